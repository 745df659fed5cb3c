#!/bin/bash
# AddressSanitizer + UBSan run of the host library (CPU build; GPU sanitizers are not available on this pool):
#   bash tools/sanitize_host.sh [pytest args]      default: the host-side test files
# Builds libsaena_host.so with -fsanitize=address,undefined into a scratch directory, points the Python bindings at it
# (SAENA_HOST_LIB) and runs the CPU tests under LD_PRELOAD=libasan.so.  The normal build is untouched.
set -e
cd "$(dirname "$0")/.."
OUT=${SAN_OUT:-/tmp/saena_san}; mkdir -p $OUT
SRC=saena_amd/csrc/host
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared -o $OUT/libsaena_host.so $SRC/*.cpp -lpthread -lrt -ldl \
    -Wl,--version-script=$SRC/exports.map
ASAN=$(g++ -print-file-name=libasan.so)
export SAENA_HOST_LIB=$OUT/libsaena_host.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$ASAN python -m pytest ${@:-tests/test_shm_comm.py tests/test_host_layout.py tests/test_amg_setup.py tests/test_sa_pins.py tests/test_aggregation_rounds.py} -x -q -m "not gpu" -p no:cacheprovider
