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
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
# (deselected: the two tests whose spawned child makes the library THROW -- a C++ exception through the preloaded ASan runtime's
#  interceptors of a Python child ends the child without a report: the interceptor's own check, not a finding; round 3 saw the same)
LD_PRELOAD=$ASAN python -m pytest ${@:-tests/test_shm_comm.py tests/test_host_layout.py tests/test_amg_setup.py tests/test_sa_pins.py tests/test_aggregation_rounds.py} -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_shm_comm.py::test_a_missing_rank_is_a_timeout_not_a_hang \
    --deselect tests/test_shm_comm.py::test_an_exchange_beyond_the_room_in_shared_memory_is_an_error_not_a_signal
