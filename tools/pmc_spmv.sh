#!/bin/bash
# HBM traffic of the fine-level SpMV kernel from rocprofv3 PMC counters (run on the GPU box from the repo root):
#   bash tools/pmc_spmv.sh <variant> <out.json>      variant: 11 = k_sellp, 9 = k_sell, 3 = k_csr_cc16<16KiB>, 0 = k_csr_stream<16KiB>
# One pass per counter, kernel pinned (PMC collection perturbs the autotune's timings), no trace domains
# next to --pmc.  tools/pmc_summarise.py applies the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md).
set -e
V=${1:-9}; OUT=${2:-gpurun_out/pmc_spmv.json}
D=gpurun_out/pmc_v$V; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_BENCH_VARIANT=$V
i=0
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum; do   # one counter per pass: pairs exceed the hardware
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-vcycle --hbm-m 0 > $D/pass$i.log 2>&1
done
python3 tools/pmc_summarise.py $D $V > $OUT
cat $OUT
