#!/bin/bash
# The round's committed measurements, one GPU-box call (run from the repo root): bash tools/round_measure.sh [outdir]
#   bench line, rocprofv3 --kernel-trace --stats of the same command (+ its split by operator), per-level V-cycle tables.
# PMC traffic of the bench kernel: bash tools/pmc_spmv.sh 9 <out.json> (separate passes, no trace domains next to --pmc).
set -o pipefail
O=${1:-gpurun_out/round}; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 2; }
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline > $O/bench_n1_under_rocprof.json 2> $O/bench_n1_under_rocprof.err || { tail -5 $O/bench_n1_under_rocprof.err; exit 3; }
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_n1_kernel_stats.csv
find $O/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/kernel_trace_split.py {} > $O/bench_n1_kernel_stats_by_grid.csv
rm -rf $O/kt
timeout -k 10 300 python -m tests.perf_vcycle 128 > $O/vcycle128_levels.log 2>&1 || exit 4
timeout -k 10 600 python -m tests.perf_vcycle 256 > $O/vcycle256_levels.log 2>&1 || exit 5
cat $O/bench_n1.json
head -6 $O/bench_n1_kernel_stats_by_grid.csv
for f in 128 256; do cut -c1-230 $O/vcycle${f}_levels.log | grep "^L[0-5]\|sum of\|raw V\|solve_pCG\|setup"; done
