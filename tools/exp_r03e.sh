mkdir -p gpurun_out/r03e
for nt in 0 1; do
  echo "== NT=$nt" >> gpurun_out/r03e/perf_nt.log
  SAENA_SELLP_NT=$nt timeout -k 10 120 python -m tests.perf_fine 128,256 9,11 3 >> gpurun_out/r03e/perf_nt.log 2>&1
done
grep -v device gpurun_out/r03e/perf_nt.log
