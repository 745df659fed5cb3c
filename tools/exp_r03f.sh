mkdir -p gpurun_out/r03f
L=gpurun_out/r03f
timeout -k 10 120 python -m tests.perf_fine 128,256 9,11 3 > $L/perf_fine_auto.log 2>&1
grep -v device $L/perf_fine_auto.log
for nt in 0 1; do
  echo "== SELL_NT=$nt (128^3 L1, v9 vs v3)" >> $L/sell_nt.log
  SAENA_SELL_NT=$nt timeout -k 10 200 python -m tests.perf_repeat 128 1 9,3 4 >> $L/sell_nt.log 2>&1
done
grep -v device $L/sell_nt.log
SAENA_SETUP_TIMING=1 timeout -k 10 300 python -m tests.perf_vcycle 128 > $L/vcycle128.log 2> $L/vcycle128.err
tail -25 $L/vcycle128.log
