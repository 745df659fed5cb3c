mkdir -p gpurun_out/r03b
for ns in 1 2 4; do
  echo "== NS=$ns" >> gpurun_out/r03b/perf_ns.log
  SAENA_SELLP_NS=$ns timeout -k 10 120 python -m tests.perf_fine 128,256 9,11 3 >> gpurun_out/r03b/perf_ns.log 2>&1
done
echo "== grid shape" >> gpurun_out/r03b/perf_ns.log
cat gpurun_out/r03b/perf_ns.log | grep -v device
