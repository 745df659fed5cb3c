mkdir -p gpurun_out/r03i
L=gpurun_out/r03i
/tmp/touch_bench > $L/touch_bench.log 2>&1 || (g++ -O2 -pthread -o /tmp/touch_bench tools/touch_bench.cpp && /tmp/touch_bench > $L/touch_bench.log 2>&1)
cat $L/touch_bench.log
nproc; free -g | head -2
export SAENA_SETUP_TIMING=1 SAENA_BENCH_NO_RCCL=1 SAENA_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --vcycle-timeout 900 > $L/bench_n2_323.json 2> $L/bench_n2_323.err
echo "rc=$?"
tail -c 1500 $L/bench_n2_323.json
grep "setup L[0-5]\]" $L/bench_n2_323.err | tail -130 | sort -s -k1,2 | awk 'NR%2==1' | tail -66
