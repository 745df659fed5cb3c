mkdir -p gpurun_out/r03j
L=gpurun_out/r03j
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plan_cache or wave_streamed or refuses_blocks" > $L/pytest_new.log 2>&1; tail -3 $L/pytest_new.log
export SAENA_SETUP_TIMING=1 SAENA_BENCH_NO_RCCL=1 SAENA_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --vcycle-timeout 900 > $L/bench_n2_323.json 2> $L/bench_n2_323.err
echo "rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r03j/bench_n2_323.json"))
v=d["vcycle_config4"]; print({k:v[k] for k in ("host_setup_s","pcg_iterations","relative_residual","vcycle_ms","residual_check")}); print(d["vcycle"]["host_setup_s"], d["vcycle"]["final_residual"])
P
grep "setup L[0-5]\]" $L/bench_n2_323.err | tail -170 | sort -s -k1,2 | awk 'NR%2==1' | tail -86
grep "autotune\|plan of" $L/bench_n2_323.err | tail -40 | cut -c1-250
