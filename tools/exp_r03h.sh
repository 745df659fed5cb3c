mkdir -p gpurun_out/r03h
L=gpurun_out/r03h
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $L/pytest_parity.log 2>&1
tail -5 $L/pytest_parity.log
export SAENA_PLAN_CACHE=$PWD/$L/plans.tsv
SAENA_SETUP_TIMING=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $L/bench1.json 2> $L/bench1.err
SAENA_SETUP_TIMING=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $L/bench2.json 2> $L/bench2.err
grep "sgpu\]" $L/bench1.err | cut -c1-330
echo ---; grep "sgpu\]" $L/bench2.err | cut -c1-200 | head -30
python - <<'P'
import json
for f in ("gpurun_out/r03h/bench1.json","gpurun_out/r03h/bench2.json"):
    d=json.load(open(f))
    print(d["value"], d["roofline"]["kernel"], d["roofline"]["us_per_launch"], d["roofline"]["frac"], d["spmv_hbm_resident"]["kernel"], d["spmv_hbm_resident"]["us_per_launch"], d["spmv_hbm_resident"]["frac"], d["vcycle"]["vcycle_ms"], d["vcycle"]["pcg_iterations_per_s"], d["vcycle"]["final_residual"], d["vcycle"]["host_setup_s"])
P
