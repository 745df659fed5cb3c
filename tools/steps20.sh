#!/bin/bash
# bench.py at the driver's --steps 20 --warmup 5 (GPU box, repo root): the fixed latency next to 20 kernels, with and without
# spinning host waits; then the whole default line at those flags.  bash tools/steps20.sh
O=gpurun_out/r02aa; mkdir -p $O
for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-vcycle --no-cpu-baseline --hbm-m 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin   ', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['kernel'])" || exit 1; done
for i in 1 2 3; do SAENA_NO_SPIN_WAIT=1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-vcycle --no-cpu-baseline --hbm-m 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no spin', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['kernel'])" || exit 1; done
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2>$O/err.log || exit 2
python3 -c "import json; d=json.loads(open('$O/bench_steps20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['vcycle']['pcg_iterations_per_s'], d['vcycle']['vcycle_ms'])"
