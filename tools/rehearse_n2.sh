#!/bin/bash
# `bench.py --gpus 2` exactly as the driver launches it, both ranks on this one card (host transport in place of RCCL), at the
# full per-rank size of configs[3] (Poisson 323^3: 16.5 M rows per rank), with the setup's phase times:
#   bash tools/rehearse_n2.sh [outdir]
O=${1:-gpurun_out/rehearse_n2}; mkdir -p $O
export SAENA_SETUP_TIMING=1 SAENA_BENCH_NO_RCCL=1 SAENA_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
S=$(date +%s)
timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --vcycle-timeout 900 > $O/bench_n2_323.json 2> $O/bench_n2_323.err
echo "rc=$? wall=$(( $(date +%s) - S )) s"
python3 - "$O" <<'P'
import json, re, sys, collections
O = sys.argv[1]
d = json.load(open(O + "/bench_n2_323.json"))
v = d["vcycle_config4"]
print({k: v[k] for k in ("host_setup_s", "pcg_iterations", "relative_residual", "vcycle_ms")}, v["residual_check"]["ok"], "| 128^3 leg:", d["vcycle"]["host_setup_s"], d["vcycle"]["final_residual"])
t = collections.defaultdict(float)
lines = [ln for ln in open(O + "/bench_n2_323.err") if ln.startswith("[setup L")]
for ln in lines[len(lines) // 2 if False else 0:]:
    m = re.match(r"\[setup L(\d+)\] (.*?)\s+([0-9.]+) s", ln)
    if m:
        t[m.group(2).strip()] += float(m.group(3))
for k, x in sorted(t.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {x / 2:7.2f} s per rank (both legs)")
P
