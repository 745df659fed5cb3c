#!/bin/bash
# HBM traffic of ONE operator of the 128^3 hierarchy from rocprofv3 PMC counters (GPU box, repo root):
#   bash tools/pmc_level.sh <level> <kind: 0 spmv, 1 jacobi> <variant> <lanes> <out.json>
# One pass per counter; the launches of tests.perf_one's timed loop (the last 40 of that kernel) are averaged.
set -e
LV=${1:-1}; KIND=${2:-1}; V=${3:-4}; G=${4:-4}; OUT=${5:-gpurun_out/pmc_level.json}
D=gpurun_out/pmc_L$LV; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_KEEP_HOST_VALUES=1      # the variant is switched after the plan-time autotune
i=0
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 -m tests.perf_one 128 $LV 0 $KIND 40 $V $G > $D/pass$i.log 2>&1
done
grep "which=" $D/pass1.log | tail -1
python3 - "$D" "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
line = [ln for ln in open(d + "/pass1.log").read().splitlines() if "which=" in ln][-1]
vals, names = defaultdict(list), defaultdict(int)
rows = []
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    if "k_csr" in r["Kernel_Name"] or "k_sell" in r["Kernel_Name"]:
        names[r["Kernel_Name"]] += 1
# the timed kernel = the k_csr_* kernel with the most launches in a pass (3 warm-up + 40 timed of the same instantiation)
kernel = max(names, key=names.get)
for r in rows:
    if r["Kernel_Name"] == kernel:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: v[-40:] for k, v in vals.items()}
mean = {k: sum(v) / len(v) for k, v in c.items()}
fetch_raw, write = mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0
rd, rd32 = mean["TCC_EA0_RDREQ_sum"], mean["TCC_EA0_RDREQ_32B_sum"]
res = {"command": "tools/pmc_level.sh: rocprofv3 --pmc <C> -- python3 -m tests.perf_one 128 <level> 0 <kind> 40 <variant> <lanes>, one pass per counter",
       "perf_one": line, "kernel": kernel,
       "counters": {k: {"n": len(v), "mean": mean[k], "min": min(v), "max": max(v)} for k, v in sorted(c.items())},
       "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected": 2.0 * fetch_raw,
       "correction": "gfx950: FETCH_SIZE counts 64 B per request, the requests of a coalesced stream are 128 B -> doubled; cross-check "
                     f"(RDREQ - 32B) x 128 B + 32B x 32 B = {(rd - rd32) * 128 + rd32 * 32:.0f}",
       "write_bytes": write, "traffic_bytes_per_launch": 2.0 * fetch_raw + write,
       "l2_hit_rate": mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "traffic_bytes_per_launch", "l2_hit_rate")}))
PY
