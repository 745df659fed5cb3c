#!/bin/bash
# Vector-L1 (TCP) behaviour of ONE operator of a Poisson hierarchy from rocprofv3 PMC counters (GPU box, repo root):
#   bash tools/pmc_l1.sh <m> <level> <which: 0 A, 1 P, 2 R> <kind: 0 spmv, 1 jacobi> <variant> <lanes> <out.txt>
# One pass per counter; prints, for the operator's kernel (the k_csr_* kernel with the most launches), the mean per launch.
set -e
M=${1:-128}; LV=${2:-2}; WH=${3:-0}; KIND=${4:-1}; V=${5:-4}; G=${6:-16}; OUT=${7:-gpurun_out/pmc_l1.txt}
D=gpurun_out/pmc_l1_L$LV; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_NO_AUTOTUNE=1 SAENA_KEEP_HOST_VALUES=1     # (the re-ordered forms are built from the host copy of the values)
i=0
for C in ${PMC_L1_COUNTERS:-TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum}; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 -m tests.perf_one $M $LV $WH $KIND 20 $V $G > $D/pass$i.log 2>&1 || true
done
python3 - "$D" "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows += list(csv.DictReader(open(f)))
names = defaultdict(int)
for r in rows:
    if "sk::k_csr" in r["Kernel_Name"] or "sk::k_sell" in r["Kernel_Name"]: names[r["Kernel_Name"]] += 1
kernel = max(names, key=names.get)
vals = defaultdict(list)
for r in rows:
    if r["Kernel_Name"] == kernel: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
line = [ln for ln in open(d + "/pass1.log").read().splitlines() if "which=" in ln]
with open(out, "w") as f:
    f.write((line[-1] if line else "") + "\n" + kernel + "\n")
    for k, v in sorted(vals.items()):
        v = v[-20:]
        f.write(f"{k:40s} n={len(v):3d} mean={sum(v) / len(v):.4g}\n")
print(open(out).read())
PY
