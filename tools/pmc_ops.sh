#!/bin/bash
# Counters of SEVERAL operators of one Poisson hierarchy, one rocprofv3 --pmc pass per counter group, the hierarchy built once per
# pass (GPU box, repo root):
#   bash tools/pmc_ops.sh <m> "<level,which,kind,variant,lanes;...>" <out.txt> [reps]
# Round 4: the transfers R1 / R2 / P1 of 256^3 -- the kernels furthest below their roofline -- had timings only; this collects what
# shows WHERE they wait: issue / wait cycles of the waves (SQ), the vector-memory unit's busy and stall cycles (TA, TCP), requests to
# the L2 and its hits.  Counter groups: the eight SQ counters share a pass (8 SQ slots), the others get one each.
set -e
M=${1:-128}; OPS=${2:-"1,2,0,-1,0"}; OUT=${3:-gpurun_out/pmc_ops.txt}; REPS=${4:-20}
D=gpurun_out/pmc_ops_$$; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_NO_AUTOTUNE=1 SAENA_KEEP_HOST_VALUES=1 SAENA_PLAN_CACHE=off
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
         "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
         "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum"
         "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE")
i=0
for C in "${GROUPS_[@]}"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 -m tests.perf_ops $M "$OPS" $REPS > $D/pass$i.log 2>&1 || echo "pass $i ($C) failed" >> $D/failed.txt
    echo "pass $i: $C done"
done
python3 - "$D" "$OUT" "$REPS" <<'PY'
import csv, glob, sys
from collections import defaultdict
d, out, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
labels = [ln for ln in open(d + "/pass1.log").read().splitlines() if ln.startswith("OP ")]
res = [defaultdict(float) for _ in labels]
kern = [""] * len(labels)
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f))]
    by_counter = defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(r)
    for cname, rs in by_counter.items():
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        # runs of 3 + reps identical (kernel, grid) dispatches of the library's kernels, in order = the operators in order
        runs, cur = [], []
        for r in rs:
            key = (r["Kernel_Name"], r["Grid_Size"])
            if cur and (cur[0]["Kernel_Name"], cur[0]["Grid_Size"]) == key:
                cur.append(r)
            else:
                if len(cur) == 3 + reps and "sk::" in cur[0]["Kernel_Name"]:
                    runs.append(cur)
                cur = [r]
        if len(cur) == 3 + reps and "sk::" in cur[0]["Kernel_Name"]:
            runs.append(cur)
        if len(runs) != len(labels):
            print(f"{cname}: {len(runs)} runs for {len(labels)} operators -- skipped")
            continue
        for k, run in enumerate(runs):
            v = [float(r["Counter_Value"]) for r in run[3:]]
            res[k][cname] = sum(v) / len(v)
            kern[k] = run[0]["Kernel_Name"][:110] + f" grid {run[0]['Grid_Size']}"
with open(out, "w") as f:
    f.write("rocprofv3 --pmc passes (one per counter group) of `python -m tests.perf_ops`; mean per launch over the timed launches\n")
    for k, lab in enumerate(labels):
        f.write("\n" + lab + "\n  " + kern[k] + "\n")
        for c, v in sorted(res[k].items()):
            f.write(f"  {c:40s} {v:.6g}\n")
        r = res[k]
        if r.get("SQ_WAVE_CYCLES"):
            f.write(f"  -> of the waves' cycles: waiting {100 * r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES']:.1f} %, issue-stalled {100 * r['SQ_WAIT_INST_ANY'] / r['SQ_WAVE_CYCLES']:.1f} %, "
                    f"issuing {100 * r['SQ_ACTIVE_INST_ANY'] / r['SQ_WAVE_CYCLES']:.1f} % (vector memory {100 * r['SQ_ACTIVE_INST_VMEM'] / r['SQ_WAVE_CYCLES']:.1f} %, LDS {100 * r['SQ_ACTIVE_INST_LDS'] / r['SQ_WAVE_CYCLES']:.1f} %)\n")
        if r.get("TCP_TCC_READ_REQ_sum") and r.get("TCC_HIT_sum") is not None:
            f.write(f"  -> L1->L2 read requests {r['TCP_TCC_READ_REQ_sum']:.4g} per launch, L2 hit rate {100 * r['TCC_HIT_sum'] / max(1.0, r['TCC_HIT_sum'] + r['TCC_MISS_sum']):.1f} %, "
                    f"fetched from memory {2 * 1024 * r.get('FETCH_SIZE', 0) / 1e6:.1f} MB (FETCH_SIZE x 2 on gfx950)\n")
print(open(out).read())
PY
