#!/bin/bash
# HBM-side traffic of the configs[4] SpMV (bench.py's `spmv_irregular`: SiH4 x 200 blocks, 1 M rows, 34.7 M entries) from rocprofv3
# PMC counters:   bash tools/pmc_spmv_irregular.sh <variant> <lanes> <out.json>
# One pass per counter, kernel pinned (counter collection perturbs the autotune), no trace domains next to --pmc; FETCH_SIZE doubled
# on gfx950 (MI355X_MICROARCH.md).
V=${1:-16}; G=${2:-16}; OUT=${3:-gpurun_out/pmc_spmv_irregular.json}
D=gpurun_out/pmc_irr_v$V; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_BENCH_VARIANT=11 SAENA_BENCH_VARIANT_IRREGULAR=$V SAENA_BENCH_LANES_IRREGULAR=$G SAENA_KEEP_HOST_VALUES=1
i=0
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vcycle --hbm-m 0 > $D/pass$i.log 2>&1 || { tail -3 $D/pass$i.log; exit 1; }
done
python3 - "$D" "$V" "$G" > "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, v, lanes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
line = json.loads([ln for ln in open(d + "/pass1.log").read().splitlines() if ln.startswith("{")][-1])["spmv_irregular"]
rows = []
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows += list(csv.DictReader(open(f)))
# the SpMV launches of the irregular leg: the kernel (SpMV epilogue: first template argument 0) with the most launches that is not k_sellp
names = defaultdict(int)
for r in rows:
    n = r["Kernel_Name"]
    if "sk::k_" in n and "<0," in n.replace(", ", ",") and "k_sellp" not in n and "k_stream_ceiling" not in n and "k_sell_scatter" not in n:
        names[n] += 1
kernel = max(names, key=names.get)
vals = defaultdict(list)
for r in rows:
    if r["Kernel_Name"] == kernel:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: x[-50:] for k, x in vals.items()}
mean = {k: sum(x) / len(x) for k, x in c.items()}
fetch, write = 2.0 * mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0
rd, rd32 = mean["TCC_EA0_RDREQ_sum"], mean["TCC_EA0_RDREQ_32B_sum"]
print(json.dumps({"workload": line["workload"], "kernel": kernel, "bench_kernel": line["kernel"], "launches": {k: len(x) for k, x in c.items()},
                  "fetch_bytes_corrected": fetch, "write_bytes": write, "cross_check_rdreq_bytes": (rd - rd32) * 128 + rd32 * 32,
                  "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": line["algorithmic_bytes"], "stored_bytes": line["working_set_bytes"],
                  "traffic_over_algorithmic": (fetch + write) / line["algorithmic_bytes"], "traffic_over_stored": (fetch + write) / line["working_set_bytes"],
                  "note": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request) + WRITE_SIZE; counts requests that leave the L2, Infinity-Cache hits included"}, indent=1))
PY
cat "$OUT"
rm -rf $D
