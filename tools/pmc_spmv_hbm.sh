#!/bin/bash
# HBM-side traffic of the HBM-resident SpMV (Poisson 256^3, the bench line's `spmv_hbm_resident`) from rocprofv3 PMC counters:
#   bash tools/pmc_spmv_hbm.sh <variant: 3 k_csr_cc16, 9 k_sell, 11 k_sellp> <out.json>
# One pass per counter, kernel pinned, no trace domains next to --pmc; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md).
V=${1:-3}; OUT=${2:-gpurun_out/pmc_spmv_hbm.json}
D=gpurun_out/pmc_hbm_v$V; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd "$OLDPWD"
export SAENA_BENCH_VARIANT=$V
i=0
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D/pass$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vcycle > $D/pass$i.log 2>&1 || { tail -3 $D/pass$i.log; exit 1; }
done
python3 - "$D" "$V" > "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, v = sys.argv[1], int(sys.argv[2])
rows = []
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows += list(csv.DictReader(open(f)))
want = {3: "k_csr_cc16<0,", 9: "k_sell<0,", 11: "k_sellp<0,", 14: "k_sellp2<0,"}[v]
g = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].replace(", ", ",")
    if want in name:
        wg = int(r["Grid_Size"]) // int(r["Workgroup_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        g[wg][r["Counter_Name"]].append(float(r["Counter_Value"]))
wg = max(g)                                   # the 256^3 operator: the largest grid of this instantiation
c = {k: x[-50:] for k, x in g[wg].items()}
mean = {k: sum(x) / len(x) for k, x in c.items()}
fetch = 2.0 * mean["FETCH_SIZE"] * 1024.0
write = mean["WRITE_SIZE"] * 1024.0
rd, rd32 = mean["TCC_EA0_RDREQ_sum"], mean["TCC_EA0_RDREQ_32B_sum"]
nnz, M = 114322352, 16387064
alg = 12 * nnz + 4 * (M + 1) + 16 * M
stored = {3: 10 * nnz + 4 * (M + 1) + 16 * M, 9: 10 * nnz + 2 * M + 16 * M, 11: 8 * nnz + 2 * M + 16 * M, 14: 8 * nnz + 2 * M + 16 * M}[v]      # values + column form + x + y
print(json.dumps({"workload": "Poisson 256^3 SpMV (16387064 rows, 114322352 nnz), 1 MI355X", "kernel_filter": want, "workgroups": wg,
                  "launches": {k: len(x) for k, x in c.items()}, "fetch_bytes_corrected": fetch, "write_bytes": write,
                  "cross_check_rdreq_bytes": (rd - rd32) * 128 + rd32 * 32,
                  "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg, "stored_bytes": stored,
                  "traffic_over_algorithmic": (fetch + write) / alg, "traffic_over_stored": (fetch + write) / stored,
                  "note": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request) + WRITE_SIZE; counts requests that leave the L2, "
                          "Infinity-Cache hits included"}, indent=1))
PY
cat "$OUT"
rm -rf $D
