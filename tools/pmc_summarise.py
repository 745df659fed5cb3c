"""Summarise the rocprofv3 --pmc passes of tools/pmc_spmv.sh into the JSON bench.py reads (`traffic`)."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, variant = sys.argv[1], int(sys.argv[2])
want = {3: "k_csr_cc16", 0: "k_csr_stream", 9: "k_sell<", 11: "k_sellp<", 14: "k_sellp2<"}[variant]
inst = {3: "<0, 1, 2048", 0: "<0, 1, 2048", 9: "k_sell<0, ", 11: "k_sellp<0, ", 14: "k_sellp2<0, "}[variant]     # the SpMV instantiation of the 128^3 fine level
vals = defaultdict(list)
kernel = None
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"] and inst in row["Kernel_Name"]:
            kernel = row["Kernel_Name"]
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
# the 5 warm-up launches come first: keep the last 40 (the timed steps)
c = {k: v[-40:] for k, v in vals.items()}
mean = {k: sum(v) / len(v) for k, v in c.items()}
fetch_raw = mean["FETCH_SIZE"] * 1024.0              # FETCH_SIZE / WRITE_SIZE are in KiB
write = mean["WRITE_SIZE"] * 1024.0
rd, rd32 = mean["TCC_EA0_RDREQ_sum"], mean["TCC_EA0_RDREQ_32B_sum"]
fetch = 2.0 * fetch_raw                               # gfx950: counted as 64 B per request, the requests are 128 B
out = {
    "command": f"SAENA_BENCH_VARIANT={variant} rocprofv3 --pmc <C> --output-format csv -- python3 bench.py --steps 40 --warmup 5 "
               "--no-cpu-baseline --no-vcycle (tools/pmc_spmv.sh: one pass per counter group, kernel pinned)",
    "kernel": kernel, "workload": "Poisson 128^3 SpMV, 1 MI355X",
    "counters": {k: {"n": len(v), "mean": mean[k], "min": min(v), "max": max(v)} for k, v in sorted(c.items())},
    "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected": fetch,
    "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests of a wide coalesced stream are 128 B -> doubled "
                  f"(MI355X_MICROARCH.md, HBM); cross-check: (TCC_EA0_RDREQ_sum - 32B) x 128 B + 32B x 32 B = {(rd - rd32) * 128 + rd32 * 32:.0f}",
    "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
    "algorithmic_bytes_per_launch": 206896036,
    "traffic_over_algorithmic": (fetch + write) / 206896036,
    "l2_hit_rate": mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]),
}
print(json.dumps(out, indent=1))
