"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) duration statistics.

`rocprofv3 --stats` averages every launch of a kernel NAME; bench.py launches its SpMV kernel on two operators
(Poisson 128^3 = the timed steps, Poisson 256^3 = the HBM-resident figure), which share one instantiation.  This splits
the trace by grid size so that each operator's average can be held against the bench line:
    python tools/kernel_trace_split.py <kernel_trace.csv> [name substring] > profiles/rNN_bench_n1_kernel_stats_by_grid.csv
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else "sk::k_"
g = collections.defaultdict(list)
for r in rows:
    if want in r["Kernel_Name"]:
        g[(r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Workgroups", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs"])
for (name, wg), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    w.writerow([name, wg, len(v), round(sum(v) / len(v), 1), v[len(v) // 2], v[0], v[-1]])
