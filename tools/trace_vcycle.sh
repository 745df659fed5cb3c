#!/bin/bash
# rocprofv3 kernel trace of a whole 256^3 run (setup, autotune, per-level timing loops, pCG, raw V-cycles): every kernel's
# duration inside the V-cycle, split by (kernel, grid).  bash tools/trace_vcycle.sh  ->  gpurun_out/r02x/split.csv
set -o pipefail
O=gpurun_out/r02x; mkdir -p $O
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 -m tests.perf_vcycle 256 > $O/vcycle256_traced.log 2>&1 || { tail -5 $O/vcycle256_traced.log; exit 3; }
find $O/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/trace.csv
rm -rf $O/kt
python3 tools/kernel_trace_split.py $O/trace.csv > $O/split.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02x/trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 400 launches: the raw V-cycle loop; print per kernel stats in that tail window
tail=rows[-1500:]
import collections
g=collections.defaultdict(list)
for r in tail:
    g[(r['Kernel_Name'][:70], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in g.values())
print("tail window total kernel ns", tot, "span", int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp']))
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:25]:
    v.sort(); print(k, len(v), "avg", sum(v)//len(v), "med", v[len(v)//2], "min", v[0], "max", v[-1])
PY
gzip -f $O/trace.csv
