"""Fine-level SpMV over operator sizes (cache regimes of the MI355X): python -m tests.perf_size_sweep [m ...]"""
import json
import subprocess
import sys

ms = [int(a) for a in sys.argv[1:]] or [48, 64, 96, 128, 160, 192, 256]
for m in ms:
    out = subprocess.run([sys.executable, "bench.py", "--m", str(m), "--no-vcycle", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(f"m={m:4d} rows={d['config']['rows_per_gpu']:9d} algorithmic {r['algorithmic_bytes'] / 1e6:8.1f} MB  {r['kernel']:34s} "
          f"{r['us_per_launch']:8.2f} us  {r['achieved']:7.0f} GB/s  {100 * r['frac']:5.1f} % of 8 TB/s", flush=True)
