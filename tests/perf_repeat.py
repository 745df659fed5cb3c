"""Run-to-run spread of an operator's kernels (development aid): python -m tests.perf_repeat m level[,level...] v1,v2,... trials [lanes,...]
Times each variant `trials` times, interleaved, with fresh input/output vectors every second trial (A of each level:
Jacobi sweep; PERF_REPEAT_WHICH=1 / 2: P / R of each level, plain product)."""
import os
import sys

import numpy as np

if not os.environ.get("PERF_REPEAT_NOKEEP"):          # (NOKEEP: the library frees the losing plans after its autotune, as in production)
    os.environ.setdefault("SAENA_KEEP_HOST_VALUES", "1")
from saena_amd import capi, host


def main():
    m, levels = int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")]
    variants = [int(v) for v in sys.argv[3].split(",")]
    trials = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    lanes_list = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
    capi.init(0)
    print("device:", capi.device_info(), flush=True)
    L = host.load("gpu")
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001)).to_device()
    for level in levels:
        which = int(os.environ.get("PERF_REPEAT_WHICH", "0"))
        op = S.device_op(level, which)
        kind = 1 if which == 0 else 0
        print(f"L{level}: autotune chose {op.variant()} lanes {op.info()['lanes_per_row']}", flush=True)
        tuned = op.info()["lanes_per_row"]
        x = y = rhs = None
        for t in range(trials):
            if t % 2 == 0:
                x, y, rhs = capi.DeviceVector(op.N_local, np.ones(op.N_local)), capi.DeviceVector(op.M), capi.DeviceVector(op.M, np.ones(op.M))
            cells = []
            for v in variants:
                for lanes in lanes_list:
                    try:
                        op.set_variant(v)
                    except capi.SgpuError:
                        cells.append(f"v{v} refused")
                        break
                    op.set_lanes_per_row(lanes or tuned)
                    op.time_kernel(kind, x, rhs, y, 3)
                    cells.append(f"v{v} G{lanes or tuned} {op.time_kernel(kind, x, rhs, y, 30) * 1e3:7.1f} us")
            print(f"trial {t}: " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
