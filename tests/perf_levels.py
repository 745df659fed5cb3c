"""Per-level kernel variant sweep (development aid): python -m tests.perf_levels [m] [maxlevel]

For every operator of the 128^3 hierarchy: time each (variant, lanes) pair, interleaved
in one process (two rounds), print the table and the winner.
"""
import sys
import time

import numpy as np

from saena_amd import capi, host

VARIANTS = {0: "s16K", 1: "s32K", 2: "vec", 3: "cc16K", 4: "cc32K", 6: "wave", 7: "cm16K", 8: "cm32K", 9: "sell", 10: "xlds"}


def sweep(name, op, kind, x, rhs, y, lanes_list, reps):
    B = op.algorithmic_bytes(kind)
    res = {}
    for rnd in range(2):
        for v in VARIANTS:
            for g in lanes_list:
                try:
                    op.set_variant(v)
                except capi.SgpuError:
                    res.setdefault((v, g), []).append(1e9)
                    continue
                op.set_lanes_per_row(g)
                op.time_kernel(kind, x, rhs, y, 3)
                us = op.time_kernel(kind, x, rhs, y, reps) * 1e3
                res.setdefault((v, g), []).append(us)
    best = min(res, key=lambda k: min(res[k]))
    cells = []
    for v in VARIANTS:
        cells.append(f"{VARIANTS[v]}: " + " ".join(f"G{g}={min(res[(v, g)]):.1f}" for g in lanes_list))
    print(f"{name}: " + " | ".join(cells), flush=True)
    print(f"   best {VARIANTS[best[0]]} G={best[1]}: {min(res[best]):.1f} us = {B / min(res[best]) / 1e3:.0f} GB/s", flush=True)
    op.set_variant(best[0]); op.set_lanes_per_row(best[1])
    return best, min(res[best])


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    capi.init(0)
    L = host.load("gpu")
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    t0 = time.time()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001)).to_device()
    print(f"setup+upload {time.time() - t0:.1f}s", flush=True)
    for l in range(min(maxl, S.num_levels)):
        info = S.level_info(l)
        avg = info["nnzA"] / info["rows"]
        opA = S.device_op(l, 0)
        M = opA.M
        x, y, rhs = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        cand = [g for g in (1, 2, 4, 8, 16, 32, 64) if avg / 16 <= g <= max(1, avg)] or [64]
        reps = 30 if info["nnzA"] > 5e6 else 100
        print(f"--- L{l}: rows {info['rows']} nnz {info['nnzA']} ({avg:.1f}/row)", flush=True)
        sweep(f"L{l} A jacobi", opA, 1, x, rhs, y, cand, reps)
        if l < S.num_levels - 1:
            opP, opR = S.device_op(l, 1), S.device_op(l, 2)
            Mc = opR.M
            xc, yc = capi.DeviceVector(Mc, np.ones(Mc)), capi.DeviceVector(Mc)
            ap, ar = opP.info()["nnz_local"] / M, opR.info()["nnz_local"] / Mc
            cp = [g for g in (1, 2, 4, 8, 16, 32, 64) if ap / 16 <= g <= max(1, ap)] or [64]
            cr = [g for g in (1, 2, 4, 8, 16, 32, 64) if ar / 16 <= g <= max(1, ar)] or [64]
            sweep(f"L{l} P ({ap:.1f}/row)", opP, 0, xc, None, y, cp, reps)
            sweep(f"L{l} R ({ar:.1f}/row)", opR, 0, x, None, yc, cr, reps)


if __name__ == "__main__":
    main()
