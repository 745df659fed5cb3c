"""Isolate the cost of the x gather on the L1 operator (development aid):
same row structure and values, columns replaced by (a) all zero (broadcast gather), (b) a dense band
(i + k: perfectly local), (c) the real columns.  python -m tests.perf_gather_probe [m]"""
import sys

import numpy as np

from saena_amd import capi, host


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    capi.init(0)
    L = host.load("gpu")
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
    for lvl in (1, 2):
        lay = S.level_layout(lvl, 0)
        M = lay["M"]
        npr = lay["nnzPerRow_local"]
        rows = np.repeat(np.arange(M, dtype=np.int64), npr)
        ptr = np.concatenate([[0], np.cumsum(npr)])
        k_in_row = np.arange(len(rows)) - ptr[rows]
        variants = {
            "real columns": lay["col_local"],
            "all zero (broadcast)": np.zeros_like(lay["col_local"]),
            "dense band i+k": np.minimum(rows + k_in_row, M - 1).astype(np.int32),
            "sorted random": None,
        }
        rng = np.random.default_rng(1)
        rnd = rng.integers(0, M, size=len(rows)).astype(np.int32)
        # sort within rows
        order = np.lexsort((rnd, rows))
        variants["sorted random"] = rnd[order]
        x, y, rhs = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        opr = S.to_device().device_op(lvl, 0) if lvl == 1 else S.device_op(lvl, 0)
        for mode in (0, 1):
            opr.debug_gather_probe(mode, x, 3)
            print(f"L{lvl} gather only, mode {mode} ({'4 consecutive nnz per lane' if mode == 0 else '64 consecutive nnz per instruction'}): "
                  f"{opr.debug_gather_probe(mode, x, 20) * 1e3:.1f} us", flush=True)
        for name, col in variants.items():
            op = capi.Operator(M=M, N_local=M, col_offset=0, nnzPerRow_local=npr, col_local=col, val_local=lay["val_local"],
                               inv_diag=lay["inv_diag"])
            B = op.algorithmic_bytes(1)
            line = f"L{lvl} {name:22s}:"
            for v, g in ((0, 4), (0, 8), (1, 8), (1, 16)):
                op.set_variant(v); op.set_lanes_per_row(g)
                op.time_kernel(1, x, rhs, y, 3)
                us = min(op.time_kernel(1, x, rhs, y, 20) for _ in range(2)) * 1e3
                line += f"  v{v}G{g}: {us:7.1f} us ({B / us / 1e3:5.0f} GB/s)"
            print(line, flush=True)
            op.destroy()


if __name__ == "__main__":
    main()
