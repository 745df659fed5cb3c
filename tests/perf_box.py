"""Kernel sweep on synthetic 3D box-stencil operators (development aid): python -m tests.perf_box [r:n ...]

A radius-r box stencil on an n^3 grid has (2r+1)^3 entries per interior row -- r = 1/2/3/5/7 gives 27/125/343/1331/3375
nnz per row, the row shapes of the smoothed-aggregation levels of the Poisson hierarchies (67, 263, 1391, 2786 nnz/row)
-- and builds in seconds, where the 256^3 hierarchy takes minutes.  Sizes are chosen beyond the 256 MiB Infinity
Cache so the numbers are HBM-bound.  Prints every (variant, lanes) time, like tests.perf_levels.
"""
import sys
import time

import numpy as np

from saena_amd import capi

VARIANTS = {0: "s16K", 1: "s32K", 2: "vec", 3: "cc16K", 4: "cc32K", 6: "wave", 7: "cm16K", 8: "cm32K", 9: "sell", 10: "xlds"}


def box_csr(n, r):
    """CSR (row lengths, columns, values) of the clipped radius-r box stencil on an n^3 grid, natural ordering"""
    idx = np.arange(n)
    lo, hi = np.maximum(idx - r, 0), np.minimum(idx + r, n - 1)
    cnt1 = (hi - lo + 1)
    # per row (k, j, i): columns = all (kk, jj, ii) in the clipped box, ascending
    K, J, I = np.meshgrid(idx, idx, idx, indexing="ij")
    npr = (cnt1[K] * cnt1[J] * cnt1[I]).ravel().astype(np.int32)
    cols = np.empty(int(npr.sum()), np.int32)
    pos = 0
    w = 2 * r + 1
    off = np.arange(-r, r + 1)
    for k in range(n):                      # one z-plane of rows at a time (vectorised inside)
        kk = off + k
        kk = kk[(kk >= 0) & (kk < n)]
        for j in range(n):
            jj = off + j
            jj = jj[(jj >= 0) & (jj < n)]
            base = (kk[:, None] * n + jj[None, :]).ravel() * n      # start of every (kk, jj) line
            for i in range(n):
                l0, h0 = lo[i], hi[i]
                seg = (base[:, None] + np.arange(l0, h0 + 1)[None, :]).ravel()
                cols[pos:pos + len(seg)] = seg
                pos += len(seg)
    assert pos == len(cols)
    vals = -1.0 / (1.0 + (np.arange(len(cols)) % 7))
    return npr, cols, vals


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
                res.setdefault((v, g), []).append(op.time_kernel(kind, x, rhs, y, reps) * 1e3)
    best = min(res, key=lambda k: min(res[k]))
    cells = [f"{VARIANTS[v]}: " + " ".join(f"G{g}={min(res[(v, g)]):.1f}" for g in lanes_list) for v in VARIANTS]
    print(f"{name}: " + " | ".join(cells), flush=True)
    op.set_variant(best[0]); op.set_lanes_per_row(best[1])
    print(f"   best {op.variant()[1]} G={best[1]}: {min(res[best]):.1f} us = {B / min(res[best]) / 1e3:.0f} GB/s", flush=True)
    return best, min(res[best])


def main():
    cases = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(3, 56), (5, 36), (7, 30)]
    capi.init(0)
    for r, n in cases:
        t0 = time.time()
        npr, cols, vals = box_csr(n, r)
        M = n ** 3
        inv = 1.0 / (1.0 + np.arange(M) % 3)
        op = capi.Operator(M=M, N_local=M, col_offset=0, nnzPerRow_local=npr, col_local=cols, val_local=vals, inv_diag=inv)
        avg = len(cols) / M
        print(f"--- box r={r} n={n}: rows {M} nnz {len(cols)} ({avg:.0f}/row, {12 * len(cols) / 1e6:.0f} MB) built in {time.time() - t0:.1f}s", flush=True)
        x, y, rhs = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        cand = [g for g in (8, 16, 32, 64) if avg / 32 <= g <= max(1, avg)] or [64]
        sweep(f"box{r} jacobi", op, 1, x, rhs, y, cand, 20)
        op.destroy()


if __name__ == "__main__":
    main()
