"""BASELINE.json configs[4] at a size where the memory system, not the launch floor, is what is measured.

The reference's config names a SuiteSparse matrix from data/florida_matrices.txt ("irregular nnz/row ... stresses load-balance of
wavefront CSR"); the ones in its checkout are 362-5041 rows -- 3-5 us of launch latency on this chip.  This generator scales the
most irregular of them, SiH4 (5 041 rows, 171 903 entries, 1-205 per row; tests/golden/matrices/SiH4.mtx.gz, a fixture the
reference ships under data/FloridaCollection), to >= 1 M rows WITHOUT making it regular:
  * `nblocks` diagonal blocks, each the matrix under its own symmetric row/column permutation (block 0 keeps the original
    order): the row-length sequence and the column scatter differ from block to block;
  * symmetric coupling entries between neighbouring blocks (n/8 pairs per block boundary), so the operator is one connected
    system with columns outside the row's own block;
  * every `hub_every`-th block carries a hub row of `hub_len` entries spread over its neighbouring blocks (and the mirrored
    column): rows longer than an LDS tile, the long-row path of the kernels.
Deterministic (seeded), diagonal kept (Jacobi needs it), symmetric.  Used by tests/test_gpu_parity.py at 3 blocks against the
oracle and by bench.py (`spmv_irregular`) at 200 blocks: 1 008 200 rows, ~34.6 M entries."""
import numpy as np

from tests import matrices


def sih4_replicated(nblocks, seed=20261005, hub_every=16, hub_len=3000):
    """-> (rows, cols, vals, M): coordinate entries (int32, int32, float64), no duplicates"""
    base, n = matrices.entries("SiH4")
    br, bc, bv = base["row"].astype(np.int64), base["col"].astype(np.int64), base["val"].astype(np.float64)
    rng = np.random.default_rng(seed)
    M = n * nblocks
    R, Cc, V = [], [], []            # the blocks: no two of these entries coincide
    XR, XC, XV = [], [], []          # coupling and hub entries: all BETWEEN blocks, so they can only coincide with one another
    for b in range(nblocks):
        p = np.arange(n) if b == 0 else rng.permutation(n)
        R.append(b * n + p[br]); Cc.append(b * n + p[bc]); V.append(bv)
        if b + 1 < nblocks:                                          # coupling to the next block, mirrored
            k = n // 8
            i = b * n + rng.choice(n, size=k, replace=False)
            j = (b + 1) * n + rng.choice(n, size=k, replace=False)
            w = 0.01 * (0.5 + rng.random(k))
            XR += [i, j]; XC += [j, i]; XV += [w, w]
        if hub_every and nblocks >= 3 and b % hub_every == (hub_every // 2 if nblocks > hub_every // 2 else nblocks // 2):
            nb = [q for q in (b - 1, b + 1) if 0 <= q < nblocks]      # columns in the neighbouring blocks only
            pool = np.concatenate([np.arange(q * n, (q + 1) * n) for q in nb])
            hub = b * n + int(rng.integers(n))
            cols = rng.choice(pool, size=min(hub_len, len(pool)), replace=False)
            w = 1e-3 * (0.5 + rng.random(len(cols)))
            XR += [np.full(len(cols), hub), cols]; XC += [cols, np.full(len(cols), hub)]; XV += [w, w]
    if XR:
        xr, xc, xv = np.concatenate(XR), np.concatenate(XC), np.concatenate(XV)
        _, first = np.unique(xr * M + xc, return_index=True)         # a hub entry may fall on a coupling entry: keep the first
        first.sort()
        R.append(xr[first]); Cc.append(xc[first]); V.append(xv[first])
    return np.concatenate(R).astype(np.int32), np.concatenate(Cc).astype(np.int32), np.concatenate(V), M


def row_length_stats(rows, M):
    ln = np.bincount(rows, minlength=M)
    q = np.percentile(ln, [50, 90, 99])
    return {"min": int(ln.min()), "mean": round(float(ln.mean()), 2), "median": int(q[0]), "p90": int(q[1]), "p99": int(q[2]), "max": int(ln.max()),
            "coefficient_of_variation": round(float(ln.std() / ln.mean()), 3)}
