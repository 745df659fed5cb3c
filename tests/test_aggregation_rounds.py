"""The setup's aggregation (host/amg_setup.cpp: aggregate(), and its row-distributed form) against the reference's rounds
restated the plain way -- every undecided row scans ALL its strong neighbours in EVERY round (aggregation_1_dist,
src/saena_object_setup1.cpp:724-995; strength: src/strength_matrix.cpp:233-453, setup1:520-719).  The product evaluates a row
again only when the one row it waits for changes state and looks at nothing but the first eligible column below the diagonal
(DESIGN.md 5); on random irregular graphs -- uneven degrees, weights over four decades, so that the strength relation is far
from symmetric in which of its two tests fires -- the aggregates must be the same row for row, at one rank and row-distributed."""
import numpy as np
import pytest

from saena_amd import host


def random_spd_graph(n, deg, seed):
    """symmetric M-matrix of a random graph: off-diagonals -w_ij (w over four decades), diagonal = sum of the row's weights + 1"""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), deg)
    c = rng.integers(0, n, size=n * deg)
    # a few long-range edges, mostly near neighbours (so that aggregates form)
    near = rng.random(n * deg) < 0.85
    c[near] = np.clip(r[near] + rng.integers(-6, 7, size=int(near.sum())), 0, n - 1)
    keep = r != c
    r, c = r[keep], c[keep]
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    e = np.unique(np.stack([lo, hi], 1), axis=0)
    w = 10.0 ** rng.uniform(-2, 2, size=len(e))
    rows = np.concatenate([e[:, 0], e[:, 1]]); cols = np.concatenate([e[:, 1], e[:, 0]]); vals = np.concatenate([-w, -w])
    diag = np.zeros(n); np.add.at(diag, rows, -vals)
    rows = np.concatenate([rows, np.arange(n)]); cols = np.concatenate([cols, np.arange(n)]); vals = np.concatenate([vals, diag + 1.0])
    return rows.astype(np.int32), cols.astype(np.int32), vals


def plain_rounds(n, rows, cols, vals, thr):
    """the reference's algorithm without shortcuts; returns the coarse id of every row (roots numbered in ascending order)"""
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))])
    off = rows != cols
    mx = np.full(n, -np.inf); np.maximum.at(mx, rows[off], -vals[off])
    strong = [[] for _ in range(n)]
    for i in range(n):
        for k in range(ptr[i], ptr[i + 1]):
            j = cols[k]
            if i == j or (-vals[k] / mx[i] > thr) or (-vals[k] / mx[j] > thr):
                strong[i].append(j)
    agg = np.arange(n); decided = np.zeros(n, bool); root = np.zeros(n, bool)
    while not decided.all():
        agg2 = agg.copy(); dec = np.ones(n, bool); rn = np.zeros(n, bool)
        for i in np.flatnonzero(~decided):
            for c in strong[i]:
                if agg[c] < agg2[i] and (not decided[c] or root[c]):
                    agg2[i], dec[i], rn[i] = agg[c], decided[c], root[c]
        for i in np.flatnonzero(~decided):
            if dec[i]:
                decided[i] = True
                if agg[i] == agg2[i]:
                    root[i] = True
                elif rn[i]:
                    agg[i] = agg2[i]
    ids = np.flatnonzero(root)
    return np.searchsorted(ids, agg), len(ids)


@pytest.mark.parametrize("n,deg,seed", [(600, 4, 1), (1500, 7, 2), (2500, 3, 3)])
def test_aggregates_equal_the_plain_rounds(n, deg, seed):
    rows, cols, vals = random_spd_graph(n, deg, seed)
    L = host.load("host")
    A = host.Matrix(host.Comm("host", "self"))
    A.set_many(rows, cols, vals)
    A.assemble()
    opts = dict(host.OPTIONS001)
    S = host.AmgSolver(A, host.options(L, **opts))
    got, ngot = S.level_aggregates(0)
    want, nwant = plain_rounds(n, rows, cols, vals, float(np.float32(opts.get("connStrength", 0.2))))
    assert ngot == nwant
    np.testing.assert_array_equal(got, want)


def _worker(rank, world, name, n, deg, seed, ret):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        rows, cols, vals = random_spd_graph(n, deg, seed)
        L = host.load("host")
        A = host.Matrix(host.Comm("host", "shm", (name, rank, world)))
        mine = rows % world == rank                     # every rank contributes a share of the entries; assemble() routes them
        A.set_many(rows[mine], cols[mine], vals[mine])
        A.assemble()
        S = host.AmgSolver(A, host.options(L, **host.OPTIONS001))
        ret[rank] = [(S.level_info(l)["rows"], S.level_info(l)["nnzA"], S.level_info(l)["nnzP"]) for l in range(S.num_levels)]
    except Exception as e:                              # noqa: BLE001 -- reported to the parent
        ret[rank] = f"{type(e).__name__}: {e}"


@pytest.mark.parametrize("world", [2, 3])
def test_row_distributed_aggregation_builds_the_one_rank_hierarchy(world):
    """the same random graph row-distributed over 2 / 3 ranks (native shared-memory communicator): every level has the rows and
    the entries of the one-rank hierarchy -- one row joining another aggregate would change them"""
    import multiprocessing as mp
    import os
    n, deg, seed = 4000, 6, 7
    rows, cols, vals = random_spd_graph(n, deg, seed)
    L = host.load("host")
    A1 = host.Matrix(host.Comm("host", "self"))
    A1.set_many(rows, cols, vals)
    A1.assemble()
    S1 = host.AmgSolver(A1, host.options(L, **host.OPTIONS001))
    want = [(S1.level_info(l)["rows"], S1.level_info(l)["nnzA"], S1.level_info(l)["nnzP"]) for l in range(S1.num_levels)]
    assert len(want) >= 3
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, f"aggr_{os.getpid()}_{world}", n, deg, seed, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        for p in procs:
            if p.is_alive():
                p.terminate()
        got = dict(ret)
    for r in range(world):
        assert got.get(r) == want, (r, got.get(r), want)
