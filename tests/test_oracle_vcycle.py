"""CPU checks of the oracle's V-cycle / solve / pCG restatement (no GPU)."""
import numpy as np

from oracle import oracle as orc
from tests import hierarchy, inputs


def _amg(nprocs, smoother="jacobi"):
    As, Ps, Rs = hierarchy.poisson_hierarchy(14, 3)
    OA, OP, OR = hierarchy.oracle_hierarchy(As, Ps, Rs, nprocs)
    for a, e in zip(OA, hierarchy.eig_estimates(As)):
        a.set_eig(e)
    return orc.OracleAmg(OA, OP, OR, smoother=smoother, max_iter=60), OA


def test_pcg_converges_and_is_rank_invariant():
    rhs = orc.laplacian3d_rhs(14)
    ref = None
    for nprocs in (1, 3):
        for smoother in ("jacobi", "chebyshev"):
            amg, OA = _amg(nprocs, smoother)
            u, it, hist = amg.solve_pCG(rhs)
            assert hist[-1] < 1e-8 * hist[0] and it < 30
            r = OA[0].residual(u, rhs)
            assert abs(np.linalg.norm(r) - hist[-1]) <= 1e-6 * hist[-1] + 1e-12
            if smoother == "jacobi":
                if ref is None:
                    ref = hist
                else:   # the reference gives the same digits at 1, 2, 4 ranks
                    assert len(hist) == len(ref) and np.all(np.abs(hist - ref) <= 1e-9 * ref)


def test_stationary_solve_matches_manual_vcycles():
    amg, OA = _amg(1)
    rhs = inputs.rhs2(OA[0].Mbig)
    u, it, hist = amg.solve(rhs)
    v = np.zeros_like(rhs)
    for _ in range(it):
        v = amg.vcycle(v, rhs)
    np.testing.assert_array_equal(u, v)


def test_solve_smoother_is_repeated_sweeps():
    """orc_solve_smoother (saena_object_solve.cpp:2017-2117) = preSmooth sweeps + residual per iteration, built from
    the smoother / residual restatements that are pinned against the compiled reference (test_oracle_pins.py)"""
    for smoother in ("jacobi", "chebyshev"):
        amg, OA = _amg(2, smoother)
        amg.set_solver(max_iter=6, tol=1e-30)
        rhs = inputs.rhs2(OA[0].Mbig)
        u, it, hist = amg.solve_smoother(rhs)
        assert it == 6 and len(hist) == 7
        v = np.zeros_like(rhs)
        want = [np.linalg.norm(OA[0].residual(v, rhs))]
        for _ in range(6):
            v = OA[0].jacobi(3, v, rhs) if smoother == "jacobi" else OA[0].chebyshev(3, v, rhs)
            want.append(np.linalg.norm(OA[0].residual(v, rhs)))
        np.testing.assert_array_equal(u, v)
        np.testing.assert_allclose(hist, want, rtol=1e-13)
        assert hist[-1] < hist[0]
