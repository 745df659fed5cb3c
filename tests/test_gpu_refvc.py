"""The HIP path against vectors computed by the COMPILED REFERENCE operators on a real smoothed-aggregation hierarchy
(tests/golden/refvc_*, oracle/ref/ref_vcycle.cpp): sgpu_vcycle against the reference's composed V-cycles (1e-11), R v
and P e -- fp64 and the fp32-halo form of restrict_matrix / prolong_matrix -- at the reference's own 2- and 4-rank
partitions through the pack / boundary kernels (1e-13 of the row scale)."""
import os

import numpy as np
import pytest

from tests import refvc, util

pytestmark = pytest.mark.gpu

TOL_TRANSFER = 1e-13
TOL_VCYCLE = 1e-11


@pytest.fixture(scope="module")
def capi():
    from saena_amd import capi as c
    c.init(0)
    return c


@pytest.mark.parametrize("fn", refvc.FIXTURES, ids=os.path.basename)
@pytest.mark.parametrize("coarse", ["cg", "direct"])
def test_gpu_vcycle_matches_the_compiled_reference_vectors(capi, fn, coarse):
    hier, ref = refvc.load(fn)
    nl = int(hier["nlevels"])
    one = [np.array([0, int(hier[f"A{l}_shape"][0])], np.int32) for l in range(nl)]
    OA, OP, OR = refvc.oracle_hierarchy(hier, one)
    GA, GP, GR = [util.gpu_operator(a) for a in OA], [util.gpu_operator(p) for p in OP], [util.gpu_operator(r) for r in OR]
    n0 = OA[0].Mbig
    u0, rhs = 0.01 * refvc.v2(n0), refvc.rhs2(n0)
    dr = capi.DeviceVector(n0, rhs)
    for name, (smoother, pre, post) in refvc.VCYCLE_CASES.items():
        G = capi.Amg(GA, GP, GR, eig_max=hier["eig"], pre=pre, post=post, smoother=smoother, coarse_solver=coarse)
        for key, start in ((f"vcycle_{name}", u0), (f"vcycle0_{name}", np.zeros(n0))):
            du = capi.DeviceVector(n0, start)
            G.vcycle(du, dr)
            got, want = du.download(), ref[key]
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert err <= TOL_VCYCLE, (key, coarse, err)
            du.upload(start)
            G.vcycle(du, dr)                                   # the graph replay gives the same bits
            np.testing.assert_array_equal(du.download(), got)
        G.destroy()


@pytest.mark.parametrize("fn", [f for f in refvc.FIXTURES if ".np1." not in f], ids=os.path.basename)
@pytest.mark.parametrize("fp32", [False, True], ids=["fp64-halo", "fp32-halo"])
def test_gpu_transfers_at_the_reference_partitions(capi, fn, fp32):
    """R_l v and P_l e with halos, rank by rank at the partitions the reference chose (nnz-balanced: uneven blocks),
    incl. matvec_sparse_float of restrict_matrix (src/restrict_matrix.cpp:746-871) and prolong_matrix (:626-758)"""
    hier, ref = refvc.load(fn)
    nl = int(hier["nlevels"])
    splits = [ref[f"split{l}"] for l in range(nl)]
    nprocs = len(splits[0]) - 1
    OA, OP, OR = refvc.oracle_hierarchy(hier, splits)
    sfx = "_float" if fp32 else ""
    for l in range(nl - 1):
        Mf, Mc = OA[l].Mbig, OA[l + 1].Mbig
        v, e = refvc.v2(Mf), refvc.ec(Mc)
        WR, WP = util.EmulatedWorld(OR[l], halo_fp32=fp32), util.EmulatedWorld(OP[l], halo_fp32=fp32)
        vs, cs = WR.slices(v, splits[l]), WR.slices(np.zeros(Mc), splits[l + 1])
        WR.exchange(vs)
        for r in range(nprocs):
            WR.g[r].spmv(vs[r], cs[r])
        bound = refvc.abs_product(hier, "P", l, v, transpose=True)
        assert np.all(np.abs(WR.gather(cs) - ref[f"R{l}_v2{sfx}"]) <= TOL_TRANSFER * bound + 1e-300), f"R{l}{sfx}"
        es, fs = WP.slices(e, splits[l + 1]), WP.slices(np.zeros(Mf), splits[l])
        WP.exchange(es)
        for r in range(nprocs):
            WP.g[r].spmv(es[r], fs[r])
        bound = refvc.abs_product(hier, "P", l, e)
        assert np.all(np.abs(WP.gather(fs) - ref[f"P{l}_ec{sfx}"]) <= TOL_TRANSFER * bound + 1e-300), f"P{l}{sfx}"
