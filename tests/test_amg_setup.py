"""Host AMG setup (smoothed aggregation restatement) against numbers printed by the
reference itself (SURVEY.md 6 / BASELINE.md 2, `mpirun ./poisson 32 data/options001.xml`):

  hierarchy 32^3: rows 27000/13500/1420/253/69, nnz 183600/833962/257610/60479/4761
  solve_pCG (Jacobi 3+3): ||r0|| = 7.227341e+03 -> 2.246251e-05 (rel 3.107991e-09), 7 iterations

The second pin runs the ORACLE's V-cycle/pCG restatement on the hierarchy built by the product's
host setup, which ties both to the reference's own output.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from saena_amd import host


def coo_from_layout(L, row_ofs=0):
    """entries (global ids) of a 1-rank operator from its reference-layout arrays"""
    rows = np.repeat(np.arange(L["M"], dtype=np.int32), L["nnzPerRow_local"]) + row_ofs
    return orc.coo_from_arrays(rows, L["col_local"], L["val_local"])


def oracle_amg_from_host(S, smoother="jacobi", **kw):
    n = S.num_levels
    OA, OP, OR = [], [], []
    for l in range(n):
        LA = S.level_layout(l, 0)
        M = LA["M"]
        OA.append(orc.OracleOp(coo_from_layout(LA), M, M, orc.split_even(M, 1)))
        OA[-1].set_eig(S.level_info(l)["eig_max"])
        if l < n - 1:
            LP, LR = S.level_layout(l, 1), S.level_layout(l, 2)
            Nc = LR["M"]
            OP.append(orc.OracleOp(coo_from_layout(LP), M, Nc, orc.split_even(M, 1), orc.split_even(Nc, 1), square=False))
            OR.append(orc.OracleOp(coo_from_layout(LR), Nc, M, orc.split_even(Nc, 1), orc.split_even(M, 1), square=False))
    return orc.OracleAmg(OA, OP, OR, smoother=smoother, **kw), OA


@pytest.fixture(scope="module")
def solver32():
    L = host.load("host")
    A = host.Matrix(host.Comm("host", "self")).laplacian3D(32).assemble()
    return A, host.AmgSolver(A, host.options(L, **host.OPTIONS001))


def test_hierarchy_sizes_32(solver32):
    _, S = solver32
    assert S.num_levels == 5
    info = [S.level_info(l) for l in range(5)]
    assert [i["rows"] for i in info] == [27000, 13500, 1420, 253, 69]
    assert [i["nnzA"] for i in info] == [183600, 833962, 257610, 60479, 4761]


def test_reference_convergence_pin_32(solver32):
    A, S = solver32
    amg, OA = oracle_amg_from_host(S, "jacobi", pre=3, post=3, max_iter=50, tol=1e-8)
    rhs = A.laplacian3D_rhs()
    u, it, hist = amg.solve_pCG(rhs)
    assert it == 7
    assert abs(hist[0] - 7.227341e+03) <= 0.5e-3 * 1e0 + 1e-6 * hist[0]        # printed with 7 digits
    assert abs(hist[0] / 7.227341e+03 - 1) < 1e-6
    assert abs(hist[-1] / 2.246251e-05 - 1) < 2e-6, hist[-1]
    assert abs(hist[-1] / hist[0] / 3.107991e-09 - 1) < 2e-6


def test_options_xml(tmp_path):
    L = host.load("host")
    xml = tmp_path / "o.xml"
    xml.write_text('<?xml version="1.0"?>\n<SAENA>\n <OPTIONS solver_max_iter="50" solver_tol="1e-8" smoother="jacobi" '
                   'preSmooth="3" postSmooth="2" PSmoother="jacobi" conn_str="0.2" dynamic_levels="1" max_level="20" '
                   'float_level="3" filter_thre="1e-14" filter_max="1e-8" filter_start="1" filter_rate="2" '
                   'switch_to_dense="0" dense_thre="0.1" dense_sz_thre="5000" petsc=""/>\n</SAENA>\n')
    o = host.options(L, xml=str(xml))
    assert (o.solver_max_iter, o.smoother, o.preSmooth, o.postSmooth, o.max_level) == (50, 0, 3, 2, 20)
    assert abs(o.connStrength - 0.2) < 1e-7 and o.filter_thre == 1e-14 and o.filter_rate == 2
    d = host.options(L)                   # saena.hpp:151-155 defaults
    assert (d.solver_max_iter, d.smoother, d.max_level, d.float_level) == (100, 1, 10, 3)


def test_chebyshev_eig_estimate():
    """find_eig: the Lanczos estimate bounds lambda_max(D^-1 A) of Poisson (< 2) from within 1%"""
    L = host.load("host")
    A = host.Matrix(host.Comm("host", "self")).laplacian3D(16).assemble()
    S = host.AmgSolver(A, host.options(L, **dict(host.OPTIONS001, smoother="chebyshev")))
    e = S.level_info(0)["eig_max"]
    n = 14
    exact = 1 + np.cos(np.pi / (n + 1))        # lambda_max(D^-1 A) of the n^3 Dirichlet Laplacian
    assert 0.97 * exact < e <= 1.0002 * exact


# ---- multi-rank: redundant setup + per-level row partition (over gloo, no GPU) ----
# agglomeration policies of the coarse levels (host/amg_setup.h), by environment:
POLICIES = {
    # round 1's rule: levels of <= 4096 rows live on rank 0, nothing else moves (keeps several distributed coarse levels at test size)
    "rows4096": {"SAENA_SHRINK_CHAIN_US": "0", "SAENA_SHRINK_ROWS": "4096"},
    # the default model (23 us exchange chain against the level's one-GPU time): at test size every coarse level goes to rank 0
    "model": {},
    # the same model with a 3.5 us chain, so that its intermediate rule fires at test size: groups of 2 ranks merge
    # (ranks 0, 2 stay active) before the smallest levels go to rank 0
    "stride": {"SAENA_SHRINK_CHAIN_US": "3.5"},
}


def _dist_worker(rank, world, port, m, ret, smoother="jacobi", slab=False, policy="rows4096", comm_kind="dist", buckets=0):
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(POLICIES[policy])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from tests.test_host_layout import assert_layout_equal, oracle_layout
        L = host.load("host")
        # the setup's collectives: torch.distributed through callbacks, or the native shared-memory communicator (the
        # gloo group stays open either way: the checks below exchange halos with it)
        comm = host.Comm("host", "dist", dist) if comm_kind == "dist" else host.Comm("host", "shm", (f"test_{port}", rank, world))
        kw = dict(host.OPTIONS001, smoother=smoother)
        if slab:      # bench.py's weak-scaled operator: even z-slabs of an m x m x (nz world + 2) grid (anisotropic: aggregates span ranks)
            nz, n2 = 8, (m - 2) ** 2
            A = host.Matrix(comm).laplacian3D(m, m, nz * world + 2)
            A.assemble(np.array([r * n2 * nz for r in range(world + 1)], np.int32))
            A1 = host.Matrix(host.Comm("host", "self")).laplacian3D(m, m, nz * world + 2).assemble()
        else:
            A = host.Matrix(comm).set_partition_buckets(buckets).laplacian3D(m).assemble()
            A1 = host.Matrix(host.Comm("host", "self")).laplacian3D(m).assemble()
            rows = np.diff(A.split).astype(np.float64)
            if buckets:      # the opt-in finer fine-level partition: a rank's rows within 5 % of the mean (the reference's nparts^2
                assert rows.max() <= 1.05 * rows.mean(), A.split      # buckets: 1.25 at 4 ranks, 1.125 at 8 on this operator)
            elif world in (4, 8):
                assert rows.max() >= 1.1 * rows.mean(), A.split
        S = host.AmgSolver(A, host.options(L, **kw))
        S1 = host.AmgSolver(A1, host.options(L, **kw))     # the same hierarchy at one rank, in this process
        assert S.num_levels == S1.num_levels
        for l in range(S.num_levels):
            a, b = S.level_info(l), S1.level_info(l)
            assert (a["rows"], a["nnzA"], a["nnzP"]) == (b["rows"], b["nnzA"], b["nnzP"]), (l, a, b)
            if smoother == "chebyshev":      # per-rank partial sums in the Lanczos dots: equal to rounding
                assert abs(a["eig_max"] - b["eig_max"]) <= 1e-12 * b["eig_max"], (l, a["eig_max"], b["eig_max"])
        splits = [S.level_split(l) for l in range(S.num_levels)]
        np.testing.assert_array_equal(splits[0], A.split)
        assert any(np.all(s[1:] == s[-1]) for s in splits[1:]), "small levels must shrink onto rank 0"
        owners = [[r for r in range(world) if s[r + 1] > s[r]] for s in splits]
        if policy == "rows4096":
            assert any(len(set(s.tolist())) == world + 1 for s in splits[1:]), "a large coarse level must stay distributed"
        elif policy == "model":
            assert all(o == [0] for o in owners[1:]), owners
        else:
            strided = [o for o in owners[1:] if 1 < len(o) < world]
            assert strided and all(o == list(range(0, world, world // len(o)))[:len(o)] or o == list(range(0, world, -(-world // len(o)))) for o in strided), owners      # groups of k merged onto ranks 0, k, 2k ...
            if world == 4:
                assert any(o == [0, 2] for o in owners[1:]), owners
            assert all(set(owners[l + 1]) <= set(owners[l]) for l in range(len(owners) - 1)), owners   # agglomeration is monotone
        # the coarse operators are re-partitioned by nnz over the ranks that still own rows (the reference's Ac->repart()):
        # every owner of a distributed coarse level holds about its share of the level's entries
        import torch
        for l in range(1, S.num_levels):
            own = owners[l]
            if len(own) < 2:
                continue
            d = S.level_layout(l, 0)
            mine = torch.tensor([float(len(d["col_local"]) + len(d["row_remote"]))], dtype=torch.float64)
            allv = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allv, mine)
            per = np.array([float(t[0]) for t in allv])[own]
            assert per.max() <= 1.6 * per.mean(), (l, per)

        def empty_in_the_middle(sp):      # an empty block followed by a non-empty one
            ne = [sp[r + 1] > sp[r] for r in range(world)]
            return any((not ne[r]) and any(ne[r + 1:]) for r in range(world))

        import scipy.sparse as sp_
        import torch
        for l in range(S.num_levels):
            assert splits[l][-1] == S1.level_info(l)["rows"]
            for which in ((0, 1, 2) if l < S.num_levels - 1 else (0,)):
                glob = S1.level_layout(l, which)
                ent = coo_from_layout(glob)
                rs = splits[l] if which != 2 else splits[l + 1]
                cs = splits[l] if which == 0 else (splits[l + 1] if which == 1 else splits[l])
                nrows, ncols = rs[-1], cs[-1]
                got = S.level_layout(l, which)
                if not (empty_in_the_middle(rs) or empty_in_the_middle(cs)):
                    O = orc.OracleOp(ent, nrows, ncols, rs, cs, square=(which == 0))
                    want = oracle_layout(O, rank)
                    want.pop("col_remote")
                    assert_layout_equal(got, want, f"level {l} op {which} rank {rank}")
                    continue
                # A partition whose owners are every k-th rank (k-rank agglomeration) is not one the reference -- hence the
                # oracle -- can lay out: its owner search (aux_functions.h:39-58) returns the FIRST of equal split values,
                # i.e. an empty rank, because the reference drops the idle ranks from the communicator instead
                # (shrink_cpu).  These layouts are checked by what they compute: y = A x through the layout arrays with
                # the halo moved over gloo by the layout's own send/recv plan, against the one-rank operator.
                r_ = np.repeat(np.arange(glob["M"]), glob["nnzPerRow_local"])
                Aglob = sp_.csr_matrix((glob["val_local"], (r_, glob["col_local"])), shape=(nrows, ncols))
                xg = np.sin(0.37 * np.arange(ncols) + 0.1)
                want_y = (Aglob @ xg)[rs[rank]:rs[rank + 1]]
                bound = (abs(Aglob) @ abs(xg))[rs[rank]:rs[rank + 1]]
                d = got
                x = xg[cs[rank]:cs[rank + 1]]
                M = d["M"]
                assert M == rs[rank + 1] - rs[rank] and d["N_local"] == len(x) and d["col_offset"] == cs[rank]
                rows_ = np.repeat(np.arange(M), d["nnzPerRow_local"])
                y = np.bincount(rows_, weights=d["val_local"] * x[d["col_local"] - d["col_offset"]], minlength=M).astype(np.float64)
                send = x[d["vIndex"]]
                reqs, bufs, so, ro = [], [], 0, 0
                for q, cnt in zip(d["sendProcRank"], d["sendProcCount"]):
                    assert cs[q + 1] > cs[q] or rs[q + 1] > rs[q]
                    reqs.append(dist.isend(torch.from_numpy(send[so:so + cnt].copy()), int(q)))
                    so += cnt
                for q, cnt in zip(d["recvProcRank"], d["recvProcCount"]):
                    assert cs[q + 1] > cs[q], "a halo source must own columns"
                    t = torch.empty(int(cnt), dtype=torch.float64)
                    bufs.append(t)
                    reqs.append(dist.irecv(t, int(q)))
                for rq in reqs:
                    rq.wait()
                recv = np.concatenate([t.numpy() for t in bufs]) if bufs else np.zeros(0)
                assert len(recv) == len(d["nnzPerCol_remote"])
                if len(recv):
                    slot = np.repeat(np.arange(len(recv)), d["nnzPerCol_remote"])
                    y += np.bincount(d["row_remote"], weights=d["val_remote"] * recv[slot], minlength=M)
                assert np.all(np.abs(y - want_y) <= 1e-13 * bound + 1e-300), f"level {l} op {which} rank {rank}"
                if which == 0 and M:
                    dg = Aglob.diagonal()[rs[rank]:rs[rank + 1]]
                    np.testing.assert_array_equal(d["inv_diag"], 1.0 / dg)
        ret[rank] = "ok"
    except BaseException as e:      # noqa
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,smoother,world,slab,policy,comm_kind", [
    ("rows", "jacobi", 3, False, "rows4096", "dist"), ("rows", "chebyshev", 4, False, "rows4096", "dist"),
    ("rows", "jacobi", 4, True, "rows4096", "dist"), ("gathered", "jacobi", 2, False, "rows4096", "dist"),
    ("rows", "jacobi", 4, False, "model", "dist"), ("rows", "chebyshev", 4, False, "stride", "dist"),
    ("gathered", "jacobi", 4, False, "stride", "dist"),
    ("rows", "chebyshev", 3, False, "rows4096", "shm"), ("rows", "jacobi", 4, True, "stride", "shm"),
    # round 4: the opt-in finer fine-level partition (the hierarchy is still the one-rank hierarchy bit for bit), and EIGHT ranks
    # over the shared-memory communicator with the k-rank agglomeration firing (configs[3]'s rank count)
    ("rows", "jacobi", 4, False, "rows4096", "shm+balanced"), ("rows", "jacobi", 8, False, "stride", "shm"),
    ("rows", "chebyshev", 8, False, "rows4096", "shm+balanced")])
def test_distributed_hierarchy_gloo(mode, smoother, world, slab, policy, comm_kind, monkeypatch):
    """The hierarchy built over several ranks -- every rank building only its rows of every level (the default), or the
    older gather-then-slice form -- is the one-rank hierarchy bit for bit: every level's A, P and R layout equals the
    oracle's layout of the one-rank operator under that level's partition.  The setup's collectives run over gloo through
    callbacks ("dist") or over the native shared-memory communicator ("shm", host/shm_comm.cpp)."""
    import torch.multiprocessing as mp
    from tests.test_host_layout import _free_port
    monkeypatch.setenv("SAENA_SETUP", mode)
    monkeypatch.setenv("SAENA_SETUP_THREADS", "2" if world < 8 else "1")
    buckets = 4096 if comm_kind.endswith("+balanced") else 0
    comm_kind = comm_kind.split("+")[0]
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_dist_worker, args=(r, world, port, 30 if slab else 24, ret, smoother, slab, policy, comm_kind, buckets)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        for p in procs:
            if p.is_alive():
                p.terminate()
        res = dict(ret)
    for r in range(world):
        assert res.get(r) == "ok", f"rank {r}: {res.get(r)}"


def test_matmat_against_scipy():
    """saena::amg::matmat (host SpGEMM): C = A B equals scipy's product; exact zeros off the diagonal are dropped"""
    import scipy.sparse as sp
    comm = host.Comm("host", "self")
    A = host.Matrix(comm).laplacian3D(9).assemble()
    B = host.Matrix(comm).band_matrix(A.num_rows, 3).assemble()
    Cm = A.matmat(B)

    def to_scipy(Mx):
        d = Mx.layout()
        rows = np.repeat(np.arange(d["M"]), d["nnzPerRow_local"])
        return sp.csr_matrix((d["val_local"], (rows, d["col_local"])), shape=(d["M"], d["M"]))
    want = (to_scipy(A) @ to_scipy(B)).tocsr()
    got = to_scipy(Cm)
    assert Cm.num_rows == A.num_rows
    diff = (got - want)
    assert abs(diff).max() <= 1e-12 * abs(want).max()
    assert got.nnz <= want.nnz and got.nnz >= want.nnz - (abs(want.data) <= 1e-14).sum()
