"""Host-side mirror (libsaena_host.so, no GPU): generators, boundary removal,
nnz-balanced partition and the storage layout must equal the oracle's (which is
pinned against the compiled reference) bit for bit -- at one rank, and at
world_size 2/3 over torch.distributed gloo (the N>1 path on CPU).
"""
import os
import socket
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from saena_amd import host
from tests import inputs, matrices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LAYOUT_KEYS = ["nnzPerRow_local", "col_local", "val_local", "nnzPerCol_remote", "row_remote", "col_remote", "val_remote",
               "recvProcRank", "recvProcCount", "sendProcRank", "sendProcCount", "vIndex", "inv_diag"]
ORC_DT = {"val_local": np.float64, "val_remote": np.float64, "inv_diag": np.float64}


def oracle_layout(A, r):
    R = A.rank(r)
    n = dict(nnzPerRow_local=R.M, col_local=R.nnz_l_local, val_local=R.nnz_l_local, nnzPerCol_remote=R.col_remote_size,
             row_remote=R.nnz_l_remote, col_remote=R.nnz_l_remote, val_remote=R.nnz_l_remote,
             recvProcRank=R.numRecvProc, recvProcCount=R.numRecvProc, sendProcRank=R.numSendProc,
             sendProcCount=R.numSendProc, vIndex=R.vIndexSize, inv_diag=R.M)
    return {k: A.rank_array(r, k, n[k], ORC_DT.get(k, np.int32)) for k in n if (k != "inv_diag" or bool(R.inv_diag))}


def assert_layout_equal(got, want, what):
    for k, w in want.items():
        g = got[k]
        if g is None and len(w) == 0:       # a rank that owns no rows of this operator
            continue
        assert g is not None, f"{what}: {k} missing"
        np.testing.assert_array_equal(g, w, err_msg=f"{what}: {k}")


@pytest.mark.parametrize("m", [5, 8, 12])
def test_poisson_single_rank(m):
    comm = host.Comm("host", "self")
    A = host.Matrix(comm).laplacian3D(m).assemble()
    entries, Mbig = orc.laplacian3d(m)
    assert (A.num_rows, A.num_local_rows, A.nnz, A.local_nnz) == (Mbig, Mbig, len(entries), len(entries))
    O = orc.OracleOp(entries, Mbig, Mbig, orc.split_nnz(entries, Mbig, 1))
    assert_layout_equal(A.layout(), oracle_layout(O, 0), f"poisson{m}")
    np.testing.assert_array_equal(A.laplacian3D_rhs(), orc.laplacian3d_rhs(m))


def test_band_and_duplicates_single_rank():
    comm = host.Comm("host", "self")
    A = host.Matrix(comm).band_matrix(300, 7).assemble()
    e = orc.band_matrix(300, 7)
    O = orc.OracleOp(e, 300, 300, orc.split_even(300, 1))
    assert_layout_equal(A.layout(), oracle_layout(O, 0), "band300_7")
    # duplicates are added (saena.hpp:46-47), tiny sums dropped (saena_matrix_setup.cpp:153)
    B = host.Matrix(comm)
    B.set_remove_boundary(False)
    for i, j, v in [(0, 0, 1.0), (0, 0, 2.0), (1, 1, 5.0), (0, 1, 1e-15), (1, 0, 0.5), (1, 0, -0.5)]:
        B.set(i, j, v)
    B.assemble()
    L = B.layout()
    np.testing.assert_array_equal(L["col_local"], [0, 1])
    np.testing.assert_array_equal(L["val_local"], [3.0, 5.0])


@pytest.mark.parametrize("name", ["plat362", "SiH4", "fxm3_6"])
def test_read_file_mtx_and_bin(name, tmp_path):
    """read_file (.mtx: real symmetric / pattern symmetric; .bin: 16-byte triples) against an independent parse"""
    comm = host.Comm("host", "self")
    A = host.Matrix(comm).read_file(matrices.path(name, tmp_path)).assemble()
    entries, M = matrices.entries(name)
    O = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    want = oracle_layout(O, 0)
    assert_layout_equal(A.layout(), want, name)
    binf = str(tmp_path / (name + ".bin"))
    A.write_bin(binf)
    assert os.path.getsize(binf) == 16 * len(entries)
    B = host.Matrix(comm).read_file(binf).assemble()
    assert_layout_equal(B.layout(), want, name + ".bin")


def test_config5_irregular_generator():
    """tests/irregular.py (bench.py's `spmv_irregular`, BASELINE configs[4] scaled up): deterministic, symmetric, no duplicate
    entries, a diagonal in every row, rows from 13 entries to a hub of ~3 000; assembled by the product's host library it is the
    oracle's layout bit for bit"""
    from tests import irregular
    r, c, v, M = irregular.sih4_replicated(3)
    r2, c2, v2, _ = irregular.sih4_replicated(3)
    assert np.array_equal(r, r2) and np.array_equal(c, c2) and np.array_equal(v, v2)
    key = r.astype(np.int64) * M + c
    assert len(np.unique(key)) == len(key)
    assert set(key.tolist()) == set((c.astype(np.int64) * M + r).tolist()), "pattern must be symmetric"
    assert np.count_nonzero(r == c) == M
    st = irregular.row_length_stats(r, M)
    assert st["min"] >= 1 and st["max"] >= 2900 and st["coefficient_of_variation"] > 0.8, st
    A = host.Matrix(host.Comm("host", "self"))
    A.set_remove_boundary(False)
    A.set_many(r, c, v)
    A.assemble()
    O = orc.OracleOp(orc.coo_from_arrays(r, c, v), M, M, orc.split_even(M, 1))
    assert_layout_equal(A.layout(), oracle_layout(O, 0), "sih4 x 3")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, ret):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        comm = host.Comm("host", "dist", dist)
        if case[0] == "file":
            import tempfile
            tmp = tempfile.mkdtemp()
            A = host.Matrix(comm).read_file(matrices.path(case[1], tmp)).assemble()     # every rank reads its chunk
            entries, Mbig = matrices.entries(case[1])
            rhs_want = None
        elif case[0] == "poisson":
            m = case[1]
            A = host.Matrix(comm).laplacian3D(m).assemble()
            entries, Mbig = orc.laplacian3d(m)
            rhs_want = orc.laplacian3d_rhs(m)
        else:
            Mloc, bw = case[1], case[2]
            A = host.Matrix(comm).band_matrix(Mloc, bw).assemble()
            Mbig = Mloc * world
            entries = orc.band_matrix(Mbig, bw)
            rhs_want = None
        split = orc.split_nnz(entries, Mbig, world)
        np.testing.assert_array_equal(A.split, split)
        golden = os.path.join(ROOT, "tests", "golden", f"ref_{case[0]}{case[1]}.np{world}.npz")
        if case[0] == "poisson" and os.path.exists(golden):                  # ... and the compiled reference's own partition at this rank count
            np.testing.assert_array_equal(A.split, np.load(golden)["split"])
        O = orc.OracleOp(entries, Mbig, Mbig, split)
        assert_layout_equal(A.layout(), oracle_layout(O, rank), f"{case} rank {rank}")
        if rhs_want is not None:
            np.testing.assert_array_equal(A.laplacian3D_rhs(), rhs_want[split[rank]:split[rank + 1]])
        # grid transfers on two partitions: P rows by `split`, columns by splitNew
        pr, pc, pv, Nc = inputs.synthetic_P(Mbig)
        splitNew = (split // 2).astype(np.int32); splitNew[-1] = Nc
        mine = (pr >= split[rank]) & (pr < split[rank + 1])
        P = host.Transfer.prolong(comm, Mbig, Nc, split, splitNew, pr[mine], pc[mine], pv[mine])
        R = P.transpose()
        OP = orc.OracleOp(orc.coo_from_arrays(pr, pc, pv), Mbig, Nc, split, splitNew, square=False)
        OR = orc.OracleOp(orc.coo_from_arrays(pc, pr, pv), Nc, Mbig, splitNew, split, square=False)
        wantP, wantR = oracle_layout(OP, rank), oracle_layout(OR, rank)
        wantP.pop("col_remote"); wantR.pop("col_remote")
        assert_layout_equal(P.layout(), wantP, f"P rank {rank}")
        assert_layout_equal(R.layout(), wantR, f"R rank {rank}")
        ret[rank] = "ok"
    except BaseException as e:      # noqa
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, ("poisson", 8)), (3, ("poisson", 12)), (2, ("band", 150, 7)), (4, ("band", 16, 15)),
                                        (3, ("file", "plat362")), (8, ("poisson", 16))])      # 8: north_star's rank count (fixture: mpirun -np 8 of the compiled reference)
def test_distributed_assemble_gloo(world, case):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, case, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
        for p in procs:
            if p.is_alive():
                p.terminate()
        res = dict(ret)
    for r in range(world):
        assert res.get(r) == "ok", f"rank {r}: {res.get(r)}"


def test_abi_symbols_present():
    """The C-ABI libraries load and export every symbol their headers declare (no compute without a GPU)."""
    import re
    from saena_amd import capi
    L = capi.lib()
    strip = lambda t: re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    hdr = strip(open(os.path.join(ROOT, "include", "saena_gpu.h")).read())
    assert "sgpu_debug" not in hdr, "test scaffolding belongs in include/saena_gpu_debug.h, not in the boundary header"
    hdr += strip(open(os.path.join(ROOT, "include", "saena_gpu_debug.h")).read())
    declared = set(re.findall(r"\b(sgpu_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name)
    hdr = strip(open(os.path.join(ROOT, "include", "saena_c.h")).read())
    declared = set(re.findall(r"\b(saena_[a-z_A-Z0-9]+)\s*\(", hdr)) - {"saena_cb_allgather", "saena_cb_alltoallv",
                                                                            "saena_cb_allreduce_i64", "saena_cb_allreduce_f64"}
    assert declared == set(host.HOST_SYMBOLS), declared ^ set(host.HOST_SYMBOLS)
    for which in ("host", "gpu"):
        H = host.load(which)
        for name in declared:
            assert hasattr(H, name)
    # the host-only library refuses the GPU communicator loudly
    with pytest.raises(capi.SgpuError):
        host.Comm("host", "rccl")


def test_gpu_path_fails_loudly_without_device():
    """No CPU fallback: on a box without an MI355X sgpu_init reports an error."""
    from saena_amd import capi
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from saena_amd import capi\n"
            "try:\n    capi.init(0)\n    print('INIT_OK')\nexcept capi.SgpuError as e:\n    print('LOUD', e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "LOUD" in out.stdout or "INIT_OK" in out.stdout, out.stdout + out.stderr


def test_fatal_signal_prints_the_measured_line_and_fails():
    """sgpu_debug_on_fatal_print (bench.py's safety net for its optional multi-rank legs): a fatal signal writes the
    registered line to stdout, a reason to stderr, and ends the process with status 128 + signal -- the failure must
    reach the launcher as a failure (no GPU needed: no compute call is made).  SIGTERM is not trapped."""
    import signal
    import subprocess
    code = ("import os, signal\n"
            "from saena_amd import capi\n"
            "capi.lib().sgpu_debug_on_fatal_print(b'{\"metric\": \"x\"}')\n"
            "os.kill(os.getpid(), signal.%s)\n"
            "print('not reached')\n")
    out = subprocess.run([sys.executable, "-c", code % "SIGABRT"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 128 + signal.SIGABRT and out.stdout == '{"metric": "x"}\n', (out.returncode, out.stdout, out.stderr[-500:])
    assert "fatal signal 6" in out.stderr
    out = subprocess.run([sys.executable, "-c", code % "SIGTERM"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == -signal.SIGTERM and out.stdout == "", (out.returncode, out.stdout)


def test_write_matrix_to_file_roundtrip(tmp_path):
    """saena::matrix::writeMatrixToFile: "<name>-r0.mtx" (header + 1-based triples) reads back to the same operator"""
    comm = host.Comm("host", "self")
    A = host.Matrix(comm).laplacian3D(9).assemble()           # stencil values are exact in 12 digits
    A.write_mtx(str(tmp_path / "lap"))
    fn = tmp_path / "lap-r0.mtx"
    head = fn.read_text().splitlines()[:2]
    assert head[0] == "%%MatrixMarket matrix coordinate real general" and head[1].split() == [str(A.num_rows), str(A.num_rows), str(A.nnz)]
    B = host.Matrix(comm)
    B.set_remove_boundary(False)
    B.read_file(str(fn)).assemble()
    assert_layout_equal(B.layout(), A.layout(), "roundtrip")
