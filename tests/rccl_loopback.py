"""RCCL paths on ONE GPU (run as a subprocess by tests/test_gpu_rccl.py; `--torch` imports torch first so
that torch's bundled HIP/RCCL runtime serves the library, the configuration of `bench.py --gpus N`).

RCCL refuses two ranks on one device, so this uses a 1-rank communicator:
  * ncclCommInitRank / ncclAllReduce (sgpu_barrier, sgpu_dot) / ncclAllGather + grouped send/recv of the
    host-side setup collectives (saena::matrix::assemble through RcclHostComm);
  * the halo exchange of sgpu_spmv with the neighbour replaced by the rank itself: pack kernel -> event ->
    ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the halo stream -> event -> remote kernel.
    Expected result: the receive buffer equals the rank's own send buffer.
"""
import sys

if "--torch" in sys.argv:
    import torch  # noqa: F401

import numpy as np

from oracle import oracle as orc
from saena_amd import capi, host
from tests import inputs, util


def main():
    capi.init(0, 0, 1, capi.get_unique_id())
    lib = capi.lib()
    capi.check(lib.sgpu_barrier())
    # the exchange chain was measured on this communicator at init (here: self send/recv) and is what the agglomeration
    # model and the one- / two-stream thresholds start from, in place of the constant of rounds 1-2
    chain = capi.chain_us()
    assert 2.0 < chain < 500.0, chain
    assert host.load("gpu").saena_measured_chain_us() == chain
    print(f"exchange chain measured at init: {chain:.1f} us", flush=True)
    n = 10007
    x = inputs.v2(n)
    dx = capi.DeviceVector(n, x)
    d = capi.dot(dx, dx)
    assert abs(d - x @ x) <= 1e-12 * (x @ x), (d, x @ x)

    # host setup collectives over RCCL (1 rank): same layout as the self communicator
    A1 = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(8).assemble().layout()
    A2 = host.Matrix(host.Comm("gpu", "self")).laplacian3D(8).assemble().layout()
    for k, v in A2.items():
        if v is not None and not np.isscalar(v):
            np.testing.assert_array_equal(A1[k], v, err_msg=k)

    # halo loopback: rank 0's share of a 2-rank Poisson operator, its neighbour rewired to itself
    entries, M = orc.laplacian3d(14)
    split = orc.split_even(M, 2)
    O = orc.OracleOp(entries, M, M, split)
    R = O.rank(0)
    arr = lambda name, cnt, dt: O.rank_array(0, name, cnt, dt)     # noqa: E731
    assert R.numSendProc == 1 and R.numRecvProc == 1 and R.vIndexSize == R.recvSize
    for fp32 in (False, True):
        op = capi.Operator(
            M=R.M, N_local=R.M, col_offset=0,
            nnzPerRow_local=arr("nnzPerRow_local", R.M, np.int32), col_local=arr("col_local", R.nnz_l_local, np.int32),
            val_local=arr("val_local", R.nnz_l_local, np.float64),
            nnzPerCol_remote=arr("nnzPerCol_remote", R.col_remote_size, np.int32),
            row_remote=arr("row_remote", R.nnz_l_remote, np.int32), val_remote=arr("val_remote", R.nnz_l_remote, np.float64),
            recvProcRank=[0], recvProcCount=arr("recvProcCount", 1, np.int32),
            sendProcRank=[0], sendProcCount=arr("sendProcCount", 1, np.int32),
            vIndex=arr("vIndex", R.vIndexSize, np.int32), inv_diag=arr("inv_diag", R.M, np.float64), halo_fp32=fp32)
        xl = inputs.v2(R.M)
        rhs = inputs.rhs2(R.M)
        # expected on the host
        npr = arr("nnzPerRow_local", R.M, np.int32)
        rows = np.repeat(np.arange(R.M), npr)
        yl = np.zeros(R.M)
        np.add.at(yl, rows, arr("val_local", R.nnz_l_local, np.float64) * xl[arr("col_local", R.nnz_l_local, np.int32)])
        recv = xl[arr("vIndex", R.vIndexSize, np.int32)]
        if fp32:
            recv = recv.astype(np.float32).astype(np.float64)
        npc = arr("nnzPerCol_remote", R.col_remote_size, np.int32)
        cols = np.repeat(np.arange(R.col_remote_size), npc)
        yr = np.zeros(R.M)
        np.add.at(yr, arr("row_remote", R.nnz_l_remote, np.int32), arr("val_remote", R.nnz_l_remote, np.float64) * recv[cols])
        want = yl + yr
        dxl, dy, dr = capi.DeviceVector(R.M, xl), capi.DeviceVector(R.M), capi.DeviceVector(R.M, rhs)
        for rep in range(3):                      # repeated exchanges reuse the persistent buffers and events
            op.spmv(dxl, dy)
        got = dy.download()
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want)), ("spmv", fp32, np.max(np.abs(got - want)))
        # a fused smoother sweep through the same exchange
        du = capi.DeviceVector(R.M, xl)
        op.jacobi(1, du, dr)
        omega = float(np.float32(2.0 / 3))
        want_u = xl - (arr("inv_diag", R.M, np.float64) * omega) * (want - rhs)
        assert np.max(np.abs(du.download() - want_u)) <= 1e-12 * np.max(np.abs(want_u)), ("jacobi", fp32)
        # residual and a 3-step Chebyshev smoother (every step a fresh exchange of the updated iterate), against the
        # same operator applied on the host: A_loop v = local v + remote v[vIndex]
        vi, invd = arr("vIndex", R.vIndexSize, np.int32), arr("inv_diag", R.M, np.float64)
        lc, lv = arr("col_local", R.nnz_l_local, np.int32), arr("val_local", R.nnz_l_local, np.float64)
        rr, rv = arr("row_remote", R.nnz_l_remote, np.int32), arr("val_remote", R.nnz_l_remote, np.float64)

        def A_loop(v):
            out = np.zeros(R.M)
            np.add.at(out, rows, lv * v[lc])
            h = v[vi].astype(np.float32).astype(np.float64) if fp32 else v[vi]
            np.add.at(out, rr, rv * h[cols])
            return out
        dres = capi.DeviceVector(R.M)
        op.residual(dxl, dr, dres)
        want_res = A_loop(xl) - rhs
        assert np.max(np.abs(dres.download() - want_res)) <= 1e-12 * np.max(np.abs(want_res)), ("residual", fp32)
        eig = 2.0
        alpha, beta = 0.13 * eig, eig
        delta, theta = (beta - alpha) / 2, (beta + alpha) / 2
        s1 = theta / delta
        rhok = 1 / s1
        u = xl.copy()
        d = (1 / theta) * invd * (rhs - A_loop(u))
        u = u + d
        for _ in range(2):
            rhokp1 = 1 / (2 * s1 - rhok)
            d = rhokp1 * rhok * d + (2 * rhokp1 / delta) * invd * (rhs - A_loop(u))
            u = u + d
            rhok = rhokp1
        du.upload(xl)
        op.chebyshev(3, eig, du, dr)
        tol = 1e-6 if fp32 else 1e-12       # the fp32 wire rounds the halo of every step
        assert np.max(np.abs(du.download() - u)) <= tol * np.max(np.abs(u)), ("chebyshev", fp32, np.max(np.abs(du.download() - u)))
        ms = op.time_kernel(0, dxl, None, dy, 20)
        print(f"loopback halo ok (fp32={fp32}): {R.vIndexSize} doubles each way, {ms * 1e3:.1f} us per SpMV incl. exchange", flush=True)
    capi.finalize()
    print("RCCL_LOOPBACK_OK", flush=True)


if __name__ == "__main__":
    main()
