"""Test helpers: feed the oracle's reference-layout arrays through the C ABI."""
import numpy as np

from oracle import oracle as orc
from saena_amd import capi


def gpu_operator(op: "orc.OracleOp", r=0, halo_fp32=False):
    """Build an sgpu_op for simulated rank r from the oracle's storage layout
    (the same arrays a Saena maintainer would pass from saena_matrix members)."""
    R = op.rank(r)
    has_diag = bool(R.inv_diag) and R.M > 0      # like the product's fill_desc: no rows -> no inv_diag pointer
    return capi.Operator(
        M=R.M, N_local=int(op.split_col[r + 1] - op.split_col[r]), col_offset=int(op.split_col[r]),
        nnzPerRow_local=op.rank_array(r, "nnzPerRow_local", R.M, np.int32),
        col_local=op.rank_array(r, "col_local", R.nnz_l_local, np.int32),
        val_local=op.rank_array(r, "val_local", R.nnz_l_local, np.float64),
        nnzPerCol_remote=op.rank_array(r, "nnzPerCol_remote", R.col_remote_size, np.int32),
        row_remote=op.rank_array(r, "row_remote", R.nnz_l_remote, np.int32),
        val_remote=op.rank_array(r, "val_remote", R.nnz_l_remote, np.float64),
        recvProcRank=op.rank_array(r, "recvProcRank", R.numRecvProc, np.int32),
        recvProcCount=op.rank_array(r, "recvProcCount", R.numRecvProc, np.int32),
        sendProcRank=op.rank_array(r, "sendProcRank", R.numSendProc, np.int32),
        sendProcCount=op.rank_array(r, "sendProcCount", R.numSendProc, np.int32),
        vIndex=op.rank_array(r, "vIndex", R.vIndexSize, np.int32),
        inv_diag=op.rank_array(r, "inv_diag", R.M, np.float64) if has_diag else None,
        halo_fp32=halo_fp32,
    )


def plan(op: "orc.OracleOp", r):
    """(sendRank, sendCount, sendDispl, recvRank, recvCount, recvDispl) of simulated rank r"""
    R = op.rank(r)
    sr = op.rank_array(r, "sendProcRank", R.numSendProc, np.int32)
    sc = op.rank_array(r, "sendProcCount", R.numSendProc, np.int32)
    rr = op.rank_array(r, "recvProcRank", R.numRecvProc, np.int32)
    rc = op.rank_array(r, "recvProcCount", R.numRecvProc, np.int32)
    sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(int) if len(sc) else np.zeros(0, int)
    rd = np.concatenate([[0], np.cumsum(rc)[:-1]]).astype(int) if len(rc) else np.zeros(0, int)
    return sr, sc, sd, rr, rc, rd


class EmulatedWorld:
    """P simulated ranks on ONE GPU: each rank's operator lives in the single
    1-rank context; the halo is packed by the GPU pack kernel, routed between
    ranks on the host, and injected (sgpu_debug_pack / sgpu_debug_inject_halo).
    Exercises K3 (pack), the remote-CSR kernel and the remote epilogues without
    RCCL, which refuses two ranks on one device."""

    def __init__(self, op: "orc.OracleOp", halo_fp32=False):
        self.op = op
        self.P = op.nprocs
        self.g = [gpu_operator(op, r, halo_fp32) for r in range(self.P)]
        self.plans = [plan(op, r) for r in range(self.P)]

    def exchange(self, xs):
        """xs[r]: DeviceVector holding rank r's slice of the input vector"""
        sends = []
        for r in range(self.P):
            n = int(self.plans[r][1].sum())
            sends.append(self.g[r].debug_pack(xs[r], n) if n else np.zeros(0))
        for r in range(self.P):
            sr, sc, sd, rr, rc, rd = self.plans[r]
            recv = np.zeros(int(rc.sum()))
            for q, cnt, dsp in zip(rr, rc, rd):
                qs = self.plans[q]
                k = list(qs[0]).index(r)
                src = sends[q][qs[2][k]:qs[2][k] + qs[1][k]]
                assert len(src) == cnt
                recv[dsp:dsp + cnt] = src
            self.g[r].debug_inject_halo(recv)

    def slices(self, v, split):
        return [capi.DeviceVector(split[r + 1] - split[r], v[split[r]:split[r + 1]]) for r in range(self.P)]

    def gather(self, ys):
        return np.concatenate([y.download() for y in ys])
