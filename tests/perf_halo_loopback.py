"""What one halo exchange costs on the halo stream, measured on ONE GPU: the middle rank's share of a 3-rank
z-slab Poisson operator (two neighbours, one 126^2 plane each way = the exchange of `bench.py --gpus N`),
with both neighbours rewired to the rank itself (RCCL refuses two ranks on one device, so the wire is a
device-local copy: this isolates pack -> event -> ncclGroup(send,recv) -> event -> remote-kernel overhead;
xGMI adds ~127 KB / 50+ GB/s = a few microseconds on a real node).

python -m tests.perf_halo_loopback [--torch] [planes per rank ...]
"""
import sys

if "--torch" in sys.argv:
    import torch  # noqa: F401

import numpy as np

from oracle import oracle as orc
from saena_amd import capi


def operator(planes, m=128, fp32=False, exchange=True):
    n = m - 2
    entries, M = orc.laplacian3d(m, m, 3 * planes + 2)
    split = np.array([0, planes * n * n, 2 * planes * n * n, 3 * planes * n * n], np.int32)
    O = orc.OracleOp(entries, M, M, split)
    R = O.rank(1)
    arr = lambda name, cnt, dt: O.rank_array(1, name, cnt, dt)     # noqa: E731
    assert R.numSendProc == 2 and R.numRecvProc == 2
    kw = dict(M=R.M, N_local=R.M, col_offset=int(split[1]),
              nnzPerRow_local=arr("nnzPerRow_local", R.M, np.int32), col_local=arr("col_local", R.nnz_l_local, np.int32),
              val_local=arr("val_local", R.nnz_l_local, np.float64), inv_diag=arr("inv_diag", R.M, np.float64))
    if exchange:
        kw.update(nnzPerCol_remote=arr("nnzPerCol_remote", R.col_remote_size, np.int32),
                  row_remote=arr("row_remote", R.nnz_l_remote, np.int32), val_remote=arr("val_remote", R.nnz_l_remote, np.float64),
                  recvProcRank=[0, 0], recvProcCount=arr("recvProcCount", 2, np.int32),
                  sendProcRank=[0, 0], sendProcCount=arr("sendProcCount", 2, np.int32),
                  vIndex=arr("vIndex", R.vIndexSize, np.int32), halo_fp32=fp32)
    return capi.Operator(**kw), int(R.M), int(R.vIndexSize)      # (R is a view into O, which dies here)


def main():
    planes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 126]
    verbose = "-v" in sys.argv
    capi.init(0, 0, 1, capi.get_unique_id())
    for p in planes:
        line = f"{p:4d} planes/rank"
        for label, kw in (("local only", dict(exchange=False)), ("fp64 halo", dict()), ("fp32 halo", dict(fp32=True))):
            op, M, nhalo = operator(p, **kw)
            if verbose:
                print(f"  [{p} {label}] created {op.info()}", flush=True)
            op.autotune()
            if verbose:
                print(f"  [{p} {label}] autotuned -> {op.variant()}", flush=True)
            x, y = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M)
            op.spmv(x, y)
            capi.check(capi.lib().sgpu_device_sync())
            if verbose:
                print(f"  [{p} {label}] one spmv done", flush=True)
            op.time_kernel(0, x, None, y, 20)
            us = op.time_kernel(0, x, None, y, 300) * 1e3
            import time
            capi.check(capi.lib().sgpu_device_sync())
            t0 = time.perf_counter()
            for _ in range(300):
                op.spmv(x, y)
            t_enq = (time.perf_counter() - t0) / 300 * 1e6          # host time to ENQUEUE one SpMV (no sync)
            capi.check(capi.lib().sgpu_device_sync())
            t_all = (time.perf_counter() - t0) / 300 * 1e6
            line += f" | {label}: {us:7.2f} us (host enqueue {t_enq:.1f}, drained {t_all:.1f})"
            if label == "local only":
                line = f"{line} ({M} rows, halo 2 x {nhalo // 2} doubles)"
            op.destroy()
        print(line, flush=True)
    capi.finalize()


if __name__ == "__main__":
    main()
