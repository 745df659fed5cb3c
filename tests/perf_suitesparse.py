"""BASELINE.json configs[4]: SpMV / Jacobi on the irregular SuiteSparse fixtures (and a wide band matrix)
through the product path.  python -m tests.perf_suitesparse   (GPU box; prints one line per matrix)

These matrices are far too small to reach the HBM roofline (0.07 - 20 MB of operator); the lines show
what the row-block planner does with irregular rows (rows/block, lanes per row, kernel chosen) and the
launch-latency floor."""
import gzip
import os
import tempfile

import numpy as np

from saena_amd import capi, host

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matrices")
FILES = {"plat362": "plat362.mtx", "SiH4": "SiH4.mtx.gz", "fxm3_6": "fxm3_6.mtx.gz"}


def mtx(name, tmp):
    fn = os.path.join(DIR, FILES[name])
    if not fn.endswith(".gz"):
        return fn
    out = os.path.join(tmp, name + ".mtx")
    with gzip.open(fn, "rb") as f, open(out, "wb") as g:
        g.write(f.read())
    return out


def report(name, A):
    op = host.device_operator(A)
    op.autotune()
    info = op.info()
    M = info["M"]
    x = capi.DeviceVector(M, np.sin(0.001 * np.arange(M)))
    y, rhs = capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
    d = host.desc_arrays(A.desc())
    rl = d["nnzPerRow_local"]
    line = (f"{name:10s} rows {M:7d} nnz {info['nnz_local']:9d} row length min/mean/max {rl.min()}/{rl.mean():.1f}/{rl.max()} "
            f"blocks {info['row_blocks']} lanes/row {info['lanes_per_row']} kernel {op.variant()[1]}")
    for kind, label in ((0, "spmv"), (1, "jacobi")):
        op.time_kernel(kind, x, rhs, y, 20)
        us = op.time_kernel(kind, x, rhs, y, 500) * 1e3
        B = op.algorithmic_bytes(kind)
        line += f" | {label} {us:6.2f} us {B / us / 1e3:7.1f} GB/s"
    print(line, flush=True)


def main():
    capi.init(0)
    comm = host.Comm("gpu", "rccl")
    with tempfile.TemporaryDirectory() as tmp:
        for name in FILES:
            A = host.Matrix(comm)
            A.read_file(mtx(name, tmp)).assemble()
            report(name, A)
    for M, bw in ((300000, 63), (1000000, 31)):
        A = host.Matrix(comm).band_matrix(M, bw).assemble()
        report(f"band{M // 1000}k_{bw}", A)
    capi.finalize()


if __name__ == "__main__":
    main()
