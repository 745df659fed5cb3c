"""Kernel timing probe (development aid, not a test): python -m tests.perf_probe [m] [reps]

Builds Poisson m^3 through the oracle's layout (test infrastructure), runs the
HIP kernels through the C ABI and prints effective GB/s per lanes-per-row
variant, interleaving the variants in one process.
"""
import sys
import time

import numpy as np

from oracle import oracle as orc
from saena_amd import capi
from tests import inputs, util


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    capi.init(0)
    t0 = time.time()
    entries, M = orc.laplacian3d(m)
    A = orc.OracleOp(entries, M, M, orc.split_even(M, 1))
    G = util.gpu_operator(A)
    print(f"setup {time.time() - t0:.1f}s  {G.info()}", flush=True)
    x = capi.DeviceVector(M, inputs.v_sin(M))
    y = capi.DeviceVector(M)
    rhs = capi.DeviceVector(M, np.ones(M))
    want = A.matvec(inputs.v_sin(M))
    for kind, name in ((0, "spmv"), (1, "jacobi"), (2, "residual"), (3, "cheby")):
        B = G.algorithmic_bytes(kind)
        for rnd in range(3):
            for lanes in (1, 2, 4):
                G.set_lanes_per_row(lanes)
                G.time_kernel(kind, x, rhs, y, 20)
                ms = G.time_kernel(kind, x, rhs, y, reps)
                print(f"{name:9s} lanes={lanes} round={rnd}: {ms * 1e3:8.2f} us  {B / ms / 1e6:8.1f} GB/s  ({B / ms / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
    G.set_lanes_per_row(1)
    G.spmv(x, y)
    print("bit-exact vs oracle:", np.array_equal(y.download(), want))


if __name__ == "__main__":
    main()
