"""Fine-level operator alone (development aid): python -m tests.perf_fine m[,m...] v1,v2,... [trials]
SpMV and Jacobi sweep of the Poisson m^3 operator on each kernel variant, interleaved; algorithmic GB/s next to each time."""
import os
import sys

import numpy as np

os.environ.setdefault("SAENA_KEEP_HOST_VALUES", "1")
from saena_amd import capi, host


def main():
    ms = [int(v) for v in sys.argv[1].split(",")]
    variants = [int(v) for v in sys.argv[2].split(",")]
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    capi.init(0)
    print("device:", capi.device_info(), flush=True)
    for m in ms:
        A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
        op = host.device_operator(A)
        M = A.num_local_rows
        x, y, rhs = capi.DeviceVector(M, np.sin(0.001 * np.arange(M))), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        B = [op.algorithmic_bytes(0), op.algorithmic_bytes(1)]
        ref = None
        for t in range(trials):
            cells = []
            for v in variants:
                try:
                    op.set_variant(v)
                except capi.SgpuError as e:
                    cells.append(f"v{v} refused ({e})")
                    continue
                op.set_lanes_per_row(1)
                if t == 0:
                    op.spmv(x, y)
                    got = y.download()
                    if ref is None:
                        ref = got
                    cells.append(f"v{v} {'bit-identical' if np.array_equal(got, ref) else 'DIFFERS max ' + str(np.max(np.abs(got - ref)))}")
                    continue
                for kind, name in ((0, "spmv"), (1, "jacobi")):
                    op.time_kernel(kind, x, rhs, y, 5)
                    us = op.time_kernel(kind, x, rhs, y, 40) * 1e3
                    cells.append(f"v{v} {name} {us:7.1f} us {B[kind] / us / 1e3:6.0f} GB/s")
            print(f"m={m} trial {t}: " + " | ".join(cells), flush=True)
        op.destroy(); A.free()


if __name__ == "__main__":
    main()
