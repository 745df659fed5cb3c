"""Time ONE operator of the hierarchy (for rocprofv3 --pmc passes): python -m tests.perf_one m level which kind reps [variant] [lanes]"""
import sys

import numpy as np

from saena_amd import capi, host


def main():
    m, level, which, kind, reps = (int(a) for a in sys.argv[1:6])
    variant = int(sys.argv[6]) if len(sys.argv) > 6 else -1
    lanes = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    capi.init(0)
    L = host.load("gpu")
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001)).to_device()
    op = S.device_op(level, which)
    if variant >= 0:
        op.set_variant(variant)
    if lanes:
        op.set_lanes_per_row(lanes)
    x, y, rhs = capi.DeviceVector(op.N_local, np.ones(op.N_local)), capi.DeviceVector(op.M), capi.DeviceVector(op.M, np.ones(op.M))
    op.time_kernel(kind, x, rhs, y, 3)
    us = op.time_kernel(kind, x, rhs, y, reps) * 1e3
    B = op.algorithmic_bytes(kind)
    print(f"L{level} which={which} kind={kind} {op.info()}: {us:.1f} us, {B} B -> {B / us / 1e3:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
