"""Several operators of one hierarchy in ONE process (for rocprofv3 --pmc passes: the setup is paid once per pass):
python -m tests.perf_ops m "level,which,kind,variant,lanes;level,which,..." reps
which: 0 A, 1 P, 2 R; kind: 0 spmv, 1 jacobi.  Every operator gets exactly 3 + reps launches of its kernel, in the order given --
tools/pmc_ops.sh finds them in the counter file as the runs of 3 + reps identical dispatches."""
import sys

import numpy as np

from saena_amd import capi, host


def main():
    m, reps = int(sys.argv[1]), int(sys.argv[3])
    ops = [tuple(int(t) for t in o.split(",")) for o in sys.argv[2].split(";") if o]
    capi.init(0)
    L = host.load("gpu")
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001)).to_device()
    for level, which, kind, variant, lanes in ops:
        op = S.device_op(level, which)
        if variant >= 0:
            op.set_variant(variant)
        if lanes:
            op.set_lanes_per_row(lanes)
        x, y, rhs = capi.DeviceVector(op.N_local, np.ones(op.N_local)), capi.DeviceVector(op.M), capi.DeviceVector(op.M, np.ones(op.M))
        op.time_kernel(kind, x, rhs, y, 3)
        us = op.time_kernel(kind, x, rhs, y, reps) * 1e3
        B = op.algorithmic_bytes(kind)
        i = op.info()
        print(f"OP L{level} which={which} kind={kind} {op.variant()[1]} lanes={i['lanes_per_row']} rows={i['M']} nnz={i['nnz_local']}: {us:.1f} us, {B} B -> {B / us / 1e3:.0f} GB/s", flush=True)
        for v in (x, y, rhs):
            v.free()


if __name__ == "__main__":
    main()
