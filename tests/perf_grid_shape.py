"""SpMV / Jacobi sweep on Poisson grids of one size and different shapes (development aid): how much of the fine level's
time is the re-fetch of x lines whose reuse distance (one grid plane of rows) exceeds the 4 MiB L2 of an XCD.
    python -m tests.perf_grid_shape"""
import numpy as np

import os
os.environ.setdefault('SAENA_KEEP_HOST_VALUES', '1')
from saena_amd import capi, host


def main():
    capi.init(0)
    for shape in ((256, 256, 256), (256, 34, 2018), (256, 10, 8066)):
        A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(*shape).assemble()
        op = host.device_operator(A)
        M = A.num_local_rows
        x, y, rhs = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        cells = []
        for v in (9, 11):
            op.set_variant(v); op.set_lanes_per_row(1)
            for kind, name in ((0, "spmv"), (1, "jacobi")):
                op.time_kernel(kind, x, rhs, y, 5)
                cells.append(f"v{v} {name} {op.time_kernel(kind, x, rhs, y, 40) * 1e3:6.1f} us")
        plane = (shape[0] - 2) * (shape[1] - 2)
        print(f"grid {shape}: {M} rows, plane of {plane} rows = {plane * 88 / 1e6:.1f} MB of stream | " + " | ".join(cells), flush=True)
        del op, A


if __name__ == "__main__":
    main()
