"""configs[4] at an HBM-bound size (tests/irregular.py), every kernel form that accepts it (development aid):
python -m tests.perf_irregular [blocks] [variants] [lanes]   -> SpMV / Jacobi time per (variant, lanes), interleaved over 3 trials"""
import os
import sys

import numpy as np

os.environ.setdefault("SAENA_KEEP_HOST_VALUES", "1")
from saena_amd import capi, host
from tests import irregular


def main():
    nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 10, 12, 9]
    lanes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
    capi.init(0)
    print("device:", capi.device_info(), flush=True)
    r, c, v, M = irregular.sih4_replicated(nblocks, hub_every=0 if os.environ.get("IRREGULAR_NO_HUBS") else 16)      # (experiment: the operator without its hub rows)
    A = host.Matrix(host.Comm("gpu", "rccl"))
    A.set_remove_boundary(False)
    A.set_many(r, c, v)
    A.assemble()
    op = host.device_operator(A)
    x, y, rhs = capi.DeviceVector(M, np.sin(0.001 * np.arange(M))), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
    B = [op.algorithmic_bytes(0), op.algorithmic_bytes(1)]
    print(f"{M} rows, {len(r)} entries, algorithmic bytes {B}", flush=True)
    op.set_variant(0); op.set_lanes_per_row(1)
    op.spmv(x, y)
    ref = y.download()
    for trial in range(3):
        for vv in variants:
            try:
                op.set_variant(vv)
            except capi.SgpuError as e:
                if trial == 0:
                    print(f"v{vv} refused: {e}", flush=True)
                continue
            for ln in lanes:
                try:
                    op.set_lanes_per_row(ln)
                except capi.SgpuError:
                    continue
                if trial == 0:
                    op.spmv(x, y)
                    err = np.max(np.abs(y.download() - ref)) / np.max(np.abs(ref))
                    print(f"v{vv} lanes {ln}: max rel diff to the sequential sum {err:.2e}", flush=True)
                    continue
                op.time_kernel(0, x, rhs, y, 5)
                us = op.time_kernel(0, x, rhs, y, 40) * 1e3
                usj = op.time_kernel(1, x, rhs, y, 40) * 1e3
                print(f"trial {trial} v{vv} {op.variant()[1]} lanes {ln}: spmv {us:7.1f} us {B[0] / us / 1e3:6.0f} GB/s | jacobi {usj:7.1f} us {B[1] / usj / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
