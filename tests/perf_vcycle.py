"""V-cycle timing probe (development aid): python -m tests.perf_vcycle [m] [smoother]

Host SA setup -> device hierarchy -> per-level kernel table (GB/s of every operator
of the V-cycle) + pCG solve rate.  Product path only (no oracle).
"""
import sys
import time

import numpy as np

from saena_amd import capi, host


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    smoother = sys.argv[2] if len(sys.argv) > 2 else "jacobi"
    capi.init(0)
    print("device:", capi.device_info(), flush=True)
    L = host.load("gpu")
    t0 = time.time()
    A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(m).assemble()
    t1 = time.time()
    S = host.AmgSolver(A, host.options(L, **dict(host.OPTIONS001, smoother=smoother)))
    t2 = time.time()
    S.to_device()
    t3 = time.time()
    print(f"assemble {t1 - t0:.1f}s  setup {t2 - t1:.1f}s  upload {t3 - t2:.1f}s  levels {S.num_levels}", flush=True)
    nl = S.num_levels
    tot_us = 0.0
    tot_bytes = 0
    for l in range(nl):
        info = S.level_info(l)
        opA = S.device_op(l, 0)
        M = opA.M
        x, y, rhs = capi.DeviceVector(M, np.ones(M)), capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
        kind = 1 if smoother == "jacobi" else 3
        opA.time_kernel(kind, x, rhs, y, 5)
        us_s = opA.time_kernel(kind, x, rhs, y, 50) * 1e3
        us_r = opA.time_kernel(2, x, rhs, y, 50) * 1e3
        Bs, Br = opA.algorithmic_bytes(kind), opA.algorithmic_bytes(2)
        line = (f"L{l}: rows {info['rows']:8d} nnz {info['nnzA']:9d} ({info['nnzA'] / info['rows']:6.1f}/row) G={opA.info()['lanes_per_row']:2d} v{opA.variant()[0]:2d} "
                f"| smooth {us_s:8.1f} us {Bs / us_s / 1e3:7.1f} GB/s | resid {us_r:8.1f} us {Br / us_r / 1e3:7.1f} GB/s")
        lvl_us, lvl_b = 0.0, 0
        if l < nl - 1:
            opP, opR = S.device_op(l, 1), S.device_op(l, 2)
            Mc = opR.M
            xc, yc = capi.DeviceVector(Mc, np.ones(Mc)), capi.DeviceVector(Mc)
            opP.time_kernel(0, xc, None, y, 5); opR.time_kernel(0, x, None, yc, 5)
            us_p = opP.time_kernel(0, xc, None, y, 50) * 1e3
            us_t = opR.time_kernel(0, x, None, yc, 50) * 1e3
            Bp, Bt = opP.algorithmic_bytes(0), opR.algorithmic_bytes(0)
            line += (f" | P {us_p:7.1f} us {Bp / us_p / 1e3:7.1f} GB/s (G={opP.info()['lanes_per_row']} v{opP.variant()[0]}, {opP.info()['nnz_local'] / opP.M:.1f}/row) | R {us_t:7.1f} us {Bt / us_t / 1e3:7.1f} GB/s (G={opR.info()['lanes_per_row']} v{opR.variant()[0]}, {opR.info()['nnz_local'] / opR.M:.1f}/row)")
            # (3,3) sweeps; on every coarse level the first pre-smoothing sweep starts from u = 0 and needs no pass over
            # the matrix (k_zero_sweep, 24 B/row), which is what sgpu_vcycle runs
            full = 6 if l == 0 else 5
            lvl_us = full * us_s + us_r + us_p + us_t
            lvl_b = full * Bs + Br + Bp + Bt + (0 if l == 0 else 24 * M)
        tot_us += lvl_us
        tot_bytes += lvl_b
        print(line + f" | level total {lvl_us:8.1f} us", flush=True)
    print(f"sum of matrix kernels per (3,3) V-cycle (arbitrary fine iterate): {tot_us:.1f} us, algorithmic {tot_bytes / 1e9:.3f} GB -> {tot_bytes / tot_us / 1e3:.1f} GB/s", flush=True)

    rhs = A.laplacian3D_rhs()
    for rep in range(2):
        t0 = time.time()
        u, it, hist, conv = S.solve_pCG(rhs)
        dt = time.time() - t0
        print(f"solve_pCG: {it} its, conv={conv}, {dt * 1e3:.1f} ms (incl. H2D/D2H) -> {it / dt:.1f} V-cycle-iterations/s; "
              f"r0={hist[0]:.6e} rN={hist[-1]:.6e} rel={hist[-1] / hist[0]:.6e}", flush=True)
    # raw V-cycles on device vectors
    M = A.num_local_rows
    du, dr = capi.DeviceVector(M, np.zeros(M)), capi.DeviceVector(M, rhs)
    h = S.device_handle()
    for _ in range(3):
        capi.check(capi.lib().sgpu_vcycle(h, du.ptr, dr.ptr))
    capi.check(capi.lib().sgpu_device_sync())
    n = 20
    t0 = time.time()
    for _ in range(n):
        capi.check(capi.lib().sgpu_vcycle(h, du.ptr, dr.ptr))
    capi.check(capi.lib().sgpu_device_sync())
    dt = (time.time() - t0) / n
    print(f"raw V-cycle: {dt * 1e3:.3f} ms -> {1 / dt:.1f} V-cycles/s", flush=True)
    nc = S.level_info(S.num_levels - 1)["rows"]
    cu, cr = capi.DeviceVector(nc, np.zeros(nc)), capi.DeviceVector(nc, np.ones(nc))
    import ctypes as C
    t0 = time.time()
    for _ in range(n):
        capi.check(capi.lib().sgpu_coarsest_solve(h, cu.ptr, cr.ptr, None))
    capi.check(capi.lib().sgpu_device_sync())
    print(f"coarsest solve: {(time.time() - t0) / n * 1e6:.1f} us", flush=True)


if __name__ == "__main__":
    main()
