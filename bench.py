#!/usr/bin/env python3
"""bench.py -- fine-level SpMV of the Saena V-cycle hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is
launched by torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* from the env).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 3D 7-point Poisson 128^3 -> 126^3 = 2 000 376
rows, 13 907 376 nnz, fp64 values / int32 indices, operator and vectors resident
in HBM.  A step = one fine-level SpMV w = A v through sgpu_spmv (the autotuned HIP kernel,
k_csr_cc16 here; for N>1 interior rows on the compute stream, pack + RCCL send/recv + boundary
rows on the halo stream).  value = algorithmic bytes of all ranks' SpMVs
/ wall time.  N>1 is weak scaling: every rank owns a 126-plane z-slab of a 128 x 128 x (126 N + 2)
grid (2 000 376 rows per rank), neighbours exchange one 126^2 plane per side.

Extra objects: `roofline` (HBM bound; kernel time from HIP events recorded on the compute stream
around the timed launches), `check` (one more SpMV, outside the timed region, against the
host-formed product incl. halo values), `vcycle` (pCG iterations/s and V-cycles/s, host-built
hierarchy; for N>1 a strong-scaled leg on the global 128^3 problem and `vcycle_weak` on the
weak-scaled operator) and `cpu_baseline` (the compiled reference's own matvec under mpirun on this
box's cores -- oracle/_ref, test infrastructure -- or the oracle's restatement as fallback;
rank 0, N=1 only; never part of the measured path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver stack

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--m", type=int, default=128, help="grid points per side (reference laplacian3D argument)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vcycle", action="store_true", help="skip the V-cycle / pCG leg (host AMG setup takes ~15 s)")
    ap.add_argument("--vcycle-timeout", type=float, default=360.0, help="watchdog of the multi-rank V-cycle legs, seconds")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline_reference(m, seconds, cores):
    """The reference's own saena_matrix::matvec, compiled from its sources (oracle/ref/Makefile -> oracle/_ref/ref_dump,
    built where /root/reference exists and shipped as a prebuilt binary), one MPI rank per core.  None if unavailable."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    if not (os.path.exists(exe) and os.path.exists(mpirun)):
        return None
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL")
    try:
        out = subprocess.run([mpirun, "-np", str(cores), exe, "/tmp", "time", str(m), str(seconds)], env=env,
                             capture_output=True, text=True, timeout=60 + 6 * seconds)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("REF_TIME_MATVEC")][-1].split()
        t, reps, ranks, nrm, Mbig, nnz = float(line[1]), int(line[2]), int(line[3]), float(line[4]), int(line[5]), int(line[6])
    except Exception:                                       # noqa: BLE001 -- fall back to the port
        return None
    if m == 128 and abs(nrm - 4185009626440892.5) > 1e-9 * nrm:      # ||A v||^2 known answer
        return None
    B = 12 * nnz + 4 * (Mbig + 1) + 16 * Mbig
    return {"value": round(B / t / 1e9, 3), "unit": "GB/s", "cores": ranks, "kind": "reference",
            "sample": f"{reps} saena_matrix::matvec calls of the compiled reference (oracle/_ref/ref_dump, -Ofast) on the same "
                      f"Poisson {m}^3 operator, mpirun -np {ranks} (one rank per core, nnz-balanced partition), {t * 1e3:.3f} ms each"}


def cpu_baseline(m, seconds):
    """CPU path timed beside the GPU one on this box's cores: the compiled reference when its prebuilt binary is here,
    else the oracle (CPU restatement of saena_matrix::matvec): P simulated MPI ranks on P threads, the reference's
    default of one thread per rank."""
    import numpy as np
    cores = max(1, min(os.cpu_count() or 1, 16))
    ref = cpu_baseline_reference(m, seconds, cores)
    if ref is not None:
        return ref
    from oracle import oracle as orc
    entries, Mbig = orc.laplacian3d(m)
    split = orc.split_nnz(entries, Mbig, cores)
    A = orc.OracleOp(entries, Mbig, Mbig, split)
    v = np.sin(0.001 * np.arange(Mbig))
    t1 = A.time_matvec(v, 3, cores)
    reps = max(5, int(seconds / max(t1, 1e-6)))
    t = A.time_matvec(v, reps, cores)
    nnz = len(entries)
    B = 12 * nnz + 4 * (Mbig + 1) + 16 * Mbig
    return {"value": round(B / t / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{reps} matvecs of the same Poisson {m}^3 operator, {cores} simulated ranks on {cores} threads "
                      f"(oracle/saena_oracle.c, -O2), {t * 1e3:.3f} ms each"}


def vcycle_leg(capi, host, A, m, dist=None):
    """Second half of BASELINE.json's metric: V-cycle iterations/s of solve_pCG (options001: Jacobi 3+3,
    tol 1e-8) on the same operator, hierarchy from the host SA setup, everything device-resident.
    With more than one rank every rank calls this (the solve is collective over RCCL); the times are
    rank 0's between barriers."""
    import ctypes as C
    import numpy as np
    L = host.load("gpu")

    def barrier():
        capi.check(capi.lib().sgpu_barrier())
        if dist is not None:
            dist.barrier()
    t0 = time.perf_counter()
    S = host.AmgSolver(A, host.options(L, **host.OPTIONS001)).to_device()
    t_setup = time.perf_counter() - t0
    M = A.num_local_rows
    du, dr = capi.DeviceVector(M), capi.DeviceVector(M, A.laplacian3D_rhs())
    h = S.device_handle()
    it = C.c_int()
    hist = np.full(64, np.nan)
    lib = capi.lib()
    PD = C.POINTER(C.c_double)
    best = None
    for _ in range(3 if dist is None else 2):            # first pass warms up; keep the best of the rest
        barrier()
        t0 = time.perf_counter()
        st = lib.sgpu_solve_pCG(h, du.ptr, dr.ptr, C.byref(it), hist.ctypes.data_as(PD), 64)
        capi.check(lib.sgpu_device_sync())
        dt = time.perf_counter() - t0
        if st != 0:
            capi.check(st)
        best = dt if best is None else min(best, dt)
    hh = hist[~np.isnan(hist)]
    for _ in range(3):
        capi.check(lib.sgpu_vcycle(h, du.ptr, dr.ptr))
    barrier()
    n = 20 if dist is None else 10
    t0 = time.perf_counter()
    for _ in range(n):
        capi.check(lib.sgpu_vcycle(h, du.ptr, dr.ptr))
    barrier()
    t_v = (time.perf_counter() - t0) / n
    levels = [S.level_info(l) for l in range(S.num_levels)]
    return {"levels": S.num_levels, "rows": [x["rows"] for x in levels], "nnz": [x["nnzA"] for x in levels],
            "pcg_iterations": it.value, "pcg_iterations_per_s": round(it.value / best, 2), "pcg_solve_ms": round(best * 1e3, 3),
            "vcycles_per_s": round(1.0 / t_v, 2), "vcycle_ms": round(t_v * 1e3, 4),
            "initial_residual": float(hh[0]), "final_residual": float(hh[-1]), "relative_residual": float(hh[-1] / hh[0]),
            "options": "data/options001.xml values: jacobi 3+3, tol 1e-8, conn_str 0.2", "host_setup_s": round(t_setup, 2)}


def verify_spmv(np, host, A, op, x, y, g0):
    """One more SpMV, checked against the same product formed on the host from this rank's layout arrays with the
    closed-form input x(g) = sin(0.001 g) -- halo values included, so with more than one rank this checks what the
    RCCL exchange delivered.  -> (max |y_gpu - y_host| / max sum_j |a_ij x_j|, rows checked)"""
    op.spmv(x, y)
    got = y.download()
    d = host.desc_arrays(A.desc())
    M = d["M"]
    f = lambda g: np.sin(0.001 * g)                                          # noqa: E731
    rows = np.repeat(np.arange(M), d["nnzPerRow_local"])
    want = np.zeros(M)
    bound = np.zeros(M)
    t = d["val_local"] * f(d["col_local"].astype(np.float64))              # col_local holds GLOBAL ids
    np.add.at(want, rows, t)
    np.add.at(bound, rows, np.abs(t))
    hc = A.halo_columns()
    if len(hc):
        slot = np.repeat(np.arange(len(hc)), d["nnzPerCol_remote"])
        t = d["val_remote"] * f(hc[slot].astype(np.float64))
        np.add.at(want, d["row_remote"], t)
        np.add.at(bound, d["row_remote"], np.abs(t))
    err = float(np.max(np.abs(got - want)) / max(float(np.max(bound)), 1e-300)) if M else 0.0
    return err, int(M)


def pmc_traffic(m, world, kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (PMC counters cannot be
    read from inside the process).  Only for the kernel those passes profiled; None for anything else."""
    table = {"k_csr_cc16<16KiB>": "r01_pmc_spmv_128_cc16.json", "k_csr_stream<16KiB>": "r01_pmc_spmv_128.json"}
    name = table.get(kernel_name)
    path = os.path.join(ROOT, "profiles", name) if name else None
    if m == 128 and world == 1 and path and os.path.exists(path):
        with open(path) as f:
            return json.load(f)["traffic_bytes_per_launch"], "profiles/" + name
    return None, None


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """Libraries (gloo, RCCL) print banners on stdout while they connect: stdout is reserved for the ONE JSON line."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        ctypes.CDLL(None).fflush(None)                   # C stdio buffers when stdout is a pipe or a file
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _init_context(capi, world, rank, device, uid, dist):
    if world > 1 and os.environ.get("SAENA_BENCH_NO_RCCL"):
        # rehearsal of the N > 1 run on ONE card (RCCL refuses several ranks per device): the multi-rank code runs
        # with halos and reductions routed through the host and gloo (sgpu_debug_init_host_transport) -- residuals
        # are the real ones, timings mean nothing
        capi.init_host_transport(device, dist)
    else:
        capi.init(device=device, rank=rank, nranks=world, unique_id=uid)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if world > 1:      # the ranks share this host's cores: split them for the threaded host setup
        os.environ.setdefault("SAENA_SETUP_THREADS", str(max(2, min(16, (os.cpu_count() or 16) // world))))
    dist = None
    if world > 1 or os.environ.get("SAENA_BENCH_IMPORT_TORCH"):
        # torch first: its bundled HIP/RCCL runtime must be the one both sides use
        import torch  # noqa: F401
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with stdout_to_stderr():
                dist.init_process_group("gloo", rank=rank, world_size=world)   # CPU rendezvous only; data rides RCCL

    import numpy as np
    from saena_amd import capi, host

    uid = None
    # SAENA_BENCH_FORCE_COMM=1: one rank WITH an RCCL communicator (self send/recv), to rehearse the N>1 code on one GPU
    multi = world > 1 or bool(os.environ.get("SAENA_BENCH_FORCE_COMM"))
    if world > 1:
        box = [capi.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    elif multi:
        uid = capi.get_unique_id()
    # SAENA_BENCH_DEVICE: rehearsal aid (several ranks on one card, if the RCCL build allows it)
    device = int(os.environ.get("SAENA_BENCH_DEVICE", local_rank))
    with stdout_to_stderr():
        _init_context(capi, world, rank, device, uid, dist)

    # ---- operator through the host mirror of saena::matrix (product path, no oracle) ----
    m = args.m
    # Setup-time collectives (assemble, hierarchy distribution): over the RCCL communicator at one rank; with more
    # ranks over the gloo group already open for the rendezvous -- that path is exercised at world_size 2-4 by the
    # CPU test-suite, the RCCL one (RcclHostComm) only at one rank so far.  SAENA_BENCH_SETUP_COMM=rccl overrides.
    # The data path (halo exchange, dots) rides RCCL either way.
    if world > 1 and os.environ.get("SAENA_BENCH_SETUP_COMM", "gloo") != "rccl":
        comm = host.Comm("gpu", "dist", dist)
    else:
        comm = host.Comm("gpu", "rccl")
    A = host.Matrix(comm)
    if world == 1:
        A.laplacian3D(m).assemble()                      # reference partitioner (trivial at one rank)
    else:
        n = m - 2
        A.laplacian3D(m, m, n * world + 2)
        split = np.array([r * n * n * n for r in range(world + 1)], np.int32)
        A.assemble(split)                                # even z-slabs: 126 planes per rank
    op = host.device_operator(A)
    if os.environ.get("SAENA_BENCH_VARIANT"):            # pin the kernel (rocprofv3 --pmc perturbs the autotune's timings)
        op.set_variant(int(os.environ["SAENA_BENCH_VARIANT"]))
    else:
        op.autotune()                                    # plan-time choice among the kernel variants
    info = op.info()
    variant, kernel_name = op.variant()
    M = info["M"]
    g0 = int(A.split[rank])
    x = capi.DeviceVector(M, np.sin(0.001 * (g0 + np.arange(M))))
    y = capi.DeviceVector(M)
    B_local = op.algorithmic_bytes(0)

    def sync_all():
        capi.check(capi.lib().sgpu_barrier())
        if dist is not None:
            dist.barrier()

    # ---- warm-up, then EXACTLY K timed steps between barriers ----
    for _ in range(args.warmup):
        op.spmv(x, y)
    sync_all()
    t0 = time.perf_counter()
    ms_kernel = op.time_kernel(0, x, None, y, args.steps)   # K launches, HIP events on the compute stream
    capi.check(capi.lib().sgpu_device_sync())
    sync_all()
    wall = time.perf_counter() - t0

    err, _ = verify_spmv(np, host, A, op, x, y, g0)        # outside the timed region
    B_total = B_local
    if dist is not None:
        import torch
        e = torch.tensor([err], dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        err = float(e[0])
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
        b = torch.tensor([float(B_local)], dtype=torch.float64)
        dist.all_reduce(b)
        B_total = float(b[0])

    if rank == 0:
        sec_per_step = wall / args.steps
        achieved = B_local / (ms_kernel * 1e-3) / 1e9
        out = {
            "metric": "fine-level SpMV effective GB/s (3D 7-pt Poisson 128^3, fp64)",
            "value": round(B_total / sec_per_step / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"Poisson {m}^3 (Saena laplacian3D, boundary rows removed): SpMV w=Av, "
                            f"{info['M']} rows x {info['nnz_local'] + info['nnz_remote']} nnz per GPU, int32 indices",
                "rows_per_gpu": info["M"], "nnz_per_gpu": info["nnz_local"] + info["nnz_remote"],
                "partition": "1 rank" if world == 1 else f"{world} even z-slabs, RCCL halo of {n * n} doubles per side",
                "pct_of_hbm_peak": round(B_total / sec_per_step / 1e9 / (HBM_PEAK_GBS * world) * 100, 2),
            },
            "check": {"what": "y = A x of one more SpMV against the host-formed product from the layout arrays, halo values included; "
                              "max over all ranks of max_i |y_gpu - y_host| / max_i sum_j |a_ij x_j|",
                      "max_rel_err": err, "ok": bool(err <= 1e-13)},
            "roofline": {
                "bound": "hbm", "kernel": f"{kernel_name}, {info['lanes_per_row']} lane(s)/row",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "us_per_launch": round(ms_kernel * 1e3, 3), "algorithmic_bytes": B_local,
                "traffic": pmc_traffic(m, world, kernel_name)[0], "traffic_source": pmc_traffic(m, world, kernel_name)[1],
            },
        }
        if world == 1 and not multi and not args.no_vcycle:
            out["vcycle"] = vcycle_leg(capi, host, A, m)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, args.cpu_seconds)

    if multi and not args.no_vcycle:
        # V-cycle / pCG over all ranks, halos over RCCL.  Runs under a watchdog and a fatal-signal printer: whatever
        # happens in here, rank 0 still prints the SpMV line measured above.
        import threading

        legs = {}

        def line(err):
            if rank != 0:
                return ""
            o = dict(out)
            o.update(legs)
            o["vcycle_error"] = err
            return json.dumps(o)

        def bail():
            if rank == 0:
                print(line(f"multi-rank V-cycle legs did not finish within {args.vcycle_timeout:.0f} s"), flush=True)
            os._exit(0)
        dog = threading.Timer(args.vcycle_timeout, bail)
        dog.daemon = True
        dog.start()
        # a native crash (GPU fault -> abort, or SIGTERM from the launcher when a sibling rank died) still prints the line
        fatal = capi.lib().sgpu_debug_on_fatal_print
        capi.check(fatal(line("a multi-rank V-cycle leg ended with a fatal signal").encode()))
        try:
            # (1) parity: the SAME global Poisson m^3 problem as at N=1 (strong scaling), row blocks from the reference's
            #     nnz-balanced partitioner -- the residuals are comparable with the reference's printed digits
            A2 = host.Matrix(comm)
            A2.laplacian3D(m).assemble()
            leg = vcycle_leg(capi, host, A2, m, dist)
            leg["scaling"] = "strong"
            leg["partition"] = f"{world} nnz-balanced row blocks of the global Poisson {m}^3 operator"
            legs["vcycle"] = leg
            capi.check(fatal(line("the weak-scaled V-cycle leg ended with a fatal signal").encode()))
            # (2) performance: the weak-scaled operator of the SpMV measurement above (2 000 376 rows per GPU); every
            #     rank builds only its rows of the hierarchy (row-distributed setup)
            leg = vcycle_leg(capi, host, A, m, dist)
            leg["scaling"] = "weak"
            leg["partition"] = f"{world} even z-slabs of Poisson {m} x {m} x {(m - 2) * world + 2}"
            legs["vcycle_weak"] = leg
        except Exception as e:                              # noqa: BLE001 -- reported, never fatal for the SpMV line
            dog.cancel()
            if rank == 0:
                print(line(f"{type(e).__name__}: {e}"), flush=True)
            os._exit(0)
        dog.cancel()
        capi.check(fatal(None))
        if rank == 0:
            out.update(legs)

    if rank == 0:
        print(json.dumps(out), flush=True)

    with stdout_to_stderr():
        capi.finalize()
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
