#!/usr/bin/env python3
"""bench.py -- fine-level SpMV of the Saena V-cycle hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is
launched by torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* from the env).  Rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json configs[1]): 3D 7-point Poisson 128^3 -> 126^3 = 2 000 376
rows, 13 907 376 nnz, fp64 values / int32 indices, operator and vectors resident
in HBM.  A step = one fine-level SpMV w = A v through sgpu_spmv (the autotuned HIP kernel,
k_csr_cc16 here; for N>1 interior rows on the compute stream, pack + RCCL send/recv + boundary
rows on the halo stream).  value = algorithmic bytes of all ranks' SpMVs
/ wall time.
Workload at N>1 (BASELINE.json configs[3]): Poisson 512^3 row-partitioned by the reference's
nnz-balanced partitioner (src/saena_matrix_repart.cpp:43-170).  At N=8 that is the whole 512^3
operator (132 651 000 rows, 16.6 M rows / 116 M nnz per GPU); at N=2 and 4 it is the cube that keeps
those 16.6 M rows per GPU (323^3 and 407^3: weak scaling, isotropic like the 512^3 problem itself),
neighbours exchange about one plane of the cube per side.

Extra objects: `roofline` (HBM bound; kernel time from HIP events recorded on the compute stream
around the timed launches; states whether the working set is Infinity-Cache resident),
`spmv_hbm_resident` (N=1: the same measurement on Poisson 256^3, 1.7 GB, beyond the 256 MiB cache;
per-GPU work of configs[3]), `check` (one more SpMV, outside the timed region, against the
host-formed product incl. halo values), `vcycle` (pCG iterations/s and V-cycles/s on the global
128^3 problem, host-built hierarchy; strong-scaled for N>1), `vcycle_config4` (N>1: the same on the configs[3]
operator, every rank building its rows of the hierarchy, with the residual of the returned iterate recomputed on the host)
and `cpu_baseline` (the compiled reference's own matvec under mpirun on
this box's cores -- oracle/_ref, test infrastructure -- or the oracle's restatement as fallback;
rank 0, N=1 only; never part of the measured path).

Exit status: 0 only when every requested leg ran; a watchdog expiry, an exception or a fatal
signal in a multi-rank leg still prints the measured line (with `vcycle_error`) and then exits 3 /
128+signal, so the launcher sees the failure.
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
    ap.add_argument("--grid-m", "--m", dest="m", type=int, default=None, help="grid points per side (reference laplacian3D argument); default 128 at "
                    "--gpus 1 (BASELINE configs[1]) and 512 at --gpus N>1 (configs[3] at N=8; at N=2/4 the cube with the same rows per GPU)")
    ap.add_argument("--hbm-m", "--m-hbm", dest="m_hbm", type=int, default=256, help="grid of the secondary HBM-resident SpMV figure at --gpus 1 (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vcycle", action="store_true", help="skip the V-cycle / pCG leg (host AMG setup takes ~15 s)")
    ap.add_argument("--vcycle-timeout", type=float, default=480.0, help="watchdog of the multi-rank V-cycle legs, seconds")
    ap.add_argument("--config4-vcycle", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--no-config4-vcycle", action="store_true",
                    help="N>1: skip the V-cycle / pCG leg on the configs[3] operator itself (its row-distributed host setup of 16.6 M rows "
                         "per rank is the longest part of the run: DESIGN.md 5 has the per-phase budget)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the cpu_baseline sample")
    ap.add_argument("--spmv-timeout", type=float, default=300.0, help="N>1: watchdog of the SpMV measurement itself, seconds")
    ap.add_argument("--assemble-timeout", type=float, default=300.0, help="N>1: watchdog of the rendezvous + assemble phase before it, seconds")
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


def vcycle_leg(capi, host, A, m, dist=None, check_residual=False):
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
    crit = {}
    if check_residual:
        # the criterion that needs no CPU reference at this size: ||A u - rhs|| of the returned iterate, formed ON THE HOST
        # from this rank's layout arrays (halo values of u fetched from their owners over the rendezvous group), against
        # tol ||rhs||; the device's own last residual must agree with it
        capi.check(lib.sgpu_solve_pCG(h, du.ptr, dr.ptr, C.byref(it), hist.ctypes.data_as(PD), 64))
        capi.check(lib.sgpu_device_sync())
        r2, b2 = host_residual_sq(np, host, A, du.download(), dr.download(), dist)
        crit = {"residual_check": {"what": "||A u - rhs||_2 of the returned iterate recomputed on the host from the layout arrays (halo of u "
                                           "exchanged over the rendezvous group), relative to ||rhs||_2; criterion: <= 2 x solver_tol (the stopping "
                                           "test is on the recursively updated residual)",
                                   "host_relative_residual": float(np.sqrt(r2 / b2)), "device_relative_residual": float(hh[-1] / hh[0]),
                                   "tol": 1e-8, "ok": bool(np.sqrt(r2 / b2) <= 2e-8)}}
    return {**crit, "levels": S.num_levels, "rows": [x["rows"] for x in levels], "nnz": [x["nnzA"] for x in levels],
            "pcg_iterations": it.value, "pcg_iterations_per_s": round(it.value / best, 2), "pcg_solve_ms": round(best * 1e3, 3),
            "vcycles_per_s": round(1.0 / t_v, 2), "vcycle_ms": round(t_v * 1e3, 4),
            "initial_residual": float(hh[0]), "final_residual": float(hh[-1]), "relative_residual": float(hh[-1] / hh[0]),
            "residual_history": [float(x) for x in hh],
            "options": "data/options001.xml values: jacobi 3+3, tol 1e-8, conn_str 0.2", "host_setup_s": round(t_setup, 2)}


def host_residual_sq(np, host, A, u, rhs, dist):
    """(sum_i (A u - rhs)_i^2, sum_i rhs_i^2) over all ranks, A u formed on the host from this rank's layout arrays -- the
    loops of saena_matrix::matvec_sparse (src/saena_matrix_matvec.cpp:9-113): pack u[vIndex], exchange with the neighbours
    (here: over the rendezvous group), local CSR part, remote CSC part."""
    d = host.desc_arrays(A.desc())
    M = d["M"]
    rows = np.repeat(np.arange(M), d["nnzPerRow_local"])
    g0 = int(A.split[dist.get_rank()]) if dist is not None else 0
    y = np.bincount(rows, weights=d["val_local"] * u[d["col_local"] - g0], minlength=M).astype(np.float64)
    if dist is not None:
        import torch
        send = u[d["vIndex"]] if len(d["vIndex"]) else np.zeros(0)
        reqs, bufs, so = [], [], 0
        for q, cnt in zip(d["sendProcRank"], d["sendProcCount"]):
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(send[so:so + cnt])), int(q)))
            so += int(cnt)
        for q, cnt in zip(d["recvProcRank"], d["recvProcCount"]):
            t = torch.empty(int(cnt), dtype=torch.float64)
            bufs.append(t)
            reqs.append(dist.irecv(t, int(q)))
        for rq in reqs:
            rq.wait()
        recv = np.concatenate([t.numpy() for t in bufs]) if bufs else np.zeros(0)
        if len(recv):
            slot = np.repeat(np.arange(len(recv)), d["nnzPerCol_remote"])
            y += np.bincount(d["row_remote"], weights=d["val_remote"] * recv[slot], minlength=M)
        t = torch.tensor([float(np.sum((y - rhs) ** 2)), float(np.sum(rhs ** 2))], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0]), float(t[1])
    return float(np.sum((y - rhs) ** 2)), float(np.sum(rhs ** 2))


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
    t = d["val_local"] * f(d["col_local"].astype(np.float64))              # col_local holds GLOBAL ids
    want = np.bincount(rows, weights=t, minlength=M).astype(np.float64)
    bound = np.bincount(rows, weights=np.abs(t), minlength=M).astype(np.float64)
    hc = A.halo_columns()
    if len(hc):
        slot = np.repeat(np.arange(len(hc)), d["nnzPerCol_remote"])
        t = d["val_remote"] * f(hc[slot].astype(np.float64))
        want += np.bincount(d["row_remote"], weights=t, minlength=M)
        bound += np.bincount(d["row_remote"], weights=np.abs(t), minlength=M)
    err = float(np.max(np.abs(got - want)) / max(float(np.max(bound)), 1e-300)) if M else 0.0
    return err, int(M)


def pmc_traffic(m, world, kernel_name):
    """Memory-side bytes per launch from the COMMITTED rocprofv3 PMC passes of this same command (PMC counters cannot
    be read from inside the process, so this is never a measurement of the present run: the line says so with
    `traffic_measured_in_run: false`).  Only for the kernel those passes profiled; None for anything else."""
    table = {"k_sellp": "r03_pmc_spmv_128_sellp.json", "k_sellp2": "r03_pmc_spmv_128_sellp2.json", "k_sell": "r02_pmc_spmv_128_sell.json", "k_csr_cc16<16KiB,4+12>": "r02_pmc_spmv_128_cc16.json",
             "k_csr_stream<16KiB>": "r01_pmc_spmv_128.json"}
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


def measure_spmv(capi, host, np, A, rank, steps, warmup, sync_all):
    """autotuned operator of A on the device; `warmup` untimed launches, then `steps` timed ones between barriers.
    -> dict(op, info, kernel_name, ms_kernel, wall, B_local, x, y)"""
    op = host.device_operator(A)
    if os.environ.get("SAENA_BENCH_VARIANT"):            # pin the kernel (rocprofv3 --pmc perturbs the autotune's timings)
        op.set_variant(int(os.environ["SAENA_BENCH_VARIANT"]))
    else:
        op.autotune()                                    # plan-time choice among the kernel variants
    info = op.info()
    _, kernel_name = op.variant()
    M = info["M"]
    g0 = int(A.split[rank])
    x = capi.DeviceVector(M, np.sin(0.001 * (g0 + np.arange(M))))
    y = capi.DeviceVector(M)
    for _ in range(warmup):
        op.spmv(x, y)
    sync_all()
    t0 = time.perf_counter()
    ms_kernel = op.time_kernel(0, x, None, y, steps)     # K launches, HIP events on the compute stream
    sync_all()                                           # device synchronize + barrier
    wall = time.perf_counter() - t0
    return dict(op=op, info=info, kernel_name=kernel_name, ms_kernel=ms_kernel, wall=wall, B_local=op.algorithmic_bytes(0),
                x=x, y=y, g0=g0)


def stored_bytes(info, kernel_name):
    """bytes the SpMV's operands occupy in HBM: values, column ids as the chosen kernel stores them, row pointers, x, y"""
    nnz = info["nnz_local"] + info["nnz_remote"]
    if kernel_name in ("k_sellp", "k_sellp2"):           # no column stream: a 16-bit pattern id per row (+ a table of a few hundred ints)
        return 8 * nnz + 2 * info["M"] + 8 * info["N_local"] + 8 * info["M"]
    if kernel_name == "k_sell":                          # 16-bit column codes, a 16-bit row length instead of the row pointer (padding < 1 % here)
        return 10 * nnz + 2 * info["M"] + 8 * info["N_local"] + 8 * info["M"]
    col_bytes = 2 if ("cc16" in kernel_name or "k_csr_cm" in kernel_name) else 4
    extra = 2 if "k_csr_cm" in kernel_name else 0       # k_csr_cm: + a 16-bit tile slot per entry
    return (8 + col_bytes + extra) * nnz + 4 * (info["M"] + 1) + 8 * info["N_local"] + 8 * info["M"]


INFINITY_CACHE_BYTES = 256 * 2 ** 20                    # MI355X: 256 MiB memory-side cache (MI355X_MICROARCH.md)


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
    m = args.m if args.m else (128 if world == 1 else 512)
    # Setup-time collectives (assemble, the row-distributed hierarchy): over the RCCL communicator at one rank; with more
    # ranks over the native shared-memory communicator (exercised at world_size 2-4 by the CPU test-suite).
    # SAENA_BENCH_SETUP_COMM=gloo routes them through the rendezvous group and Python callbacks instead, =rccl through
    # RcclHostComm (staged through the device).  The data path (halo exchange, dots) rides RCCL either way.
    def watchdog(seconds, what, code):
        import threading

        def bail():
            print(f"bench.py rank {rank}: {what} did not finish within {seconds:.0f} s", file=sys.stderr, flush=True)
            os._exit(code)
        t = threading.Timer(seconds, bail)
        t.daemon = True
        t.start()
        return t
    # (N > 1: a stall in the rendezvous or in the assemble's collectives must not look like a long run either)
    asm_dog = watchdog(args.assemble_timeout, "the rendezvous + assemble phase", 5) if world > 1 else None
    setup_comm = os.environ.get("SAENA_BENCH_SETUP_COMM", "shm")
    if world > 1 and setup_comm == "shm":
        # the job's ranks sit on one node (the launch contract): the setup's collectives are memory copies through the native
        # shared-memory communicator (saena_amd/csrc/host/shm_comm.cpp); the name is fresh for every job
        box = [f"{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}_{int(time.time() * 1e3) % 10 ** 9}" if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = host.Comm("gpu", "shm", (box[0], rank, world))
    elif world > 1 and setup_comm == "gloo":
        comm = host.Comm("gpu", "dist", dist)
    else:
        comm = host.Comm("gpu", "rccl")
    A = host.Matrix(comm)
    if world == 1 or world == 8:
        mg = m
    else:                                                # the cube that gives every GPU the rows it holds at N = 8: (N/8)^(1/3) (m - 2) interior points per side
        mg = int(round((world / 8.0) ** (1.0 / 3.0) * (m - 2))) + 2
    grid = (mg, mg, mg)
    A.laplacian3D(*grid).assemble()                      # reference partitioner: nnz-balanced contiguous row blocks
    if asm_dog is not None:
        asm_dog.cancel()
    grid_s = f"{mg}^3"

    def sync_all():
        capi.check(capi.lib().sgpu_barrier())
        if dist is not None:
            dist.barrier()

    # ---- warm-up, then EXACTLY K timed steps between barriers ----
    # (N > 1: the first exchange between two DEVICES happens in here -- it has only ever run through a 1-rank communicator
    #  and the host transport.  A hang must not look like a long run: a watchdog ends the process non-zero with a reason.)
    spmv_dog = None
    if world > 1:
        import threading

        def spmv_bail():
            print(f"bench.py rank {rank}: the multi-GPU SpMV measurement did not finish within {args.spmv_timeout:.0f} s "
                  "(halo exchange over RCCL hung?)", file=sys.stderr, flush=True)
            os._exit(4)
        spmv_dog = threading.Timer(args.spmv_timeout, spmv_bail)
        spmv_dog.daemon = True
        spmv_dog.start()
    R = measure_spmv(capi, host, np, A, rank, args.steps, args.warmup, sync_all)
    if spmv_dog is not None:
        spmv_dog.cancel()
    op, info, kernel_name, ms_kernel, wall, B_local = R["op"], R["info"], R["kernel_name"], R["ms_kernel"], R["wall"], R["B_local"]

    err, _ = verify_spmv(np, host, A, op, R["x"], R["y"], R["g0"])        # outside the timed region
    B_total = B_local
    rows_all, halo_all = [info["M"]], [info["nnz_remote"]]
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
        g = [None] * world
        dist.all_gather_object(g, (info["M"], info["nnz_remote"]))
        rows_all, halo_all = [x[0] for x in g], [x[1] for x in g]

    out = None
    if rank == 0:
        sec_per_step = wall / args.steps
        achieved = B_local / (ms_kernel * 1e-3) / 1e9
        ws = stored_bytes(info, kernel_name)
        traffic, traffic_src = pmc_traffic(m, world, kernel_name)
        out = {
            "metric": f"fine-level SpMV effective GB/s (3D 7-pt Poisson {grid_s}, fp64)",
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
            "device": capi.device_info(),
            "data": "synthetic",
            "config": {
                "workload": (f"Poisson {grid_s} (Saena laplacian3D, boundary rows removed): SpMV w=Av, "
                             f"{info['M']} rows x {info['nnz_local'] + info['nnz_remote']} nnz on rank 0, int32 indices"
                             + ("; BASELINE configs[1]" if world == 1 and m == 128 else "")
                             + (f"; BASELINE configs[3] (Poisson {m}^3 over 8 GPUs)" + ("" if world == 8 else
                                f" weak-scaled to {world} GPUs: the cube with the same {(m - 2) ** 3 // 8} rows per GPU")
                                if world > 1 else "")),
                "rows_per_gpu": rows_all, "nnz_per_gpu": info["nnz_local"] + info["nnz_remote"],
                "partition": "1 rank" if world == 1 else
                             f"{world} nnz-balanced contiguous row blocks (reference partitioner), RCCL halo of ~{grid[0] - 2}^2 doubles per side "
                             f"(remote nnz per rank {halo_all})",
                "pct_of_hbm_peak": round(B_total / sec_per_step / 1e9 / (HBM_PEAK_GBS * world) * 100, 2),
                **({"weak_scaling_reference": "the 1-GPU figure at THIS per-GPU size is `spmv_hbm_resident` of the N=1 line (Poisson 256^3, "
                                              "16.4 M rows, HBM-resident: 5.8-5.9 TB/s, profiles/r02_bench_n1.json); the N=1 `value` is the "
                                              "Infinity-Cache-resident configs[1] operator (8.1 TB/s) and not the denominator of a weak-scaling ratio"}
                   if world > 1 else {}),
            },
            "check": {"what": "y = A x of one more SpMV against the host-formed product from the layout arrays, halo values included; "
                              "max over all ranks of max_i |y_gpu - y_host| / max_i sum_j |a_ij x_j|",
                      "max_rel_err": err, "ok": bool(err <= 1e-13)},
            "roofline": {
                "bound": "hbm", "kernel": f"{kernel_name}, {1 if kernel_name in ('k_sell', 'k_sellp') else 0.5 if kernel_name == 'k_sellp2' else info['lanes_per_row']} lane(s)/row",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "us_per_launch": round(ms_kernel * 1e3, 3), "algorithmic_bytes": B_local,
                "working_set_bytes": ws, "stored_bytes_rate_gbs": round(ws / (ms_kernel * 1e-3) / 1e9, 2), "cache_resident": bool(ws <= INFINITY_CACHE_BYTES),
                "note": ("the operator and vectors fit the 256 MiB Infinity Cache: repeated launches are served by it, so `achieved` "
                         "can exceed what HBM alone sustains (~6.1 TB/s reads); see spmv_hbm_resident for the HBM-bound figure"
                         if ws <= INFINITY_CACHE_BYTES else "working set beyond the 256 MiB Infinity Cache: HBM-bound"),
                "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_run": False if traffic is not None else None,
            },
        }
    # free the big operator's vectors before the legs allocate theirs
    if world == 1 and not multi and args.m_hbm and args.m_hbm != m:
        # the same measurement on an operator beyond the Infinity Cache (Poisson 256^3: 16.4 M rows, 114 M nnz, 1.7 GB
        # algorithmic -- also the per-GPU share of configs[3])
        A3 = host.Matrix(comm)
        A3.laplacian3D(args.m_hbm).assemble()
        steps3, warm3 = max(50, args.steps // 8), max(10, args.warmup // 8)       # (0.3 ms each)
        R3 = measure_spmv(capi, host, np, A3, 0, steps3, warm3, sync_all)
        e3, _ = verify_spmv(np, host, A3, R3["op"], R3["x"], R3["y"], 0)
        a3 = R3["B_local"] / (R3["ms_kernel"] * 1e-3) / 1e9
        ws3 = stored_bytes(R3["info"], R3["kernel_name"])
        out["spmv_hbm_resident"] = {
            "workload": f"Poisson {args.m_hbm}^3: {R3['info']['M']} rows x {R3['info']['nnz_local']} nnz, same kernel path",
            "kernel": f"{R3['kernel_name']}, {1 if R3['kernel_name'] in ('k_sell', 'k_sellp') else 0.5 if R3['kernel_name'] == 'k_sellp2' else R3['info']['lanes_per_row']} lane(s)/row", "steps": steps3, "warmup": warm3,
            "us_per_launch": round(R3["ms_kernel"] * 1e3, 3), "algorithmic_bytes": R3["B_local"], "working_set_bytes": ws3,
            "stored_bytes_rate_gbs": round(ws3 / (R3["ms_kernel"] * 1e-3) / 1e9, 2),
            "cache_resident": bool(ws3 <= INFINITY_CACHE_BYTES), "achieved": round(a3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(a3 / HBM_PEAK_GBS, 4), "check_max_rel_err": e3,
        }
        # bytes leaving the L2 per launch from the COMMITTED PMC passes of this leg (tools/pmc_spmv_hbm.sh), same kernel only
        pmc_name = {"k_sellp": "r03_pmc_spmv_256_sellp.json", "k_sellp2": "r03_pmc_spmv_256_sellp2.json"}.get(R3["kernel_name"], "r02_pmc_spmv_256_cc16.json" if "k_csr_cc16" in R3["kernel_name"] else None)
        pmc3 = os.path.join(ROOT, "profiles", pmc_name) if pmc_name else None
        if args.m_hbm == 256 and pmc3 and os.path.exists(pmc3):
            with open(pmc3) as f:
                out["spmv_hbm_resident"].update(traffic=json.load(f)["traffic_bytes_per_launch"], traffic_source="profiles/" + pmc_name,
                                                traffic_measured_in_run=False)
        R3["op"].destroy()
        for k in ("x", "y"):
            R3[k].free()
        A3.free()
        del R3, A3

    if rank == 0:
        if world == 1 and not multi and not args.no_vcycle:
            out["vcycle"] = vcycle_leg(capi, host, A, m)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, args.cpu_seconds)

    if multi and not args.no_vcycle:
        # V-cycle / pCG over all ranks, halos over RCCL.  Runs under a watchdog and a fatal-signal printer: whatever
        # happens in here, rank 0 still prints the SpMV line measured above -- and the process then ends with a
        # NON-ZERO status (3 for a watchdog expiry or an exception, 128 + signal for a fatal signal), so the launcher
        # records the failure.
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
            msg = f"multi-rank V-cycle legs did not finish within {args.vcycle_timeout:.0f} s"
            if rank == 0:
                print(line(msg), flush=True)
            print(f"bench.py rank {rank}: {msg}", file=sys.stderr, flush=True)
            os._exit(3)
        dog = threading.Timer(args.vcycle_timeout, bail)
        dog.daemon = True
        dog.start()
        # a native crash (GPU fault -> abort) still prints the line, then exits 128 + signal
        fatal = capi.lib().sgpu_debug_on_fatal_print
        capi.check(fatal(line("the 128^3 multi-rank V-cycle leg ended with a fatal signal").encode()))
        try:
            # (1) parity: the SAME global Poisson 128^3 problem as at N=1 (strong scaling), row blocks from the reference's
            #     nnz-balanced partitioner -- the residuals are comparable with the reference's printed digits
            A2 = host.Matrix(comm)
            A2.laplacian3D(128 if m >= 128 else m).assemble()
            leg = vcycle_leg(capi, host, A2, m, dist)
            leg["scaling"] = "strong"
            leg["partition"] = f"{world} nnz-balanced row blocks of the global Poisson {128 if m >= 128 else m}^3 operator"
            legs["vcycle"] = leg
            A2.free()
            if args.no_config4_vcycle:
                legs["vcycle_config4"] = {"skipped": "--no-config4-vcycle"}
            else:
                capi.check(fatal(line("the configs[3] V-cycle leg ended with a fatal signal").encode()))
                # (2) configs[3]: the operator of the SpMV measurement above (16.6 M rows per GPU at m = 512); every
                #     rank builds only its rows of the hierarchy (row-distributed setup)
                leg = vcycle_leg(capi, host, A, m, dist, check_residual=True)
                leg["scaling"] = "weak"
                leg["partition"] = f"{world} nnz-balanced row blocks of Poisson {grid_s}"
                legs["vcycle_config4"] = leg
                if not leg["residual_check"]["ok"]:
                    raise RuntimeError(f"configs[3] V-cycle leg: host-recomputed relative residual {leg['residual_check']['host_relative_residual']:.3e} > 2e-8")
        except Exception as e:                              # noqa: BLE001 -- reported with the SpMV line, then a failing status
            dog.cancel()
            msg = f"{type(e).__name__}: {e}"
            if rank == 0:
                print(line(msg), flush=True)
            print(f"bench.py rank {rank}: multi-rank V-cycle leg failed: {msg}", file=sys.stderr, flush=True)
            os._exit(3)
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
