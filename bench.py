#!/usr/bin/env python3
"""bench.py -- fine-level SpMV of the Saena V-cycle hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is
launched by torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* from the env).  Rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json configs[1]): 3D 7-point Poisson 128^3 -> 126^3 = 2 000 376
rows, 13 907 376 nnz, fp64 values / int32 indices, operator and vectors resident
in HBM.  A step = one fine-level SpMV w = A v through sgpu_spmv (the autotuned HIP kernel,
k_sellp here; for N>1 interior rows on the compute stream, pack + RCCL send/recv + boundary
rows on the halo stream).  value = algorithmic bytes of all ranks' SpMVs
/ wall time.
Workload at N>1 (BASELINE.json configs[3]): Poisson 512^3 row-partitioned by the reference's
nnz-balanced partitioner (src/saena_matrix_repart.cpp:43-170).  At N=8 that is the whole 512^3
operator (132 651 000 rows, 16.6 M rows / 116 M nnz per GPU); at N=2 and 4 it is the cube that keeps
those 16.6 M rows per GPU (323^3 and 407^3: weak scaling, isotropic like the 512^3 problem itself),
neighbours exchange about one plane of the cube per side.

Extra objects: `roofline` (kernel time from HIP events recorded on the compute stream around the timed launches; `bound` says
whether the working set is Infinity-Cache or HBM resident; the streaming ceiling of the operator's stored bytes is measured in the
same run -- `peak_measured`, `frac_of_measured` <= 1 -- and is what a cache-resident operator's `frac` is taken against),
`spmv_irregular` (N=1: BASELINE configs[4] at 1 M rows -- the reference's SiH4 replicated, tests/irregular.py), `vcycle_256` (N=1:
BASELINE configs[2]), at N>1 `config.partition_imbalance`, `value_balanced` / `balanced_partition` (the same measurement under the
opt-in finer row partition) and `vcycle_balanced_partition`,
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
    ap.add_argument("--irregular-blocks", type=int, default=200, help="--gpus 1: diagonal blocks of the configs[4] operator (SiH4 replicated with per-block "
                    "permutations, coupling and hub rows: tests/irregular.py; 200 -> 1 008 200 rows, 34.7 M entries; 0 = skip)")
    ap.add_argument("--no-vcycle", action="store_true", help="skip the V-cycle / pCG leg (host AMG setup takes ~15 s)")
    ap.add_argument("--no-config2-vcycle", action="store_true", help="--gpus 1: skip the V-cycle / pCG leg on Poisson 256^3 (BASELINE configs[2]; "
                    "host setup of its 10-level, 1.0 G-entry hierarchy ~20 s, create + plan ~5 s)")
    ap.add_argument("--vcycle-timeout", type=float, default=480.0, help="watchdog of the multi-rank V-cycle legs, seconds")
    ap.add_argument("--config4-vcycle", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--no-config4-vcycle", action="store_true",
                    help="N>1: skip the V-cycle / pCG leg on the configs[3] operator itself (its row-distributed host setup of 16.6 M rows "
                         "per rank is the longest part of the run: DESIGN.md 5 has the per-phase budget)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the cpu_baseline sample")
    ap.add_argument("--spmv-timeout", type=float, default=300.0, help="N>1: watchdog of the SpMV measurement itself, seconds")
    ap.add_argument("--assemble-timeout", type=float, default=300.0, help="N>1: watchdog of the rendezvous + assemble phase before it, seconds")
    ap.add_argument("--no-balanced", action="store_true",
                    help="N>1: skip the second SpMV measurement (and the second 128^3 leg) under the opt-in finer row partition (`value_balanced`)")
    ap.add_argument("--balanced-buckets", type=int, default=4096, help="N>1: row buckets of the opt-in finer partition (the reference uses nparts^2)")
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
    if dist is None:
        # algorithmic bytes of one (3,3) V-cycle as sgpu_vcycle runs it (BASELINE.md section 3 per operator; every coarse level's first
        # pre-smoothing sweep starts from u = 0 and is the 24 B/row zero sweep, not a pass over the matrix) -> its rate against the peak
        pre, post = host.OPTIONS001["preSmooth"], host.OPTIONS001["postSmooth"]
        tot = 0
        for l in range(S.num_levels - 1):
            opA, opP, opR = S.device_op(l, 0), S.device_op(l, 1), S.device_op(l, 2)
            full = pre + post if l == 0 else pre + post - 1
            tot += full * opA.algorithmic_bytes(1) + opA.algorithmic_bytes(2) + opP.algorithmic_bytes(0) + opR.algorithmic_bytes(0) + (0 if l == 0 else 24 * opA.M)
        crit["vcycle_algorithmic_bytes"] = int(tot)
        crit["vcycle_algorithmic_gbs"] = round(tot / t_v / 1e9, 1)
        crit["vcycle_frac_of_hbm_peak"] = round(tot / t_v / 1e9 / HBM_PEAK_GBS, 4)
    if dist is not None:      # where the levels live: ranks that own rows of each level (all of them -> every k-th -> rank 0: the agglomeration)
        crit["ranks_per_level"] = [int(np.count_nonzero(np.diff(S.level_split(l)) > 0)) for l in range(S.num_levels)]
    if check_residual:
        # the criterion that needs no CPU reference at this size: ||A u - rhs|| of the returned iterate, formed ON THE HOST
        # from this rank's layout arrays (halo values of u fetched from their owners over the rendezvous group), against
        # tol ||rhs||; the device's own last residual must agree with it
        capi.check(lib.sgpu_solve_pCG(h, du.ptr, dr.ptr, C.byref(it), hist.ctypes.data_as(PD), 64))
        capi.check(lib.sgpu_device_sync())
        r2, b2 = host_residual_sq(np, host, A, du.download(), dr.download(), dist)
        crit["residual_check"] = {"what": "||A u - rhs||_2 of the returned iterate recomputed on the host from the layout arrays (halo of u "
                                           "exchanged over the rendezvous group), relative to ||rhs||_2; criterion: <= 2 x solver_tol (the stopping "
                                           "test is on the recursively updated residual)",
                                   "host_relative_residual": float(np.sqrt(r2 / b2)), "device_relative_residual": float(hh[-1] / hh[0]),
                                   "tol": 1e-8, "ok": bool(np.sqrt(r2 / b2) <= 2e-8)}
    return {**crit, "levels": S.num_levels, "rows": [x["rows"] for x in levels], "nnz": [x["nnzA"] for x in levels],
            "pcg_iterations": it.value, "pcg_iterations_per_s": round(it.value / best, 2), "pcg_solve_ms": round(best * 1e3, 3),
            "vcycles_per_s": round(1.0 / t_v, 2), "vcycle_ms": round(t_v * 1e3, 4),
            "initial_residual": float(hh[0]), "final_residual": float(hh[-1]), "relative_residual": float(hh[-1] / hh[0]),
            "residual_history": [float(x) for x in hh],
            "options": "data/options001.xml values: jacobi 3+3, tol 1e-8, conn_str 0.2", "host_setup_s": round(t_setup, 2),
            "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 2 ** 20, 2)}


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


# committed rocprofv3 --pmc passes of this command, by operator size and kernel FAMILY (most specific first): a kernel the passes did
# not profile falls back to its family's file and the line says which kernel the counters belong to (round-3 review: a third pick of
# the autotune must not drop `traffic` silently)
PMC_FILES = {
    128: [("k_sellp2", "r03_pmc_spmv_128_sellp2.json"), ("k_sellp", "r03_pmc_spmv_128_sellp.json"), ("k_sell", "r02_pmc_spmv_128_sell.json"),
          ("k_csr_cc16", "r02_pmc_spmv_128_cc16.json"), ("k_csr_stream", "r01_pmc_spmv_128.json")],
    256: [("k_sellp2", "r03_pmc_spmv_256_sellp2.json"), ("k_sellp", "r03_pmc_spmv_256_sellp.json"), ("k_csr_cc16", "r02_pmc_spmv_256_cc16.json")],
}


def pmc_traffic(m, world, kernel_name):
    """Memory-side bytes per launch from the COMMITTED rocprofv3 PMC passes of this same command (PMC counters cannot
    be read from inside the process, so this is never a measurement of the present run: the line says so with
    `traffic_measured_in_run: false`).  -> (bytes, file, kernel the passes profiled) or (None, None, None)."""
    if world != 1:
        return None, None, None
    fams = PMC_FILES.get(m, [])
    # the kernel's own family first, then (same stored bytes) its nearest relative: k_sellp2 <-> k_sellp
    order = [f for f in fams if kernel_name.startswith(f[0])] + [f for f in fams if kernel_name.startswith("k_sellp") and f[0].startswith("k_sellp")]
    for fam, name in order:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f)["traffic_bytes_per_launch"], "profiles/" + name, fam
    return None, None, None


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
    # Before the W warm-up steps: ~0.1-0.3 s of the same launches, untimed, so that the card's clocks have left whatever state the
    # host-side setup left them in.  (Round 4: on one fresh box the first 440 launches of the process ran at 42.5 us each and the
    # streaming-ceiling kernel measured right after them at its usual 19.9 us -- profiles/r04_bench_n1_first_process_slow.json; the
    # driver's default run times 20 steps.)  The count depends on the GLOBAL size only: every rank launches the same number.
    settle = 4000 if A.num_rows <= 4_000_000 else 1000
    op.time_kernel(0, x, None, y, settle)
    for _ in range(warmup):
        op.spmv(x, y)
    sync_all()
    t0 = time.perf_counter()
    ms_kernel = op.time_kernel(0, x, None, y, steps)     # K launches, HIP events on the compute stream
    sync_all()                                           # device synchronize + barrier
    wall = time.perf_counter() - t0
    return dict(op=op, info=info, kernel_name=kernel_name, ms_kernel=ms_kernel, wall=wall, B_local=op.algorithmic_bytes(0),
                x=x, y=y, g0=g0, settle_launches=settle)


def stored_bytes(info, kernel_name):
    """bytes the SpMV's operands occupy in HBM: values, column ids as the chosen kernel stores them, row pointers, x, y"""
    nnz = info["nnz_local"] + info["nnz_remote"]
    vec = 8 * info["N_local"] + 8 * info["M"]
    if kernel_name == "k_rowt":                          # (opt-in) a 16-bit template id per row, nothing per entry
        return 2 * info["M"] + vec
    if kernel_name.startswith("k_sellp"):                # k_sellp, k_sellp2, their <wide> forms, k_sellpx: no column stream, a 16-bit pattern id per row
        return 8 * nnz + (6 if "rowbase" in kernel_name else 2) * info["M"] + vec   # (+ a table of a few hundred ints / a few KiB per workgroup; rowbase: + the row's first column)
    if kernel_name in ("k_sell", "k_sell<sorted>", "k_sellx", "k_csr_xlds", "k_csr_xldsr"):   # 16-bit column codes; k_sell: a 16-bit row length instead of the row pointer (padding < 1 % here)
        per_row = {"k_sell": 2, "k_sell<sorted>": 6}.get(kernel_name, 4)             # (sorted: + the 32-bit row a slice position holds)
        return 10 * nnz + per_row * info["M"] + vec
    if kernel_name == "k_dense_rows":
        return 8 * info["M"] * info["N_local"] + vec
    col_bytes = 2 if ("cc16" in kernel_name or "k_csr_cm" in kernel_name) else 4
    extra = 2 if "k_csr_cm" in kernel_name else 0       # k_csr_cm: + a 16-bit tile slot per entry
    return (8 + col_bytes + extra) * nnz + 4 * (info["M"] + 1) + vec


INFINITY_CACHE_BYTES = 256 * 2 ** 20                    # MI355X: 256 MiB memory-side cache (MI355X_MICROARCH.md)


def irregular_leg(capi, host, np, comm, nblocks, sync_all):
    """BASELINE.json configs[4] ("SuiteSparse/Florida matrix ... irregular nnz/row, stresses load-balance of wavefront CSR") at a size
    where it is bound by the memory system: the reference's SiH4 (data/FloridaCollection, 5 041 rows -- 4 us of launch floor on this
    chip) replicated to >= 1 M rows with per-block permutations, coupling between blocks and hub rows (tests/irregular.py).  SpMV and
    Jacobi sweep through the same entry points, kernel from the plan-time autotune, one more SpMV checked against the host product."""
    from tests import irregular
    t0 = time.perf_counter()
    r, c, v, M = irregular.sih4_replicated(nblocks)
    lens = irregular.row_length_stats(r, M)
    Ai = host.Matrix(comm)
    Ai.set_remove_boundary(False)
    Ai.set_many(r, c, v)
    del r, c, v
    Ai.assemble()
    t_build = time.perf_counter() - t0
    op = host.device_operator(Ai)
    plan = op.block_plan(0)                              # before the autotune frees the host copies it does not need
    if os.environ.get("SAENA_BENCH_VARIANT_IRREGULAR"):      # pinned for the rocprofv3 --pmc passes (tools/pmc_spmv_irregular.sh)
        op.set_variant(int(os.environ["SAENA_BENCH_VARIANT_IRREGULAR"]))
        if os.environ.get("SAENA_BENCH_LANES_IRREGULAR"):
            op.set_lanes_per_row(int(os.environ["SAENA_BENCH_LANES_IRREGULAR"]))
    else:
        op.autotune()
    info = op.info()
    _, kname = op.variant()
    x = capi.DeviceVector(M, np.sin(0.001 * np.arange(M)))
    y, rhs = capi.DeviceVector(M), capi.DeviceVector(M, np.ones(M))
    for _ in range(5):
        op.spmv(x, y)
    sync_all()
    ms = op.time_kernel(0, x, None, y, 50)
    ms_j = op.time_kernel(1, x, rhs, y, 50)
    # check: one more SpMV against the host-formed product from the layout arrays
    op.spmv(x, y)
    got = y.download()
    d = host.desc_arrays(Ai.desc())
    rows = np.repeat(np.arange(M), d["nnzPerRow_local"])
    t = d["val_local"] * np.sin(0.001 * d["col_local"].astype(np.float64))
    want = np.bincount(rows, weights=t, minlength=M)
    bound = float(np.bincount(rows, weights=np.abs(t), minlength=M).max())
    err = float(np.max(np.abs(got - want)) / bound)
    nnz = info["nnz_local"]
    B, Bj = op.algorithmic_bytes(0), op.algorithmic_bytes(1)
    ws = stored_bytes(info, kname)
    # bytes leaving the L2 per launch from the committed PMC passes of this leg (same kernel family only)
    tr, tr_src, tr_k = None, None, None
    pmc = os.path.join(ROOT, "profiles", "r04_pmc_spmv_irregular.json")
    if nblocks == 200 and os.path.exists(pmc):
        with open(pmc) as f:
            pj = json.load(f)
        if pj["bench_kernel"].split(",")[0] == kname:
            tr, tr_src, tr_k = pj["traffic_bytes_per_launch"], "profiles/r04_pmc_spmv_irregular.json", kname
    rf = roofline_object(capi, info, kname, ms, B, B / (ms * 1e-3) / 1e9, ws, tr, tr_src, 20, tr_k)
    out = {"workload": f"SiH4 (reference data/FloridaCollection) x {nblocks} permuted diagonal blocks + coupling + hub rows: {M} rows x {nnz} nnz; BASELINE configs[4] scaled to an HBM-bound size",
           "rows": M, "nnz": nnz, "row_lengths": lens,
           "row_block_plan": {**plan, "mean_nnz_per_block": round(plan["nnz"] / max(1, plan["blocks"]), 1),
                              "max_over_mean": round(plan["max_nnz_per_block"] * plan["blocks"] / max(1, plan["nnz"]), 3),
                              "what": "the tile kernels' row blocks (<= 2048 products and <= 256 rows each; a longer row gets a block of its own): the "
                                      "spread of work a workgroup gets -- blocks are dealt to the CUs dynamically, so the spread costs a tail, not a stall"},
           **rf, "stored_bytes_per_nnz": round((ws - 16 * M) / nnz, 3),
           "jacobi": {"us_per_sweep": round(ms_j * 1e3, 3), "achieved": round(Bj / (ms_j * 1e-3) / 1e9, 2), "frac": round(Bj / (ms_j * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "unit": "GB/s"},
           "check_max_rel_err": err, "check_ok": bool(err <= 1e-13), "build_s": round(t_build, 2)}
    op.destroy()
    for vec in (x, y, rhs):
        vec.free()
    Ai.free()
    return out


def roofline_object(capi, info, kernel_name, ms_kernel, B_alg, achieved, ws, traffic, traffic_src, reps, traffic_kernel=None):
    """The contract's roofline object.  `achieved` = algorithmic bytes / kernel time.  What it is divided by depends on where the
    working set lives (round-3 review: the spec-peak fraction of a cache-resident operator was 1.10 -- not an HBM fraction):
      * beyond the 256 MiB Infinity Cache: bound "hbm", `peak` = the 8 TB/s spec peak, `frac` = achieved / peak (algorithmic bytes:
        a form that stores 8 B per entry where the contract counts 12 can pass 1; `stored_frac` = the bytes actually stored / peak
        cannot), plus `peak_measured` / `frac_of_measured` against the in-run streaming ceiling;
      * inside it: bound "infinity_cache", `peak` = `peak_measured` (the in-run streaming ceiling of this byte mix, which the cache
        serves as well), `frac` = `frac_of_measured` <= 1; the spec-peak ratio is kept as `frac_of_hbm_spec_peak` with a note."""
    cache = bool(ws <= INFINITY_CACHE_BYTES)
    lanes = 1 if kernel_name in ("k_sell", "k_sellp", "k_sellp<wide>", "k_sellpx") else 0.5 if kernel_name.startswith("k_sellp2") else info["lanes_per_row"]
    mc = measured_ceiling(capi, info, kernel_name, ms_kernel, B_alg, reps)
    o = {"bound": "infinity_cache" if cache else "hbm", "kernel": f"{kernel_name}, {lanes} lane(s)/row", "achieved": round(achieved, 2), "unit": "GB/s"}
    if cache:
        o.update(peak=mc["peak_measured"], frac=mc["frac_of_measured"], peak_hbm_spec=HBM_PEAK_GBS, frac_of_hbm_spec_peak=round(achieved / HBM_PEAK_GBS, 4),
                 note="the operator and vectors fit the 256 MiB Infinity Cache: repeated launches are served by it, so the rate is not an HBM rate and "
                      "`frac_of_hbm_spec_peak` may pass 1; `peak` is the streaming ceiling of the same stored bytes measured in this run (in the "
                      "contract's algorithmic bytes), `frac` = time(ceiling) / time(kernel); see spmv_hbm_resident for the HBM-bound figure")
    else:
        o.update(peak=HBM_PEAK_GBS, frac=round(achieved / HBM_PEAK_GBS, 4), stored_frac=round(ws / (ms_kernel * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 note="working set beyond the 256 MiB Infinity Cache: HBM-bound; `frac` is in the contract's algorithmic bytes (12 B per entry: a form that "
                      "stores fewer can pass 1), `stored_frac` in the bytes the chosen form stores, `frac_of_measured` against the in-run streaming ceiling")
    o.update(mc)
    o.update(us_per_launch=round(ms_kernel * 1e3, 3), algorithmic_bytes=B_alg, working_set_bytes=ws,
             stored_bytes_rate_gbs=round(ws / (ms_kernel * 1e-3) / 1e9, 2), cache_resident=cache,
             traffic=traffic, traffic_source=traffic_src, traffic_measured_in_run=False if traffic is not None else None)
    if traffic is not None:
        o["traffic_kernel"] = traffic_kernel             # the kernel the committed counter passes profiled (its family's, if not this one)
    return o


def measured_ceiling(capi, info, kernel_name, ms_kernel, B_alg, reps):
    """What THIS device gives a kernel that moves the bytes the chosen form stores (values, its column / pattern ids, x once,
    y once) with coalesced 16-byte loads and does nothing else -- sgpu_debug_stream_ceiling, timed in this process right after
    the operator's own launches, the fastest of plain / non-temporal loads and stores.  `peak_measured` is that rate in the
    contract's ALGORITHMIC bytes (so that achieved / peak_measured = time(ceiling) / time(kernel)), `frac_of_measured` <= 1 by
    construction: the operator's kernel moves the same bytes and gathers on top."""
    ws = stored_bytes(info, kernel_name)
    wr = 8 * info["M"]
    us, mode, moved = capi.stream_ceiling(ws - wr, wr, reps)
    us_scaled = us * ws / moved                          # (the ceiling's read stream is rounded to whole 16-byte loads per row)
    return {"peak_measured": round(B_alg / (us_scaled * 1e-6) / 1e9, 2), "frac_of_measured": round(us_scaled / (ms_kernel * 1e3), 4),
            "ceiling": {"us_per_launch": round(us_scaled, 3), "form": mode, "bytes_moved": int(moved), "stored_bytes": int(ws),
                        "stored_bytes_rate_gbs": round(ws / (us_scaled * 1e-6) / 1e9, 2),
                        "what": "pure streaming kernel over the operator's stored byte mix (no gathers, no LDS), same process, back-to-back launches"}}


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
    comm_state = {"kind": setup_comm if world > 1 else "rccl", "note": None}
    if world > 1 and setup_comm == "shm":
        # the shared-memory segments live in /dev/shm: a container's default is 64 MB while the exchanges of the configs[3] setup
        # move gigabytes per rank (round-3 advisor finding) -- not enough room there means the gloo transport from the start
        try:
            st = os.statvfs("/dev/shm")
            free = st.f_bavail * st.f_frsize
        except OSError:
            free = 0
        need = world * (2 << 30)                           # the largest exchange of the setup is 58.5 MiB per rank at 1 M rows per rank (2 ranks, measured:
                                                           # SAENA_SETUP_TIMING prints the segment growing), ~1 GiB at configs[3] size; a refusal later on falls back as well
        box = [free if rank == 0 else None]                # one decision for the job: rank 0's view
        dist.broadcast_object_list(box, src=0)
        if box[0] < need:
            comm_state.update(kind="gloo", note=f"/dev/shm has {box[0] / 2 ** 30:.1f} GiB free, below the {need / 2 ** 30:.0f} GiB kept for {world} ranks: setup collectives over gloo")

    def make_comm(kind):
        if kind == "shm":
            # the job's ranks sit on one node (the launch contract): the setup's collectives are memory copies through the native
            # shared-memory communicator (saena_amd/csrc/host/shm_comm.cpp); the name is fresh for every job and every communicator
            box = [f"{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}_{time.time_ns() % 10 ** 12}" if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return host.Comm("gpu", "shm", (box[0], rank, world))
        if kind == "gloo":
            return host.Comm("gpu", "dist", dist)
        return host.Comm("gpu", "rccl")

    def with_shm_fallback(fn):
        """fn(comm) -> result, collective.  A failure of the shared-memory communicator (no room in /dev/shm: reported by every
        rank, the others through the failed flag) is not the end of the run: the ranks agree over the rendezvous group, switch
        the setup's collectives to gloo and run fn again."""
        err, res = None, None
        try:
            res = fn(comm_state["comm"])
        except Exception as e:                              # noqa: BLE001
            if comm_state["kind"] != "shm" or "shared-memory communicator" not in str(e):
                raise
            err = e
        if world > 1 and comm_state["kind"] == "shm":
            import torch
            t = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t[0]) > 0:
                comm_state.update(kind="gloo", note=f"shared-memory communicator failed ({err or 'on another rank'}): setup collectives over gloo")
                print(f"bench.py rank {rank}: {comm_state['note']}", file=sys.stderr, flush=True)
                comm_state["comm"] = make_comm("gloo")
                res = fn(comm_state["comm"])
        return res

    comm_state["comm"] = make_comm(comm_state["kind"])
    if world == 1 or world == 8:
        mg = m
    else:                                                # the cube that gives every GPU the rows it holds at N = 8: (N/8)^(1/3) (m - 2) interior points per side
        mg = int(round((world / 8.0) ** (1.0 / 3.0) * (m - 2))) + 2
    grid = (mg, mg, mg)

    def build_fine(c, buckets=0):
        Ax = host.Matrix(c)
        if buckets:
            Ax.set_partition_buckets(buckets)            # opt-in: NOT the reference's partition
        return Ax.laplacian3D(*grid).assemble()
    A = with_shm_fallback(build_fine)                    # reference partitioner: nnz-balanced contiguous row blocks
    comm = comm_state["comm"]
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

    def reduce_measurement(Rx, errx):
        """-> (max error, max wall, total algorithmic bytes, rows per rank, remote nnz per rank, nnz per rank)"""
        ix = Rx["info"]
        nz = ix["nnz_local"] + ix["nnz_remote"]
        if dist is None:
            return errx, Rx["wall"], Rx["B_local"], [ix["M"]], [ix["nnz_remote"]], [nz]
        import torch
        t = torch.tensor([errx, Rx["wall"]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        b = torch.tensor([float(Rx["B_local"])], dtype=torch.float64)
        dist.all_reduce(b)
        gg = [None] * world
        dist.all_gather_object(gg, (ix["M"], ix["nnz_remote"], nz))
        return float(t[0]), float(t[1]), float(b[0]), [x[0] for x in gg], [x[1] for x in gg], [x[2] for x in gg]

    def imbalance(rows, nnzs):
        """what the slowest rank carries against the mean: `value` divides all ranks' bytes by the slowest rank's wall time, so a
        weak-scaling efficiency cannot exceed mean / max before any exchange cost"""
        mr, mn = sum(rows) / len(rows), sum(nnzs) / len(nnzs)
        return {"rows_max_over_mean": round(max(rows) / mr, 4), "nnz_max_over_mean": round(max(nnzs) / mn, 4),
                "efficiency_cap": round(mn / max(nnzs), 4)}
    err, wall, B_total, rows_all, halo_all, nnz_all = reduce_measurement(R, err)

    balanced = None
    if world > 1 and not args.no_balanced:
        # The reference's partitioner works on nparts^2 row buckets (src/saena_matrix_repart.cpp:43-170): a rank gets 7-9 of 64
        # buckets at N = 8, 3-5 of 16 at N = 4.  The same SpMV measurement under the OPT-IN finer histogram
        # (saena_matrix_set_partition_buckets): reported next to `value`, never instead of it.
        bal_dog = watchdog(args.assemble_timeout + args.spmv_timeout, "the SpMV measurement under the finer partition", 4)
        Ab = with_shm_fallback(lambda c: build_fine(c, args.balanced_buckets))
        Rb = measure_spmv(capi, host, np, Ab, rank, args.steps, args.warmup, sync_all)
        eb, _ = verify_spmv(np, host, Ab, Rb["op"], Rb["x"], Rb["y"], Rb["g0"])
        eb, wb, Bb, rows_b, halo_b, nnz_b = reduce_measurement(Rb, eb)
        bal_dog.cancel()
        balanced = {"what": f"the same measurement with at least {args.balanced_buckets} row buckets in the nnz-balanced partitioner instead of the "
                            f"reference's nparts^2 = {world * world} (opt-in: saena_matrix_set_partition_buckets / SAENA_FINE_PARTITION_BUCKETS)",
                    "value": round(Bb / (wb / args.steps) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(wb / args.steps * 1e3, 6),
                    "kernel": Rb["kernel_name"], "us_per_launch_rank0": round(Rb["ms_kernel"] * 1e3, 3), "rows_per_gpu": rows_b,
                    "partition_imbalance": imbalance(rows_b, nnz_b), "check_max_rel_err": eb, "check_ok": bool(eb <= 1e-13)}
        Rb["op"].destroy()
        for k in ("x", "y"):
            Rb[k].free()
        Ab.free()
        del Rb, Ab

    out = None
    if rank == 0:
        sec_per_step = wall / args.steps
        achieved = B_local / (ms_kernel * 1e-3) / 1e9
        ws = stored_bytes(info, kernel_name)
        traffic, traffic_src, traffic_kernel = pmc_traffic(m, world, kernel_name)
        out = {
            "metric": f"fine-level SpMV effective GB/s (3D 7-pt Poisson {grid_s}, fp64)",
            "value": round(B_total / sec_per_step / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_launches": R["settle_launches"],      # untimed launches of the same kernel in front of the W warm-up steps (clock state)
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
                **({"partition_imbalance": {**imbalance(rows_all, nnz_all),
                                            "note": f"the reference's partitioner splits at multiples of 1/{world * world} of the rows (nparts^2 buckets); "
                                                    "`value` = all ranks' bytes / the slowest rank's time, so this caps the weak-scaling efficiency; "
                                                    "`value_balanced` is the same measurement under the opt-in finer histogram"},
                    "setup_comm": comm_state["kind"] + (f" ({comm_state['note']})" if comm_state["note"] else "")} if world > 1 else {}),
                "partition": "1 rank" if world == 1 else
                             f"{world} nnz-balanced contiguous row blocks (reference partitioner), RCCL halo of ~{grid[0] - 2}^2 doubles per side "
                             f"(remote nnz per rank {halo_all})",
                "pct_of_hbm_peak": round(B_total / sec_per_step / 1e9 / (HBM_PEAK_GBS * world) * 100, 2),
                **({"weak_scaling_reference": "the 1-GPU figure at THIS per-GPU size is `spmv_hbm_resident` of the N=1 line (Poisson 256^3, "
                                              "16.4 M rows, HBM-resident: 7.2-7.7 TB/s in algorithmic bytes, profiles/r04_bench_n1.json); the N=1 `value` "
                                              "is the Infinity-Cache-resident configs[1] operator (8.8 TB/s) and not the denominator of a weak-scaling "
                                              "ratio; `partition_imbalance.efficiency_cap` is what the reference's partition allows before any exchange cost"}
                   if world > 1 else {}),
            },
            **({"value_balanced": balanced["value"], "balanced_partition": balanced} if balanced else {}),
            "check": {"what": "y = A x of one more SpMV against the host-formed product from the layout arrays, halo values included; "
                              "max over all ranks of max_i |y_gpu - y_host| / max_i sum_j |a_ij x_j|",
                      "max_rel_err": err, "ok": bool(err <= 1e-13)},
            "roofline": roofline_object(capi, info, kernel_name, ms_kernel, B_local, achieved, ws, traffic, traffic_src, max(20, args.steps // 4), traffic_kernel),
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
        rf3 = roofline_object(capi, R3["info"], R3["kernel_name"], R3["ms_kernel"], R3["B_local"], a3, ws3, None, None, 20)
        out["spmv_hbm_resident"] = {"workload": f"Poisson {args.m_hbm}^3: {R3['info']['M']} rows x {R3['info']['nnz_local']} nnz, same kernel path",
                                    "steps": steps3, "warmup": warm3, **rf3, "check_max_rel_err": e3}
        # bytes leaving the L2 per launch from the COMMITTED PMC passes of this leg (tools/pmc_spmv_hbm.sh), by kernel family
        t3, src3, k3 = pmc_traffic(args.m_hbm, world, R3["kernel_name"])
        if t3 is not None:
            out["spmv_hbm_resident"].update(traffic=t3, traffic_source=src3, traffic_measured_in_run=False, traffic_kernel=k3)
        R3["op"].destroy()
        for k in ("x", "y"):
            R3[k].free()
        A3.free()
        del R3, A3

    if world == 1 and not multi and args.irregular_blocks:
        out["spmv_irregular"] = irregular_leg(capi, host, np, comm, args.irregular_blocks, sync_all)

    if rank == 0:
        if world == 1 and not multi and not args.no_vcycle:
            out["vcycle"] = vcycle_leg(capi, host, A, m)
        if world == 1 and not multi and not args.no_vcycle and not args.no_config2_vcycle and m == 128:
            # BASELINE configs[2]: the full V-cycle with its R / P transfers on Poisson 256^3 (16.4 M rows, 10 levels, 1.0 G entries)
            A256 = host.Matrix(comm).laplacian3D(256).assemble()
            leg = vcycle_leg(capi, host, A256, 256)
            leg["workload"] = "BASELINE configs[2]: Poisson 256^3, 16387064 rows, full (3,3)-Jacobi V-cycle with R / P transfers, solve_pCG, 1 MI355X"
            out["vcycle_256"] = leg
            A256.free()
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
            m128 = 128 if m >= 128 else m

            def leg128(c, buckets=0):
                A2 = host.Matrix(c)
                if buckets:
                    A2.set_partition_buckets(buckets)
                A2.laplacian3D(m128).assemble()
                lg = vcycle_leg(capi, host, A2, m, dist)
                lg["scaling"] = "strong"
                lg["rows_per_gpu"] = [int(x) for x in np.diff(A2.split)]
                A2.free()
                return lg
            leg = with_shm_fallback(leg128)
            leg["partition"] = f"{world} nnz-balanced row blocks of the global Poisson {m128}^3 operator (reference partitioner)"
            legs["vcycle"] = leg
            if not args.no_balanced:
                # the same solve under the opt-in finer partition: the residual history does not depend on the partition
                # (every ||r_k|| within 1e-10 ||r_0|| of the reference-partition run, checked here and in tests/test_gpu_rccl.py)
                capi.check(fatal(line("the 128^3 multi-rank V-cycle leg under the finer partition ended with a fatal signal").encode()))
                lb = with_shm_fallback(lambda c: leg128(c, args.balanced_buckets))
                lb["partition"] = f"{world} row blocks of the same operator from at least {args.balanced_buckets} buckets (opt-in)"
                h1, h2 = leg["residual_history"], lb["residual_history"]
                same = len(h1) == len(h2) and all(abs(a - b) <= 1e-10 * h1[0] for a, b in zip(h1, h2))
                lb["history_matches_reference_partition"] = {"criterion": "same iteration count, every ||r_k|| within 1e-10 ||r_0||",
                                                             "max_abs_diff_over_r0": (max(abs(a - b) for a, b in zip(h1, h2)) / h1[0]) if len(h1) == len(h2) else None,
                                                             "ok": bool(same)}
                legs["vcycle_balanced_partition"] = lb
                if not same:
                    raise RuntimeError("the 128^3 residual history under the finer partition differs from the reference-partition history")
            if args.no_config4_vcycle:
                legs["vcycle_config4"] = {"skipped": "--no-config4-vcycle"}
            else:
                capi.check(fatal(line("the configs[3] V-cycle leg ended with a fatal signal").encode()))
                # (2) configs[3]: the operator of the SpMV measurement above (16.6 M rows per GPU at m = 512); every
                #     rank builds only its rows of the hierarchy (row-distributed setup)
                fine = {"A": A, "comm": comm_state["comm"]}

                def leg4(c):
                    if fine["comm"] is not c:              # the setup's transport changed under us: the operator is tied to the old one
                        fine.update(A=build_fine(c), comm=c)
                    return vcycle_leg(capi, host, fine["A"], m, dist, check_residual=True)
                leg = with_shm_fallback(leg4)
                leg["scaling"] = "weak"
                leg["partition"] = f"{world} nnz-balanced row blocks of Poisson {grid_s} (reference partitioner)"
                legs["vcycle_config4"] = leg
                if not leg["residual_check"]["ok"]:
                    raise RuntimeError(f"configs[3] V-cycle leg: host-recomputed relative residual {leg['residual_check']['host_relative_residual']:.3e} > 2e-8")
            if comm_state["note"] and rank == 0:
                out["config"]["setup_comm"] = comm_state["kind"] + f" ({comm_state['note']})"
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
