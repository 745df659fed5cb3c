"""ctypes binding of libsaena_amd.so (include/saena_gpu.h).

There is no CPU fallback: importing works anywhere (so the symbol table can be
checked without a GPU), but every compute entry point needs `init()` to have
found an MI355X, and a missing shared library raises immediately.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsaena_amd.so")

_PI = C.POINTER(C.c_int)
_PD = C.POINTER(C.c_double)


class SgpuError(RuntimeError):
    pass


class OpDesc(C.Structure):
    """sgpu_op_desc (include/saena_gpu.h)"""
    _fields_ = [
        ("M", C.c_int), ("N_local", C.c_int), ("col_offset", C.c_int),
        ("nnz_l_local", C.c_long), ("nnzPerRow_local", _PI), ("col_local", _PI), ("val_local", _PD),
        ("nnz_l_remote", C.c_long), ("col_remote_size", C.c_int),
        ("nnzPerCol_remote", _PI), ("row_remote", _PI), ("val_remote", _PD),
        ("numRecvProc", C.c_int), ("numSendProc", C.c_int),
        ("recvProcRank", _PI), ("recvProcCount", _PI), ("sendProcRank", _PI), ("sendProcCount", _PI),
        ("vIndexSize", C.c_int), ("vIndex", _PI), ("inv_diag", _PD), ("halo_fp32", C.c_int),
    ]


class AmgParams(C.Structure):
    """sgpu_amg_params"""
    _fields_ = [
        ("preSmooth", C.c_int), ("postSmooth", C.c_int), ("smoother", C.c_int), ("jacobi_omega", C.c_double),
        ("coarse_solver", C.c_int), ("CG_coarsest_max_iter", C.c_int), ("CG_coarsest_tol", C.c_double),
        ("solver_max_iter", C.c_int), ("solver_tol", C.c_double), ("use_graph", C.c_int),
    ]


# every symbol include/saena_gpu.h (the boundary) and include/saena_gpu_debug.h (test scaffolding) declare:
# name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "sgpu_last_error": (C.c_char_p, []),
    "sgpu_get_unique_id": (C.c_int, [_VP]),
    "sgpu_init": (C.c_int, [C.c_int, C.c_int, C.c_int, _VP]),
    "sgpu_finalize": (C.c_int, []),
    "sgpu_rank": (C.c_int, []),
    "sgpu_nranks": (C.c_int, []),
    "sgpu_device_sync": (C.c_int, []),
    "sgpu_barrier": (C.c_int, []),
    "sgpu_vec_alloc": (C.c_int, [C.POINTER(_VP), C.c_size_t]),
    "sgpu_vec_free": (C.c_int, [_VP]),
    "sgpu_vec_upload": (C.c_int, [_VP, _VP, C.c_size_t]),
    "sgpu_vec_download": (C.c_int, [_VP, _VP, C.c_size_t]),
    "sgpu_vec_fill": (C.c_int, [_VP, C.c_double, C.c_size_t]),
    "sgpu_vec_copy": (C.c_int, [_VP, _VP, C.c_size_t]),
    "sgpu_vec_axpby": (C.c_int, [C.c_double, _VP, C.c_double, _VP, C.c_size_t]),
    "sgpu_dot": (C.c_int, [_VP, _VP, C.c_size_t, _PD]),
    "sgpu_op_create": (C.c_int, [C.POINTER(OpDesc), C.POINTER(_VP)]),
    "sgpu_op_destroy": (C.c_int, [_VP]),
    "sgpu_op_info": (C.c_int, [_VP, _PI, _PI, C.POINTER(C.c_long), C.POINTER(C.c_long), _PI, _PI]),
    "sgpu_op_set_lanes_per_row": (C.c_int, [_VP, C.c_int]),
    "sgpu_op_set_variant": (C.c_int, [_VP, C.c_int]),
    "sgpu_op_autotune": (C.c_int, [_VP]),
    "sgpu_op_get_variant": (C.c_int, [_VP, _PI, C.POINTER(C.c_char_p)]),
    "sgpu_spmv": (C.c_int, [_VP, _VP, _VP]),
    "sgpu_residual": (C.c_int, [_VP, _VP, _VP, _VP]),
    "sgpu_residual_negative": (C.c_int, [_VP, _VP, _VP, _VP]),
    "sgpu_residual_multiply": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_double]),
    "sgpu_jacobi": (C.c_int, [_VP, C.c_int, C.c_double, _VP, _VP]),
    "sgpu_chebyshev": (C.c_int, [_VP, C.c_int, C.c_double, _VP, _VP]),
    "sgpu_prolong_correct": (C.c_int, [_VP, _VP, _VP]),
    "sgpu_debug_pack": (C.c_int, [_VP, _VP, _PD]),
    "sgpu_debug_gather_probe": (C.c_int, [_VP, C.c_int, _VP, C.c_int, C.POINTER(C.c_float)]),
    "sgpu_debug_block_plan": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_long)]),
    "sgpu_debug_stream_ceiling": (C.c_int, [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "sgpu_debug_inject_halo": (C.c_int, [_VP, _PD]),
    "sgpu_spmv_host": (C.c_int, [_VP, _PD, _PD]),
    "sgpu_jacobi_host": (C.c_int, [_VP, C.c_int, C.c_double, _PD, _PD]),
    "sgpu_chebyshev_host": (C.c_int, [_VP, C.c_int, C.c_double, _PD, _PD]),
    "sgpu_amg_default_params": (C.c_int, [C.POINTER(AmgParams)]),
    "sgpu_amg_create": (C.c_int, [C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), _PD, C.POINTER(AmgParams), C.POINTER(_VP)]),
    "sgpu_amg_destroy": (C.c_int, [_VP]),
    "sgpu_vcycle": (C.c_int, [_VP, _VP, _VP]),
    "sgpu_solve": (C.c_int, [_VP, _VP, _VP, _PI, _PD, C.c_int]),
    "sgpu_solve_pCG": (C.c_int, [_VP, _VP, _VP, _PI, _PD, C.c_int]),
    "sgpu_solve_CG": (C.c_int, [_VP, _VP, _VP, _PI, _PD, C.c_int]),
    "sgpu_solve_smoother": (C.c_int, [_VP, _VP, _VP, _PI, _PD, C.c_int]),
    "sgpu_amg_set_solve_params": (C.c_int, [_VP, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]),
    "sgpu_amg_profile_matvecs": (C.c_int, [_VP, C.c_int, _PD]),
    "sgpu_coarsest_solve": (C.c_int, [_VP, _VP, _VP, _PI]),
    "sgpu_debug_on_fatal_print": (C.c_int, [C.c_char_p]),
    "sgpu_debug_launch_count": (C.c_int, [C.POINTER(C.c_long)]),
    "sgpu_debug_chain_us": (C.c_int, [C.POINTER(C.c_double)]),
    "sgpu_debug_device_info": (C.c_int, [C.c_char_p, C.c_int]),
    "sgpu_debug_allow_local_only": (C.c_int, [_VP, C.c_int]),
    "sgpu_debug_init_host_transport": (C.c_int, [C.c_int, C.c_int, C.c_int, _VP, _VP, _VP]),
    "sgpu_time_kernel": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, C.c_int, C.POINTER(C.c_float)]),
    "sgpu_algorithmic_bytes": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64)]),
}

_lib = None


def lib():
    """Load libsaena_amd.so (built in-tree by __graft_entry__.build()); raise if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgpuError(f"{LIB_PATH} is missing: build the HIP extension first "
                            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        msg = lib().sgpu_last_error().decode(errors="replace")
        raise SgpuError(f"libsaena_amd status {status}: {msg}")


_initialised = False


_transport_keep = None


def init_host_transport(device, dist):
    """sgpu_debug_init_host_transport over a torch.distributed group that moves CPU tensors (gloo): this process is
    rank dist.get_rank() of dist.get_world_size(); halos and scalar reductions are routed through the host.  For
    validating the multi-rank code with several processes on ONE card (RCCL needs one device per rank)."""
    global _initialised, _transport_keep
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    XF = C.CFUNCTYPE(C.c_int, _VP, _VP, _PI, _PI, C.c_int, _VP, _PI, _PI, C.c_int, C.c_int)
    AF = C.CFUNCTYPE(C.c_int, _VP, _PD, C.c_int)

    def view(ptr, nbytes):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)) if nbytes else np.zeros(0, np.uint8)

    def exchange(user, send, srank, scount, nsend, recv, rrank, rcount, nrecv, eb):
        try:
            reqs, outs, keep = [], [], []
            so = 0
            for i in range(nsend):
                n = scount[i] * eb
                t = torch.from_numpy(view(send + so if send else None, n).copy())
                keep.append(t)
                reqs.append(dist.isend(t, srank[i]))
                so += n
            ro = 0
            for i in range(nrecv):
                n = rcount[i] * eb
                t = torch.empty(n, dtype=torch.uint8)
                outs.append((ro, n, t))
                reqs.append(dist.irecv(t, rrank[i]))
                ro += n
            for rq in reqs:
                rq.wait()
            for ro, n, t in outs:
                if n:
                    view(recv + ro, n)[:] = t.numpy()
            return 0
        except Exception as e:      # pragma: no cover
            print("host transport exchange failed:", e)
            return 1

    def allreduce(user, v, n):
        try:
            a = np.ctypeslib.as_array(v, shape=(n,))
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t)
            a[:] = t.numpy()
            return 0
        except Exception as e:      # pragma: no cover
            print("host transport allreduce failed:", e)
            return 1
    xf, af = XF(exchange), AF(allreduce)
    _transport_keep = (xf, af)
    check(lib().sgpu_debug_init_host_transport(int(device), rank, world, C.cast(xf, _VP), C.cast(af, _VP), None))
    _initialised = True


def init(device=0, rank=0, nranks=1, unique_id=None):
    """sgpu_init; raises SgpuError when no MI355X is visible."""
    global _initialised
    if _initialised:
        return
    buf = None
    if unique_id is not None:
        buf = C.create_string_buffer(bytes(unique_id), 128)
    check(lib().sgpu_init(device, rank, nranks, buf))
    _initialised = True


def finalize():
    global _initialised
    if _initialised:
        check(lib().sgpu_finalize())
        _initialised = False


def get_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().sgpu_get_unique_id(buf))
    return buf.raw


class DeviceVector:
    """A row slice in HBM (raw device pointer + length)."""

    def __init__(self, n, host=None):
        self.n = int(n)
        p = _VP()
        check(lib().sgpu_vec_alloc(C.byref(p), self.n))
        self.ptr = p
        if host is not None:
            self.upload(host)

    def upload(self, host):
        host = np.ascontiguousarray(host, np.float64)
        assert host.size == self.n
        check(lib().sgpu_vec_upload(self.ptr, host.ctypes.data, self.n))
        return self

    def download(self):
        out = np.empty(self.n, np.float64)
        check(lib().sgpu_vec_download(out.ctypes.data, self.ptr, self.n))
        return out

    def fill(self, a):
        check(lib().sgpu_vec_fill(self.ptr, float(a), self.n))
        return self

    def free(self):
        if self.ptr:
            lib().sgpu_vec_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if _initialised:
                self.free()
        except Exception:
            pass


def _ai(a):
    return np.ascontiguousarray(a, np.int32)


def _ad(a):
    return np.ascontiguousarray(a, np.float64)


class Operator:
    """sgpu_op built from one rank's arrays in the reference's storage layout."""

    def __init__(self, *, M, N_local, col_offset, nnzPerRow_local, col_local, val_local,
                 nnzPerCol_remote=(), row_remote=(), val_remote=(),
                 recvProcRank=(), recvProcCount=(), sendProcRank=(), sendProcCount=(), vIndex=(),
                 inv_diag=None, halo_fp32=False):
        keep = self._keep = {}
        keep["npr"], keep["col"], keep["val"] = _ai(nnzPerRow_local), _ai(col_local), _ad(val_local)
        keep["npc"], keep["rr"], keep["rv"] = _ai(nnzPerCol_remote), _ai(row_remote), _ad(val_remote)
        keep["rpr"], keep["rpc"] = _ai(recvProcRank), _ai(recvProcCount)
        keep["spr"], keep["spc"], keep["vi"] = _ai(sendProcRank), _ai(sendProcCount), _ai(vIndex)
        keep["inv"] = None if inv_diag is None else _ad(inv_diag)
        d = OpDesc()
        d.M, d.N_local, d.col_offset = int(M), int(N_local), int(col_offset)
        d.nnz_l_local = len(keep["col"])
        d.nnzPerRow_local = keep["npr"].ctypes.data_as(_PI)
        d.col_local = keep["col"].ctypes.data_as(_PI)
        d.val_local = keep["val"].ctypes.data_as(_PD)
        d.nnz_l_remote = len(keep["rr"])
        d.col_remote_size = len(keep["npc"])
        d.nnzPerCol_remote = keep["npc"].ctypes.data_as(_PI)
        d.row_remote = keep["rr"].ctypes.data_as(_PI)
        d.val_remote = keep["rv"].ctypes.data_as(_PD)
        d.numRecvProc, d.numSendProc = len(keep["rpr"]), len(keep["spr"])
        d.recvProcRank = keep["rpr"].ctypes.data_as(_PI)
        d.recvProcCount = keep["rpc"].ctypes.data_as(_PI)
        d.sendProcRank = keep["spr"].ctypes.data_as(_PI)
        d.sendProcCount = keep["spc"].ctypes.data_as(_PI)
        d.vIndexSize = len(keep["vi"])
        d.vIndex = keep["vi"].ctypes.data_as(_PI)
        d.inv_diag = keep["inv"].ctypes.data_as(_PD) if keep["inv"] is not None else None
        d.halo_fp32 = 1 if halo_fp32 else 0
        h = _VP()
        check(lib().sgpu_op_create(C.byref(d), C.byref(h)))
        self.h = h
        self.M, self.N_local = int(M), int(N_local)
        self._keep = None      # the library copied everything

    def info(self):
        M, N, nb, ln = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        nl, nr = C.c_long(), C.c_long()
        check(lib().sgpu_op_info(self.h, C.byref(M), C.byref(N), C.byref(nl), C.byref(nr), C.byref(nb), C.byref(ln)))
        return dict(M=M.value, N_local=N.value, nnz_local=nl.value, nnz_remote=nr.value, row_blocks=nb.value,
                    lanes_per_row=ln.value)

    def set_lanes_per_row(self, lanes):
        check(lib().sgpu_op_set_lanes_per_row(self.h, int(lanes)))

    def block_plan(self, big=0):
        """spread of work over the row blocks of the tile kernels' plan (sgpu_debug_block_plan)"""
        o = (C.c_long * 8)()
        check(lib().sgpu_debug_block_plan(self.h, int(big), o))
        k = ("blocks", "min_nnz_per_block", "max_nnz_per_block", "nnz", "min_rows_per_block", "max_rows_per_block", "long_rows", "longest_row")
        return dict(zip(k, (int(x) for x in o)))

    def variant(self):
        v, name = C.c_int(), C.c_char_p()
        check(lib().sgpu_op_get_variant(self.h, C.byref(v), C.byref(name)))
        return v.value, name.value.decode()

    def autotune(self):
        check(lib().sgpu_op_autotune(self.h))

    def set_variant(self, variant):
        check(lib().sgpu_op_set_variant(self.h, int(variant)))

    def spmv(self, v, w):
        check(lib().sgpu_spmv(self.h, v.ptr, w.ptr))

    def residual(self, u, rhs, res):
        check(lib().sgpu_residual(self.h, u.ptr, rhs.ptr, res.ptr))

    def residual_negative(self, u, rhs, res):
        check(lib().sgpu_residual_negative(self.h, u.ptr, rhs.ptr, res.ptr))

    def residual_multiply(self, u, rhs, res, w, c):
        check(lib().sgpu_residual_multiply(self.h, u.ptr, rhs.ptr, res.ptr, w.ptr, float(c)))

    def jacobi(self, it, u, rhs, omega=0.0):
        check(lib().sgpu_jacobi(self.h, int(it), float(omega), u.ptr, rhs.ptr))

    def chebyshev(self, it, eig_max, u, rhs):
        check(lib().sgpu_chebyshev(self.h, int(it), float(eig_max), u.ptr, rhs.ptr))

    def prolong_correct(self, e_coarse, u):
        check(lib().sgpu_prolong_correct(self.h, e_coarse.ptr, u.ptr))

    def debug_gather_probe(self, mode, x, reps=20):
        ms = C.c_float()
        check(lib().sgpu_debug_gather_probe(self.h, mode, x.ptr, reps, C.byref(ms)))
        return ms.value

    def debug_pack(self, v, n_send):
        out = np.empty(n_send)
        check(lib().sgpu_debug_pack(self.h, v.ptr, out.ctypes.data_as(_PD)))
        return out

    def debug_allow_local_only(self, allow=True):
        check(lib().sgpu_debug_allow_local_only(self.h, 1 if allow else 0))

    def debug_inject_halo(self, recv):
        recv = _ad(recv)
        check(lib().sgpu_debug_inject_halo(self.h, recv.ctypes.data_as(_PD)))

    def spmv_host(self, v):
        v = _ad(v)
        w = np.empty(self.M)
        check(lib().sgpu_spmv_host(self.h, v.ctypes.data_as(_PD), w.ctypes.data_as(_PD)))
        return w

    def algorithmic_bytes(self, kind=0):
        b = C.c_int64()
        check(lib().sgpu_algorithmic_bytes(self.h, kind, C.byref(b)))
        return b.value

    def time_kernel(self, kind, x, rhs, y, reps):
        ms = C.c_float()
        check(lib().sgpu_time_kernel(self.h, kind, x.ptr, rhs.ptr if rhs is not None else None, y.ptr, reps, C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.h:
            lib().sgpu_op_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if _initialised:
                self.destroy()
        except Exception:
            pass


class Amg:
    """sgpu_amg over Operators A[l], P[l], R[l]."""

    def __init__(self, A, P, R, eig_max=None, pre=3, post=3, smoother="jacobi", max_iter=100, tol=1e-8, use_graph=True, coarse_solver="direct"):
        self.A, self.P, self.R = list(A), list(P), list(R)
        n = len(A)
        prm = AmgParams()
        check(lib().sgpu_amg_default_params(C.byref(prm)))
        prm.preSmooth, prm.postSmooth = pre, post
        prm.smoother = 0 if smoother == "jacobi" else 1
        prm.solver_max_iter, prm.solver_tol = max_iter, tol
        prm.use_graph = 1 if use_graph else 0
        prm.coarse_solver = 1 if coarse_solver == "direct" else 0
        HA = (_VP * n)(*[a.h for a in A])
        HP = (_VP * n)(*([p.h for p in P] + [None] * (n - len(P))))
        HR = (_VP * n)(*([r.h for r in R] + [None] * (n - len(R))))
        eig = None
        if eig_max is not None:
            eig = _ad(eig_max)
        h = _VP()
        check(lib().sgpu_amg_create(n, HA, HP, HR, eig.ctypes.data_as(_PD) if eig is not None else None, C.byref(prm), C.byref(h)))
        self.h = h

    def vcycle(self, u, rhs):
        check(lib().sgpu_vcycle(self.h, u.ptr, rhs.ptr))

    def coarsest_solve(self, u, rhs):
        it = C.c_int()
        check(lib().sgpu_coarsest_solve(self.h, u.ptr, rhs.ptr, C.byref(it)))
        return it.value

    def _solve(self, fn, u, rhs, cap=256):
        it = C.c_int()
        hist = np.full(cap, np.nan)
        st = fn(self.h, u.ptr, rhs.ptr, C.byref(it), hist.ctypes.data_as(_PD), cap)
        if st not in (0, -6):
            check(st)
        return it.value, hist[~np.isnan(hist)], st == 0

    def solve(self, u, rhs):
        return self._solve(lib().sgpu_solve, u, rhs)

    def solve_pCG(self, u, rhs):
        return self._solve(lib().sgpu_solve_pCG, u, rhs)

    def solve_CG(self, u, rhs):
        return self._solve(lib().sgpu_solve_CG, u, rhs, cap=2048)

    def solve_smoother(self, u, rhs):
        return self._solve(lib().sgpu_solve_smoother, u, rhs, cap=2048)

    def set_solve_params(self, max_iter, tol, smoother, pre, post):
        check(lib().sgpu_amg_set_solve_params(self.h, int(max_iter), float(tol), 0 if smoother == "jacobi" else 1, int(pre), int(post)))

    def profile_matvecs(self, iters=5):
        us = np.zeros(len(self.A))
        check(lib().sgpu_amg_profile_matvecs(self.h, int(iters), us.ctypes.data_as(_PD)))
        return us

    def destroy(self):
        if self.h:
            lib().sgpu_amg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if _initialised:
                self.destroy()
        except Exception:
            pass


def stream_ceiling(read_bytes, write_bytes, reps=20):
    """-> (microseconds, mode name, bytes moved) of the fastest pure streaming kernel over this byte mix (sgpu_debug_stream_ceiling)"""
    us, mode, moved = C.c_float(), C.c_int(), C.c_size_t()
    check(lib().sgpu_debug_stream_ceiling(int(read_bytes), int(write_bytes), int(reps), C.byref(us), C.byref(mode), C.byref(moved)))
    return us.value, ("plain loads and stores", "non-temporal loads", "non-temporal stores", "non-temporal loads and stores")[mode.value], moved.value


def device_info():
    """one line about the context's device, plus the amdgpu partition modes where sysfs shows them"""
    buf = C.create_string_buffer(512)
    check(lib().sgpu_debug_device_info(buf, 512))
    line = buf.value.decode()
    import glob
    for key in ("current_compute_partition", "current_memory_partition"):
        vals = set()
        for f in glob.glob(f"/sys/class/drm/card*/device/{key}"):
            try:
                vals.add(open(f).read().strip())
            except OSError:
                pass
        if vals:
            line += f", {key.replace('current_', '')} {'/'.join(sorted(vals))}"
    return line


def chain_us():
    """the exchange chain sgpu_init measured on the communicator, microseconds (0 without one)"""
    v = C.c_double()
    check(lib().sgpu_debug_chain_us(C.byref(v)))
    return v.value


def launch_count():
    n = C.c_long()
    check(lib().sgpu_debug_launch_count(C.byref(n)))
    return n.value


def dot(x, y):
    out = C.c_double()
    check(lib().sgpu_dot(x.ptr, y.ptr, x.n, C.byref(out)))
    return out.value
