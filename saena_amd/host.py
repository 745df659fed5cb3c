"""ctypes binding of the host-side mirror (include/saena_c.h).

`HostLib("host")` loads libsaena_host.so (g++, no GPU: partition, layout,
generators); `HostLib("gpu")` loads libsaena_amd.so, which carries the same host
code plus the HIP path.  Both fail loudly when the shared object is missing.
"""
import ctypes as C
import os

import numpy as np

from .capi import OpDesc, SgpuError

_HERE = os.path.dirname(os.path.abspath(__file__))
_VP = C.c_void_p
_PI = C.POINTER(C.c_int)
_PD = C.POINTER(C.c_double)
_PS = C.POINTER(C.c_size_t)

CB_ALLGATHER = C.CFUNCTYPE(C.c_int, _VP, _VP, _VP, C.c_size_t)
CB_ALLTOALLV = C.CFUNCTYPE(C.c_int, _VP, _VP, _PS, _PS, _VP, _PS, _PS)
CB_I64 = C.CFUNCTYPE(C.c_int, _VP, C.POINTER(C.c_long), C.c_int)
CB_F64 = C.CFUNCTYPE(C.c_int, _VP, _PD, C.c_int)

HOST_SYMBOLS = {
    "saena_last_error": (C.c_char_p, []),
    "saena_comm_self": (_VP, []),
    "saena_comm_callbacks": (_VP, [C.c_int, C.c_int, _VP, CB_ALLGATHER, CB_ALLTOALLV, CB_I64, CB_F64]),
    "saena_comm_rccl": (_VP, []),
    "saena_comm_shm": (_VP, [C.c_char_p, C.c_int, C.c_int]),
    "saena_measured_chain_us": (C.c_double, []),
    "saena_comm_test_alltoallv": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "saena_comm_test_allreduce_f64": (C.c_int, [_VP, _VP, C.c_int]),
    "saena_comm_test_allreduce_i64": (C.c_int, [_VP, _VP, C.c_int]),
    "saena_comm_free": (None, [_VP]),
    "saena_matrix_new": (_VP, [_VP]),
    "saena_matrix_free": (None, [_VP]),
    "saena_matrix_set": (C.c_int, [_VP, C.c_int, C.c_int, C.c_double]),
    "saena_matrix_set_many": (C.c_int, [_VP, _PI, _PI, _PD, C.c_long]),
    "saena_matrix_read_file": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "saena_matrix_write_bin": (C.c_int, [_VP, C.c_char_p]),
    "saena_matrix_write_mtx": (C.c_int, [_VP, C.c_char_p]),
    "saena_matrix_set_remove_boundary": (C.c_int, [_VP, C.c_int]),
    "saena_matrix_set_partition_buckets": (C.c_int, [_VP, C.c_int]),
    "saena_matrix_add_duplicates": (C.c_int, [_VP, C.c_int]),
    "saena_matrix_set_eig": (C.c_int, [_VP, C.c_double]),
    "saena_matrix_assemble": (C.c_int, [_VP]),
    "saena_matrix_assemble_with_split": (C.c_int, [_VP, _PI]),
    "saena_matrix_get_num_rows": (C.c_int, [_VP]),
    "saena_matrix_get_num_local_rows": (C.c_int, [_VP]),
    "saena_matrix_get_nnz": (C.c_long, [_VP]),
    "saena_matrix_get_local_nnz": (C.c_long, [_VP]),
    "saena_matrix_get_split": (C.c_int, [_VP, _PI]),
    "saena_matrix_get_desc": (C.c_int, [_VP, C.POINTER(OpDesc)]),
    "saena_matrix_get_layout_extra": (C.c_int, [_VP, C.POINTER(_PI), C.POINTER(C.POINTER(C.c_long))]),
    "saena_matrix_get_halo_columns": (C.c_int, [_VP, C.POINTER(_PI)]),
    "saena_laplacian3D": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    "saena_laplacian3D_set_rhs": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _PD]),
    "saena_band_matrix": (C.c_int, [_VP, C.c_int, C.c_uint]),
    "saena_matmat": (C.c_int, [_VP, _VP, _VP]),
    "saena_prolong_new": (_VP, [_VP, C.c_int, C.c_int, _PI, _PI, _PI, _PI, _PD, C.c_long]),
    "saena_restrict_from_prolong": (_VP, [_VP]),
    "saena_transfer_free": (None, [_VP]),
    "saena_transfer_get_desc": (C.c_int, [_VP, C.POINTER(OpDesc)]),
    "saena_transfer_get_local_nnz": (C.c_long, [_VP]),
}



class OptionsC(C.Structure):
    """saena_options_c"""
    _fields_ = [("solver_max_iter", C.c_int), ("relative_tol", C.c_double), ("smoother", C.c_int),
                ("preSmooth", C.c_int), ("postSmooth", C.c_int), ("connStrength", C.c_float),
                ("dynamic_levels", C.c_int), ("max_level", C.c_int), ("float_level", C.c_int),
                ("filter_thre", C.c_double), ("filter_max", C.c_double), ("filter_start", C.c_int), ("filter_rate", C.c_int),
                ("switch_to_dense", C.c_int), ("dense_thre", C.c_float), ("dense_sz_thre", C.c_int)]


HOST_SYMBOLS.update({
    "saena_options_default": (C.c_int, [C.POINTER(OptionsC)]),
    "saena_options_from_file": (C.c_int, [C.c_char_p, C.POINTER(OptionsC)]),
    "saena_amg_new": (_VP, []),
    "saena_amg_free": (None, [_VP]),
    "saena_amg_set_matrix": (C.c_int, [_VP, _VP, C.POINTER(OptionsC)]),
    "saena_amg_num_levels": (C.c_int, [_VP]),
    "saena_amg_level_info": (C.c_int, [_VP, C.c_int, _PI, C.POINTER(C.c_long), C.POINTER(C.c_long), _PD]),
    "saena_amg_level_aggregates": (C.c_int, [_VP, C.c_int, _PI, _PI]),
    "saena_amg_level_desc": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(OpDesc)]),
    "saena_amg_level_split": (C.c_int, [_VP, C.c_int, _PI]),
    "saena_amg_to_device": (C.c_int, [_VP]),
    "saena_amg_device_handle": (_VP, [_VP]),
    "saena_amg_device_op": (_VP, [_VP, C.c_int, C.c_int]),
    "saena_amg_solve": (C.c_int, [_VP, _PD, _PD, _PI, _PD, C.c_int]),
    "saena_amg_solve_pCG": (C.c_int, [_VP, _PD, _PD, _PI, _PD, C.c_int]),
})

_libs = {}


def load(which="host"):
    """which: 'host' -> libsaena_host.so, 'gpu' -> libsaena_amd.so"""
    if which not in _libs:
        path = os.path.join(_HERE, "libsaena_host.so" if which == "host" else "libsaena_amd.so")
        if which == "host" and os.environ.get("SAENA_HOST_LIB"):      # tools/sanitize_host.sh: the sanitizer build of the host library
            path = os.environ["SAENA_HOST_LIB"]
        if not os.path.exists(path):
            raise SgpuError(f"{path} is missing: run __graft_entry__.build()")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in HOST_SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _libs[which] = L
    return _libs[which]


def _ai(a):
    return np.ascontiguousarray(a, np.int32)


def _ad(a):
    return np.ascontiguousarray(a, np.float64)


class Comm:
    """Setup-time communicator: self, the GPU runtime's RCCL communicator, the native shared-memory communicator of the
    ranks of one node, or torch.distributed (any backend that moves CPU tensors, e.g. gloo) through callbacks."""

    def __init__(self, which="host", kind="self", dist=None):
        self.L = load(which)
        self.kind = kind
        self.rank, self.nranks = 0, 1
        self._cbs = None
        if kind == "self":
            self.h = self.L.saena_comm_self()
        elif kind == "rccl":
            self.h = self.L.saena_comm_rccl()
            if not self.h:
                raise SgpuError(self.L.saena_last_error().decode())
        elif kind == "shm":              # dist = (name, rank, nranks): the ranks of one node through shared memory (native)
            name, self.rank, self.nranks = dist
            self.h = self.L.saena_comm_shm(str(name).encode(), int(self.rank), int(self.nranks))
            if not self.h:
                raise SgpuError(self.L.saena_last_error().decode())
        elif kind == "dist":
            self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
            self._make_dist_callbacks(dist)
            self.h = self.L.saena_comm_callbacks(self.rank, self.nranks, None, *self._cbs)
        else:
            raise ValueError(kind)

    def _make_dist_callbacks(self, dist):
        import torch
        np_ = self.nranks

        def buf(ptr, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)) if n else np.zeros(0, np.uint8)

        def allgather(user, send, recv, nbytes):
            try:
                s = torch.from_numpy(buf(send, nbytes).copy())
                outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(np_)]
                dist.all_gather(outs, s)
                r = buf(recv, nbytes * np_)
                for q in range(np_):
                    r[q * nbytes:(q + 1) * nbytes] = outs[q].numpy()
                return 0
            except Exception as e:      # pragma: no cover
                print("allgather callback failed:", e)
                return 1

        def alltoallv(user, send, sc, sd, recv, rc, rd):
            try:
                ins = [torch.from_numpy(buf(send + sd[q] if send else None, sc[q]).copy()) if sc[q] else torch.empty(0, dtype=torch.uint8)
                       for q in range(np_)]
                outs = [torch.empty(rc[q], dtype=torch.uint8) for q in range(np_)]
                # pairwise exchange (works on every backend that has send/recv)
                me = self.rank
                outs[me] = ins[me].clone()
                for step in range(1, np_):
                    to, frm = (me + step) % np_, (me - step) % np_
                    reqs = []
                    if sc[to]:
                        reqs.append(dist.isend(ins[to], to))
                    if rc[frm]:
                        reqs.append(dist.irecv(outs[frm], frm))
                    for rq in reqs:
                        rq.wait()
                for q in range(np_):
                    if rc[q]:
                        buf(recv + rd[q], rc[q])[:] = outs[q].numpy()
                return 0
            except Exception as e:      # pragma: no cover
                print("alltoallv callback failed:", e)
                return 1

        def ar_i64(user, v, n):
            t = torch.from_numpy(np.ctypeslib.as_array(v, shape=(n,)).copy())
            dist.all_reduce(t)
            np.ctypeslib.as_array(v, shape=(n,))[:] = t.numpy()
            return 0

        def ar_f64(user, v, n):
            t = torch.from_numpy(np.ctypeslib.as_array(v, shape=(n,)).copy())
            dist.all_reduce(t)
            np.ctypeslib.as_array(v, shape=(n,))[:] = t.numpy()
            return 0

        self._cbs = (CB_ALLGATHER(allgather), CB_ALLTOALLV(alltoallv), CB_I64(ar_i64), CB_F64(ar_f64))


def _check(L, status):
    if status != 0:
        raise SgpuError("saena host: " + L.saena_last_error().decode(errors="replace"))


def desc_arrays(d: OpDesc, nranks_for_scan=None):
    """Copy every array an sgpu_op_desc points at into numpy (for layout tests)."""
    def ai(p, n):
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.int32)

    def ad(p, n):
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float64)

    return dict(
        M=d.M, N_local=d.N_local, col_offset=d.col_offset,
        nnzPerRow_local=ai(d.nnzPerRow_local, d.M), col_local=ai(d.col_local, d.nnz_l_local), val_local=ad(d.val_local, d.nnz_l_local),
        nnzPerCol_remote=ai(d.nnzPerCol_remote, d.col_remote_size), row_remote=ai(d.row_remote, d.nnz_l_remote),
        val_remote=ad(d.val_remote, d.nnz_l_remote),
        recvProcRank=ai(d.recvProcRank, d.numRecvProc), recvProcCount=ai(d.recvProcCount, d.numRecvProc),
        sendProcRank=ai(d.sendProcRank, d.numSendProc), sendProcCount=ai(d.sendProcCount, d.numSendProc),
        vIndex=ai(d.vIndex, d.vIndexSize), inv_diag=ad(d.inv_diag, d.M) if d.inv_diag else None,
    )


class Matrix:
    """saena::matrix mirror (reference include/saena.hpp:14-73)."""

    def __init__(self, comm: Comm):
        self.comm, self.L = comm, comm.L
        self.h = self.L.saena_matrix_new(comm.h)

    def set(self, i, j, v):
        _check(self.L, self.L.saena_matrix_set(self.h, int(i), int(j), float(v)))

    def set_many(self, rows, cols, vals):
        r, c, v = _ai(rows), _ai(cols), _ad(vals)
        _check(self.L, self.L.saena_matrix_set_many(self.h, r.ctypes.data_as(_PI), c.ctypes.data_as(_PI), v.ctypes.data_as(_PD), len(r)))

    def read_file(self, name, input_type=""):
        _check(self.L, self.L.saena_matrix_read_file(self.h, os.fsencode(name), input_type.encode()))
        return self

    def write_bin(self, name):
        _check(self.L, self.L.saena_matrix_write_bin(self.h, os.fsencode(name)))

    def write_mtx(self, name):
        """saena::matrix::writeMatrixToFile: this rank's entries as <name>-r<rank>.mtx"""
        _check(self.L, self.L.saena_matrix_write_mtx(self.h, os.fsencode(name)))

    def set_remove_boundary(self, flag):
        self.L.saena_matrix_set_remove_boundary(self.h, 1 if flag else 0)

    def set_partition_buckets(self, n):
        """opt-in: at least n row buckets in the nnz-balanced partition of assemble() instead of the reference's nparts^2"""
        _check(self.L, self.L.saena_matrix_set_partition_buckets(self.h, int(n)))
        return self

    def laplacian3D(self, mx, my=None, mz=None):
        my = mx if my is None else my
        mz = mx if mz is None else mz
        self._grid = (mx, my, mz)
        _check(self.L, self.L.saena_laplacian3D(self.h, mx, my, mz))
        return self

    def band_matrix(self, M, bw):
        _check(self.L, self.L.saena_band_matrix(self.h, M, bw))
        return self

    def assemble(self, split=None):
        if split is None:
            _check(self.L, self.L.saena_matrix_assemble(self.h))
        else:
            s = _ai(split)
            _check(self.L, self.L.saena_matrix_assemble_with_split(self.h, s.ctypes.data_as(_PI)))
        return self

    def matmat(self, B):
        """saena::amg::matmat: C = self * B (host SpGEMM, one rank)"""
        Cm = Matrix(self.comm)
        _check(self.L, self.L.saena_matmat(self.h, B.h, Cm.h))
        return Cm

    def laplacian3D_rhs(self):
        mx, my, mz = self._grid
        out = np.empty(self.num_local_rows)
        _check(self.L, self.L.saena_laplacian3D_set_rhs(self.h, mx, my, mz, out.ctypes.data_as(_PD)))
        return out

    @property
    def num_rows(self):
        return self.L.saena_matrix_get_num_rows(self.h)

    @property
    def num_local_rows(self):
        return self.L.saena_matrix_get_num_local_rows(self.h)

    @property
    def nnz(self):
        return self.L.saena_matrix_get_nnz(self.h)

    @property
    def local_nnz(self):
        return self.L.saena_matrix_get_local_nnz(self.h)

    @property
    def split(self):
        s = np.zeros(self.comm.nranks + 1, np.int32)
        _check(self.L, self.L.saena_matrix_get_split(self.h, s.ctypes.data_as(_PI)))
        return s

    def desc(self):
        d = OpDesc()
        _check(self.L, self.L.saena_matrix_get_desc(self.h, C.byref(d)))
        return d

    def layout(self):
        d = self.desc()
        out = desc_arrays(d)
        cr, sc = _PI(), C.POINTER(C.c_long)()
        self.L.saena_matrix_get_layout_extra(self.h, C.byref(cr), C.byref(sc))
        out["col_remote"] = np.ctypeslib.as_array(cr, shape=(d.nnz_l_remote,)).copy() if d.nnz_l_remote else np.zeros(0, np.int32)
        out["nnzPerProcScan"] = np.ctypeslib.as_array(sc, shape=(self.comm.nranks + 1,)).copy()
        return out

    def halo_columns(self):
        """global column id of every slot of this rank's receive (halo) buffer"""
        d = self.desc()
        p = _PI()
        self.L.saena_matrix_get_halo_columns(self.h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(d.col_remote_size,)).copy() if d.col_remote_size else np.zeros(0, np.int32)

    def free(self):
        if self.h:
            self.L.saena_matrix_free(self.h)
            self.h = None


class Transfer:
    """prolong_matrix / restrict_matrix mirror."""

    def __init__(self, comm, h):
        self.comm, self.L, self.h = comm, comm.L, h

    @classmethod
    def prolong(cls, comm, Mbig, Nbig, split_row, split_col, rows, cols, vals):
        sr, sc, r, c, v = _ai(split_row), _ai(split_col), _ai(rows), _ai(cols), _ad(vals)
        h = comm.L.saena_prolong_new(comm.h, Mbig, Nbig, sr.ctypes.data_as(_PI), sc.ctypes.data_as(_PI),
                                     r.ctypes.data_as(_PI), c.ctypes.data_as(_PI), v.ctypes.data_as(_PD), len(r))
        if not h:
            raise SgpuError(comm.L.saena_last_error().decode())
        return cls(comm, h)

    def transpose(self):
        h = self.L.saena_restrict_from_prolong(self.h)
        if not h:
            raise SgpuError(self.L.saena_last_error().decode())
        return Transfer(self.comm, h)

    def desc(self):
        d = OpDesc()
        _check(self.L, self.L.saena_transfer_get_desc(self.h, C.byref(d)))
        return d

    def layout(self):
        return desc_arrays(self.desc())


def device_operator(obj, halo_fp32=False):
    """sgpu_op_create from a host Matrix/Transfer (GPU library only)."""
    from . import capi
    d = obj.desc()
    d.halo_fp32 = 1 if halo_fp32 else 0
    op = capi.Operator.__new__(capi.Operator)
    h = _VP()
    capi.check(capi.lib().sgpu_op_create(C.byref(d), C.byref(h)))
    op.h, op.M, op.N_local, op._keep = h, d.M, d.N_local, None
    return op


def options(L, xml=None, **kw):
    """saena::options: defaults (saena.hpp:151-155), an options XML of the reference's format, then overrides"""
    o = OptionsC()
    L.saena_options_default(C.byref(o))
    if xml is not None:
        _check(L, L.saena_options_from_file(os.fsencode(xml), C.byref(o)))
    for k, v in kw.items():
        if k == "smoother":
            v = 0 if v == "jacobi" else 1
        setattr(o, k, v)
    return o


# the reference's data/options001.xml (Jacobi 3+3, tol 1e-8, conn_str 0.2, filter 1e-14 -> 1e-8 rate 2)
OPTIONS001 = dict(solver_max_iter=50, relative_tol=1e-8, smoother="jacobi", preSmooth=3, postSmooth=3, connStrength=0.2,
                  dynamic_levels=1, max_level=20, float_level=3, filter_thre=1e-14, filter_max=1e-8, filter_start=1, filter_rate=2)


class AmgSolver:
    """saena::amg mirror (reference include/saena.hpp:195-265): set_matrix on the host, solve on the GPU."""

    def __init__(self, A: Matrix, opts: OptionsC):
        self.A, self.L = A, A.L
        self.h = self.L.saena_amg_new()
        _check(self.L, self.L.saena_amg_set_matrix(self.h, A.h, C.byref(opts)))
        self.opts = opts

    @property
    def num_levels(self):
        return self.L.saena_amg_num_levels(self.h)

    def level_aggregates(self, l):
        """coarse id of every fine row of level l (one-rank setups), and the number of aggregates"""
        rows = self.level_info(l)["rows"]
        out, n = np.zeros(rows, np.int32), C.c_int()
        _check(self.L, self.L.saena_amg_level_aggregates(self.h, l, out.ctypes.data_as(_PI), C.byref(n)))
        return out, n.value

    def level_info(self, l):
        rows, na, npp, eig = C.c_int(), C.c_long(), C.c_long(), C.c_double()
        _check(self.L, self.L.saena_amg_level_info(self.h, l, C.byref(rows), C.byref(na), C.byref(npp), C.byref(eig)))
        return dict(rows=rows.value, nnzA=na.value, nnzP=npp.value, eig_max=eig.value)

    def level_split(self, l):
        s = np.zeros(self.A.comm.nranks + 1, np.int32)
        _check(self.L, self.L.saena_amg_level_split(self.h, l, s.ctypes.data_as(_PI)))
        return s

    def level_layout(self, l, which):
        d = OpDesc()
        _check(self.L, self.L.saena_amg_level_desc(self.h, l, which, C.byref(d)))
        return desc_arrays(d)

    def to_device(self):
        _check(self.L, self.L.saena_amg_to_device(self.h))
        return self

    def device_handle(self):
        return self.L.saena_amg_device_handle(self.h)

    def device_op(self, l, which=0):
        from . import capi
        op = capi.Operator.__new__(capi.Operator)
        op.h = _VP(self.L.saena_amg_device_op(self.h, l, which))
        info = None
        op._keep = None
        op.destroy = lambda: None          # owned by the solver
        info = op.info()
        op.M, op.N_local = info["M"], info["N_local"]
        return op

    def _solve(self, fn, rhs, cap=256):
        rhs = _ad(rhs)
        u = np.zeros_like(rhs)
        it = C.c_int()
        hist = np.full(cap, np.nan)
        st = fn(self.h, rhs.ctypes.data_as(_PD), u.ctypes.data_as(_PD), C.byref(it), hist.ctypes.data_as(_PD), cap)
        if st not in (0, -6):
            _check(self.L, st)
        return u, it.value, hist[~np.isnan(hist)], st == 0

    def solve(self, rhs):
        return self._solve(self.L.saena_amg_solve, rhs)

    def solve_pCG(self, rhs):
        return self._solve(self.L.saena_amg_solve_pCG, rhs)

    def free(self):
        if self.h:
            self.L.saena_amg_free(self.h)
            self.h = None
