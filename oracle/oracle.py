"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under saena_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

index_t = C.c_int
nnz_t = C.c_long
value_t = C.c_double


class Coo(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("val", C.c_double)]


COO_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("val", np.float64)])

_PI = C.POINTER(C.c_int)
_PD = C.POINTER(C.c_double)
_PF = C.POINTER(C.c_float)
_PL = C.POINTER(C.c_long)


class RankOp(C.Structure):
    _fields_ = [
        ("rank", C.c_int), ("nprocs", C.c_int),
        ("M", C.c_int), ("row_ofst", C.c_int), ("col_ofst", C.c_int),
        ("nnz_l", C.c_long), ("nnz_l_local", C.c_long), ("nnz_l_remote", C.c_long),
        ("col_remote_size", C.c_int),
        ("nnzPerRow_local", _PI), ("row_local", _PI), ("col_local", _PI), ("val_local", _PD),
        ("nnzPerCol_remote", _PI), ("row_remote", _PI), ("col_remote", _PI), ("col_remote2", _PI),
        ("val_remote", _PD), ("nnzPerProcScan", _PL),
        ("numRecvProc", C.c_int), ("numSendProc", C.c_int),
        ("recvProcRank", _PI), ("recvProcCount", _PI), ("sendProcRank", _PI), ("sendProcCount", _PI),
        ("recvCount", _PI), ("sendCount", _PI), ("vdispls", _PI), ("rdispls", _PI),
        ("vIndexSize", C.c_int), ("recvSize", C.c_int),
        ("vIndex", _PI), ("vSend", _PD), ("vecValues", _PD), ("vSend_f", _PF), ("vecValues_f", _PF),
        ("inv_diag", _PD), ("temp1", _PD), ("temp2", _PD),
    ]


class Op(C.Structure):
    _fields_ = [
        ("nprocs", C.c_int), ("Mbig", C.c_int), ("Nbig", C.c_int), ("nnz_g", C.c_long),
        ("split_row", _PI), ("split_col", _PI), ("r", C.POINTER(RankOp)),
        ("eig_max_of_invdiagXA", C.c_double), ("jacobi_omega", C.c_float), ("use_double", C.c_int),
    ]


class Grid(C.Structure):
    pass


Grid._fields_ = [
    ("level", C.c_int), ("A", C.POINTER(Op)), ("P", C.POINTER(Op)), ("R", C.POINTER(Op)),
    ("coarse", C.POINTER(Grid)), ("res", _PD), ("uCorr", _PD), ("res_coarse", _PD), ("uCorrCoarse", _PD),
]


class Amg(C.Structure):
    _fields_ = [
        ("max_level", C.c_int), ("grids", C.POINTER(Grid)),
        ("preSmooth", C.c_int), ("postSmooth", C.c_int), ("smoother", C.c_int),
        ("CG_coarsest_max_iter", C.c_int), ("CG_coarsest_tol", C.c_double),
        ("solver_max_iter", C.c_int), ("solver_tol", C.c_double),
    ]


def build(force=False):
    """(Re)build liboracle.so with the committed Makefile."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "saena_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        pc = C.POINTER(Coo)
        L.orc_laplacian3d.restype = C.c_long
        L.orc_laplacian3d.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(pc), _PI]
        L.orc_laplacian3d_rhs.argtypes = [C.c_int, C.c_int, C.c_int, _PD]
        L.orc_band_matrix.restype = C.c_long
        L.orc_band_matrix.argtypes = [C.c_int, C.c_int, C.POINTER(pc)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_sort_colmajor.argtypes = [C.c_void_p, C.c_long]
        L.orc_split_even.argtypes = [C.c_int, C.c_int, _PI]
        L.orc_split_nnz.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, _PI]
        L.orc_op_build.restype = C.POINTER(Op)
        L.orc_op_build.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, _PI, _PI, C.c_int, C.c_int]
        L.orc_op_free.argtypes = [C.POINTER(Op)]
        for f in ("orc_matvec", "orc_matvec_float"):
            getattr(L, f).argtypes = [C.POINTER(Op), _PD, _PD]
        L.orc_matvec_dense.argtypes = [C.POINTER(Op), _PD, _PD, C.c_int]
        for f in ("orc_residual", "orc_residual_negative"):
            getattr(L, f).argtypes = [C.POINTER(Op), _PD, _PD, _PD]
        L.orc_jacobi.argtypes = [C.POINTER(Op), C.c_int, _PD, _PD]
        L.orc_chebyshev.argtypes = [C.POINTER(Op), C.c_int, _PD, _PD]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [_PD, _PD, _PI, C.c_int]
        L.orc_amg_create.restype = C.POINTER(Amg)
        L.orc_amg_create.argtypes = [C.c_int, C.POINTER(C.POINTER(Op)), C.POINTER(C.POINTER(Op)), C.POINTER(C.POINTER(Op))]
        L.orc_amg_free.argtypes = [C.POINTER(Amg)]
        L.orc_solve_coarsest_CG.restype = C.c_int
        L.orc_solve_coarsest_CG.argtypes = [C.POINTER(Amg), C.POINTER(Op), _PD, _PD]
        L.orc_vcycle.argtypes = [C.POINTER(Amg), C.POINTER(Grid), _PD, _PD]
        for f in ("orc_solve", "orc_solve_pCG", "orc_solve_CG", "orc_solve_smoother"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.POINTER(Amg), _PD, _PD, _PD, C.c_int]
        L.orc_time_matvec.restype = C.c_double
        L.orc_time_matvec.argtypes = [C.POINTER(Op), _PD, _PD, C.c_int, C.c_int]
        L.orc_time_jacobi.restype = C.c_double
        L.orc_time_jacobi.argtypes = [C.POINTER(Op), _PD, _PD, C.c_int, C.c_int]
        _LIB = L
    return _LIB


def _pd(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_PD)


def _pi(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_PI)


def _take_coo(ptr, n):
    buf = (Coo * n).from_address(C.addressof(ptr.contents))
    arr = np.frombuffer(buf, dtype=COO_DTYPE, count=n).copy()
    lib().orc_free(ptr)
    return arr


def laplacian3d(mx, my=None, mz=None):
    """-> (coo entries column-major, Mbig): the reference's laplacian3D after boundary removal."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    p = C.POINTER(Coo)()
    mbig = C.c_int()
    n = lib().orc_laplacian3d(mx, my, mz, C.byref(p), C.byref(mbig))
    return _take_coo(p, n), mbig.value


def laplacian3d_rhs(mx, my=None, mz=None):
    my = mx if my is None else my
    mz = mx if mz is None else mz
    rhs = np.empty((mx - 2) * (my - 2) * (mz - 2))
    lib().orc_laplacian3d_rhs(mx, my, mz, _pd(rhs))
    return rhs


def band_matrix(M, bw):
    p = C.POINTER(Coo)()
    n = lib().orc_band_matrix(M, bw, C.byref(p))
    return _take_coo(p, n)


def coo_from_arrays(row, col, val):
    e = np.empty(len(row), dtype=COO_DTYPE)
    e["row"], e["col"], e["val"] = row, col, val
    lib().orc_sort_colmajor(e.ctypes.data, len(e))
    return e


def split_even(Mbig, nprocs):
    s = np.zeros(nprocs + 1, np.int32)
    lib().orc_split_even(Mbig, nprocs, _pi(s))
    return s


def split_nnz(entries, Mbig, nprocs):
    s = np.zeros(nprocs + 1, np.int32)
    lib().orc_split_nnz(entries.ctypes.data, len(entries), Mbig, nprocs, _pi(s))
    return s


class OracleOp:
    """A distributed operator (all simulated ranks) in the reference's layout."""

    def __init__(self, entries, Mbig, Nbig, split_row, split_col=None, square=True):
        entries = np.ascontiguousarray(entries, dtype=COO_DTYPE)
        split_row = np.ascontiguousarray(split_row, np.int32)
        split_col = split_row if split_col is None else np.ascontiguousarray(split_col, np.int32)
        self.nprocs = len(split_row) - 1
        self.Mbig, self.Nbig = int(Mbig), int(Nbig)
        self.split_row, self.split_col = split_row.copy(), split_col.copy()
        self.p = lib().orc_op_build(entries.ctypes.data, len(entries), Mbig, Nbig, _pi(split_row), _pi(split_col),
                                    self.nprocs, 1 if square else 0)
        self.c = self.p.contents

    def __del__(self):
        if getattr(self, "p", None) is not None and _LIB is not None:
            _LIB.orc_op_free(self.p)
            self.p = None

    # --- layout access (copies) ---
    def rank(self, r):
        return self.c.r[r]

    def rank_array(self, r, name, n, dtype):
        ptr = getattr(self.c.r[r], name)
        if n == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    def set_eig(self, e):
        self.c.eig_max_of_invdiagXA = float(e)

    def set_use_double(self, flag):
        self.c.use_double = 1 if flag else 0

    # --- hot path ---
    def matvec(self, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.size == self.Nbig
        w = np.empty(self.Mbig)
        lib().orc_matvec(self.p, _pd(v), _pd(w))
        return w

    def matvec_float(self, v):
        v = np.ascontiguousarray(v, np.float64)
        w = np.empty(self.Mbig)
        lib().orc_matvec_float(self.p, _pd(v), _pd(w))
        return w

    def matvec_dense(self, v, as_float=False):
        """saena_matrix_dense::matvec_dense(_float) on the dense form of this (square) operator"""
        v = np.ascontiguousarray(v, np.float64)
        w = np.empty(self.Mbig)
        lib().orc_matvec_dense(self.p, _pd(v), _pd(w), 1 if as_float else 0)
        return w

    def residual(self, u, rhs):
        res = np.empty(self.Mbig)
        lib().orc_residual(self.p, _pd(np.ascontiguousarray(u)), _pd(np.ascontiguousarray(rhs)), _pd(res))
        return res

    def residual_negative(self, u, rhs):
        """res = rhs - A u: saena_matrix::residual_negative (include/saena_matrix.tpp:26-33) -- matvec, then res[i] = rhs[i] - res[i]"""
        return np.asarray(rhs, np.float64) - self.matvec(u)

    def residual_multiply(self, u, rhs, w, c):
        """res = c * w o (rhs - A u): saena_matrix::residual_multiply (include/saena_matrix.tpp:35-43) -- matvec, then
        res[i] = c * w[i] * (rhs[i] - res[i]), the product left to right"""
        return (float(c) * np.asarray(w, np.float64)) * (np.asarray(rhs, np.float64) - self.matvec(u))

    def jacobi(self, it, u, rhs):
        u = np.array(u, np.float64)
        lib().orc_jacobi(self.p, it, _pd(u), _pd(np.ascontiguousarray(rhs, np.float64)))
        return u

    def chebyshev(self, it, u, rhs):
        u = np.array(u, np.float64)
        lib().orc_chebyshev(self.p, it, _pd(u), _pd(np.ascontiguousarray(rhs, np.float64)))
        return u

    def time_matvec(self, v, reps, threads):
        w = np.empty(self.Mbig)
        return lib().orc_time_matvec(self.p, _pd(np.ascontiguousarray(v)), _pd(w), reps, threads)

    def time_jacobi(self, u, rhs, reps, threads):
        u = np.array(u, np.float64)
        return lib().orc_time_jacobi(self.p, _pd(u), _pd(np.ascontiguousarray(rhs)), reps, threads)


class OracleAmg:
    """Multigrid hierarchy over OracleOps (A[l], P[l], R[l])."""

    def __init__(self, A, P, R, pre=3, post=3, smoother="jacobi", max_iter=100, tol=1e-8):
        self.A, self.P, self.R = list(A), list(P), list(R)
        n = len(A)
        PA = (C.POINTER(Op) * n)(*[a.p for a in A])
        PP = (C.POINTER(Op) * n)(*([p.p for p in P] + [None] * (n - len(P))))
        PR = (C.POINTER(Op) * n)(*([r.p for r in R] + [None] * (n - len(R))))
        self.p = lib().orc_amg_create(n, PA, PP, PR)
        c = self.p.contents
        c.preSmooth, c.postSmooth = pre, post
        c.smoother = 0 if smoother == "jacobi" else 1
        c.solver_max_iter, c.solver_tol = max_iter, tol

    def __del__(self):
        if getattr(self, "p", None) is not None and _LIB is not None:
            _LIB.orc_amg_free(self.p)
            self.p = None

    def set_solver(self, max_iter, tol):
        self.p.contents.solver_max_iter, self.p.contents.solver_tol = max_iter, tol

    def vcycle(self, u, rhs):
        u = np.array(u, np.float64)
        rhs = np.ascontiguousarray(rhs, np.float64)
        lib().orc_vcycle(self.p, self.p.contents.grids, _pd(u), _pd(rhs))
        return u

    def coarsest_cg(self, rhs):
        A = self.A[-1]
        u = np.zeros(A.Mbig)
        it = lib().orc_solve_coarsest_CG(self.p, A.p, _pd(u), _pd(np.ascontiguousarray(rhs, np.float64)))
        return u, it

    def _solve(self, fn, rhs, cap=256):
        u = np.zeros(self.A[0].Mbig)
        hist = np.full(cap, np.nan)
        it = fn(self.p, _pd(u), _pd(np.ascontiguousarray(rhs, np.float64)), _pd(hist), cap)
        return u, it, hist[~np.isnan(hist)]

    def solve(self, rhs):
        return self._solve(lib().orc_solve, rhs)

    def solve_pCG(self, rhs):
        return self._solve(lib().orc_solve_pCG, rhs)

    def solve_CG(self, rhs, cap=2048):
        return self._solve(lib().orc_solve_CG, rhs, cap)

    def solve_smoother(self, rhs, cap=2048):
        return self._solve(lib().orc_solve_smoother, rhs, cap)
