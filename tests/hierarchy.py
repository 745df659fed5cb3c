"""Synthetic multigrid hierarchies for V-cycle parity tests (test infrastructure).

Built with scipy so the tests do not depend on the product's own AMG setup:
tentative 2x2x2 aggregation, one damped-Jacobi smoothing step of P, R = P^T,
Ac = R A P.  The same operators go to the oracle and to the GPU.
"""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc


def coo_to_scipy(entries, M, N):
    return sp.csr_matrix((entries["val"], (entries["row"], entries["col"])), shape=(M, N))


def scipy_to_coo(A):
    A = A.tocoo()
    return orc.coo_from_arrays(A.row.astype(np.int32), A.col.astype(np.int32), A.data.astype(np.float64))


def poisson_hierarchy(m, nlevels, omega=2.0 / 3):
    """-> lists A[l], P[l], R[l] of scipy CSR matrices for the interior (m-2)^3 Poisson system."""
    entries, M = orc.laplacian3d(m)
    A0 = coo_to_scipy(entries, M, M)
    n = m - 2
    As, Ps, Rs = [A0], [], []
    dims = (n, n, n)
    for _ in range(nlevels - 1):
        A = As[-1]
        nx, ny, nz = dims
        cx, cy, cz = (nx + 1) // 2, (ny + 1) // 2, (nz + 1) // 2
        k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        agg = ((k // 2) * cy + (j // 2)) * cx + (i // 2)
        Nc = cx * cy * cz
        T = sp.csr_matrix((np.ones(A.shape[0]), (np.arange(A.shape[0]), agg.ravel())), shape=(A.shape[0], Nc))
        Dinv = sp.diags(1.0 / A.diagonal())
        P = (T - omega * (Dinv @ A @ T)).tocsr()
        P.eliminate_zeros()
        R = P.T.tocsr()
        Ac = (R @ A @ P).tocsr()
        Ac.eliminate_zeros()
        Ps.append(P); Rs.append(R); As.append(Ac)
        dims = (cx, cy, cz)
    return As, Ps, Rs


def oracle_hierarchy(As, Ps, Rs, nprocs=1):
    """OracleOps on an even row partition per level"""
    splits = [orc.split_even(A.shape[0], nprocs) for A in As]
    OA = [orc.OracleOp(scipy_to_coo(A), A.shape[0], A.shape[0], splits[l]) for l, A in enumerate(As)]
    OP = [orc.OracleOp(scipy_to_coo(P), P.shape[0], P.shape[1], splits[l], splits[l + 1], square=False) for l, P in enumerate(Ps)]
    OR = [orc.OracleOp(scipy_to_coo(R), R.shape[0], R.shape[1], splits[l + 1], splits[l], square=False) for l, R in enumerate(Rs)]
    return OA, OP, OR


def eig_estimates(As):
    """upper bounds of lambda_max(D^-1 A) (Gershgorin) -- an INPUT of Chebyshev, shared by both sides"""
    out = []
    for A in As:
        d = A.diagonal()
        out.append(float(np.max(np.abs(A).sum(axis=1).A1 / np.abs(d))))
    return out
