"""Loader of the refvc_* fixtures (oracle/ref/make_golden_vcycle.py): a real smoothed-aggregation hierarchy (input)
and what the COMPILED REFERENCE operators computed on it (transfers, composed V-cycles) at 1/2/4 ranks."""
import glob
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "refvc_*.np*.npz")))
VCYCLE_CASES = {"jacobi33": ("jacobi", 3, 3), "jacobi21": ("jacobi", 2, 1), "cheby33": ("chebyshev", 3, 3), "cheby12": ("chebyshev", 1, 2)}


def load(fn):
    tag = os.path.basename(fn).split(".")[0]
    hier = dict(np.load(os.path.join(GOLDEN, tag + ".hier.npz")))
    return hier, dict(np.load(fn))


def coo(hier, name, l):
    M = int(hier[f"{name}{l}_shape"][0])
    rows = np.repeat(np.arange(M, dtype=np.int32), hier[f"{name}{l}_npr"])
    return orc.coo_from_arrays(rows, hier[f"{name}{l}_col"], hier[f"{name}{l}_val"])


def oracle_hierarchy(hier, splits):
    """OracleOps of every level on the given per-level row partitions (R = P^T laid out by the oracle itself)"""
    nl = int(hier["nlevels"])
    OA, OP, OR = [], [], []
    for l in range(nl):
        M = int(hier[f"A{l}_shape"][0])
        OA.append(orc.OracleOp(coo(hier, "A", l), M, M, splits[l]))
        OA[-1].set_eig(float(hier["eig"][l]))
        if l < nl - 1:
            Nc = int(hier[f"P{l}_shape"][1])
            P = coo(hier, "P", l)
            OP.append(orc.OracleOp(P, M, Nc, splits[l], splits[l + 1], square=False))
            Rt = orc.coo_from_arrays(P["col"].copy(), P["row"].copy(), P["val"].copy())
            OR.append(orc.OracleOp(Rt, Nc, M, splits[l + 1], splits[l], square=False))
    return OA, OP, OR


def v2(n):
    g = np.arange(n, dtype=np.float64)
    return np.sin(0.37 * g + 0.1) + 0.25 * np.cos(1.3 * g)


def rhs2(n):
    return np.cos(0.05 * np.arange(n, dtype=np.float64)) - 0.3


def ec(n):
    return np.sin(0.21 * np.arange(n, dtype=np.float64) + 0.4)


def abs_product(hier, name, l, x, transpose=False):
    """sum_j |a_ij x_j| per output row (the scale of a row sum's rounding error)"""
    M, N = (int(v) for v in hier[f"{name}{l}_shape"])
    rows = np.repeat(np.arange(M), hier[f"{name}{l}_npr"])
    cols, vals = hier[f"{name}{l}_col"], hier[f"{name}{l}_val"]
    if transpose:
        return np.bincount(cols, weights=np.abs(vals * x[rows]), minlength=N)
    return np.bincount(rows, weights=np.abs(vals * x[cols]), minlength=M)
