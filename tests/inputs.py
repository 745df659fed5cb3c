"""Closed-form test inputs shared by the fixture generator (oracle/ref/ref_dump.cpp) and the tests."""
import numpy as np


def v_sin(n, ofs=0):
    g = np.arange(ofs, ofs + n, dtype=np.float64)
    return np.sin(0.001 * g)


def v2(n, ofs=0):
    g = np.arange(ofs, ofs + n, dtype=np.float64)
    return np.sin(0.37 * g + 0.1) + 0.25 * np.cos(1.3 * g)


def rhs2(n, ofs=0):
    g = np.arange(ofs, ofs + n, dtype=np.float64)
    return np.cos(0.05 * g) - 0.3


def ec(n, ofs=0):
    g = np.arange(ofs, ofs + n, dtype=np.float64)
    return np.sin(0.21 * g + 0.4)


def synthetic_P(Mbig):
    """P(i,j) = 1/(1+|i-2j|) + 0.001 i, j in {i/2-1, i/2, i/2+1} clipped to [0, Nc)."""
    Nc = (Mbig + 1) // 2
    rows, cols, vals = [], [], []
    for i in range(Mbig):
        for j in range(i // 2 - 1, i // 2 + 2):
            if 0 <= j < Nc:
                rows.append(i)
                cols.append(j)
                vals.append(1.0 / (1 + abs(i - 2 * j)) + 0.001 * i)
    return np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals, np.float64), Nc
