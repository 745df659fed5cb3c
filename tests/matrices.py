"""Independent (test-side) reader for the SuiteSparse fixtures under tests/golden/matrices/,
applying the rules of the reference's read_file + remove_duplicates (src/saena_matrix.cpp:17-203,
src/saena_matrix_setup.cpp:118-165): symmetric -> mirrored, pattern -> value 1 and mirrored,
duplicates added, |v| <= 1e-14 dropped.  Used to cross-check the product's reader."""
import gzip
import os

import numpy as np

from oracle import oracle as orc

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matrices")
FILES = {"plat362": "plat362.mtx", "SiH4": "SiH4.mtx.gz", "fxm3_6": "fxm3_6.mtx.gz"}


def path(name, tmpdir=None):
    """path of an uncompressed .mtx (decompressing into tmpdir when needed)"""
    fn = os.path.join(DIR, FILES[name])
    if not fn.endswith(".gz"):
        return fn
    out = os.path.join(str(tmpdir), name + ".mtx")
    if not os.path.exists(out):
        with gzip.open(fn, "rb") as f, open(out, "wb") as g:
            g.write(f.read())
    return out


def entries(name):
    """-> (coo entries column-major, M) as the assembled reference matrix holds them"""
    fn = os.path.join(DIR, FILES[name])
    opener = gzip.open if fn.endswith(".gz") else open
    with opener(fn, "rt") as f:
        header = f.readline().split()
        pattern, symmetric = header[3] == "pattern", header[4] == "symmetric"
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        M, N, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, ndmin=2)
    r, c = data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1
    v = np.ones(len(r)) if pattern else data[:, 2]
    if symmetric or pattern:
        off = r != c
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, v[off]])
    key = r * N + c
    order = np.argsort(key, kind="stable")
    key, v = key[order], v[order]
    uk, start = np.unique(key, return_index=True)
    vs = np.add.reduceat(v, start)
    keep = np.abs(vs) > 1e-14
    uk, vs = uk[keep], vs[keep]
    return orc.coo_from_arrays((uk // N).astype(np.int32), (uk % N).astype(np.int32), vs), M
