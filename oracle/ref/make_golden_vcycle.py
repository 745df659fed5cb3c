#!/usr/bin/env python3
"""Vector-level fixtures of the grid transfers and of a whole V-cycle on a REAL smoothed-aggregation hierarchy, from
the COMPILED REFERENCE operators (oracle/_ref/ref_vcycle, see ref_vcycle.cpp).

TEST INFRASTRUCTURE ONLY; runs only in the build container.  The hierarchy (A_l, P_l) is built by the product's host
setup (libsaena_host.so, CPU) and handed to the reference's own saena_matrix / prolong_matrix / restrict_matrix as
coordinate lists; every OUTPUT array in a fixture (R v, P e in fp64 and fp32-halo form, composed V-cycles) was computed
by the reference's classes.  The fixture also stores the hierarchy, so it is self-contained: inputs + expected outputs.

    python oracle/ref/make_golden_vcycle.py
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from saena_amd import host                     # noqa: E402  (host-only library: no GPU involved)

REFBIN = os.path.join(ROOT, "oracle", "_ref", "ref_vcycle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MPIRUN = "/opt/conda/bin/mpirun"
REFDATA = "/root/reference/data"

CASES = [
    ("poisson16", dict(kind="poisson", m=16), (1, 2, 4, 8)),
    ("plat362", dict(kind="file", path=f"{REFDATA}/old/plat362.mtx"), (1, 2)),
]


def hierarchy(case):
    L = host.load("host")
    A = host.Matrix(host.Comm("host", "self"))
    if case["kind"] == "poisson":
        A.laplacian3D(case["m"])
    else:
        A.read_file(case["path"])
    A.assemble()
    S = host.AmgSolver(A, host.options(L, **dict(host.OPTIONS001, smoother="chebyshev")))      # chebyshev: eig estimates are computed
    out = {"nlevels": S.num_levels, "eig": np.array([S.level_info(l)["eig_max"] for l in range(S.num_levels)])}
    for l in range(S.num_levels):
        for which, name in ((0, "A"), (1, "P")):
            if which == 1 and l == S.num_levels - 1:
                continue
            d = S.level_layout(l, which)
            out[f"{name}{l}_npr"], out[f"{name}{l}_col"], out[f"{name}{l}_val"] = d["nnzPerRow_local"], d["col_local"], d["val_local"]
            out[f"{name}{l}_shape"] = np.array([d["M"], d["N_local"]], np.int64)
    return out


def write_coo(fn, npr, col, val, M, N):
    rows = np.repeat(np.arange(M, dtype=np.int32), npr)
    with open(fn, "wb") as f:
        f.write(struct.pack("<iiq", M, N, len(col)))
        rec = np.zeros(len(col), dtype=[("r", "<i4"), ("c", "<i4"), ("v", "<f8")])
        rec["r"], rec["c"], rec["v"] = rows, col, val
        f.write(rec.tobytes())


def main():
    if not os.path.exists(REFBIN):
        sys.exit("build the reference driver first: make -C oracle ref")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib", OMP_NUM_THREADS="1")
    DT = {"f64": np.float64, "i32": np.int32, "i64": np.int64}
    for tag, case, nps in CASES:
        H = hierarchy(case)
        tin, tout = tempfile.mkdtemp(prefix="refvc_in_"), tempfile.mkdtemp(prefix="refvc_out_")
        try:
            nl = H["nlevels"]
            for l in range(nl):
                M, N = H[f"A{l}_shape"]
                write_coo(os.path.join(tin, f"A{l}.coo"), H[f"A{l}_npr"], H[f"A{l}_col"], H[f"A{l}_val"], int(M), int(N))
                if l < nl - 1:
                    M, N = H[f"P{l}_shape"]
                    write_coo(os.path.join(tin, f"P{l}.coo"), H[f"P{l}_npr"], H[f"P{l}_col"], H[f"P{l}_val"], int(M), int(N))
            np.savetxt(os.path.join(tin, "eig.txt"), H["eig"], fmt="%.17g")
            for p in nps:
                out = subprocess.run([MPIRUN, "-np", str(p), REFBIN, tin, tout, tag, str(nl)], env=env, check=True,
                                     capture_output=True, text=True, timeout=1800)
                sys.stdout.write(out.stdout)
            groups = {}
            for fn in sorted(os.listdir(tout)):
                t, npart, name, dt = fn.split(".")
                groups.setdefault(npart, {})[name] = np.fromfile(os.path.join(tout, fn), dtype=DT[dt])
            # the hierarchy (the INPUT) once per case, the reference's outputs once per rank count
            np.savez_compressed(os.path.join(GOLDEN, f"refvc_{tag}.hier.npz"), **{k: v for k, v in H.items() if k != "nlevels"}, nlevels=np.int64(nl))
            for npart, arrs in groups.items():
                np.savez_compressed(os.path.join(GOLDEN, f"refvc_{tag}.{npart}.npz"), **arrs)
                print("wrote", f"refvc_{tag}.{npart}.npz", "levels", nl, "rows", [int(H[f'A{l}_shape'][0]) for l in range(nl)])
        finally:
            shutil.rmtree(tin, ignore_errors=True)
            shutil.rmtree(tout, ignore_errors=True)


if __name__ == "__main__":
    main()
