#!/usr/bin/env python3
"""Vector-level fixtures of the smoothed-aggregation setup from the COMPILED REFERENCE (oracle/_ref/ref_sa, see
ref_sa.cpp): for every level of a hierarchy the coarse id of every fine row (find_aggregation + aggregate_index_update),
the coarse partition, and the smoothed prolongation P = (I - omega D^-1 A) P_t entry by entry (saena_object::SA,
src/saena_object_setup1.cpp:8-254), at 1, 2 and 4 MPI ranks.

TEST INFRASTRUCTURE ONLY; runs only in the build container.  The INPUT of level l is the operator A_l of the product's
host hierarchy (the reference's own Galerkin product needs MKL and is not buildable here); the reference assembles it itself
and every OUTPUT array in a fixture was computed by the reference's classes.  tests/test_sa_pins.py compares the product's
aggregates (bit-exact) and P (pattern exact, values to 1e-14) with them.

    python oracle/ref/make_golden_sa.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_vcycle import hierarchy, write_coo, MPIRUN, REFDATA, GOLDEN      # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref", "ref_sa")
CONN = 0.2                                     # data/options001.xml conn_str (host.OPTIONS001)

CASES = [
    ("poisson8", dict(kind="poisson", m=8), (1, 2, 4)),
    ("poisson12", dict(kind="poisson", m=12), (1, 2, 4)),
    ("poisson16", dict(kind="poisson", m=16), (1, 2, 4, 8)),
    ("plat362", dict(kind="file", path=f"{REFDATA}/old/plat362.mtx"), (1, 2)),
    ("poisson24", dict(kind="poisson", m=24), (1, 3, 8)),
]


def main():
    if not os.path.exists(REFBIN):
        sys.exit("build the reference driver first: make -C oracle ref")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib", OMP_NUM_THREADS="1")
    DT = {"f64": np.float64, "i32": np.int32, "i64": np.int64}
    for tag, case, nps in CASES:
        H = hierarchy(case)
        nl = H["nlevels"]
        if nl < 2:
            print(tag, ": a one-level hierarchy pins nothing, skipped")
            continue
        tin, tout = tempfile.mkdtemp(prefix="refsa_in_"), tempfile.mkdtemp(prefix="refsa_out_")
        try:
            for l in range(nl):
                M, N = H[f"A{l}_shape"]
                write_coo(os.path.join(tin, f"A{l}.coo"), H[f"A{l}_npr"], H[f"A{l}_col"], H[f"A{l}_val"], int(M), int(N))
            for p in nps:
                out = subprocess.run([MPIRUN, "-np", str(p), REFBIN, tin, tout, tag, str(nl), repr(CONN)], env=env, check=True,
                                     capture_output=True, text=True, timeout=1800)
                sys.stdout.write(out.stdout)
            groups = {}
            for fn in sorted(os.listdir(tout)):
                t, npart, name, dt = fn.split(".")
                groups.setdefault(npart, {})[name] = np.fromfile(os.path.join(tout, fn), dtype=DT[dt])
            # the operators A_l (the INPUT) once per case, the reference's outputs once per rank count
            np.savez_compressed(os.path.join(GOLDEN, f"refsa_{tag}.hier.npz"), nlevels=np.int64(nl), conn=np.float64(CONN),
                                **{k: v for k, v in H.items() if k.startswith("A")})
            for npart, arrs in groups.items():
                np.savez_compressed(os.path.join(GOLDEN, f"refsa_{tag}.{npart}.npz"), **arrs)
                print("wrote", f"refsa_{tag}.{npart}.npz", "levels", nl, "rows", [int(H[f'A{l}_shape'][0]) for l in range(nl)],
                      "aggregates", [int(arrs[f"Pshape{l}"][1]) for l in range(nl - 1)])
        finally:
            shutil.rmtree(tin, ignore_errors=True)
            shutil.rmtree(tout, ignore_errors=True)


if __name__ == "__main__":
    main()
