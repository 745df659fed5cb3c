#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/ref_dump).

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (it needs
/root/reference to have been compiled by `make -C oracle ref`); the GPU box
consumes the committed .npz files.  Every array in a fixture is either a
closed-form input parameter or an output of the reference's own operators.

    python oracle/ref/make_golden.py            # small fixtures (committed)
    python oracle/ref/make_golden.py --norms    # + norm pins at 32^3 / 128^3
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MPIRUN = "/opt/conda/bin/mpirun"

DT = {"f64": np.float64, "i32": np.int32, "i64": np.int64}

# (kind, args) x rank counts
CASES = [
    (("poisson", "8"), (1, 2, 4)),
    (("poisson", "12"), (1, 2, 4)),
    (("poisson", "16"), (1, 3, 8)),      # 8 ranks: north_star's count (343 / 300 / 386-row parts: the nparts^2-bucket partition)
    (("band", "300", "7"), (1, 2, 4)),
    (("band", "64", "63"), (1, 2)),
]
# SuiteSparse matrices shipped in the reference's data/ (data/florida_matrices.txt names bcsstk28 and plat362)
REFDATA = "/root/reference/data"
FILE_CASES = [
    (("file", f"{REFDATA}/old/plat362.mtx", "plat362"), (1, 2)),
    (("file", f"{REFDATA}/FloridaCollection/SiH4.mtx", "SiH4"), (1, 4)),          # irregular rows (1..~250 nnz)
    (("file", f"{REFDATA}/FloridaCollection/fxm3_6.mtx", "fxm3_6"), (1, 3)),      # pattern symmetric
    # (bcsstk28 aborts inside the reference itself: "Saena not working on this one", data/florida_matrices.txt)
]
NORM_CASES = [(("norms", "32"), (1, 2, 4)), (("norms", "128"), (1, 8))]


def run(kind_args, nprocs, outdir):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/usr/lib/x86_64-linux-gnu:/opt/conda/lib"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [MPIRUN, "-np", str(nprocs), REFBIN, outdir, *kind_args]
    out = subprocess.run(cmd, env=env, check=True, capture_output=True, text=True, timeout=3600)
    sys.stdout.write(out.stdout)


def pack(outdir):
    """raw files  <tag>.np<P>.<name>.<dtype>  ->  {tag.npP: {name: array}}"""
    groups = {}
    for fn in sorted(os.listdir(outdir)):
        tag, npart, name, dt = fn.split(".")
        arr = np.fromfile(os.path.join(outdir, fn), dtype=DT[dt])
        groups.setdefault(f"{tag}.{npart}", {})[name] = arr
    return groups


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--norms", action="store_true", help="also run the 32^3/128^3 norm pins (slow)")
    args = ap.parse_args()
    if not os.path.exists(REFBIN):
        sys.exit("build the reference driver first: make -C oracle ref")
    os.makedirs(GOLDEN, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="refdump_")
    try:
        for kind_args, nps in CASES + FILE_CASES:
            for p in nps:
                run(kind_args, p, tmp)
        for key, arrs in pack(tmp).items():
            np.savez_compressed(os.path.join(GOLDEN, f"ref_{key}.npz"), **arrs)
        shutil.rmtree(tmp)
        if args.norms:
            tmp = tempfile.mkdtemp(prefix="refdump_")
            pins = {}
            for kind_args, nps in NORM_CASES:
                for p in nps:
                    run(kind_args, p, tmp)
            for key, arrs in pack(tmp).items():
                pins[key] = {"Av_sq": arrs["pins"][0], "jacobi3_sq": arrs["pins"][1], "cheby3_sq": arrs["pins"][2]}
            with open(os.path.join(GOLDEN, "ref_norm_pins.json"), "w") as f:
                json.dump(pins, f, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
