"""The GPU SpGEMM of the AMG setup (sgpu_spgemm.hip: Ac = (R A) P on the device) against the host kernel: the result
must be the host's BIT FOR BIT -- same pattern after the drop rule, same values (every entry adds its products in the
host's order) -- so that the hierarchy stays the one pinned against the reference's printed sizes and vectors."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys, json, hashlib
import numpy as np
sys.path.insert(0, %(root)r)
from saena_amd import capi, host
capi.init(0)
L = host.load("gpu")
A = host.Matrix(host.Comm("gpu", "rccl")).laplacian3D(%(m)d).assemble()
S = host.AmgSolver(A, host.options(L, **dict(host.OPTIONS001, smoother="chebyshev")))
out = {"levels": S.num_levels}
for l in range(S.num_levels):
    for which in (0, 1, 2):
        if which and l == S.num_levels - 1:
            continue
        d = S.level_layout(l, which)
        h = hashlib.sha256()
        for k in ("nnzPerRow_local", "col_local", "val_local"):
            h.update(np.ascontiguousarray(d[k]).tobytes())
        out[f"{l}.{which}"] = [int(d["M"]), int(len(d["col_local"])), h.hexdigest()]
    out[f"eig{l}"] = S.level_info(l)["eig_max"]
# the public product C = A A as well (saena::amg::matmat)
C = A.matmat(A)
d = C.layout()
out["AA"] = [int(len(d["col_local"])), hashlib.sha256(np.ascontiguousarray(d["col_local"]).tobytes() + np.ascontiguousarray(d["val_local"]).tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
"""


def _run(m, host_spgemm, hbm_accumulator=False):
    env = dict(os.environ)
    if host_spgemm:
        env["SAENA_HOST_SPGEMM"] = "1"
    if hbm_accumulator:
        env["SAENA_SPGEMM_NO_LDS"] = "1"            # heavy rows on the dense accumulator in HBM (the only form before round 3)
    env["SAENA_SETUP_TIMING"] = "1"
    out = subprocess.run([sys.executable, "-c", WORKER % dict(root=ROOT, m=m)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    return res, out.stderr


@pytest.mark.parametrize("m", [24, 64])
def test_gpu_spgemm_builds_the_host_hierarchy_bit_for_bit(m):
    gpu, err_gpu = _run(m, host_spgemm=False)
    ref, err_host = _run(m, host_spgemm=True)
    assert "[spgemm gpu]" in err_gpu and "[spgemm gpu]" not in err_host, "the first run must have used the device kernel, the second the host's"
    assert gpu == ref, {k: (gpu.get(k), ref.get(k)) for k in set(gpu) | set(ref) if gpu.get(k) != ref.get(k)}
    if m == 64:       # all accumulators were exercised: light (wave/row), medium (workgroup/row), heavy (dense, in LDS windows)
        assert gpu["levels"] >= 6
        hbm, err_hbm = _run(m, host_spgemm=False, hbm_accumulator=True)        # ... and heavy with the dense accumulator in HBM
        assert "[spgemm gpu]" in err_hbm and hbm == ref
