"""RCCL code paths on one GPU through a 1-rank communicator (see tests/rccl_loopback.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# how the two streams of a multi-rank apply depend on each other: flags polled by kernels (default),
# stream value operations, events
SYNC_MODES = {
    "kernel-flags": {},
    "stream-value-ops": {"SAENA_NO_INKERNEL_SYNC": "1"},
    "events": {"SAENA_NO_INKERNEL_SYNC": "1", "SAENA_NO_STREAM_VALUE_OPS": "1"},
}


@pytest.mark.parametrize("torch_first,mode", [(False, "kernel-flags"), (True, "kernel-flags"), (True, "stream-value-ops"), (True, "events")],
                         ids=["system-rocm", "torch-runtime", "torch-runtime-value-ops", "torch-runtime-events"])
def test_rccl_loopback(torch_first, mode):
    cmd = [sys.executable, "-m", "tests.rccl_loopback"] + (["--torch"] if torch_first else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **SYNC_MODES[mode])
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_LOOPBACK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2` exactly as the driver launches it, with both ranks on this one card: SAENA_BENCH_NO_RCCL=1
    routes halos through the host transport (RCCL refuses two ranks per device).  stdout must be ONE JSON line whose
    SpMV self-check -- halo values included -- passes."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SAENA_BENCH_NO_RCCL="1", SAENA_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-vcycle"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rows_per_gpu"] == 2000376
    assert d["check"]["ok"] is True, d["check"]
