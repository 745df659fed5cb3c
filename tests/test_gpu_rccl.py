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
