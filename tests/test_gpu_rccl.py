"""RCCL code paths on one GPU through a 1-rank communicator (see tests/rccl_loopback.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("torch_first", [False, True], ids=["system-rocm", "torch-runtime"])
def test_rccl_loopback(torch_first):
    cmd = [sys.executable, "-m", "tests.rccl_loopback"] + (["--torch"] if torch_first else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_LOOPBACK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
