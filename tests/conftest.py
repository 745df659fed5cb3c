import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the plan cache (what the plan-time autotune chose, kept between processes) is off under test unless a test points it at
# a directory of its own: tests must not depend on what an earlier run left in ~/.cache
os.environ.setdefault("SAENA_PLAN_CACHE", "off")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
