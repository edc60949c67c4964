import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU oracle runs")


def _ensure_library():
    """The HIP library is git-ignored (built artefact); build it on first use so a fresh checkout can run the suite."""
    lib = os.path.join(ROOT, "mustache_amd", "libmustache_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mustache_amd", "csrc"), "-j4"])


_ensure_library()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
