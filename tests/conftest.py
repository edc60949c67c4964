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
    """The libraries are git-ignored (built artefacts); build them on first use so a fresh checkout can run the suite."""
    libs = [os.path.join(ROOT, "mustache_amd", n) for n in ("libmustache_hip.so", "libmustache_io.so")]
    if not all(os.path.exists(lib) for lib in libs):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mustache_amd", "csrc"), "-j4"])


_ensure_library()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
