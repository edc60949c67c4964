"""Replayed hipGraphs (MST_FLAG_GRAPH) against plain launches under the conditions that exposed the memset-node fault of
LABBOOK.md R4.6: several call signatures alternating, more than the per-thread caches hold (graphs are evicted, destroyed and
captured again), the finish's scratch and the host summary poisoned before every call.  The summary -- flags word, record counts,
tested-pixel counts -- must equal the first plain pass of the same configuration every time."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_replayed_graphs_equal_plain_launches_under_poison():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "graph_stress.py"), "600", "1", "9"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "failures none" in out.stdout, out.stdout[-2000:]
