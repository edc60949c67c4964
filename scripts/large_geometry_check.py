"""Whole per-chromosome GPU pipeline against the CPU oracle at geometries beyond BASELINE's (GPU box; the oracle needs about
a minute per 8000 x 8000 block):  500 bp resolution (distance limit 4000 px -> blocks of 8000 x 8000, window 4000) and an odd
distance limit (dpx 3011 -> blocks of 6022), both tile-sharing modes (tests/fuzz_cases.geometry_case; the second geometry also
runs under `pytest -m gpu`).
    python scripts/large_geometry_check.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fuzz_cases
from mustache_amd.pipeline import ChromosomePipeline
pipe = ChromosomePipeline(fuzz_cases.OCT)
bad = total = 0
t0 = time.time()
for n, dpx, res, depth in ((9000, 4000, 500, 200.0), (7000, 3011, 1000, 60.0)):
    ok, nl, d = fuzz_cases.geometry_case(pipe, n, dpx, res, depth, share_modes=(True, False))
    total += nl
    bad += not ok
    print("%s %s  [%.0f s]" % ("ok" if ok else "MISMATCH", d, time.time() - t0), flush=True)
print("done: %d mismatches; loops compared: %d" % (bad, total))
