"""Randomised sweep of the two-sample path: the band-direct route (both samples' sigma loops from their bands, difference image
and its blurs inside mst_diff_dog_band, pair p-values, tail) and the dense route (diff_mustache() on dense blocks) against the
CPU oracle's restatement of the reference's diff_mustache(), on random block sizes, distance limits, depths and thresholds
(case generator: tests/fuzz_cases.diff_case; a seeded slice runs under `pytest -m gpu`).
    python scripts/fuzz_diff.py [n_cases]        (GPU box; the oracle needs a few seconds per case; FUZZ_SEED=...)"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fuzz_cases
from mustache_amd.engine import ScaleSpaceEngine
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 4242)))
eng = ScaleSpaceEngine(fuzz_cases.OCT)
bad = total = 0
for case in range(ncases):
    ok, n, d = fuzz_cases.diff_case(rng, eng)
    total += n
    bad += not ok
    print("case %2d %s %s" % (case, "ok" if ok else "MISMATCH", d), flush=True)
print("done: %d cases, %d mismatches; loops compared: %d" % (ncases, bad, total))
