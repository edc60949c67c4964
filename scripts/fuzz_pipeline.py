"""Randomised end-to-end sweep: the whole per-chromosome GPU pipeline (COO -> band -> normalisation -> band-direct fused
kernel -> device FDR/selection -> batched tail -> overlap masks) against the CPU oracle's regulator restatement, on random
chromosome lengths, distance limits, resolutions (both normalisation branches), depths and thresholds, in both tile-sharing
modes (case generator: tests/fuzz_cases.pipeline_case; a seeded slice runs under `pytest -m gpu`).
    python scripts/fuzz_pipeline.py [n_cases]      (GPU box; the oracle needs ~3 s per 2000 x 2000 block; FUZZ_SEED=...)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fuzz_cases
from mustache_amd.pipeline import ChromosomePipeline
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 77)))
pipe = ChromosomePipeline(fuzz_cases.OCT)
bad = total = 0
t0 = time.time()
for case in range(ncases):
    ok, n, d = fuzz_cases.pipeline_case(rng, pipe, share_modes=(True, False), wide=(case % 5 == 4))
    total += n
    bad += not ok
    print("case %2d %s %s  [%.0f s]" % (case, "ok" if ok else "MISMATCH", d, time.time() - t0), flush=True)
print("done: %d cases, %d mismatches; loops compared: %d" % (ncases, bad, total))
