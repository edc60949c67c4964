"""Randomised sweep of the whole-genome batched path: random sets of chromosomes (lengths from shorter than one block to
several blocks, one distance limit / resolution per set) through ChromosomePipeline.run_genome -- all chromosomes side by side
in one band, their blocks mixed in the same launches -- against the CPU oracle's regulator restatement on each chromosome
(case generator: tests/fuzz_cases.genome_case; a seeded slice runs under `pytest -m gpu`).
    python scripts/fuzz_genome.py [n_sets]      (GPU box; FUZZ_SEED=...)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fuzz_cases
from mustache_amd.pipeline import ChromosomePipeline
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 99)))
pipe = ChromosomePipeline(fuzz_cases.OCT)
bad = total = chroms = 0
t0 = time.time()
for case in range(nsets):
    ok, n, d = fuzz_cases.genome_case(rng, pipe)
    total += n
    chroms += d["chromosomes"]
    bad += not ok
    print("set %d %s %s (%d loops so far, %.0f s)" % (case, "ok" if ok else "MISMATCH", d, total, time.time() - t0), flush=True)
print("FUZZ_GENOME sets=%d chromosomes=%d loops=%d mismatches=%d" % (nsets, chroms, total, bad))
