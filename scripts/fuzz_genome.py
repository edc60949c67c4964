"""Randomised sweep of the whole-genome batched path: random sets of chromosomes (lengths from shorter than one block to
several blocks, one distance limit / resolution per set) through ChromosomePipeline.run_genome -- all chromosomes side by side
in one band, their blocks mixed in the same launches -- against the CPU oracle's regulator restatement on each chromosome.
    python scripts/fuzz_genome.py [n_sets]      (GPU box)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
import oracle
from mustache_amd.pipeline import ChromosomePipeline
from mustache_amd.synth import synth_coo

nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 99)))
pipe = ChromosomePipeline([1.6, 3.2])
bad = total = chroms = 0
t0 = time.time()
for case in range(nsets):
    dpx = int(rng.integers(60, 420))
    res = int(rng.choice([1000, 2000, 5000, 10000]))
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.1, 0.3]))
    pipe.blocks_per_launch = lambda CH, k=int(rng.integers(1, 9)): k          # blocks of several chromosomes per launch
    k = int(rng.integers(2, 6))
    coos = []
    for c in range(k):
        n = int(rng.integers(max(dpx + 50, 300), 5200))
        depth = float(rng.choice([5.0, 40.0, 300.0]))
        coos.append(synth_coo(n, dpx, depth=depth, seed=int(rng.integers(0, 10 ** 6)), nloops=max(n // 25, 4)))
    bands, ns = zip(*[pipe.normalized_band(x, y, v.copy(), res, dpx) for x, y, v in coos])
    got_all = pipe.run_genome(list(bands), list(ns), dpx, st, pt)
    for c, ((x, y, v), got) in enumerate(zip(coos, got_all)):
        exp = sorted(oracle.regulator_coo(x, y, v.copy(), res, dpx, [1.6, 3.2], st, pt), key=lambda r: (int(r[0]), int(r[1])))
        got = sorted(got, key=lambda r: (int(r[0]), int(r[1])))
        same = [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
        qerr = max([abs(g[2] - e[2]) / max(e[2], 1e-300) for g, e in zip(got, exp)], default=0.0) if same else float("nan")
        total += len(exp)
        chroms += 1
        if not same or qerr > 1e-6:
            bad += 1
            print("MISMATCH set %d chromosome %d: n=%d dpx=%d res=%d st=%g pt=%g  got %d exp %d qerr %g"
                  % (case, c, ns[c], dpx, res, st, pt, len(got), len(exp), qerr), flush=True)
    print("set %d: %d chromosomes, dpx=%d res=%d ok so far (%d loops, %.0f s)" % (case, k, dpx, res, total, time.time() - t0),
          flush=True)
print("FUZZ_GENOME sets=%d chromosomes=%d loops=%d mismatches=%d" % (nsets, chroms, total, bad))
