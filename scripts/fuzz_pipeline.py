"""Randomised end-to-end sweep: the whole per-chromosome GPU pipeline (COO -> band -> normalisation -> band-direct fused
kernel -> device FDR/selection -> batched tail -> overlap masks) against the CPU oracle's regulator restatement, on random
chromosome lengths, distance limits, resolutions (both normalisation branches), depths and thresholds.
    python scripts/fuzz_pipeline.py [n_cases]      (GPU box; the oracle needs ~10 s per 2000 x 2000 block)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle
from mustache_amd.pipeline import ChromosomePipeline
from mustache_amd.synth import synth_coo

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(__import__('os').environ.get('FUZZ_SEED', 77)))   # FUZZ_SEED=... draws another sweep
pipe = ChromosomePipeline([1.6, 3.2])
bad = total = 0
t0 = time.time()
for case in range(ncases):
    dpx = int(rng.integers(60, 420))
    n = int(rng.integers(max(2 * dpx, 500), 5200))
    res = int(rng.choice([1000, 2000, 5000, 10000, 25000]))
    depth = float(rng.choice([5.0, 40.0, 300.0]))
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.1, 0.3]))
    x, y, v = synth_coo(n, dpx, depth=depth, seed=int(rng.integers(0, 10 ** 6)), nloops=max(n // 25, 4))
    exp = oracle.regulator_coo(x, y, v.copy(), res, dpx, [1.6, 3.2], st, pt)
    got = sorted(pipe.run(x, y, v.copy(), res, dpx, st, pt), key=lambda r: (int(r[0]), int(r[1])))
    exp = sorted(exp, key=lambda r: (int(r[0]), int(r[1])))
    same = [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    qerr = max([abs(g[2] - e[2]) / max(e[2], 1e-300) for g, e in zip(got, exp)], default=0.0) if same else float("nan")
    total += len(exp)
    if not same or qerr > 1e-6:
        bad += 1
    print("case %2d n %5d dpx %3d res %5d depth %5.1f st %.2f pt %.2f branch %s loops %4d/%4d %s q-err %.1e  [%.0f s]"
          % (case, n, dpx, res, depth, st, pt, "A" if (n - dpx) * res > 2e6 else "B", len(got), len(exp),
             "ok" if same else "MISMATCH", qerr, time.time() - t0), flush=True)
print("done: %d cases, %d mismatches; loops compared: %d" % (ncases, bad, total))
