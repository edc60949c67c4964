"""Whole per-chromosome GPU pipeline against the CPU oracle at geometries beyond BASELINE's (GPU box; the oracle needs about
a minute per 8000 x 8000 block):  500 bp resolution (distance limit 4000 px -> blocks of 8000 x 8000, window 4000) and an odd
distance limit (dpx 3011 -> blocks of 6022).
    python scripts/large_geometry_check.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle
from mustache_amd.pipeline import ChromosomePipeline
from mustache_amd.synth import synth_coo

pipe = ChromosomePipeline([1.6, 3.2])
bad = total = 0
t0 = time.time()
for n, dpx, res, depth in ((9000, 4000, 500, 200.0), (7000, 3011, 1000, 60.0)):
    st, pt = 0.88, 0.1
    x, y, v = synth_coo(n, dpx, depth=depth, seed=n, nloops=n // 25)
    exp = oracle.regulator_coo(x, y, v.copy(), res, dpx, [1.6, 3.2], st, pt)
    got = sorted(pipe.run(x, y, v.copy(), res, dpx, st, pt), key=lambda r: (int(r[0]), int(r[1])))
    exp = sorted(exp, key=lambda r: (int(r[0]), int(r[1])))
    same = [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    qerr = max([abs(g[2] - e[2]) / max(e[2], 1e-300) for g, e in zip(got, exp)], default=0.0) if same else float("nan")
    total += len(exp)
    bad += (not same) or qerr > 1e-6
    print("n %5d dpx %4d res %5d depth %5.1f nnz %d loops %4d/%4d %s q-err %.1e  [%.0f s]"
          % (n, dpx, res, depth, len(v), len(got), len(exp), "ok" if same else "MISMATCH", qerr, time.time() - t0), flush=True)
print("done: %d mismatches; loops compared: %d" % (bad, total))
