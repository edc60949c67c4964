"""Randomised sweep of the two-sample path: the band-direct route (both samples' sigma loops from their bands, difference image
and its blurs inside mst_diff_dog_band, pair p-values, tail) and the dense route (diff_mustache() on dense blocks) against the
CPU oracle's restatement of the reference's diff_mustache(), on random block sizes, distance limits, depths and thresholds.
    python scripts/fuzz_diff.py [n_cases]        (GPU box; the oracle needs a few seconds per case)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from mustache_amd.diff_mustache import _pair_tail, diff_mustache
from mustache_amd.engine import ScaleSpaceEngine
from mustache_amd.normalize import band_from_coo
from mustache_amd.synth import synth_coo

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(int(__import__('os').environ.get('FUZZ_SEED', 4242)))   # FUZZ_SEED=... draws another sweep
eng = ScaleSpaceEngine([1.6, 3.2])
OCT = [1.6, 3.2]


def key(lists):
    return [[(int(a), int(b), float(s)) for a, b, _, s in l] for l in lists]


bad = total = 0
for case in range(ncases):
    n = int(rng.integers(260, 900))
    dpx = int(rng.integers(60, max(61, min(n - 20, 300))))
    start = int(rng.integers(0, 5000))
    st, pt, pt2 = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.1, 0.3])), float(rng.choice([0.1, 0.3]))
    cs = []
    for s in range(2):
        x, y, v = synth_coo(n, dpx, depth=float(rng.choice([40.0, 150.0, 300.0])), seed=int(rng.integers(0, 10 ** 6)),
                            nloops=max(n // 15, 2))
        oracle.normalize_sparse(x, y, v, 50000, dpx)
        c = np.zeros((n, n))
        c[x, y] = v
        cs.append(c)
    exp = oracle.diff_block(cs[0].copy(), cs[1].copy(), start, dpx, OCT, st, pt, pt2)
    bands = []
    for c in cs:
        xx, yy = np.nonzero(np.triu(c))
        bands.append(band_from_coo(torch.from_numpy(xx).cuda(), torch.from_numpy(yy).cuda(), torch.from_numpy(c[xx, yy]).cuda(),
                                   n, dpx))
    batch = eng.run_band_pairs(bands, n, dpx, [0], n, select_below=pt)         # the driver's form (selected records only)
    got_band = _pair_tail(batch, 0, 1, start, pt, pt2, st, True)
    full = _pair_tail(eng.run_band_pairs(bands, n, dpx, [0], n), 0, 1, start, pt, pt2, st, True)   # whole found sets
    if [[tuple(l) for l in ls] for ls in full] != [[tuple(l) for l in ls] for ls in got_band]:
        print("case %d: selected-only and full-download forms differ" % case, flush=True)
        bad += 1
    got_dense = diff_mustache(cs[0].copy(), cs[1].copy(), "1", "1", 5000, start, start + n, 0, dpx, OCT, st, pt, pt2)
    ok = key(got_band) == key(exp) and key(got_dense) == key(exp)
    qe = 0.0
    if ok:
        for g, e in zip(got_band, exp):
            for a, b in zip(g, e):
                qe = max(qe, abs(a[2] - b[2]) / max(b[2], 1e-300))
    nl = sum(len(l) for l in exp)
    total += nl
    if not ok or qe > 1e-6:
        bad += 1
    print("case %2d n %3d dpx %3d st %.2f pt %.1f pt2 %.1f lists %s %s q-err %.1e" % (
        case, n, dpx, st, pt, pt2, [len(l) for l in exp], "ok" if ok else "MISMATCH", qe), flush=True)
print("done: %d cases, %d mismatches; loops compared: %d" % (ncases, bad, total))
