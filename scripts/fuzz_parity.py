"""Randomised parity sweep: HIP found set / values / levels vs the CPU oracle on many block shapes, densities and seeds.
    python scripts/fuzz_parity.py [n_cases]        (GPU box; ~1 s per case)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from mustache_amd.engine import ScaleSpaceEngine
from mustache_amd.synth import synth_coo

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(__import__('os').environ.get('FUZZ_SEED', 2024)))   # FUZZ_SEED=... draws another sweep
eng = ScaleSpaceEngine([1.6, 3.2])
bad = 0
for case in range(ncases):
    n = int(rng.integers(90, 760))
    dpx = int(rng.integers(30, max(31, min(n - 10, 420))))
    depth = float(rng.choice([0.8, 2.0, 8.0, 40.0, 300.0]))
    seed = int(rng.integers(0, 10 ** 6))
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=max(n // 20, 1))
    if len(v) < 100:
        continue
    oracle.normalize_sparse(x, y, v, 50000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    if rng.random() < 0.3:                      # punch rectangular holes (unmappable regions)
        a, b = sorted(rng.integers(0, n, 2))
        c[a:b, :] = 0
        c[:, a:b] = 0
    ref = c.copy()
    nz = oracle.block_prologue(ref, dpx)
    if nz.sum() < 50:
        continue
    ss = oracle.scale_space_levels(ref, nz, [1.6, 3.2], blur="scipy")
    f = ss.pval != 2
    dev = torch.from_numpy(c.copy()).cuda().unsqueeze(0)
    nzd, cnt = eng.prologue(dev, dpx, True)
    for skip in (True, False):
        found, fits = eng.sigma_loop(dev, nzd, cnt, skip_empty=skip)
        r = found[0]
        ok = (np.array_equal(r["pixel"].astype(np.int64), np.flatnonzero(nz.ravel())[f]) and
              np.array_equal(r["value"], ss.best[f]) and np.array_equal(r["level"].astype(np.int64), ss.level[f]) and
              np.allclose(r["pval"], ss.pval[f], rtol=1e-9, atol=0, equal_nan=True) and
              np.array_equal(fits[0][0], np.array([t["loc"] for t in ss.tested])))
        if not ok:
            bad += 1
            print("MISMATCH case", case, dict(n=n, dpx=dpx, depth=depth, seed=seed, skip=skip, nz=int(nz.sum()), found=int(f.sum()),
                                              got=len(r["pixel"])))
    # whole block through the drop-in (BH, filters, clustering) vs the oracle's tail
    from mustache_amd.mustache import mustache
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.2, 0.5]))
    exp = oracle.mustache_block(c.copy(), 17, dpx, [1.6, 3.2], st, pt)
    got = mustache(c.copy(), "1", "1", 5000, [], 17, n + 17, 0, dpx, [1.6, 3.2], st, pt)
    if [(int(a), int(b), s_) for a, b, _, s_ in got] != [(int(a), int(b), s_) for a, b, _, s_ in exp] or \
            not np.allclose([q for _, _, q, _ in got], [q for _, _, q, _ in exp], rtol=1e-9):
        bad += 1
        print("TAIL MISMATCH case", case, dict(n=n, dpx=dpx, depth=depth, seed=seed, st=st, pt=pt, exp=len(exp), got=len(got)))
    nloops_total = globals().get("nloops_total", 0) + len(exp)
    if case % 10 == 0:
        print("case", case, "n", n, "dpx", dpx, "depth", depth, "nz", int(nz.sum()), "found", int(f.sum()), flush=True)
print("done:", ncases, "cases,", bad, "mismatches; loops compared:", globals().get("nloops_total", 0))
