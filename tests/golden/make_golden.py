#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (dev container only).

    python -B tests/golden/make_golden.py

The reference (ay-lab/mustache v1.3.3, read-only at /root/reference) has no tests or golden outputs of its own
(SURVEY.md section 4), so parity is pinned on outputs of the reference itself: this script imports it (see
_refimport.py for the three stand-ins the import needs), feeds it small deterministic synthetic inputs
(mustache_amd/synth.py) and stores inputs + outputs + selected intermediates as compressed .npz files.
Only data is stored -- no reference source text.  The fixtures travel to the GPU box; the reference does not.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _refimport import load_reference  # noqa: E402
from mustache_amd.synth import synth_coo  # noqa: E402

OCTAVES = [1.6, 3.2]


def subsample(a):
    f = np.ascontiguousarray(a).ravel()
    return f[::97].copy()


def run_mustache_traced(ref, c, start, dpx, st, pt, octaves=None):
    """Call ref.mustache and capture SciPy-call arguments/results and the function's final locals."""
    import scipy.stats
    cap = dict(gauss=[], maxf=[], fit=[], bh=[])
    g0, m0, fit0, bh0 = ref.gaussian_filter, ref.maximum_filter, ref.expon.fit, ref.multipletests

    def g(inp, sigma, **kw):
        out = g0(inp, sigma, **kw)
        cap["gauss"].append((float(sigma), float(kw["truncate"]), float(out.sum()), subsample(out)))
        return out

    def mf(inp, **kw):
        out = m0(inp, **kw)
        cap["maxf"].append((float(inp.sum()), subsample(inp), float(out.sum()), subsample(out)))
        return out

    def fit(data, *a, **kw):
        r = fit0(data, *a, **kw)
        cap["fit"].append((float(r[0]), float(r[1]), float(np.sum(data))))
        return r

    def bh(p, **kw):
        r = bh0(p, **kw)
        cap["bh"].append((np.array(p, copy=True), np.array(r[1], copy=True)))
        return r

    locs = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name != "mustache":
            return None

        def local(frame, event, arg):
            if event == "return":
                for k in ("nz", "pAll", "Scales", "vAll", "o", "so", "x", "y"):
                    if k in frame.f_locals:
                        locs[k] = np.array(frame.f_locals[k], copy=True)
            return local
        return local

    ref.gaussian_filter, ref.maximum_filter, ref.multipletests = g, mf, bh
    ref.expon.fit = fit
    sys.settrace(tracer)
    try:
        loops = ref.mustache(c, "1", "1", 5000, [], start, start + c.shape[0], 0, dpx, octaves or OCTAVES, st, pt)
    finally:
        sys.settrace(None)
        ref.gaussian_filter, ref.maximum_filter, ref.multipletests = g0, m0, bh0
        ref.expon.fit = fit0
    return loops, cap, locs


def loops_array(loops):
    if not loops:
        return np.zeros((0, 4))
    return np.array([[float(a), float(b), float(q), float(s)] for a, b, q, s in loops], dtype=np.float64)


def dense(x, y, v, n):
    c = np.zeros((n, n))
    c[x, y] = v
    return c


def make_normalize(ref):
    # branch A: window = int(2e6/res) = 40 bins; holes exercise cnt<30, an empty diagonal, a constant run
    n, dpx, res = 600, 80, 50000
    x, y, v = synth_coo(n, dpx, depth=120.0, seed=11)
    d = y - x
    keep = np.ones(len(v), bool)
    keep &= ~((x >= 200) & (x < 290) & (d % 3 != 0))        # sparse stretch -> window counts < 30
    keep &= d != 37                                           # an empty diagonal
    x, y, v = x[keep], y[keep], v[keep].copy()
    v[(d[keep] == 20) & (x >= 400) & (x < 480)] = 5.0        # constant run -> zero local variance
    perm = np.random.default_rng(5).permutation(len(v))       # entry order matters for np.mean/np.std
    x, y, v = x[perm], y[perm], v[perm]
    vin = v.copy()
    w = ref.normalize_sparse(x, y, v, res, dpx)
    np.savez_compressed(os.path.join(HERE, "normalize_A.npz"), x=x.astype(np.int32), y=y.astype(np.int32),
                        v_in=vin, v_out=v, weights=np.array(w), res=res, dpx=dpx)
    # branch B: (n - dpx) * res <= 2e6
    n, dpx, res = 300, 80, 5000
    x, y, v = synth_coo(n, dpx, depth=120.0, seed=12)
    perm = np.random.default_rng(6).permutation(len(v))
    x, y, v = x[perm], y[perm], v[perm]
    vin = v.copy()
    w = ref.normalize_sparse(x, y, v, res, dpx)
    np.savez_compressed(os.path.join(HERE, "normalize_B.npz"), x=x.astype(np.int32), y=y.astype(np.int32),
                        v_in=vin, v_out=v, weights=np.array(w), res=res, dpx=dpx)
    print("normalize fixtures done")


def make_normalize_real(ref):
    """Branch A of normalize_sparse at the REAL window sizes of BASELINE's configurations: 400 bins (5 kb) and 2000 bins (1 kb)
    -- normalize_A's window is 40.  Integer counts (what a contact map holds), a thinned stretch (window counts < 30 on the far
    diagonals) and an empty diagonal; the chromosomes are short and the bands narrow so that the reference's O(n * window)
    np.convolve calls take seconds and the fixtures stay ~1 MB: the window length is what is under test."""
    for name, n, dpx, res, seed in (("normalize_C.npz", 2600, 56, 5000, 13), ("normalize_D.npz", 5200, 28, 1000, 14)):
        x, y, v = synth_coo(n, dpx, depth=60.0, seed=seed)
        d = y - x
        v = np.round(v) + 1.0
        keep = np.ones(len(v), bool)
        keep &= ~((x >= n // 3) & (x < n // 3 + 3 * (2000000 // res) // 2) & ((x + 7 * d) % 23 != 0))   # ~4 % kept over 1.5 windows
        keep &= d != 17
        x, y, v = x[keep], y[keep], v[keep].copy()
        vin = v.copy()
        w = ref.normalize_sparse(x, y, v, res, dpx)
        assert (n - dpx) * res > 2000000 and len(w) > 0
        np.savez_compressed(os.path.join(HERE, name), x=x.astype(np.int32), y=y.astype(np.int32), v_in=vin.astype(np.uint16),
                            v_out=v, weights=np.array(w), res=res, dpx=dpx, window=int(2000000 / res))
        assert np.array_equal(vin, vin.astype(np.uint16))
        print(name, len(v), "records, window", int(2000000 / res))


def make_block(ref, name, n, dpx, seed, start, st, pt, depth=300.0, nloops=None, res=50000):
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=nloops)
    ref.normalize_sparse(x, y, v, res, dpx)
    c = dense(x, y, v, n)
    cin = c.copy()
    loops, cap, locs = run_mustache_traced(ref, c, start, dpx, st, pt)
    out = dict(x=x.astype(np.int32), y=y.astype(np.int32), v=v, n=n, dpx=dpx, start=start, st=st, pt=pt,
               loops=loops_array(loops), c_after_sum=float(c.sum()),
               c_changed=int((c != cin).sum()))
    out["g_sigma"] = np.array([g[0] for g in cap["gauss"]])
    out["g_trunc"] = np.array([g[1] for g in cap["gauss"]])
    out["g_sum"] = np.array([g[2] for g in cap["gauss"]])
    out["g_sub"] = np.array([g[3] for g in cap["gauss"]])
    out["d_sum"] = np.array([m[0] for m in cap["maxf"]])
    out["d_sub"] = np.array([m[1] for m in cap["maxf"]])
    out["m_sum"] = np.array([m[2] for m in cap["maxf"]])
    out["m_sub"] = np.array([m[3] for m in cap["maxf"]])
    out["fit"] = np.array(cap["fit"]).reshape(-1, 3)
    if cap["bh"]:
        out["bh_in"], out["bh_out"] = cap["bh"][0]
    for k in ("nz", "pAll", "Scales", "vAll"):
        if k in locs:
            out["loc_" + k] = locs[k] if k != "nz" else np.packbits(locs[k])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "nz", int(locs["nz"].sum()) if "nz" in locs else None, "found",
          int((locs["pAll"] != 2).sum()) if "pAll" in locs else None, "loops", len(loops),
          "pzero", int((locs["pAll"] == 0).sum()) if "pAll" in locs else None)


def make_big_block(ref, name="block_2000", n=2000, dpx=400, seed=3, start=3200, st=0.8, pt=0.1, octaves=None, full=False):
    """One block of BASELINE's 5 kb geometry (2000 x 2000, distance limit 400 px) -- or, as `block_4000`, of the headline
    1 kb geometry (4000 x 4000, distance limit 2000 px; the reference needs ~80 s and ~3 GB for it).  The input is the raw
    synthetic map (regenerated from the seed by the tests, so only its checksum is stored) and the outputs are kept
    compact: loops, the 18 expon fits, per-level Gaussian sums, and order-independent checksums of the found set.
    `octaves`: the list the reference's main() builds from -sz / -oc (mustache.py:874): sigma0 * 2^i, i < octaves."""
    x, y, v = synth_coo(n, dpx, depth=300.0, seed=seed)
    c = dense(x, y, v, n)
    loops, cap, locs = run_mustache_traced(ref, c, start, dpx, st, pt, octaves)
    nz = locs["nz"]
    found = locs["pAll"] != 2
    pix = np.flatnonzero(nz.ravel())[found].astype(np.int64)
    sig = locs["Scales"][found]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, dpx=dpx, seed=seed, depth=300.0, start=start, st=st,
                        pt=pt, in_nnz=len(v), in_checksum=float(v.sum()), nz_count=int(nz.sum()),
                        loops=loops_array(loops), fit=np.array(cap["fit"]).reshape(-1, 3),
                        octaves=np.array(octaves or OCTAVES, dtype=np.float64), g_sigma=np.array([g[0] for g in cap["gauss"]]),
                        g_sum=np.array([g[2] for g in cap["gauss"]]), found_count=int(found.sum()),
                        found_pixel_sum=int(pix.sum()), found_pixel_xor=int(np.bitwise_xor.reduce(pix)),
                        found_sigma_sum=float(np.sum(sig)), found_value_sum=float(np.sum(locs["vAll"][found])),
                        found_value_max=float(np.max(locs["vAll"][found])),
                        found_pvalue_sum=float(np.sum(locs["pAll"][found])),
                        found_pixels_head=pix[:4096].astype(np.int32), found_values_head=locs["vAll"][found][:4096])
    print(name, "nz", int(nz.sum()), "found", int(found.sum()), "loops", len(loops))
    if full:
        # the COMPLETE found set of the reference, pixel for pixel: row-major pixel index, the recorded scale as an index into
        # the sorted distinct Scales values (= the 1-based tested level of the GPU records minus one when all 18 occur), the
        # winning DoG value vAll (bit pattern) and the BH q-value the reference leaves in pAll (mustache.py:778-779)
        sig_values = np.unique(sig)
        np.savez_compressed(os.path.join(HERE, name + "_full.npz"), n=n, dpx=dpx, seed=seed, depth=300.0,
                            pixel=pix.astype(np.uint32), sigma_values=sig_values,
                            sigma_index=np.searchsorted(sig_values, sig).astype(np.uint8),
                            value=locs["vAll"][found], q=locs["pAll"][found])
        print(name + "_full", len(pix), "records;", len(sig_values), "distinct scales")


def make_edges(ref):
    # < 50 tested pixels -> [] (mustache.py:701); 50 <= nz < 10000 -> [] (mustache.py:775)
    n, dpx = 200, 60
    x, y, v = synth_coo(n, dpx, depth=200.0, seed=21)
    ref.normalize_sparse(x, y, v, 50000, dpx)
    few = (x < 8) & (y - x >= 4) & (y < 13)
    c1 = dense(x[few], y[few], v[few], n)
    l1 = ref.mustache(c1.copy(), "1", "1", 5000, [], 0, n, 0, dpx, OCTAVES, 0.8, 0.1)
    c2 = dense(x, y, v, n)
    l2 = ref.mustache(c2.copy(), "1", "1", 5000, [], 0, n, 0, dpx, OCTAVES, 0.8, 0.1)
    np.savez_compressed(os.path.join(HERE, "block_edges.npz"), x=x.astype(np.int32), y=y.astype(np.int32), v=v,
                        few=few, n=n, dpx=dpx, nz1=int(np.logical_and(c1 != 0, np.triu(c1, 4)).sum()),
                        nz2=int(np.logical_and(c2 != 0, np.triu(c2, 4)).sum()),
                        loops1=loops_array(l1), loops2=loops_array(l2))
    print("edges", len(l1), len(l2))


class _InlineProcess:
    """multiprocessing.Process stand-in that runs the target in-process (fixtures need determinism, not speed)."""

    def __init__(self, target, args):
        self._t, self._a = target, args

    def start(self):
        self._t(*self._a)

    def join(self):
        pass


class _PlainManager:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def list(self):
        return []


def make_tiling(ref):
    """(n, dpx) -> starts / ends / mask sizes as regulator + process_block compute them (mustache.py:896-910, :948-953)."""
    rec = []
    real = (ref.Process, ref.Manager, ref.mustache, ref.normalize_sparse, ref.read_pd)
    calls = []

    def fake_mustache(cc, ch, ch2, res, w, start, end, mask, dpx, octs, st, pt):
        calls.append((start, end, mask, cc.shape[0]))
        return []

    ref.Process, ref.Manager = _InlineProcess, _PlainManager
    ref.mustache = fake_mustache
    ref.normalize_sparse = lambda *a, **k: []
    cases = [(1500, 400), (2000, 400), (2001, 400), (3600, 400), (3601, 400), (9630, 400), (4001, 2000),
             (12000, 2000), (8000, 1000), (4200, 200), (2399, 400)]
    try:
        for n, dpx in cases:
            calls.clear()
            xx = np.array([0, n - 1])
            ref.read_pd = lambda *a, **k: (xx, xx.copy(), np.array([1.0, 1.0]))
            ref.regulator("dummy.txt", False, False, "out", res=5000, distance_filter=dpx * 5000,
                          chromosome="1", nprocesses=1)
            rec.append((n, dpx, [c[0] for c in calls], [c[1] for c in calls], [c[2] for c in calls], calls[0][3]))
    finally:
        ref.Process, ref.Manager, ref.mustache, ref.normalize_sparse, ref.read_pd = real
    np.savez_compressed(os.path.join(HERE, "tiling.npz"),
                        n=np.array([r[0] for r in rec]), dpx=np.array([r[1] for r in rec]),
                        chunk=np.array([r[5] for r in rec]),
                        starts=np.array([np.array(r[2]) for r in rec], dtype=object),
                        ends=np.array([np.array(r[3]) for r in rec], dtype=object),
                        masks=np.array([np.array(r[4]) for r in rec], dtype=object))
    print("tiling done", [(r[0], r[1], len(r[2])) for r in rec])


def make_regulator(ref, name="regulator_3blocks", n=4200, dpx=200, res=10000, seed=31, nloops=None):
    """End to end through the reference's text reader, normalisation, tiling and block loop (3 blocks).
    `regulator_1kb_4blocks` (n=9000, res=1000, dpx=2000): the HEADLINE geometry -- 4000^2 blocks at stride 2000 (half
    overlap), a right-aligned last block [5000, 9000) whose mask_size (3000) exceeds dpx (mustache.py:909-910, :948-953),
    and normalize_sparse at its real 2000-bin window (:628-669).  ~10 min in the reference."""
    x, y, v = synth_coo(n, dpx, depth=300.0, seed=seed, nloops=nloops)
    # raw integer-ish counts and a bias file, as a 3-column RAWobserved-style text + KRnorm-style vector
    rng = np.random.default_rng(9)
    bias = rng.uniform(0.5, 1.5, n)
    bias[rng.choice(n, 40, replace=False)] = np.nan
    bias[rng.choice(n, 40, replace=False)] = 0.1
    with tempfile.TemporaryDirectory() as td:
        fpath = os.path.join(td, "chrS.RAWobserved")
        bpath = os.path.join(td, "chrS.KRnorm")
        with open(fpath, "w") as f:
            for a, b, c in zip(x, y, v):
                f.write("%d\t%d\t%r\n" % (a * res, b * res, float(c)))
        with open(bpath, "w") as f:
            for b in bias:
                f.write("%r\n" % float(b) if not np.isnan(b) else "NaN\n")
        real = (ref.Process, ref.Manager)
        ref.Process, ref.Manager = _InlineProcess, _PlainManager
        try:
            rx, ry, rv = ref.read_pd(fpath, dpx * res, bpath, "S", res)
            loops = ref.regulator(fpath, False, False, "out", res=res, pt=0.1, st=0.8, distance_filter=dpx * res,
                                  bias=bpath, chromosome="S", nprocesses=1)
        finally:
            ref.Process, ref.Manager = real
    rx, ry, rv = np.asarray(rx), np.asarray(ry), np.asarray(rv)
    la = loops_array(loops)
    la = la[np.lexsort((la[:, 1], la[:, 0]))]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, dpx=dpx, res=res, seed=seed, depth=300.0,
                        nloops=-1 if nloops is None else nloops,
                        bias=bias, in_checksum=float(v.sum()), in_nnz=len(v),
                        read_nnz=len(rv), read_vsum=float(rv.sum()), read_xsum=int(rx.sum()), read_ysum=int(ry.sum()),
                        loops=la)
    print("regulator loops", len(la))


def make_diff(name="diff_320", n=320, dpx=80, start=640, res=50000, nloops=30, pt=0.3, pt2=0.3, compact=False,
              normalize=True):
    """Two-sample path: run the reference's diff_mustache() on one block pair and keep its locals.  `diff_320`: a small
    pair with full inputs and locals; `diff_2000` (compact=True): BASELINE config 5's 5 kb block geometry (2000 x 2000,
    distance limit 400 px), inputs regenerated from the seeds by the tests, outputs as checksums + the four loop lists."""
    ref = load_reference("mustache")
    dref = load_reference("diff_mustache")
    xa, ya, va = synth_coo(n, dpx, depth=300.0, seed=51, nloops=nloops)
    xb, yb, vb = synth_coo(n, dpx, depth=260.0, seed=52, nloops=nloops)
    sums = (float(va.sum()), float(vb.sum()), len(va), len(vb))
    if normalize:          # (the big fixture feeds the raw maps: window sums through np.convolve are BLAS-build dependent, so
        ref.normalize_sparse(xa, ya, va, res, dpx)      # a normalised input could not be regenerated bit for bit elsewhere)
        ref.normalize_sparse(xb, yb, vb, res, dpx)
    c1, c2 = dense(xa, ya, va, n), dense(xb, yb, vb, n)
    fits = []
    fit0 = dref.norm.fit

    def nfit(data, *a, **k):
        r = fit0(data, *a, **k)
        fits.append((float(r[0]), float(r[1])))
        return r

    locs = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name != "diff_mustache":
            return None

        def local(frame, event, arg):
            if event == "return":
                for k in ("nz1", "nz2", "pAll1", "pAll2", "pPair1", "pPair2", "vAll1", "vAll2", "Scales1", "Scales2"):
                    if k in frame.f_locals:
                        locs[k] = np.array(frame.f_locals[k], copy=True)
            return local
        return local

    dref.norm.fit = nfit
    sys.settrace(tracer)
    try:
        out = dref.diff_mustache(c1.copy(), c2.copy(), "1", "1", 5000, start, start + n, 0, dpx, OCTAVES, 0.8, pt, pt2)
    finally:
        sys.settrace(None)
        dref.norm.fit = fit0
    lists = dict(loops1=loops_array(out[0]), diff1=loops_array(out[1]), loops2=loops_array(out[2]),
                 diff2=loops_array(out[3]))
    if compact:
        extra = {}
        for s_ in ("1", "2"):
            f = locs["pAll" + s_] != 2
            pix = np.flatnonzero(locs["nz" + s_].ravel())[f].astype(np.int64)
            extra.update({"found_count" + s_: int(f.sum()), "found_pixel_sum" + s_: int(pix.sum()),
                          "found_pixel_xor" + s_: int(np.bitwise_xor.reduce(pix)),
                          "found_value_sum" + s_: float(np.sum(locs["vAll" + s_][f])),
                          "found_pvalue_sum" + s_: float(np.sum(locs["pAll" + s_][f])),
                          "found_pair_sum" + s_: float(np.sum(locs["pPair" + s_][f])),
                          "found_sigma_sum" + s_: float(np.sum(locs["Scales" + s_][f])),
                          "nz_count" + s_: int(locs["nz" + s_].sum())})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, dpx=dpx, start=start, res=res, nloops=nloops,
                            st=0.8, pt=pt, pt2=pt2, in_sums=np.array(sums), norm_fit=np.array(fits), **lists, **extra)
    else:
        locs.pop("nz1", None), locs.pop("nz2", None)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), xa=xa.astype(np.int32), ya=ya.astype(np.int32), va=va,
                            xb=xb.astype(np.int32), yb=yb.astype(np.int32), vb=vb, n=n, dpx=dpx, start=start,
                            st=0.8, pt=pt, pt2=pt2, norm_fit=np.array(fits), **lists,
                            **{"loc_" + k: v for k, v in locs.items()})
    print(name, [len(o) for o in out], {k: v.shape for k, v in locs.items()})


def make_diff_regulator(name="diff_regulator_5kb_3blocks", n=5200, dpx=400, res=5000, pt=0.1, pt2=0.2, st=0.8):
    """The reference's own TWO-SAMPLE regulator() (diff_mustache.py:572-716) end to end at BASELINE config 5's geometry: two
    text files at 5 kb, distance limit 400 bins -> both samples normalised (window 400), three blocks of 2000 x 2000 at stride
    1600 ([0, 2000), [1600, 3600), right-aligned [3200, 5200)), diff_mustache() per block pair, the overlap masks, the four
    tags.  ~4 min in the reference.  Stored: the generator's parameters + the sorted tagged rows [x, y, fdr, sigma, tag]."""
    load_reference("mustache")
    dref = load_reference("diff_mustache")
    xa, ya, va = synth_coo(n, dpx, depth=300.0, seed=71)
    xb, yb, vb = synth_coo(n, dpx, depth=260.0, seed=72)
    sums = (float(va.sum()), float(vb.sum()), len(va), len(vb))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for tag, (x, y, v) in (("a", (xa, ya, va)), ("b", (xb, yb, vb))):
            fp = os.path.join(td, "s%s.txt" % tag)
            with open(fp, "w") as f:
                for a, b, c in zip(x, y, v):
                    f.write("%d\t%d\t%r\n" % (a * res, b * res, float(c)))
            paths.append(fp)
        real = (dref.Process, dref.Manager)
        dref.Process, dref.Manager = _InlineProcess, _PlainManager
        try:
            rows = dref.regulator(paths[0], paths[1], False, False, "out", res=res, pt=pt, pt2=pt2, st=st,
                                  distance_filter=dpx * res, chromosome="S", nprocesses=1)
        finally:
            dref.Process, dref.Manager = real
    out = np.array(sorted([float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4])] for r in rows)).reshape(-1, 5)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, dpx=dpx, res=res, pt=pt, pt2=pt2, st=st, seeds=np.array([71, 72]),
                        depths=np.array([300.0, 260.0]), in_sums=np.array(sums), rows=out)
    print("diff regulator rows", len(out), "per tag", [int((out[:, 4] == t).sum()) for t in (1, 2, 3, 4)])


def make_krnorm(ref):
    """The one real data file the reference bundles, data/chr21_5kb.KRnorm (3 columns: chr21, position, KR bias of HMEC chr21
    at 5 kb; 9630 rows), through the reference's read_bias (mustache.py:218-251, the 3-column branch with `is_chr` and
    position // res).  Fixture = the file's numeric content (positions, values) + what read_bias returned for it."""
    path = "/root/reference/data/chr21_5kb.KRnorm"
    pos, val, names = [], [], set()
    for line in open(path):
        a, b, c = line.strip().split("\t")
        names.add(a)
        pos.append(int(b))
        val.append(float(c))
    d = ref.read_bias(path, "21", 5000)
    keys = np.array(sorted(d), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "krnorm_chr21_5kb.npz"), chrom=np.array(sorted(names)),
                        pos=np.array(pos, dtype=np.int64), value=np.array(val), res=5000,
                        bias_keys=keys, bias_values=np.array([d[k] for k in keys]),
                        other_chrom_entries=len(ref.read_bias(path, "20", 5000)))
    v = np.array(val)
    print("krnorm rows", len(v), "finite", int(np.isfinite(v).sum()), "nan", int(np.isnan(v).sum()),
          "finite<0.2", int((v[np.isfinite(v)] < 0.2).sum()), "dict", len(keys))


def hic_header_case(path):
    """The version-8 `.hic` file the header fixture is made from (tests/hic_writer.py, deterministic); shared with the test."""
    sys.path.insert(0, os.path.dirname(HERE))
    from hic_writer import write_hic
    rng = np.random.default_rng(77)
    chroms = [("All", 5000), ("chr1", 1234567), ("2", 987654), ("chrX", 150000), ("MT", 16571)]
    mats = {}
    for ci in (1, 2, 3):
        per = {}
        for res in (5000, 25000):
            nb = chroms[ci][1] // res + 1
            x = rng.integers(0, nb, 300)
            y = np.minimum(nb - 1, x + rng.integers(0, 40, 300))
            per[res] = (x, y, rng.integers(1, 50, 300).astype(float))
        mats[ci] = per
    norms = {("KR", ci, res): rng.uniform(0.5, 1.5, chroms[ci][1] // res + 1) for ci in (1, 2, 3) for res in (5000, 25000)}
    write_hic(path, chroms, mats, norms=norms, version=8)


def make_hic_header():
    """Pin the native `.hic` reader's HEADER parse on code the reference itself holds: diff_mustache.py:182-249
    (readcstr / read_header, carried over from hic2cool, never called by the reference's own main).  It is imported here
    and run on a version-8 file written by tests/hic_writer.py; what it returns is the fixture."""
    import hashlib
    dref = load_reference("diff_mustache")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "hdr.hic")
        hic_header_case(path)
        blob = open(path, "rb").read()
        with open(path, "rb") as fh:
            chrs, resolutions, masterindex, genome, metadata = dref.read_header(fh)
    idx = sorted(chrs)
    np.savez_compressed(os.path.join(HERE, "hic_header_v8.npz"), sha256=hashlib.sha256(blob).hexdigest(),
                        file_bytes=len(blob), version=int(dref.version), masterindex=int(masterindex), genome=genome,
                        chr_index=np.array(idx), chr_name=np.array([chrs[i][1] for i in idx]),
                        chr_length=np.array([int(chrs[i][2]) for i in idx], dtype=np.int64),
                        resolutions=np.array(resolutions, dtype=np.int64),
                        meta_keys=np.array(sorted(metadata)), meta_values=np.array([metadata[k] for k in sorted(metadata)]))
    print("hic header", dref.version, masterindex, genome, chrs, resolutions, metadata)



def make_readers_ref():
    """readers_ref.npz: what the REFERENCE's own read_cooler / read_mcooler / read_hic_file (mustache.py:300-592) return for
    the three-chromosome container of tests/readers_case.py, served through the stand-in `cooler` module and a list-backed
    `hicstraw` (both absent offline; the reference only ever sees the module objects named `cooler` / `hicstraw`), plus the
    chromosome lists main() builds for a whole-genome run (mustache.py:1019-1033, captured by running main() with
    regulator() replaced by a recorder).  Pins the window walk, the set-difference de-duplication between consecutive
    windows, the NaN / zero / negative-value handling and the distance filter on the reference itself."""
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(HERE))
    import readers_case as rc
    ref = load_reference("mustache")

    class Numpy1(object):
        """The reference calls np.nan_to_num(<scipy sparse matrix>, copy=False, ...) (mustache.py:428, :526).  Under NumPy 1.x,
        which the reference was written for, that wraps the matrix in a 0-d object array and returns it untouched (object
        is not an inexact type) -- the NaNs are dealt with later (`val[np.isnan(val)] = 0`, `val > 0`); NumPy 2 raises
        instead ("Unable to avoid copy").  The reference module therefore sees NumPy through this proxy, which restores the
        1.x behaviour of that one call and forwards everything else."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def nan_to_num(x, *a, **kw):
            from scipy import sparse as sp
            return x if sp.issparse(x) else np.nan_to_num(x, *a, **kw)
    ref.np = Numpy1()
    out = {}

    def put(key, x, y, v):
        x, y, v = rc.as_sorted(x, y, v)
        out[key + "_x"], out[key + "_y"], out[key + "_v"] = x.astype(np.int32), y.astype(np.int32), v
    with tempfile.TemporaryDirectory() as td:
        cool, mcool = rc.write_cool_files(td)
        ref.cooler = rc.standin_cooler()                # tests/standins/cooler.py
        fake = rc.fake_hicstraw(rc.RES)
        ref.hicstraw = fake
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            for name, size in rc.CHROMS:
                x, y, v, res = ref.read_cooler(cool, rc.DIST, name, name, False)
                assert res == rc.RES
                put("cool_" + name, x, y, v)
                x, y, v = ref.read_mcooler(mcool, rc.DIST, name, name, 2 * rc.RES, False)
                put("mcool_" + name, x, y, v)
                x, y, v = ref.read_hic_file("case.hic", False, False, rc.DIST, name, name, rc.RES)
                put("hic_" + name, x, y, v)
            # the whole-genome GPU test's container (tests/test_gpu_pipeline.py): record count and digest per chromosome
            gcool = os.path.join(td, "genome.cool")
            ref.cooler.write_cool(gcool, rc.GENOME_RES, rc.genome_container())
            for name, _, _ in rc.GENOME:
                x, y, v, res = ref.read_cooler(gcool, rc.GENOME_DPX * rc.GENOME_RES, name, name, False)
                out["genome_%s_count" % name] = np.array(len(x))
                out["genome_%s_sha256" % name] = np.array(rc.digest(x, y, v))
            # chromosome size handed in by the caller (-cz / the whole-genome loop) and an explicit normalisation name
            del fake.calls[:]
            x, y, v = ref.read_hic_file("case.hic", "VC", rc.CHROMS[0][1], rc.DIST, "chrA", "chrA", rc.RES)
            put("hic_chrA_sized_VC", x, y, v)
            out["hic_calls_chrA_sized_VC"] = np.array([[c[2], c[3]] for c in fake.calls], dtype=np.int64)
            out["hic_norm_default"] = np.array("KR")
            # whole-genome chromosome lists of main(): regulator() replaced by a recorder that reports no loops
            seen = []

            def recorder(f, norm_method, CHRM_SIZE, outdir, **kw):
                seen.append((str(kw["chromosome"]), str(kw["chromosome2"]), int(CHRM_SIZE) if CHRM_SIZE else 0,
                             int(kw["distance_filter"]), int(kw["res"])))
                return []
            ref.regulator = recorder
            hicp = os.path.join(td, "case.hic")
            open(hicp, "wb").write(b"")                  # main() only checks that the path exists
            argv0 = sys.argv
            try:
                for key, path, r in (("cool", cool, "5kb"), ("mcool", mcool, "10kb"), ("hic", hicp, "5kb")):
                    del seen[:]
                    sys.argv = ["mustache", "-f", path, "-r", r, "-o", os.path.join(td, "o.tsv")]
                    ref.main()
                    out["main_%s_chroms" % key] = np.array([s[0] for s in seen])
                    out["main_%s_sizes" % key] = np.array([s[2] for s in seen], dtype=np.int64)
                    out["main_%s_dist_res" % key] = np.array([seen[0][3], seen[0][4]], dtype=np.int64)
            finally:
                sys.argv = argv0
    np.savez_compressed(os.path.join(HERE, "readers_ref.npz"), **out)
    print("readers_ref", {k: (v.shape if v.ndim else str(v)) for k, v in out.items()})


if __name__ == "__main__":
    if sys.argv[1:] == ["hic"]:
        make_hic_header()
        sys.exit(0)
    if sys.argv[1:] == ["readers"]:
        make_readers_ref()
        sys.exit(0)
    if sys.argv[1:] == ["diff"]:
        make_diff()
        sys.exit(0)
    if sys.argv[1:] == ["diffregulator"]:   # the two-sample regulator() at config 5's geometry (~4 min in the reference)
        make_diff_regulator()
        sys.exit(0)
    if sys.argv[1:] == ["diff2000"]:      # BASELINE config 5's block geometry (~1 min in the reference)
        make_diff("diff_2000", n=2000, dpx=400, start=3200, res=5000, nloops=None, pt=0.1, pt2=0.2, compact=True,
                  normalize=False)
        sys.exit(0)
    ref = load_reference("mustache")
    which = sys.argv[1:] or ["norm", "blocks", "big", "edges", "tiling", "regulator", "krnorm"]
    if "norm" in which:
        make_normalize(ref)
    if "normreal" in which:         # real window sizes (400 / 2000 bins); ~1 min in the reference
        make_normalize_real(ref)
    if "blocks" in which:
        make_block(ref, "block_320", 320, 80, seed=1, start=1600, st=0.8, pt=0.2, nloops=30)
        make_block(ref, "block_512", 512, 128, seed=2, start=0, st=0.7, pt=0.2, depth=200.0, nloops=40)
    if "big" in which:
        make_big_block(ref)
    if "big4000" in which:          # BASELINE config 4's block geometry (not in the default list: ~2 min, ~3 GB)
        make_big_block(ref, "block_4000", n=4000, dpx=2000, seed=4, start=8000, st=0.8, pt=0.1, full=True)
    if "octaves" in which:          # -sz / -oc variants (not in the default list): 3 octaves, and sigma0 = 2.0
        make_big_block(ref, "block_700_oc3", n=700, dpx=160, seed=21, start=1400, st=0.7, pt=0.2, octaves=[1.6, 3.2, 6.4])
        make_big_block(ref, "block_640_sz2", n=640, dpx=150, seed=22, start=0, st=0.7, pt=0.2, octaves=[2.0, 4.0])
    if "edges" in which:
        make_edges(ref)
    if "tiling" in which:
        make_tiling(ref)
    if "regulator" in which:
        make_regulator(ref)
    if "regulator1kb" in which:     # headline geometry (not in the default list: ~10 min, ~2 GB)
        make_regulator(ref, "regulator_1kb_4blocks", n=9000, dpx=2000, res=1000, seed=32, nloops=400)
    if "krnorm" in which:
        make_krnorm(ref)
