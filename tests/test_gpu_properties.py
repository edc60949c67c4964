"""Size-independent properties at BASELINE's full block sizes, non-default scale-space parameters, determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _normalised_block(n, dpx, seed, res):
    """A dense near-diagonal block cut from a synthetic chromosome, normalised on the GPU."""
    import torch
    from mustache_amd.normalize import normalize_band
    from mustache_amd.pipeline import ChromosomePipeline
    from mustache_amd.synth import band_counts
    dev = "cuda"
    N = n + dpx
    band = band_counts(N, dpx, 300.0, max(N // 32, 1), seed, device=dev)
    band, _, _ = normalize_band(band, N, dpx, res)
    pipe = ChromosomePipeline([1.6, 3.2])
    c, nz, cnt = pipe.blocks_from_band(band, N, dpx, [dpx // 2], n)
    return pipe, c, nz, cnt


@pytest.mark.parametrize("n,dpx,res", [(4000, 2000, 1000), (2000, 400, 5000)])
def test_full_size_block_properties(n, dpx, res):
    """BASELINE shapes (4000^2 @ 1 kb, 2000^2 @ 5 kb): skipping empty tiles changes nothing; two runs are bit-identical
    (no float atomics); every found pixel is a tested pixel inside the band with a positive response and p in [0, 1]."""
    pipe, c, nz, cnt = _normalised_block(n, dpx, 5, res)
    eng = pipe.engine
    a, fa = eng.sigma_loop(c, nz, cnt, skip_empty=True)
    b, fb = eng.sigma_loop(c, nz, cnt, skip_empty=False)
    a2, fa2 = eng.sigma_loop(c, nz, cnt, skip_empty=True)
    ra, rb, ra2 = a[0], b[0], a2[0]
    for k in ("pixel", "level", "value", "pval"):
        assert np.array_equal(ra[k], rb[k]), "dense and band-skipping runs must agree exactly (%s)" % k
        assert np.array_equal(ra[k], ra2[k]), "run-to-run determinism (%s)" % k
    assert np.array_equal(fa[0][0], fb[0][0]) and np.array_equal(fa[0][1], fb[0][1])
    m = len(ra["pixel"])
    assert m > 1000
    assert np.all(np.diff(ra["pixel"].astype(np.int64)) > 0), "records sorted by pixel, no duplicates"
    x, y = ra["pixel"].astype(np.int64) // n, ra["pixel"].astype(np.int64) % n
    nzh = nz[0].cpu().numpy().astype(bool)
    assert nzh[x, y].all() and np.all(y - x >= 4) and np.all(y - x <= dpx + 1)
    assert np.all(ra["value"] > 0) and np.all((ra["level"] >= 1) & (ra["level"] <= 18))
    assert np.all((ra["pval"] >= 0) & (ra["pval"] <= 1))
    # a found pixel is a strict local maximum of its level among found neighbours of the same level? not required;
    # but no two 8-adjacent pixels can both be found at the SAME level with equal value unless tied
    loc, scale = fa[0]
    assert np.all(scale > 0) and np.all(loc >= 0)


@pytest.mark.parametrize("octaves,sz", [([1.6, 3.2, 6.4], None), ([2.0, 4.0], None), ([1.2], None)])
def test_non_default_octaves_vs_oracle(octaves, sz):
    """-sz / -oc variants: other radii (up to 28 uses the small-tile instantiation), 1 or 3 octaves, no level reuse."""
    import oracle
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    n, dpx = 384, 96
    x, y, v = synth_coo(n, dpx, depth=300.0, seed=9, nloops=30)
    oracle.normalize_sparse(x, y, v, 50000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    exp, mid = oracle.mustache_block(c.copy(), 0, dpx, octaves, 0.7, 0.3, return_intermediate=True)
    got = mustache(c, "1", "1", 5000, [], 0, n, 0, dpx, octaves, 0.7, 0.3)
    assert len(exp) > 0
    assert [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    np.testing.assert_allclose([q for _, _, q, _ in got], [q for _, _, q, _ in exp], rtol=1e-9)


def test_unsupported_radius_fails_loudly():
    from mustache_amd.mustache import mustache
    c = np.zeros((256, 256))
    with pytest.raises(ValueError, match="radius"):
        mustache(c, "1", "1", 5000, [], 0, 256, 0, 60, [1.6, 3.2, 6.4, 12.8], 0.8, 0.1)


def test_ragged_and_tiny_blocks():
    """Odd block edge (scalar prologue path, partial tiles), a block smaller than one tile, and an all-zero block."""
    import oracle
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    for n, dpx in ((333, 90), (61, 40)):
        x, y, v = synth_coo(n, dpx, depth=300.0, seed=3)
        oracle.normalize_sparse(x, y, v, 50000, dpx)
        c = np.zeros((n, n))
        c[x, y] = v
        exp = oracle.mustache_block(c.copy(), 7, dpx, [1.6, 3.2], 0.7, 0.3)
        got = mustache(c, "1", "1", 5000, [], 7, n + 7, 0, dpx, [1.6, 3.2], 0.7, 0.3)
        assert [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    assert mustache(np.zeros((128, 128)), "1", "1", 5000, [], 0, 128, 0, 40, [1.6, 3.2], 0.8, 0.1) == []


@pytest.mark.parametrize("depth,seed", [(6.0, 17), (1.5, 18)])
def test_sparse_block_vs_oracle(depth, seed):
    """Sparse contact maps (most far-diagonal pixels are exact zeros, like real 1 kb data): partially tested tiles, runs
    of identical DoG values, tiles with a handful of nz pixels -- found set and loops must still equal the oracle's."""
    import torch
    import oracle
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    n, dpx = 700, 300
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=40)
    oracle.normalize_sparse(x, y, v, 20000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    frac = (c != 0).sum() / (n * dpx)
    assert frac < 0.8
    ref = c.copy()
    nz = oracle.block_prologue(ref, dpx)
    ss = oracle.scale_space_levels(ref, nz, [1.6, 3.2])
    eng = ScaleSpaceEngine([1.6, 3.2])
    dev = torch.from_numpy(c.copy()).cuda().unsqueeze(0)
    nzd, cnt = eng.prologue(dev, dpx, True)
    found, fits = eng.sigma_loop(dev, nzd, cnt)
    f = ss.pval != 2
    assert np.array_equal(found[0]["pixel"].astype(np.int64), np.flatnonzero(nz.ravel())[f])
    assert np.array_equal(found[0]["value"], ss.best[f])
    assert np.array_equal(found[0]["level"].astype(np.int64), ss.level[f])
    np.testing.assert_allclose(found[0]["pval"], ss.pval[f], rtol=1e-9)
    exp = oracle.mustache_block(c.copy(), 0, dpx, [1.6, 3.2], 0.5, 0.3)
    got = mustache(c, "1", "1", 5000, [], 0, n, 0, dpx, [1.6, 3.2], 0.5, 0.3)
    assert [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]


def test_found_capacity_overflow_is_reported_and_recovered():
    """A too-small record capacity must surface as MST_E_OVERFLOW (never silently dropped records); the engine then
    re-runs with a larger buffer and gets the same records."""
    import torch
    from mustache_amd import _lib
    from mustache_amd.engine import ScaleSpaceEngine, _ptr, _stream
    import ctypes
    pipe, c, nz, cnt = _normalised_block(2000, 400, 7, 5000)
    eng = pipe.engine
    ref, _ = eng.sigma_loop(c, nz, cnt)
    m = len(ref[0]["pixel"])
    assert m > 5000
    got, _ = eng.sigma_loop(c, nz, cnt, found_cap=1024)          # overflows, engine grows x4 until it fits
    assert np.array_equal(got[0]["pixel"], ref[0]["pixel"]) and np.array_equal(got[0]["pval"], ref[0]["pval"])
    eng._found_cap.clear()
    # and the raw ABI reports it
    lv = ctypes.byref(eng._lv_struct)
    B, CH = 1, 2000
    cap = 512
    found = torch.empty((B, cap, 2), dtype=torch.int64, device="cuda")
    count = torch.empty(B, dtype=torch.int32, device="cuda")
    stats = torch.empty((B, 48, 2), dtype=torch.float64, device="cuda")
    fit = torch.empty((B, 48, 2), dtype=torch.float64, device="cuda")
    pval = torch.empty((B, cap), dtype=torch.float64, device="cuda")
    wsb = int(eng.lib.mst_scale_space_workspace_bytes(B, CH, lv))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    assert eng.lib.mst_scale_space(_ptr(c), _ptr(nz), B, CH, lv, _ptr(found), cap, _ptr(count), _ptr(stats), 1, _ptr(ws),
                                   wsb, _stream()) == 0
    rc = eng.lib.mst_found_pvalues(_ptr(found), cap, _ptr(count), _ptr(cnt), _ptr(stats), B, 18, _ptr(pval), _ptr(fit),
                                   _stream())
    assert rc == _lib.MST_E_OVERFLOW and b"capacity" in eng.lib.mst_last_error()
    assert int(count.cpu()[0]) == m, "the counter keeps counting past the capacity, so the caller knows the need"


def test_non_finite_input_is_an_error_not_garbage():
    """The reference raises ValueError inside expon.fit for NaN/inf DoG values; the library returns MST_E_NONFINITE."""
    from mustache_amd import _lib
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    import oracle
    n, dpx = 320, 80
    x, y, v = synth_coo(n, dpx, depth=300.0, seed=1)
    oracle.normalize_sparse(x, y, v, 50000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    c[100, 130] = np.nan
    with pytest.raises(_lib.MstError) as e:
        mustache(c, "1", "1", 5000, [], 0, n, 0, dpx, [1.6, 3.2], 0.8, 0.1)
    assert e.value.code == _lib.MST_E_NONFINITE


def test_bad_arguments_are_rejected():
    from mustache_amd import _lib
    lib = _lib.load()
    assert lib.mst_block_prologue(None, None, None, 1, 16, 4, 1, None) == _lib.MST_E_ARG
    assert b"mst_block_prologue" in lib.mst_last_error()
    assert lib.mst_scale_space_workspace_bytes(0, 100, None) == 0


def test_opt_in_fma_mode_within_north_star_tolerance():
    """MST_FLAG_FMA (off by default) fuses each tap's multiply-add.  It must keep the found SET identical on the reference
    fixture geometry, and DoG / p-values within the tolerance north_star states (1e-5 relative; observed ~1e-13)."""
    pipe, c, nz, cnt = _normalised_block(2000, 400, 5, 5000)
    eng = pipe.engine
    a, fa = eng.sigma_loop(c, nz, cnt, fma=False)
    b, fb = eng.sigma_loop(c, nz, cnt, fma=True)
    ra, rb = a[0], b[0]
    assert np.array_equal(ra["pixel"], rb["pixel"]) and np.array_equal(ra["level"], rb["level"])
    assert not np.array_equal(ra["value"], rb["value"]), "the relaxed mode really is a different rounding sequence"
    np.testing.assert_allclose(rb["value"], ra["value"], rtol=1e-9)
    np.testing.assert_allclose(rb["pval"], ra["pval"], rtol=1e-5, atol=1e-300)
    np.testing.assert_allclose(fb[0][1], fa[0][1], rtol=1e-9)


def test_exact_zero_pvalue_and_top_edge_candidates():
    """SURVEY 8c edge cases: (1) a response so strong that 1 - (-expm1(-z)) is exactly 0.0; (2) strong candidates within
    2*ceil(sigma) rows of the block's top edge, which the reference drops through a negative-start (empty) slice."""
    import torch
    import oracle
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    n, dpx = 360, 90
    x, y, v = synth_coo(n, dpx, depth=300.0, seed=23, nloops=20)
    oracle.normalize_sparse(x, y, v, 50000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    yy, xx = np.mgrid[0:n, 0:n]
    for (cy, cx, amp) in ((5, 45, 60.0), (2, 30, 40.0), (200, 250, 4000.0)):      # two blobs at the top edge, one enormous
        c += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.5 ** 2)) * (c != 0)
    ref = c.copy()
    nz = oracle.block_prologue(ref, dpx)
    ss = oracle.scale_space_levels(ref, nz, [1.6, 3.2])
    f = ss.pval != 2
    assert (ss.pval[f] == 0.0).sum() >= 1, "fixture must contain an exactly-zero p-value"
    eng = ScaleSpaceEngine([1.6, 3.2])
    dev = torch.from_numpy(c.copy()).cuda().unsqueeze(0)
    nzd, cnt = eng.prologue(dev, dpx, True)
    found, _ = eng.sigma_loop(dev, nzd, cnt)
    assert np.array_equal(found[0]["pixel"].astype(np.int64), np.flatnonzero(nz.ravel())[f])
    assert np.array_equal(found[0]["pval"] == 0.0, ss.pval[f] == 0.0), "exact zeros must coincide"
    np.testing.assert_allclose(found[0]["pval"], ss.pval[f], rtol=1e-9, atol=0)
    exp, mid = oracle.mustache_block(c.copy(), 0, dpx, [1.6, 3.2], 0.7, 0.2, return_intermediate=True)
    got = mustache(c, "1", "1", 5000, [], 0, n, 0, dpx, [1.6, 3.2], 0.7, 0.2)
    assert [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    assert not any(int(a) < 8 for a, _, _, _ in exp), "top-edge candidates are dropped by the reference's slice semantics"
    # the top-edge blobs were genuinely found and significant -- they are dropped by the filter, not missed
    q = oracle.benjamini_hochberg(ss.pval[f])
    rows = np.flatnonzero(nz.ravel())[f] // n
    assert ((rows < 8) & (q < 0.2)).any()


def test_large_distance_limit_blocks_of_10000():
    """-d 5 Mb at 1 kb: dpx = 5000 -> blocks of 10000 x 10000 (54k tiles per block, 80 KB of LDS for a diagonal mean).  Too big
    for the oracle in test time, so size-independent properties: the band-direct kernel and the dense-block path give the
    same loops, skipping empty tiles changes nothing, two runs are identical, every loop lies inside the distance limit."""
    import torch
    from mustache_amd.normalize import band_from_coo, normalize_band
    from mustache_amd.pipeline import ChromosomePipeline, block_tiling
    from mustache_amd.synth import synth_coo
    n, dpx, res = 12000, 5000, 1000
    x, y, v = synth_coo(n, dpx, depth=60.0, seed=71, nloops=600)
    pipe = ChromosomePipeline([1.6, 3.2])
    band = band_from_coo(*(torch.from_numpy(a).to(pipe.device) for a in (x, y, v)), n, dpx)
    band, _, _ = normalize_band(band, n, dpx, res)
    CH, start, end = block_tiling(n, dpx)
    assert CH == 10000 and len(start) == 2
    key = lambda r: (int(r[0]), int(r[1]), float(r[2]), float(r[3]))
    a = sorted(key(r) for r in pipe.run_band(band, n, dpx, 0.3, 0.3, distributed=False))
    b = sorted(key(r) for r in pipe.run_band(band, n, dpx, 0.3, 0.3, distributed=False, skip_empty=False))
    c = sorted(key(r) for r in pipe.run_band(band, n, dpx, 0.3, 0.3, distributed=False, dense=True))
    d = sorted(key(r) for r in pipe.run_band(band, n, dpx, 0.3, 0.3, distributed=False))
    assert a == b == d and len(a) > 50                      # with / without the tile list, and run to run: bit for bit
    # the dense-block route cuts its tiles on each block's own lattice, the band route on the chromosome's (shared between
    # overlapping blocks): the level sums are grouped differently, q may differ in its last bits
    assert [(r[0], r[1], r[3]) for r in a] == [(r[0], r[1], r[3]) for r in c]
    np.testing.assert_allclose([r[2] for r in a], [r[2] for r in c], rtol=1e-9)
    assert all(0 <= r[1] - r[0] <= dpx for r in a)
