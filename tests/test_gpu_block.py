"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the golden
fixtures produced by the reference.  Integer / index results must be identical; DoG-derived floats are bit-exact
by construction (same operation order, no FMA); p-values are compared at 1e-9 relative (north_star asks 1e-5)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OCT = [1.6, 3.2]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=True)


def _dense(g):
    n = int(g["n"])
    c = np.zeros((n, n))
    c[g["x"], g["y"]] = g["v"]
    return c


@pytest.fixture(scope="module")
def eng():
    from mustache_amd.engine import ScaleSpaceEngine
    return ScaleSpaceEngine(OCT)


def test_library_loaded_is_in_tree():
    from mustache_amd import _lib
    lib = _lib.load()
    assert lib.mst_abi_version() == _lib.MST_ABI_VERSION
    assert os.path.dirname(_lib.LIB_PATH).endswith("mustache_amd")


@pytest.mark.parametrize("shape", [(1, 300, 517), (3, 64, 40), (1, 9, 2000)])
def test_gauss_blur_bit_exact(eng, shape):
    import torch
    import oracle
    rng = np.random.default_rng(3)
    img = rng.normal(size=shape) * 3 + 1
    dev = torch.from_numpy(img).cuda()
    for lv in oracle.level_table(OCT + [6.4])[::5]:
        if lv["radius"] > 32:
            continue
        out = eng.gauss_blur(dev, lv["weights"][lv["radius"]:]).cpu().numpy()
        for b in range(shape[0]):
            exp = oracle.blur_scipy(img[b], lv["sigma"], lv["truncate"])
            assert np.array_equal(out[b], exp), "sigma=%r radius=%d" % (lv["sigma"], lv["radius"])


def _run_block(eng, c, dpx, skip_empty=True):
    import torch
    dev = torch.from_numpy(c).cuda().unsqueeze(0)
    nz, nzc = eng.prologue(dev, dpx, True)
    found, fits = eng.sigma_loop(dev, nz, nzc, skip_empty=skip_empty)
    return dev, nz, int(nzc.cpu().numpy()[0]), found[0], fits[0]


@pytest.mark.parametrize("name", ["block_320.npz", "block_512.npz"])
@pytest.mark.parametrize("skip_empty", [True, False])
def test_sigma_loop_vs_reference_fixture(eng, golden_dir, name, skip_empty):
    g = _load(golden_dir, name)
    c = _dense(g)
    n, dpx = int(g["n"]), int(g["dpx"])
    dev, nz_d, nzc, found, fit = _run_block(eng, c.copy(), dpx, skip_empty)
    nz = np.unpackbits(g["loc_nz"]).astype(bool)[:n * n].reshape(n, n)
    assert np.array_equal(nz_d[0].cpu().numpy().astype(bool), nz)
    assert nzc == int(nz.sum())
    assert float(dev[0].sum().cpu()) == float(g["c_after_sum"]) or np.isclose(float(dev[0].sum().cpu()), float(g["c_after_sum"]), rtol=1e-13)
    # found set, best DoG value and recorded scale: bit-exact against the reference's locals
    pall, vall, scl = g["loc_pAll"], g["loc_vAll"], g["loc_Scales"]
    nz_idx = np.flatnonzero(nz.ravel())
    ref_found = pall != 2
    assert np.array_equal(found["pixel"].astype(np.int64), nz_idx[ref_found])
    assert np.array_equal(found["value"], vall[ref_found])
    sig = np.asarray(eng.levels.tested_sigma)
    assert np.array_equal(sig[found["level"].astype(int) - 1], scl[ref_found])
    # expon.fit: loc is a min (exact); scale is a mean (different but deterministic summation order)
    assert np.array_equal(fit[0], g["fit"][:, 0])
    np.testing.assert_allclose(fit[1], g["fit"][:, 1], rtol=1e-12)
    # p-values (before BH) against the reference's multipletests input
    np.testing.assert_allclose(found["pval"], g["bh_in"], rtol=1e-9, atol=0)


@pytest.mark.parametrize("name", ["block_320.npz", "block_512.npz"])
def test_mustache_dropin_vs_reference_fixture(golden_dir, name):
    from mustache_amd.mustache import mustache
    g = _load(golden_dir, name)
    c = _dense(g)
    n, dpx, start = int(g["n"]), int(g["dpx"]), int(g["start"])
    loops = mustache(c, "1", "1", 5000, [], start, start + n, 0, dpx, OCT, float(g["st"]), float(g["pt"]))
    assert c.sum() == float(g["c_after_sum"]), "block must be mutated in place like the reference does"
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    ref = torch.from_numpy(_dense(g)[None].copy()).cuda()
    ScaleSpaceEngine(OCT).prologue(ref, dpx, True)
    assert np.array_equal(ref[0].cpu().numpy(), c), "host-side fills == the device copy's fills (mst_block_prologue)"
    exp = g["loops"]
    assert len(loops) == len(exp) > 0
    got = np.array([[float(a), float(b), q, s] for a, b, q, s in loops])
    assert np.array_equal(got[:, :2], exp[:, :2]), "loop coordinates must be identical, in the reference's order"
    assert np.array_equal(got[:, 3], exp[:, 3])
    np.testing.assert_allclose(got[:, 2], exp[:, 2], rtol=1e-9)
    assert isinstance(loops[0][0], np.int64) and isinstance(loops[0][2], np.float64)


def test_edge_blocks(golden_dir):
    from mustache_amd.mustache import mustache
    g = _load(golden_dir, "block_edges.npz")
    n, dpx = int(g["n"]), int(g["dpx"])
    few = g["few"]
    c1 = np.zeros((n, n))
    c1[g["x"][few], g["y"][few]] = g["v"][few]
    before = c1.copy()
    assert mustache(c1, "1", "1", 5000, [], 0, n, 0, dpx, OCT, 0.8, 0.1) == []
    assert np.array_equal(c1, before)      # < 50 tested pixels: the reference returns before it fills c (mustache.py:701-703)
    c2 = np.zeros((n, n))
    c2[g["x"], g["y"]] = g["v"]
    assert mustache(c2, "1", "1", 5000, [], 0, n, 0, dpx, OCT, 0.8, 0.1) == []
    assert c2[0, 0] == 2.0 and c2[0, n - 1] == 2.0     # >= 50 tested pixels: filled in place (:703-706) before the :775 exit
    c3 = np.zeros((n, n))          # empty block
    assert mustache(c3, "1", "1", 5000, [], 0, n, 0, dpx, OCT, 0.8, 0.1) == []


@pytest.mark.parametrize("n,dpx,seed", [(2000, 400, 0)])
def test_full_size_block_vs_oracle(eng, n, dpx, seed):
    """BASELINE config 2: one dense 2000x2000 near-diagonal block at the 5 kb shape, HIP vs the oracle end to end."""
    import oracle
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    x, y, v = synth_coo(n + 1600, dpx, depth=300.0, seed=seed)
    oracle.normalize_sparse(x, y, v, 5000, dpx)
    sel = (x >= 1600) & (y >= 1600)
    c = np.zeros((n, n))
    c[x[sel] - 1600, y[sel] - 1600] = v[sel]
    exp, mid = oracle.mustache_block(c.copy(), 1600, dpx, OCT, 0.8, 0.1, return_intermediate=True)
    got = mustache(c, "1", "1", 5000, [], 1600, 1600 + n, 0, dpx, OCT, 0.8, 0.1)
    assert len(exp) > 5
    assert [(int(a), int(b)) for a, b, _, _ in got] == [(int(a), int(b)) for a, b, _, _ in exp]
    assert [s for _, _, _, s in got] == [s for _, _, _, s in exp]
    np.testing.assert_allclose([q for _, _, q, _ in got], [q for _, _, q, _ in exp], rtol=1e-9)


def test_device_bh_equals_numpy_bh(eng, golden_dir):
    """mst_bh_fdr (hipCUB segmented sort + suffix minimum) == the NumPy restatement of statsmodels' fdr_bh, bit for bit,
    and == the reference's multipletests output captured in the fixture."""
    from mustache_amd.tail import benjamini_hochberg
    g = _load(golden_dir, "block_512.npz")
    c = _dense(g)
    dev, nz_d, nzc, found, fit = _run_block(eng, c, int(g["dpx"]))
    assert "q" in found
    assert np.array_equal(found["q"], benjamini_hochberg(found["pval"]))
    np.testing.assert_allclose(found["q"], g["bh_out"], rtol=1e-9)


def _big_block(g):
    from mustache_amd.synth import synth_coo
    n, dpx = int(g["n"]), int(g["dpx"])
    x, y, v = synth_coo(n, dpx, depth=float(g["depth"]), seed=int(g["seed"]))
    assert len(v) == int(g["in_nnz"]) and v.sum() == float(g["in_checksum"]), "synthetic generator drifted"
    c = np.zeros((n, n))
    c[x, y] = v
    return c, n, dpx


def _check_found_set(eng, g, nzc, found, fit):
    assert nzc == int(g["nz_count"])
    pix = found["pixel"].astype(np.int64)
    assert len(pix) == int(g["found_count"])
    assert int(pix.sum()) == int(g["found_pixel_sum"]) and int(np.bitwise_xor.reduce(pix)) == int(g["found_pixel_xor"])
    assert np.array_equal(pix[:4096], g["found_pixels_head"].astype(np.int64))
    assert np.array_equal(found["value"][:4096], g["found_values_head"])
    assert float(np.max(found["value"])) == float(g["found_value_max"])
    sig = np.asarray(eng.levels.tested_sigma)[found["level"].astype(int) - 1]
    np.testing.assert_allclose(float(np.sum(sig)), float(g["found_sigma_sum"]), rtol=1e-13)
    np.testing.assert_allclose(float(np.sum(found["value"])), float(g["found_value_sum"]), rtol=1e-12)
    # (the reference overwrites pAll with the BH q-values before it returns, mustache.py:778-779: the fixture's "pvalue" sum is the q sum)
    np.testing.assert_allclose(float(np.sum(found["q"])), float(g["found_pvalue_sum"]), rtol=1e-9)
    assert np.array_equal(fit[0], g["fit"][:, 0])
    np.testing.assert_allclose(fit[1], g["fit"][:, 1], rtol=1e-12)
    if "full" in g:
        # the reference's COMPLETE found set (block_4000_full.npz: every found pixel, its recorded scale, its winning DoG value
        # and its q-value) -- a regression is localised to the pixel, not to a checksum
        f = g["full"]
        assert np.array_equal(found["pixel"], f["pixel"]), np.flatnonzero(found["pixel"] != f["pixel"])[:10]
        sig_ref = f["sigma_values"][f["sigma_index"]]
        assert np.array_equal(sig, sig_ref), np.flatnonzero(sig != sig_ref)[:10]
        assert np.array_equal(found["value"], f["value"]), np.flatnonzero(found["value"] != f["value"])[:10]
        np.testing.assert_allclose(found["q"], f["q"], rtol=1e-9)


@pytest.mark.parametrize("name", ["block_2000.npz", "block_4000.npz"])
def test_baseline_size_block_vs_reference_fixture(eng, golden_dir, name):
    """BASELINE's block geometries -- 5 kb: 2000 x 2000, distance limit 400 px; 1 kb (the headline workload): 4000 x 4000,
    distance limit 2000 px -- against outputs of the REFERENCE ITSELF (tests/golden/block_*.npz, made by make_golden.py):
    tested-pixel count, the whole found set through order-independent checksums plus its first 4096 records, the 18 expon
    fits, and the final loop list of the drop-in mustache().  Every form of the fused kernel is held to it: dense-block
    source and band-direct source, each with and without empty-tile skipping."""
    import torch
    from mustache_amd.mustache import mustache
    from mustache_amd.normalize import band_from_coo
    g = _load(golden_dir, name)
    full = os.path.join(golden_dir, name.replace(".npz", "_full.npz"))
    if os.path.exists(full):
        g = dict(g.items())
        g["full"] = dict(np.load(full).items())
        assert len(g["full"]["pixel"]) == int(g["found_count"]) > 100_000
    c, n, dpx = _big_block(g)
    for skip_empty in (True, False):
        dev, nz_d, nzc, found, fit = _run_block(eng, c.copy(), dpx, skip_empty)
        _check_found_set(eng, g, nzc, found, fit)
        del dev, nz_d
    x, y = np.nonzero(c)
    band = band_from_coo(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(c[x, y]).cuda(), n, dpx)
    for skip_empty in (True, False):
        found, fits, nzc = eng.sigma_loop_band(band, n, dpx, [0], n, skip_empty=skip_empty)
        _check_found_set(eng, g, int(nzc.cpu().numpy().view(np.uint32)[0]), found[0], fits[0])
    start = int(g["start"])
    loops = mustache(c, "1", "1", 5000, [], start, start + n, 0, dpx, OCT, float(g["st"]), float(g["pt"]))
    exp = g["loops"]
    got = np.array([[float(a), float(b), q, s] for a, b, q, s in loops])
    assert got.shape == exp.shape and len(exp) > 100
    assert np.array_equal(got[:, :2], exp[:, :2]) and np.array_equal(got[:, 3], exp[:, 3])
    np.testing.assert_allclose(got[:, 2], exp[:, 2], rtol=1e-9)


@pytest.mark.parametrize("name", ["block_700_oc3.npz", "block_640_sz2.npz"])
def test_sigma_zero_and_octave_variants_vs_reference_fixture(golden_dir, name):
    """-sz / -oc (mustache.py:874: octave_values = sigma0 * 2^i) against outputs of the REFERENCE ITSELF: three octaves
    (1.6, 3.2, 6.4 -- radii up to 28: the wide-halo tile, 27 tested levels) and sigma0 = 2.0 (2.0, 4.0).  Found set, expon
    fits and final loops as in the BASELINE-size test; dense-block and band-direct source, with and without tile skipping."""
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.mustache import mustache
    from mustache_amd.normalize import band_from_coo
    g = _load(golden_dir, name)
    octs = [float(o) for o in g["octaves"]]
    eng = ScaleSpaceEngine(octs)
    assert eng.levels.n_tested == len(g["fit"])
    c, n, dpx = _big_block(g)
    for skip_empty in (True, False):
        dev, nz_d, nzc, found, fit = _run_block(eng, c.copy(), dpx, skip_empty)
        _check_found_set(eng, g, nzc, found, fit)
    x, y = np.nonzero(c)
    band = band_from_coo(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(c[x, y]).cuda(), n, dpx)
    for skip_empty in (True, False):
        found, fits, nzc = eng.sigma_loop_band(band, n, dpx, [0], n, skip_empty=skip_empty)
        _check_found_set(eng, g, int(nzc.cpu().numpy().view(np.uint32)[0]), found[0], fits[0])
    start = int(g["start"])
    loops = mustache(c, "1", "1", 5000, [], start, start + n, 0, dpx, octs, float(g["st"]), float(g["pt"]))
    exp = g["loops"]
    got = np.array([[float(a), float(b), q, s] for a, b, q, s in loops])
    assert got.shape == exp.shape and len(exp) > 10
    assert np.array_equal(got[:, :2], exp[:, :2]) and np.array_equal(got[:, 3], exp[:, 3])
    np.testing.assert_allclose(got[:, 2], exp[:, 2], rtol=1e-9)


def test_complete_found_set_at_headline_geometry_vs_oracle():
    """BASELINE config 4's block geometry (4000 x 4000, distance limit 2000) against the CPU oracle on the COMPLETE found
    set -- not the checksums + first records block_4000.npz holds: three consecutive blocks of a synthetic 1 kb chromosome
    go through ONE launch of the band-direct kernel, so the middle block receives the tiles it shares with the block before
    it and gives those it shares with the block after it; its tested-pixel count, every found pixel, level and DoG value must
    equal the oracle's (rows 3-7: SciPy gaussian_filter / maximum_filter, ~10 s), loc exactly, scale to 1e-12, p to 1e-9."""
    import torch
    import oracle
    from mustache_amd.normalize import normalize_band
    from mustache_amd.pipeline import ChromosomePipeline, block_tiling
    from mustache_amd.synth import band_counts
    n, dpx, res = 8000, 2000, 1000
    pipe = ChromosomePipeline([1.6, 3.2])
    raw = band_counts(n, dpx, 400.0, 300, 5, device=pipe.device)
    band, _, _ = normalize_band(raw, n, dpx, res)
    CH, start, end = block_tiling(n, dpx)
    assert CH == 4000 and len(start) == 3
    recs, fits, nzc = pipe.engine.sigma_loop_band(band, n, dpx, start, CH, skip_empty=True, with_q=False)
    g = {k: v.copy() for k, v in recs[1].items()}
    s = start[1]
    slab = band[:, s:s + CH].cpu().numpy()
    c = np.zeros((CH, CH))
    r = np.arange(CH)
    for d in range(dpx + 2):
        L = CH - d
        c[r[:L], r[:L] + d] = slab[d, :L]
    nz = oracle.block_prologue(c, dpx)
    ss = oracle.scale_space_levels(c, nz, [1.6, 3.2], blur="scipy")
    hit = ss.pval != 2
    assert int(nzc.cpu().numpy().view(np.uint32)[1]) == int(nz.sum()) > 3_000_000
    pix = np.flatnonzero(nz.ravel())[hit].astype(np.uint32)
    assert len(pix) > 100_000
    assert np.array_equal(g["pixel"], pix)
    assert np.array_equal(g["level"], ss.level[hit].astype(np.uint32))
    assert np.array_equal(g["value"], ss.best[hit])
    np.testing.assert_allclose(g["pval"], ss.pval[hit], rtol=1e-9)
    np.testing.assert_array_equal(fits[1][0][:18], np.array([t["loc"] for t in ss.tested]))
    np.testing.assert_allclose(fits[1][1][:18], np.array([t["scale"] for t in ss.tested]), rtol=1e-12)
