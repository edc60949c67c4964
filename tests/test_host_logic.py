"""CPU-side tests (no GPU): C-ABI library loads and exports every declared symbol; host logic (tiling, CLI
helpers, readers, tail, sharding) against the oracle and the reference-generated fixtures."""
import os
import re

import numpy as np
import pytest

OCT = [1.6, 3.2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mustache_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mustache_hip.h")).read()
    tagged = re.findall(r"^(MST_STABLE|MST_INTERNAL) (?:int|uint64_t|const char \*)\s*\*?\s*(mst_[a-z_0-9]+)\s*\(", hdr, flags=re.M)
    declared = set(name for _tag, name in tagged)
    assert len(declared) == len(tagged) >= 14
    # every entry point carries exactly one stability tag; no untagged declaration is left
    assert not re.findall(r"^(?:int|uint64_t|const char \*)\s*\*?\s*mst_[a-z_0-9]+\s*\(", hdr, flags=re.M)
    internal = set(name for tag, name in tagged if tag == "MST_INTERNAL")
    assert internal == {"mst_bh_select_nowait", "mst_band_scatter_packed", "mst_band_scatter_hic_rows",
                        "mst_band_verify_packed", "mst_scale_space_band_tiles", "mst_scale_space_band_items",
                        "mst_candidate_features_band_multi", "mst_diag_means_band_multi", "mst_scale_space_band_stage"}
    # the five of SURVEY 8(b) and the band forms the per-chromosome driver uses are on the stable side
    for name in ("mst_normalize_band", "mst_scatter_blocks", "mst_gauss_blur", "mst_scale_space", "mst_found_pvalues",
                 "mst_found_finish", "mst_scale_space_band", "mst_bh_select", "mst_cluster_representatives", "mst_diff_dog_band"):
        assert name in declared - internal, name
    lib = _lib.load()                      # raises if any bound symbol is missing
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.exported_symbols())
    assert lib.mst_abi_version() == _lib.MST_ABI_VERSION == 3


def test_level_table_matches_oracle():
    import oracle
    from mustache_amd.levels import LevelTable
    lt = LevelTable(OCT)
    lv = oracle.level_table(OCT)
    assert lt.radius == [l["radius"] for l in lv]
    assert lt.sigma == [l["sigma"] for l in lv]
    for a, b in zip(lt.taps, lv):
        assert np.array_equal(a, b["weights"][b["radius"]:])
    assert lt.n_tested == 18 and lt.tested_sigma[0] == lv[2]["sigma"] and lt.tested_sigma[-1] == lv[22]["sigma"]
    with pytest.raises(ValueError):
        LevelTable([1.6, 3.2, 6.4, 12.8])   # radius > 28: not supported by the kernel, must fail loudly


def test_tiling_matches_reference(golden_dir):
    from mustache_amd.mustache import block_tiling, block_mask_size
    g = np.load(os.path.join(golden_dir, "tiling.npz"), allow_pickle=True)
    for i in range(len(g["n"])):
        n, dpx = int(g["n"][i]), int(g["dpx"][i])
        chunk, start, end = block_tiling(n, dpx)
        assert chunk == int(g["chunk"][i]) and start == list(g["starts"][i]) and end == list(g["ends"][i])
        assert [block_mask_size(b, start, end, dpx) for b in range(len(start))] == list(g["masks"][i])


def test_cli_helpers():
    from mustache_amd.mustache import parseBP, resolve_distance_filter, parse_args, is_chr
    assert parseBP("5kb") == 5000 and parseBP("1mb") == 1000000 and parseBP("2500") == 2500
    assert parseBP("kb") is False and parseBP("") is False and parseBP("5xb") is False
    assert resolve_distance_filter(None, 5000, quiet=True) == 2000000      # mustache.py:1004-1006
    assert resolve_distance_filter(None, 1000, quiet=True) == 2000000      # 2000 * res
    assert resolve_distance_filter(None, 25000, quiet=True) == 5000000     # 200 * res
    assert resolve_distance_filter("100kb", 5000, quiet=True) == 1000000   # clamp to 200 * res
    assert resolve_distance_filter("90mb", 5000, quiet=True) == 50000000   # clamp to 10000 * res
    a = parse_args(["-f", "x", "-r", "5kb", "-o", "o.tsv", "-ch", "21"])
    assert (a.pt, a.st, a.s_z, a.octaves, a.nprocesses) == (0.2, 0.88, 1.6, 2, 4) and a.chromosome == ["21"]
    assert is_chr("chr21", "21") and is_chr(21, "chr21") and not is_chr("chr2", "21")


def test_text_reader(tmp_path):
    from mustache_amd.mustache import read_pd, read_bias
    res = 5000
    f = tmp_path / "c.txt"
    f.write_text("0\t5000\t10\n5000\t20000\t4\n10000\t10000\t7\n0\t3000000\t9\n15000\t20000\t3\n")
    b = tmp_path / "b.txt"
    b.write_text("chr1\t0\t1.0\nchr1\t5000\t2.0\nchr1\t10000\tNaN\nchr1\t15000\t0.1\nchr1\t20000\t0.5\nchr2\t0\t9\n")
    bias = read_bias(str(b), "1", res)
    assert bias[0] == 1.0 and bias[1] == 2.0 and np.isinf(bias[2]) and np.isinf(bias[3]) and bias[4] == 0.5
    assert bias[77] == 1.0
    x, y, v = read_pd(str(f), 2000000, str(b), "1", res)
    # row 3 (diagonal at the NaN bin) and row 5 (bias < 0.2) vanish; row 4 is beyond the distance limit
    assert list(x) == [0, 1] and list(y) == [1, 4] and list(v) == [5.0, 4.0]
    f5 = tmp_path / "c5.txt"
    f5.write_text("chr1 0 chr1 5000 10\nchr2 0 chr2 5000 3\nchr1 20000 chr1 5000 4\n")
    x, y, v = read_pd(str(f5), 2000000, False, "1", res)
    assert list(x) == [0, 1] and list(y) == [1, 4] and list(v) == [10, 4]


class _FakeBatch:
    """BlockBatch stand-in whose gathers come from host arrays: checks the host tail logic without a GPU."""

    def __init__(self, c, nz, found, levels):
        from types import SimpleNamespace
        self.c_h, self.nz_h, self.CH, self.B = c, nz, c.shape[0], 1
        self.nz_count = [int(nz.sum())]
        self.found = [found]
        self.engine = SimpleNamespace(levels=levels)

    def candidate_features(self, b, pixel, half):
        CH = self.CH
        c1, c2, cv = [], [], []
        for p, h in zip(pixel, half):
            x, y = int(p) // CH, int(p) % CH
            c1.append(int(np.sum(self.nz_h[x - h:x + h + 1, y - h:y + h + 1])))
            c2.append(int(np.sum(self.nz_h[x - 2 * h:x + 2 * h + 1, y - 2 * h:y + 2 * h + 1])))
            cv.append(self.c_h[x, y])
        return np.array(c1, np.uint32), np.array(c2, np.uint32), np.array(cv)

    def diagonals(self, b, ks):
        out = np.zeros((len(ks), self.CH))
        for i, k in enumerate(ks):
            d = np.diagonal(self.c_h, int(k))
            out[i, :len(d)] = d
        return out


@pytest.mark.parametrize("name", ["block_320.npz", "block_512.npz"])
def test_host_tail_vs_reference_fixture(golden_dir, name):
    import oracle
    from mustache_amd.levels import LevelTable
    from mustache_amd.tail import block_tail, benjamini_hochberg
    g = np.load(os.path.join(golden_dir, name), allow_pickle=True)
    n, dpx = int(g["n"]), int(g["dpx"])
    c = np.zeros((n, n))
    c[g["x"], g["y"]] = g["v"]
    nz = oracle.block_prologue(c, dpx)
    ss = oracle.scale_space_levels(c, nz, OCT)
    fm = ss.pval != 2
    found = dict(pixel=np.flatnonzero(nz.ravel())[fm].astype(np.uint32), level=ss.level[fm].astype(np.uint32),
                 value=ss.best[fm], pval=ss.pval[fm])
    assert np.array_equal(benjamini_hochberg(found["pval"]), g["bh_out"])
    loops = block_tail(_FakeBatch(c, nz, found, LevelTable(OCT)), 0, int(g["start"]), float(g["pt"]), float(g["st"]))
    got = np.array([[float(a), float(b), q, s] for a, b, q, s in loops]).reshape(-1, 4)
    assert np.array_equal(got, g["loops"]), "host tail must reproduce the reference's loops bit for bit, in order"


def test_components_match_scipy_label():
    from scipy.ndimage import label
    from mustache_amd.tail import _components
    rng = np.random.default_rng(0)
    for _ in range(20):
        m = rng.random((40, 40)) < 0.18
        px, py = np.nonzero(m)
        order, labels, n = _components(px, py)
        ref, nref = label(m, structure=np.ones((3, 3)))
        assert n == nref
        assert np.array_equal(ref[px[order], py[order]] - 1, labels)


def test_read_bias_on_the_reference_bundled_krnorm(tmp_path, golden_dir):
    """data/chr21_5kb.KRnorm, the one real data file the reference ships (3 columns `chr21 <tab> position <tab> KR bias`,
    9630 rows: 6813 finite, 2817 NaN, 185 finite below 0.2), through read_bias's 3-column branch (mustache.py:229-238):
    the fixture holds the file's numbers and the dictionary the REFERENCE's read_bias returned for it."""
    from mustache_amd.mustache import read_bias
    g = np.load(os.path.join(golden_dir, "krnorm_chr21_5kb.npz"))
    path = tmp_path / "chr21_5kb.KRnorm"
    with open(path, "w") as fh:
        for p, v in zip(g["pos"], g["value"]):
            fh.write("chr21\t%d\t%s\n" % (int(p), "NaN" if np.isnan(v) else repr(float(v))))
    res = int(g["res"])
    d = read_bias(str(path), "21", res)                       # "21" matches "chr21" (is_chr)
    keys = np.array(sorted(d), dtype=np.float64)
    assert np.array_equal(keys, g["bias_keys"])
    got = np.array([d[k] for k in keys])
    assert np.array_equal(got, g["bias_values"])              # inf where the factor is NaN or < 0.2, the factor otherwise
    assert int(np.isinf(got).sum()) == 2817 + 185
    assert d[1e9] == 1.0                                      # bins the file does not list default to 1
    assert len(read_bias(str(path), "20", res)) == int(g["other_chrom_entries"]) == 0


@pytest.mark.parametrize("quantum", [0.05, 0.01])
def test_cluster_representative_with_tied_q_values(golden_dir, quantum):
    """BH maps runs of records to one q-value, and the reference takes the FIRST arg-min of q over a cluster's pixels in raster
    order (mustache.py:843-848).  Stress that rule: the found p-values of a fixture block are quantised so that most clusters
    hold several pixels with exactly equal q; the product's host tail must still pick the same representatives as the oracle's
    restatement of the reference's tail."""
    import oracle
    from mustache_amd.levels import LevelTable
    from mustache_amd.tail import block_tail
    g = np.load(os.path.join(golden_dir, "block_512.npz"), allow_pickle=True)
    n, dpx = int(g["n"]), int(g["dpx"])
    c = np.zeros((n, n))
    c[g["x"], g["y"]] = g["v"]
    nz = oracle.block_prologue(c, dpx)
    ss = oracle.scale_space_levels(c, nz, OCT)
    fm = ss.pval != 2
    pq = ss.pval.copy()
    pq[fm] = np.minimum(1.0, np.ceil(ss.pval[fm] / quantum) * quantum * 0.1)      # few distinct values, small enough to be selected
    exp = oracle.block_tail(c, nz, pq, ss.scale, int(g["start"]), 0.5, 0.5)
    found = dict(pixel=np.flatnonzero(nz.ravel())[fm].astype(np.uint32), level=ss.level[fm].astype(np.uint32),
                 value=ss.best[fm], pval=pq[fm])
    got = block_tail(_FakeBatch(c, nz, found, LevelTable(OCT)), 0, int(g["start"]), 0.5, 0.5)
    ga = np.array([[float(a), float(b), q, s] for a, b, q, s in got]).reshape(-1, 4)
    ea = np.array([[float(a), float(b), q, s] for a, b, q, s in exp]).reshape(-1, 4)
    assert len(ea) > 20 and np.array_equal(ga, ea)
    # the stress is real: many selected records share their q-value with another one
    from mustache_amd.tail import benjamini_hochberg
    q = benjamini_hochberg(found["pval"])
    sel = q[q < 0.5]
    assert len(sel) - len(np.unique(sel)) > len(sel) // 2


def test_launched_tile_count_matches_the_geometric_rule():
    """mst_scale_space_band_tiles (what bench.py uses to price the skipping mode) == a direct restatement of the rule: a
    30 x 62 tile is launched iff its pixels can reach the tested band 4 <= col - row <= dpx + 1."""
    import ctypes
    from mustache_amd import _lib
    from mustache_amd.levels import LevelTable
    lib = _lib.load()
    lv = LevelTable(OCT).as_struct()
    for CH, dpx in ((2000, 400), (4000, 2000), (700, 160), (320, 80), (8000, 4000), (2000, 5)):
        ty, tx = -(-CH // 30), -(-CH // 62)
        m = 0
        for j in range(ty):
            r_lo, r_hi = j * 30, min(j * 30 + 29, CH - 1)
            for i in range(tx):
                c_lo, c_hi = i * 62, min(i * 62 + 61, CH - 1)
                m += (c_hi - r_lo >= 4) and (c_lo - r_hi <= dpx + 1)
        total = ctypes.c_int32(0)
        assert lib.mst_scale_space_band_tiles(CH, dpx, ctypes.byref(lv), ctypes.byref(total)) == m
        assert total.value == ty * tx
    assert lib.mst_scale_space_band_tiles(0, 5, ctypes.byref(lv), None) < 0


def test_genome_layout_windows_equal_the_chromosome_windows():
    """pipeline.GenomeLayout (whole-genome batching): every block window [start, start + CHUNK) of the side-by-side band
    holds exactly the chromosome's own window, zero past the chromosome's end, never the next chromosome's data."""
    import torch
    from mustache_amd.pipeline import GenomeLayout
    from mustache_amd.mustache import block_tiling
    dpx = 400
    ns = [5200, 1500, 2000, 4300, 700, 2001]
    lay = GenomeLayout(ns, dpx)
    assert lay.CH == 2000 and all(s >= max(n, 2000) and s % 64 == 0 for s, n in zip(lay.slot, ns))
    assert lay.off == [sum(lay.slot[:i]) for i in range(len(ns))] and lay.N == sum(lay.slot)
    rng = np.random.default_rng(0)
    bands = [torch.from_numpy(rng.uniform(0.5, 2.0, (dpx + 2, n))) for n in ns]
    g = lay.band(bands, "cpu")
    assert g.shape == (dpx + 2, lay.N)
    k = 0
    for c, n in enumerate(ns):
        CH, start, end = block_tiling(n, dpx)
        for i, s in enumerate(start):
            assert lay.blocks[k] == (c, i, s, lay.off[c] + s)
            win = g[:, lay.off[c] + s: lay.off[c] + s + CH]
            own = torch.zeros((dpx + 2, CH), dtype=torch.float64)
            own[:, :min(CH, n - s)] = bands[c][:, s:s + CH]
            assert torch.equal(win, own), (c, i)
            k += 1
    assert k == len(lay.blocks)


def _work_list_counts(start, CH, dpx, share, band_only, ITR=30, ITC=62, RMAX=14):
    """The tile-sharing rule of mst_scale_space_band restated (DESIGN 3.1 "tile sharing"): tiles of ITR x ITC owned pixels on a
    lattice anchored at chromosome coordinate 0 (at the block's origin without sharing); a tile is listed for block b unless
    block b - 1 delivers it; it is delivered to block b + 1 as well iff its staged window -- owned pixels + 1-pixel ring + RMAX
    halo -- lies inside BOTH blocks.  Returns (work items, lattice cells visited, shared tiles)."""
    items = tiles = shared = 0
    given = set()
    for b, s in enumerate(start):
        s0 = s if share else 0
        nxt = share and b + 1 < len(start) and 0 < start[b + 1] - s < CH
        given_next = set()
        for a in range(s0 // ITR, (s0 + CH - 1) // ITR + 1):
            for c in range(s0 // ITC, (s0 + CH - 1) // ITC + 1):
                tiles += 1
                y0, x0 = a * ITR - s0 - 1, c * ITC - s0 - 1                  # block coordinates of the ring's corner
                r_lo, r_hi = max(y0 + 1, 0), min(y0 + ITR, CH - 1)
                q_lo, q_hi = max(x0 + 1, 0), min(x0 + ITC, CH - 1)
                if r_lo > r_hi or q_lo > q_hi:
                    continue
                if band_only and not (q_hi - r_lo >= 4 and q_lo - r_hi <= dpx + 1):
                    continue
                if (a, c) in given:
                    continue
                items += 1
                if nxt:
                    d = s - start[b + 1]
                    Y0, X0, H, W = y0 - RMAX, x0 - RMAX, ITR + 2 + 2 * RMAX, ITC + 2 + 2 * RMAX
                    inside = lambda o: Y0 + o >= 0 and X0 + o >= 0 and Y0 + o + H <= CH and X0 + o + W <= CH
                    if inside(0) and inside(d):
                        shared += 1
                        given_next.add((a, c))
        given = given_next
    return items, tiles, shared


def test_work_list_of_a_launch_matches_the_sharing_rule():
    """mst_scale_space_band_items (the launch's work list, host code of the C ABI) against the rule restated above, for the
    headline geometry, 5 kb, an irregular last block, a chromosome of one block, and with MST_FLAG_NO_SHARE / the tile list."""
    import ctypes
    from mustache_amd import _lib
    from mustache_amd.levels import LevelTable
    from mustache_amd.pipeline import block_tiling
    lib = _lib.load()
    lv = LevelTable(OCT).as_struct()
    for n, dpx in ((31119, 2000), (9630, 400), (14321, 2000), (5230, 150), (1800, 400), (12345, 1507)):
        CH, start, end = block_tiling(n, dpx)
        st = (ctypes.c_int64 * len(start))(*[int(s) for s in start])
        for flags in (0, 1, 4, 5):                    # 1 = MST_FLAG_SKIP_EMPTY (tile list), 4 = MST_FLAG_NO_SHARE
            t, s = ctypes.c_int64(0), ctypes.c_int64(0)
            got = lib.mst_scale_space_band_items(st, len(start), CH, dpx, ctypes.byref(lv), flags, ctypes.byref(t), ctypes.byref(s))
            exp = _work_list_counts([int(v) for v in start], CH, dpx, not (flags & 4), bool(flags & 1))
            assert (got, s.value) == (exp[0], exp[2]), (n, dpx, flags, got, s.value, exp)
            assert t.value == got + s.value          # "tiles" = what the blocks would run one by one
            if flags == 4:                            # every block on its own lattice, every tile once per block
                assert s.value == 0 and got == len(start) * (-(-CH // 30)) * (-(-CH // 62))
    # at 1 kb (blocks overlap by half their edge) the sharing removes a fifth of the dense work and more of the tile list
    CH, start, end = block_tiling(248957, 2000)
    st = (ctypes.c_int64 * len(start))(*[int(s) for s in start])
    t, s = ctypes.c_int64(0), ctypes.c_int64(0)
    it = lib.mst_scale_space_band_items(st, len(start), CH, 2000, ctypes.byref(lv), 0, ctypes.byref(t), ctypes.byref(s))
    assert 0.20 < s.value / t.value < 0.25 and it + s.value <= t.value


def test_settle_gc_freezes_once_and_can_be_switched_off(monkeypatch):
    """engine.settle_gc: everything alive at the first engine's construction leaves the cyclic collector's generations
    (gc.freeze), once per process; MUSTACHE_GC_FREEZE=0 leaves the collector alone."""
    import gc
    from mustache_amd import engine
    monkeypatch.setattr(engine, "_GC_SETTLED", False)
    monkeypatch.setenv("MUSTACHE_GC_FREEZE", "0")
    before = gc.get_freeze_count()
    engine.settle_gc()
    assert gc.get_freeze_count() == before and engine._GC_SETTLED is False
    monkeypatch.setenv("MUSTACHE_GC_FREEZE", "1")
    try:
        engine.settle_gc()
        frozen = gc.get_freeze_count()
        assert frozen > before and engine._GC_SETTLED is True
        junk = [[i] for i in range(1000)]               # objects made afterwards are collected as ever
        engine.settle_gc()
        assert gc.get_freeze_count() == frozen and len(junk) == 1000
    finally:
        gc.unfreeze()                                   # leave the test process as it was


def test_fill_like_reference_equals_numpy_row_slices():
    """mustache.fill_like_reference (mst_host_fill_block for C-contiguous rows, NumPy slices otherwise) == the fills of
    mustache.py:703-706 spelled as row slices, incl. views with a larger row stride and non-contiguous layouts."""
    from mustache_amd.mustache import fill_like_reference

    def spelled(c, dpx, intra):
        n = c.shape[0]
        for r in range(n):
            c[r, :min(n, r + 5)] = 2.0
            if intra:
                c[r, r + dpx + 1:] = 2.0

    rng = np.random.default_rng(5)
    for n, dpx, intra in ((1, 0, True), (5, 0, True), (7, 2, False), (300, 50, True), (300, 500, True), (1025, 333, True)):
        a = rng.uniform(size=(n, n))
        b = a.copy()
        fill_like_reference(a, dpx, intra)
        spelled(b, dpx, intra)
        assert np.array_equal(a, b), (n, dpx, intra)
    big = rng.uniform(size=(80, 100))
    view, rest = big[:, :80], big[:, 80:].copy()
    want = view.copy()
    fill_like_reference(view, 10, True)                 # rows 100 doubles apart
    spelled(want, 10, True)
    assert np.array_equal(view, want) and np.array_equal(big[:, 80:], rest)
    f = np.asfortranarray(rng.uniform(size=(50, 50)))   # column-major: the NumPy form
    want = np.ascontiguousarray(f)
    fill_like_reference(f, 10, True)
    spelled(want, 10, True)
    assert np.array_equal(f, want)


def test_write_loops_text_equals_the_references_str_of_numpy_scalars(tmp_path):
    """write_loops formats rows with int() / repr(float()); the reference concatenates str() of NumPy scalars
    (mustache.py:1098-1103).  Same bytes, incl. tiny and huge q-values, integer-valued floats and large coordinates."""
    from mustache_amd.mustache import write_loops
    rng = np.random.default_rng(11)
    q = np.concatenate([rng.uniform(0, 0.2, 3000), 10.0 ** rng.uniform(-300, -3, 3000), [0.0, 1.0, 1e-5, 1e-4, 9.999e-05, 5e-324, 0.1]])
    m = len(q)
    xs, ys = rng.integers(0, 3_000_000, m), rng.integers(0, 3_000_000, m)
    sig = rng.choice([1.8379173679952558, 2.111212657236631, 3.2, 6.4, 4.0], m)
    loops = [[np.int64(a), np.int64(b), np.float64(c), np.float64(d)] for a, b, c, d in zip(xs, ys, q, sig)]
    res = 1000
    for chrom in ("chr1", 21):
        p = str(tmp_path / ("out_%s.tsv" % chrom))
        write_loops(p, chrom, chrom, res, loops, first=True)
        want = "BIN1_CHR\tBIN1_START\tBIN1_END\tBIN2_CHROMOSOME\tBIN2_START\tBIN2_END\tFDR\tDETECTION_SCALE\n" + "".join(
            str(chrom) + '\t' + str(lp[0] * res) + '\t' + str((lp[0] + 1) * res) + '\t' + str(chrom) + '\t' + str(lp[1] * res) + '\t' +
            str((lp[1] + 1) * res) + '\t' + str(lp[2]) + '\t' + str(lp[3]) + '\n' for lp in loops)
        assert open(p).read() == want
    vals = np.concatenate([rng.uniform(0, 1, 20000), 10.0 ** rng.uniform(-320, 308, 20000), rng.uniform(1e15, 1e17, 2000)])
    assert all(repr(float(v)) == str(np.float64(v)) for v in vals)


def test_engine_constructor_leaves_the_collector_alone(monkeypatch):
    """Library use (a maintainer changes imports only): constructing the engine must not freeze the
    host application's garbage collector; only the command-line entry points and bench.py call settle_gc(), and
    MUSTACHE_GC_FREEZE=1 is the library's opt-in."""
    import gc
    import inspect
    from mustache_amd import engine, mustache as mm, diff_mustache as dm
    monkeypatch.setattr(engine, "_GC_SETTLED", False)
    monkeypatch.setattr(engine, "require_gpu", lambda: object())
    monkeypatch.delenv("MUSTACHE_GC_FREEZE", raising=False)
    before = gc.get_freeze_count()
    engine.ScaleSpaceEngine((1.6, 3.2), device="cpu")
    assert gc.get_freeze_count() == before and engine._GC_SETTLED is False
    monkeypatch.setenv("MUSTACHE_GC_FREEZE", "1")
    try:
        engine.ScaleSpaceEngine((1.6, 3.2), device="cpu")
        assert gc.get_freeze_count() > before and engine._GC_SETTLED is True
    finally:
        gc.unfreeze()
    assert "settle_gc()" in inspect.getsource(mm.main) and "settle_gc()" in inspect.getsource(dm.main)
    assert "settle_gc()" not in inspect.getsource(mm.mustache) and "settle_gc()" not in inspect.getsource(mm.regulator)


def test_trim_band_in_place_moves_rows_inside_one_allocation():
    """normalize._trim_band_in_place: [rows, n_alloc] -> [rows, n] without a second band (groups of rows through a bounded
    temporary; the result is a view of the same storage) -- for every group size, including overlapping source / destination rows."""
    import torch
    from mustache_amd.normalize import _trim_band_in_place
    for rows, na, n, tb in ((7, 10, 9, 80), (2002, 500, 497, 8 * 497 * 50), (5, 6, 1, 8), (9, 100, 3, 64), (4, 9, 8, 8 * 8 * 4), (1, 9, 8, 1)):
        b = torch.arange(rows * na, dtype=torch.float64).view(rows, na).clone()
        want = b[:, :n].clone()
        got = _trim_band_in_place(b, n, tb)
        assert torch.equal(got, want) and got.is_contiguous() and got.data_ptr() == b.data_ptr(), (rows, na, n)
    assert _trim_band_in_place(torch.ones(3, 4, dtype=torch.float64), 0).shape == (3, 0)


def test_scalar_text_fast_path_is_limited_to_doubles():
    """mustache._scalar_text: repr(float(v)) equals the reference's str(v) for Python floats and np.float64 only; a float32
    (str() prints its own shortest repr) and everything else go through str()."""
    from mustache_amd.mustache import _scalar_text
    for v in (0.1, np.float64(1) / 3, np.float64(1e-300), 2.0):
        assert _scalar_text(v) == str(v)
    assert _scalar_text(np.float32(0.1)) == str(np.float32(0.1)) == "0.1"
    assert _scalar_text(np.int64(7)) == "7"


def test_split_groups_keeps_order_and_makes_the_last_stage_small():
    """pipeline.split_groups: groups of `per` items in order; a last group of six or more blocks and ~100 Mpix gives all but
    6 % (at least two) of its items to a group in front of it -- only the last stage's finish is exposed; small launches stay whole."""
    from mustache_amd.pipeline import split_groups
    for n, per, px, want in ((124, 16, 16e6, [16] * 7 + [10, 2]), (6, 64, 4e6, [6]), (63, 64, 4e6, [59, 4]), (20, 64, 4e6, [20]),
                             (16, 16, 16e6, [14, 2]), (17, 16, 16e6, [16, 1]), (5, 16, 16e6, [5]), (0, 16, 16e6, [])):
        g = split_groups(list(range(n)), per, px)
        assert [len(a) for a in g] == want and sum(g, []) == list(range(n)), (n, per)


def test_bench_stage_cuts():
    """bench.stage_cuts: the stages of a step's one launch -- small launches whole, the last stage small (the only one whose
    finish is exposed), two equal stages in front from 24 blocks up; the driver's rank counts on the 124-block workload."""
    import bench
    assert bench.stage_cuts(124, 4000) == [0, 59, 117, 124] and bench.stage_cuts(62, 4000) == [0, 29, 58, 62]
    assert bench.stage_cuts(31, 4000) == [0, 15, 29, 31] and bench.stage_cuts(16, 4000) == [0, 15, 16] and bench.stage_cuts(15, 4000) == [0, 14, 15]
    assert bench.stage_cuts(6, 2000) == [0, 6] and bench.stage_cuts(3, 4000) == [0, 3] and bench.stage_cuts(12, 4000) == [0, 11, 12]
    assert bench.stage_cuts(124, 4000, overlap=1) == [0, 124] and bench.stage_cuts(124, 4000, overlap=2) == [0, 117, 124]
    assert bench.stage_cuts(16, 4000, shares="0.5,0.3,0.2") == [0, 8, 13, 16]
    for nb in range(1, 130):
        c = bench.stage_cuts(nb, 4000)
        assert c[0] == 0 and c[-1] == nb and all(b > a for a, b in zip(c[:-1], c[1:]))
