"""GPU parity of the two-sample path (reference diff_mustache.py:260-569) against a fixture produced by the reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OCT = [1.6, 3.2]


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "diff_320.npz"), allow_pickle=True)


def _blocks(g):
    n = int(g["n"])
    c1 = np.zeros((n, n)); c1[g["xa"], g["ya"]] = g["va"]
    c2 = np.zeros((n, n)); c2[g["xb"], g["yb"]] = g["vb"]
    return c1, c2


def test_pair_records_vs_reference(golden_dir):
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    g = _load(golden_dir)
    c1, c2 = _blocks(g)
    eng = ScaleSpaceEngine(OCT)
    batch = eng.run_block_pairs(torch.from_numpy(np.stack([c1, c2])).cuda(), int(g["dpx"]))
    # norm.fit of the difference DoG: one (loc, scale) per octave, repeated for its 9 tested levels in the reference
    ref_fit = g["norm_fit"]
    np.testing.assert_allclose(batch.norm_fit[0, 0], ref_fit[0], rtol=1e-9)
    np.testing.assert_allclose(batch.norm_fit[1, 0], ref_fit[9], rtol=1e-9)
    for b, nm in ((0, "1"), (1, "2")):
        rec = batch.found[b]
        pall, ppair, vall = g["loc_pAll" + nm], g["loc_pPair" + nm], g["loc_vAll" + nm]
        # the reference keeps per-nz arrays; found pixels are those whose pPair was written (pPair != 2)
        f = ppair != 2
        assert f.sum() == len(rec["pixel"])
        assert np.array_equal(rec["value"], vall[f]), "winning DoG values must be bit-identical"
        np.testing.assert_allclose(rec["pair"], ppair[f], rtol=1e-5, atol=1e-300)


def _band_of(c, n, dpx):
    """raw dense block -> band tensor [dpx+2, n] on the GPU (what the per-chromosome driver holds)."""
    import torch
    from mustache_amd.normalize import band_from_coo
    x, y = np.nonzero(np.triu(c))
    return band_from_coo(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(c[x, y]).cuda(), n, dpx)


def test_band_direct_pairs_equal_dense_pairs_and_reference(golden_dir):
    """engine.run_band_pairs (both samples band-direct, difference image / G_2 / G_3 only ever in LDS: mst_diff_dog_band)
    against engine.run_block_pairs (the reference's dense data flow) and the reference's locals: found sets and winning DoG
    values identical, p-values and q identical, pair p-values within 1e-9 of the dense route (the norm.fit scale comes from
    one pass instead of two) and within 1e-5 of the reference."""
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    g = _load(golden_dir)
    c1, c2 = _blocks(g)
    n, dpx = int(g["n"]), int(g["dpx"])
    eng = ScaleSpaceEngine(OCT)
    dense = eng.run_block_pairs(torch.from_numpy(np.stack([c1, c2])).cuda(), dpx)
    band = eng.run_band_pairs([_band_of(c1, n, dpx), _band_of(c2, n, dpx)], n, dpx, [0], n)
    np.testing.assert_allclose(band.norm_fit, dense.norm_fit, rtol=1e-10)
    assert list(band.nz_count) == list(dense.nz_count)
    for b, nm in ((0, "1"), (1, "2")):
        ra, rb = dense.found[b], band.found[b]
        for k in ("pixel", "level", "value", "pval", "q"):
            assert np.array_equal(ra[k], rb[k]), k
        np.testing.assert_allclose(rb["pair"], ra["pair"], rtol=1e-9, atol=1e-300)
        f = g["loc_pPair" + nm] != 2
        np.testing.assert_allclose(rb["pair"], g["loc_pPair" + nm][f], rtol=1e-5, atol=1e-300)
        # the gathers of the tail read the band: same window counts and values as the dense blocks give
        pix = ra["pixel"][:200]
        half = np.full(len(pix), 2, np.int64)
        fa, fb = dense.candidate_features(b, pix, half), band.candidate_features(b, pix, half)
        for u, v in zip(fa, fb):
            assert np.array_equal(u, v)


def test_baseline_geometry_pairs_vs_reference_fixture(golden_dir):
    """BASELINE config 5's block geometry (2000 x 2000, distance limit 400 px) against the reference's own diff_mustache()
    (tests/golden/diff_2000.npz): both samples' found sets through checksums, the norm.fit pairs, and the four loop lists
    -- through the band-direct route the driver uses."""
    from mustache_amd.diff_mustache import _pair_tail
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.synth import synth_coo
    g = np.load(os.path.join(golden_dir, "diff_2000.npz"), allow_pickle=True)
    n, dpx, start = int(g["n"]), int(g["dpx"]), int(g["start"])
    eng = ScaleSpaceEngine(OCT)
    bands = []
    for depth, seed, chk in ((300.0, 51, 0), (260.0, 52, 1)):        # the raw synthetic maps, as the fixture was made
        x, y, v = synth_coo(n, dpx, depth=depth, seed=seed)
        assert float(v.sum()) == float(g["in_sums"][chk]) and len(v) == int(g["in_sums"][2 + chk]), "generator drifted"
        c = np.zeros((n, n))
        c[x, y] = v
        bands.append(_band_of(c, n, dpx))
    batch = eng.run_band_pairs(bands, n, dpx, [0], n)
    sig = np.asarray(eng.levels.tested_sigma)
    for b, nm in ((0, "1"), (1, "2")):
        rec = batch.found[b]
        pix = rec["pixel"].astype(np.int64)
        assert int(batch.nz_count[b]) == int(g["nz_count" + nm])
        assert len(pix) == int(g["found_count" + nm]) and int(pix.sum()) == int(g["found_pixel_sum" + nm])
        assert int(np.bitwise_xor.reduce(pix)) == int(g["found_pixel_xor" + nm])
        np.testing.assert_allclose(float(rec["value"].sum()), float(g["found_value_sum" + nm]), rtol=1e-9)
        np.testing.assert_allclose(float(sig[rec["level"].astype(int) - 1].sum()), float(g["found_sigma_sum" + nm]), rtol=1e-12)
        np.testing.assert_allclose(float(rec["q"].sum()), float(g["found_pvalue_sum" + nm]), rtol=1e-7)
        np.testing.assert_allclose(float(rec["pair"].sum()), float(g["found_pair_sum" + nm]), rtol=1e-6)
    np.testing.assert_allclose(batch.norm_fit[0, 0], g["norm_fit"][0], rtol=1e-7)
    np.testing.assert_allclose(batch.norm_fit[1, 0], g["norm_fit"][9], rtol=1e-7)
    # the driver's form: BH, selection and the partner look-ups on the device, only the selected records downloaded
    lean = eng.run_band_pairs(bands, n, dpx, [0], n, select_below=float(g["pt"]))
    for b in (0, 1):
        full, sel = batch.found[b], lean.found[b]
        keep = full["q"] < float(g["pt"])
        assert keep.sum() > 30 and np.array_equal(full["pixel"][keep], sel["pixel"])
        for key in ("level", "q", "pair", "value"):
            assert np.array_equal(full[key][keep], sel[key]), key
        other = batch.found[1 - b]
        k = np.searchsorted(other["pixel"], sel["pixel"])
        hit = (k < len(other["pixel"])) & (other["pixel"][np.minimum(k, len(other["pixel"]) - 1)] == sel["pixel"])
        assert hit.any() and (~hit).any()
        assert np.array_equal(sel["v_other"][hit], other["value"][k[hit]]) and np.isnan(sel["v_other"][~hit]).all()
    for form in (batch, lean):
        out = _pair_tail(form, 0, 1, start, float(g["pt"]), float(g["pt2"]), float(g["st"]), True)
        for got, key in zip(out, ("loops1", "diff1", "loops2", "diff2")):
            exp = g[key]
            arr = np.array([[float(a), float(b), q, s] for a, b, q, s in got]).reshape(-1, 4)
            assert arr.shape == exp.shape and len(exp) > 30, key
            assert np.array_equal(arr[:, :2], exp[:, :2]) and np.array_equal(arr[:, 3], exp[:, 3]), key
            np.testing.assert_allclose(arr[:, 2], exp[:, 2], rtol=1e-7)


def test_diff_mustache_dropin_vs_reference(golden_dir):
    from mustache_amd.diff_mustache import diff_mustache
    g = _load(golden_dir)
    c1, c2 = _blocks(g)
    n, dpx, start = int(g["n"]), int(g["dpx"]), int(g["start"])
    out = diff_mustache(c1, c2, "1", "1", 5000, start, start + n, 0, dpx, OCT, float(g["st"]), float(g["pt"]),
                        float(g["pt2"]))
    assert c1[0, 0] == 2 and c2[5, 5] == 2, "both blocks are filled in place like the reference does"
    # ... with exactly the values the device copies hold after mst_block_prologue (the fills are written on the host)
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    r1, r2 = _blocks(g)
    dev = torch.from_numpy(np.stack([r1, r2])).cuda()
    ScaleSpaceEngine(OCT).prologue(dev, dpx, True)
    assert np.array_equal(dev[0].cpu().numpy(), c1) and np.array_equal(dev[1].cpu().numpy(), c2)
    for got, key in zip(out, ("loops1", "diff1", "loops2", "diff2")):
        exp = g[key]
        arr = np.array([[float(a), float(b), q, s] for a, b, q, s in got]).reshape(-1, 4)
        assert arr.shape == exp.shape, key
        assert np.array_equal(arr[:, :2], exp[:, :2]) and np.array_equal(arr[:, 3], exp[:, 3]), key
        np.testing.assert_allclose(arr[:, 2], exp[:, 2], rtol=1e-9)
    assert len(g["loops1"]) > 5 and len(g["diff1"]) > 0 and len(g["loops2"]) > 5


def test_diff_regulator_equals_blockwise_oracle(tmp_path):
    """Two text files -> regulator(): 4 tagged lists; checked against the oracle run block by block."""
    import oracle
    from mustache_amd.diff_mustache import regulator
    from mustache_amd.synth import synth_coo
    n, dpx, res = 2300, 200, 10000
    files = []
    coos = []
    for seed in (61, 62):
        x, y, v = synth_coo(n, dpx, depth=300.0, seed=seed)
        f = str(tmp_path / ("s%d.txt" % seed))
        with open(f, "w") as fh:
            for a, b, c in zip(x, y, v):
                fh.write("%d\t%d\t%r\n" % (a * res, b * res, float(c)))
        files.append(f)
        coos.append((x, y, v))
    got = regulator(files[0], files[1], False, False, "unused", res=res, pt=0.2, pt2=0.2, st=0.8,
                    distance_filter=dpx * res, chromosome="S", verbose=False)
    # oracle: normalise each sample, same tiling, diff_block per block pair, overlap mask, tags 1..4
    for x, y, v in coos:
        oracle.normalize_sparse(x, y, v, res, dpx)
    CH, start, end = oracle.block_bounds(n, dpx)
    exp = []
    for i in range(len(start)):
        cc = [oracle.dense_block(x, y, v, start[i], end[i], CH) for x, y, v in coos]
        res4 = oracle.diff_block(cc[0], cc[1], start[i], dpx, OCT, 0.8, 0.2, 0.2)
        mask = oracle.block_mask_size(i, start, end, dpx)
        for tag, loops in enumerate(res4, start=1):
            for lp in loops:
                if lp[0] >= start[i] + mask or lp[1] >= start[i] + mask:
                    exp.append((int(lp[0]), int(lp[1]), lp[3], tag))
    assert len(exp) > 10
    assert sorted((int(r[0]), int(r[1]), r[3], r[4]) for r in got) == sorted(exp)


def test_diff_cli_two_ranks_equal_one_process(tmp_path):
    """`torchrun --nproc-per-node 2 -m mustache_amd.diff_mustache ...` (both ranks on this box's GPU, gloo group: the test
    hooks of sharding.init_from_env) writes the same four files as the single-process CLI."""
    import subprocess
    import sys
    from mustache_amd.diff_mustache import main, SUFFIX
    from mustache_amd.synth import synth_coo
    n, dpx, res = 2300, 200, 10000
    files = []
    for seed in (61, 62):
        x, y, v = synth_coo(n, dpx, depth=300.0, seed=seed)
        f = str(tmp_path / ("s%d.txt" % seed))
        with open(f, "w") as fh:
            for a, b, c in zip(x, y, v):
                fh.write("%d\t%d\t%r\n" % (a * res, b * res, float(c)))
        files.append(f)
    common = ["-f1", files[0], "-f2", files[1], "-ch", "chrS", "chrT", "chrU", "-r", "10kb", "-pt", "0.2", "-pt2", "0.2", "-st",
              "0.8", "-d", str(dpx * res)]
    single, multi = str(tmp_path / "one"), str(tmp_path / "two")
    main(common + ["-o", single])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MUSTACHE_DIST_BACKEND="gloo", MUSTACHE_ONE_DEVICE="1", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", "-m", "mustache_amd.diff_mustache"] + common + ["-o", multi]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    total = 0
    for suf in SUFFIX.values():
        a, b = open(single + suf).read(), open(multi + suf).read()
        assert a == b, suf
        total += a.count("\n") - 1
    assert total > 30


def test_pair_genome_batched_equals_chromosome_by_chromosome():
    """run_pair_genome (both samples' bands of several chromosomes side by side, all block pairs through the same launches:
    BASELINE config 5) == call_diff_loops_coo on each chromosome alone: same tagged rows in the same order."""
    from mustache_amd.diff_mustache import call_diff_loops_coo, normalized_pair_bands, run_pair_genome
    from mustache_amd.pipeline import ChromosomePipeline
    from mustache_amd.synth import synth_coo
    dpx, res = 200, 10000
    chroms = [(synth_coo(n, dpx, depth=300.0, seed=s1), synth_coo(n2, dpx, depth=300.0, seed=s2))
              for n, n2, s1, s2 in ((2300, 2300, 61, 62), (1700, 1650, 63, 64), (3100, 3100, 65, 66))]
    alone = [call_diff_loops_coo((a[0], a[1], a[2].copy()), (b[0], b[1], b[2].copy()), res, dpx, OCT, 0.8, 0.2, 0.2,
                                 verbose=False) for a, b in chroms]
    pipe = ChromosomePipeline(OCT)
    pairs = [normalized_pair_bands(pipe, (a[0], a[1], a[2].copy()), (b[0], b[1], b[2].copy()), res, dpx) for a, b in chroms]
    together = run_pair_genome(pipe, pairs, dpx, 0.8, 0.2, 0.2)
    assert sum(len(o) for o in alone) > 30
    for a, b in zip(alone, together):
        assert [(int(r[0]), int(r[1]), float(r[2]), float(r[3]), int(r[4])) for r in a] == \
               [(int(r[0]), int(r[1]), float(r[2]), float(r[3]), int(r[4])) for r in b]
    # the same with the block pairs cut into several groups: the device work of group i + 1 is queued before group i is
    # collected (engine.run_band_pairs_overlapped) -- same rows
    pipe.blocks_per_launch = lambda CH: 1
    pairs = [normalized_pair_bands(pipe, (a[0], a[1], a[2].copy()), (b[0], b[1], b[2].copy()), res, dpx) for a, b in chroms]
    piped = run_pair_genome(pipe, pairs, dpx, 0.8, 0.2, 0.2)
    for a, b in zip(alone, piped):
        assert [(int(r[0]), int(r[1]), float(r[2]), float(r[3]), int(r[4])) for r in a] == \
               [(int(r[0]), int(r[1]), float(r[2]), float(r[3]), int(r[4])) for r in b]


def test_diff_cli_from_hic_files_equals_cli_from_text(tmp_path):
    """BASELINE config 5's input form: both samples as `.hic` files through the native reader's packed / streamed path
    (`-f1 a.hic -f2 b.hic`, no -ch: every chromosome of the file) write the same four files as the same contacts given as
    text -- integer counts, NONE normalisation, so that the float32 the `.hic` path carries is exact."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hic_writer import write_hic
    from mustache_amd.diff_mustache import main, SUFFIX
    from mustache_amd.synth import synth_coo
    dpx, res = 200, 10000
    chroms = [("All", 1), ("chrS", 2300 * res), ("chrT", 1800 * res)]
    hics, txts = [], []
    for smp, seeds in enumerate(((61, 63), (62, 64))):
        mats, rows = {}, []
        for ci, (name, length) in enumerate(chroms[1:], start=1):
            x, y, v = synth_coo(length // res, dpx, depth=300.0, seed=seeds[ci - 1])
            near = (y - x) <= dpx
            x, y, c = x[near], y[near], np.round(v[near]) + 1.0
            mats[ci] = {res: (x, y, c)}
            rows.append((name, x, y, c))
        h = str(tmp_path / ("s%d.hic" % smp))
        write_hic(h, chroms, mats, {}, version=8, block_bin_count=200, float_counts=False)
        hics.append(h)
        per = []
        for name, x, y, c in rows:
            t = str(tmp_path / ("s%d_%s.txt" % (smp, name)))
            with open(t, "w") as fh:
                for a, b, cc in zip(x, y, c):
                    fh.write("%d\t%d\t%r\n" % (a * res, b * res, float(cc)))
            per.append(t)
        txts.append(per)
    common = ["-r", "10kb", "-pt", "0.2", "-pt2", "0.2", "-st", "0.8", "-d", str(dpx * res), "-norm", "NONE"]
    out_h = str(tmp_path / "from_hic")
    main(["-f1", hics[0], "-f2", hics[1], "-o", out_h] + common)
    total = 0
    for k, name in enumerate(("chrS", "chrT")):
        out_t = str(tmp_path / ("from_txt_" + name))
        main(["-f1", txts[0][k], "-f2", txts[1][k], "-ch", name, "-o", out_t] + common)
        for suf in SUFFIX.values():
            want = [l for l in open(out_t + suf).read().strip().split("\n")[1:] if l]
            got = [l for l in open(out_h + suf).read().strip().split("\n")[1:] if l.startswith(name + "\t")]
            assert got == want, (name, suf, len(got), len(want))
            total += len(want)
    assert total > 30


def test_diff_regulator_5kb_three_blocks_vs_reference(golden_dir, tmp_path):
    """The reference's own two-sample regulator() (diff_mustache.py:572-716) at BASELINE config 5's geometry -- 5 kb, distance
    limit 400 bins, three 2000 x 2000 block pairs at stride 1600 with a right-aligned last one, both samples normalised with
    their 400-bin windows -- against regulator() of this package from the SAME two text files: the tagged rows (common loops of
    each sample, differential loops of each) have identical coordinates, scales and tags, q within 1e-6."""
    import pandas as pd
    from mustache_amd.diff_mustache import regulator
    from mustache_amd.synth import synth_coo
    g = np.load(os.path.join(golden_dir, "diff_regulator_5kb_3blocks.npz"))
    n, dpx, res = int(g["n"]), int(g["dpx"]), int(g["res"])
    files = []
    for k, (seed, depth) in enumerate(zip(g["seeds"], g["depths"])):
        x, y, v = synth_coo(n, dpx, depth=float(depth), seed=int(seed))
        assert len(v) == int(g["in_sums"][2 + k]) and float(v.sum()) == float(g["in_sums"][k]), "synthetic generator drifted"
        f = str(tmp_path / ("s%d.txt" % k))
        pd.DataFrame({"a": x * res, "b": y * res, "c": v}).to_csv(f, sep="\t", header=False, index=False)
        files.append(f)
    got = regulator(files[0], files[1], False, False, "unused", res=res, pt=float(g["pt"]), pt2=float(g["pt2"]), st=float(g["st"]),
                    distance_filter=dpx * res, chromosome="S", verbose=False)
    got = np.array(sorted([float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4])] for r in got)).reshape(-1, 5)
    exp = g["rows"]
    assert got.shape == exp.shape and len(exp) > 50 and all((exp[:, 4] == t).any() for t in (1, 2, 3, 4))
    # sorted by (x, y, fdr, ...): rows that differ only in fdr digits keep their order as long as fdr agrees to 1e-6
    assert np.array_equal(got[:, [0, 1, 3, 4]], exp[:, [0, 1, 3, 4]]), "coordinates, scales and tags must match the reference exactly"
    np.testing.assert_allclose(got[:, 2], exp[:, 2], rtol=1e-6)


def test_pair_launch_equals_each_samples_own_launch():
    """mst_scale_space_band_pair (ONE fused launch over both samples' blocks, what run_band_pairs queues) == mst_scale_space_band
    on each sample alone: same records per block, same tested-pixel counts -- on overlapping blocks, where tiles are shared
    inside a sample and must not be across the split.  And the entry refuses a split inside a run of overlapping blocks."""
    import ctypes
    import torch
    from mustache_amd import _lib
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.normalize import band_from_coo
    from mustache_amd.synth import synth_coo
    n, dpx, CH = 4300, 300, 2000
    starts = [0, 1000, 2000, 2300]
    dev = torch.device("cuda:0")
    bands = []
    for seed, depth in ((71, 300.0), (72, 240.0)):
        x, y, v = synth_coo(n, dpx, depth=depth, seed=seed)
        bands.append(band_from_coo(torch.as_tensor(x, device=dev), torch.as_tensor(y, device=dev), torch.as_tensor(v, device=dev), n, dpx))
    eng = ScaleSpaceEngine(OCT)
    pair = eng.run_band_pairs(bands, n, dpx, starts, CH)
    pair_found = [{k: np.array(v) for k, v in r.items()} for r in pair.found]      # (views of staging memory that the engine's
    pair_nz = np.array(pair.nz_count)                                              #  second-next download reuses: keep copies)
    P = len(starts)
    total = 0
    for k, band in enumerate(bands):
        recs, fits, nzc = eng.sigma_loop_band(band, n, dpx, starts, CH)
        assert np.array_equal(nzc.cpu().numpy(), pair_nz[k * P:(k + 1) * P])
        for b in range(P):
            one, two = recs[b], pair_found[k * P + b]
            assert np.array_equal(one["pixel"], two["pixel"]) and np.array_equal(one["level"], two["level"])
            assert np.array_equal(one["value"], two["value"])
            total += len(one["pixel"])
    assert total > 2000
    assert eng.band_items(starts, CH, dpx, skip_empty=True)[2] > 0          # sharing was on inside each sample
    lib = _lib.load()
    st = (ctypes.c_int64 * 4)(0, 1000, 2000, 3000)
    nzc = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = lib.mst_scale_space_band_pair(bands[0].data_ptr(), bands[1].data_ptr(), 2, n, dpx, st, 4, CH, ctypes.byref(eng._lv_struct),
                                       None, 0, None, None, nzc.data_ptr(), 0, None, 0, None)
    assert rc == _lib.MST_E_ARG and b"must not continue" in lib.mst_last_error()
    # a record capacity that is far too small: the pair launch is re-run with more room (both bands again), single call and
    # pipelined groups alike
    small = ScaleSpaceEngine(OCT)
    small._found_cap[CH] = 128
    again = small.run_band_pairs(bands, n, dpx, starts, CH)
    assert small._found_cap[CH] > 128
    for b in range(2 * P):
        assert np.array_equal(again.found[b]["pixel"], pair_found[b]["pixel"]) and np.array_equal(again.found[b]["q"], pair_found[b]["q"])
    groups = [starts[:2], starts[2:]]
    want = []
    for g in groups:
        w = eng.run_band_pairs(bands, n, dpx, g, CH, select_below=0.5)
        want.append([{k: np.array(v) for k, v in r.items()} for r in w.found])
    small = ScaleSpaceEngine(OCT)
    small._found_cap[CH] = 128
    for got, ref in zip(small.run_band_pairs_overlapped(bands, n, dpx, groups, CH, select_below=0.5), want):
        assert len(got.found) == len(ref)
        for a, b in zip(got.found, ref):
            assert len(a["pixel"]) > 0 and np.array_equal(a["pixel"], b["pixel"]) and np.array_equal(a["pair"], b["pair"])
