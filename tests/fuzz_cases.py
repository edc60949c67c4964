"""Randomised parity cases shared by the GPU test slice (tests/test_gpu_fuzz.py, seeded, a few minutes) and the long sweeps
(scripts/fuzz_*.py, any seed / case count).  TEST INFRASTRUCTURE: every case runs the HIP path through the package and checks
it against the CPU oracle (oracle/), which only tests, smoke() and bench.py's CPU leg may import.

Each *_case(rng, ...) draws one case from `rng`, runs it and returns (ok, n_loops, description); ok = coordinates, scales
(and for the block case the complete found set, DoG values, levels, `loc`) identical to the oracle, p / q within the stated
tolerance."""
import os

import numpy as np

# FUZZ_OCT="3.2,6.4" / "1.6,3.2,6.4": the sweeps on the wide-radius instantiations (-sz 3.2, -oc 3)
OCT = [float(t) for t in os.environ["FUZZ_OCT"].split(",")] if os.environ.get("FUZZ_OCT") else [1.6, 3.2]


# ---- the oracle ahead of the GPU.  The CPU oracle is the slow side of every pipeline case (seconds to a minute per chromosome,
# one core); tests/test_gpu_fuzz.py first runs its cases in PLAN mode (plan = a list: the case draws its parameters, appends
# the oracle jobs it will need and returns), hands the jobs to worker processes (start_ahead) and then runs the cases for real
# with the same seed -- the same draws -- picking the results up as they are needed.  Without start_ahead (scripts/fuzz_*.py)
# the oracle runs inline, as before.
_AHEAD = {}
_POOL = None


def oracle_job(args):
    """(n, dpx, depth, seed, nloops, res, st, pt, octaves) -> the oracle's loops of that synthetic chromosome"""
    import oracle
    import torch
    from mustache_amd.synth import synth_coo
    n, dpx, depth, seed, nloops, res, st, pt, octs = args
    torch.set_num_threads(1)
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=nloops)
    return oracle.regulator_coo(x, y, v, res, dpx, list(octs), st, pt)


def _oracle_on_coo(x, y, v, res, dpx, octs, st, pt):
    """what a worker process runs: NumPy / SciPy only (no torch import: on a fresh box that alone takes a minute per process)"""
    import oracle
    return oracle.regulator_coo(x, y, v, res, dpx, list(octs), st, pt)


def start_ahead(jobs, workers=None):
    """run the oracle jobs (oracle_job arguments) in worker processes; _oracle() collects them.  The synthetic contacts are drawn
    here, in the parent (torch is loaded already), and handed over as arrays."""
    global _POOL
    import concurrent.futures as cf
    import multiprocessing as mp
    from mustache_amd.synth import synth_coo
    if _POOL is None:
        _POOL = cf.ProcessPoolExecutor(max_workers=workers or max(2, min(12, (os.cpu_count() or 4) // 2)),
                                       mp_context=mp.get_context("spawn"))
    for j in jobs:
        if j not in _AHEAD:
            n, dpx, depth, seed, nloops, res, st, pt, octs = j
            x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=nloops)
            _AHEAD[j] = _POOL.submit(_oracle_on_coo, x, y, v, res, dpx, octs, st, pt)


def stop_ahead():
    global _POOL
    if _POOL is not None:
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None
    _AHEAD.clear()


def _oracle(job):
    fut = _AHEAD.pop(job, None)
    if fut is not None:
        try:
            return fut.result()
        except Exception as e:            # a broken pool must not fail a parity test: the oracle runs inline instead
            print("fuzz_cases: oracle worker failed (%r), running inline" % (e,), flush=True)
    return oracle_job(job)


def _same_loops(got, exp, qtol=1e-6):
    got = sorted(got, key=lambda r: (int(r[0]), int(r[1])))
    exp = sorted(exp, key=lambda r: (int(r[0]), int(r[1])))
    same = [(int(a), int(b), s) for a, b, _, s in got] == [(int(a), int(b), s) for a, b, _, s in exp]
    qerr = max([abs(g[2] - e[2]) / max(e[2], 1e-300) for g, e in zip(got, exp)], default=0.0) if same else float("nan")
    return bool(same and qerr <= qtol), qerr


def parity_case(rng, eng):
    """One random dense block (edge 90-760, distance limit 30-420, dense to very sparse, sometimes with rectangular
    unmappable holes): found set / values / levels / loc identical, p 1e-9, with and without empty-tile skipping; then the
    whole block through the drop-in mustache() against the oracle's tail."""
    import torch
    import oracle
    from mustache_amd.mustache import mustache
    from mustache_amd.synth import synth_coo
    n = int(rng.integers(90, 760))
    dpx = int(rng.integers(30, max(31, min(n - 10, 420))))
    depth = float(rng.choice([0.8, 2.0, 8.0, 40.0, 300.0]))
    seed = int(rng.integers(0, 10 ** 6))
    hole = rng.random() < 0.3
    a, b = sorted(rng.integers(0, n, 2))
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.2, 0.5]))
    desc = dict(kind="block", n=n, dpx=dpx, depth=depth, seed=seed, hole=bool(hole), st=st, pt=pt)
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=max(n // 20, 1))
    if len(v) < 100:
        return True, 0, dict(desc, skipped="fewer than 100 contacts")
    oracle.normalize_sparse(x, y, v, 50000, dpx)
    c = np.zeros((n, n))
    c[x, y] = v
    if hole:
        c[a:b, :] = 0
        c[:, a:b] = 0
    ref = c.copy()
    nz = oracle.block_prologue(ref, dpx)
    if nz.sum() < 50:
        return True, 0, dict(desc, skipped="fewer than 50 tested pixels")
    ss = oracle.scale_space_levels(ref, nz, OCT, blur="scipy")
    f = ss.pval != 2
    dev = torch.from_numpy(c.copy()).cuda().unsqueeze(0)
    nzd, cnt = eng.prologue(dev, dpx, True)
    ok = True
    for skip in (True, False):
        found, fits = eng.sigma_loop(dev, nzd, cnt, skip_empty=skip)
        r = found[0]
        ok = ok and (np.array_equal(r["pixel"].astype(np.int64), np.flatnonzero(nz.ravel())[f]) and
                     np.array_equal(r["value"], ss.best[f]) and np.array_equal(r["level"].astype(np.int64), ss.level[f]) and
                     np.allclose(r["pval"], ss.pval[f], rtol=1e-9, atol=0, equal_nan=True) and
                     np.array_equal(fits[0][0], np.array([t["loc"] for t in ss.tested])))
    exp = oracle.mustache_block(c.copy(), 17, dpx, OCT, st, pt)
    got = mustache(c.copy(), "1", "1", 5000, [], 17, n + 17, 0, dpx, OCT, st, pt)
    tail_ok = ([(int(a_), int(b_), s_) for a_, b_, _, s_ in got] == [(int(a_), int(b_), s_) for a_, b_, _, s_ in exp] and
               np.allclose([q for _, _, q, _ in got], [q for _, _, q, _ in exp], rtol=1e-9))
    return bool(ok and tail_ok), len(exp), dict(desc, found=int(f.sum()), found_ok=bool(ok), tail_ok=bool(tail_ok))


def pipeline_case(rng, pipe, max_n=5200, share_modes=(True,), wide=False, plan=None):
    """One random chromosome through the whole GPU pipeline (COO -> band -> normalisation -> band-direct fused kernel -> device
    BH / selection -> batched tail -> overlap masks) against the oracle's regulator restatement.  share_modes: the values of
    engine.share_tiles to run (tiles of overlapping blocks computed once / every tile once per block) -- the oracle runs once.
    wide=True draws a distance limit of 1000-1300 px, where consecutive blocks overlap by half their edge and most tiles are
    shared (the 1 kb geometry of BASELINE config 4 in small)."""
    import oracle
    from mustache_amd.synth import synth_coo
    if wide:
        dpx = int(rng.integers(1000, 1300))
        n = int(rng.integers(3 * dpx, 3 * dpx + 900))
    else:
        dpx = int(rng.integers(60, 420))
        n = int(rng.integers(max(2 * dpx, 500), max_n))
    res = int(rng.choice([1000, 2000, 5000, 10000, 25000]))
    depth = float(rng.choice([5.0, 40.0, 300.0]))
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.1, 0.3]))
    seed = int(rng.integers(0, 10 ** 6))
    job = (n, dpx, depth, seed, max(n // 25, 4), res, st, pt, tuple(OCT))
    if plan is not None:
        plan.append(job)
        return None
    x, y, v = synth_coo(n, dpx, depth=depth, seed=seed, nloops=max(n // 25, 4))
    import os
    if os.environ.get("FUZZ_VERBOSE"):
        print("  pipeline_case:", dict(n=n, dpx=dpx, res=res, depth=depth, st=st, pt=pt, seed=seed, wide=wide), flush=True)
    exp = _oracle(job) if not os.environ.get("FUZZ_SKIP_ORACLE") else []
    ok, qerr, counts = True, 0.0, []
    keep = pipe.engine.share_tiles
    try:
        for share in share_modes:
            pipe.engine.share_tiles = share
            got = pipe.run(x, y, v.copy(), res, dpx, st, pt, distributed=False)
            s, q = _same_loops(got, exp)
            ok, qerr = ok and s, max(qerr, q if s else float("inf"))
            counts.append(len(got))
    finally:
        pipe.engine.share_tiles = keep
    return ok, len(exp), dict(kind="chromosome", n=n, dpx=dpx, res=res, depth=depth, st=st, pt=pt, seed=seed,
                              branch="A" if (n - dpx) * res > 2e6 else "B", got=counts, exp=len(exp), qerr=qerr,
                              share_modes=list(share_modes))


def genome_case(rng, pipe, max_n=5200, max_chroms=5, plan=None):
    """A random set of 2-max_chroms chromosomes (from shorter than one block up) through run_genome -- side by side in one
    band, a random number of blocks per launch so that blocks of several chromosomes share launches -- against the oracle's
    regulator restatement per chromosome."""
    import oracle
    from mustache_amd.synth import synth_coo
    dpx = int(rng.integers(60, 420))
    res = int(rng.choice([1000, 2000, 5000, 10000]))
    st, pt = float(rng.choice([0.5, 0.7, 0.88])), float(rng.choice([0.05, 0.1, 0.3]))
    per = int(rng.integers(1, 9))
    k = int(rng.integers(2, max_chroms + 1))
    coos, jobs = [], []
    for c in range(k):
        n = int(rng.integers(max(dpx + 50, 300), max_n))
        depth = float(rng.choice([5.0, 40.0, 300.0]))
        seed = int(rng.integers(0, 10 ** 6))
        jobs.append((n, dpx, depth, seed, max(n // 25, 4), res, st, pt, tuple(OCT)))
        if plan is None:
            coos.append(synth_coo(n, dpx, depth=depth, seed=seed, nloops=max(n // 25, 4)))
    if plan is not None:
        plan.extend(jobs)
        return None
    keep = pipe.__dict__.get("blocks_per_launch")
    pipe.blocks_per_launch = lambda CH, k_=per: k_
    try:
        bands, ns = zip(*[pipe.normalized_band(x, y, v.copy(), res, dpx) for x, y, v in coos])
        got_all = pipe.run_genome(list(bands), list(ns), dpx, st, pt)
    finally:
        if keep is None:
            pipe.__dict__.pop("blocks_per_launch", None)
        else:
            pipe.blocks_per_launch = keep
    ok, total, bad = True, 0, []
    for c, ((x, y, v), got) in enumerate(zip(coos, got_all)):
        exp = _oracle(jobs[c])
        s, q = _same_loops(got, exp)
        total += len(exp)
        if not s:
            ok = False
            bad.append(dict(chromosome=c, n=ns[c], got=len(got), exp=len(exp), qerr=q))
    return ok, total, dict(kind="genome", chromosomes=k, ns=list(ns), dpx=dpx, res=res, st=st, pt=pt, blocks_per_launch=per,
                           bad=bad)


def diff_case(rng, eng):
    """One random block pair through the two-sample path: the band-direct route in both forms (selected records only + device
    look-ups = what the driver runs; whole found sets) and the dense drop-in diff_mustache(), against the oracle's
    restatement of the reference's diff_mustache(): the four loop lists identical in coordinates and scales, q within 1e-6."""
    import torch
    import oracle
    from mustache_amd.diff_mustache import _pair_tail, diff_mustache
    from mustache_amd.normalize import band_from_coo
    from mustache_amd.synth import synth_coo

    def key(lists):
        return [[(int(a), int(b), float(s)) for a, b, _, s in l] for l in lists]

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
    batch = eng.run_band_pairs(bands, n, dpx, [0], n, select_below=pt)
    got_band = _pair_tail(batch, 0, 1, start, pt, pt2, st, True)
    full = _pair_tail(eng.run_band_pairs(bands, n, dpx, [0], n), 0, 1, start, pt, pt2, st, True)
    forms_agree = [[tuple(l) for l in ls] for ls in full] == [[tuple(l) for l in ls] for ls in got_band]
    got_dense = diff_mustache(cs[0].copy(), cs[1].copy(), "1", "1", 5000, start, start + n, 0, dpx, OCT, st, pt, pt2)
    ok = key(got_band) == key(exp) and key(got_dense) == key(exp)
    qe = 0.0
    if ok:
        for g, e in zip(got_band, exp):
            for a, b in zip(g, e):
                qe = max(qe, abs(a[2] - b[2]) / max(b[2], 1e-300))
    return bool(ok and forms_agree and qe <= 1e-6), sum(len(l) for l in exp), dict(
        kind="block pair", n=n, dpx=dpx, st=st, pt=pt, pt2=pt2, lists=[len(l) for l in exp], forms_agree=forms_agree, qerr=qe)


def geometry_case(pipe, n, dpx, res, depth, st=0.88, pt=0.1, share_modes=(True,), plan=None):
    """A fixed chromosome at a geometry beyond BASELINE's through the whole pipeline against the oracle's regulator."""
    from mustache_amd.synth import synth_coo
    job = (n, dpx, depth, n, n // 25, res, st, pt, tuple(OCT))
    if plan is not None:
        plan.append(job)
        return None
    x, y, v = synth_coo(n, dpx, depth=depth, seed=n, nloops=n // 25)
    exp = _oracle(job)
    ok, qerr = True, 0.0
    keep = pipe.engine.share_tiles
    try:
        for share in share_modes:
            pipe.engine.share_tiles = share
            s, q = _same_loops(pipe.run(x, y, v.copy(), res, dpx, st, pt, distributed=False), exp)
            ok, qerr = ok and s, max(qerr, q if s else float("inf"))
    finally:
        pipe.engine.share_tiles = keep
    return ok, len(exp), dict(kind="geometry", n=n, dpx=dpx, res=res, depth=depth, nnz=len(v), exp=len(exp), qerr=qerr)


def hic_rows_case(rng, path, dev):
    """One random `.hic` file (versions 8-9, row lists / dense grids, short / long coordinates, integer / float counts, block
    sizes, NaN norm entries, random slab sizes and thread counts, with and without a caller's chromosome size): the band of the
    raw streamed read (rows decoded on the device, mst_band_scatter_hic_rows) must equal the band of the packed streamed read
    (host decoder) bit for bit, with the same n and record count.  Returns the record count; asserts on a mismatch."""
    import os
    import torch
    from hic_writer import write_hic
    from mustache_amd.hicfile import HicFile
    from mustache_amd.normalize import band_from_packed, read_hic_stream_to_device
    n = int(rng.integers(300, 5000))
    res = int(rng.choice([1000, 5000, 25000]))
    spread = int(rng.integers(20, 600))
    dpx = int(rng.integers(10, spread + 20))
    m = int(rng.integers(500, 150000))
    x = rng.integers(0, n, m)
    y = np.minimum(x + rng.integers(0, spread, m), n - 1 - int(rng.integers(0, 3)))
    key = np.unique(np.minimum(x, y) * 1000003 + np.maximum(x, y))
    x, y = key // 1000003, key % 1000003
    floats = bool(rng.integers(2))
    c = rng.uniform(0.25, 40, len(x)).astype(np.float32).astype(np.float64) if floats else rng.integers(1, 900, len(x)).astype(np.float64)
    version = int(rng.choice([8, 9]))
    norm = rng.uniform(0.5, 2.0, n + 1)
    norm[rng.integers(0, n, 4)] = np.nan
    write_hic(path, [("All", 7500), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): norm}, version=version,
              block_bin_count=int(rng.choice([16, 64, 128, 500])), float_counts=floats, dense_blocks=bool(rng.integers(4) == 0),
              short_coords=bool(rng.integers(2)) if version == 9 else True)
    size_bp = 0 if rng.integers(2) else int((n - rng.integers(1, 40)) * res)
    norm_name = "KR" if rng.integers(3) else "NONE"
    slab = int(rng.choice([410, 1000, 20000, 1 << 19]))
    with HicFile(path) as h:
        a = read_hic_stream_to_device(h, "chr1", res, norm_name, dpx, size_bp, dev, threads=int(rng.integers(1, 6)), slab_records=slab, raw=True)
        b = read_hic_stream_to_device(h, "chr1", res, norm_name, dpx, size_bp, dev, threads=3, slab_records=4096, raw=False)
    assert a.device_band is not None and b.device_parts is not None
    ba, bb = band_from_packed(a, dpx, dev), band_from_packed(b, dpx, dev)
    assert a.count == b.count and ba.shape == bb.shape and torch.equal(ba, bb), (n, dpx, version, floats, slab)
    os.remove(path)
    return a.count

