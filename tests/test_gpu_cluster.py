"""mst_cluster_representatives (device clustering, mustache.py:830-848) against the host restatement in tail.py -- itself
held to scipy.ndimage.label's numbering (tests/test_host_logic.py) and to the reference's loop fixtures -- on crafted and
random record sets: touching / overlapping halos, chains, candidates at the block's edges, q ties, selected records that are
not candidates but win a component's arg-min (a reference quirk: the filters drop candidates, not their q)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(CH, founds):
    import torch
    from mustache_amd.engine import ScaleSpaceEngine, _MultiGather

    class Recs(_MultiGather):
        pass
    b = Recs()
    b.engine = ScaleSpaceEngine((1.6, 3.2))
    b.CH = CH
    b.found = founds
    b._device = lambda: b.engine.device
    return b


def _case(rng, CH, n_sel, spread, frac_cand, quantum, pt=0.1):
    # selected records: clumps, so that components of many shapes and sizes appear
    cx = rng.integers(1, CH - 1, max(1, n_sel // 6))
    cy = rng.integers(1, CH - 1, max(1, n_sel // 6))
    k = rng.integers(0, len(cx), n_sel)
    x = np.clip(cx[k] + rng.integers(-spread, spread + 1, n_sel), 1, CH - 1)
    y = np.clip(cy[k] + rng.integers(-spread, spread + 1, n_sel), 0, CH - 1)
    pix = np.unique(x.astype(np.int64) * CH + y)
    q = rng.uniform(0.0, pt, len(pix))
    if quantum:
        q = np.floor(q / quantum) * quantum              # plateaus of equal q, as BH produces
    extra = np.unique(rng.integers(CH, CH * CH, 50))     # found-but-not-selected records (q >= pt) in between
    extra = extra[~np.isin(extra, pix)]
    allpix = np.concatenate([pix, extra])
    allq = np.concatenate([q, rng.uniform(pt, 1.0, len(extra))])
    o = np.argsort(allpix)
    allpix, allq = allpix[o], allq[o]
    sel = np.nonzero(allq < pt)[0]
    cand = np.sort(rng.choice(sel, max(1, int(len(sel) * frac_cand)), replace=False))
    rec = {"pixel": allpix.astype(np.uint32), "level": np.ones(len(allpix), np.uint32), "q": allq}
    return rec, allq, cand


def test_device_clustering_equals_host_clustering():
    from mustache_amd.tail import cluster_representatives
    rng = np.random.default_rng(5)
    cases = []
    for CH, n_sel, spread, frac, quantum in ((300, 400, 3, 0.5, 0.0), (300, 900, 6, 0.3, 0.01), (2000, 3000, 4, 0.6, 0.02),
                                             (64, 500, 2, 0.7, 0.05), (4000, 6000, 9, 0.4, 0.0), (120, 40, 1, 1.0, 0.0),
                                             (500, 2500, 12, 0.9, 0.005)):
        for rep in range(3):
            cases.append((CH,) + _case(rng, CH, n_sel, spread, frac, quantum))
    total = 0
    for CH in sorted({c[0] for c in cases}):
        group = [c for c in cases if c[0] == CH]
        batch = _batch(CH, [c[1] for c in group])
        bs = list(range(len(group)))
        got = batch.cluster_representatives_multi(bs, [c[2] for c in group], [c[3] for c in group], 0.1)
        for b, (_, rec, q, cand) in enumerate(group):
            exp = cluster_representatives(batch, b, q, cand)
            assert got[b] == exp, (CH, b)
            total += len(exp)
    assert total > 500


def test_device_clustering_crafted_cases():
    """Two candidates 3 apart share a component, 4 apart do not; a non-candidate selected record next to a candidate takes the
    arg-min; equal q -> the raster-first member; an empty block in the middle of the batch."""
    from mustache_amd.tail import cluster_representatives
    CH = 100

    def rec_of(pixels, qs):
        o = np.argsort(pixels)
        return {"pixel": np.asarray(pixels, np.uint32)[o], "level": np.ones(len(pixels), np.uint32), "q": np.asarray(qs, float)[o]}
    P = lambda x, y: x * CH + y
    blocks = [
        (rec_of([P(10, 20), P(10, 23), P(10, 27), P(50, 60)], [0.05, 0.04, 0.03, 0.02]), [0, 1, 2, 3]),
        (rec_of([P(30, 40), P(31, 41)], [0.05, 0.01]), [0]),                     # the neighbour (not a candidate) wins
        (rec_of([P(5, 9)], [0.05]), []),                                          # no candidate at all
        (rec_of([P(70, 80), P(70, 81), P(71, 80)], [0.02, 0.02, 0.02]), [1, 2]),  # ties: raster-first member = P(70, 80)
        (rec_of([P(1, 5), P(98, 99), P(99, 99)], [0.03, 0.02, 0.01]), [0, 1]),    # block edges
    ]
    batch = _batch(CH, [b[0] for b in blocks])
    qs = [b[0]["q"] for b in blocks]
    idxs = [np.asarray(b[1], dtype=np.int64) for b in blocks]
    got = batch.cluster_representatives_multi(list(range(len(blocks))), qs, idxs, 0.1)
    exp = [cluster_representatives(batch, b, qs[b], idxs[b]) if len(idxs[b]) else [] for b in range(len(blocks))]
    assert got == exp
    assert got[0] == [1, 2, 3] and got[1] == [1] and got[2] == [] and got[3] == [0] and got[4] == [0, 2]
