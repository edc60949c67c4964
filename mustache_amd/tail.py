"""Per-block tail after the sigma loop: FDR, candidate selection, sparsity / diagonal-mean filters, clustering.

Mirrors reference mustache/mustache.py:774-850 on the compacted "found" records the GPU returns (a few thousand
pixels per block), so no dense CH x CH array is ever built on the host.  The window densities, c[x, y] and the
diagonals come from small device gathers (BlockBatch.candidate_features / .diagonals).
"""
import numpy as np


def benjamini_hochberg(p):
    """statsmodels.stats.multitest.multipletests(p, method='fdr_bh')[1]  (mustache.py:778): sort, p*m/rank,
    cumulative minimum from the right, clip at 1, back to input order.  Ties receive equal q-values, so the result
    per pixel does not depend on the input order."""
    p = np.asarray(p, dtype=np.float64)
    m = p.size
    if m == 0:
        return p.copy()
    order = np.argsort(p)
    adj = p[order] / (np.arange(1, m + 1) / float(m))
    adj = np.minimum.accumulate(adj[::-1])[::-1]
    adj[adj > 1] = 1
    q = np.empty_like(adj)
    q[order] = adj
    return q


def _components(px, py):
    """8-connected components of a pixel set, numbered in raster order of their first pixel -- the numbering
    scipy.ndimage.label(structure=ones((3,3))) produces (mustache.py:840-841).  Returns (order, labels, n) with
    labels given for the pixels in raster order (px[order], py[order]).

    Min-label propagation with pointer jumping over the (at most four) backward neighbour links of every pixel; a
    component's label converges to the raster index of its first pixel, so numbering the distinct labels in
    ascending order is the raster order of first pixels."""
    key = px.astype(np.int64) * CH_KEY + py.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key = key[order]
    n = len(key)
    src, dst = [], []
    for off in (-1, -CH_KEY - 1, -CH_KEY, -CH_KEY + 1):          # (a, b-1), (a-1, b-1), (a-1, b), (a-1, b+1)
        nk = key + off
        pos = np.minimum(np.searchsorted(key, nk), n - 1)
        hit = np.nonzero(key[pos] == nk)[0]
        src.append(hit)
        dst.append(pos[hit])
    src = np.concatenate(src)
    dst = np.concatenate(dst)
    lab = np.arange(n)
    while True:
        new = lab.copy()
        np.minimum.at(new, src, lab[dst])
        np.minimum.at(new, dst, lab[src])
        new = new[new]
        if np.array_equal(new, lab):
            break
        lab = new
    uniq, labels = np.unique(lab, return_inverse=True)      # labels are minimal raster indices -> raster order
    return order, labels, len(uniq)


def _features_multi(batch, bs, pixs, halfs):
    """candidate_features for several blocks; one upload / download when the batch offers it."""
    f = getattr(batch, "candidate_features_multi", None)
    if f is not None:
        return f(bs, pixs, halfs)
    return [batch.candidate_features(b, p, h) for b, p, h in zip(bs, pixs, halfs)]


def _diagonal_means_multi(batch, bs, kss):
    """mean of the non-zero entries of each requested diagonal of each block (mustache.py:816-820)."""
    f = getattr(batch, "diagonal_means_multi", None)
    if f is not None:
        return f(bs, kss)
    CH = batch.CH                                   # batches that only expose whole diagonals (host-side test doubles)
    out = []
    for b, ks in zip(bs, kss):
        dg_all = batch.diagonals(b, ks)
        means = np.empty(len(ks))
        for i, k in enumerate(ks):
            dg = dg_all[i, :CH - int(k)]
            means[i] = np.mean(dg[dg != 0])
        out.append(means)
    return out


def fdr_candidates_multi(batch, bs, pt, st):
    """BH over the found p-values of each block in bs, selection q < pt and the sparsity filter (mustache.py:778-811).
    Returns [(q, idx, cval)]: q per record, idx = indices (into the block's record arrays) of the surviving candidates,
    cval = c[x, y] of the filled block at those candidates (what the diagonal-mean filter compares, :824)."""
    CH = batch.CH
    if not pt <= 1:
        raise ValueError("pThreshold must be <= 1")
    sigma_t = np.asarray(batch.engine.levels.tested_sigma)
    qs, sels, pixs, halfs = [], [], [], []
    for b in bs:
        rec = batch.found[b]
        q = rec["q"] if "q" in rec else benjamini_hochberg(rec["pval"])   # (:778-779) BH ran on the device when "q" is present
        sel = np.nonzero(q < pt)[0]                     # (:789-797)  o < pt can only hold at found pixels (pt <= 1)
        scale = sigma_t[rec["level"][sel].astype(np.int64) - 1]
        qs.append(q)
        sels.append(sel)
        pixs.append(rec["pixel"][sel])
        halfs.append(np.ceil(scale).astype(np.int64))   # s = math.ceil(xyScales[i])  (:802)
    feats = _features_multi(batch, bs, pixs, halfs)
    out = []
    for q, sel, pix, half, (cnt1, cnt2, cval) in zip(qs, sels, pixs, halfs, feats):
        if sel.size == 0:
            out.append((q, sel, np.zeros(0)))
            continue
        x = (pix // CH).astype(np.int64)
        c1 = cnt1 / ((2 * half + 1) ** 2)               # (:803-804)
        c2 = cnt2 / ((4 * half + 1) ** 2)               # (:805-807)
        keep = (x != 0) & ~((c1 < st) | (c2 < 0.6))     # (:800, :808)
        out.append((q, sel[keep], cval[keep]))
    return out


def fdr_candidates(batch, b, pt, st):
    q, idx, _ = fdr_candidates_multi(batch, [b], pt, st)[0]
    return q, idx


def diag_mean_filter_multi(batch, bs, idxs, cvals=None):
    """c[x, y] > 2 * mean(non-zero entries of that diagonal of the filled block)   (mustache.py:816-828)."""
    CH = batch.CH
    if cvals is None:
        pixs = [batch.found[b]["pixel"][idx] for b, idx in zip(bs, idxs)]
        feats = _features_multi(batch, bs, pixs, [np.zeros(len(p), np.int64) for p in pixs])
        cvals = [f[2] for f in feats]
    kss, invs = [], []
    for b, idx in zip(bs, idxs):
        pix = batch.found[b]["pixel"][idx]
        x = (pix // CH).astype(np.int64)
        y = (pix % CH).astype(np.int64)
        ks, inv = np.unique(y - x, return_inverse=True)
        kss.append(ks)
        invs.append(inv)
    means = _diagonal_means_multi(batch, bs, kss)
    return [idx[cval > 2 * mean[inv]] if idx.size else idx for idx, cval, inv, mean in zip(idxs, cvals, invs, means)]


def diag_mean_filter(batch, b, idx):
    return diag_mean_filter_multi(batch, [b], [idx])[0]


def cluster_representatives(batch, b, q, idx):
    """Clustering (mustache.py:830-848): candidates + their 8 neighbours, 8-connected components in raster order,
    representative = first arg-min of o over ALL member pixels.  Returns record indices of the representatives."""
    CH = batch.CH
    rec = batch.found[b]
    pix = rec["pixel"][idx]
    x = (pix // CH).astype(np.int64)
    y = (pix % CH).astype(np.int64)
    offs = np.array([(0, 0), (1, 0), (1, 1), (0, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (-1, 1)])
    hx = (x[:, None] + offs[None, :, 0]).ravel()
    hy = (y[:, None] + offs[None, :, 1]).ravel()
    hp = np.unique(hx * CH_KEY + hy)
    hx, hy = hp // CH_KEY, hp % CH_KEY
    order, labels, ncomp = _components(hx, hy)
    hx, hy = hx[order], hy[order]
    # o at the member pixels: q where the pixel was found, otherwise >= 1 (1 off-nz, 2 on-nz) -- never the minimum
    fpix = rec["pixel"].astype(np.int64)
    mp = hx * CH + hy
    inside = (hx < CH) & (hy < CH)
    pos = np.searchsorted(fpix, mp)
    pos_c = np.minimum(pos, len(fpix) - 1)
    hit = inside & (fpix[pos_c] == mp)
    o = np.where(hit, q[pos_c], 1.5)
    # per label the FIRST arg-min of o in raster order: stable sort by (label, o), first entry of each label
    by = np.lexsort((o, labels))
    first = by[np.searchsorted(labels[by], np.arange(ncomp))]
    if not hit[first].all():
        raise AssertionError("cluster without a found pixel")
    return [int(r) for r in pos_c[first]]


def loops_from_reps(batch, b, q, reps, start):
    """The reference's output rows [x + start, y + start, fdr, sigma] (np.int64, np.int64, np.float64, np.float64;
    mustache.py:848) for the representatives `reps` (record indices) of block b."""
    if len(reps) == 0:
        return []
    CH = batch.CH
    rec = batch.found[b]
    r = np.asarray(reps, dtype=np.int64)
    px = rec["pixel"][r].astype(np.int64)
    sigma_t = np.asarray(batch.engine.levels.tested_sigma, dtype=np.float64)
    xs = px // CH + np.int64(start)
    ys = px % CH + np.int64(start)
    qs = np.asarray(q, dtype=np.float64)[r]
    sg = sigma_t[rec["level"][r].astype(np.int64) - 1]
    return [[a, b_, c, d] for a, b_, c, d in zip(xs, ys, qs, sg)]


def batch_tail(batch, bs, starts, pt, st, intra=True):
    """Loops of the blocks bs as the reference's lists of [x+start, y+start, fdr, sigma] (mustache.py:848), one list per
    block.  The stages run block-batched so the device gathers cost one round trip per stage, not per block."""
    out = [[] for _ in bs]
    live = [j for j, b in enumerate(bs)
            if batch.nz_count[b] >= 50 and batch.nz_count[b] >= 10000]      # (:701), (:775) len(pFound) counts the nz pixels
    if not live:
        return out
    cand = fdr_candidates_multi(batch, [bs[j] for j in live], pt, st)
    keep = [(j, q, idx, cval) for j, (q, idx, cval) in zip(live, cand) if idx.size]        # (:813)
    if intra and keep:                                                       # (:822-828)
        idxs = diag_mean_filter_multi(batch, [bs[j] for j, _, _, _ in keep], [k[2] for k in keep], [k[3] for k in keep])
        keep = [(j, q, idx, None) for (j, q, _, _), idx in zip(keep, idxs) if idx.size]
    multi = getattr(batch, "cluster_representatives_multi", None)
    if multi is not None and keep:              # device clustering, all blocks in one launch (mst_cluster_representatives)
        reps_all = multi([bs[j] for j, _, _, _ in keep], [k[1] for k in keep], [k[2] for k in keep], pt)
    else:                                       # host form (test doubles without a device; the cross-check of the kernel)
        reps_all = [cluster_representatives(batch, bs[j], q, idx) for j, q, idx, _ in keep]
    for (j, q, idx, _), reps in zip(keep, reps_all):
        out[j] = loops_from_reps(batch, bs[j], q, reps, starts[j])
    return out


def block_tail(batch, b, start, pt, st, intra=True):
    """Loops of block b as the reference's list of [x+start, y+start, fdr, sigma] (mustache.py:848)."""
    return batch_tail(batch, [b], [start], pt, st, intra)[0]


CH_KEY = 1 << 20   # pixel-key radix for the halo set (coordinates can reach CH, one past the block edge)
