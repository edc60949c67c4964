"""Per-block tail after the sigma loop: FDR, candidate selection, sparsity / diagonal-mean filters, clustering.

Mirrors reference mustache/mustache.py:774-850 on the compacted "found" records the GPU returns (a few thousand
pixels per block), so no dense CH x CH array is ever built on the host.  The window densities, c[x, y] and the
diagonals come from small device gathers (BlockBatch.candidate_features / .diagonals).
"""
import math

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
    scipy.ndimage.label(structure=ones((3,3))) produces (mustache.py:840-841).  Returns (labels, n)."""
    order = np.lexsort((py, px))
    px, py = px[order], py[order]
    index = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(px, py))}
    parent = list(range(len(px)))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for i, (a, b) in enumerate(zip(px, py)):
        a, b = int(a), int(b)
        for nb in ((a, b - 1), (a - 1, b - 1), (a - 1, b), (a - 1, b + 1)):
            j = index.get(nb)
            if j is not None:
                ri, rj = find(i), find(j)
                if ri != rj:
                    if ri < rj:
                        parent[rj] = ri
                    else:
                        parent[ri] = rj
    roots = np.array([find(i) for i in range(len(px))])
    uniq, labels = np.unique(roots, return_inverse=True)   # roots are minimal raster indices -> raster order
    return order, labels, len(uniq)


def fdr_candidates(batch, b, pt, st):
    """BH over the found p-values of block b, selection q < pt and the sparsity filter (mustache.py:778-811).
    Returns (q, idx): q per record, idx = indices (into the block's record arrays) of the surviving candidates."""
    CH = batch.CH
    if not pt <= 1:
        raise ValueError("pThreshold must be <= 1")
    rec = batch.found[b]
    sigma_t = np.asarray(batch.engine.levels.tested_sigma)
    q = rec["q"] if "q" in rec else benjamini_hochberg(rec["pval"])   # (:778-779) BH ran on the device when "q" is present
    sel = np.nonzero(q < pt)[0]                     # (:789-797)  o < pt can only hold at found pixels (pt <= 1)
    if sel.size == 0:
        return q, sel
    pix = rec["pixel"][sel]
    x = (pix // CH).astype(np.int64)
    scale = sigma_t[rec["level"][sel].astype(np.int64) - 1]
    half = np.ceil(scale).astype(np.int64)          # s = math.ceil(xyScales[i])  (:802)
    cnt1, cnt2, _ = batch.candidate_features(b, pix, half)
    c1 = cnt1 / ((2 * half + 1) ** 2)               # (:803-804)
    c2 = cnt2 / ((4 * half + 1) ** 2)               # (:805-807)
    keep = (x != 0) & ~((c1 < st) | (c2 < 0.6))     # (:800, :808)
    return q, sel[keep]


def diag_mean_filter(batch, b, idx):
    """c[x, y] > 2 * mean(non-zero entries of that diagonal of the filled block)   (mustache.py:816-828)."""
    CH = batch.CH
    if idx.size == 0:
        return idx
    pix = batch.found[b]["pixel"][idx]
    x = (pix // CH).astype(np.int64)
    y = (pix % CH).astype(np.int64)
    _, _, cval = batch.candidate_features(b, pix, np.zeros(len(pix), np.int64))
    ks, inv = np.unique(y - x, return_inverse=True)
    diags = batch.diagonals(b, ks)
    means = np.empty(len(ks))
    for i, k in enumerate(ks):
        dg = diags[i, :CH - int(k)]
        means[i] = np.mean(dg[dg != 0])
    return idx[cval > 2 * means[inv]]


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
    reps = []
    for lb in range(ncomp):
        mem = np.nonzero(labels == lb)[0]           # raster order (hx, hy sorted)
        i = mem[np.argmin(o[mem])]
        if not hit[i]:
            raise AssertionError("cluster without a found pixel")
        reps.append(int(pos_c[i]))
    return reps


def loops_from_reps(batch, b, q, reps, start):
    CH = batch.CH
    rec = batch.found[b]
    sigma_t = np.asarray(batch.engine.levels.tested_sigma)
    out = []
    for r in reps:
        px = int(rec["pixel"][r])
        out.append([np.int64(px // CH + start), np.int64(px % CH + start), np.float64(q[r]),
                    np.float64(sigma_t[int(rec["level"][r]) - 1])])
    return out


def block_tail(batch, b, start, pt, st, intra=True):
    """Loops of block b as the reference's list of [x+start, y+start, fdr, sigma] (mustache.py:848)."""
    if batch.nz_count[b] < 50:                      # (:701)
        return []
    if batch.nz_count[b] < 10000:                   # (:775)  len(pFound) counts the nz pixels
        return []
    q, idx = fdr_candidates(batch, b, pt, st)
    if idx.size == 0:                               # (:813)
        return []
    if intra:                                       # (:822-828)
        idx = diag_mean_filter(batch, b, idx)
        if idx.size == 0:
            return []
    reps = cluster_representatives(batch, b, q, idx)
    return loops_from_reps(batch, b, q, reps, start)


CH_KEY = 1 << 20   # pixel-key radix for the halo set (coordinates can reach CH, one past the block edge)
