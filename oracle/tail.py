"""Oracle: FDR, candidate selection, sparsity / diagonal-mean filters, clustering (mustache.py:774-850).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import math

import numpy as np


def benjamini_hochberg(p):
    """`multipletests(p, method='fdr_bh')[1]` (mustache.py:778).

    statsmodels is a third-party dependency that is absent here; this restates its published algorithm:
    sort ascending, divide by the empirical CDF rank/m, cumulative minimum from the right, clip at 1,
    scatter back to the input order.
    """
    p = np.asarray(p, dtype=np.float64)
    m = p.size
    if m == 0:
        return p.copy()
    order = np.argsort(p)
    ranked = p[order] / (np.arange(1, m + 1) / float(m))
    ranked = np.minimum.accumulate(ranked[::-1])[::-1]
    ranked[ranked > 1] = 1
    q = np.empty_like(ranked)
    q[order] = ranked
    return q


def _window_density(nz, x, y, s):
    """sum(nz[x-s:x+s+1, y-s:y+s+1]) / (2s+1)^2 with Python slice semantics (mustache.py:803-807):
    a negative start wraps around and yields an empty window (density 0); a stop past the edge truncates."""
    return np.sum(nz[x - s:x + s + 1, y - s:y + s + 1]) / ((2 * s + 1) ** 2)


def block_tail(c, nz, pval, scale, start, pt, st, intra=True, min_nz=10000):
    """Rows 8-9 of SURVEY.md section 8a.  ``c`` is the prologue'd (filled) block, ``pval``/``scale`` are the
    per-nz arrays pAll / Scales after the sigma loop.  Returns the reference's list of
    [x+start, y+start, fdr, sigma] (mustache.py:848)."""
    pval = pval.copy()
    found = pval != 2
    if len(found) < min_nz:                       # counts nz pixels, not found pixels (:775)
        return []
    pval[found] = benjamini_hochberg(pval[found])  # (:778-779)

    o = np.ones_like(c)
    o[nz] = pval                                  # (:789-790) not-found nz stay 2
    so = np.ones_like(c)
    so[nz] = scale                                # (:793-794)
    x, y = np.nonzero(o < pt)                     # the set the argsort at :791-797 selects
    if x.size:
        srt = np.argsort(o[x, y], kind="stable")
        x, y = x[srt], y[srt]

    keep = x != 0                                 # (:800)
    for i in range(x.size):
        s = math.ceil(so[x[i], y[i]])
        c1 = _window_density(nz, x[i], y[i], s)
        c2 = _window_density(nz, x[i], y[i], 2 * s)
        if c1 < st or c2 < 0.6:                   # (:808)
            keep[i] = False
    x, y = x[keep], y[keep]
    if x.size == 0:                               # (:813)
        return []

    if intra:                                     # (:822-828)
        n = c.shape[0]
        means = np.empty(x.size)
        for i in range(x.size):
            k = int(y[i] - x[i])
            diag = c[np.arange(0, n - k), np.arange(k, n)] if k >= 0 else c[np.arange(-k, n), np.arange(0, n + k)]
            means[i] = np.mean(diag[diag != 0])
        ok = c[x, y] > 2 * means
        if ok.size == 0 or ok.sum() == 0:
            return []
        x, y = x[ok], y[ok]

    # clustering (:830-841): candidates and their 8 neighbours, 8-connected components
    from scipy.ndimage import label
    size = int(np.max(y)) + 2
    lab = np.zeros((size, size), dtype=np.float32)
    lab[x, y] = o[x, y] + 1
    for dx, dy in ((1, 0), (1, 1), (0, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (-1, 1)):
        lab[x + dx, y + dy] = 2
    lab_i, nfeat = label(lab, structure=np.ones((3, 3)))

    # (:843-848) per label: argwhere(label_matrix == label) -- the label's pixels in row-major order, halo included -- and the
    # first minimum of o over them.  The reference scans the whole matrix once per label (minutes for 10^4 labels); here the
    # labelled pixels are listed ONCE in row-major order and grouped by label with a stable sort, which leaves every label's
    # pixels in exactly the order its own argwhere would give: the same argmin, the same representative.
    rr, cc = np.nonzero(lab_i)
    labs = lab_i[rr, cc]
    order = np.argsort(labs, kind="stable")
    rr, cc, labs = rr[order], cc[order], labs[order]
    bounds = np.searchsorted(labs, np.arange(1, nfeat + 2))
    out = []
    for lb in range(1, nfeat + 1):
        a, b = bounds[lb - 1], bounds[lb]
        i = a + np.argmin(o[rr[a:b], cc[a:b]])
        _x, _y = rr[i], cc[i]
        out.append([_x + start, _y + start, o[_x, _y], so[_x, _y]])
    return out
