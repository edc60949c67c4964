"""Oracle: two-sample differential loop calling, one block pair (reference mustache/diff_mustache.py:260-569).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import math

import numpy as np
import scipy.special as _sc

from .scale_space import level_table, blur_scipy, blur_explicit, maxfilter3_zero, maxfilter3_scipy, expon_pvalue
from .tail import benjamini_hochberg, _window_density


def _two_sided_normal(x, loc, scale):
    """2 * min(cdf, 1 - cdf) of N(loc, scale) with the reference's nan handling (diff_mustache.py:372-385)."""
    cdf = _sc.ndtr((x - loc) / scale)
    np.nan_to_num(cdf, copy=False, posinf=1, neginf=1, nan=1)
    hi = cdf > 0.5
    cdf[hi] = 1 - cdf[hi]
    return cdf * 2


def _filters(c, nz, o, so, pt, st, intra):
    """selection (row-major np.where, :458/:473) + sparsity filter (:479-505) for one sample"""
    x, y = np.where(o < pt)
    keep = x != 0
    for i in range(x.size):
        s = math.ceil(so[x[i], y[i]])
        if _window_density(nz, x[i], y[i], s) < st or _window_density(nz, x[i], y[i], 2 * s) < 0.6:
            keep[i] = False
    return x[keep], y[keep]


def _diag_filter(c, x, y):
    n = c.shape[0]
    means = np.empty(x.size)
    for i in range(x.size):
        k = int(y[i] - x[i])
        d = c[np.arange(0, n - k), np.arange(k, n)]
        means[i] = np.mean(d[d != 0])
    return c[x, y] > 2 * means


def _clusters(x, y, o, so, start):
    from scipy.ndimage import label
    size = int(np.max(y)) + 2
    lab = np.zeros((size, size), dtype=np.float32)
    lab[x, y] = o[x, y] + 1
    for dx, dy in ((1, 0), (1, 1), (0, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (-1, 1)):
        lab[x + dx, y + dy] = 2
    lab_i, nfeat = label(lab, structure=np.ones((3, 3)))
    out = []
    for lb in range(1, nfeat + 1):
        idx = np.argwhere(lab_i == lb)
        i = np.argmin(o[idx[:, 0], idx[:, 1]])
        _x, _y = idx[i, 0], idx[i, 1]
        out.append([_x + start, _y + start, o[_x, _y], so[_x, _y]])
    return out


def diff_block(c1, c2, start, distance_in_px, octave_values, st, pt, pt2, intra=True, blur="scipy", s=10,
               return_intermediate=False):
    """(loops1, diff_loops1, loops2, diff_loops2) like the reference's diff_mustache(); mutates c1 and c2."""
    n = c1.shape[0]
    off = np.arange(n)[None, :] - np.arange(n)[:, None]
    nz1 = np.logical_and(c1 != 0, off >= 4)                    # (:262-264)
    nz2 = np.logical_and(c2 != 0, off >= 4)
    nzb = np.logical_and(nz1, nz2)
    empty = ([], [], [], [])
    if nz1.sum() < 50 or nz2.sum() < 50:                       # (:266)
        return (empty, None) if return_intermediate else empty
    for c in (c1, c2):                                         # (:268-273)
        c[off <= 4] = 2
        if intra:
            c[off >= distance_in_px + 1] = 2
    cd = np.zeros(c1.shape)
    cd[nzb] = c1[nzb] - c2[nzb]                                # (:275-276)

    levels = level_table(octave_values, s)
    if blur == "scipy":
        G = lambda img, lv: blur_scipy(img, lv["sigma"], lv["truncate"])
        mx = maxfilter3_scipy
    else:
        G = lambda img, lv: blur_explicit(img, lv["weights"], lv["radius"])
        mx = maxfilter3_zero
    # NB: the reference never advances the difference image's DoG inside the level loop -- `Lc = Gc - Gn` is set once
    # per octave at :336 and only `Ln` is recomputed at :363 (there is no `Lc = Ln` next to :413-425).  Every tested level
    # of an octave therefore scores the difference with the SAME image D_2 = G_2 - G_3 of that octave.  Reproduced as is.
    imgs = (None, c1, c2)
    nzs = (None, nz1, nz2)
    st_best = [None, np.zeros(int(nz1.sum())), np.zeros(int(nz2.sum()))]
    st_scale = [None, np.ones(int(nz1.sum())), np.ones(int(nz2.sum()))]
    st_p = [None, np.ones(int(nz1.sum())) * 2, np.ones(int(nz2.sum())) * 2]
    st_pair = [None, np.ones(int(nz1.sum())) * 2, np.ones(int(nz2.sum())) * 2]
    fits = []
    per_oct = s + 2
    for oi in range(len(octave_values)):
        lv = levels[oi * per_oct:(oi + 1) * per_oct]
        d_diff = G(cd, lv[1]) - G(cd, lv[2])                                 # Lc of the difference image (:315, :330, :336)
        g_cur = [None] + [G(im, lv[1]) for im in imgs[1:]]
        d_p = [None] + [G(im, lv[0]) - gc for im, gc in zip(imgs[1:], g_cur[1:])]   # (:307-325)
        g_nxt = [None] + [G(im, lv[2]) for im in imgs[1:]]
        d_c = [None] + [gc - gn for gc, gn in zip(g_cur[1:], g_nxt[1:])]            # (:337-338)
        m_p = [None, mx(d_p[1]), mx(d_p[2])]
        m_c = [None, mx(d_c[1]), mx(d_c[2])]
        for i in range(3, s + 2):                                            # (:349)
            g_cur = g_nxt
            g_nxt = [None] + [G(im, lv[i]) for im in imgs[1:]]
            d_n = [None] + [gc - gn for gc, gn in zip(g_cur[1:], g_nxt[1:])]
            loc = float(np.mean(d_diff[nzb]))                                # norm.fit (:371)
            scl = float(np.sqrt(np.mean((d_diff[nzb] - loc) ** 2)))
            fits.append((loc, scl))
            for k in (1, 2):
                nz = nzs[k]
                absd = np.abs(d_c[k][nz])
                eloc = absd.min()
                escl = absd.mean() - eloc
                p = expon_pvalue(absd, eloc, escl)                           # (:367-370)
                np.nan_to_num(p, copy=False, posinf=1, neginf=1, nan=1)      # (:386-387)
                pp = _two_sided_normal(d_diff[nz], loc, scl)                 # (:372-385)
                m_n = mx(d_n[k])
                dc = d_c[k][nz]
                upd = np.logical_and.reduce((dc > st_best[k], dc == m_c[k][nz],
                                             np.logical_or(d_p[k][nz] == m_p[k][nz], d_n[k][nz] == m_n[nz]),
                                             dc > m_p[k][nz], dc > m_n[nz]))  # (:396-407)
                st_best[k][upd] = dc[upd]
                st_scale[k][upd] = lv[i - 1]["sigma"]
                st_p[k][upd] = p[upd]
                st_pair[k][upd] = pp[upd]
                m_p[k], m_c[k] = m_c[k], m_n
            d_p, d_c = d_c, d_n

    if len(st_p[1]) < 10000 or len(st_p[2]) < 10000:                         # (:430)
        return (empty, None) if return_intermediate else empty
    o, so, pair, v = [None] * 3, [None] * 3, [None] * 3, [None] * 3
    for k in (1, 2):
        f = st_p[k] != 2
        st_p[k][f] = benjamini_hochberg(st_p[k][f])                          # (:432-436)
        o[k] = np.ones_like(c1)
        o[k][nzs[k]] = st_p[k]
        so[k] = np.ones_like(c1)
        so[k][nzs[k]] = st_scale[k]
        pair[k] = np.ones_like(c1)
        pair[k][nzs[k]] = st_pair[k]
        v[k] = np.ones_like(c1)
        v[k][nzs[k]] = st_best[k]
    mid = dict(nz1=nz1, nz2=nz2, best=st_best, scale=st_scale, p=st_p, pair=st_pair, fits=fits)
    x1, y1 = _filters(c1, nz1, o[1], so[1], pt, st, intra)
    x2, y2 = _filters(c2, nz2, o[2], so[2], pt, st, intra)
    if len(x1) == 0 or len(x2) == 0:                                         # (:507)
        return (empty, mid) if return_intermediate else empty
    if intra:                                                                # (:516-529)
        ok = _diag_filter(c1, x1, y1)
        if ok.size == 0 or ok.sum() == 0:
            return (empty, mid) if return_intermediate else empty
        x1, y1 = x1[ok], y1[ok]
        ok = _diag_filter(c2, x2, y2)
        if ok.size == 0 or ok.sum() == 0:
            return (empty, mid) if return_intermediate else empty
        x2, y2 = x2[ok], y2[ok]
    out1 = _clusters(x1, y1, o[1], so[1], start)
    out2 = _clusters(x2, y2, o[2], so[2], start)
    d1 = [r for r in out1 if pair[1][r[0] - start, r[1] - start] < pt2
          and v[1][r[0] - start, r[1] - start] > v[2][r[0] - start, r[1] - start]]      # (:567)
    d2 = [r for r in out2 if pair[2][r[0] - start, r[1] - start] < pt2
          and v[2][r[0] - start, r[1] - start] > v[1][r[0] - start, r[1] - start]]      # (:568)
    res = (out1, d1, out2, d2)
    return (res, mid) if return_intermediate else res
