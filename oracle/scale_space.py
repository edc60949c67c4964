"""Oracle: prologue, Gaussian sigma-stack, DoG, 3x3 max, (x, y, sigma) sieve, per-level p-values.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates mustache.py:699-772 and the SciPy kernels it
calls (scipy/ndimage/_filters.py, scipy 1.15.3; scipy/stats/_continuous_distns.py) in plain NumPy.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import scipy.special as _sc


# ----------------------------------------------------------------------------------------------------------
# level table  (mustache.py:714-752)
# ----------------------------------------------------------------------------------------------------------
def gaussian_weights(sigma, radius):
    """Normalised 1-D Gaussian taps for x in [-radius, radius].

    Follows scipy/ndimage/_filters.py:226-236 (`_gaussian_kernel1d`, order 0): exp(-0.5/sigma^2 * x^2)
    divided by its `ndarray.sum()`.  The taps are bit-symmetric, so the reversal at _filters.py:322 is a no-op.
    """
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / sigma2 * x ** 2)
    return phi / phi.sum()


def level_table(octave_values, s=10):
    """All Gaussian levels the reference evaluates, in evaluation order.

    Per octave value ``o`` the reference blurs the block with sigma_k = o * 2**((k-1)/s), k = 1..s+2
    (mustache.py:716, :722, :731, :748; `s = 10` is hard-wired at :711).  The truncate argument is
    ``t = ((w-1)/2 - 0.5)/sigma`` with ``w = 2*ceil(2*sigma)+1`` (:717-718) and SciPy turns it into the
    integer radius ``int(t*sigma + 0.5)`` (_filters.py:314-316).

    Returns a list of dicts: octave index, k, sigma, truncate, radius, weights.
    """
    out = []
    for oi, o in enumerate(octave_values):
        for k in range(1, s + 3):
            sigma = o if k == 1 else o * 2 ** ((k - 1) / s)
            w = 2 * math.ceil(2 * sigma) + 1
            t = (((w - 1) / 2) - 0.5) / sigma
            radius = int(t * float(sigma) + 0.5)
            out.append(dict(octave=oi, k=k, sigma=sigma, truncate=t, radius=radius,
                            weights=gaussian_weights(sigma, radius)))
    return out


# ----------------------------------------------------------------------------------------------------------
# separable Gaussian (mustache.py:719 etc. -> scipy gaussian_filter -> C correlate1d)
# ----------------------------------------------------------------------------------------------------------
def _reflect_extend_rows(a, r):
    """'reflect' (half-sample symmetric) extension along axis 0 by r rows each side: d c b a | a b c d | d c b a."""
    n = a.shape[0]
    if r == 0:
        return a
    idx = np.arange(-r, n + r)
    # general multi-fold reflection (period 2n), valid also when r > n
    idx = np.mod(idx, 2 * n)
    idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    return a[idx]


def _correlate_sym_axis0(a, w, r):
    """SciPy's C `correlate1d` on a symmetric kernel, along axis 0, in the same floating-point order.

    The C loop (scipy/ndimage/src/ni_filters.c, symmetric branch) computes, per output sample,
        t = x[c] * w[0];  for j = -r .. -1:  t += (x[c+j] + x[c-j]) * w[j]
    i.e. centre tap first, then the pairs from the outermost inwards, each pair summed before the multiply,
    no fused multiply-add.  NumPy elementwise ops reproduce this bit-for-bit (verified).
    """
    n = a.shape[0]
    ext = _reflect_extend_rows(a, r)
    acc = ext[r:r + n] * w[r]
    for j in range(r, 0, -1):
        acc = acc + (ext[r - j:r - j + n] + ext[r + j:r + j + n]) * w[r - j]
    return acc


def blur_explicit(c, weights, radius):
    """2-D separable Gaussian exactly as `gaussian_filter` evaluates it: axis 0 first, then axis 1
    (scipy/ndimage/_filters.py:424-427), float64 intermediate, mode='reflect'."""
    v = _correlate_sym_axis0(np.ascontiguousarray(c), weights, radius)
    g = _correlate_sym_axis0(np.ascontiguousarray(v.T), weights, radius).T
    return np.ascontiguousarray(g)


def blur_scipy(c, sigma, truncate):
    """The call the reference makes (mustache.py:719): third-party SciPy, used for pinning and CPU timing."""
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(c, sigma, truncate=truncate, order=0)


def maxfilter3_zero(d):
    """3x3 maximum with zero padding == maximum_filter(d, footprint=ones((3,3)), mode='constant')
    (mustache.py:740-743, :757-758; scipy/ndimage/_filters.py:1303-1354, cval=0.0)."""
    h, w = d.shape
    p = np.zeros((h + 2, w + 2), dtype=d.dtype)
    p[1:-1, 1:-1] = d
    m = p[0:h, 0:w].copy()
    for dy in range(3):
        for dx in range(3):
            if dy == 0 and dx == 0:
                continue
            np.maximum(m, p[dy:dy + h, dx:dx + w], out=m)
    return m


def maxfilter3_scipy(d):
    from scipy.ndimage import maximum_filter
    return maximum_filter(d, footprint=np.ones((3, 3)), mode='constant')


# ----------------------------------------------------------------------------------------------------------
# prologue (mustache.py:699-706)
# ----------------------------------------------------------------------------------------------------------
def block_prologue(c, distance_in_px, intra=True):
    """Tested-pixel mask and constant fills.  Mutates ``c`` like the reference does.

    nz = (c != 0) AND (col-row >= 4), taken BEFORE the fills (mustache.py:699); then every pixel with
    col-row <= 4 becomes 2 (:703) and, for an intra-chromosomal block, every pixel with
    col-row >= distance_in_px+1 becomes 2 (:705-706).
    """
    n = c.shape[0]
    off = np.arange(n)[None, :] - np.arange(n)[:, None]          # col - row
    nz = np.logical_and(c != 0, off >= 4)
    c[off <= 4] = 2
    if intra:
        c[off >= distance_in_px + 1] = 2
    return nz


# ----------------------------------------------------------------------------------------------------------
# sigma loop (mustache.py:708-772)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class ScaleSpaceResult:
    best: np.ndarray            # vAll  : best DoG response per nz pixel (0 where never updated)
    scale: np.ndarray           # Scales: recorded sigma per nz pixel (1 where never updated)
    pval: np.ndarray            # pAll  : p-value per nz pixel (2 where never updated)
    level: np.ndarray           # 0 = never updated, else 1 + index into `tested` (oracle-side convenience)
    tested: list = field(default_factory=list)   # one dict per tested level: octave, k, sigma, loc, scale


def expon_pvalue(absd, loc, scale):
    """1 - expon.cdf(|D|, loc, scale)  (mustache.py:756).

    scipy: cdf = -expm1(-x) on the open support x > 0 and 0 at x <= 0
    (_distn_infrastructure.py:2127-2139, _continuous_distns.py:2087-2088).
    """
    x = (absd - loc) / scale
    cdf = np.zeros_like(x)
    pos = x > 0
    cdf[pos] = -_sc.expm1(-x[pos])
    cdf[np.isnan(x)] = np.nan
    return 1 - cdf


def scale_space_levels(c, nz, octave_values, s=10, blur="scipy", keep_levels=False):
    """Run the two-octave sigma loop on a prologue'd block.

    ``blur``: "scipy" calls scipy.ndimage like the reference; "explicit" uses the NumPy restatement above
    (bit-identical, asserted by tests/test_oracle_golden.py).
    """
    levels = level_table(octave_values, s)
    if blur == "scipy":
        def G(lv):
            return blur_scipy(c, lv["sigma"], lv["truncate"])
        mx = maxfilter3_scipy
    else:
        def G(lv):
            return blur_explicit(c, lv["weights"], lv["radius"])
        mx = maxfilter3_zero

    nnz = int(nz.sum())
    best = np.zeros(nnz)
    scale = np.ones(nnz)
    pval = np.ones(nnz) * 2
    level = np.zeros(nnz, dtype=np.int32)
    tested = []
    kept = {}
    per_oct = s + 2
    for oi in range(len(octave_values)):
        lv = levels[oi * per_oct:(oi + 1) * per_oct]
        g_prev = G(lv[0])                       # k = 1   (:719)
        g_cur = G(lv[1])                        # k = 2   (:725)
        d_p = g_prev - g_cur                    # D_1     (:728)
        g_nxt = G(lv[2])                        # k = 3   (:734)
        d_c = g_cur - g_nxt                     # D_2     (:738)
        m_p = mx(d_p)
        m_c = mx(d_c)
        if keep_levels:
            kept[(oi, 1)] = d_p
            kept[(oi, 2)] = d_c
        for i in range(3, s + 2):               # (:744)  tested level = D_{i-1}, neighbours D_{i-2}, D_i
            g_cur = g_nxt
            g_nxt = G(lv[i])                    # k = i+1 (:748-751)
            d_n = g_cur - g_nxt                 # D_i     (:754)
            absd = np.abs(d_c[nz])
            loc = absd.min()                    # expon.fit: loc = min, scale = mean - loc (:755)
            scl = absd.mean() - loc
            p = expon_pvalue(absd, loc, scl)    # (:756)
            m_n = mx(d_n)
            dc = d_c[nz]
            upd = np.logical_and.reduce((
                dc > best, dc == m_c[nz],
                np.logical_or(d_p[nz] == m_p[nz], d_n[nz] == m_n[nz]),
                dc > m_p[nz], dc > m_n[nz]))    # (:760-765)
            best[upd] = dc[upd]
            scale[upd] = lv[i - 1]["sigma"]     # scales[o][i] = sigma_i  (:767)
            pval[upd] = p[upd]
            tested.append(dict(octave=oi, k=i, sigma=lv[i - 1]["sigma"], loc=loc, scale=scl))
            level[upd] = len(tested)
            if keep_levels:
                kept[(oi, i)] = d_n
            d_p, d_c = d_c, d_n
            m_p, m_c = m_c, m_n
    res = ScaleSpaceResult(best=best, scale=scale, pval=pval, level=level, tested=tested)
    if keep_levels:
        res.dog = kept
    return res
