"""Deterministic synthetic banded contact maps (SURVEY.md section 8d shapes).

No real Hi-C map is available offline (the bundled RAWobserved is missing), so tests and
``bench.py`` run on synthetic chromosomes: a distance-decay band ``E[d] = depth/(1+d) + 0.3`` with planted
blobs (loops), count noise, and per-bin biases.  Every draw is a pure function of ``(seed, diagonal, bin)``
through a 64-bit integer mix, evaluated with torch tensor ops, so the same chromosome comes out

  * on CPU and on the GPU (only IEEE add/mul/div/sqrt/floor are used after the integer hash), and
  * for any column range, which lets each rank generate exactly the part of the band its blocks cover.

The generator is bench/test input plumbing, not part of the loop-calling path.
"""
import math

import torch

_M1 = -4658895280553007687    # 0xBF58476D1CE4E5B9 as int64
_M2 = -7723592293110705685    # 0x94D049BB133111EB as int64
_K_SEED = -7046029254386353131  # 0x9E3779B97F4A7C15
_K_D = 0x632BE59BD9B4E019
_K_I = 0x2545F4914F6CDD1D


def _lsr(z, k):
    """logical shift right of an int64 tensor"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix(z):
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _i64(v):
    """wrap a Python int into the signed 64-bit range"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _uniform(seed, d, i, k):
    """U[0,1) float64, pure function of the integer arguments (tensors broadcast)."""
    z = (d * _K_D + i * _K_I) + _i64(seed * _K_SEED + k * 0x1000193)
    z = _mix(_mix(z) + k)
    return (_lsr(z, 11)).to(torch.float64) * (1.0 / 9007199254740992.0)


def bin_bias(seed, i0, i1, device="cpu"):
    """Per-bin multiplicative bias in [0.6, 1.6)."""
    i = torch.arange(i0, i1, dtype=torch.int64, device=device)
    zero = torch.zeros((), dtype=torch.int64, device=device)
    return 0.6 + _uniform(seed + 7919, zero - 1, i, 0)


def loop_list(n, dpx, nloops, seed):
    """Planted loops: anchors (a, a+b), amplitude in [2,6), width sigma in [0.8, 2.5).  Host-side, tiny."""
    idx = torch.arange(nloops, dtype=torch.int64)
    zero = torch.zeros((), dtype=torch.int64)
    ua = _uniform(seed + 101, zero - 2, idx, 0)
    ub = _uniform(seed + 101, zero - 2, idx, 1)
    uamp = _uniform(seed + 101, zero - 2, idx, 2)
    usig = _uniform(seed + 101, zero - 2, idx, 3)
    b = 8 + torch.floor(ub * max(dpx - 16, 1)).to(torch.int64)
    a = torch.floor(ua * (n - b - 1).clamp(min=1).to(torch.float64)).to(torch.int64)
    amp = 2.0 + 4.0 * uamp
    sig = 0.8 + 1.7 * usig
    return a, b, amp, sig


def band_counts(n, dpx, depth, nloops, seed, i0=0, i1=None, device="cpu"):
    """Raw (bias-divided) contact values on the band, diagonal-major: out[d, i-i0] = value of pixel (i, i+d)
    for d in [0, dpx+1], i in [i0, i1); 0 where there is no contact or i+d >= n."""
    if i1 is None:
        i1 = n
    nd = dpx + 2
    d = torch.arange(nd, dtype=torch.int64, device=device)[:, None]
    i = torch.arange(i0, i1, dtype=torch.int64, device=device)[None, :]
    e = depth / (1.0 + d.to(torch.float64)) + 0.3
    e = e.expand(nd, i1 - i0).clone()
    # planted blobs: additive bump  E += E0 * (amp-1) / (1 + r2/(2 sig^2))^2 on a (2R+1)^2 stamp
    a, b, amp, sig = loop_list(n, dpx, nloops, seed)
    sel = (a + 12 >= i0 - 12) & (a - 12 < i1)
    a, b, amp, sig = a[sel].to(device), b[sel].to(device), amp[sel].to(device), sig[sel].to(device)
    if a.numel():
        R = 10
        off = torch.arange(-R, R + 1, dtype=torch.int64, device=device)
        di = off[None, :, None]                       # row offset
        dj = off[None, None, :]                       # col offset
        pi = a[:, None, None] + di                    # pixel row
        pj = (a + b)[:, None, None] + dj              # pixel col
        pd = pj - pi
        r2 = (di * di + dj * dj).to(torch.float64)
        bump = (amp[:, None, None] - 1.0) / (1.0 + r2 / (2.0 * sig[:, None, None] ** 2)) ** 2
        base = depth / (1.0 + pd.clamp(min=0).to(torch.float64)) + 0.3
        ok = (pi >= i0) & (pi < i1) & (pd >= 0) & (pd < nd) & (pj < n)
        flat = (pd * (i1 - i0) + (pi - i0))[ok]
        e.view(-1).index_add_(0, flat, (base * bump)[ok])
    g = (_uniform(seed, d, i, 0) + _uniform(seed, d, i, 1) + _uniform(seed, d, i, 2) + _uniform(seed, d, i, 3)
         - 2.0) * math.sqrt(3.0)
    cnt = torch.floor(e + torch.sqrt(e) * g + 0.5).clamp(min=0.0)
    bias_i = bin_bias(seed, i0, i1, device)[None, :]
    j = (i + d)
    bias_j = 0.6 + _uniform(seed + 7919, torch.zeros((), dtype=torch.int64, device=device) - 1, j, 0)
    val = cnt / (bias_i * bias_j)
    val = torch.where(j < n, val, torch.zeros_like(val))
    return val


def band_to_coo(band, i0=0):
    """Diagonal-major band -> upper-triangular COO (x, y, v) with v > 0, ordered by (d, i)."""
    d, c = torch.nonzero(band > 0, as_tuple=True)
    x = c + i0
    return x, x + d, band[d, c]


def synth_coo(n, dpx, depth=300.0, nloops=None, seed=0, device="cpu"):
    """Whole synthetic chromosome as COO numpy arrays (host) -- for tests and small configurations."""
    if nloops is None:
        nloops = max(n // 32, 1)
    band = band_counts(n, dpx, depth, nloops, seed, device=device)
    x, y, v = band_to_coo(band)
    return x.cpu().numpy(), y.cpu().numpy(), v.cpu().numpy()
