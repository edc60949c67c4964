#!/usr/bin/env python3
"""How far is mst_normalize_band (branch A, prefix-sum kernel and blocked-sum kernel) from the EXACT window arithmetic?

The reference's own window sums (np.convolve -> BLAS ddot) depend on the BLAS build, so "the" float64 answer does not exist;
the yardstick here is the same formula (mustache.py:645-667) evaluated with window sums in extended precision (numpy
longdouble: 64-bit mantissa prefix sums over the whole diagonal, 2^-64 ~ 5e-20 per operation) and every later step in
longdouble as well.  Printed per shape: the largest |z_gpu - z_exact| relative to max(|z_exact|, 1e-3 * diagonal std) and
the largest absolute error, for both kernels, next to the same figures for the float64 oracle (np.convolve).

    python scripts/norm_accuracy.py            (GPU box)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def exact_band(raw, n, dpx, res):
    """longdouble restatement of normalize_sparse branch A on a [dpx+2, n] band (0 = no contact)."""
    W = int(2000000 / res)
    out = np.zeros_like(raw)
    L = np.longdouble
    for d in range(dpx + 2):
        row = raw[d, :n - d]
        nzm = row != 0
        if not nzm.any():
            continue
        vals = np.where(nzm, row + 0.001, 0.0)
        v = row[nzm]
        mean, std = float(np.mean(v)), float(np.std(v))
        if np.isnan(mean):
            mean = 0.0
        if np.isnan(std):
            std = 1.0
        m = len(row)
        left = W // 2
        pc = np.concatenate([[0], np.cumsum(nzm.astype(np.int64))])
        p1 = np.concatenate([[L(0)], np.cumsum(vals.astype(L))])
        p2 = np.concatenate([[L(0)], np.cumsum(vals.astype(L) ** 2)])
        lo = np.clip(np.arange(m) - left, 0, m)
        hi = np.clip(np.arange(m) - left + W, 0, m)
        c = (pc[hi] - pc[lo]).astype(L)
        s1, s2 = p1[hi] - p1[lo], p2[hi] - p2[lo]
        with np.errstate(all="ignore"):
            var = (s2 - s1 * s1 / c) / (c - 1)
            var = np.where(np.isfinite(var), var, L(std) ** 2)
            mu = s1 / c
            small = c < 30
            mu = np.where(small, L(mean), mu)
            var = np.where(small, L(std) ** 2, var)
            mu = np.where(np.isfinite(mu), mu, L(mean))
            z = (vals.astype(L) - mu) / np.sqrt(var)
            z = np.where(np.isfinite(z), z, L(0))
        z = z * (L(1) + np.log(L(1) + L(mean)) / np.log(L(30)))
        out[d, :m] = np.where(nzm, z, 0).astype(np.float64)          # rounding the exact z to float64 is the last step
    return out


def main():
    import torch
    import oracle
    from mustache_amd.normalize import normalize_band
    from mustache_amd.synth import band_counts
    dev = torch.device("cuda:0")
    for name, n, dpx, res, depth, seed in (("chr21@5kb", 9630, 400, 5000, 300.0, 0), ("chr1@1kb slice", 30000, 2000, 1000, 400.0, 1),
                                           ("sparse 2kb", 12000, 150, 2000, 3.0, 5)):
        raw = band_counts(n, dpx, depth, max(n // 32, 1), seed, device=dev)
        rh = raw.cpu().numpy()
        ex = exact_band(rh, n, dpx, res)
        sd = np.array([np.std(ex[d][ex[d] != 0]) if (ex[d] != 0).any() else 1.0 for d in range(dpx + 2)])[:, None]
        den = np.maximum(np.abs(ex), 1e-3 * sd)
        line = "%-16s n=%d dpx=%d window=%d nnz=%d:" % (name, n, dpx, int(2000000 / res), int((rh != 0).sum()))
        for label, kern in (("walk", "auto"), ("segment", "segment"), ("blocked", "blocked")):
            got = normalize_band(raw, n, dpx, res, kernel=kern)[0].cpu().numpy()
            err = np.abs(got - ex)
            line += "  %s rel %.2e abs %.2e" % (label, float((err / den).max()), float(err.max()))
        # the float64 oracle (np.convolve) against the same yardstick
        d_idx, i_idx = np.nonzero(rh)
        x, y, v = i_idx.astype(np.int64), (i_idx + d_idx).astype(np.int64), rh[d_idx, i_idx].copy()
        oracle.normalize_sparse(x, y, v, res, dpx)
        err = np.abs(v - ex[d_idx, i_idx])
        line += "  float64 oracle rel %.2e abs %.2e" % (float((err / den[d_idx, i_idx]).max()), float(err.max()))
        print(line, flush=True)


if __name__ == "__main__":
    main()
