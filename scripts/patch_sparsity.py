#!/usr/bin/env python3
"""How often would PATCH-LEVEL sparsity pay in the fused kernel?  (round-4 verdict item 3b, measured on the CPU.)

A wave of the fused kernel owns a 32 x 16 pixel patch of a tile (mst_scale_space.hip: two column groups of 8 pixels x 32 rows).
Today a wave without a tested pixel skips its 3x3 maxima, sieve and statistics; the proposal is to let a wave whose patch PLUS
its 1-pixel ring holds no tested pixel also skip its axis-1 pass and DoG (35 % of a level).  That only pays where such patches
exist inside the band.  This script counts them on the bench's thinned chr1 @ 1 kb workload (pixel (i, i + d) kept with
probability min(1, keep / (d + 1)), bench.py write_synthetic_hic) for several `keep`, over the band tiles of one 4000 x 4000
block, and prints the share of empty 32 x 16 patches (ring included) next to the share of tested pixels.

    python scripts/patch_sparsity.py            (CPU, ~1 min)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mustache_amd.synth import band_counts, _uniform   # noqa: E402

n, dpx, CH = 12000, 2000, 4000
val = band_counts(n, dpx, 400.0, 400, 1, i0=4000, i1=8000 + dpx + 2, device="cpu")          # rows d, columns i
d = torch.arange(dpx + 2, dtype=torch.int64)[:, None]
i = torch.arange(4000, 8000 + dpx + 2, dtype=torch.int64)[None, :]
u = _uniform(6, d, i, 9) * (d + 1).to(torch.float64)
print("%8s %14s %22s %26s" % ("keep", "tested share", "empty 32x16 patches", "empty incl. 1-pixel ring"))
for keep in (1e9, 200.0, 50.0, 20.0, 5.0, 2.0):
    raw = torch.where(u < keep, val, torch.zeros_like(val)).numpy()
    # dense block rows 0..CH-1 (chromosome rows 4000..7999): tested = raw != 0 and 4 <= d <= dpx + 1
    blk = np.zeros((CH, CH), dtype=bool)
    r = np.arange(CH)
    for dd in range(4, dpx + 2):
        L = CH - dd
        blk[r[:L], r[:L] + dd] = raw[dd, :L] != 0
    inband = np.zeros((CH, CH), dtype=bool)
    for dd in range(4, dpx + 2):
        L = CH - dd
        inband[r[:L], r[:L] + dd] = True
    # patches on a 32 x 16 lattice that lie entirely inside the tested band
    P = blk[:CH // 32 * 32, :CH // 16 * 16].reshape(CH // 32, 32, CH // 16, 16)
    B = inband[:CH // 32 * 32, :CH // 16 * 16].reshape(CH // 32, 32, CH // 16, 16)
    full = B.all(axis=(1, 3))
    any_t = P.any(axis=(1, 3))
    # ring: dilate the tested mask by one pixel, then the same count
    dil = blk.copy()
    dil[1:, :] |= blk[:-1, :]; dil[:-1, :] |= blk[1:, :]
    d2 = dil.copy()
    d2[:, 1:] |= dil[:, :-1]; d2[:, :-1] |= dil[:, 1:]
    any_r = d2[:CH // 32 * 32, :CH // 16 * 16].reshape(CH // 32, 32, CH // 16, 16).any(axis=(1, 3))
    print("%8s %13.2f%% %21.3f%% %25.3f%%" % ("all" if keep > 1e8 else "%g" % keep, 100.0 * blk[inband].mean(),
                                                100.0 * (~any_t[full]).mean(), 100.0 * (~any_r[full]).mean()))
