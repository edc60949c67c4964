#!/usr/bin/env python3
"""Timing of the device BH + selection stage (mst_bh_select) next to the fused kernel (GPU box).

    python scripts/select_time.py            # both BASELINE shapes: 4000^2 / dpx 2000 (12 blocks) and 2000^2 / dpx 400 (6 blocks)

Prints, per shape: fused-kernel time (empty tiles skipped), found records per block, records with p < pt per block (what
the stage sorts), the stage's device time (events, median of 7) for pt = 0.1 and pt = 0.05, and the selected counts."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shape(dpx, res, blocks, depth):
    import numpy as np
    import torch
    from mustache_amd.pipeline import ChromosomePipeline, block_tiling
    from mustache_amd.synth import band_counts
    from mustache_amd.normalize import normalize_band
    dev = torch.device("cuda:0")
    CH0 = max(2 * dpx, 2000)
    n = CH0 + (blocks - 1) * (CH0 - dpx)
    raw = band_counts(n, dpx, depth, 100 * blocks, 1, device=dev)
    band, _, _ = normalize_band(raw, n, dpx, res)
    del raw
    pipe = ChromosomePipeline((1.6, 3.2), device=dev)
    CH, start, end = block_tiling(n, dpx)
    eng = pipe.engine
    tm = []
    for it in range(3):
        found, pval, count, fit, cap, nzc = eng.sigma_loop_band(band, n, dpx, start, CH, skip_empty=True, download=False,
                                                                timing=tm)
    torch.cuda.synchronize()
    out = {"CH": CH, "dpx": dpx, "blocks": len(start), "kernel_ms": round(tm[-1][0].elapsed_time(tm[-1][1]), 3)}
    cnt = count.cpu().numpy().view(np.uint32).astype(np.int64)
    out["found_per_block"] = [int(c) for c in cnt]
    pv = pval.cpu().numpy()
    nt = eng.levels.n_tested
    for pt in (0.1, 0.05):
        out["p_below_%g" % pt] = [int((pv[b, :cnt[b]] < pt).sum()) for b in range(len(cnt))]
        ms = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sel, fits = eng._download_selected(found, pval, count, fit, nt, cap, pt)
            e1.record()
            torch.cuda.synchronize()
            if it:
                ms.append(e0.elapsed_time(e1))
        ms.sort()
        out["select_ms_%g" % pt] = round(ms[len(ms) // 2], 3)
        out["selected_%g" % pt] = [len(s["pixel"]) for s in sel]
    print("SELECT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    shape(2000, 1000, 12, 400.0)
    shape(400, 5000, 6, 300.0)
