#!/usr/bin/env python3
"""Where a whole per-chromosome run spends its wall time on the GPU box (chr21 @ 5 kb shape by default).

    python scripts/e2e_time.py [n dpx res depth]

Prints the pipeline's own stage timings (median of 5 runs from the normalised band) and a cProfile top list of one run."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from mustache_amd.pipeline import ChromosomePipeline
    from mustache_amd.synth import band_counts
    from mustache_amd.normalize import normalize_band
    a = sys.argv[1:]
    n, dpx, res, depth = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (9630, 400, 5000, 300.0)
    dev = torch.device("cuda:0")
    pipe = ChromosomePipeline((1.6, 3.2), device=dev)
    pipe.engine.share_tiles = os.environ.get("E2E_SHARE", "1") != "0"      # E2E_SHARE=0: MST_FLAG_NO_SHARE
    if os.environ.get("E2E_BLOCKS"):
        pipe.overlap_blocks = int(os.environ["E2E_BLOCKS"])                # blocks per fused-kernel launch
    raw = band_counts(n, dpx, depth, max(10, n // 32), 0, device=dev)
    torch.cuda.synchronize()
    rows = []
    for it in range(6):
        t0 = time.time()
        band, _, _ = normalize_band(raw.clone(), n, dpx, res)
        torch.cuda.synchronize()
        t1 = time.time()
        tm = {}
        loops = pipe.run_band(band, n, dpx, 0.88, 0.1, timings=tm, distributed=False)
        torch.cuda.synchronize()
        t2 = time.time()
        if it:
            rows.append(dict(normalize_ms=(t1 - t0) * 1e3, run_band_ms=(t2 - t1) * 1e3, scale_space_ms=tm["scale_space_s"] * 1e3, fused_launches_ms=tm.get("kernel_ms", 0.0),
                             tail_ms=tm["tail_s"] * 1e3, loops=len(loops), blocks=tm["blocks"]))
    rows.sort(key=lambda r: r["run_band_ms"])
    print("E2E " + json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rows[len(rows) // 2].items()}))
    pr = cProfile.Profile()
    pr.enable()
    pipe.run_band(band, n, dpx, 0.88, 0.1, distributed=False)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue())


if __name__ == "__main__":
    main()
