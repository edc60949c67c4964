"""Normalisation (row 1) timing on the synthetic chr1@1kb band: HIP events around mst_normalize_band."""
import sys, torch
sys.path.insert(0, ".")
from mustache_amd.normalize import normalize_band
from mustache_amd.synth import band_counts
n, dpx, res = 248957, 2000, 1000
dev = torch.device("cuda", 0)
raw = torch.empty((dpx + 2, n), dtype=torch.float64, device=dev)
for i0 in range(0, n, 16384):
    i1 = min(n, i0 + 16384)
    raw[:, i0:i1] = band_counts(n, dpx, 400.0, 8000, 1, i0=i0, i1=i1, device=dev)
import os
for kern in os.environ.get("NORM_KERNELS", "auto,segment,blocked").split(","):
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out, _, _ = normalize_band(raw, n, dpx, res, kernel=kern)
        e1.record()
        torch.cuda.synchronize()
        gb = 2 * raw.numel() * 8 / 1e9
        print("normalize[%s] %.2f ms  (%.1f GB algorithmic -> %.2f TB/s)" % (kern, e0.elapsed_time(e1), gb, gb / e0.elapsed_time(e1)))
