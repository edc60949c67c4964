"""Early kernel timing: B dense 1 kb-shape blocks (CH=4000, dpx=2000) built with torch from the synthetic band,
then the HIP prologue + fused sigma loop.  Scratch script (superseded by bench.py)."""
import sys, time
import torch, numpy as np
from mustache_amd.engine import ScaleSpaceEngine
from mustache_amd.synth import band_counts

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dpx = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = "cuda"
eng = ScaleSpaceEngine()
n = CH + (B - 1) * (CH - dpx)
blocks = torch.zeros((B, CH, CH), dtype=torch.float64, device=dev)
r = torch.arange(CH, device=dev)
for b in range(B):
    s = b * (CH - dpx)
    band = band_counts(n, dpx, 400.0 if dpx == 2000 else 300.0, max(n // 32, 1), 1, i0=s, i1=s + CH, device=dev)
    # crude per-diagonal z-score (input preparation only)
    m = band.sum(1, keepdim=True) / (band > 0).sum(1, keepdim=True).clamp(min=1)
    sd = torch.sqrt((((band - m) ** 2) * (band > 0)).sum(1, keepdim=True) / (band > 0).sum(1, keepdim=True).clamp(min=1)).clamp(min=1e-9)
    z = torch.where(band > 0, (band - m) / sd, torch.zeros_like(band))
    for d in range(dpx + 2):
        L = CH - d
        blocks[b, r[:L], r[:L] + d] = z[d, :L]
torch.cuda.synchronize()
raw = blocks.clone()
for skip in (False, True):
    for it in range(3):
        blocks.copy_(raw)
        torch.cuda.synchronize(); t0 = time.time()
        nz, nzc = eng.prologue(blocks, dpx, True)
        torch.cuda.synchronize(); t1 = time.time()
        out = eng.sigma_loop(blocks, nz, nzc, skip_empty=skip, download=False)
        torch.cuda.synchronize(); t2 = time.time()
    cnt = out[2].cpu().numpy()
    mpix = B * CH * CH / 1e6
    print("skip_empty=%s prologue %.2f ms  sigma_loop %.2f ms  -> %.1f Mpix/s (kernel only)  found/block %s nz %s"
          % (skip, (t1 - t0) * 1e3, (t2 - t1) * 1e3, mpix / (t2 - t1), cnt[:4], nzc.cpu().numpy()[:4]))
