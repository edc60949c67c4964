#!/usr/bin/env python3
"""The step of ONE rank's share of chr1 @ 1 kb at the driver's rank counts, on one GPU: rank 0's contiguous block range of a world of
1, 2, 4, 8 (bench.py's strong split), timed like bench.py's step (fused kernels + p-values + records to the host).  What it
shows is the fixed cost per step that strong scaling cannot divide -- not a scaling projection: every rank of a real run also
waits for the slowest.     python scripts/rank_share_time.py [steps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch      # noqa: E402
import bench      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
w = bench.Workload("chr1@1kb synthetic", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
base = None
for world in (1, 2, 4, 8):
    w.rank, w.world = 0, world
    w.set_scaling("strong")
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    w.kernel_ms.clear()
    t0 = time.time()
    for _ in range(steps):
        w.step(False)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    k = sum(a.elapsed_time(b) for a, b in w.kernel_ms) / steps
    base = base or ms
    print("world %d: rank 0 runs %3d blocks in %d launches: %.3f ms per step (fused kernels %.3f ms, rest %.3f); "
          "N x this = %.1f ms vs %.1f ms at N = 1 -> %.3f" % (world, len(w.mine), len(w.groups), ms, k, ms - k, world * ms, base, base / (world * ms)))
