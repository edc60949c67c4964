#!/usr/bin/env python3
"""Strong-scaling PROJECTION from one GPU (no multi-GPU box is available to the builder): the path has no data-path
collective -- ranks share nothing but the final gather of a few hundred records -- so a rank's step time at N ranks is the
time of its own share of the blocks.  For N = 1, 2, 4, 8 this times rank 0's share (the largest: ceil(124 / N) blocks) of
bench.py's workload on the one GPU present, with bench.py's own step function, and prints the projected whole-job rate
(all blocks / slowest rank's time) and efficiency.  It is a projection, not a measurement of xGMI or RCCL: what it shows is
how the fixed per-step costs (launch tails, the last group's post-processing) weigh as the share shrinks.

    python scripts/project_scaling.py [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    base = None
    rows = []
    w = bench.Workload("chr1@1kb", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
    from mustache_amd.sharding import shard_blocks
    for N in (1, 2, 4, 8):
        w.mine = shard_blocks(len(w.start), 0, N)
        for _ in range(2):
            w.step(False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            w.step(False)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / steps
        rate = w.total_mpix / dt
        base = base or rate
        rows.append({"ranks": N, "blocks_rank0": len(w.mine), "ms_per_step": round(dt * 1e3, 3), "projected_mpix_s": round(rate, 1),
                     "efficiency": round(rate / (N * base), 4),
                     "split_bound": round(len(w.start) / (N * len(w.mine)), 4)})
        print("SCALE " + json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
