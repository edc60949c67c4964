"""Step time of rank 0's block range at the given world sizes for several splits of the range into launches / stages
(MST_BENCH_SHARES): what bench.py's split rule was chosen from.   python scripts/share_split_time.py [world ...]"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
w = bench.Workload("chr1@1kb synthetic", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
steps = 30
for world in [int(a) for a in sys.argv[1:]] or (8, 4):
    w.rank, w.world = 0, world
    w.set_scaling("strong")
    for sh in os.environ.get("SPLITS", "").split(";") if os.environ.get("SPLITS") else ("1", "0.6,0.4", "0.7,0.3", "0.8,0.2", "0.85,0.15", "0.9,0.1", "0.94,0.06", "0.5,0.3,0.2", "0.45,0.4,0.15", "0.5,0.44,0.06",
               "0.29,0.29,0.29,0.13", "0.35,0.35,0.2,0.1", "0.32,0.32,0.3,0.06"):
        os.environ["MST_BENCH_SHARES"] = sh
        for _ in range(4):
            w.step(False)
        torch.cuda.synchronize()
        w.kernel_ms.clear()
        t0 = time.time()
        for _ in range(steps):
            w.step(False)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        k = sum(a.elapsed_time(b) for a, b in w.kernel_ms) / steps
        print("world %d shares %-22s groups %s: %.3f ms per step (kernels %.3f, rest %.3f)" % (world, sh, [len(g) for g in w.groups], ms, k, ms - k), flush=True)
