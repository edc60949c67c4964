#!/usr/bin/env python3
"""The small-launch floor: bench.py's chr21 @ 5 kb step (6 blocks of 2000 x 2000 in one launch) -- wall time per step, the fused
kernel's own time (HIP events) and what is left (launch preparation, p-values, the host round trips).  GPU box.
    python scripts/small_launch_time.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch       # noqa: E402
import bench       # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
w = bench.Workload("chr21@5kb synthetic", 9630, 400, 5000, 300.0, 300, 0, dev, 0, 1)
for _ in range(5):
    w.step(False)
w.kernel_ms.clear()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    w.step(False)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
k = sum(a.elapsed_time(b) for a, b in w.kernel_ms) / steps
print("chr21 @ 5 kb: %.3f ms per step = %.0f Mpix/s; fused kernel %.3f ms (%.0f Mpix/s); everything else %.3f ms"
      % (dt * 1e3, w.total_mpix / dt, k, w.total_mpix / (k * 1e-3), dt * 1e3 - k))

# where the rest goes: host time of the enqueue (_ss_launch: allocations, work list, uploads, kernel launch) and of the finish
# (mst_found_finish incl. its wait for the kernel, then building the per-block records)
eng = w.pipe.engine
acc = {"launch": 0.0, "finish": 0.0, "results": 0.0}
for name in ("_ss_launch", "_ss_finish", "_ss_results"):
    def wrap(fn, key):
        def inner(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            acc[key] += time.perf_counter() - t
            return r
        return inner
    setattr(eng, name, wrap(getattr(eng, name), name.split("_")[-1]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    w.step(False)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / steps
print("host view per step: total %.3f ms | _ss_launch (enqueue) %.3f | _ss_finish (incl. waiting for the kernel) %.3f | "
      "_ss_results %.3f | python around them %.3f"
      % (tot * 1e3, acc["launch"] / steps * 1e3, acc["finish"] / steps * 1e3, acc["results"] / steps * 1e3,
         (tot - sum(acc.values()) / steps) * 1e3))
