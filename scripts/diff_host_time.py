#!/usr/bin/env python3
"""Host view of the two-sample call (chr21 @ 5 kb shape): how long the host needs to queue everything, how long it then waits
for the device, and what it does after the wait (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mustache_amd.diff_mustache import _pairs_from_filled
dev = torch.device("cuda:0")
w5 = bench.Workload("chr21@5kb", 9630, 400, 5000, 300.0, 300, 0, dev, 0, 1)
band_b, _ = bench.make_band(9630, 400, 260.0, 300, 7, 5000, dev)
call = lambda: _pairs_from_filled(w5.pipe.engine, w5.pipe, [w5.band, band_b], w5.n, w5.dpx, w5.start, w5.CH, pt=0.1)
for _ in range(5):
    call()
marks = {}
real_sync = torch.cuda.Stream.synchronize
def sync(self):
    marks["before"] = time.perf_counter()
    real_sync(self)
    marks["after"] = time.perf_counter()
torch.cuda.Stream.synchronize = sync
rows = []
for _ in range(30):
    t0 = time.perf_counter()
    call()
    t1 = time.perf_counter()
    rows.append((marks["before"] - t0, marks["after"] - marks["before"], t1 - marks["after"]))
rows.sort(key=lambda r: sum(r))
q, wt, post = rows[len(rows) // 2]
print("HOST median call %.3f ms = queueing %.3f + waiting for the device %.3f + after the wait %.3f" % ((q + wt + post) * 1e3, q * 1e3, wt * 1e3, post * 1e3))
eng = w5.pipe.engine
acc = {}
def wrap(obj, name):
    fn = getattr(obj, name)
    def inner(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, inner)
for nm in ("_ss_launch", "_carve", "_pairs_one_wait"):
    wrap(eng, nm)
for nm in ("mst_diff_dog_band", "mst_found_finish", "mst_pair_pvalues_dog", "mst_bh_select_nowait", "mst_pair_gather", "mst_scale_space_band"):
    pass
for _ in range(30):
    call()
print({k: round(v / 30 * 1e3, 3) for k, v in acc.items()})
