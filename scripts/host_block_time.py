#!/usr/bin/env python3
"""What the reference's own per-block seam costs: mustache(c, ...) with a dense NumPy block of 4000 x 4000 (pageable host memory in,
fills written back in place, loops out) -- the PCIe-inclusive rate of the boundary, never `value`.  GPU box.
    python scripts/host_block_time.py [calls]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from mustache_amd.mustache import mustache      # noqa: E402
from mustache_amd.synth import synth_coo        # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, dpx = 4000, 2000
x, y, v = synth_coo(n, dpx, depth=300.0, seed=3, nloops=150)
c0 = np.zeros((n, n))
c0[x, y] = v
ts = []
for i in range(calls + 2):
    c = c0.copy()
    t0 = time.time()
    loops = mustache(c, "chr1", "chr1", 1000, None, 0, n, -1, dpx, [1.6, 3.2], 0.88, 0.1)
    torch.cuda.synchronize()
    if i >= 2:
        ts.append(time.time() - t0)
ts.sort()
med = ts[len(ts) // 2]
print("mustache(c) on a host block of %d x %d: median %.1f ms per call (min %.1f) = %.0f Mpix/s incl. the upload of 128 MB from pageable "
      "memory, the fills written into the caller's block and the host tail; %d loops" % (n, n, med * 1e3, ts[0] * 1e3, n * n / 1e6 / med, len(loops)))

# stage by stage: the form that downloaded the filled block (until round 5) next to the host-side fills used now
from mustache_amd.mustache import _engine, block_tail      # noqa: E402
eng = _engine([1.6, 3.2])
acc = {}
for i in range(6):
    c = c0.copy()
    marks = [time.time()]
    dev = torch.from_numpy(np.ascontiguousarray(c)).to(eng.device).unsqueeze(0)
    torch.cuda.synchronize(); marks.append(time.time())
    batch = eng.run_blocks(dev, dpx, intra=True)
    torch.cuda.synchronize(); marks.append(time.time())
    h = batch.c[0].cpu().numpy()
    marks.append(time.time())
    c[...] = h
    marks.append(time.time())
    loops = block_tail(batch, 0, 0, 0.1, 0.88, intra=True)
    marks.append(time.time())
    # the alternative to the download: the fills of mustache.py:703-706 applied on the host
    c2 = c0.copy()
    t0 = time.time()
    for r in range(n):
        c2[r, :min(n, r + 5)] = 2.0
        c2[r, r + dpx + 1:] = 2.0
    t_fill = time.time() - t0
    assert np.array_equal(c2, c)
    if i >= 2:
        for k, a, b in zip(("upload", "run_blocks", "download", "copy into c", "tail"), marks[:-1], marks[1:]):
            acc[k] = acc.get(k, 0.0) + (b - a) / 4
        acc["host fill instead"] = acc.get("host fill instead", 0.0) + t_fill / 4
print("stages (ms): " + ", ".join("%s %.1f" % (k, v * 1e3) for k, v in acc.items()))
