"""Per-stage timing of one bench step (host wall-clock with syncs between stages) -- profiling aid."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from mustache_amd.engine import BlockBatch
small = "--full" not in sys.argv
dev = torch.device("cuda", 0)
n = 4000 + 11 * 2000 if small else 248957
w = bench.Workload("x", n, 2000, 1000, 400.0, 800 if small else 8000, 1, dev, 0, 1)
pipe, eng = w.pipe, w.pipe.engine
def sync(): torch.cuda.synchronize()
for rep in range(3):
    t = {}
    sync(); t0 = time.time()
    c, nz, nzc = pipe.blocks_from_band(w.band, w.n, w.dpx, [w.start[i] for i in w.mine], w.CH)
    sync(); t1 = time.time()
    found, pval, count, fit, cap = eng.sigma_loop(c, nz, nzc, skip_empty=False, download=False)
    sync(); t2 = time.time()
    out = eng._download(found, pval, count, fit, eng.levels.n_tested)
    sync(); t3 = time.time()
    print("blocks_from_band %.2f ms | sigma_loop(+pvalues, alloc) %.2f ms | sort+download+split %.2f ms | total %.2f"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
