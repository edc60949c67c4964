"""End-to-end per-chromosome timing on the synthetic chr1@1kb band (or --small): rows 2-7 + host tail."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
small = "--full" not in sys.argv
dev = torch.device("cuda", 0)
n = 4000 + 11 * 2000 if small else 248957
w = bench.Workload("x", n, 2000, 1000, 400.0, 800 if small else 8000, 1, dev, 0, 1)
for rep in range(2):
    tm = {}
    t0 = time.time()
    loops = w.pipe.run_band(w.band, w.n, w.dpx, 0.88, 0.1, timings=tm, distributed=False)
    print("e2e %.2f s  loops %d  %s" % (time.time() - t0, len(loops), tm))
