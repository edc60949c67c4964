"""cProfile of the host tail (rows 8-9) on the synthetic chr1@1kb band (--full) or 12 blocks of it."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import bench
small = "--full" not in sys.argv
dev = torch.device("cuda", 0)
n = 4000 + 11 * 2000 if small else 248957
w = bench.Workload("x", n, 2000, 1000, 400.0, 800 if small else 8000, 1, dev, 0, 1)
w.pipe.run_band(w.band, w.n, w.dpx, 0.88, 0.1, distributed=False)
pr = cProfile.Profile()
tm = {}
pr.enable()
loops = w.pipe.run_band(w.band, w.n, w.dpx, 0.88, 0.1, timings=tm, distributed=False)
pr.disable()
print(len(loops), tm)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
