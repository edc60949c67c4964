#!/usr/bin/env python3
"""Does a whole-genome run cost the same whatever ran before it in the process?  (GPU box.)
    python scripts/stream_aliasing_check.py fresh      # the genome workload in a fresh process
    python scripts/stream_aliasing_check.py after      # ... after a chr1 @ 1 kb workload of another engine ran
Round 4 found the second form 0.032 -> 0.053 s: every engine created its own side streams, HIP maps streams to a few hardware
queues in creation order, and the genome engine's launch stream landed on the default stream's queue -- the host tail's small
kernels then waited behind the next launch's fused kernel.  engine.device_streams() (one fixed set per device and process)
removed the dependence; this script is the check."""
import sys, time, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
import bench
import bench_extra
dev = torch.device('cuda:0')
def instrument(eng):
    acc = {}
    for name in ("_ss_launch", "_ss_finish", "_ss_results", "_download_selected"):
        def wrap(fn, key):
            def inner(*a, **k):
                t = time.perf_counter(); r = fn(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t; return r
            return inner
        setattr(eng, name, wrap(getattr(eng, name), name))
    return acc
def run(wg, tag):
    acc = instrument(wg.pipe.engine)
    wg.pipe.run_layout(wg.layout, wg.band, 0.88, 0.1)
    acc.clear()
    ts = []; tms = []
    for _ in range(5):
        tm = {}
        torch.cuda.synchronize(); t0 = time.time()
        wg.pipe.run_layout(wg.layout, wg.band, 0.88, 0.1, timings=tm)
        torch.cuda.synchronize(); ts.append(time.time() - t0); tms.append(tm)
    print(tag, "e2e median %.4f" % sorted(ts)[2], "scale_space_s %.4f tail_s %.4f" % (tms[2]["scale_space_s"], tms[2]["tail_s"]),
          {k: round(v / 5 * 1e3, 2) for k, v in acc.items()}, flush=True)
which = sys.argv[1]
if which == "fresh":
    wg = bench_extra.genome_workload(bench, "g", 5000, 400, 300.0, 1000, dev, two_samples=False)
    run(wg, "fresh")
else:
    w = bench.Workload("c", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
    for _ in range(2): w.step(False)
    wg = bench_extra.genome_workload(bench, "g", 5000, 400, 300.0, 1000, dev, two_samples=False)
    run(wg, "after chr1")
