#!/usr/bin/env python3
"""Where the two-sample whole-genome call (run_pair_layout, hg19-shaped genome @ 5 kb) spends its wall time (GPU box)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import bench
    import bench_extra
    from mustache_amd.diff_mustache import run_pair_layout
    dev = torch.device("cuda:0")
    wg = bench_extra.genome_workload(bench, "hg19-shaped genome @5kb synthetic", 5000, 400, 300.0, 1000, dev, two_samples=True)
    for _ in range(2):
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.time()
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    print("PAIR_GENOME s per call", [round(t, 4) for t in ts])
    eng = wg.pipe.engine
    ts = []
    for _ in range(3):
        t0 = time.time()
        eng.run_band_pairs([wg.band, wg.band2], wg.n, wg.dpx, wg.start, wg.CH, select_below=0.1)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    print("DEVICE_PART s per call", [round(t, 4) for t in ts])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())


if __name__ == "__main__":
    main()
