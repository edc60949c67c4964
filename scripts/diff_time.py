#!/usr/bin/env python3
"""Where the two-sample call spends its wall time (chr21 @ 5 kb shape, 6 block pairs; GPU box)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from mustache_amd.diff_mustache import _pairs_from_filled
    dev = torch.device("cuda:0")
    w5 = bench.Workload("chr21@5kb", 9630, 400, 5000, 300.0, 300, 0, dev, 0, 1)
    band_b, _ = bench.make_band(9630, 400, 260.0, 300, 7, 5000, dev)
    for _ in range(3):
        _pairs_from_filled(w5.pipe.engine, w5.pipe, [w5.band, band_b], w5.n, w5.dpx, w5.start, w5.CH, pt=0.1)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        _pairs_from_filled(w5.pipe.engine, w5.pipe, [w5.band, band_b], w5.n, w5.dpx, w5.start, w5.CH, pt=0.1)
    torch.cuda.synchronize()
    print("DIFF ms per call %.3f" % ((time.time() - t0) / 10 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        _pairs_from_filled(w5.pipe.engine, w5.pipe, [w5.band, band_b], w5.n, w5.dpx, w5.start, w5.CH, pt=0.1)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
    print(s.getvalue())


if __name__ == "__main__":
    main()
