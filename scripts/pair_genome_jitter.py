#!/usr/bin/env python3
"""Per-call wall time of the two-sample whole-genome call next to the container's CFS throttling counters: are the slow calls
the ones the CPU quota interrupts?  (GPU box.)   python scripts/pair_genome_jitter.py [calls]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def cpu_stat():
    out = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


def main():
    import torch
    import bench
    import bench_extra
    from mustache_amd.diff_mustache import run_pair_layout
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    import gc
    mode = os.environ.get("MST_GC", "")
    if mode == "off":
        gc.disable()
    elif mode == "freeze":
        gc.collect()
        gc.freeze()
    print("gc mode:", mode or "default", "thresholds", gc.get_threshold(), "tracked objects", len(gc.get_objects()))
    print("torch threads", torch.get_num_threads(), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
    wg = bench_extra.genome_workload(bench, "hg19-shaped genome @5kb synthetic", 5000, 400, 300.0, 1000, dev, two_samples=True)
    for _ in range(3):
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
    torch.cuda.synchronize()
    for i in range(calls):
        s0, c0, t0 = cpu_stat(), time.process_time(), time.time()
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
        torch.cuda.synchronize()
        dt, cpu, s1 = time.time() - t0, time.process_time() - c0, cpu_stat()
        print("call %2d  %.4f s  process cpu %.3f s  throttled periods +%d  throttled time +%.1f ms" % (
            i, dt, cpu, s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0),
            (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3))


if __name__ == "__main__":
    main()
