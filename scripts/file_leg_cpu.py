#!/usr/bin/env python3
"""Process CPU seconds (all threads) of every stage of bench.py's file leg, pass by pass, next to the container's CFS throttling
counters: how much CPU does a pass burn OUTSIDE the reader, and which passes does the quota interrupt?  (GPU box.)
    python scripts/file_leg_cpu.py [passes]"""
import os
import shutil
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    from hic_writer import write_synthetic_hic
    from mustache_amd.hicfile import HicFile
    from mustache_amd.normalize import band_from_packed, normalize_band, read_hic_stream_to_device
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if os.environ.get("MST_AFFINITY"):          # "first": one hardware thread per core (the lower half of the CPU numbers); "a-b": that range
        spec = os.environ["MST_AFFINITY"]
        ncpu = os.cpu_count()
        cpus = range(ncpu // 2) if spec == "first" else range(int(spec.split("-")[0]), int(spec.split("-")[1]) + 1)
        os.sched_setaffinity(0, cpus)
        print("affinity:", spec, "->", len(os.sched_getaffinity(0)), "CPUs")
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        print("cpu0 thread siblings:", sib)
    except OSError:
        pass
    dev = torch.device("cuda:0")
    import gc
    mode = os.environ.get("MST_GC", "")
    if mode == "off":
        gc.disable()
    elif mode == "freeze":
        gc.collect()
        gc.freeze()
    print("gc mode:", mode or "default", "thresholds", gc.get_threshold(), "tracked objects", len(gc.get_objects()))
    print("torch threads", torch.get_num_threads(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
    w = bench.Workload("chr1@1kb synthetic", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
    tmp = tempfile.mkdtemp(prefix="mst_flc_")
    try:
        path = os.path.join(tmp, "chr1_1kb.hic")
        write_synthetic_hic(path, w.n, w.dpx, w.res, 400.0, 8000, 1, 200.0, dev)
        h = HicFile(path)
        for rep in range(passes):
            torch.cuda.synchronize()
            s0 = cpu_stat()
            marks = [(time.time(), time.process_time())]
            pc = read_hic_stream_to_device(h, "chr1", w.res, "KR", w.dpx, 0, dev, part=(0, 1))
            marks.append((time.time(), time.process_time()))
            band = band_from_packed(pc, w.dpx, dev)
            n = int(band.shape[1])
            torch.cuda.synchronize()
            marks.append((time.time(), time.process_time()))
            nb, _, _ = normalize_band(band, n, w.dpx, w.res)
            torch.cuda.synchronize()
            marks.append((time.time(), time.process_time()))
            loops = w.pipe.run_band(nb, n, w.dpx, 0.88, 0.1, timings={})
            torch.cuda.synchronize()
            marks.append((time.time(), time.process_time()))
            s1 = cpu_stat()
            names = ("read", "scatter", "normalize", "kernels+tail")
            print("pass %d total %.4f s | " % (rep, marks[-1][0] - marks[0][0]) + " | ".join(
                "%s %.4f s wall / %.3f cpu" % (nm, b[0] - a[0], b[1] - a[1]) for nm, a, b in zip(names, marks[:-1], marks[1:]))
                + " | throttled +%d periods, +%.1f ms | loops %d" % (
                    s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0),
                    (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3, len(loops)))
            del pc, band, nb
        h.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
