#!/usr/bin/env python3
"""Upper bound of a producer / consumer ("wave roles") split of the fused kernel, measured without building it (PROFILE library,
GPU box):  the axis-0 half of every level (MST_ABLATE=32: staging + V passes + barriers) and the other half (MST_ABLATE=24:
staging + H passes + DoG + maxima + sieve + statistics, no V pass) are launched AT THE SAME TIME on two streams over the same
12 blocks, so that every SIMD carries waves of complementary phases as it would with dedicated producer and consumer waves --
minus the barriers that would couple them.  Compared with the full kernel on the same blocks, and with each half alone.

    MUSTACHE_HIP_LIB=mustache_amd/libmustache_hip_profile.so MST_IGNORE_NONFINITE=1 python scripts/exp_roles.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from mustache_amd.normalize import normalize_band            # noqa: E402
from mustache_amd.pipeline import ChromosomePipeline, block_tiling   # noqa: E402
from mustache_amd.synth import band_counts                   # noqa: E402

dev = torch.device("cuda:0")
blocks, dpx, res = 12, 2000, 1000
n = 4000 + (blocks - 1) * 2000
band, _, _ = normalize_band(band_counts(n, dpx, 400.0, 800, 1, device=dev), n, dpx, res)
CH, start, end = block_tiling(n, dpx)
engs = [ChromosomePipeline((1.6, 3.2), device=dev).engine for _ in range(2)]
for e in engs:
    e.share_tiles = False
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
keep = []


def launch(eng, variant, stream, skip):
    os.environ["MST_ABLATE"] = str(variant)
    with torch.cuda.stream(stream):                 # the launch only (no finish: it would wait for the kernel)
        nzc = torch.empty(len(start), dtype=torch.int32, device=dev)
        keep.append(eng._ss_launch(None, None, nzc, skip, None, None, False, (band, int(n), int(dpx), [int(v) for v in start], int(CH))))


def timed(jobs, skip, reps=5):
    out = []
    for it in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for (eng, variant, stream) in jobs:
            launch(eng, variant, stream, skip)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        keep.clear()
        if it:
            out.append(e0.elapsed_time(e1))
    out.sort()
    return out[len(out) // 2]


for skip in (False, True):
    full = timed([(engs[0], 0, streams[0])], skip)
    two_full = timed([(engs[0], 0, streams[0]), (engs[1], 0, streams[1])], skip)
    v_only = timed([(engs[0], 32, streams[0])], skip)
    hs_only = timed([(engs[0], 24, streams[0])], skip)
    both = timed([(engs[0], 32, streams[0]), (engs[1], 24, streams[1])], skip)
    print("ROLES %s: full kernel %.2f ms (two of them at once %.2f = %.2f each) | V half alone %.2f | H + sieve half alone %.2f | both "
          "halves at once %.2f ms = %.3f of the full kernel (sum of the halves alone: %.2f)"
          % ("tile list" if skip else "dense", full, two_full, two_full / 2, v_only, hs_only, both, both / full, v_only + hs_only), flush=True)
