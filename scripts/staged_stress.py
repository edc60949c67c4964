#!/usr/bin/env python3
"""Stress of the staged launch path (GPU box): hundreds of steps over SIX different block ranges in rotation -- more work lists than
the library's four-entry cache holds, so device copies of the lists are evicted, re-uploaded and their buffers regrown all the
time -- dense and tile list alternating, every step's found records (pixels, levels, p-values of every block) compared with the
first step of its range.  A race between a stage's kernel, the finish behind its event and the list cache would show as a
differing record.     python scripts/staged_stress.py [steps]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
w = bench.Workload("chr1@1kb synthetic", 248957, 2000, 1000, 400.0, 8000, 1, dev, 0, 1)
ranges = [(0, 8), (3, 8), (7, 8), (1, 4), (5, 16), (15, 16)]         # (rank, world) of the strong split: 16 / 15 / 31 / 8 blocks


def digest(w, skip):
    """every group's records are hashed AS THEY ARE HANDED OUT: they are views of two alternating page-locked staging sets, valid
    until the second-next download (engine._pinned) -- a caller that keeps the first group of three until the end reads the
    third group's bytes"""
    w.step(skip)                      # (sets up w._group_starts for the current range)
    h = hashlib.sha256()
    for recs, fits, nzc in w.pipe.engine.sigma_loop_band_overlapped(w.band, w.n, w.dpx, w._group_starts, w.CH, skip_empty=skip,
                                                                   sort=False, with_value=False, with_q=False):
        for r in recs:
            o = np.argsort(r["pixel"], kind="stable")
            h.update(r["pixel"][o].tobytes()); h.update(r["level"][o].tobytes()); h.update(r["pval"][o].tobytes())
        h.update(np.asarray(nzc).tobytes())
    return h.hexdigest()


want, bad, t0 = {}, 0, time.time()
for it in range(steps):
    rank, world = ranges[it % len(ranges)]
    skip = bool((it // len(ranges)) & 1)
    w.rank, w.world = rank, world
    w.set_scaling("strong")
    d = digest(w, skip)
    key = (rank, world, skip)
    if key not in want:
        want[key] = d
    elif want[key] != d:
        bad += 1
        print("MISMATCH step", it, key, flush=True)
    if it % 100 == 0:
        print("step", it, key, [len(g) for g in w.groups], "%.0f s" % (time.time() - t0), flush=True)
print("done: %d steps over %d (range, launch form) combinations, %d mismatches" % (steps, len(want), bad))
