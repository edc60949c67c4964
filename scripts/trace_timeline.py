#!/usr/bin/env python3
"""Phase timeline of the fused kernel from a PROFILE build's MST_TRACE file (see mst_scale_space.hip, MST_STAMP).

    MUSTACHE_HIP_LIB=mustache_amd/libmustache_hip_profile.so MST_TRACE=/tmp/t.bin python scripts/exp_variants.py --one
    python scripts/trace_timeline.py /tmp/t.bin

Stamps per (workgroup, wave, level): 0 level start, 1 V pass done, 2 after the V->H barrier, 3 H pass done, 4 DoG + edge
strip written, 5 after the second barrier, 6 sieve phase done.  Prints mean cycles per segment, the barrier waits, and
the per-wave spread."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
nwg, nw, nl, ns = (int(v) for v in raw[:4])
t = raw[4:].reshape(nwg, nw, nl, ns).astype(np.int64)
extra = t[:, :, nl - 1, :]
t = t[:, :, :nl - 1, :]
seg_names = ["taps+V pass", "wait barrier 1", "H pass", "DoG+edge", "wait barrier 2", "max/sieve/stats"]
rows = []
for wg in range(nwg):
    if t[wg, 0, 0, 0] == 0:
        continue
    has_nz = bool((extra[wg, :, 2] >> 40).any())
    lv = [l for l in range(nl - 1) if t[wg, 0, l, 0] != 0]
    for l in lv:
        s = t[wg, :, l, :]                      # [wave][stamp]
        seg = np.diff(s[:, :7], axis=1)         # 6 segments
        if s[0, 6] == 0:                        # level without a sieve phase (first two levels of octave 1)
            seg[:, 5] = 0
        rows.append((wg, l, has_nz, seg))
print("workgroups sampled:", len({r[0] for r in rows}), " with tested pixels:", len({r[0] for r in rows if r[2]}))
for flag in (True, False):
    sel = [r[3] for r in rows if r[2] == flag]
    if not sel:
        continue
    a = np.array(sel)                            # [n][wave][seg]
    print("\n%s tiles: mean cycles per wave per level (%d level samples)" % ("band" if flag else "empty", len(a)))
    tot = a.sum(axis=2).mean()
    for i, nme in enumerate(seg_names):
        print("  %-18s %8.0f  (%4.1f %%)   wave spread (max-min over the 4 waves) %6.0f" % (
            nme, a[:, :, i].mean(), 100 * a[:, :, i].mean() / tot, (a[:, :, i].max(axis=1) - a[:, :, i].min(axis=1)).mean()))
    print("  %-18s %8.0f" % ("level total", tot))
# whole-tile time
for wg in range(nwg):
    if t[wg, 0, 0, 0] == 0:
        continue
    lv = [l for l in range(nl - 1) if t[wg, 0, l, 0] != 0]
    first, last = t[wg, :, lv[0], 0].min(), t[wg, :, lv[-1], :7].max()
    stg = extra[wg, :, 1].max()
    print("wg %2d tile %6d levels %d  level loop %7d cycles (staging end -> first level start %d)" % (
        wg, int(extra[wg, 0, 2] & 0xFFFFFFFF), len(lv), last - first, first - stg))
    if wg > 6:
        break
