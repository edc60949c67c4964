#!/usr/bin/env python3
"""Device idle time inside the product path's chromosome run (pipeline.run_band, chr1 @ 1 kb): from a rocprofv3 kernel trace of
scripts/e2e_time.py, the last run's window from its first fused kernel to its last kernel -- busy time, idle time, the largest
gaps and what ran around them.
    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python scripts/e2e_time.py 248957 2000 1000 400
    python scripts/run_band_gaps.py out/.../t_kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
fused = [i for i, r in enumerate(rows) if "scale_space_kernel" in r[2]]
# runs = clusters of fused kernels separated by > 20 ms without one
runs, cur = [], [fused[0]]
for a, b in zip(fused[:-1], fused[1:]):
    if rows[b][0] - rows[a][1] > 20e6:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
for run in runs[-3:-1]:                   # the last timed runs (the very last one is the cProfile run)
    a, b = run[0], run[-1]
    while b + 1 < len(rows) and rows[b + 1][0] - rows[b][1] < 2e6:      # the tail kernels behind the last fused kernel
        b += 1
    t0, t1 = rows[a][0], max(r[1] for r in rows[a:b + 1])
    busy, edge = 0, t0
    gaps = []
    for s, e, name in rows[a:b + 1]:
        if s > edge:
            gaps.append((s - edge, name))
        busy += max(0, e - max(s, edge))
        edge = max(edge, e)
    fk = sum(rows[i][1] - rows[i][0] for i in run)
    print("run of %d fused kernels: window %.2f ms, device busy %.2f ms (fused kernels %.2f ms), idle %.2f ms in %d gaps" % (
        len(run), (t1 - t0) / 1e6, busy / 1e6, fk / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
    for g, name in sorted(gaps, reverse=True)[:8]:
        print("   gap %7.1f us before %s" % (g / 1e3, name[:80]))
