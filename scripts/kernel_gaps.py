#!/usr/bin/env python3
"""Timeline of the kernels of a short repeated call from a rocprofv3 --kernel-trace CSV: for the last `calls` repetitions of the
sequence that starts with a kernel whose name contains `first`, the offset and duration of every kernel and the idle time of
the device between them.

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python scripts/diff_time.py
    python scripts/kernel_gaps.py out/*/t_kernel_trace.csv zero_counts 3"""
import csv
import sys

path, first = sys.argv[1], sys.argv[2]
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
# a call = from one occurrence of `first` that follows a gap of > 50 us to the next such occurrence
heads = [i for i in starts if i == 0 or rows[i][0] - rows[i - 1][1] > 50000]
for a, b in list(zip(heads[:-1], heads[1:]))[-calls:]:
    t0 = rows[a][0]
    busy = 0
    print("---- call of %d kernels, %.1f us from first start to last end; gap before it %.1f us" % (
        b - a, (rows[b - 1][1] - t0) / 1e3, (rows[a][0] - rows[a - 1][1]) / 1e3 if a else 0.0))
    prev_end = t0
    for s, e, name in rows[a:b]:
        print("  +%8.1f us  %8.1f us  (idle before %6.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3, name[:90]))
        busy += e - s
        prev_end = max(prev_end, e)
    print("  busy %.1f us of %.1f" % (busy / 1e3, (rows[b - 1][1] - t0) / 1e3))
