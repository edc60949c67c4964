#!/bin/bash
# Stage markers of one CLI run (GPU box): a synthetic .hic (chr1-like at 1 kb, 60,000 bins = 29 blocks of 4000 x 4000) through
# `python -m mustache_amd -f x.hic ...` with the PROFILE library (roctx ranges per C-ABI entry point: read / normalise / launch /
# finish / tail) and MUSTACHE_ROCTX=1 (the host stages of the Python side), under rocprofv3 --marker-trace --kernel-trace.
# usage: scripts/marker_trace.sh <tag>   -> gpurun_out/marker_<tag>/summary.md   (copy to profiles/<tag>_marker_trace.md)
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/marker_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO:$REPO/tests
python - <<PY > $OUT/make_hic.log 2>&1
import torch, hic_writer, os, time
t0 = time.time()
n = hic_writer.write_synthetic_hic("/tmp/marker.hic", 60000, 2000, 1000, 400.0, 2000, 1, 200.0, torch.device("cuda:0"))
print("wrote %d records, %d bytes in %.1f s" % (n, os.path.getsize("/tmp/marker.hic"), time.time() - t0))
PY
MUSTACHE_HIP_LIB=$REPO/mustache_amd/libmustache_hip_profile.so MUSTACHE_ROCTX=1 \
  rocprofv3 --marker-trace --kernel-trace --stats -d $OUT/trace -o mt --output-format csv -- \
  python -m mustache_amd -f /tmp/marker.hic -ch chr1 -r 1kb -norm KR -pt 0.1 -st 0.88 -o /tmp/marker.tsv > $OUT/cli.log 2>&1
python - <<PY > $OUT/summary.md
import csv, glob, collections
out = "$OUT"
f = glob.glob(out + "/trace/**/*marker_api_trace.csv", recursive=True)
print("# Stage markers of one CLI run (rocprofv3 --marker-trace --kernel-trace; PROFILE library + MUSTACHE_ROCTX=1)\n")
print("command: python -m mustache_amd -f marker.hic -ch chr1 -r 1kb -norm KR -pt 0.1 -st 0.88   (synthetic .hic: 60,000 bins at 1 kb, 29 blocks of 4000 x 4000)\n")
print("\`\`\`")
print("\n".join(l.rstrip() for l in open(out + "/cli.log") if ("loops found" in l or "[gpu]" in l or "Normalizing" in l)))
print("\`\`\`\n")
if f:
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.OrderedDict()
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        name = r.get("Function") or r.get("Name") or "?"
        a = agg.setdefault(name, [0, 0, None])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if a[2] is None:
            a[2] = int(r["Start_Timestamp"]) - t0
    print("| range | calls | total ms | mean ms | first start (ms after the first marker) |\n|---|---|---|---|---|")
    for name, (c, tot, first) in sorted(agg.items(), key=lambda kv: kv[1][2]):
        print("| %s | %d | %.3f | %.3f | %.3f |" % (name, c, tot / 1e6, tot / 1e6 / c, first / 1e6))
else:
    print("no marker csv found:", glob.glob(out + "/trace/**/*.csv", recursive=True))
k = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if k:
    print("\nkernels (rocprofv3 --stats):\n\n| kernel | calls | total ms | mean us | % |\n|---|---|---|---|---|")
    for r in list(csv.DictReader(open(k[0])))[:12]:
        print("| %s | %s | %.3f | %.1f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -30 $OUT/summary.md
