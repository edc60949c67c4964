#!/bin/bash
# one PMC group on the small bench; usage: pmc_one.sh "<counters>" [env assignments...]
C="$1"; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
env "$@" rocprofv3 --pmc $C --kernel-trace -d $OUT/g -o pmc --output-format csv -- python $REPO/bench.py --small --steps 1 --warmup 0 --no-cpu > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
per = collections.defaultdict(dict)
for r in csv.DictReader(open("$OUT/g/pmc_counter_collection.csv")):
    if "scale_space_kernel" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["_g"] = int(r["Grid_Size"])
big = max(v["_g"] for v in per.values())
first = min(k for k, v in per.items() if v["_g"] == big)
print({k: "%.4g" % v for k, v in per[first].items() if k != "_g"})
PY
