#!/bin/bash
# PMC of ONE tile-list launch (12 blocks of the chr1 @ 1 kb shape, scripts/exp_variants.py --one) per ablation of the PROFILE
# library: what the max / sieve / statistics phase adds in instructions, in busy VALU cycles and in time (round-6 pricing of
# the intra-wave phase overlap).  usage: scripts/pmc_phase.sh <tag> [ablation values...]   (default: 0 1)
TAG=${1:-x}; shift
ABL=${@:-0 1}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmcp_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO MUSTACHE_HIP_LIB=${PMC_LIB:-$REPO/mustache_amd/libmustache_hip_profile.so} MST_IGNORE_NONFINITE=1 EXP_MODES=${EXP_MODES:-skip}
for a in $ABL; do
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_BRANCH"; do
    i=$((i+1))
    MST_ABLATE=$a rocprofv3 --pmc $C --kernel-trace -d $OUT/a${a}_g$i -o pmc --output-format csv -- python $REPO/scripts/exp_variants.py --one > $OUT/log_a${a}_g$i.txt 2>&1
  done
  MST_ABLATE=$a python $REPO/scripts/exp_variants.py --one 2>&1 | grep "^EXP" | cut -c1-260 > $OUT/time_a$a.txt
done
python - <<PY
import csv, glob, collections, re
for a in "$ABL".split():
    allc = {}
    for f in sorted(glob.glob("$OUT/a%s_g*/pmc_counter_collection.csv" % a)):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "scale_space_kernel" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        last = max(per)
        allc.update(per[last])
    t = open("$OUT/time_a%s.txt" % a).read()
    m = re.search(r'"skip": \{"ms": ([0-9.]+)', t) or re.search(r'"dense": \{"ms": ([0-9.]+)', t)
    print("ablation %s  ms %s  " % (a, m.group(1) if m else "?") + "  ".join("%s=%.5g" % (k, allc[k]) for k in sorted(allc)))
PY
