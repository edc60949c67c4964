#!/bin/bash
# Detailed PMC passes on the small bench (12 blocks); one rocprofv3 run per counter group.
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmcd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $REPO/bench.py --small --steps 1 --warmup 0 --core > $OUT/log_g$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
allc = {}
for f in sorted(glob.glob("$OUT/g*/pmc_counter_collection.csv")):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "scale_space_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            per[int(r["Dispatch_Id"])]["_g"] = int(r["Grid_Size"])
    big = max(v["_g"] for v in per.values())
    first = min(k for k, v in per.items() if v["_g"] == big)
    allc.update({k: v for k, v in per[first].items() if k != "_g"})
for k in sorted(allc):
    print("%-28s %.6g" % (k, allc[k]))
PY
