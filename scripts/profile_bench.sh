#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (own runs, as the guide prescribes).
# usage: scripts/profile_bench.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  # one launch for all 12 blocks (MST_BENCH_OVERLAP=1), so the counters of a dispatch cover a known pixel count
  MST_BENCH_OVERLAP=1 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$N -o pmc --output-format csv -- python $REPO/bench.py --small --steps 1 --warmup 0 --no-cpu > $OUT/bench_pmc_$N.log 2>&1
done
ls -R $OUT | head -50
