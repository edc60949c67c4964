#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats of the default bench run, then PMC passes (each its own run with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes), then the summary -- all in ONE session, so every figure of
# profiles/<tag>_summary.md comes from the same build on the same box.
# usage: scripts/profile_bench.sh <tag>     (outputs under gpurun_out/prof_<tag>/; copy summary + json + csv into profiles/)
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-file > $OUT/bench_trace.log 2>&1
# PMC: one launch for all 12 blocks of the --small workload (MST_BENCH_OVERLAP=1), so a dispatch's counters cover a known
# pixel count; the same runs also hold the normalisation kernels (n = 26,000 bins) and the two-sample kernels (chr21 shape)
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  MST_BENCH_OVERLAP=1 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$N -o pmc --output-format csv -- python $REPO/bench.py --small --steps 1 --warmup 0 --core > $OUT/bench_pmc_$N.log 2>&1
done
cd $REPO && python scripts/summarize_profile.py $TAG $OUT > $OUT/summarize.log 2>&1
tail -5 $OUT/summarize.log
