#!/bin/bash
# Instruction-cache counters (SQ_IFETCH, SQ_IFETCH_LEVEL, ...) of the fused kernel on the small bench: does the 79 KB kernel thrash the 64 KB cache?
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
rocprofv3 -L > $OUT/avail.txt 2>&1
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  MST_BENCH_OVERLAP=1 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$N -o pmc --output-format csv -- python $REPO/bench.py --small --steps 1 --warmup 0 --core > $OUT/log_$N.txt 2>&1
done
ls $OUT/*/ | head
