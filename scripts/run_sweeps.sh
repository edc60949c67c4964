#!/bin/bash
# All randomised parity sweeps in one go on the GPU box (the HIP path against the oracle on randomly drawn cases; tests may import
# oracle/): usage  scripts/run_sweeps.sh <tag> <seed>   -> gpurun_out/sweeps_<tag>.txt (copy to profiles/<tag>_fuzz_sweeps.txt)
TAG=${1:-r06}; SEED=${2:-9601}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/sweeps_$TAG.txt
cd $REPO
T0=$(date +%s)
{
echo "# Fuzz sweeps, FUZZ_SEED=$SEED, one MI355X box; last lines of each log"
echo; echo "== scripts/fuzz_parity.py 300";   FUZZ_SEED=$SEED python scripts/fuzz_parity.py 300 2>&1 | tail -2
echo; echo "== scripts/fuzz_pipeline.py 60";  FUZZ_SEED=$SEED python scripts/fuzz_pipeline.py 60 2>&1 | tail -2
echo; echo "== scripts/fuzz_diff.py 100";     FUZZ_SEED=$SEED python scripts/fuzz_diff.py 100 2>&1 | tail -2
echo; echo "== scripts/fuzz_genome.py 12";    FUZZ_SEED=$SEED python scripts/fuzz_genome.py 12 2>&1 | tail -2
echo; echo "== scripts/fuzz_hic_stream.py 150"; FUZZ_SEED=$SEED python scripts/fuzz_hic_stream.py 150 2>&1 | tail -2
echo; echo "== scripts/fuzz_hic_rows_gpu.py 300 $SEED"; python scripts/fuzz_hic_rows_gpu.py 300 $SEED 2>&1 | tail -2
echo; echo "== wide-radius instantiations: FUZZ_OCT=3.2,6.4 scripts/fuzz_parity.py 40; FUZZ_OCT=1.6,3.2,6.4 scripts/fuzz_pipeline.py 6"
FUZZ_OCT=3.2,6.4 FUZZ_SEED=$SEED python scripts/fuzz_parity.py 40 2>&1 | tail -1
FUZZ_OCT=1.6,3.2,6.4 FUZZ_SEED=$SEED python scripts/fuzz_pipeline.py 6 2>&1 | tail -1
echo; echo "box time: $(( $(date +%s) - T0 )) s"
} > $OUT 2>&1
cat $OUT
