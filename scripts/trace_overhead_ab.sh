#!/bin/bash
# A/B in one session: bench.py --core untraced and under rocprofv3 --kernel-trace, staged launches and a launch per group (profiles/r06b_trace_overhead.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
pick() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], 'ms/step', d['ms_per_step'], 'kernel/step', d['roofline']['kernel_ms_per_step'], 'band_skip kernel', d['band_skip']['kernel_ms_per_step'], 'no_share', d['no_share']['kernel_ms_per_step'], d['no_share']['band_skip']['kernel_ms_per_step'])
" $1 "$2"; }
for mode in 1 0; do
  MUSTACHE_STAGED=$mode python $R/bench.py --steps 5 --warmup 2 --core > $R/gpurun_out/tab_u$mode.json 2>/dev/null; pick $R/gpurun_out/tab_u$mode.json "untraced staged=$mode"
  MUSTACHE_STAGED=$mode rocprofv3 --kernel-trace -d $R/gpurun_out/tab_t$mode -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --core > $R/gpurun_out/tab_t$mode.json 2>/dev/null; pick $R/gpurun_out/tab_t$mode.json "traced   staged=$mode"
done
