#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats_b -o r -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward > $OUT/stats_b.log 2>&1
grep -o '"kernel_avg_ms": [0-9.]*\|"ms_per_step": [0-9.]*' $OUT/stats_b.log
