#!/bin/bash
# round-2 evidence refresh after the lazy candidates / short exact sequences: tests, bench lines, rocprofv3 stats + PMC,
# the per-dispatch timeline of a step (with and without the timing events), variant B's redundancy
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
tools/gpu_r02.sh tests bench prof metric hopper512 halfcheetah1024 humanoidrun4096 humanoidtrack2048demo 2>&1 | tee $OUT/r02d.log | tail -40
MODES="lazy-ev lazy-noev nolazy-noev" tools/gpu_timeline.sh > /dev/null 2>&1; cp $OUT/timeline.log $OUT/profiles/r02_timeline.txt
python tools/gpu_exchange.py 2>/dev/null | tee $OUT/exchange.log
tools/gpu_noev.sh | tail -12
