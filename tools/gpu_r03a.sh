#!/bin/bash
# round 3, first measurement pass: the two-candidates-per-lane kernel against the one-candidate kernel (same box), the
# 8-seed sweep three ways, and rocprofv3 kernel statistics of a humanoidrun N=8192 plan and of the sweep
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 300 python tools/gpu_pk2_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/pk2_ab.txt
timeout 300 python tools/gpu_sweep.py 2>&1 | grep -v "override\|amdgpu.ids\|init sigma" | tee gpurun_out/r03/sweep.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03/prof_sweep -o sweep -- python $GRAFT_REPO_ROOT/tools/gpu_sweep.py humanoidrun 1024 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/r03/prof_sweep > gpurun_out/r03/kernel_stats_sweep.md 2>&1 || true
ls -R gpurun_out/r03 | head -30
