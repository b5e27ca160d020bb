#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/probes/rollout_timeline.py 2>/dev/null | grep prologue
tools/gpu_ab2.sh notests metric humanoidtrack2048demo humanoidrun4096
