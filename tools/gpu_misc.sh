#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/gpu_ab2.sh notests metric hopper512 halfcheetah1024 humanoidtrack2048demo
