#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/gpu_exchange.py 2>/dev/null | tee gpurun_out/exchange.log
