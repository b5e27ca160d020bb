#!/bin/bash
# first GPU contact: tests + a quick timing of the metric config
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
