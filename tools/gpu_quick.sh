#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps/s %.1f  ms/step %.3f  rollout_kernel_ms %.3f' % (d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
