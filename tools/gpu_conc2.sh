#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for R in 0 98304; do echo "== MBD_LDS_RESERVE=$R"; MBD_LDS_RESERVE=$R bash tools/gpu_concurrent.sh | grep "K= 2\|K= 4\|K= 8\|K= 1 "; 
MBD_LDS_RESERVE=$R python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench steps/s %.1f rollout %.4f' % (d['steps_per_sec'], d['roofline']['kernel_avg_ms']))"; done
