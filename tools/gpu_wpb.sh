#!/bin/bash
# wavefronts per workgroup of the rollout kernel vs candidate count (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
for N in 1024 2048 4096 8192 16384; do
  for W in 1 2 4; do
    MBD_WPB=$W MBD_BENCH_N=$N python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=%5d wpb=%d  ms/step %.3f  rollout_ms %.4f  (%.0f 1024-candidate steps/s)' % ($N, $W, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['value']))"
  done
done
