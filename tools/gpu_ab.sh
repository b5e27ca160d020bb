#!/bin/bash
# within-box A/B of rollout-kernel variants (box-to-box variance is ~10 %: never compare across calls)
cd "$GRAFT_REPO_ROOT" || exit 1
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants
run() { MBD_HIP_LIB=$2 python bench.py --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$1', 'steps/s %.1f  ms/step %.3f  rollout_kernel_ms %.4f' % (d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
for round in 1 2 3; do
  run head ""
  for f in $V/*.so; do run $(basename $f .so) $f; done
done
