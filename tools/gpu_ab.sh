#!/bin/bash
# within-box A/B of rollout-kernel variants (box-to-box variance is ~10 %: never compare across calls)
# usage: gpu_ab.sh [config] [rounds]   — benches libmbd_hip.so ("head") and every lib/variants/*.so
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CFG=${1:-metric}; ROUNDS=${2:-3}
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants
run() { MBD_HIP_LIB=$2 python bench.py --config $CFG --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-10s' % ('$CFG', '$1'), 'steps/s %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
for round in $(seq $ROUNDS); do
  run head ""
  for f in $V/*.so; do run $(basename $f .so) $f; done
done | tee gpurun_out/ab_$CFG.log
