#!/bin/bash
# the full GPU suite on head, then within-box A/B of head against every lib/variants/*.so on the given configs
# usage: gpu_ab2.sh [notests] config...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" != notests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/ab2_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/ab2_tests.log
  tail -3 gpurun_out/ab2_tests.log
else shift; fi
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants
run() { MBD_HIP_LIB=$3 python bench.py --config $1 --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-8s' % ('$1', '$2'), 'steps/s %.1f  async %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
{
for c in ${@:-metric}; do
  for round in 1 2; do
    run $c head ""
    for f in $V/*.so; do run $c $(basename $f .so) $f; done
  done
done
} | tee gpurun_out/ab2.log
