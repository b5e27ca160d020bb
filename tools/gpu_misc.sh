#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { env $3 python bench.py --config $1 --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-8s' % ('$1', '$2'), 'steps/s %.1f  async %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
for c in hopper512 halfcheetah1024; do
  for r in 1 2; do run $c consts "X=1"; run $c runtime "MBD_NO_PLANAR_FLAGS=1"; done
done | tee gpurun_out/planar_flags.log
