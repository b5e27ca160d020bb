#!/bin/bash
# lazy candidates + fused noise workgroups: the full GPU suite, then a within-box A/B against the materialised path
# (MBD_NO_LAZY=1) and against aux-stream generation (MBD_NO_FUSED_NOISE=1) on every single-GPU config
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/lazy_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/lazy_tests.log
tail -5 gpurun_out/lazy_tests.log
run() { env $3 python bench.py --config $1 --steps 198 --warmup 20 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-8s' % ('$1', '$2'), 'steps/s %.1f  async %.1f  ms/step %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
{
for round in 1 2; do
  run metric lazy "X=1"
  run metric nolazy "MBD_NO_LAZY=1"
done
for c in hopper512 halfcheetah1024 humanoidrun4096 humanoidtrack2048demo; do
  for round in 1 2; do run $c lazy "X=1"; run $c nolazy "MBD_NO_LAZY=1"; done
done
} | tee gpurun_out/lazy_ab.log
