#!/bin/bash
# the ring of three noise buffers + progress word (head) against the two-buffer, event-ordered form (variant), same box
cd "$GRAFT_REPO_ROOT" || exit 1
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants/libmbd_hip_head_before_ring.so
run() { MBD_HIP_LIB=$3 python bench.py --config $1 --steps 150 --warmup 10 --no-cpu-baseline --no-final-reward --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-18s %-8s' % ('$1', '$2'), 'value %.1f  async %.1f  ms/step %.4f  async ms %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d.get('ms_per_step_async', 0), d['roofline']['kernel_avg_ms']))"; }
{
for c in ${CONFIGS:-sweep8 humanoidrun8192 humanoidrun4096 metric}; do
  for round in 1 2; do run $c head ""; run $c before $V; done
done
for lib in "" $V; do echo "mbd_plan_run, lib=${lib:-head}"; MBD_HIP_LIB=$lib python tools/gpu_planrun.py 2>&1 | grep -v amdgpu.ids | tail -3; done
} | tee gpurun_out/ring_ab.log
