#!/bin/bash
# head against every library under lib/variants, same box, alternating: CONFIGS="metric ..." tools/gpu_lib_ab.sh
cd "$GRAFT_REPO_ROOT" || exit 1
V=$GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants
run() { MBD_HIP_LIB=$3 python bench.py --config $1 --steps ${STEPS:-150} --warmup 10 --no-cpu-baseline --no-final-reward --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-14s' % ('$1', '$2'), 'value %.1f  async %.1f  ms/step %.4f  async ms %.4f  rollout_kernel_ms %.4f' % (d['value'], d['value_async'], d['ms_per_step'], d.get('ms_per_step_async', 0), d['roofline']['kernel_avg_ms']))"; }
{
for c in ${CONFIGS:-metric humanoidrun4096 humanoidrun8192}; do
  for round in 1 2; do
    run $c head ""
    for f in $V/*.so; do [[ $f == *plain* ]] && continue; run $c $(basename $f .so | sed s/libmbd_hip_//) $f; done
  done
done
} | tee gpurun_out/lib_ab.log
