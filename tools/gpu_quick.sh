#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for lib in "" $GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/libmbd_hip_noslp.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  MBD_HIP_LIB=$lib python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ.get('MBD_HIP_LIB','')[-20:], 'steps/s %.1f  ms/step %.3f  rollout_kernel_ms %.3f' % (d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
