#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for N in 1024 2048 4096 8192; do
MBD_BENCH_N=$N python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-final-reward 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench(torch stream) N=$N', 'steps/s %.1f  ms/step %.3f  rollout_kernel_ms %.4f' % (d['steps_per_sec'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
python - <<'PY' 2>&1 | grep "N="
import sys
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
for N in (1024, 2048, 4096, 8192):
    a = Args(seed=0, env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=40, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    r, d = run_diffusion(a, return_details=True); r, d = run_diffusion(a, return_details=True)
    print("plan.run(own stream) N=%d: %.1f steps/s %.3f ms/step" % (N, d["steps_per_sec"], 1e3/d["steps_per_sec"]))
PY
