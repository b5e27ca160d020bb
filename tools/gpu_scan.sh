#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
for N in (64, 256, 512, 1024, 2048, 4096, 8192, 16384):
    a = Args(seed=0, env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=40, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    r, d = run_diffusion(a, return_details=True)
    r, d = run_diffusion(a, return_details=True)
    print("humanoidrun N=%5d: %8.1f steps/s  %7.3f ms/step  %9.0f rollouts/s  rew_final %.3f" % (N, d["steps_per_sec"], 1e3/d["steps_per_sec"], N*d["steps_per_sec"], r))
for env, N, T in (("hopper",512,0.1),("halfcheetah",1024,0.4),("humanoidtrack",2048,0.1),("car2d",128,0.1)):
    a = Args(seed=0, env_name=env, Nsample=N, Hsample=50 if env!="car2d" else 30, Ndiffuse=50, temp_sample=T, disable_recommended_params=True, not_render=True, enable_demo=(env=="humanoidtrack"))
    r, d = run_diffusion(a, return_details=True)
    r, d = run_diffusion(a, return_details=True)
    print("%s N=%d: %.1f steps/s rew_final %.3f" % (env, N, d["steps_per_sec"], r))
PY
