#!/bin/bash
# first GPU contact: tests + a quick timing of the metric config
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
python - <<'PY' 2>&1 | tail -8
import sys, time
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
for N in (1024, 4096):
    a = Args(seed=0, env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=30, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    r, d = run_diffusion(a, return_details=True)
    print("humanoidrun N=%d: %.1f steps/s, rew_final %.4f" % (N, d["steps_per_sec"], r), d["rew_means"][[0,-1]])
PY
