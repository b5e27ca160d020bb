#!/bin/bash
# mbd_plan_run with and without the hipGraph replay of the reverse loop
cd "$GRAFT_REPO_ROOT" || exit 1
for G in 1 0; do
MBD_GRAPH=$G python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|override\|init sigma"
import os, sys
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
for env, kw in [("car2d", dict(Nsample=128, Hsample=30, Ndiffuse=50)), ("car2d", dict(Nsample=1024)), ("cartpole", {}), ("humanoidtrack", {}),
                ("humanoidrun", dict(Nsample=1024, Ndiffuse=100, disable_recommended_params=True))]:
    best = 0
    for rep in range(3):
        a = Args(seed=0, env_name=env, not_render=True, **kw)
        r, d = run_diffusion(a, return_details=True)
        best = max(best, d["steps_per_sec"])
    print("MBD_GRAPH=%s %-14s N=%5d: %8.1f steps/s  rew %.3f" % (os.environ["MBD_GRAPH"], env, a.Nsample, best, r))
PY
done
