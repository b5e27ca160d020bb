#!/bin/bash
# N3: K independent humanoidrun plans (seeds 0..K-1, metric config) run concurrently on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|override"
import sys
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args
from mbd_hip.scripts.run_mbd import run_concurrent
for K in (1, 2, 4, 8, 16):
    plans = [Args(seed=s, env_name="humanoidrun", Nsample=1024, Hsample=50, Ndiffuse=100, temp_sample=0.1,
                  disable_recommended_params=True, not_render=True) for s in range(K)]
    run_concurrent(plans[:1])
    rews, mus, secs = run_concurrent(plans)
    print("K=%2d concurrent plans: %.3f s  -> %.0f plan-steps/s (%.0f per plan), mean rew_final %.3f" % (
        K, secs, K * 99 / secs, 99 / secs, sum(rews) / K))
PY
