#!/bin/bash
# do long plans stay finite?  (library variants under lib/variants, plus the in-tree build)
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" $(ls $GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants/*.so 2>/dev/null); do
MBD_HIP_LIB=$lib python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|override\|init sigma"
import sys, os
sys.path.insert(0, "model-based-diffusion_amd")
import numpy as np
from mbd_hip.planners.mbd_planner import Args, run_diffusion
for env, kw in [("humanoidstandup", {}), ("humanoidrun", {}), ("humanoidrun", dict(disable_recommended_params=True, Nsample=1024, Ndiffuse=100))]:
    a = Args(seed=0, env_name=env, not_render=True, **kw)
    r, d = run_diffusion(a, return_details=True)
    rm = d["rew_means"]
    bad = np.where(~np.isfinite(rm))[0]
    print(os.path.basename(os.environ.get("MBD_HIP_LIB", "")) or "in-tree", env, a.Nsample, "rew_final %.3f" % r, "first non-finite step:", (int(bad[0]) if len(bad) else None))
PY
done
