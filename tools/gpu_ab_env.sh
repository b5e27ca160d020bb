#!/bin/bash
# same-box A/B of library variants on another env: MBD_AB_ENV (default hopper), reference default arguments
cd "$GRAFT_REPO_ROOT" || exit 1
E=${MBD_AB_ENV:-hopper}
for round in 1 2 3; do
for lib in "" $(ls $GRAFT_REPO_ROOT/model-based-diffusion_amd/lib/variants/*.so 2>/dev/null); do
MBD_HIP_LIB=$lib MBD_AB_ENV=$E python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|override\|init sigma"
import os, sys
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
a = Args(seed=0, env_name=os.environ["MBD_AB_ENV"], not_render=True)
r, d = run_diffusion(a, return_details=True)
print("%-10s %-12s %.1f steps/s" % (os.path.basename(os.environ.get("MBD_HIP_LIB", "")) or "head", a.env_name, d["steps_per_sec"]))
PY
done; done
