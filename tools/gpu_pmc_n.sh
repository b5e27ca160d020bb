#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cat > /tmp/runn.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
N = int(sys.argv[1])
a = Args(seed=0, env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=12, temp_sample=0.1, disable_recommended_params=True, not_render=True)
r, d = run_diffusion(a, return_details=True)
print("N", N, "steps/s", d["steps_per_sec"])
PY
cd /tmp && export TMPDIR=/tmp
for N in 1024 2048 4096 8192; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmcn_$N -o r -- python /tmp/runn.py $N > $OUT/pmcn_$N.log 2>&1
done
