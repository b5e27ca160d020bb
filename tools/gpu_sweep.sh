#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests -m gpu -x -q -k "concurrent" 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time
sys.path.insert(0, "model-based-diffusion_amd")
from mbd_hip.planners.mbd_planner import Args, run_diffusion
from mbd_hip.scripts.run_mbd import run_concurrent
mk = lambda s: Args(seed=s, env_name="humanoidrun", Nsample=1024, Hsample=50, Ndiffuse=100, temp_sample=0.1, disable_recommended_params=True, not_render=True)
run_concurrent([mk(0)])
t=time.time(); seq=[run_diffusion(mk(s)) for s in range(8)]; tseq=time.time()-t
rews, mus, secs = run_concurrent([mk(s) for s in range(8)])
print("8 seeds sequential: %.3f s (incl. env/plan setup) | concurrent reverse loops: %.3f s -> %.0f plan-steps/s" % (tseq, secs, 8*99/secs))
print("rewards equal:", [float(a)==float(b) for a,b in zip(seq,rews)], "mean %.3f" % (sum(rews)/8))
PY
