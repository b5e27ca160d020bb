#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|init sigma\|override"
import sys, time
sys.path.insert(0, "model-based-diffusion_amd")
import numpy as np
from mbd_hip.planners.mbd_planner import Args, run_diffusion
from mbd_hip.planners.path_integral import Args as PArgs, run_path_integral
rows = []
for env, kw in [("car2d", dict(Nsample=1024, enable_demo=True)), ("car2d", dict(Nsample=1024)), ("cartpole", {}), ("hopper", {}), ("walker2d", {}),
                ("halfcheetah", {}), ("ant", {}), ("humanoidstandup", {}), ("humanoidtrack", dict(enable_demo=True)), ("humanoidtrack", {}), ("humanoidrun", {})]:
    a = Args(seed=0, env_name=env, not_render=True, **kw)   # the reference's defaults incl. recommended overrides
    t = time.time(); r, d = run_diffusion(a, return_details=True); dt = time.time() - t
    print("%-16s demo=%d N=%5d Nd=%3d temp=%.2f: rew_final %8.3f  first/last step mean %7.3f -> %7.3f  finite=%s  %.1f steps/s (%.2f s total)" % (
        env, a.enable_demo, a.Nsample, a.Ndiffuse, a.temp_sample, r, d["rew_means"][0], d["rew_means"][-1], bool(np.isfinite(d["mu_0ts"]).all()), d["steps_per_sec"], dt))
for meth in ("mppi", "cma-es", "cem"):
    a = PArgs(seed=0, env_name="hopper", update_method=meth)
    r, d = run_path_integral(a, return_details=True)
    print("path_integral %-7s hopper: rew %.3f sigma_final %.4f" % (meth, r, d["sigma_final"]))
# the reference's own sweeps with its default arguments (mbd/scripts/run_mbd.py:17-64): 8 seeds / 8 temperatures
from mbd_hip.scripts import run_mbd
for env in ("hopper", "ant", "humanoidrun"):
    t = time.time(); rews, secs = run_mbd.run_multiple_seed(run_mbd.Args(algo="mbd", mode="seed", env_name=env)); dt = time.time() - t
    print("run_multiple_seed mbd %-12s: rews %s  (%.2f s of lockstep loop, %.2f s wall)" % (env, np.round(rews, 3), secs, dt))
rews, secs = run_mbd.run_multiple_seed(run_mbd.Args(algo="path_integral", update_method="cma-es", mode="seed", env_name="hopper"))
print("run_multiple_seed path_integral cma-es hopper: rews %s (%.2f s)" % (np.round(rews, 3), secs))
rews, best = run_mbd.run_multiple_temp(run_mbd.Args(algo="mbd", mode="temp", env_name="hopper"))
print("run_multiple_temp mbd hopper: best_temp %.2f" % best)
PY
