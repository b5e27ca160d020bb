"""The reference's seed sweep (mbd/scripts/run_mbd.py:17-39: 8 plans, seeds 0..7) at the metric's sizes (humanoidrun
N=1024 H=50): plan-steps/s as ONE sweep (mbd_sweep_run: one rollout launch per step over the 8192 candidates) against
8 concurrent plans on 8 streams (round-robin from the host) and against the plans one after another.  Same box, one
process; MBD_PK2=0 in the environment keeps the sweep on the one-candidate-per-lane kernel (A/B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np
from mbd_hip import _capi
from mbd_hip.planners.mbd_planner import Args, run_diffusion
from mbd_hip.scripts.run_mbd import run_concurrent
env_name = sys.argv[1] if len(sys.argv) > 1 else "humanoidrun"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Nd = 100
plans = [Args(seed=s, env_name=env_name, Nsample=N, Hsample=50, Ndiffuse=Nd, temp_sample=0.1,
              disable_recommended_params=True, not_render=True) for s in range(P)]
out = {}
for label, kw, env in (("sweep (two candidates per lane)", dict(batched=True), {}),
                       ("sweep (one candidate per lane)", dict(batched=True), {"MBD_PK2": "0"}),
                       ("8 streams, round-robin", dict(batched=False), {})):
    for k, v in env.items():
        _capi.debug_set(k, int(v))
    best = None
    for rep in range(3):
        rews, mus, secs = run_concurrent(plans, **kw)
        best = secs if best is None or secs < best else best
    for k in env:
        _capi.debug_set(k, -1)
    out[label] = (best, rews, mus)
    print("%-36s %7.1f ms for %d plans x %d steps = %8.0f plan-steps/s   rew %.3f +- %.3f" %
          (label, best * 1e3, P, Nd - 1, P * (Nd - 1) / best, np.mean(rews), np.std(rews)), flush=True)
ref = out["8 streams, round-robin"]
for label, (_, rews, mus) in out.items():
    same = all(np.array_equal(a, b) for a, b in zip(mus, ref[2])) and np.array_equal(np.float32(rews), np.float32(ref[1]))
    print("  %-34s bit-identical to the concurrent plans: %s" % (label, same))
t0 = time.time()
r, det = run_diffusion(plans[0], return_details=True)
print("one plan alone: %.1f ms -> %.0f steps/s" % ((time.time() - t0) * 1e3, 0))
