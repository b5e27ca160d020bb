"""Per-control-step overhead of the rollout kernels: kernel time against n_frames (a model parameter) at fixed B, H —
slope = one substep, intercept / H = what a control step costs outside its substeps (reward, action fetch, branches)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_model
from mbd_hip import _capi
from mbd_hip.envs.base import RigidBodyEnv
H = 50
for name, B in (("humanoidrun", 1024), ("hopper", 512), ("halfcheetah", 1024)):
    res = []
    for nf in (1, 2, 4, 8, 16):
        m = load_model(name)
        m.fields["n_frames"] = nf
        env = RigidBodyEnv(name, model=m)
        st = env.reset(_capi.prng_key(1))
        g = np.random.default_rng(0)
        us = torch.tensor(np.clip(g.normal(size=(B, H, env.action_size)) * 0.2, -1, 1).astype(np.float32), device="cuda")
        for _ in range(3):
            env.rollout(st, us)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(15):
            e0.record(); env.rollout(st, us); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        res.append((nf, float(np.median(ts))))
        del env
    x = np.array([r[0] for r in res], float); y = np.array([r[1] for r in res])
    k, c = np.polyfit(x, y, 1)
    print(name, " ".join("nf=%d:%.0fus" % r for r in res), "| per substep %.3f us, per control step outside substeps %.3f us (intercept %.0f us incl. ~35 us launch/prologue)" % (k / H, (c - 35.0) / H, c))
