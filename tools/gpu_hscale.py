"""Fixed cost of a rollout launch (prologue: per-lane model constants; epilogue): kernel time against H at fixed B,
linear fit — the intercept is what a launch pays before and after its H x n_frames substeps."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
for name, B in (("humanoidrun", 1024), ("hopper", 512), ("halfcheetah", 1024)):
    env = get_env(name)
    st = env.reset(_capi.prng_key(1))
    g = np.random.default_rng(0)
    res = []
    for H in (1, 2, 5, 10, 25, 50, 100):
        us = torch.tensor(np.clip(g.normal(size=(B, H, env.action_size)) * 0.3, -1, 1).astype(np.float32), device="cuda")
        for _ in range(3):
            env.rollout(st, us)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(); env.rollout(st, us); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        res.append((H, float(np.median(ts))))
    Hs = np.array([r[0] for r in res], float); T = np.array([r[1] for r in res])
    k, c = np.polyfit(Hs[2:], T[2:], 1)
    print(name, "B=%d" % B, " ".join("H=%d:%.1fus" % r for r in res), "| per control step %.2f us, intercept %.1f us" % (k, c))
