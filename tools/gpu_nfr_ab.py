"""n_frames as a compile-time constant (NFR) against the run-time loops, per built-in model, in ONE process: the switch
MBD_NO_NFR_CONST is read per launch.  Kernel time of env.rollout at the BASELINE sizes, median of 15, alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
H = 50
for name, B in (("humanoidrun", 1024), ("humanoidtrack", 2048), ("humanoidstandup", 1024), ("ant", 1024), ("hopper", 512),
                ("walker2d", 1024), ("cartpole", 1024)):
    env = get_env(name)
    st = env.reset(_capi.prng_key(1))
    g = np.random.default_rng(0)
    us = torch.tensor(np.clip(g.normal(size=(B, H, env.action_size)) * 0.2, -1, 1).astype(np.float32), device="cuda")
    ts = {"0": [], "1": []}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(36):
        k = str(it & 1)
        _capi.debug_set("MBD_NO_NFR_CONST", int(k))
        e0.record(); env.rollout(st, us); e1.record(); e1.synchronize()
        if it >= 6:
            ts[k].append(e0.elapsed_time(e1) * 1e3)
    a, b = float(np.median(ts["0"])), float(np.median(ts["1"]))
    print("%-16s B=%4d  NFR constant %.1f us   run-time n_frames %.1f us   (%+.2f %%)" % (name, B, a, b, 100.0 * (b - a) / b))
_capi.debug_set("MBD_NO_NFR_CONST", -1)
