"""Device-memory drift over create / run / destroy cycles of the handles round 4 added or changed: MBD sweeps, path-integral
sweeps (all three update rules), envs on the general instantiation (a random custom model), the exchange (world 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.envs.base import RigidBodyEnv
from mbd_hip.planners.mbd_planner import Args, Sweep
from mbd_hip.planners import path_integral
from random_models import stable_random_model
from test_random_models import _comp
free0 = None
for it in range(40):
    env = get_env("humanoidrun")
    a = Args(env_name="humanoidrun", Nsample=256, Hsample=20, Ndiffuse=6, disable_recommended_params=True, not_render=True)
    keys = np.array([_capi.prng_key(k) for k in range(4)], np.uint32)
    for um in (0, 1, 2, 3):
        args = a if um == 0 else path_integral.Args(env_name="humanoidrun", Nsample=256, Hsample=20, Nrefine=6, disable_recommended_params=True)
        sw = Sweep(env, args, 4, update_method=um)
        for k in range(4):
            sw.set_state0(k, env.reset(_capi.prng_key(k)))
        sw.run(keys)
        sw.close()
    _, m = stable_random_model(it % 8, _comp)
    e2 = RigidBodyEnv("hopper", model=m)
    us = np.zeros((8, 5, e2.action_size), np.float32)
    e2.rollout(e2.reset(_capi.prng_key(it)), us)
    for e in (env, e2):
        e.close() if hasattr(e, "close") else None
    del env, e2, sw
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 5:
        free0 = free
print("free after 5 iterations %.1f MB, after 40 %.1f MB, drift %.2f MB" % (free0 / 2**20, free / 2**20, (free0 - free) / 2**20))
