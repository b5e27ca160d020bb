"""Fuzz beyond the committed seeds: random MJCF models (tests/random_models.py) through the general kernels against the checker,
rollouts bit for bit.  usage (GPU box): python tools/gpu_fuzz_models.py FIRST COUNT [planar | planar3d]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from random_models import stable_random_model
from test_random_models import _comp
from mbd_hip import _capi
from mbd_hip.envs.base import RigidBodyEnv
from oracle import oracle as orc_mod
orc_mod.build()
orc = orc_mod.Oracle("f32")
first, count = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "3d"
bad = refused = 0
kinds = {}
for seed in range(first, first + count):
    if mode == "3d":
        _, m = stable_random_model(seed, _comp)
    else:
        _, m = stable_random_model(seed, lambda x: _comp(x, env_name="halfcheetah", planar=None if mode == "planar" else False),
                                   planar=True, max_bodies=10)
    try:
        env = RigidBodyEnv("hopper" if mode == "3d" else "halfcheetah", model=m)
    except Exception as e:  # a shape the library refuses (says so)
        refused += 1
        print(seed, "refused:", str(e)[:120])
        continue
    st = env.reset(_capi.prng_key(seed))
    us = np.clip(np.random.default_rng(seed).normal(size=(21, 25, env.action_size)) * 0.6, -1.3, 1.3).astype(np.float32)
    got = env.rollout(st, us).cpu().numpy()
    ref = orc.rollout(m.to_struct(), np.asarray(st.pipeline_state, np.float32), us)
    if not np.array_equal(got, ref):
        bad += 1
        print(seed, "MISMATCH max|d|", np.abs(got - ref).max(), "links", m.n_links)
print(f"{mode} seeds {first}..{first + count - 1}: {bad} mismatches, {refused} refused")
