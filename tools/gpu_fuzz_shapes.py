"""Fuzz of plan shapes: one diffusion step (mbd_plan_reverse_once) at tiny and random (N, H, Ndiffuse), both threefry layouts,
against the checker — candidates, rewards, weights, Ybar, key, bit for bit (tests/test_gpu_parity._one_step).  GPU box only."""
import sys, os
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import test_gpu_parity as T
from mbd_hip import _capi
from oracle import oracle as orc_mod
orc_mod.build(); orc = orc_mod.Oracle("f32")
gpu = _capi
gpu.check = _capi.check
bad = 0
g = np.random.default_rng(0)
cases = [("humanoidrun", 1, 1, 2), ("humanoidrun", 2, 1, 3), ("humanoidrun", 3, 2, 5), ("humanoidrun", 5, 51, 4), ("hopper", 1, 1, 2), ("hopper", 17, 3, 7),
         ("car2d", 1, 1, 2), ("car2d", 2, 3, 3), ("halfcheetah", 7, 2, 3), ("ant", 1, 2, 2), ("humanoidstandup", 9, 1, 2), ("walker2d", 33, 5, 3), ("cartpole", 1, 1, 2)]
for _ in range(40):
    cases.append((str(g.choice(["humanoidrun", "hopper", "halfcheetah", "ant", "car2d", "walker2d", "humanoidstandup", "cartpole"])), int(g.integers(1, 400)), int(g.integers(1, 64)), int(g.integers(2, 9))))
for name, N, H, Nd in cases:
    for impl in ((0, 1) if N * H < 2000 else (1,)):
        os.environ["MBD_THREEFRY_PARTITIONABLE"] = str(impl)
        try:
            T._one_step(gpu, orc, name, N, H, Nd, 0.1, impl, False, i=int(g.integers(1, Nd)))
        except AssertionError as e:
            bad += 1; print("MISMATCH", name, N, H, Nd, impl, str(e)[:100])
        except Exception as e:
            bad += 1; print("ERROR", name, N, H, Nd, impl, type(e).__name__, str(e)[:160])
print("cases", len(cases), "bad", bad)
