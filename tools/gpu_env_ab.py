"""Kernel time of mbd_env_rollout per env and batch size, head against every library under lib/variants (one process per
library, alternating): usage gpu_env_ab.py ENV N [ENV N ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "model-based-diffusion_amd"))
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
a = sys.argv[1:]
for k in range(0, len(a) - 1, 2):
    name, B = a[k], int(a[k + 1])
    env = get_env(name)
    st = env.reset(_capi.prng_key(1))
    g = np.random.default_rng(0)
    us = torch.tensor(np.clip(g.normal(size=(B, 50, env.action_size)) * 0.5, -1, 1).astype(np.float32), device="cuda")
    for _ in range(3): r = env.rollout(st, us)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        e0.record(); r = env.rollout(st, us); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("%%-16s B=%%5d  %%8.1f us  checksum %%.9g" %% (name, B, float(np.median(ts)), float(r.double().sum())))
''' % ROOT
V = os.path.join(ROOT, "model-based-diffusion_amd", "lib", "variants")
libs = [("head", "")] + [(f[len("libmbd_hip_"):-3], os.path.join(V, f)) for f in sorted(os.listdir(V)) if f.endswith(".so") and "plain" not in f]
for rnd in range(2):
    for tag, lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD] + sys.argv[1:], env=dict(os.environ, MBD_HIP_LIB=lib), capture_output=True, text=True)
        for l in out.stdout.splitlines():
            print("%-14s %s" % (tag, l))
        if out.returncode: print(out.stderr[-500:])
