"""Candidates per wavefront (MBD_CPW) against the plan's size, per planar env: steps/s of a 40-step plan.  usage (GPU box):
python tools/gpu_cpw_sizes.py   (spawns one process per (env, N, cpw): the lever is read when the library loads)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "model-based-diffusion_amd"))
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.planners.mbd_planner import Args, Plan
name, N, temp = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
env = get_env(name)
st = env.reset(_capi.prng_key(1))
best = 0.0
for rep in range(3):
    p = Plan(env, Args(env_name=name, Nsample=N, Hsample=50, Ndiffuse=41, temp_sample=temp, disable_recommended_params=True, not_render=True))
    p.set_state0(st)
    secs = p.run(_capi.prng_key(3))[3]
    p.close()
    best = max(best, 40.0 / secs)
print(f"{best:.0f}")
"""
print("| env | N | " + " | ".join(f"MBD_CPW={c}" for c in (0, 1, 2, 4)) + " |")
print("|---|---:|" + "---:|" * 4)
for name, temp in (("hopper", 0.1), ("walker2d", 0.1), ("halfcheetah", 0.4)):
    for N in (128, 256, 512, 1024, 2048, 4096):
        row = []
        for cpw in (0, 1, 2, 4):
            env = dict(os.environ, MBD_CPW=str(cpw))
            r = subprocess.run([sys.executable, "-c", ONE, ROOT, name, str(N), str(temp)], env=env, capture_output=True, text=True)
            row.append(r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "err")
        print(f"| {name} | {N} | " + " | ".join(row) + " |", flush=True)
