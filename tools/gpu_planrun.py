"""Loop seconds of mbd_plan_run per step (the C loop: no Python between the steps).  usage: gpu_planrun.py [env N Nd]..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.planners.mbd_planner import Args, Plan
a = sys.argv[1:] or ["humanoidrun", "8192", "100", "humanoidrun", "4096", "100", "humanoidrun", "1024", "100"]
for k in range(0, len(a) - 2, 3):
    name, N, Nd = a[k], int(a[k + 1]), int(a[k + 2])
    env = get_env(name)
    args = Args(env_name=name, Nsample=N, Hsample=50, Ndiffuse=Nd, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    st = env.reset(_capi.prng_key(1))
    ts = []
    for rep in range(4):
        p = Plan(env, args); p.set_state0(st)
        mu, rm, rf, secs = p.run(_capi.prng_key(5)); p.close()
        ts.append(secs / (Nd - 1) * 1e6)
    print(f"{name} N={N}: mbd_plan_run {min(ts[1:]):.1f} us/step ({1e6 / min(ts[1:]):.1f} steps/s)  rew_final {rf:.4f}")
