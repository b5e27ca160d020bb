"""Where the per-step host read's turnaround goes: host-side durations of the calls of a step driven like bench.py's
`value` leg (sample_rollout, score_update, the wait for the step's mean reward), per N."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.planners.mbd_planner import Args, Plan, HostProgress
dev = torch.device("cuda:0")
for N in [int(x) for x in (sys.argv[1:] or ["8192", "4096", "1024"])]:
    env = get_env("humanoidrun")
    ND = 100
    a = Args(env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=ND, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    plan = Plan(env, a); plan.set_state0(env.reset(_capi.prng_key(1)))
    lib = plan.lib
    HNu = 50 * 17
    Y = [torch.zeros(HNu, device=dev), torch.zeros(HNu, device=dev)]
    local = torch.zeros(N, device=dev)
    host = HostProgress(1, dev)
    rng = np.asarray(_capi.prng_key(7), np.uint32)
    stream = None
    keys = _capi.prng_split(rng, 2)
    T = {"sample_rollout": [], "score_update": [], "prepare": [], "wait": [], "step": []}
    i = ND - 1
    t_prev = None
    for it in range(80):
        rng, ks = keys[0], _capi.key_array(keys[1])
        keys = _capi.prng_split(rng, 2)
        _capi.check(lib.mbd_plan_prefetch_noise(plan.h, _capi.key_array(keys[1]), stream))
        t0 = time.perf_counter()
        _capi.check(lib.mbd_plan_sample_rollout(plan.h, i, ks, Y[0].data_ptr(), local.data_ptr(), None, stream))
        t1 = time.perf_counter()
        host.reset(0)
        _capi.check(lib.mbd_plan_score_update(plan.h, i, ks, Y[0].data_ptr(), local.data_ptr(), None, Y[1].data_ptr(), host.ptr(0), stream))
        t2 = time.perf_counter()
        Y.reverse()
        i = i - 1 if i > 1 else ND - 1
        host.wait(0)
        t3 = time.perf_counter()
        if it >= 20:
            T["sample_rollout"].append(t1 - t0); T["score_update"].append(t2 - t1); T["wait"].append(t3 - t2)
            if t_prev is not None: T["step"].append(t3 - t_prev)
        t_prev = t3
    torch.cuda.synchronize()
    print(f"N={N}: " + "  ".join(f"{k} {np.median(v) * 1e6:.1f} us" for k, v in T.items() if v))
    plan.close()
