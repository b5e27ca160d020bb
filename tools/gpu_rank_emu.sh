#!/bin/bash
# One rank of a G-GPU weak-scaling run emulated on one GPU: a plan that owns 1024 of G*1024 candidates, the
# all-gather replaced by a device copy of its own rewards into every shard slot.  Step time vs G, with the other
# ranks' rows sampled behind the rollout (default) or on the same stream (MBD_NO_AUX=1).
cd "$GRAFT_REPO_ROOT" || exit 1
for NOAUX in 0 1; do
MBD_NO_AUX=$NOAUX python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, "model-based-diffusion_amd")
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.planners.mbd_planner import Args, Plan
env = get_env("humanoidrun")
st = env.reset(_capi.prng_key(1))
for G in (1, 2, 4, 8):
    N = 1024 * G
    a = Args(env_name="humanoidrun", Nsample=N, Hsample=50, Ndiffuse=100, temp_sample=0.1, disable_recommended_params=True, not_render=True)
    plan = Plan(env, a, shard_begin=0, shard_count=1024)
    plan.set_state0(st)
    HNu = 50 * 17
    Ybar, Yn = torch.zeros(HNu, device="cuda"), torch.zeros(HNu, device="cuda")
    loc, allv, rm = torch.zeros(1024, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(1, device="cuda")
    ks = _capi.key_array(_capi.prng_key(5))
    def step(i):
        global Ybar, Yn
        _capi.check(plan.lib.mbd_plan_sample_rollout(plan.h, i, ks, Ybar.data_ptr(), loc.data_ptr(), None, None))
        allv.view(G, 1024).copy_(loc.unsqueeze(0).expand(G, 1024))
        _capi.check(plan.lib.mbd_plan_score_update(plan.h, i, ks, Ybar.data_ptr(), allv.data_ptr(), None, Yn.data_ptr(), rm.data_ptr(), None))
        Ybar, Yn = Yn, Ybar
    for i in range(99, 89, -1): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(89, 29, -1): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 60
    print("MBD_NO_AUX=%s G=%d (N_total=%5d): %.3f ms per step" % (os.environ["MBD_NO_AUX"], G, N, dt * 1e3))
    plan.close()
PY
done
