"""What does a rank pay for variant B's redundancy (every rank samples, scores and averages over ALL N_total
candidates so that ONE all-gather per step suffices)?  One rank of a G=8 job emulated on one GPU: a plan that owns
N_total/8 candidates; the all-gather is replaced by a device copy.  Prints, per N_total:
  a  ms per step of the sharded plan (the other ranks' rows sampled on the aux stream behind the rollout)
  a' the same with MBD_NO_AUX=1 semantics measured through fenced phases: phase 1 / exchange / phase 2 (HIP events)
  b  ms per step of an UNSHARDED plan of N_total/8 candidates (the floor: no redundant work at all)
and the exposed share (a - b) / a.  Run on the GPU box: python tools/gpu_exchange.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mbd_hip import _capi  # noqa: E402
from mbd_hip.envs import get_env  # noqa: E402
from mbd_hip.planners.mbd_planner import Args, Plan  # noqa: E402

G, H, ENV = 8, 50, "humanoidrun"
env = get_env(ENV)
st = env.reset(_capi.prng_key(1))
HNu = H * env.action_size


def loop(plan, N, sh, fenced):
    Ybar, Yn = torch.zeros(HNu, device="cuda"), torch.zeros(HNu, device="cuda")
    loc, allv, rm = torch.zeros(sh, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(1, device="cuda")
    ks = _capi.key_array(_capi.prng_key(5))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]

    def step(i):
        nonlocal Ybar, Yn
        if fenced:
            ev[0].record()
        _capi.check(plan.lib.mbd_plan_sample_rollout(plan.h, i, ks, Ybar.data_ptr(), loc.data_ptr(), None, None))
        if fenced:
            ev[1].record()
        allv.view(N // sh, sh).copy_(loc.unsqueeze(0).expand(N // sh, sh))
        if fenced:
            ev[2].record()
        _capi.check(plan.lib.mbd_plan_score_update(plan.h, i, ks, Ybar.data_ptr(), allv.data_ptr(), None, Yn.data_ptr(),
                                                   rm.data_ptr(), None))
        Ybar, Yn = Yn, Ybar
        if fenced:
            ev[3].record()
            ev[3].synchronize()
            for k in range(3):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
    for i in range(99, 89, -1):
        step(i)
    torch.cuda.synchronize()
    acc[:] = [0.0, 0.0, 0.0]
    t = time.perf_counter()
    for i in range(89, 29, -1):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 60
    return dt * 1e3, [a / 60 for a in acc]


for N in (4096, 8192):
    sh = N // G
    mk = lambda n, b, c: Plan(env, Args(env_name=ENV, Nsample=n, Hsample=H, Ndiffuse=100, temp_sample=0.1,
                                        disable_recommended_params=True, not_render=True), shard_begin=b, shard_count=c)
    p = mk(N, 0, sh)
    p.set_state0(st)
    a, _ = loop(p, N, sh, False)
    _, ph = loop(p, N, sh, True)
    p.close()
    q = mk(sh, 0, sh)
    q.set_state0(st)
    b, _ = loop(q, sh, sh, False)
    _, phb = loop(q, sh, sh, True)
    q.close()
    print(f"N_total={N} G={G} shard={sh}: sharded {a:.3f} ms/step | fenced phases: phase1 {ph[0]:.3f} exchange(copy) {ph[1]:.3f} "
          f"phase2 {ph[2]:.3f} | unsharded N={sh}: {b:.3f} ms/step (phase1 {phb[0]:.3f} phase2 {phb[2]:.3f}) | "
          f"exposed redundancy {(a - b) / a * 100:.1f} %")
