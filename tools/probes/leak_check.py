import sys
sys.path.insert(0, "model-based-diffusion_amd")
import torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
from mbd_hip.planners.mbd_planner import Args, Plan
free0 = None
for it in range(60):
    env = get_env("humanoidrun")
    a = Args(env_name="humanoidrun", Nsample=4096, Hsample=50, Ndiffuse=20, disable_recommended_params=True, not_render=True)
    p = Plan(env, a, shard_begin=0, shard_count=512)
    p.set_state0(env.reset(_capi.prng_key(it)))
    Y = torch.zeros(850, device="cuda"); loc = torch.zeros(512, device="cuda")
    _capi.check(p.lib.mbd_plan_sample_rollout(p.h, 5, _capi.key_array(_capi.prng_key(1)), Y.data_ptr(), loc.data_ptr(), None, None))
    _capi.check(p.lib.mbd_plan_prefetch_noise(p.h, _capi.key_array(_capi.prng_key(2)), None))  # (declared after the fact: the next step generates its own)
    _capi.check(p.lib.mbd_plan_sample_rollout(p.h, 4, _capi.key_array(_capi.prng_key(2)), Y.data_ptr(), loc.data_ptr(), None, None))
    torch.cuda.synchronize()
    p.close(); env.close() if hasattr(env, "close") else None
    del p, env
    free, total = torch.cuda.mem_get_info()
    if it == 5: free0 = free
print("free after 5 iterations %.1f MB, after 60 %.1f MB, drift %.2f MB" % (free0 / 2**20, free / 2**20, (free0 - free) / 2**20))
