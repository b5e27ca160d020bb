"""Two candidates per lane (mbd_pk2.h, MBD_PK2=1) against one (MBD_PK2=0), same box, ONE process: the lever (mbd_debug_set) is read per
launch.  Kernel time of env.rollout (HIP events on the launch stream), median of 9, alternating; plus a bit-for-bit
comparison of the two kernels' rewards at every size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import numpy as np, torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
H = 50
cases = [("humanoidrun", B) for B in (1024, 2048, 4096, 8192, 16384, 32768)] + [("humanoidtrack", 8192), ("humanoidstandup", 8192), ("ant", 8192), ("ant", 16384)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for name, B in cases:
    env = get_env(name)
    st = env.reset(_capi.prng_key(1))
    g = np.random.default_rng(0)
    us = torch.tensor(np.clip(g.normal(size=(B, H, env.action_size)) * 0.4, -1, 1).astype(np.float32), device="cuda")
    ts = {"0": [], "1": []}
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(22):
        k = str(it & 1)
        _capi.debug_set("MBD_PK2", int(k))
        e0.record(); r = env.rollout(st, us); e1.record(); e1.synchronize()
        out[k] = r.cpu().numpy()
        if it >= 4:
            ts[k].append(e0.elapsed_time(e1) * 1e3)
    a, b = float(np.median(ts["0"])), float(np.median(ts["1"]))
    same = np.array_equal(out["0"], out["1"])
    print("%-16s B=%6d  one/lane %9.1f us   two/lane %9.1f us   x%.3f   bit-identical: %s" % (name, B, a, b, a / b, same), flush=True)
_capi.debug_set("MBD_PK2", -1)
