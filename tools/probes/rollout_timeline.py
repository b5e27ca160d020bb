"""Per-workgroup timeline of the rollout kernel (start/end ticks, CU/SIMD placement) for a few N."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "model-based-diffusion_amd"))
import torch
from mbd_hip import _capi
from mbd_hip.envs import get_env
lib = _capi.load()
lib.mbd_debug_set_clock_buffer.argtypes = [C.c_void_p, C.c_void_p]
env = get_env("humanoidrun")
st = env.reset(_capi.prng_key(0))
for N in (1024, 4096):
    us = (torch.randn(N, 50, 17, device="cuda") * 0.3).clamp(-1, 1)
    env.rollout(st, us); torch.cuda.synchronize()
    G = N // 4  # wavefronts
    buf = torch.zeros(G * 6, dtype=torch.int64, device="cuda")
    lib.mbd_debug_set_clock_buffer(env.handle, C.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(st, us); e1.record(); torch.cuda.synchronize()
    lib.mbd_debug_set_clock_buffer(env.handle, None)
    b = buf.cpu().numpy().reshape(G, 6)
    t0, t1, hw = b[:, 0], b[:, 1], b[:, 2]
    tick = 1e-8  # s_memtime: 100 MHz
    start = (t0 - t0.min()) * tick * 1e6; dur = (t1 - t0) * tick * 1e6; end = (t1 - t0.min()) * tick * 1e6
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    per_simd = np.bincount(np.unique(key, return_inverse=True)[1])
    pro, s1, s2 = (b[:, 3] - t0) * tick * 1e6, (b[:, 4] - b[:, 3]) * tick * 1e6, (b[:, 5] - b[:, 4]) * tick * 1e6
    print(f"   prologue us: p50 {np.median(pro):.1f} max {pro.max():.1f} | first control step p50 {np.median(s1):.1f} | second p50 {np.median(s2):.1f}"
          f" | after the last control step to the end: {np.median(dur - pro - s1 - s2) :.1f} - 47 steps")
    print(f"N={N}: kernel {e0.elapsed_time(e1)*1e3:.0f} us | WG start us: p50 {np.median(start):.0f} p90 {np.percentile(start,90):.0f} max {start.max():.0f} | "
          f"WG dur us: min {dur.min():.0f} p50 {np.median(dur):.0f} max {dur.max():.0f} | last end {end.max():.0f} | SIMDs used {len(per_simd)} waves/SIMD max {per_simd.max()} "
          f"hist {np.bincount(per_simd)[1:].tolist()}")
