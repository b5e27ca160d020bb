"""ctypes loader for the CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (model-based-diffusion_amd/) never does.  Models are passed in as ctypes structs laid out like
``mbd_model_t`` (include/mbd_hip.h); this module does not import the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")

LINK_STATE = 13


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (seconds). Building the checker is not using it."""
    libs = ["liboracle_f32.so", "liboracle_f64.so", "liboracle_f32_omp.so", "liboracle_count.so"]
    srcs = [os.path.join(_HERE, s) for s in ("mbd_oracle_core.c", "mbd_oracle_physics.c", "mbd_oracle_planar.h", "spec_math.h", "count_ops.cc")]
    srcs.append(os.path.join(_HERE, "..", "include", "mbd_hip.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    stale = force or any(
        not os.path.exists(os.path.join(_BUILD, l)) or os.path.getmtime(os.path.join(_BUILD, l)) < newest
        for l in libs)
    if stale:
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)


def count_substep(model_struct, state, action):
    """Op counts of ONE physics substep as executed by the restatement (oracle/count_ops.cc): a dict with
    add, mul, fma, div, sqrt, cmp and flops = add + mul + 2 fma + div + sqrt (F_sub(model))."""
    build()
    lib = C.CDLL(os.path.join(_BUILD, "liboracle_count.so"))
    lib.orc_count_substep.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, np.ctypeslib.ndpointer(np.uint64)]
    st = np.ascontiguousarray(state, np.float32).reshape(-1)
    out, cnt = np.zeros_like(st), np.zeros(6, np.uint64)
    lib.orc_count_substep(C.addressof(model_struct), st, np.ascontiguousarray(action, np.float32), out, cnt)
    d = dict(zip(("add", "mul", "fma", "div", "sqrt", "cmp"), (int(x) for x in cnt)))
    d["flops"] = d["add"] + d["mul"] + 2 * d["fma"] + d["div"] + d["sqrt"]
    return d, out


def depth_substeps(model_struct, state, action, n_sub: int = 8):
    """Dependency depth of n_sub consecutive substeps (oracle/count_ops.cc orc_depth_substeps): an array [n_sub][L + 1] —
    per link, and in the last column over all links — of the longest chain of dependent operations behind the state after
    each substep, in issue slots of the kernels' sequences (division 7, square root 5, everything else 1)."""
    build()
    lib = C.CDLL(os.path.join(_BUILD, "liboracle_count.so"))
    lib.orc_depth_substeps.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, np.ctypeslib.ndpointer(np.uint32)]
    L = int(model_struct.n_links)
    out = np.zeros((n_sub, L + 1), np.uint32)
    lib.orc_depth_substeps(C.addressof(model_struct), np.ascontiguousarray(state, np.float32).reshape(-1),
                           np.ascontiguousarray(action, np.float32), n_sub, out)
    return out


class Oracle:
    def __init__(self, variant: str = "f32"):
        path = os.path.join(_BUILD, f"liboracle_{variant}.so")
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        lib.orc_threefry2x32.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        lib.orc_prng_key.argtypes = [C.c_uint64, _u32p]
        lib.orc_prng_split.argtypes = [_u32p, C.c_int, C.c_int, _u32p]
        lib.orc_random_bits32.argtypes = [_u32p, C.c_int, C.c_uint64, C.c_uint64]
        lib.orc_random_bits32.restype = C.c_uint32
        lib.orc_uniform.argtypes = [_u32p, C.c_int, C.c_uint64, C.c_float, C.c_float, _f32p]
        lib.orc_erfinv_f32.argtypes = [C.c_float]
        lib.orc_erfinv_f32.restype = C.c_float
        lib.orc_normal.argtypes = [_u32p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
        lib.orc_schedule.argtypes = [C.c_float, C.c_float, C.c_int, _f32p, _f32p, _f32p]
        lib.orc_sample.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p,
                                   _f32p, C.c_void_p]
        lib.orc_score_update.argtypes = [C.c_int, C.c_int, _f32p, C.c_void_p, C.c_float, C.c_float, _f32p,
                                         _f32p, C.c_float, C.c_float, C.c_float, C.c_int, _f32p, _f32p]
        lib.orc_score_update.restype = C.c_float
        lib.orc_car2d_reward.argtypes = [_f32p]
        lib.orc_car2d_reward.restype = C.c_float
        lib.orc_car2d_reset.argtypes = [_f32p]
        lib.orc_car2d_step.argtypes = [_f32p, _f32p, _f32p]
        lib.orc_car2d_step.restype = C.c_float
        lib.orc_car2d_rollout.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_void_p]
        lib.orc_car2d_xref_logpd.argtypes = [_f32p, _f32p, C.c_int]
        lib.orc_car2d_xref_logpd.restype = C.c_float
        lib.orc_track_xref_logpd.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
        lib.orc_track_xref_logpd.restype = C.c_float
        lib.orc_env_step.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        lib.orc_env_step.restype = C.c_float
        lib.orc_substep.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        lib.orc_rollout.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_void_p,
                                    C.c_void_p]
        lib.orc_forward.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        lib.orc_link_positions.argtypes = [C.c_void_p, _f32p, _f32p]
        lib.orc_joint_angles.argtypes = [C.c_void_p, _f32p, _f32p]
        for fn in ("orc_sp_atan2", "orc_sp_asin", "orc_sp_exp", "orc_sp_log1p"):
            getattr(lib, fn).restype = C.c_float
        lib.orc_sp_atan2.argtypes = [C.c_float, C.c_float]
        lib.orc_sp_asin.argtypes = [C.c_float]
        lib.orc_sp_exp.argtypes = [C.c_float]
        lib.orc_sp_log1p.argtypes = [C.c_float]
        lib.orc_sp_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.orc_sp_rot.argtypes = [_f32p, _f32p, _f32p]
        lib.orc_sp_qmul.argtypes = [_f32p, _f32p, _f32p]
        lib.orc_sp_sum.argtypes = [_f32p, C.c_int]
        lib.orc_sp_sum.restype = C.c_float
        lib.orc_pi_update.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_float, _f32p, _f32p, _f32p, _f32p, _f32p]
        lib.orc_pi_update.restype = C.c_float
        lib.orc_mean_h.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.orc_real_bytes.restype = C.c_int
        lib.orc_model_bytes.restype = C.c_int

    # ---- PRNG -------------------------------------------------------------------------------------
    def threefry2x32(self, k0, k1, c0, c1):
        o0, o1 = C.c_uint32(), C.c_uint32()
        self.lib.orc_threefry2x32(k0, k1, c0, c1, C.byref(o0), C.byref(o1))
        return o0.value, o1.value

    def prng_key(self, seed: int) -> np.ndarray:
        k = np.zeros(2, np.uint32)
        self.lib.orc_prng_key(seed, k)
        return k

    def split(self, key, num=2, impl=1) -> np.ndarray:
        out = np.zeros((num, 2), np.uint32)
        self.lib.orc_prng_split(np.ascontiguousarray(key, np.uint32), num, impl, out)
        return out

    def uniform(self, key, size, lo, hi, impl=1) -> np.ndarray:
        out = np.zeros(size, np.float32)
        self.lib.orc_uniform(np.ascontiguousarray(key, np.uint32), impl, size, lo, hi, out)
        return out

    def normal(self, key, shape, impl=1) -> np.ndarray:
        size = int(np.prod(shape))
        out = np.zeros(size, np.float32)
        self.lib.orc_normal(np.ascontiguousarray(key, np.uint32), impl, 0, size, size, out)
        return out.reshape(shape)

    def erfinv(self, x: float) -> float:
        return self.lib.orc_erfinv_f32(x)

    # ---- planner algebra --------------------------------------------------------------------------
    def schedule(self, beta0, betaT, Nd):
        a, ab, s = (np.zeros(Nd, np.float32) for _ in range(3))
        self.lib.orc_schedule(beta0, betaT, Nd, a, ab, s)
        return a, ab, s

    def sample(self, key, impl, N, H, Nu, begin, count, sigma, Ybar, want_eps=False):
        Y0s = np.zeros((count, H, Nu), np.float32)
        eps = np.zeros((count, H, Nu), np.float32) if want_eps else None
        self.lib.orc_sample(np.ascontiguousarray(key, np.uint32), impl, N, H * Nu, begin, count, sigma,
                            np.ascontiguousarray(Ybar, np.float32), Y0s,
                            eps.ctypes.data if want_eps else None)
        return (Y0s, eps) if want_eps else Y0s

    def score_update(self, rews, Y0s, Ybar_i, alpha_i, ab_i, ab_im1, temp, lp_demo=None, rew_xref=0.0,
                     literal=True):
        N = rews.shape[0]
        HNu = int(np.prod(Ybar_i.shape))
        w = np.zeros(N, np.float32)
        out = np.zeros(Ybar_i.shape, np.float32)
        lp = None if lp_demo is None else np.ascontiguousarray(lp_demo, np.float32)
        m = self.lib.orc_score_update(N, HNu, np.ascontiguousarray(rews, np.float32),
                                      None if lp is None else lp.ctypes.data, rew_xref, temp,
                                      np.ascontiguousarray(Y0s, np.float32),
                                      np.ascontiguousarray(Ybar_i, np.float32), alpha_i, ab_i, ab_im1,
                                      int(literal), w, out)
        return out, w, m

    def pi_update(self, method, rews, Y0s, mu_t, sigma, temp):
        """path_integral.py update rules. method: 1 mppi, 2 cma-es, 3 cem. Returns (mu, sigma, weights, rew_mean)."""
        N = rews.shape[0]
        HNu = int(np.prod(mu_t.shape))
        w = np.zeros(N, np.float32)
        out = np.zeros(mu_t.shape, np.float32)
        sig = np.array([sigma], np.float32)
        m = self.lib.orc_pi_update(method, N, HNu, np.ascontiguousarray(rews, np.float32), temp,
                                   np.ascontiguousarray(Y0s, np.float32), np.ascontiguousarray(mu_t, np.float32),
                                   sig, w, out)
        return out, float(sig[0]), w, m

    # ---- car2d --------------------------------------------------------------------------------------
    def car2d_reset(self):
        q = np.zeros(3, np.float32)
        self.lib.orc_car2d_reset(q)
        return q

    def car2d_step(self, q, a):
        out = np.zeros(3, np.float32)
        r = self.lib.orc_car2d_step(np.ascontiguousarray(q, np.float32),
                                    np.ascontiguousarray(a, np.float32), out)
        return out, r

    def car2d_reward(self, q):
        return self.lib.orc_car2d_reward(np.ascontiguousarray(q, np.float32))

    def car2d_rollout(self, q0, us, want_qs=False):
        B, H, _ = us.shape
        rewss = np.zeros((B, H), np.float32)
        qs = np.zeros((B, H, 3), np.float32) if want_qs else None
        self.lib.orc_car2d_rollout(np.ascontiguousarray(q0, np.float32),
                                   np.ascontiguousarray(us, np.float32), B, H, rewss,
                                   qs.ctypes.data if want_qs else None)
        return (rewss, qs) if want_qs else rewss

    def car2d_xref_logpd(self, xs, xref):
        return self.lib.orc_car2d_xref_logpd(np.ascontiguousarray(xs, np.float32),
                                             np.ascontiguousarray(xref, np.float32), xs.shape[0])

    def track_xref_logpd(self, xpos, xref):
        H, K, _ = xpos.shape
        return self.lib.orc_track_xref_logpd(np.ascontiguousarray(xpos, np.float32),
                                             np.ascontiguousarray(xref, np.float32), H, K)

    # ---- rigid-body physics (model = ctypes struct laid out as mbd_model_t) ---------------------------
    def forward(self, model, q, qd):
        s = np.zeros((model.n_links, LINK_STATE), np.float32)
        self.lib.orc_forward(C.addressof(model), np.ascontiguousarray(q, np.float32),
                             np.ascontiguousarray(qd, np.float32), s)
        return s

    def env_step(self, model, state, action):
        out = np.zeros_like(state, dtype=np.float32)
        r = self.lib.orc_env_step(C.addressof(model), np.ascontiguousarray(state, np.float32),
                                  np.ascontiguousarray(action, np.float32), out)
        return out, r

    def substep(self, model, state, action):
        out = np.zeros_like(state, dtype=np.float32)
        self.lib.orc_substep(C.addressof(model), np.ascontiguousarray(state, np.float32),
                             np.ascontiguousarray(action, np.float32), out)
        return out

    def rollout(self, model, state0, us, want_xpos=False, want_final=False):
        B, H, _ = us.shape
        rewss = np.zeros((B, H), np.float32)
        xpos = np.zeros((B, H, model.n_track, 3), np.float32) if want_xpos else None
        fin = np.zeros((B, model.n_links, LINK_STATE), np.float32) if want_final else None
        self.lib.orc_rollout(C.addressof(model), np.ascontiguousarray(state0, np.float32),
                             np.ascontiguousarray(us, np.float32), B, H, rewss,
                             xpos.ctypes.data if want_xpos else None,
                             fin.ctypes.data if want_final else None)
        res = [rewss]
        if want_xpos:
            res.append(xpos)
        if want_final:
            res.append(fin)
        return res[0] if len(res) == 1 else tuple(res)

    def link_positions(self, model, state):
        out = np.zeros((model.n_links, 3), np.float32)
        self.lib.orc_link_positions(C.addressof(model), np.ascontiguousarray(state, np.float32), out)
        return out

    def joint_angles(self, model, state):
        out = np.zeros((model.n_links, 3), np.float32)
        self.lib.orc_joint_angles(C.addressof(model), np.ascontiguousarray(state, np.float32), out)
        return out
