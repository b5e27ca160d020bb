"""CPU restatement of run_diffusion / reverse_once (mbd/planners/mbd_planner.py:38-182) on top of the C
oracle — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg)."""
from __future__ import annotations

import numpy as np

from .oracle import Oracle


class OracleEnv:
    """Minimal env adaptor: kind 'car2d' or a compiled model struct (ctypes, laid out as mbd_model_t)."""

    def __init__(self, orc: Oracle, name: str, model_struct=None, xref=None, rew_xref: float = 0.0,
                 init_q=None):
        self.orc, self.name, self.ms, self.xref, self.rew_xref = orc, name, model_struct, xref, rew_xref
        self.init_q = init_q
        self.Nu = 2 if name == "car2d" else model_struct.n_act

    def reset_qqd(self, key, impl):
        """(q, qd) the env's reset hands to pipeline_init: humanoidrun.py:19-27 / hopper.py:20-28 / walker2d.py:19-27 /
        humanoidstandup.py:19-27 / cartpole.py:20-28 (its [0, pi] offset is part of the compiled init_q) /
        humanoidtrack.py:48-52 (deterministic); brax ant / half_cheetah: qvel = hi * normal(rng2)"""
        m = self.ms
        q = np.array(self.init_q, np.float32)
        qd = np.zeros(m.n_qd, np.float32)
        s = np.float32(m.reset_noise)
        if s > 0:
            keys = self.orc.split(key, 3, impl)
            q = (q + self.orc.uniform(keys[1], m.n_q, -s, s, impl)).astype(np.float32)
            if self.name in ("halfcheetah", "ant"):
                qd = (s * self.orc.normal(keys[2], (m.n_qd,), impl)).astype(np.float32)
            else:
                qd = self.orc.uniform(keys[2], m.n_qd, -s, s, impl)
        return q, qd

    def reset(self, key, impl):
        """humanoidrun.py:19-32 / hopper.py:20-34 / humanoidtrack.py:48-61 / car2d.py:73-75"""
        if self.name == "car2d":
            return self.orc.car2d_reset()
        q, qd = self.reset_qqd(key, impl)
        return self.orc.forward(self.ms, q, qd)

    def rollout(self, state0, us, want_xpos=False):
        if self.name == "car2d":
            return self.orc.car2d_rollout(state0, us, want_qs=want_xpos)
        return self.orc.rollout(self.ms, state0, us, want_xpos=want_xpos)

    def logpd(self, xpos):
        if self.name == "car2d":
            return np.array([self.orc.car2d_xref_logpd(x, self.xref) for x in xpos], np.float32)
        return np.array([self.orc.track_xref_logpd(x, self.xref) for x in xpos], np.float32)


def mean_h(orc: Oracle, rewss):
    out = np.zeros(rewss.shape[0], np.float32)
    rewss = np.ascontiguousarray(rewss, np.float32)
    orc.lib.orc_mean_h(rewss.ctypes.data, rewss.shape[0], rewss.shape[1], out.ctypes.data)
    return out


def reverse_once(orc: Oracle, env: OracleEnv, state0, i, rng, Ybar_i, sched, N, H, temp, impl,
                 enable_demo=False, literal=True, timers=None):
    """One step of mbd_planner.py:97-135. Returns (rng', Ybar_im1, rew_mean, details).  ``timers``: a dict that
    accumulates the seconds of the three phases (sample / rollout / score) — bench.py's cpu_baseline reports them."""
    import time
    alphas, alphas_bar, sigmas = sched
    t0 = time.perf_counter()
    keys = orc.split(rng, 2, impl)  # rng, Y0s_rng = split(rng)  (:103)
    rng, ks = keys[0], keys[1]
    Y0s = orc.sample(ks, impl, N, H, env.Nu, 0, N, float(sigmas[i]), Ybar_i)  # :104-106
    t1 = time.perf_counter()
    lp = xpos = None
    if enable_demo:
        rewss, xpos = env.rollout(state0, Y0s, want_xpos=True)  # :109
        lp = env.logpd(xpos)  # :118
    else:
        rewss = env.rollout(state0, Y0s)
    t2 = time.perf_counter()
    rews = mean_h(orc, np.ascontiguousarray(rewss))  # :110
    Ybar_im1, w, rew_mean = orc.score_update(rews, Y0s, Ybar_i, float(alphas[i]), float(alphas_bar[i]),
                                             float(alphas_bar[i - 1]), temp, lp_demo=lp,
                                             rew_xref=env.rew_xref, literal=literal)  # :111-135
    if timers is not None:
        t3 = time.perf_counter()
        for k, v in (("sample", t1 - t0), ("rollout", t2 - t1), ("score", t3 - t2)):
            timers[k] = timers.get(k, 0.0) + v
    return rng, Ybar_im1, rew_mean, dict(Y0s=Y0s, rewss=rewss, rews=rews, weights=w, lp=lp, xpos=xpos)


def run_diffusion(orc: Oracle, env: OracleEnv, seed, N, H, Nd, temp, beta0=1e-4, betaT=1e-2, impl=1,
                  enable_demo=False, literal=True, max_steps=None):
    """mbd_planner.py:38-182 (RNG chain :40,79,150). Returns dict(mu_0ts, rew_means, rew_final, state_init)."""
    rng = orc.prng_key(seed)
    rng, rng_reset = orc.split(rng, 2, impl)
    state0 = env.reset(rng_reset, impl)
    sched = orc.schedule(beta0, betaT, Nd)
    rng_exp, rng = orc.split(rng, 2, impl)
    Ybar = np.zeros((H, env.Nu), np.float32)
    mus, rms = [], []
    r = rng_exp
    steps = 0
    for i in range(Nd - 1, 0, -1):
        r, Ybar, rm, _ = reverse_once(orc, env, state0, i, r, Ybar, sched, N, H, temp, impl, enable_demo, literal)
        mus.append(Ybar)
        rms.append(rm)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
    rew_final = mean_h(orc, np.ascontiguousarray(env.rollout(state0, Ybar[None])))[0]  # :179-180
    return dict(mu_0ts=np.stack(mus), rew_means=np.array(rms, np.float32), rew_final=rew_final,
                state_init=state0)


PI_METHODS = {"mppi": 1, "cma-es": 2, "cem": 3}


def run_path_integral(orc: Oracle, env: OracleEnv, seed, N, H, Nrefine, temp, update_method="mppi", impl=1,
                      max_steps=None):
    """mbd/planners/path_integral.py:55-148 (RNG chain :57,99,144; sigma starts at 1.0 :131)."""
    rng = orc.prng_key(seed)
    rng, rng_reset = orc.split(rng, 2, impl)
    state0 = env.reset(rng_reset, impl)
    rng_exp, rng = orc.split(rng, 2, impl)
    mu = np.zeros((H, env.Nu), np.float32)
    sigma = np.float32(1.0)
    mus, rms, sigmas = [], [], []
    r = rng_exp
    steps = 0
    for t in range(Nrefine - 1, 0, -1):
        keys = orc.split(r, 2, impl)
        r, ks = keys[0], keys[1]
        Y0s = orc.sample(ks, impl, N, H, env.Nu, 0, N, float(sigma), mu)  # :115-118
        rews = mean_h(orc, np.ascontiguousarray(env.rollout(state0, Y0s)))  # :121
        mu, sigma, _, rm = orc.pi_update(PI_METHODS[update_method], rews, Y0s, mu, float(sigma), temp)  # :122-125
        mus.append(mu)
        rms.append(rm)
        sigmas.append(sigma)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
    rew_final = mean_h(orc, np.ascontiguousarray(env.rollout(state0, mu[None])))[0]  # :146
    return dict(mu_0ts=np.stack(mus), rew_means=np.array(rms, np.float32), sigmas=np.array(sigmas, np.float32),
                rew_final=rew_final, state_init=state0)
