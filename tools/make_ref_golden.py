"""Golden vectors from the REFERENCE'S OWN SOURCE, executed — not restated.

jax, brax, flax, mujoco and tyro are absent from every box this repo builds or runs on, so the reference cannot run as it
is.  Its in-tree, Brax-free path can be EXECUTED, though, if something stands in for the array runtime: this script loads
`/root/reference/mbd/planners/mbd_planner.py` and `mbd/envs/car2d.py` UNCHANGED (by import, from where they lie) under a small
numpy stand-in for the `jax` / `jax.numpy` / `flax.struct` names they use (float32 everywhere, python scalars weakly typed —
numpy 2's promotion rules are JAX's here), runs `run_diffusion(Args(env_name="car2d", ...))` and records what every call of
the reference's `reverse_once` took and returned, plus the values in between (normals, candidates' rewards, softmax weights).

What is the reference's: the planner algebra (mbd_planner.py:84-135,138-151,179-180), the rollout (mbd/utils.py:14-20), the
car2d env (car2d.py:10-102) — every line of them runs.  What is NOT: (1) `jax.random.{PRNGKey, split, normal}` are served by
this repo's threefry / ErfInv restatement (oracle/, pinned by Random123 KATs and by values the real JAX prints in its docs);
(2) numpy evaluates sin / cos / exp with libm-grade routines and sums pairwise where XLA has its own polynomials and tree
reductions: values agree to float32 round-off, not bit for bit.  So these fixtures pin the contract at ~1e-6, which is the
tolerance tests/test_ref_golden.py uses and states.  They are NOT outputs of JAX; no fixture claims to be.

    python tools/make_ref_golden.py [/root/reference]      ->  tests/golden/ref_car2d_*.npz
"""
import dataclasses
import importlib
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # the reference tree is read-only to this repo: importing from it must not leave __pycache__ there

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

REC = {"steps": [], "cur": {}}


# ---- float32 arrays with jax's `.at[idx].set(v)` ------------------------------------------------------------------------
class F32(np.ndarray):
    @property
    def at(self):
        arr = self

        class _At:
            def __getitem__(self, idx):
                class _Set:
                    def set(self, v):
                        out = np.array(arr, copy=True)
                        out[idx] = v
                        return _w(out)
                return _Set()
        return _At()


def _w(x):
    a = np.asarray(x)
    if a.dtype.kind == "f" and a.dtype != np.float32:
        a = a.astype(np.float32)
    return a.view(F32) if a.ndim else a[()]


def _wrapfn(fn):
    def g(*a, **k):
        return _w(fn(*a, **k))
    return g


def make_jnp():
    m = types.ModuleType("jax.numpy")
    for name in ("sqrt", "roll", "clip", "where", "einsum", "concatenate", "sin", "cos", "any", "diff", "arctan2", "append",
                 "cumprod", "exp", "abs", "sum", "mean", "stack", "maximum", "minimum", "arange", "argsort"):
        setattr(m, name, _wrapfn(getattr(np, name)))
    m.pi = np.pi
    m.ndarray = np.ndarray
    m.float32 = np.float32
    m.array = lambda x, dtype=None: _w(np.array([np.asarray(e) for e in x]) if isinstance(x, (list, tuple)) else np.asarray(x))
    m.zeros = lambda shape, dtype=None: _w(np.zeros(shape, np.float32))
    m.ones = lambda shape, dtype=None: _w(np.ones(shape, np.float32))
    m.linspace = lambda a, b, n: _w(np.linspace(a, b, n))          # (float64 inside, rounded once: not jax's f32 formula)
    m.load = lambda path: _w(np.load(path))                         # (x64 is off in the reference: float64 files arrive as f32)
    m.save = lambda path, a: np.save(path, np.asarray(a))
    la = types.ModuleType("jax.numpy.linalg")
    la.norm = _wrapfn(np.linalg.norm)
    m.linalg = la
    return m


class _Node:
    """a pytree node of the stand-in (brax.envs.base.State, brax.base.State / Transform / Motion look-alikes): an attribute bag
    with flax's .replace; stacked, indexed and measured field by field (what lax.scan / vmap do to pytrees)"""
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def replace(self, **kw):
        d = dict(self.__dict__)
        d.update(kw)
        return type(self)(**d)

    def __len__(self):
        return len(next(v for v in self.__dict__.values() if isinstance(v, (np.ndarray, _Node))))

    def __getitem__(self, k):
        return type(self)(**{n: (v[k] if isinstance(v, (np.ndarray, _Node)) else v) for n, v in self.__dict__.items()})


class BraxEnvState(_Node):
    """brax.envs.base.State(pipeline_state, obs, reward, done, metrics, info)"""
    def __init__(self, pipeline_state=None, obs=None, reward=None, done=None, metrics=None, info=None):
        super().__init__(pipeline_state=pipeline_state, obs=obs, reward=reward, done=done,
                         metrics={} if metrics is None else metrics, info={} if info is None else info)


def _tree_stack(items):
    x = items[0]
    if isinstance(x, tuple):
        return tuple(_tree_stack([it[k] for it in items]) for k in range(len(x)))
    if isinstance(x, _Node):
        return type(x)(**{n: _tree_stack([getattr(it, n) for it in items]) for n in x.__dict__})
    if isinstance(x, dict):
        return {n: _tree_stack([it[n] for it in items]) for n in x}
    if x is None:
        return None
    return _w(np.stack([np.asarray(it) for it in items]))


def make_jax(orc, impl):
    jax = types.ModuleType("jax")
    jnp = make_jnp()
    jax.numpy = jnp
    jax.Array = np.ndarray

    def jit(f=None, **kw):
        if f is None:
            return lambda g: jit(g, **kw)
        if getattr(f, "__name__", "") == "reverse_once":   # the reference's step: record what goes in and what comes out
            def wrapped(carry, unused):
                i, rng, Ybar_i = carry
                REC["cur"] = {"i": int(i), "rng_in": np.array(rng, np.uint32), "Ybar_i": np.array(Ybar_i, np.float32)}
                (i2, rng2, Yn), rew = f(carry, unused)
                REC["cur"].update(rng_out=np.array(rng2, np.uint32), Ybar_im1=np.array(Yn, np.float32),
                                  rew_mean=np.float32(rew))
                REC["steps"].append(REC["cur"])
                return (i2, rng2, Yn), rew
            return wrapped
        if getattr(f, "__name__", "") == "update_once":     # path_integral.py:111-127: the same, for the baselines' step
            def wrapped_pi(carry, unused):
                t, rng, mu, sigma = carry
                REC["cur"] = {"t": int(t), "rng_in": np.array(rng, np.uint32), "mu_in": np.array(mu, np.float32),
                              "sigma_in": np.float32(sigma)}
                (t2, rng2, mu2, sigma2), rew = f(carry, unused)
                REC["cur"].update(rng_out=np.array(rng2, np.uint32), mu_out=np.array(mu2, np.float32),
                                  sigma_out=np.float32(sigma2), rew_mean=np.float32(rew))
                REC["steps"].append(REC["cur"])
                return (t2, rng2, mu2, sigma2), rew
            return wrapped_pi
        return f
    jax.jit = jit

    def vmap(f, in_axes=0):
        def g(*args):
            axes = in_axes if isinstance(in_axes, tuple) else (in_axes,) * len(args)
            n = next(len(a) for a, ax in zip(args, axes) if ax is not None)
            outs = [f(*[(a if ax is None else a[k]) for a, ax in zip(args, axes)]) for k in range(n)]
            out = _tree_stack(outs)
            if isinstance(in_axes, tuple) and in_axes == (None, 0) and not isinstance(out, tuple) and "t" in REC["cur"]:
                REC["cur"]["rewss"] = np.array(out, np.float32)                                   # vmap(eval_us)
            if isinstance(in_axes, tuple) and in_axes == (None, 0) and isinstance(out, tuple):   # vmap(rollout_us)
                REC["cur"]["rewss"] = np.array(out[0], np.float32)
                REC["cur"]["qs"] = np.array(out[1].x.pos if isinstance(out[1], _Node) else out[1], np.float32)
            return out
        return g
    jax.vmap = vmap
    lax = types.ModuleType("jax.lax")

    def scan(step, init, xs):
        carry, ys = init, []
        for t in range(len(xs)):
            carry, y = step(carry, xs[t])
            ys.append(y)
        return carry, _tree_stack(ys)
    lax.scan = scan
    jax.lax = lax
    rnd = types.ModuleType("jax.random")
    rnd.PRNGKey = lambda seed=0: orc.prng_key(int(seed))
    rnd.split = lambda key, num=2: orc.split(np.asarray(key, np.uint32), num, impl)

    def normal(key, shape):
        eps = _w(orc.normal(np.asarray(key, np.uint32), tuple(shape), impl))
        REC["cur"]["eps"] = np.array(eps, np.float32)
        return eps
    rnd.normal = normal
    rnd.uniform = lambda key, shape, minval=0.0, maxval=1.0: _w(orc.uniform(np.asarray(key, np.uint32), int(np.prod(shape)),
                                                                          float(minval), float(maxval), impl).reshape(shape))
    jax.random = rnd
    nn = types.ModuleType("jax.nn")

    def softmax(x):
        x = np.asarray(x, np.float32)
        e = np.exp(x - x.max())
        w = _w(e / e.sum())
        REC["cur"]["logp0"], REC["cur"]["weights"] = np.array(x, np.float32), np.array(w, np.float32)
        return w
    nn.softmax = softmax
    jax.nn = nn
    cfg = types.SimpleNamespace(update=lambda *a, **k: None)
    jax.config = cfg
    return jax, jnp


class _AnyBase:
    def __init__(self, *a, **k):
        pass


class _Any(types.ModuleType):
    """a module whose every attribute exists: classes usable as bases, callables returning None (brax, etils, tyro, ...)"""
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = sys.modules.get(f"{self.__name__}.{name}")
        if sub is not None:
            return sub
        return type(name, (_AnyBase,), {})


class _Sys(types.SimpleNamespace):
    def replace(self, **kw):      # (cartpole.py:18 sys.replace(dt=0.005))
        d = dict(self.__dict__)
        d.update(kw)
        return _Sys(**d)


def _fake_mjcf_load(path):
    """stands in for brax.io.mjcf.load where an env's constructor only asks the system for its link NAMES
    (humanoidtrack.py:26-31): bodies that own a joint, in document order — Brax's link order"""
    import xml.etree.ElementTree as ET
    names = []
    if not os.path.exists(str(path)):   # (hopper.py:13 / walker2d.py:14 load their XML from inside the Brax wheel)
        return _compiled_sys(path, None) if BRAX["orc"] is not None else _Sys(link_names=names)

    def walk(e):
        for b in e.findall("body"):
            if b.find("joint") is not None or b.find("freejoint") is not None:
                names.append(b.get("name"))
            walk(b)
    walk(ET.parse(str(path)).getroot().find("worldbody"))
    return _compiled_sys(path, names) if BRAX["orc"] is not None else _Sys(link_names=names)


BRAX = {"orc": None, "last_init": None}


def _compiled_sys(path, names_from_xml):
    """what brax.io.mjcf.load returns, as far as the reference's wrappers ask (init_q, sizes, link names, dt): THIS repo's
    compiled model of the file of that name.  Not the reference's code — the physics underneath it."""
    from mbd_hip.model import Model
    name = os.path.splitext(os.path.basename(str(path)))[0]
    with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")) as f:
        m = Model.from_json(f.read())
    init_q = m.init_q.copy()
    if name == "cartpole":       # (the compiled model carries the reset offset of cartpole.py:26 in init_q already)
        init_q = init_q - np.array([0.0, np.pi], np.float32)
    names = names_from_xml or list(m.link_names)
    return _Sys(link_names=names, init_q=_w(init_q), q_size=lambda: m.q_size(), qd_size=lambda: m.qd_size(),
                act_size=lambda: m.act_size(), dt=float(m.fields["dt"]), _model=m, _ms=m.to_struct(), _name=name,
                _slot=[names.index(n) for n in m.link_names])


def _pstate(sysobj, st):
    """brax.base.State of a checker state [L, 13]: x.pos = the link frames' origins, x.rot, xd.vel = the origins' velocities
    (v_com - w x (R com)), xd.ang; listed in the XML's link order (marker bodies this repo's models drop keep zeros); q / qd
    only where a wrapper's REWARD reads them (cartpole.py:44: hinge angle of the pole, slide velocity of the cart)"""
    orc, ms, m = BRAX["orc"], sysobj._ms, sysobj._model
    o = orc.link_positions(ms, st)
    vel = (st[:, 7:10] - np.cross(st[:, 10:13], st[:, 0:3] - o)).astype(np.float32)
    n = len(sysobj.link_names)
    full = lambda a, w: (lambda z: (z.__setitem__(sysobj._slot, a), _w(z))[1])(np.zeros((n, w), np.float32))
    q, qd = np.zeros(m.q_size(), np.float32), np.zeros(m.qd_size(), np.float32)
    if sysobj._name == "cartpole":
        ax = np.asarray(m.fields["slide_axis"][0][0], np.float32)
        q[0], q[1] = float(o[0] @ ax), float(orc.joint_angles(ms, st)[1][0])
        qd[0] = float(vel[0] @ ax)
    return _Node(q=_w(q), qd=_w(qd), x=_Node(pos=full(o, 3), rot=full(st[:, 3:7], 4)),
                 xd=_Node(vel=full(vel, 3), ang=full(st[:, 10:13], 3)), _st=np.array(st, np.float32))


class OrcPipelineEnv:
    """stands in for brax.envs.base.PipelineEnv under the reference's wrappers: pipeline_init = this repo's forward
    kinematics, pipeline_step = n_frames substeps of this repo's CPU checker with the action held.  Everything above it —
    reset, step, rewards, the demo, rollout_us, the planner — is the reference's own code, executed."""
    def __init__(self, sys=None, backend="generalized", n_frames=1, debug=False, **kw):
        assert backend == "positional", backend
        self.sys, self._n_frames, self._backend = sys, int(n_frames), backend
        assert self._n_frames == int(sys._model.fields["n_frames"]), (sys._name, n_frames)   # (envs/specs.py agrees with the wrapper)

    def pipeline_init(self, q, qd):
        st = BRAX["orc"].forward(self.sys._ms, np.asarray(q, np.float32), np.asarray(qd, np.float32))
        BRAX["last_init"] = np.array(st, np.float32)
        return _pstate(self.sys, st)

    def pipeline_step(self, pipeline_state, action):
        st = np.asarray(pipeline_state._st, np.float32)
        act = np.ascontiguousarray(np.asarray(action, np.float32))
        for _ in range(self._n_frames):
            st = BRAX["orc"].substep(self.sys._ms, st, act)
        return _pstate(self.sys, st)

    @property
    def dt(self):
        return self.sys.dt * self._n_frames

    @property
    def action_size(self):
        return self.sys.act_size()

    @property
    def observation_size(self):
        return int(np.asarray(self.reset(BRAX["orc"].prng_key(0)).obs).shape[-1])


def _reconstruct_array(fun, args, arr_state, aval_state):
    """jax._src.array._reconstruct_array for the demo pickle (jog_xref.pkl holds pickled jax Arrays): the numpy array inside"""
    a = fun(*args)
    a.__setstate__(arr_state)
    return _w(a)


def install(orc, impl):
    jax, jnp = make_jax(orc, impl)
    jnp.tile = _wrapfn(np.tile)
    jnp.int32 = np.int32
    src = types.ModuleType("jax._src")
    src_arr = types.ModuleType("jax._src.array")
    src_arr._reconstruct_array = _reconstruct_array
    src.array = src_arr
    jax._src = src
    mods = {"jax": jax, "jax.numpy": jnp, "jax.lax": jax.lax, "jax.random": jax.random, "jax.nn": jax.nn, "jax._src": src,
            "jax._src.array": src_arr}
    flax = types.ModuleType("flax")
    struct = types.ModuleType("flax.struct")

    def dataclass(cls):
        cls = dataclasses.dataclass(cls)
        cls.replace = lambda self, **kw: dataclasses.replace(self, **kw)
        return cls
    struct.dataclass = dataclass
    flax.struct = struct
    mods.update({"flax": flax, "flax.struct": struct})
    for name in ("brax", "brax.base", "brax.envs", "brax.envs.base", "brax.io", "brax.io.html", "brax.io.mjcf", "brax.generalized",
                 "brax.generalized.pipeline", "brax.positional", "brax.training", "etils", "etils.epath", "tyro", "mujoco"):
        mods[name] = _Any(name)
    mods["brax.io.mjcf"].load = _fake_mjcf_load
    import pathlib
    mods["etils.epath"].resource_path = lambda name: pathlib.Path("/nonexistent-wheel") / name
    mods["etils"].epath = mods["etils.epath"]
    sys.modules.update(mods)


def run(ref, name, seed, N, H, Nd, temp, demo):
    planner = importlib.import_module("mbd.planners.mbd_planner")
    REC["steps"], REC["cur"] = [], {}
    args = planner.Args(seed=seed, env_name="car2d", Nsample=N, Hsample=H, Ndiffuse=Nd, temp_sample=temp, enable_demo=demo,
                        disable_recommended_params=True, not_render=True)
    rew_final = planner.run_diffusion(args)
    out = dict(seed=seed, N=N, H=H, Nd=Nd, temp=np.float32(temp), demo=demo, impl=1, rew_final=np.float32(rew_final),
               made_by="tools/make_ref_golden.py: the reference's mbd_planner.py + car2d.py executed under a numpy stand-in "
                       "for jax (NOT outputs of JAX; PRNG from this repo's threefry restatement)")
    for key in ("i", "rng_in", "rng_out", "Ybar_i", "Ybar_im1", "rew_mean", "rewss", "logp0", "weights"):
        out[key] = np.stack([np.asarray(s[key]) for s in REC["steps"]])
    # (the normals and the visited states of the first two steps only: they are what makes the files big, and the later
    # steps' normals follow from rng_in through the same generator)
    for key in ("eps", "qs"):
        out[key] = np.stack([np.asarray(s[key]) for s in REC["steps"][:2]])
    env = importlib.import_module("mbd.envs.car2d").Car2d()
    out["rew_xref"] = np.float32(env.rew_xref)
    out["state_init"] = np.asarray(env.x0, np.float32)
    path = os.path.join(ROOT, "tests", "golden", f"ref_car2d_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(REC['steps'])} steps, rew_final = {float(rew_final):.6f}")


def env_rollouts():
    """The reference's env alone — `mbd.utils.rollout_us(Car2d.step, state, us)` — from 96 start poses spread over the arena
    (next to obstacles, next to the goal: collisions and non-zero rewards both occur) under random actions, and
    `eval_xref_logpd` of the visited states."""
    car = importlib.import_module("mbd.envs.car2d")
    utils = importlib.import_module("mbd.utils")
    env = car.Car2d()
    g = np.random.default_rng(7)
    B, H = 96, 50
    q0 = np.stack([g.uniform(-1.2, 0.9, B), g.uniform(-0.9, 0.9, B), g.uniform(0, 2 * np.pi, B)], 1).astype(np.float32)
    q0[:16, :2] = (np.array([0.5, 0.0]) + g.normal(size=(16, 2)) * 0.12).astype(np.float32)   # around the goal
    us = np.clip(g.normal(size=(B, H, 2)) * 0.7, -1.3, 1.3).astype(np.float32)                 # (beyond +-1: the env clips)
    rewss, qs, lp = [], [], []
    for b in range(B):
        st = car.State(_w(q0[b]), _w(q0[b]), 0.0, 0.0)
        r, q = utils.rollout_us(env.step, st, _w(us[b]))
        rewss.append(np.asarray(r, np.float32)); qs.append(np.asarray(q, np.float32))
        lp.append(np.float32(env.eval_xref_logpd(q)))
    path = os.path.join(ROOT, "tests", "golden", "ref_car2d_env.npz")
    np.savez_compressed(path, q0=q0, us=us, rewss=np.stack(rewss), qs=np.stack(qs), logpd=np.array(lp, np.float32),
                        made_by="tools/make_ref_golden.py: mbd/utils.py rollout_us over mbd/envs/car2d.py, numpy stand-in for jax")
    moved = np.abs(np.diff(np.stack(qs)[:, :, :2], axis=1)).sum(-1) > 0
    print(f"wrote {path}: {B} rollouts, {int((np.stack(rewss) > 0).sum())} rewarded steps, "
          f"{int((~moved).sum())} blocked steps")


def pi_updates(orc):
    """The reference's path-integral update rules — mbd/planners/path_integral.py:33-52 softmax_update / cma_es_update /
    cem_update, module-level functions, executed — on softmax weights of random rewards (the weights themselves come from
    this repo's checker: update_once, which forms them, is a closure the reference's car2d cannot reach — it asks the env for
    `sys`)."""
    pi = importlib.import_module("mbd.planners.path_integral")
    g = np.random.default_rng(11)
    out = {}
    for case, (N, H, Nu) in enumerate(((64, 12, 3), (256, 10, 6), (400, 5, 17))):
        rews = g.normal(size=N).astype(np.float32) * 0.3 + 1.0
        Y0s = np.clip(g.normal(size=(N, H, Nu)) * 0.5, -1, 1).astype(np.float32)
        mu = (g.normal(size=(H, Nu)) * 0.2).astype(np.float32)
        sigma, temp = np.float32(0.7), 0.1
        _, _, w, _ = orc.pi_update(1, rews, Y0s, mu, float(sigma), temp)
        out.update({f"rews{case}": rews, f"Y0s{case}": Y0s, f"mu{case}": mu, f"sigma{case}": sigma, f"temp{case}": np.float32(temp),
                    f"weights{case}": np.asarray(w, np.float32)})
        for mname, fn in (("mppi", pi.softmax_update), ("cmaes", pi.cma_es_update), ("cem", pi.cem_update)):
            m2, s2 = fn(_w(w), _w(Y0s), sigma, _w(mu))
            out[f"{mname}_mu{case}"], out[f"{mname}_sigma{case}"] = np.asarray(m2, np.float32), np.float32(s2)
    path = os.path.join(ROOT, "tests", "golden", "ref_pi_updates.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def env_rewards():
    """The Brax-backed env wrappers' OWN reward code, executed on synthetic pipeline states (their physics is Brax's and
    cannot run; their `_get_reward` / `eval_xref_logpd` are plain array expressions and can): humanoidrun.py:46-51,
    hopper.py:57-65, walker2d.py:57-62, humanoidstandup.py:50-56, humanoidtrack.py:87-96 (from the INCOMING state) and
    :98-106 with the demo built by :33-43 from jog_xref.pkl."""
    g = np.random.default_rng(5)
    B = 64
    pos = (g.normal(size=(B, 3)) * np.array([2.0, 0.5, 0.6]) + np.array([0.5, 0.0, 1.2])).astype(np.float32)
    pos[:4, 2] = [1.3, 1.0, 1.1, -0.9]        # (at the clip's kink and far outside it)
    vel = (g.normal(size=(B, 3)) * 1.5 + np.array([1.6, 0, 0])).astype(np.float32)
    out = dict(root_pos=pos, root_vel=vel)

    def ps(k):
        x = types.SimpleNamespace(pos=_w(np.concatenate([pos[k][None], np.zeros((15, 3), np.float32)])))
        xd = types.SimpleNamespace(vel=_w(np.concatenate([vel[k][None], np.zeros((15, 3), np.float32)])))
        return types.SimpleNamespace(x=x, xd=xd)
    for mod, cls in (("humanoidrun", "HumanoidRun"), ("hopper", "Hopper"), ("walker2d", "Walker2d"),
                     ("humanoidstandup", "HumanoidStandup"), ("humanoidtrack", "HumanoidTrack")):
        env = getattr(importlib.import_module(f"mbd.envs.{mod}"), cls)()
        if mod == "humanoidtrack":
            out["humanoidtrack_reward"] = np.array([env._get_reward(types.SimpleNamespace(pipeline_state=ps(k))) for k in range(B)], np.float32)
            out["humanoidtrack_xref"] = np.asarray(env.xref, np.float32)
            out["humanoidtrack_track_idx"] = np.asarray(env.track_body_idx, np.int32)
            out["humanoidtrack_rew_xref"] = np.float32(env.rew_xref)
            L = int(max(env.ref_body_idx)) + 1
            xpos = (np.asarray(env.xref).transpose(1, 0, 2)[None] + g.normal(size=(24, 50, 5, 3)) *
                    g.uniform(0.02, 0.6, size=(24, 1, 1, 1))).astype(np.float32)     # [24][H][5][3] around the demo
            full = np.zeros((24, 50, L, 3), np.float32)
            full[:, :, np.asarray(env.track_body_idx)] = xpos
            out["humanoidtrack_xpos"] = xpos
            out["humanoidtrack_logpd"] = np.array([env.eval_xref_logpd(types.SimpleNamespace(x=types.SimpleNamespace(pos=_w(full[b]))))
                                                   for b in range(24)], np.float32)
        else:
            out[f"{mod}_reward"] = np.array([env._get_reward(ps(k)) for k in range(B)], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "ref_env_rewards.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def env_resets():
    """The wrappers' own `reset(rng)` executed (humanoidrun.py:19-32, hopper.py:20-34, walker2d.py:19-33, humanoidstandup.py:19-32,
    cartpole.py:20-37, humanoidtrack.py:48-61): which sub-key perturbs which coordinates, the noise ranges, cartpole's [0, pi]
    offset — with `sys.init_q` / sizes handed in from this repo's compiled models and Brax's `pipeline_init` (forward
    kinematics: not the reference's code) replaced by a recorder of the (q, qd) it is given."""
    from mbd_hip.model import Model
    out = {}
    for mod, cls in (("humanoidrun", "HumanoidRun"), ("hopper", "Hopper"), ("walker2d", "Walker2d"),
                     ("humanoidstandup", "HumanoidStandup"), ("cartpole", "Cartpole"), ("humanoidtrack", "HumanoidTrack")):
        with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{mod}.json")) as f:
            m = Model.from_json(f.read())
        init_q = m.init_q.copy()
        if mod == "cartpole":       # (the compiled model carries the reset offset of cartpole.py:26 in init_q already)
            init_q = init_q - np.array([0.0, np.pi], np.float32)
        env = getattr(importlib.import_module(f"mbd.envs.{mod}"), cls)()
        env.sys = types.SimpleNamespace(init_q=_w(init_q), q_size=lambda m=m: m.q_size(), qd_size=lambda m=m: m.qd_size(),
                                        act_size=lambda m=m: m.act_size())
        got = {}

        def pipeline_init(q, qd, got=got, L=m.n_links):
            got["q"], got["qd"] = np.asarray(q, np.float32), np.asarray(qd, np.float32)
            return types.SimpleNamespace(q=_w(got["q"]), qd=_w(got["qd"]), x=types.SimpleNamespace(pos=_w(np.zeros((L, 3), np.float32))))
        env.pipeline_init = pipeline_init
        for seed in (0, 3):
            key = orc_key(seed)
            env.reset(key)
            out[f"{mod}_key{seed}"], out[f"{mod}_q{seed}"], out[f"{mod}_qd{seed}"] = key, got["q"], got["qd"]
    path = os.path.join(ROOT, "tests", "golden", "ref_env_resets.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def env_obs(orc):
    """The wrappers' own `_get_obs` executed (humanoidrun.py:43-44, hopper.py:49-55, walker2d.py:50-56, humanoidstandup.py,
    cartpole.py:54-56, humanoidtrack.py:84-85) on a pipeline state with given q, qd and — for the two that read it — the
    torso origin's height `x.pos[0, 2]` (from this repo's forward kinematics: an INPUT here, not the reference's code).
    Two states per env: joint angles off their rest values with qd = 0, and the rest pose with |qd| up to 15 (beyond
    hopper's / walker2d's +-10 clip)."""
    from mbd_hip.model import Model
    out = {}
    g = np.random.default_rng(11)
    for mod, cls in (("humanoidrun", "HumanoidRun"), ("hopper", "Hopper"), ("walker2d", "Walker2d"),
                     ("humanoidstandup", "HumanoidStandup"), ("cartpole", "Cartpole"), ("humanoidtrack", "HumanoidTrack")):
        with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{mod}.json")) as f:
            m = Model.from_json(f.read())
        ms = m.to_struct()
        env = getattr(importlib.import_module(f"mbd.envs.{mod}"), cls)()
        env.sys = types.SimpleNamespace(act_size=lambda m=m: m.act_size())
        q_a = m.init_q.copy()
        for l in range(m.n_links):
            if m.fields["n_rot"][l] < 0:
                continue
            qi, ns = int(m.fields["q_idx"][l]), int(m.fields["n_slide"][l])
            for k in range(int(m.fields["n_rot"][l])):
                sg = float(m.fields["rot_sign"][l][k])
                lo, hi = sorted((sg * max(m.fields["rot_lo"][l][k], -0.6), sg * min(m.fields["rot_hi"][l][k], 0.6)))
                q_a[qi + ns + k] = g.uniform(lo, hi) * 0.5
            for k in range(ns):
                q_a[qi + k] += g.uniform(-0.2, 0.2)
        cases = {"a": (q_a.astype(np.float32), np.zeros(m.qd_size(), np.float32)),
                 "b": (m.init_q.astype(np.float32), g.uniform(-15, 15, m.qd_size()).astype(np.float32))}
        for tag, (q, qd) in cases.items():
            xpos = orc.link_positions(ms, orc.forward(ms, q, qd))
            ps = types.SimpleNamespace(q=_w(q), qd=_w(qd), x=types.SimpleNamespace(pos=_w(xpos)))
            try:
                obs = env._get_obs(ps)
            except TypeError:  # (humanoidrun / humanoidstandup: _get_obs(pipeline_state, action))
                obs = env._get_obs(ps, _w(np.zeros(m.act_size(), np.float32)))
            out[f"{mod}_{tag}_q"], out[f"{mod}_{tag}_qd"], out[f"{mod}_{tag}_obs"] = q, qd, np.asarray(obs, np.float32)
    path = os.path.join(ROOT, "tests", "golden", "ref_env_obs.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def run_brax(orc, env_name, seed, N, H, Nd, temp, demo):
    """The reference's WHOLE planning run for a Brax-backed env — mbd_planner.run_diffusion, mbd.utils.rollout_us, the env's
    wrapper (reset, step, reward, the demo's log-density), all executed unchanged — with brax.envs.base.PipelineEnv served by
    this repo's CPU checker (OrcPipelineEnv: forward kinematics + n_frames substeps) and jax by the numpy stand-in.  What
    the files pin: how the reference's code COMPOSES a step around the physics (which state a reward reads, humanoidtrack's
    lagged reward and step counter, n_frames per wrapper, the key chain, the score) — for every wrapper, in situ.  What they
    do not: Brax's physics (the checker stands in for it)."""
    BRAX["orc"] = orc
    base = sys.modules["brax.envs.base"]
    base.PipelineEnv, base.State = OrcPipelineEnv, BraxEnvState
    for k in [k for k in sys.modules if k == "mbd" or k.startswith("mbd.")]:   # (their classes were built on the empty stand-in)
        del sys.modules[k]
    planner = importlib.import_module("mbd.planners.mbd_planner")
    REC["steps"], REC["cur"] = [], {}
    args = planner.Args(seed=seed, env_name=env_name, Nsample=N, Hsample=H, Ndiffuse=Nd, temp_sample=temp, enable_demo=demo,
                        disable_recommended_params=True, not_render=True)
    rew_final = planner.run_diffusion(args)
    out = dict(env=env_name, seed=seed, N=N, H=H, Nd=Nd, temp=np.float32(temp), demo=demo, impl=1, rew_final=np.float32(rew_final),
               state_init=BRAX["last_init"],
               made_by="tools/make_ref_golden.py: the reference's mbd_planner.py + utils.py + mbd/envs/%s.py executed under a "
                       "numpy stand-in for jax, with brax's PipelineEnv served by this repo's CPU checker (NOT outputs of JAX "
                       "or Brax)" % env_name)
    for key in ("i", "rng_in", "rng_out", "Ybar_i", "Ybar_im1", "rew_mean", "rewss", "logp0", "weights"):
        out[key] = np.stack([np.asarray(st[key]) for st in REC["steps"]])
    out["eps"] = np.stack([np.asarray(st["eps"])[:32] for st in REC["steps"][:1]])   # (the first 32 candidates' normals: file size)
    if demo:   # the tracked links' positions of the first step (what eval_xref_logpd read), in the wrapper's own order
        env = importlib.import_module("mbd.envs").get_env(env_name)
        out["xpos_tracked"] = np.asarray(REC["steps"][0]["qs"])[:64, :, np.asarray(env.track_body_idx)]   # (64 candidates: file size)
    path = os.path.join(ROOT, "tests", "golden", f"ref_run_{env_name}{'_demo' if demo else ''}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(REC['steps'])} steps, rew_final = {float(rew_final):.6f}")


def run_brax_pi(orc, env_name, method, seed, N, H, Nr, temp):
    """The same for the baselines: the reference's run_path_integral (path_integral.py:55-148) — its sampling, eval_us, the
    standardisation without a zero-spread guard, softmax and update rule — executed whole over the checker-backed PipelineEnv."""
    BRAX["orc"] = orc
    base = sys.modules["brax.envs.base"]
    base.PipelineEnv, base.State = OrcPipelineEnv, BraxEnvState
    for k in [k for k in sys.modules if k == "mbd" or k.startswith("mbd.")]:
        del sys.modules[k]
    pi = importlib.import_module("mbd.planners.path_integral")
    REC["steps"], REC["cur"] = [], {}
    args = pi.Args(seed=seed, env_name=env_name, Nsample=N, Hsample=H, Nrefine=Nr, temp_sample=temp, update_method=method,
                   disable_recommended_params=True)
    rew_final = pi.run_path_integral(args)
    out = dict(env=env_name, method=method, seed=seed, N=N, H=H, Nr=Nr, temp=np.float32(temp), impl=1,
               rew_final=np.float32(rew_final), state_init=BRAX["last_init"],
               made_by="tools/make_ref_golden.py: the reference's path_integral.py + utils.py + mbd/envs/%s.py executed under a "
                       "numpy stand-in for jax, brax's PipelineEnv served by this repo's CPU checker (NOT outputs of JAX or Brax)" % env_name)
    for key in ("t", "rng_in", "rng_out", "mu_in", "mu_out", "sigma_in", "sigma_out", "rew_mean", "rewss", "weights"):
        out[key] = np.stack([np.asarray(st[key]) for st in REC["steps"]])
    path = os.path.join(ROOT, "tests", "golden", f"ref_pi_run_{env_name}_{method}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(REC['steps'])} steps, rew_final = {float(rew_final):.6f}, sigma {float(out['sigma_out'][-1]):.4g}")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    os.environ.setdefault("TQDM_DISABLE", "1")
    from oracle import oracle as orc_mod
    orc_mod.build()
    orc = orc_mod.Oracle("f32")
    install(orc, 1)
    sys.path.insert(0, ref)
    # BASELINE config 1 (car2d, N=128, H=30, 50 diffusion steps), and the demo-conditioned score at the demo's H = 50
    run(ref, "config1", 0, 128, 30, 50, 0.1, False)
    run(ref, "demo", 1, 64, 50, 12, 0.1, True)
    run(ref, "demo256", 2, 256, 50, 30, 0.1, True)
    env_rollouts()
    pi_updates(orc)
    env_rewards()
    global orc_key
    orc_key = lambda seed: orc.split(orc.prng_key(seed), 2, 1)[1]   # rng_reset of mbd_planner.py:79
    env_resets()
    env_obs(orc)
    # config 1 with --enable_demo: impossible at H = 30 (the demo has 50 rows, car2d.py:96-102) — what the reference does
    # with it is part of the record (it raises; the product refuses the plan: mbd_plan_create, MBD_ERR_INVALID)
    try:
        run(ref, "demo_h30", 0, 128, 30, 50, 0.1, True)
        rec = dict(raised=False, error="")
    except Exception as e:  # noqa: BLE001
        rec = dict(raised=True, error=f"{type(e).__name__}: {e}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_car2d_demo_h30.npz"), N=128, H=30, Nd=50, **rec)
    print("config 1 with enable_demo at H=30:", rec)
    # round 5: N >= 256 and >= 10 recorded steps per wrapper (round 4: N = 16-48, 3-5 steps), and the
    # demo-conditioned score at config 5's N = 2048 (H = 50 is the demo's length; 2 steps: 2 x 102 400 env steps in Python)
    for env_name, N, H, Nd, demo in (("humanoidrun", 256, 20, 11, False), ("hopper", 256, 12, 11, False), ("walker2d", 256, 10, 11, False),
                                     ("humanoidstandup", 256, 12, 11, False), ("cartpole", 256, 20, 11, False),
                                     ("humanoidtrack", 256, 20, 11, False), ("humanoidtrack", 2048, 50, 3, True)):
        run_brax(orc, env_name, 1, N, H, Nd, 0.1, demo)
    for env_name, method, N, H, Nr in (("hopper", "mppi", 256, 15, 11), ("hopper", "cma-es", 256, 15, 11), ("hopper", "cem", 256, 15, 11),
                                       ("humanoidrun", "mppi", 256, 10, 11), ("humanoidrun", "cma-es", 256, 10, 11)):
        run_brax_pi(orc, env_name, method, 2, N, H, Nr, 0.1)


if __name__ == "__main__":
    main()
