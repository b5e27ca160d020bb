"""Env objects exposing the reference's plugin surface over the C ABI.

``reset(rng) -> State``, ``step(state, action) -> State``, ``action_size``, ``observation_size``,
``eval_xref_logpd``, ``rew_xref``, ``xref``, ``sys``, ``dt`` — what mbd_planner.py:70-80,109,118,121
touches — plus the batched ``rollout`` fast path (the vmap at mbd_planner.py:109).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import Any, Dict, Optional

import numpy as np

from .. import _capi
from ..model import LINK_STATE, MbdModel, Model
from . import specs

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "assets")


def prng_impl() -> int:
    """jax_threefry_partitionable: JAX >= 0.5.0 defaults to True; MBD_THREEFRY_PARTITIONABLE=0 selects
    the legacy layout (the reference does not pin its JAX version, setup.py:20)."""
    return int(os.environ.get("MBD_THREEFRY_PARTITIONABLE", "1") != "0")


@dataclasses.dataclass
class State:
    """brax.envs.base.State / car2d.State (car2d.py:35-40) look-alike; arrays are numpy."""
    pipeline_state: np.ndarray
    obs: Optional[np.ndarray]
    reward: np.float32
    done: np.float32
    metrics: Dict[str, Any] = dataclasses.field(default_factory=dict)

    def replace(self, **kw) -> "State":
        return dataclasses.replace(self, **kw)


def _names(env_name: str):
    """Link / actuator names of a built-in model (diagnostics only; the numbers come from the library)."""
    import json
    try:
        with open(os.path.join(_ASSETS, "compiled", f"{env_name}.json")) as f:
            d = json.load(f)
        return d["link_names"], d["actuator_names"]
    except OSError:
        return (), ()


class _EnvBase:
    _h: Optional[C.c_void_p] = None

    def _info(self):
        a, o, s, l, f = (C.c_int() for _ in range(5))
        dt = C.c_float()
        _capi.check(self._lib.mbd_env_info(self._h, C.byref(a), C.byref(o), C.byref(s), C.byref(l),
                                           C.byref(f), C.byref(dt)))
        self._action_size, self._observation_size, self._state_size = a.value, o.value, s.value
        self.dt = dt.value

    @property
    def action_size(self) -> int:
        return self._action_size

    @property
    def observation_size(self) -> int:
        return self._observation_size

    @property
    def handle(self):
        return self._h

    def reset(self, rng) -> State:
        st = np.zeros(self._state_size, np.float32)
        _capi.check(self._lib.mbd_env_reset(self._h, _capi.key_array(rng), prng_impl(), _capi.np_ptr(st)))
        return State(self._shape_state(st), self.observe(st), np.float32(0.0), np.float32(0.0), {})

    def pipeline_init(self, q, qd=None):
        """PipelineEnv.pipeline_init(q, qd) (humanoidrun.py:29): the pipeline state of generalized coordinates — what
        ``Plan.set_state0`` / ``rollout_us`` take; a state to plan from that is not a reset.  ``qd=None``: at rest (zeros of
        the model's qd size; car2d has no velocities)."""
        q = np.ascontiguousarray(q, np.float32).reshape(-1)
        if qd is None:
            sys_ = getattr(self, "sys", None)
            qd = np.zeros(sys_.qd_size() if sys_ is not None else 0, np.float32)
        qd = np.ascontiguousarray(qd, np.float32).reshape(-1)
        st = np.zeros(self._state_size, np.float32)
        _capi.check(self._lib.mbd_env_pipeline_init(self._h, _capi.np_ptr(q), q.size, _capi.np_ptr(qd) if qd.size else None,
                                                    qd.size, _capi.np_ptr(st)))
        return self._shape_state(st)

    def step(self, state: State, action) -> State:
        s_in = np.ascontiguousarray(state.pipeline_state, np.float32).reshape(-1)
        a = np.ascontiguousarray(action, np.float32).reshape(-1)
        s_out = np.zeros_like(s_in)
        rew = np.zeros(1, np.float32)
        obs = np.zeros(self.observation_size, np.float32)
        _capi.check(self._lib.mbd_env_step(self._h, _capi.np_ptr(s_in), _capi.np_ptr(a), _capi.np_ptr(s_out),
                                           _capi.np_ptr(rew), _capi.np_ptr(obs)))
        return state.replace(pipeline_state=self._shape_state(s_out), obs=obs, reward=rew[0],
                             done=self._next_done(state))

    def observe(self, pipeline_state) -> np.ndarray:
        """_get_obs of one state through the C ABI (host arithmetic inside the library)."""
        st = np.ascontiguousarray(pipeline_state, np.float32).reshape(-1)
        obs = np.zeros(self.observation_size, np.float32)
        _capi.check(self._lib.mbd_env_observe(self._h, _capi.np_ptr(st), _capi.np_ptr(obs)))
        return obs

    def eval_xref_logpd_batch(self, xpos):
        """jax.vmap(env.eval_xref_logpd)(qs) (mbd_planner.py:118) on the GPU: ``xpos`` [B,H,K,3] (car2d: [B,H,3])
        CUDA tensor as returned by ``rollout(..., want_xpos=True)`` -> [B] CUDA tensor."""
        import torch
        dev = torch.device("cuda", self.device)
        x = torch.as_tensor(xpos, dtype=torch.float32, device=dev).contiguous()
        out = torch.empty(x.shape[0], dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(self._lib.mbd_env_xref_logpd(self._h, x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), stream))
        return out

    def _next_done(self, state):
        return np.float32(0.0)

    def rollout(self, state: State, us, want_xpos: bool = False, want_final: bool = False):
        """Batched rollout on the GPU. ``us``: [B,H,Nu] numpy array or CUDA torch tensor.
        Returns rewss [B,H] (and xpos [B,H,K,3]; and the final pipeline states [B, state_size]) as CUDA torch tensors."""
        import torch
        dev = torch.device("cuda", self.device)
        us_t = torch.as_tensor(us, dtype=torch.float32, device=dev).contiguous()
        B, H, Nu = us_t.shape
        assert Nu == self.action_size
        s0 = torch.as_tensor(np.ascontiguousarray(state.pipeline_state, np.float32).reshape(-1), device=dev)
        rewss = torch.empty((B, H), dtype=torch.float32, device=dev)
        xpos = None
        if want_xpos:
            xpos = torch.empty((B, H) + self._xpos_shape(), dtype=torch.float32, device=dev)
        final = torch.empty((B, s0.numel()), dtype=torch.float32, device=dev) if want_final else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(self._lib.mbd_env_rollout(self._h, s0.data_ptr(), us_t.data_ptr(), B, H, rewss.data_ptr(),
                                              xpos.data_ptr() if want_xpos else None,
                                              final.data_ptr() if want_final else None, stream))
        out = (rewss,) + ((xpos,) if want_xpos else ()) + ((final,) if want_final else ())
        return out if len(out) > 1 else rewss

    def __del__(self):
        try:
            if self._h is not None:
                self._lib.mbd_env_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Car2d(_EnvBase):
    """mbd/envs/car2d.py:43-110 behind the C ABI."""

    def __init__(self, device: int = 0):
        self._lib = _capi.load()
        self.device = device
        self.H = 50
        h = C.c_void_p()
        _capi.check(self._lib.mbd_env_create(b"car2d", device, C.byref(h)))
        self._h = h
        self._info()
        self.xref = np.zeros((50, 2), np.float32)  # car2d.py:66
        _capi.check(self._lib.mbd_env_xref(self._h, _capi.np_ptr(self.xref), 100, None))
        rx = C.c_float()
        _capi.check(self._lib.mbd_env_rew_xref(self._h, C.byref(rx)))
        self.rew_xref = rx.value
        self.x0 = np.array([-0.5, 0.0, np.pi * 3 / 2], np.float32)
        self.xg = np.array([0.5, 0.0, 0.0], np.float32)

    def _shape_state(self, st):
        return st.reshape(3)

    def _xpos_shape(self):
        return (3,)

    def reset(self, rng=None) -> State:
        return super().reset(np.zeros(2, np.uint32) if rng is None else rng)

    def eval_xref_logpd(self, xs) -> np.float32:
        """car2d.py:95-102 for ONE trajectory xs [H,3] (host; the planner uses the batched kernel)."""
        xs = np.asarray(xs, np.float32)
        err = np.linalg.norm(xs[:, :2] - self.xref[:, :2], axis=-1)
        return np.float32(0.0 - ((np.clip(err, 0.0, 0.5) / 0.5) ** 2).mean())


class RigidBodyEnv(_EnvBase):
    """HumanoidRun / HumanoidTrack / Hopper / Halfcheetah (mbd/envs/*.py) behind the C ABI."""

    def __init__(self, env_name: str, device: int = 0, model: Optional[Model] = None):
        self._lib = _capi.load()
        self.device = device
        self.env_name = env_name
        spec = specs.SPECS[env_name]
        h = C.c_void_p()
        if model is None:
            # by name, like mbd.envs.get_env: the compiled model and the demo live inside the library
            _capi.check(self._lib.mbd_env_create(env_name.encode(), device, C.byref(h)))
        else:  # a caller-compiled model (mbd_hip.mjcf.load): custom MJCF files, experiments
            xref_ptr, rew_xref = None, 0.0
            if env_name == "humanoidtrack":
                xr = np.ascontiguousarray(np.load(os.path.join(_ASSETS, "compiled", "jog_xref.npy")), np.float32)
                xref_ptr, rew_xref = _capi.np_ptr(xr), 1.0  # humanoidtrack.py:44
            st = model.to_struct()
            _capi.check(self._lib.mbd_env_create_model(env_name.encode(), device, C.byref(st), xref_ptr, rew_xref,
                                                       C.byref(h)))
        self._h = h
        self._info()
        st = MbdModel()
        _capi.check(self._lib.mbd_env_get_model(self._h, C.byref(st)))
        names = (model.link_names, model.actuator_names) if model is not None else _names(env_name)
        self.sys = Model.from_struct(st, *names, env_name=env_name)  # env.sys (mbd_planner.py:174)
        self._struct = st
        self.xref = None
        n = C.c_int()
        _capi.check(self._lib.mbd_env_xref(self._h, None, 0, C.byref(n)))
        if n.value:  # env.xref (humanoidtrack.py:36-43): [n_track][50][3]
            self.xref = np.zeros((int(st.n_track), 50, 3), np.float32)
            _capi.check(self._lib.mbd_env_xref(self._h, _capi.np_ptr(self.xref), n.value, None))
        rx = C.c_float()
        _capi.check(self._lib.mbd_env_rew_xref(self._h, C.byref(rx)))
        self.rew_xref = rx.value
        if env_name == "humanoidtrack":
            self.H = 50  # humanoidtrack.py:17
            self.track_body_names = list(spec["track"])
            self.track_body_idx = np.asarray(self.sys.fields["track_link"], np.int32)

    def _shape_state(self, st):
        return st.reshape(self.sys.n_links, LINK_STATE)

    def _xpos_shape(self):
        return (int(self.sys.fields["n_track"]), 3)

    def _next_done(self, state):
        # humanoidtrack abuses `done` as a time counter (humanoidtrack.py:71,81)
        return np.float32(state.done + 1) if self.env_name == "humanoidtrack" else np.float32(0.0)

    # ---- kinematics.inverse on the host (observations only; the planner never reads obs) ---------------
    def generalized(self, pipeline_state):
        """(q, qd) from a [L,13] COM-frame state (mbd_model_observe, host arithmetic inside the library): free
        root = link-frame pose / velocity; slides = anchor offset / relative anchor velocity along the slide axes;
        hinges = joint-frame Euler angles (x, y', z'') times the MJCF axis handedness and the relative angular
        velocity projected on the gimbal axes."""
        st = np.ascontiguousarray(pipeline_state, np.float32).reshape(-1)
        q, qd = np.zeros(self.sys.q_size(), np.float32), np.zeros(self.sys.qd_size(), np.float32)
        _capi.check(self._lib.mbd_model_observe(C.byref(self._struct), _capi.np_ptr(st), _capi.np_ptr(q),
                                                _capi.np_ptr(qd), None))
        return q, qd

    def _get_obs(self, pipeline_state) -> np.ndarray:
        st = np.ascontiguousarray(pipeline_state, np.float32).reshape(-1)
        obs = np.zeros(self.observation_size, np.float32)
        _capi.check(self._lib.mbd_model_observe(C.byref(self._struct), _capi.np_ptr(st), None, None, _capi.np_ptr(obs)))
        return obs

    @property
    def observation_size(self) -> int:
        n = self.sys.q_size() + self.sys.qd_size()
        return n - {"halfcheetah": 1, "ant": 2}.get(self.env_name, 0)

    def link_positions(self, pipeline_state) -> np.ndarray:
        """x.pos of every link (world position of the link-frame origin) from a [L,13] state."""
        from ..mjcf import _rot
        s = np.asarray(pipeline_state, np.float64).reshape(-1, LINK_STATE)
        com = np.asarray(self.sys.fields["com"], np.float64)
        return np.stack([s[l, :3] - _rot(com[l], s[l, 3:7]) for l in range(self.sys.n_links)]).astype(np.float32)

    def eval_xref_logpd(self, xpos) -> np.float32:
        """humanoidtrack.py:98-106 for ONE trajectory: xpos [H,K,3] tracked link positions (host)."""
        xs = np.asarray(xpos, np.float32).transpose(1, 0, 2)
        err = np.linalg.norm(xs - self.xref, axis=-1)
        return np.float32(0.0 - ((np.clip(err, 0.0, 0.5) / 0.5) ** 2).mean())
