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
from ..model import LINK_STATE, Model
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
        return State(self._shape_state(st), None, np.float32(0.0), np.float32(0.0), {})

    def step(self, state: State, action) -> State:
        s_in = np.ascontiguousarray(state.pipeline_state, np.float32).reshape(-1)
        a = np.ascontiguousarray(action, np.float32).reshape(-1)
        s_out = np.zeros_like(s_in)
        rew = np.zeros(1, np.float32)
        _capi.check(self._lib.mbd_env_step(self._h, _capi.np_ptr(s_in), _capi.np_ptr(a), _capi.np_ptr(s_out),
                                           _capi.np_ptr(rew), None))
        return state.replace(pipeline_state=self._shape_state(s_out), reward=rew[0],
                             done=self._next_done(state))

    def _next_done(self, state):
        return np.float32(0.0)

    def rollout(self, state: State, us, want_xpos: bool = False):
        """Batched rollout on the GPU. ``us``: [B,H,Nu] numpy array or CUDA torch tensor.
        Returns rewss [B,H] (and xpos [B,H,K,3]) as CUDA torch tensors."""
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
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(self._lib.mbd_env_rollout(self._h, s0.data_ptr(), us_t.data_ptr(), B, H, rewss.data_ptr(),
                                              xpos.data_ptr() if want_xpos else None, None, stream))
        return (rewss, xpos) if want_xpos else rewss

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
        self.xref = np.load(os.path.join(_ASSETS, "compiled", "car2d_xref.npy")).astype(np.float32)
        h = C.c_void_p()
        _capi.check(self._lib.mbd_env_create_car2d(device, _capi.np_ptr(self.xref), C.byref(h)))
        self._h = h
        self._info()
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
        st = super().reset(np.zeros(2, np.uint32) if rng is None else rng)
        return st.replace(obs=st.pipeline_state.copy())

    def step(self, state: State, action) -> State:  # car2d.py:86: obs = q
        st = super().step(state, action)
        return st.replace(obs=np.asarray(st.pipeline_state).copy())

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
        if model is None:
            with open(os.path.join(_ASSETS, "compiled", f"{env_name}.json")) as f:
                model = Model.from_json(f.read())
        self.sys = model
        self._struct = model.to_struct()
        self.xref = None
        self.rew_xref = 0.0
        xref_ptr = None
        if env_name == "humanoidtrack":
            self.H = 50  # humanoidtrack.py:17
            self.xref = np.ascontiguousarray(np.load(os.path.join(_ASSETS, "compiled", "jog_xref.npy")), np.float32)
            self.rew_xref = 1.0  # humanoidtrack.py:44
            self.track_body_names = list(spec["track"])
            self.track_body_idx = np.asarray(model.fields["track_link"], np.int32)
            xref_ptr = _capi.np_ptr(self.xref)
        h = C.c_void_p()
        _capi.check(self._lib.mbd_env_create_model(env_name.encode(), device, C.byref(self._struct), xref_ptr,
                                                   self.rew_xref, C.byref(h)))
        self._h = h
        self._info()

    def _shape_state(self, st):
        return st.reshape(self.sys.n_links, LINK_STATE)

    def _xpos_shape(self):
        return (int(self.sys.fields["n_track"]), 3)

    def _next_done(self, state):
        # humanoidtrack abuses `done` as a time counter (humanoidtrack.py:71,81)
        return np.float32(state.done + 1) if self.env_name == "humanoidtrack" else np.float32(0.0)

    # ---- kinematics.inverse on the host (observations only; the planner never reads obs) ---------------
    def generalized(self, pipeline_state):
        """(q, qd) from a [L,13] COM-frame state: free root = link-frame pose / velocity; slides = anchor
        offset / relative anchor velocity along the slide axes; hinges = joint-frame Euler angles (x, y', z'')
        times the MJCF axis handedness and the relative angular velocity projected on the gimbal axes."""
        from ..mjcf import _q2mat, _qmul
        F = self.sys.fields
        s = np.asarray(pipeline_state, np.float64).reshape(-1, LINK_STATE)
        q = np.zeros(self.sys.q_size())
        qd = np.zeros(self.sys.qd_size())
        for l in range(self.sys.n_links):
            p, r, v, w = s[l, :3], s[l, 3:7], s[l, 7:10], s[l, 10:13]
            R = _q2mat(r)
            qi, di = int(F["q_idx"][l]), int(F["qd_idx"][l])
            if F["n_rot"][l] < 0:
                c = R @ np.asarray(F["com"][l], float)
                q[qi:qi + 3], q[qi + 3:qi + 7] = p - c, r
                qd[di:di + 3], qd[di + 3:di + 6] = v - np.cross(w, c), w
                continue
            par = int(F["parent"][l])
            if par >= 0:
                Pp, Pr, Pv, Pw = s[par, :3], s[par, 3:7], s[par, 7:10], s[par, 10:13]
            else:
                Pp, Pr, Pv, Pw = np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3), np.zeros(3)
            RP = _q2mat(Pr)
            ap = Pp + RP @ np.asarray(F["ap_pos"][l], float)
            ac = p + R @ np.asarray(F["ac_pos"][l], float)
            A = _q2mat(_qmul(Pr, np.asarray(F["ap_rot"][l], float)))
            Cm = _q2mat(_qmul(r, np.asarray(F["ac_rot"][l], float)))
            Xp, Yp, Zp = A[:, 0], A[:, 1], A[:, 2]
            Xc, Yc, Zc = Cm[:, 0], Cm[:, 1], Cm[:, 2]
            ang = [np.arctan2(-Zc @ Yp, Zc @ Zp), np.arcsin(np.clip(Zc @ Xp, -1, 1)), np.arctan2(-Yc @ Xp, Xc @ Xp)]
            n1 = np.cross(Zc, Xp)
            axes = [Xp, n1 / (np.linalg.norm(n1) + 1e-12), Zc]
            rel_w = w - Pw
            rel_v = (v + np.cross(w, ac - p)) - (Pv + np.cross(Pw, ap - Pp))
            ns, nr = int(F["n_slide"][l]), int(F["n_rot"][l])
            for k in range(ns):
                sk = A @ np.asarray(F["slide_axis"][l][k], float)
                q[qi + k], qd[di + k] = (ac - ap) @ sk, rel_v @ sk
            for k in range(nr):
                sg = float(F["rot_sign"][l][k])
                q[qi + ns + k], qd[di + ns + k] = sg * ang[k], sg * (rel_w @ axes[k])
        return q.astype(np.float32), qd.astype(np.float32)

    def _get_obs(self, pipeline_state) -> np.ndarray:
        q, qd = self.generalized(pipeline_state)
        if self.env_name in ("hopper", "walker2d"):  # hopper.py:49-55 / walker2d.py:49-55
            pos = q.copy()
            pos[1] = self.link_positions(pipeline_state)[0, 2]
            return np.concatenate([pos, np.clip(qd, -10, 10)]).astype(np.float32)
        if self.env_name == "ant":  # brax ant: root x, y excluded
            return np.concatenate([q[2:], qd]).astype(np.float32)
        if self.env_name == "halfcheetah":  # brax half_cheetah: the root x position is excluded
            return np.concatenate([q[1:], qd]).astype(np.float32)
        return np.concatenate([q, qd]).astype(np.float32)  # humanoidrun.py:43-44 etc.

    def reset(self, rng) -> State:
        st = super().reset(rng)
        return st.replace(obs=self._get_obs(st.pipeline_state))

    def step(self, state: State, action) -> State:
        st = super().step(state, action)
        return st.replace(obs=self._get_obs(st.pipeline_state))

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
