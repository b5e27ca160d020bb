"""ctypes mirror of ``mbd_model_t`` (include/mbd_hip.h) + JSON (de)serialisation of compiled models.

The compiled model is what ``brax.io.mjcf.load`` hands the reference's envs as ``sys``
(mbd/envs/humanoidrun.py:15, hopper.py:14, humanoidtrack.py:16), flattened to one POD struct that the
HIP rollout kernel reads from constant memory.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Any, Dict, List

import numpy as np

MAX_LINKS = 16
MAX_Q = 40
MAX_ACT = 24
MAX_COL = 16
MAX_TRACK = 8
LINK_STATE = 13

REWARD_KINDS = {"humanoidrun": 0, "hopper": 1, "halfcheetah": 2, "humanoidtrack": 3, "walker2d": 1,
                "humanoidstandup": 4, "cartpole": 5, "ant": 6}

# mbd_model_flags (include/mbd_hip.h).  The SPEC_* bits are the specification switches of DESIGN.md §9: code-level guesses
# about Brax's positional pipeline (default word: DEFAULT_SPEC below); tools/compare_golden.py --search tries every combination against a golden.
FLAG_RESET_QUAT_RAW, FLAG_PLANAR = 1, 2
SPEC_FLAGS = {"contact_avg": 4, "contact6_gauss_seidel": 8, "friction_vel_bound": 16, "restitution_min": 32,
              "euler_extrinsic": 64, "gyroscopic": 128}
SPEC_MASK = sum(SPEC_FLAGS.values())
# the default word (include/mbd_hip.h MBD_DEFAULT_SPEC): what mjcf.load gives a model and the shipped library's tuned kernels compile in
DEFAULT_SPEC = SPEC_FLAGS["contact_avg"]


def spec_bits(*names: str) -> int:
    """The flag word of the named specification switches: spec_bits("contact_avg", "contact6_gauss_seidel") -> 12."""
    return sum(SPEC_FLAGS[n] for n in names)


def spec_names(flags: int):
    return [n for n, b in SPEC_FLAGS.items() if flags & b]


_L, _A, _K, _T = MAX_LINKS, MAX_ACT, MAX_COL, MAX_TRACK
_f, _i = C.c_float, C.c_int32


class MbdModel(C.Structure):
    """Field order and sizes must match include/mbd_hip.h exactly (checked by tests/test_capi.py)."""

    _fields_ = [
        ("n_links", _i), ("n_q", _i), ("n_qd", _i), ("n_act", _i), ("n_col", _i), ("n_track", _i),
        ("n_frames", _i), ("reward_kind", _i), ("iso_inertia", _i), ("flags", _i), ("reserved_i", _i * 2),
        ("dt", _f), ("vel_fac", _f), ("ang_fac", _f), ("joint_scale_pos", _f), ("joint_scale_ang", _f),
        ("collide_scale", _f), ("friction", _f), ("elasticity", _f), ("gravity", _f * 3),
        ("reset_noise", _f), ("reward_params", _f * 8),
        ("parent", _i * _L), ("n_rot", _i * _L), ("n_slide", _i * _L), ("q_idx", _i * _L),
        ("qd_idx", _i * _L),
        ("inv_mass", _f * _L), ("inv_inertia", (_f * 6) * _L), ("com", (_f * 3) * _L),
        ("ap_pos", (_f * 3) * _L), ("ap_rot", (_f * 4) * _L), ("ac_pos", (_f * 3) * _L),
        ("ac_rot", (_f * 4) * _L),
        ("ang_damp", _f * _L), ("vel_damp", _f * _L),
        ("rot_lo", (_f * 3) * _L), ("rot_hi", (_f * 3) * _L), ("rot_stiff", (_f * 3) * _L),
        ("rot_damp", (_f * 3) * _L), ("rot_sign", (_f * 3) * _L),
        ("slide_axis", ((_f * 3) * 3) * _L),
        ("slide_lo", (_f * 3) * _L), ("slide_hi", (_f * 3) * _L), ("slide_damp", (_f * 3) * _L),
        ("act_link", _i * _A), ("act_slot", _i * _A), ("act_gear", _f * _A), ("act_lo", _f * _A),
        ("act_hi", _f * _A),
        ("col_link", _i * _K), ("col_pos", (_f * 3) * _K), ("col_radius", _f * _K),
        ("link_pos", (_f * 3) * _L), ("link_rot", (_f * 4) * _L), ("joint_pos", (_f * 3) * _L),
        ("rot_axis", ((_f * 3) * 3) * _L), ("slide_axis_body", ((_f * 3) * 3) * _L),
        ("init_q", _f * MAX_Q),
        ("track_link", _i * _T),
    ]


_SCALARS = ["n_links", "n_q", "n_qd", "n_act", "n_col", "n_track", "n_frames", "reward_kind",
            "iso_inertia", "flags", "dt", "vel_fac", "ang_fac", "joint_scale_pos", "joint_scale_ang",
            "collide_scale", "friction", "elasticity", "reset_noise"]
_ARRAYS = [n for n, _t in MbdModel._fields_ if n not in _SCALARS and n != "reserved_i"]


class Model:
    """A compiled model: plain numpy arrays keyed like the struct fields, plus names for the shim."""

    def __init__(self, fields: Dict[str, Any], link_names: List[str], actuator_names: List[str],
                 env_name: str = ""):
        self.fields = fields
        self.link_names = list(link_names)
        self.actuator_names = list(actuator_names)
        self.env_name = env_name

    # -- sizes the reference reads off `sys` --------------------------------------------------------
    def q_size(self) -> int:
        return int(self.fields["n_q"])

    def qd_size(self) -> int:
        return int(self.fields["n_qd"])

    def act_size(self) -> int:
        return int(self.fields["n_act"])

    @property
    def n_links(self) -> int:
        return int(self.fields["n_links"])

    @property
    def init_q(self) -> np.ndarray:
        return np.asarray(self.fields["init_q"], np.float32)[: self.q_size()].copy()

    def with_spec(self, flags: int) -> "Model":
        """A copy of the model whose specification switches (SPEC_FLAGS bits) are ``flags``; everything else unchanged."""
        f = dict(self.fields)
        f["flags"] = (int(f.get("flags", 0)) & ~SPEC_MASK) | (int(flags) & SPEC_MASK)
        m = Model(f, self.link_names, self.actuator_names, self.env_name)
        for extra in ("masses", "inertias"):
            if hasattr(self, extra):
                setattr(m, extra, getattr(self, extra))
        return m

    def to_struct(self) -> MbdModel:
        s = MbdModel()
        for name in _SCALARS:
            setattr(s, name, self.fields.get(name, 0) if name == "flags" else self.fields[name])
        for name in _ARRAYS:
            ctype_arr = getattr(s, name)
            dst = np.ctypeslib.as_array(ctype_arr)
            src = np.asarray(self.fields[name])
            if src.size == 0:
                continue
            view = dst[tuple(slice(0, n) for n in src.shape)] if src.ndim else dst
            view[...] = src
        return s

    @staticmethod
    def from_struct(st: MbdModel, link_names: List[str] = (), actuator_names: List[str] = (),
                    env_name: str = "") -> "Model":
        """The inverse of ``to_struct``: arrays trimmed to the model's own sizes (what ``from_json`` yields)."""
        L, A, K, T, Q = st.n_links, st.n_act, st.n_col, st.n_track, st.n_q
        lead = {"act_link": A, "act_slot": A, "act_gear": A, "act_lo": A, "act_hi": A, "col_link": K, "col_pos": K,
                "col_radius": K, "track_link": T, "init_q": Q, "gravity": 3, "reward_params": 8}
        fields: Dict[str, Any] = {}
        for name in _SCALARS:
            fields[name] = getattr(st, name)
        for name in _ARRAYS:
            a = np.ctypeslib.as_array(getattr(st, name)).copy()
            fields[name] = a[: lead.get(name, L)]
        return Model(fields, list(link_names) or [f"link{l}" for l in range(L)],
                     list(actuator_names) or [f"act{a}" for a in range(A)], env_name)

    def to_json(self) -> str:
        out = {"env_name": self.env_name, "link_names": self.link_names,
               "actuator_names": self.actuator_names, "fields": {}}
        for k, v in self.fields.items():
            a = np.asarray(v)
            if a.ndim == 0:
                out["fields"][k] = a.item()
            else:
                # float32 values round-trip exactly through repr of the python float
                out["fields"][k] = a.astype(np.float64 if a.dtype.kind == "f" else np.int64).tolist()
        return json.dumps(out, indent=1)

    @staticmethod
    def from_json(text: str) -> "Model":
        d = json.loads(text)
        fields: Dict[str, Any] = {}
        ftypes = dict((n, t) for n, t in MbdModel._fields_)
        fields["flags"] = 0  # (models compiled before the field existed)
        for k, v in d["fields"].items():
            if isinstance(v, list):
                base = ftypes[k]
                while hasattr(base, "_type_") and not isinstance(base._type_, str):
                    base = base._type_
                kind = np.int32 if base._type_ == "i" else np.float32
                fields[k] = np.asarray(v, dtype=kind)
            else:
                fields[k] = v
        return Model(fields, d["link_names"], d["actuator_names"], d.get("env_name", ""))
