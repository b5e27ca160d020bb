"""MJCF -> compiled positional-dynamics model (host side, numpy only).

Stands in for ``brax.io.mjcf.load`` as called by the reference's envs (mbd/envs/humanoidrun.py:15,
humanoidtrack.py:16, hopper.py:13-14) for the subset of MJCF those models use: one top-level
``<default>``, capsule/sphere geoms, free/hinge/slide joints, motor actuators, a floor plane, and
Brax's ``<custom><numeric>`` solver parameters (mbd/assets/humanoidrun.xml:10-23).

What MuJoCo's compiler contributes inside ``mjcf.load`` is restated here from the MuJoCo
documentation (third-party, absent from this container): ``inertiafromgeom`` masses/inertias at
density 1000, ``fromto`` capsules, degree angles, default classes.  What Brax adds: joint-less bodies
are fused into their parent, one *link* per remaining body, link/joint frames, and the
``spring_mass_scale`` / ``spring_inertia_scale`` exponents applied to mass and principal inertias
(brax/com.py) — with the humanoid's ``spring_inertia_scale = 1`` every inertia tensor becomes the
identity.

PARITY UNPINNED: none of this can be checked against a Brax/MuJoCo install here; the unit tests pin it
against closed-form masses, centres of mass and inertias instead (tests/test_mjcf.py).
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence

import numpy as np

from .model import (DEFAULT_SPEC, MAX_ACT, MAX_COL, MAX_LINKS, MAX_Q, MAX_TRACK, Model, REWARD_KINDS, SPEC_MASK)

# defaults of brax.io.mjcf.load for <custom><numeric> entries that are absent (recollection)
_CUSTOM_DEFAULTS = {
    "vel_damping": 0.0, "ang_damping": 0.0, "joint_scale_pos": 0.5, "joint_scale_ang": 0.2,
    "collide_scale": 1.0, "spring_mass_scale": 0.0, "spring_inertia_scale": 0.0, "elasticity": 0.0,
    "constraint_stiffness": 2000.0, "constraint_limit_stiffness": 1000.0,
    "constraint_vel_damping": 0.0, "constraint_ang_damping": 0.0,
}
_BIG = 1.0e9


# ---- small float64 rotation helpers ---------------------------------------------------------------
def _qmul(a, b):
    w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]
    x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]
    y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1]
    z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]
    return np.array([w, x, y, z])


def _qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _q2mat(q):
    q = np.asarray(q, float) / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def _mat2q(m):
    m = np.asarray(m, float)
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    q = q / np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def _from_to(a, b):
    """Shortest-arc quaternion rotating unit vector a onto unit vector b (brax math.from_to)."""
    a = np.asarray(a, float) / np.linalg.norm(a)
    b = np.asarray(b, float) / np.linalg.norm(b)
    d = float(np.dot(a, b))
    if d > 1 - 1e-12:
        return np.array([1.0, 0, 0, 0])
    if d < -1 + 1e-12:
        o = np.cross(a, [0.0, 0.0, 1.0])
        if np.linalg.norm(o) < 1e-6:
            o = np.cross(a, [0.0, 1.0, 0.0])
        o = o / np.linalg.norm(o)
        return np.array([0.0, *o])
    c = np.cross(a, b)
    q = np.array([1.0 + d, *c])
    return q / np.linalg.norm(q)


def _rot(v, q):
    return _q2mat(q) @ np.asarray(v, float)


def _floats(s: Optional[str], n: Optional[int] = None, default=None):
    if s is None:
        return default
    v = [float(t) for t in s.split()]
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return np.array(v)


# ---- geoms -----------------------------------------------------------------------------------------
class _Geom:
    def __init__(self, kind, pos, quat, radius, half, density, contype, conaffinity, friction, name):
        self.kind, self.pos, self.quat = kind, np.asarray(pos, float), np.asarray(quat, float)
        self.radius, self.half, self.density = radius, half, density
        self.contype, self.conaffinity, self.friction, self.name = contype, conaffinity, friction, name

    def mass_inertia(self):
        """(mass, inertia about the geom centre in the BODY frame) — MuJoCo inertiafromgeom."""
        r, h, rho = self.radius, self.half, self.density
        if self.kind == "sphere":
            m = rho * 4.0 / 3.0 * math.pi * r ** 3
            local = np.eye(3) * (0.4 * m * r * r)
        elif self.kind == "capsule":
            mc = rho * math.pi * r * r * (2 * h)
            ms = rho * 4.0 / 3.0 * math.pi * r ** 3
            m = mc + ms
            izz = mc * r * r / 2 + ms * 0.4 * r * r
            ixx = mc * ((2 * h) ** 2 / 12 + r * r / 4) + ms * (0.4 * r * r + h * h + 0.75 * h * r)
            local = np.diag([ixx, ixx, izz])
        else:
            raise ValueError(f"geom type {self.kind!r} is outside the hot-path subset")
        R = _q2mat(self.quat)
        return m, R @ local @ R.T

    def transformed(self, pos, quat):
        """The same geom expressed in the parent frame of a body at (pos, quat)."""
        return _Geom(self.kind, np.asarray(pos) + _rot(self.pos, quat), _qmul(quat, self.quat),
                     self.radius, self.half, self.density, self.contype, self.conaffinity,
                     self.friction, self.name)

    def sphere_points(self):
        """Sphere colliders this geom contributes against a plane (capsule = its two end spheres)."""
        if self.kind == "sphere":
            return [(self.pos, self.radius)]
        if self.kind == "capsule":
            ax = _rot([0, 0, self.half], self.quat)
            return [(self.pos + ax, self.radius), (self.pos - ax, self.radius)]
        raise ValueError(self.kind)


class _Body:
    def __init__(self, name, pos, quat):
        self.name, self.pos, self.quat = name, np.asarray(pos, float), np.asarray(quat, float)
        self.joints: List[dict] = []
        self.geoms: List[_Geom] = []
        self.children: List["_Body"] = []


def _merged(defaults: Dict[str, Dict[str, str]], tag: str, elem) -> Dict[str, str]:
    d = dict(defaults.get(tag, {}))
    d.update(elem.attrib)
    return d


def _parse_geom(elem, defaults, angle_scale) -> Optional[_Geom]:
    a = _merged(defaults, "geom", elem)
    kind = a.get("type", "sphere")
    if kind == "plane":
        return None
    size = _floats(a.get("size"), None, np.array([0.0]))
    quat = _floats(a.get("quat"), 4, np.array([1.0, 0, 0, 0]))
    if "axisangle" in a:
        aa = _floats(a["axisangle"], 4)
        ang = aa[3] * angle_scale
        ax = aa[:3] / np.linalg.norm(aa[:3])
        quat = np.array([math.cos(ang / 2), *(math.sin(ang / 2) * ax)])
    pos = _floats(a.get("pos"), 3, np.zeros(3))
    half = 0.0
    if kind == "capsule":
        if "fromto" in a:
            ft = _floats(a["fromto"], 6)
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            half = 0.5 * float(np.linalg.norm(p1 - p0))
            quat = _from_to([0, 0, 1.0], (p1 - p0))
        else:
            half = float(size[1])
    elif kind != "sphere":
        raise ValueError(f"geom type {kind!r} is outside the hot-path subset (sphere, capsule)")
    fr = _floats(a.get("friction"), None, np.array([1.0, 0.005, 0.0001]))
    return _Geom(kind, pos, quat, float(size[0]), half, float(a.get("density", 1000.0)),
                 int(a.get("contype", 1)), int(a.get("conaffinity", 1)), float(fr[0]),
                 a.get("name", ""))


def _parse_body(elem, defaults, angle_scale) -> _Body:
    b = _Body(elem.get("name", ""), _floats(elem.get("pos"), 3, np.zeros(3)),
              _floats(elem.get("quat"), 4, np.array([1.0, 0, 0, 0])))
    b.quat = b.quat / np.linalg.norm(b.quat)
    for ch in elem:
        if ch.tag == "joint" or ch.tag == "freejoint":
            a = _merged(defaults, "joint", ch) if ch.tag == "joint" else dict(ch.attrib, type="free")
            b.joints.append(a)
        elif ch.tag == "geom":
            g = _parse_geom(ch, defaults, angle_scale)
            if g is not None:
                b.geoms.append(g)
        elif ch.tag == "body":
            b.children.append(_parse_body(ch, defaults, angle_scale))
    return b


def _fuse(b: _Body) -> None:
    """Brax fuses joint-less bodies into their parent (feet -> shins in the humanoid)."""
    new_children: List[_Body] = []
    for c in b.children:
        _fuse(c)
        if not c.joints:
            for g in c.geoms:
                b.geoms.append(g.transformed(c.pos, c.quat))
            for gc in c.children:
                gc.pos = c.pos + _rot(gc.pos, c.quat)
                gc.quat = _qmul(c.quat, gc.quat)
                new_children.append(gc)
        else:
            new_children.append(c)
    b.children = new_children


def is_planar(F) -> bool:
    """A model the planar restatement applies to (MBD_FLAG_PLANAR, include/mbd_hip.h): every joint is a hinge about the
    world y axis (joint frames = a quarter turn about z, identical on both sides) or hinge-less, slides only on
    world-parented links and in the x-z plane, link frames un-rotated, all offsets in the plane, diagonal inertia, no
    gravity along y, at most two slide dofs, at most FOUR sphere colliders per link (what the planar kernels are built for:
    two as one packed pair, three or four — the halfcheetah under collide_all_capsules, whose torso carries four — one by one)."""
    L = int(F["n_links"])
    if L < 1 or abs(float(np.asarray(F["gravity"])[1])) != 0.0:
        return False
    for l in range(L):
        nr, ns = int(F["n_rot"][l]), int(F["n_slide"][l])
        if nr < 0 or nr > 1 or ns > 2 or (ns > 0 and int(F["parent"][l]) >= 0):
            return False
        apr, acr = np.asarray(F["ap_rot"][l], np.float64), np.asarray(F["ac_rot"][l], np.float64)
        if not np.array_equal(apr, acr) or apr[1] != 0.0 or apr[2] != 0.0:
            return False
        if nr == 1 and abs(abs(2.0 * apr[0] * apr[3]) - 1.0) > 1e-6:   # joint X axis = +-y
            return False
        if nr == 0 and not np.array_equal(apr, [1.0, 0.0, 0.0, 0.0]):
            return False
        if not np.array_equal(np.asarray(F["link_rot"][l], np.float64), [1.0, 0.0, 0.0, 0.0]):
            return False
        for key in ("ap_pos", "ac_pos", "com"):
            if float(np.asarray(F[key][l])[1]) != 0.0:
                return False
        if np.any(np.asarray(F["inv_inertia"][l])[3:] != 0.0):
            return False
        for k in range(ns):
            s_w = _rot(np.asarray(F["slide_axis"][l][k], np.float64), apr)
            if abs(s_w[1]) > 1e-9:
                return False
    ncol = int(np.asarray(F["col_link"]).shape[0]) if np.ndim(F["col_link"]) else 0
    per_link = {}
    for k in range(ncol):
        if float(np.asarray(F["col_pos"])[k][1]) != 0.0:
            return False
        per_link[int(np.asarray(F["col_link"])[k])] = per_link.get(int(np.asarray(F["col_link"])[k]), 0) + 1
    if per_link and max(per_link.values()) > 4:
        return False
    return int(np.asarray(F["track_link"]).size) == 0


def stability_report(model) -> List[str]:
    """What a custom model asks of an explicit, Jacobi-summed position-based step that it cannot give (found by fuzzing
    random models, tests/random_models.py; properties of the algorithm, not of this implementation).  Returns one line per
    finding; empty for every built-in model.
      * constraint / joint dampers act explicitly: a damper d between two links is stable while
        dt * d * (lambda_max(I_child^-1) + lambda_max(I_parent^-1)) < 2 (angular) resp. dt * d * (1/m_c + 1/m_p) < 2 —
        which is why the reference's XMLs pair constraint_ang_damping = 30 with spring_inertia_scale = 1 (unit tensors);
      * the joint stage SUMS the corrections of all joints of a link: with its share w_link / (w_link + w_other) of each,
        joint_scale_pos * (sum of shares) must stay below 4/3 — beyond it the pose-difference velocity feeds the overshoot
        back and the links fly apart even in free fall;
      * a joint with two or three hinge dofs reads its angles as Euler angles (x, y', z''): at a middle angle of +-90 degrees
        they are undefined (1 / cos) and the step returns non-finite values — the middle dof needs a range inside (-90, 90)
        (the reference's humanoids: abdomen_y -75..30, hip_z -60..35, shoulder2 -85..60; round-6 fuzz, seed 2123)."""
    F, L = model.fields, model.n_links
    dt = float(F["dt"])
    lam = [float(np.linalg.eigvalsh(np.array([[i[0], i[3], i[4]], [i[3], i[1], i[5]], [i[4], i[5], i[2]]], np.float64)).max())
           for i in np.asarray(F["inv_inertia"][:L], np.float64)]
    w = np.asarray(F["inv_mass"][:L], np.float64)
    out, load = [], np.zeros(L)
    for l in range(L):
        p = int(F["parent"][l])
        if int(F["n_rot"][l]) < 0:
            continue
        lp, wp = (lam[p], w[p]) if p >= 0 else (0.0, 0.0)
        d_ang = float(F["ang_damp"][l]) + float(np.max(np.asarray(F["rot_damp"][l])[:max(int(F["n_rot"][l]), 0)], initial=0.0))
        if dt * d_ang * (lam[l] + lp) >= 2.0:
            out.append(f"link {model.link_names[l]!r}: angular damping {d_ang:g} is explicit-unstable at dt {dt:g} "
                       f"(dt d (1/I_c + 1/I_p) = {dt * d_ang * (lam[l] + lp):.3g} >= 2); lower it or use spring_inertia_scale = 1")
        d_vel = float(F["vel_damp"][l])
        if dt * d_vel * (w[l] + wp) >= 2.0:
            out.append(f"link {model.link_names[l]!r}: constraint_vel_damping {d_vel:g} is explicit-unstable at dt {dt:g} "
                       f"(dt d (1/m_c + 1/m_p) = {dt * d_vel * (w[l] + wp):.3g} >= 2)")
        load[l] += w[l] / (w[l] + wp)
        if p >= 0:
            load[p] += wp / (w[l] + wp)
        if int(F["n_rot"][l]) >= 2:
            lo, hi = float(np.asarray(F["rot_lo"][l])[1]), float(np.asarray(F["rot_hi"][l])[1])
            if lo <= -0.5 * math.pi + 0.01 or hi >= 0.5 * math.pi - 0.01:
                out.append(f"link {model.link_names[l]!r}: the middle hinge of its {int(F['n_rot'][l])}-dof joint may reach +-90 degrees "
                           f"(range {math.degrees(max(lo, -1e3)):.0f} .. {math.degrees(min(hi, 1e3)):.0f}): its Euler angles are undefined there; limit it inside (-90, 90)")
    jsp = float(F["joint_scale_pos"])
    if L and jsp * float(load.max()) >= 4.0 / 3.0:
        k = int(load.argmax())
        out.append(f"link {model.link_names[k]!r}: joint_scale_pos {jsp:g} x {load[k]:.2f} (its share of its joints' summed "
                   f"corrections) = {jsp * load[k]:.2f} >= 4/3: the joint stage overshoots; lower joint_scale_pos to <= {1.2 / load[k]:.2f}")
    return out


def load(path: str, env_name: str = "", n_frames: int = 1, drop_link_suffix: Optional[str] = None,
         track_names: Sequence[str] = (), reset_noise: float = 0.0,
         reward_params: Sequence[float] = (), dt_override: Optional[float] = None,
         init_q_offset: Sequence[float] = (), gear_override: Sequence[float] = (),
         passive_joint_forces: bool = True, reset_quat_raw: bool = False, planar: Optional[bool] = None,
         spec_flags: Optional[int] = None, warn_unstable: bool = True, collide_all_capsules: bool = False) -> Model:
    """Compile an MJCF file. ``n_frames`` is the env's physics substeps per control step
    (humanoidrun.py:17 -> 7, humanoidtrack.py:46 -> 5, hopper.py:18 -> 20).

    Named switches for the places where this engine had to GUESS what Brax does — each is a
    recompile of the MODEL, the kernels and the checker stay as they are:
      passive_joint_forces  MJCF joint ``stiffness`` / ``damping`` act as passive joint forces on top of the
                            <custom> constraint_{ang,vel}_damping (default True); False zeroes them in the model.
      reset_quat_raw        reset() leaves the noise-perturbed root quaternion un-normalised
                            (MBD_FLAG_RESET_QUAT_RAW; default False: normalised).
      gear_override         the env class's replacement of sys.actuator.gear (brax ant / half_cheetah).
      planar                None: set MBD_FLAG_PLANAR when the model qualifies (see ``is_planar``); False: keep
                            the general 3-D arithmetic for a planar model.
      spec_flags            the CODE-level guesses as flag bits (model.SPEC_FLAGS / model.spec_bits: contact_avg,
                            contact6_gauss_seidel, friction_vel_bound, restitution_min, euler_extrinsic, gyroscopic;
                            include/mbd_hip.h mbd_model_flags): checker and kernels honour them alike.  None: the default
                            word, model.DEFAULT_SPEC (contact_avg since round 6) — what the shipped library's tuned
                            kernels compile in; any other word runs the general kernels.
      collide_all_capsules  a DATA-level guess about the re-authored hopper / walker2d / halfcheetah files (Brax ships its own
                            in its wheel): False — only the geoms whose contype / conaffinity meet the floor's collide (the
                            FEET in those files: a body that tips over sinks through the floor and keeps collecting its
                            forward reward); True — every sphere and capsule of every link collides, capsule ends as
                            spheres (the collider type the kernels already have), whatever its masks say.
    Reward-side switches live in reward_params (ant: [5] = terminate_when_unhealthy).
    warn_unstable: emit ``stability_report``'s findings as warnings (custom models; the built-in ones have none)."""
    root = ET.parse(path).getroot()
    comp = root.find("compiler")
    angle_scale = 1.0 if (comp is not None and comp.get("angle", "degree") == "radian") else math.pi / 180
    defaults: Dict[str, Dict[str, str]] = {}
    dflt = root.find("default")
    if dflt is not None:
        for ch in dflt:
            if ch.tag == "default":
                raise ValueError("nested <default> classes are outside the hot-path subset")
            defaults[ch.tag] = dict(ch.attrib)
    opt = root.find("option")
    dt = float(opt.get("timestep", 0.002)) if opt is not None else 0.002
    if dt_override is not None:  # e.g. cartpole.py:18 sys.replace(dt=0.005)
        dt = float(dt_override)
    gravity = _floats(opt.get("gravity") if opt is not None else None, 3, np.array([0, 0, -9.81]))
    custom = dict(_CUSTOM_DEFAULTS)
    cst = root.find("custom")
    if cst is not None:
        for n in cst.findall("numeric"):
            custom[n.get("name")] = float(n.get("data").split()[0])

    world = root.find("worldbody")
    floor = None
    for g in world.findall("geom"):
        a = _merged(defaults, "geom", g)
        if a.get("type") == "plane":
            fr = _floats(a.get("friction"), None, np.array([1.0, 0.005, 0.0001]))
            floor = dict(contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
                         friction=float(fr[0]))
    top = _Body("world", np.zeros(3), np.array([1.0, 0, 0, 0]))
    for bel in world.findall("body"):
        top.children.append(_parse_body(bel, defaults, angle_scale))
    _fuse(top)

    # ---- flatten depth-first: one link per body -----------------------------------------------------
    links: List[dict] = []

    def visit(b: _Body, parent: int):
        if drop_link_suffix and b.name.endswith(drop_link_suffix):
            return
        idx = len(links)
        links.append(dict(body=b, parent=parent))
        for c in b.children:
            visit(c, idx)

    for c in top.children:
        visit(c, -1)
    L = len(links)
    if L > MAX_LINKS:
        raise ValueError(f"{L} links > MBD_MAX_LINKS={MAX_LINKS}")

    F: Dict[str, np.ndarray] = {}
    z = lambda *s, dt_=np.float32: np.zeros(s, dt_)
    F.update(parent=z(L, dt_=np.int32), n_rot=z(L, dt_=np.int32), n_slide=z(L, dt_=np.int32),
             q_idx=z(L, dt_=np.int32), qd_idx=z(L, dt_=np.int32), inv_mass=z(L),
             inv_inertia=z(L, 6), com=z(L, 3), ap_pos=z(L, 3), ap_rot=z(L, 4), ac_pos=z(L, 3),
             ac_rot=z(L, 4), ang_damp=z(L), vel_damp=z(L), rot_lo=z(L, 3), rot_hi=z(L, 3),
             rot_stiff=z(L, 3), rot_damp=z(L, 3), rot_sign=z(L, 3), slide_axis=z(L, 3, 3),
             slide_lo=z(L, 3), slide_hi=z(L, 3), slide_damp=z(L, 3),
             link_pos=z(L, 3), link_rot=z(L, 4), joint_pos=z(L, 3), rot_axis=z(L, 3, 3),
             slide_axis_body=z(L, 3, 3))
    F["ap_rot"][:, 0] = 1
    F["ac_rot"][:, 0] = 1
    F["link_rot"][:, 0] = 1
    F["rot_sign"][:] = 1
    F["slide_lo"][:] = -_BIG
    F["slide_hi"][:] = _BIG
    init_q: List[float] = []
    joint_slot: Dict[str, tuple] = {}
    com64 = np.zeros((L, 3))
    col_link: List[int] = []
    col_pos: List[np.ndarray] = []
    col_rad: List[float] = []
    friction = floor["friction"] if floor else 1.0
    iso = True
    nq = nqd = 0
    # <compiler settotalmass="M">: MuJoCo rescales every body's mass and inertia so that the model weighs M (the stock
    # half_cheetah.xml: settotalmass="14")
    settotal = float(comp.get("settotalmass", "-1")) if comp is not None else -1.0
    mscale = 1.0
    if settotal > 0:
        mscale = settotal / sum(g.mass_inertia()[0] for ent in links for g in ent["body"].geoms)
    for l, ent in enumerate(links):
        b: _Body = ent["body"]
        F["parent"][l] = ent["parent"]
        # ---- inertia from geoms -------------------------------------------------------------------
        if not b.geoms:
            raise ValueError(f"body {b.name!r} has no geoms: inertiafromgeom needs at least one")
        ms, cs, Is = [], [], []
        for g in b.geoms:
            m_g, I_g = g.mass_inertia()
            ms.append(m_g * mscale); cs.append(g.pos); Is.append(I_g * mscale)
        mass = float(sum(ms))
        com = sum(m_g * c for m_g, c in zip(ms, cs)) / mass
        I = np.zeros((3, 3))
        for m_g, c, I_g in zip(ms, cs, Is):
            d = c - com
            I += I_g + m_g * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        com64[l] = com
        lam, V = np.linalg.eigh(I)
        lam_s = lam ** (1.0 - custom["spring_inertia_scale"])
        Iinv = V @ np.diag(1.0 / lam_s) @ V.T
        # round-off of the eigen-decomposition is not structure: off-diagonal entries 1e-12 below the trace are
        # exact zeros (axis-aligned capsules and boxes then have exactly diagonal tensors, which the kernels use)
        Iinv[np.abs(Iinv) < 1e-12 * np.trace(Iinv)] = 0.0
        F["inv_mass"][l] = 1.0 / (mass ** (1.0 - custom["spring_mass_scale"]))
        F["inv_inertia"][l] = [Iinv[0, 0], Iinv[1, 1], Iinv[2, 2], Iinv[0, 1], Iinv[0, 2], Iinv[1, 2]]
        if not (np.allclose(Iinv, Iinv[0, 0] * np.eye(3), rtol=1e-6, atol=1e-9)):
            iso = False
        F["com"][l] = com
        ent["mass"], ent["inertia"] = mass, I
        # ---- joints ---------------------------------------------------------------------------------
        kinds = [j.get("type", "hinge") for j in b.joints]
        F["q_idx"][l], F["qd_idx"][l] = nq, nqd
        F["ang_damp"][l] = custom["constraint_ang_damping"]
        F["vel_damp"][l] = custom["constraint_vel_damping"]
        if kinds == ["free"]:
            F["n_rot"][l] = -1
            if ent["parent"] != -1:
                raise ValueError("free joint below the root")
            init_q += list(b.pos) + list(b.quat)
            nq += 7; nqd += 6
            # free links carry no link transform (q holds the world pose)
            F["link_pos"][l] = 0
            continue
        if "free" in kinds or "ball" in kinds:
            raise ValueError(f"body {b.name!r}: joint mix {kinds} is outside the hot-path subset")
        slides = [j for j in b.joints if j.get("type", "hinge") == "slide"]
        hinges = [j for j in b.joints if j.get("type", "hinge") == "hinge"]
        if b.joints[: len(slides)] != slides:
            raise ValueError(f"body {b.name!r}: slide joints must precede hinge joints")
        if len(slides) > 3 or len(hinges) > 3:
            raise ValueError(f"body {b.name!r}: more than 3 dofs of one kind")
        F["n_rot"][l], F["n_slide"][l] = len(hinges), len(slides)
        F["link_pos"][l], F["link_rot"][l] = b.pos, b.quat
        anchor = _floats((hinges or slides)[0].get("pos"), 3, np.zeros(3))
        for j in hinges:
            if not np.allclose(_floats(j.get("pos"), 3, np.zeros(3)), anchor):
                raise ValueError(f"body {b.name!r}: hinge joints of one link must share their anchor")
        F["joint_pos"][l] = anchor
        axes = [(_floats(j.get("axis"), 3, np.array([0, 0, 1.0]))) for j in hinges]
        axes = [a / np.linalg.norm(a) for a in axes]
        sign3 = 1.0
        if len(axes) == 0:
            Rj = np.eye(3)
        elif len(axes) == 1:
            Rj = _q2mat(_from_to([1.0, 0, 0], axes[0]))
        else:
            if abs(np.dot(axes[0], axes[1])) > 1e-6:
                raise ValueError(f"body {b.name!r}: hinge axes must be orthogonal")
            e3 = np.cross(axes[0], axes[1])
            if len(axes) == 3:
                d3 = float(np.dot(e3, axes[2]))
                if abs(abs(d3) - 1) > 1e-6:
                    raise ValueError(f"body {b.name!r}: third hinge axis must be +-(a1 x a2)")
                sign3 = 1.0 if d3 > 0 else -1.0
            Rj = np.stack([axes[0], axes[1], e3], axis=1)
        qj = _mat2q(Rj)
        for k, (j, a) in enumerate(zip(hinges, axes)):
            F["rot_axis"][l, k] = a
            s = sign3 if k == 2 else 1.0
            F["rot_sign"][l, k] = s
            limited = j.get("limited")
            has_range = "range" in j
            lim = (limited == "true") if limited in ("true", "false") else has_range
            if lim and has_range:
                lo, hi = _floats(j["range"], 2) * angle_scale
                lo, hi = (lo, hi) if s > 0 else (-hi, -lo)
            else:
                lo, hi = -_BIG, _BIG
            F["rot_lo"][l, k], F["rot_hi"][l, k] = lo, hi
            F["rot_stiff"][l, k] = float(j.get("stiffness", 0.0)) if passive_joint_forces else 0.0
            F["rot_damp"][l, k] = float(j.get("damping", 0.0)) if passive_joint_forces else 0.0
            joint_slot[j.get("name", f"{b.name}_h{k}")] = (l, k, s)
        for k, j in enumerate(slides):
            a = _floats(j.get("axis"), 3, np.array([0, 0, 1.0]))
            a = a / np.linalg.norm(a)
            F["slide_axis_body"][l, k] = a
            F["slide_axis"][l, k] = Rj.T @ a
            limited = j.get("limited")
            if (limited == "true" or (limited not in ("true", "false") and "range" in j)) and "range" in j:
                F["slide_lo"][l, k], F["slide_hi"][l, k] = _floats(j["range"], 2)
            F["slide_damp"][l, k] = float(j.get("damping", 0.0)) if passive_joint_forces else 0.0
            if float(j.get("stiffness", 0.0)) != 0.0:
                raise ValueError(f"body {b.name!r}: slide joint stiffness is outside the hot-path subset")
            joint_slot[j.get("name", f"{b.name}_s{k}")] = (l, 3 + k, 1.0)
        p = ent["parent"]
        pcom = com64[p] if p >= 0 else np.zeros(3)
        F["ap_pos"][l] = b.pos + _rot(anchor, b.quat) - pcom
        F["ap_rot"][l] = _qmul(b.quat, qj)
        F["ac_pos"][l] = anchor - com
        F["ac_rot"][l] = qj
        init_q += [0.0] * (len(slides) + len(hinges))
        nq += len(slides) + len(hinges); nqd += len(slides) + len(hinges)
        # ---- colliders (vs the floor plane) -------------------------------------------------------
    for l, ent in enumerate(links):
        for g in ent["body"].geoms:
            if floor is None:
                continue
            if (g.contype & floor["conaffinity"]) | (floor["contype"] & g.conaffinity) or \
                    (collide_all_capsules and g.kind in ("sphere", "capsule")):
                for pos, rad in g.sphere_points():
                    col_link.append(l); col_pos.append(pos - com64[l]); col_rad.append(rad)
                friction = max(friction, g.friction)
    if len(col_link) > MAX_COL:
        raise ValueError(f"{len(col_link)} colliders > MBD_MAX_COL={MAX_COL}")
    if nq > MAX_Q:
        raise ValueError(f"n_q {nq} > MBD_MAX_Q")

    # ---- actuators ----------------------------------------------------------------------------------
    act_link, act_slot, act_gear, act_lo, act_hi, act_names = [], [], [], [], [], []
    act = root.find("actuator")
    if act is not None:
        for mtr in act:
            if mtr.tag != "motor":
                raise ValueError(f"actuator <{mtr.tag}> is outside the hot-path subset (motor only)")
            a = _merged(defaults, "motor", mtr)
            l, slot, s = joint_slot[a["joint"]]
            gear = float(a.get("gear", "1").split()[0])
            limited = a.get("ctrllimited", "auto")
            if limited == "true" or (limited == "auto" and "ctrlrange" in a):
                lo, hi = _floats(a["ctrlrange"], 2)
            else:
                lo, hi = -_BIG, _BIG
            act_link.append(l); act_slot.append(slot); act_gear.append(gear * s)
            act_lo.append(lo); act_hi.append(hi); act_names.append(a.get("name", a["joint"]))
    if len(act_link) > MAX_ACT:
        raise ValueError("too many actuators")
    if len(gear_override):  # the env class replaces sys.actuator.gear (brax ant / half_cheetah, positional backend)
        if len(gear_override) != len(act_gear):
            raise ValueError(f"gear_override has {len(gear_override)} entries for {len(act_gear)} actuators")
        act_gear = [math.copysign(float(g), old) for g, old in zip(gear_override, act_gear)]
    for k, off in enumerate(init_q_offset):
        init_q[k] += float(off)
    names = [ent["body"].name for ent in links]
    track = [names.index(n) for n in track_names]
    if len(track) > MAX_TRACK:
        raise ValueError("too many tracked links")

    F.update(
        n_links=L, n_q=nq, n_qd=nqd, n_act=len(act_link), n_col=len(col_link), n_track=len(track),
        n_frames=int(n_frames), reward_kind=REWARD_KINDS.get(env_name, 0), iso_inertia=int(iso),
        # (MBD_FLAG_PLANAR is added below, once the model is complete; spec_flags: the specification switches, model.SPEC_FLAGS)
        flags=int(1 if reset_quat_raw else 0) | (int(DEFAULT_SPEC if spec_flags is None else spec_flags) & SPEC_MASK),
        dt=np.float32(dt), vel_fac=np.float32(math.exp(custom["vel_damping"] * dt)),
        ang_fac=np.float32(math.exp(custom["ang_damping"] * dt)),
        joint_scale_pos=np.float32(custom["joint_scale_pos"]),
        joint_scale_ang=np.float32(custom["joint_scale_ang"]),
        collide_scale=np.float32(custom["collide_scale"]), friction=np.float32(friction),
        elasticity=np.float32(custom["elasticity"]), gravity=np.asarray(gravity, np.float32),
        reset_noise=np.float32(reset_noise),
        reward_params=np.asarray(list(reward_params) + [0.0] * (8 - len(reward_params)), np.float32),
        act_link=np.asarray(act_link, np.int32), act_slot=np.asarray(act_slot, np.int32),
        act_gear=np.asarray(act_gear, np.float32), act_lo=np.asarray(act_lo, np.float32),
        act_hi=np.asarray(act_hi, np.float32),
        col_link=np.asarray(col_link, np.int32),
        col_pos=np.asarray(col_pos, np.float32).reshape(-1, 3),
        col_radius=np.asarray(col_rad, np.float32),
        init_q=np.asarray(init_q, np.float32), track_link=np.asarray(track, np.int32),
    )
    # python scalars for the scalar fields
    for k in ("dt", "vel_fac", "ang_fac", "joint_scale_pos", "joint_scale_ang", "collide_scale",
              "friction", "elasticity", "reset_noise"):
        F[k] = float(F[k])
    if planar is None or planar:
        ok = is_planar(F)
        if planar and not ok:
            raise ValueError("planar=True but the model does not move in the x-z plane")
        if ok:
            F["flags"] = int(F["flags"]) | 2  # MBD_FLAG_PLANAR
    model = Model(F, names, act_names, env_name)
    model.masses = np.array([ent["mass"] for ent in links])      # diagnostics / tests only
    model.inertias = np.stack([ent["inertia"] for ent in links])
    if warn_unstable:
        import warnings
        for line in stability_report(model):
            warnings.warn(f"{os.path.basename(path)}: {line}", stacklevel=2)
    return model
