"""TEST INFRASTRUCTURE ONLY — a second, deliberately dumb reader of the MJCF model files (round-3 verdict item 4).

The product's MJCF compiler (mbd_hip/mjcf.py) feeds BOTH the kernels and the CPU checker, so a compile error is invisible
to every bit-exact GPU-vs-checker test (round 3 found three such defects by other means).  This reader shares no code with
it and works differently on purpose: no model struct, no link frames, no fusing of bodies, no joint frames — it walks the
XML once, carries every body's WORLD transform at the pose the file is written in, and returns per link (= body that owns
joints; geoms of joint-less descendants are simply attributed to it) plain world-frame facts:

    mass, centre of mass, inertia tensor about it (capsule = cylinder + two hemispheres, assembled with the parallel-axis
    theorem from textbook closed forms), joint anchor, hinge / slide axes, the spheres that can touch the floor, and the
    actuators (joint name, gear) in file order.

tests/test_model_crosscheck.py compares them with what the compiled models (assets/compiled/*.json) say about the same
quantities after the checker's forward kinematics.  Never imported by the product."""
import math
import xml.etree.ElementTree as ET

import numpy as np


def _q2m(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _nums(s, default):
    return np.array([float(t) for t in s.split()]) if s is not None else np.asarray(default, float)


def _solid(kind, a, b, r, rho):
    """(mass, com, inertia about com) of a sphere at `a` or a capsule from `a` to `b`, radius r, density rho — world frame.
    Pieces: point-like solids with known central tensors, combined by the parallel-axis theorem."""
    pieces = []  # (mass, centre, central inertia tensor)
    if kind == "sphere":
        m = rho * 4.0 / 3.0 * math.pi * r ** 3
        pieces.append((m, a, 0.4 * m * r * r * np.eye(3)))
    else:
        L = float(np.linalg.norm(b - a))
        u = (b - a) / L
        uu = np.outer(u, u)
        mc = rho * math.pi * r * r * L                                   # the cylinder
        pieces.append((mc, 0.5 * (a + b), mc * r * r / 2.0 * uu + mc * (3 * r * r + L * L) / 12.0 * (np.eye(3) - uu)))
        mh = rho * 2.0 / 3.0 * math.pi * r ** 3                           # a hemisphere: COM 3r/8 from its flat face
        Ih = 0.4 * mh * r * r * uu + (83.0 / 320.0) * mh * r * r * (np.eye(3) - uu)
        pieces.append((mh, b + 0.375 * r * u, Ih))
        pieces.append((mh, a - 0.375 * r * u, Ih))
    return pieces


def combine(pieces):
    m = sum(p[0] for p in pieces)
    c = sum(p[0] * p[1] for p in pieces) / m
    I = np.zeros((3, 3))
    for mp, cp, Ip in pieces:
        d = cp - c
        I += Ip + mp * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, I


def read(path, drop_suffix=None):
    root = ET.parse(path).getroot()
    comp = root.find("compiler")
    deg = not (comp is not None and comp.get("angle", "degree") == "radian")
    dflt = {}
    if root.find("default") is not None:
        for ch in root.find("default"):
            dflt[ch.tag] = dict(ch.attrib)
    custom = {}
    if root.find("custom") is not None:
        for n in root.find("custom").findall("numeric"):
            custom[n.get("name")] = float(n.get("data").split()[0])
    world = root.find("worldbody")
    floor = None
    for g in world.findall("geom"):
        a = dict(dflt.get("geom", {}), **g.attrib)
        if a.get("type") == "plane":
            floor = (int(a.get("contype", 1)), int(a.get("conaffinity", 1)))
    links = []

    def geoms_of(body, pos, R, link):
        for g in body.findall("geom"):
            a = dict(dflt.get("geom", {}), **g.attrib)
            kind = a.get("type", "sphere")
            if kind == "plane":
                continue
            size = _nums(a.get("size"), [0.0])
            rho = float(a.get("density", 1000.0))
            if kind == "capsule" and "fromto" in a:
                ft = _nums(a["fromto"], None)
                p0, p1 = pos + R @ ft[:3], pos + R @ ft[3:]
            elif kind == "capsule":
                Rg = np.eye(3)
                if "axisangle" in a:
                    aa = _nums(a["axisangle"], None)
                    ang = aa[3] * (math.pi / 180 if deg else 1.0)
                    k = aa[:3] / np.linalg.norm(aa[:3])
                    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
                    Rg = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)   # Rodrigues
                elif "quat" in a:
                    Rg = _q2m(_nums(a["quat"], None))
                c = pos + R @ _nums(a.get("pos"), [0, 0, 0])
                h = R @ Rg @ np.array([0, 0, size[1]])
                p0, p1 = c - h, c + h
            elif kind == "sphere":
                p0 = p1 = pos + R @ _nums(a.get("pos"), [0, 0, 0])
            else:
                raise ValueError(kind)
            link["pieces"] += _solid(kind, p0, p1, float(size[0]), rho)
            ct, ca = int(a.get("contype", 1)), int(a.get("conaffinity", 1))
            if floor is not None and ((ct & floor[1]) | (floor[0] & ca)):
                ends = [p0] if kind == "sphere" else [p1, p0]
                link["colliders"] += [(e, float(size[0])) for e in ends]

    def walk(body, ppos, pR, owner):
        name = body.get("name", "")
        if drop_suffix and name.endswith(drop_suffix):
            return
        pos = ppos + pR @ _nums(body.get("pos"), [0, 0, 0])
        R = pR @ _q2m(_nums(body.get("quat"), [1, 0, 0, 0]))
        joints = [dict(dflt.get("joint", {}), **j.attrib) for j in body.findall("joint")]
        link = owner
        if joints:
            link = dict(name=name, parent=(links.index(owner) if owner is not None else -1), pieces=[], colliders=[],
                        pos=pos, R=R, joints=[])
            links.append(link)
            for j in joints:
                kind = j.get("type", "hinge")
                ax = _nums(j.get("axis"), [0, 0, 1])
                link["joints"].append(dict(name=j.get("name", ""), kind=kind,
                                           anchor=pos + R @ _nums(j.get("pos"), [0, 0, 0]),
                                           axis=R @ (ax / np.linalg.norm(ax)),
                                           range=_nums(j.get("range"), [0, 0]) * (math.pi / 180 if deg and kind == "hinge" else 1.0),
                                           limited=j.get("limited", "auto"), has_range="range" in j))
        if link is None:
            raise ValueError(f"body {name!r} has no jointed ancestor")
        geoms_of(body, pos, R, link)
        for ch in body.findall("body"):
            walk(ch, pos, R, link)

    for b in world.findall("body"):
        walk(b, np.zeros(3), np.eye(3), None)
    total = sum(p[0] for l in links for p in l["pieces"])
    scale = 1.0
    if comp is not None and float(comp.get("settotalmass", "-1")) > 0:
        scale = float(comp.get("settotalmass")) / total
    for l in links:
        m, c, I = combine(l["pieces"])
        l["mass"], l["com"], l["inertia"] = m * scale, c, I * scale
        del l["pieces"]
    acts = []
    if root.find("actuator") is not None:
        for mtr in root.find("actuator"):
            a = dict(dflt.get(mtr.tag, {}), **mtr.attrib)
            acts.append((a["joint"], float(a.get("gear", "1").split()[0]),
                         _nums(a.get("ctrlrange"), [-1e9, 1e9]) if a.get("ctrllimited", "auto") != "false" and "ctrlrange" in a else np.array([-1e9, 1e9])))
    return dict(links=links, actuators=acts, custom=custom, total_mass=total * scale)


def monte_carlo(kind, a, b, r, rho, n=400_000, seed=0):
    """mass, com, inertia of the same solid by sampling its bounding box — the check of the closed forms themselves"""
    g = np.random.default_rng(seed)
    lo, hi = np.minimum(a, b) - r, np.maximum(a, b) + r
    x = lo + (hi - lo) * g.random((n, 3))
    if kind == "sphere":
        inside = np.linalg.norm(x - a, axis=1) <= r
    else:
        ab = b - a
        t = np.clip(((x - a) @ ab) / (ab @ ab), 0.0, 1.0)
        inside = np.linalg.norm(x - (a + t[:, None] * ab), axis=1) <= r
    vol = np.prod(hi - lo) * inside.mean()
    pts = x[inside]
    c = pts.mean(axis=0)
    d = pts - c
    I = rho * vol * ((d * d).sum(axis=1).mean() * np.eye(3) - (d[:, :, None] * d[:, None, :]).mean(axis=0))
    return rho * vol, c, I
