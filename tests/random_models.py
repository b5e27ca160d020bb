"""Random MJCF models inside the hot-path subset of mbd_hip/mjcf.py, for fuzzing what only custom files reach: link trees of
random shape (up to 4 children on a link, chains up to depth 5), 1 / 2 / 3-dof hinge joints, slide + hinge joints, fused
(joint-less) bodies, skew capsules (full inertia tensors), joint springs and dampers, limited and unlimited ranges, one or
two sphere colliders on some links, actuators on a shuffled subset of the dofs, random <custom> numerics.

    random_mjcf(seed) -> XML text            (deterministic in the seed)

Used by tests/test_random_models.py: compile -> the independent reader's facts (oracle/model_reader.py) -> checker rollouts
stay finite -> (GPU) the general kernels reproduce the checker bit for bit."""
import numpy as np


def _v(x):
    return " ".join(f"{float(t):.5g}" for t in x)


def random_mjcf(seed, max_bodies=14, kinds=("h1", "h1", "h2", "h3", "sh", "none"), probs=(0.3, 0.2, 0.2, 0.12, 0.1, 0.08),
                springs=True, sis=(0, 0.5, 1.0), planar=False):
    """planar=True: a model that moves in the x-z plane only (root = slide x, slide z, hinge y like hopper / walker2d /
    halfcheetah; every other joint one hinge about +-y; geoms in the plane) — what the planar kernels take."""
    g = np.random.default_rng(seed)
    n_bodies = int(g.integers(min(3, max_bodies), max_bodies + 1))
    parent, depth, kids = [-1], [0], [0]
    for b in range(1, n_bodies):
        cand = [p for p in range(b) if kids[p] < (4 if p == 0 else 3) and depth[p] < 5]
        p = int(g.choice(cand))
        parent.append(p); depth.append(depth[p] + 1); kids.append(0); kids[p] += 1
    sis_v = float(g.choice(list(sis)))
    # a capsule per body from its origin to a tip; children hang at the tip or half way
    tip, rad = [], []
    for b in range(n_bodies):
        d = g.normal(size=3) * np.array([1.0, 0.0 if planar else 1.0, 0.6]) + (np.array([0, 0, -1.0]) if b else 0)
        if planar and sis_v < 1.0:   # (the planar restatement wants diagonal tensors: true tensors of axis-aligned capsules, or
            d = np.array([[1.0, 0, 0], [-1.0, 0, 0], [0, 0, -1.0]][int(g.integers(0, 3 if b else 2))])   # spring_inertia_scale = 1)
        d = d / np.linalg.norm(d) * g.uniform(0.14, 0.32)
        # (parents heavier than their children: the joint stage sums the corrections of ALL joints of a link, Jacobi-style;
        # a light link between heavy neighbours overshoots — e.g. 4 joints x joint_scale_pos 0.7 on a 1.5 kg root carrying
        # 2-5 kg children diverges in free fall — which is why the reference's models have heavy torsos and scales 0.5 / 0.2)
        tip.append(d); rad.append(float(g.uniform(0.085, 0.1) if b == 0 else g.uniform(0.03, 0.06) * 0.85 ** depth[b]))
    n_col, n_act_max = 0, 22
    # explicit dampers and motors are stable only below ~2 I / dt: with spring_inertia_scale = 1 every tensor is the identity
    # (1 kg m^2, what the reference's humanoids use with constraint_ang_damping = 30); below it a thin capsule's axial inertia
    # is ~5e-4 kg m^2 and the same coefficients are far outside the stable range of ANY explicit integrator
    soft = sis_v < 1.0
    joints_of, lines = {}, []
    dofs = []   # (joint name, kind)

    def body(b, indent):
        nonlocal n_col
        pad = " " * indent
        pos = np.array([0.0, 0.0, 0.45 + 0.3 * max(depth)]) if b == 0 else tip[parent[b]] * (1.0 if g.random() < 0.7 else 0.5)
        out = [f'{pad}<body name="b{b}" pos="{_v(pos)}">']
        if b == 0 and planar:
            out += [f'{pad} <joint name="rootx" type="slide" axis="1 0 0"/>', f'{pad} <joint name="rootz" type="slide" axis="0 0 1"/>',
                    f'{pad} <joint name="rooty" type="hinge" axis="0 1 0"/>']
        elif b == 0:
            out.append(f'{pad} <joint type="free"/>')
        else:
            kind = "p1" if planar else g.choice(list(kinds), p=np.asarray(probs) / np.sum(probs))
            if kind == "none" and any(parent[c] == b for c in range(n_bodies)):
                kind = "h1"   # (a fused body in the middle of a chain would move its children onto the grandparent's link)
            js = []
            if kind == "p1":
                js.append(("hinge", np.array([0, 1.0 if g.random() < 0.5 else -1.0, 0])))
            elif kind == "h1":
                a = g.normal(size=3); a /= np.linalg.norm(a)
                js.append(("hinge", a))
            elif kind == "h2":
                js += [("hinge", np.array([0, 1.0, 0])), ("hinge", np.array([1.0, 0, 0]))]
            elif kind == "h3":
                js += [("hinge", np.array([0, 0, 1.0])), ("hinge", np.array([0, 1.0, 0])), ("hinge", np.array([1.0, 0, 0]))]
            elif kind == "sh":
                a = g.normal(size=3); a /= np.linalg.norm(a)
                js += [("slide", a), ("hinge", np.array([0, 1.0, 0]))]
            for k, (jk, ax) in enumerate(js):
                name = f"j{b}_{k}"
                lo, hi = -float(g.uniform(0.2, 1.2)), float(g.uniform(0.2, 1.2))
                if jk == "slide":
                    lo, hi = -float(g.uniform(0.02, 0.1)), float(g.uniform(0.02, 0.15))
                attrs = f'name="{name}" type="{jk}" axis="{_v(ax)}"'
                # (the MIDDLE hinge of a two- or three-dof joint is always limited: free to reach +-90 degrees it runs the joint's
                # Euler angles into their pole, where they are undefined — mjcf.stability_report names such joints; round-6 fuzz,
                # seed 2123: NaN in checker and kernel alike)
                limited = g.random() < 0.8
                if limited or (k == 1 and kind in ("h2", "h3")):
                    attrs += f' range="{lo:.4g} {hi:.4g}"'
                if g.random() < 0.3:
                    attrs += f' damping="{g.uniform(0.05, 1.0) * (0.001 if soft and jk == "hinge" else 1.0):.3g}"'
                if springs and jk == "hinge" and g.random() < 0.25:
                    attrs += f' stiffness="{g.uniform(0.5, 5.0) * (0.2 if soft else 1.0):.3g}"'
                out.append(f"{pad} <joint {attrs}/>")
                dofs.append((name, jk))
        out.append(f'{pad} <geom type="capsule" fromto="0 0 0 {_v(tip[b])}" size="{rad[b]:.4g}"/>')
        leaf = not any(parent[c] == b for c in range(n_bodies))
        if (leaf or g.random() < 0.2) and n_col < 15:
            out.append(f'{pad} <geom type="sphere" pos="{_v(tip[b])}" size="{rad[b] + 0.012:.4g}" contype="1" conaffinity="1"/>')
            n_col += 1
            if g.random() < 0.2 and n_col < 15:
                out.append(f'{pad} <geom type="sphere" pos="{_v(tip[b] * 0.4)}" size="{rad[b] + 0.008:.4g}" contype="1" conaffinity="1"/>')
                n_col += 1
        for c in range(n_bodies):
            if parent[c] == b:
                out += body(c, indent + 1)
        out.append(f"{pad}</body>")
        return out

    tree = body(0, 0)
    order = list(range(len(dofs)))
    g.shuffle(order)
    keep = order[:max(1, min(n_act_max, int(round(len(dofs) * g.uniform(0.5, 1.0)))))]
    acts = []
    for k in keep:
        name, jk = dofs[k]
        gear = g.uniform(10, 60) * (0.03 if soft and jk == "hinge" else 1.0) * (1.0 if jk == "hinge" else 2.0) * (1 if g.random() < 0.85 else -1)
        cr = 1.0 if g.random() < 0.7 else 0.5
        acts.append(f'<motor joint="{name}" gear="{gear:.4g}" ctrlrange="{-cr} {cr}"/>')
    custom = [f'<numeric name="spring_inertia_scale" data="{sis_v}"/>',
              f'<numeric name="spring_mass_scale" data="0"/>',
              f'<numeric name="constraint_ang_damping" data="{g.choice([0, 10, 30]) * (0.0001 if soft else 1.0)}"/>',
              f'<numeric name="constraint_vel_damping" data="{g.choice([0, 0.5, 5])}"/>']
    if g.random() < 0.5:
        custom += [f'<numeric name="joint_scale_pos" data="{g.choice([0.3, 0.5])}"/>',
                   f'<numeric name="joint_scale_ang" data="{g.choice([0.1, 0.2])}"/>']
    if g.random() < 0.3:
        custom.append(f'<numeric name="elasticity" data="{g.choice([0.1, 0.4])}"/>')
    xml = ['<mujoco><compiler angle="radian"/>', f'<option timestep="{g.choice([0.003, 0.004, 0.005])}"/>',
           "<custom>" + "".join(custom) + "</custom>",
           f'<default><geom contype="0" conaffinity="0" density="{g.choice([500, 900, 1000])}"/></default>',
           '<worldbody><geom type="plane" size="10 10 1" contype="1" conaffinity="1" friction="0.9 0.005 0.0001"/>']
    xml += tree
    xml += ["</worldbody>", "<actuator>" + "".join(acts) + "</actuator></mujoco>"]
    return "\n".join(xml)


def jacobi_load(model):
    """max over links of sum over its joints of w_link / (w_link + w_other) (translational inverse masses): how much of the
    summed joint corrections lands on the most loaded link.  The joint stage applies joint_scale_pos x this sum to a link
    per substep, and position-based dynamics (velocity = pose difference) diverges beyond ~4/3: the error recurrence of a
    relaxation alpha is e' = (1 - alpha)(2 e - e_prev), whose roots leave the unit circle at alpha = 4/3."""
    F, L = model.fields, model.n_links
    w = np.asarray(F["inv_mass"][:L], np.float64)
    load = np.zeros(L)
    for l in range(1, L):
        p = int(F["parent"][l])
        load[l] += w[l] / (w[l] + w[p])
        load[p] += w[p] / (w[l] + w[p])
    return float(load.max())


def stable_random_model(seed, compile_fn, **kw):
    """random_mjcf(seed) with joint_scale_pos / joint_scale_ang lowered, where needed, to what its most loaded link
    tolerates (jacobi_load); returns (xml, compiled model)."""
    import re
    xml = random_mjcf(seed, **kw)
    m = compile_fn(xml)
    load = jacobi_load(m)
    jsp = min(float(m.fields["joint_scale_pos"]), 0.9 / load)
    jsa = min(float(m.fields["joint_scale_ang"]), 0.36 / load)
    xml = re.sub(r'<numeric name="joint_scale_(pos|ang)" data="[^"]*"/>', "", xml)
    xml = xml.replace("</custom>", f'<numeric name="joint_scale_pos" data="{jsp:.4g}"/><numeric name="joint_scale_ang" data="{jsa:.4g}"/></custom>')
    return xml, compile_fn(xml)
