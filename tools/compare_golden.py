"""Replay the substep of a golden file (tools/dump_golden.py, record (B)) through this repo's CPU oracle STAGE BY
STAGE and name the first stage and link whose pose or velocity differs from Brax's by more than the tolerance —
i.e. which of DESIGN.md §9's guesses is wrong — after first comparing the compiled system (masses, inertias).

    python tools/compare_golden.py tests/golden/golden_humanoidrun_N64_H50.npz [--tol 1e-5] [--flags N] [--search]

--flags N   replay under the word of specification switches N (include/mbd_hip.h mbd_model_flags / mbd_hip.model.SPEC_FLAGS:
            contact_avg 4, contact6_gauss_seidel 8, friction_vel_bound 16, restitution_min 32, euler_extrinsic 64, gyroscopic 128);
            without it: the model as compiled = the default word, 4 (contact_avg) since round 6
--search    replay under EVERY combination of the six switches and rank them: most stages within tolerance first, then the
            smallest error at the first mismatching stage, then the fewest switches flipped against the default word — the line
            to read is the first one; a winner other than the default word names the code-level guesses of DESIGN.md §9 that Brax
            decides the other way, and
            the model is then recompiled with that flag word (mjcf.load(spec_flags=...)): no kernel or checker rewrite.

Exit code 0: every stage within tolerance (--search: under the best combination); 1: a mismatch was reported; 2: the file
has no stage records.
Importable: compare(path, tol, flags=0) -> list of report lines, first_mismatch (None or (stage, link, quantity, err));
search(path, tol) -> list of (flags, names, first_mismatch, stages_ok, err_at_first) best first."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

STAGES = ["1_acceleration", "2_integrate", "3_joint_position", "4_contact_position", "5_project", "6_contact_velocity"]
SUSPECTS = {
    "sys": "the MJCF compile (inertia from geoms, body fusing, frames): mbd_hip/mjcf.py",
    "1_acceleration": "actuator gears / ctrlrange, the Euler-angle convention of multi-dof joints, MJCF joint stiffness & "
                      "damping applied on top of constraint_{ang,vel}_damping (mjcf.load(passive_joint_forces=False))",
    "2_integrate": "velocity damping factors exp(damping*dt), the first-order quaternion update",
    "3_joint_position": "joint_scale_pos / joint_scale_ang, the alignment error by joint type, the Euler-angle limits",
    "4_contact_position": "contact point / normal of sphere-plane pairs, collide_scale, static friction",
    "5_project": "velocity from the pose difference (quaternion difference convention)",
    "6_contact_velocity": "the friction-impulse bound mu*lambda_n/dt, restitution",
}


def _state(g, prefix):
    return np.concatenate([g[f"{prefix}_x_pos"], g[f"{prefix}_x_rot"], g[f"{prefix}_xd_vel"], g[f"{prefix}_xd_ang"]],
                          axis=1).astype(np.float32)


def compare(path, tol=1e-5, flags=None):
    from mbd_hip.model import Model
    from oracle import oracle as orc_mod
    import ctypes as C
    g = np.load(path)
    name = os.path.basename(path).split("_")[1]
    with open(os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")) as f:
        m = Model.from_json(f.read())
    if flags is not None:  # (None: the model as compiled — the default word, model.DEFAULT_SPEC)
        m = m.with_spec(flags)
    ms = m.to_struct()
    L = m.n_links
    lines, first = [], None
    if "substep_in_x_pos" not in g:
        return [f"{path}: no substep records (made by an older tools/dump_golden.py)"], ("none", -1, "", 0.0)
    # ---- the compiled system ------------------------------------------------------------------------------
    if "sys_link_mass" in g:
        mass = 1.0 / np.asarray(m.fields["inv_mass"], np.float64)[:L]
        gm = np.asarray(g["sys_link_mass"], np.float64)
        if gm.shape[0] != L:
            lines.append(f"sys: Brax has {gm.shape[0]} links, the compiled model {L} (dropped marker links?)")
            gm = gm[:L]
        err = np.abs(mass - gm) / np.maximum(np.abs(gm), 1e-12)
        lines.append(f"sys: link masses max rel err {err.max():.3g} (link {int(err.argmax())})")
        if err.max() > 1e-4 and first is None:
            first = ("sys", int(err.argmax()), "mass", float(err.max()))
        gear = np.asarray(g["sys_actuator_gear"], np.float64)
        mine = np.abs(np.asarray(m.fields["act_gear"], np.float64))
        if gear.shape == mine.shape:
            e = np.abs(np.abs(gear) - mine).max()
            lines.append(f"sys: actuator gears max abs err {e:.3g}")
            if e > 1e-4 and first is None:
                first = ("sys", int(np.abs(np.abs(gear) - mine).argmax()), "gear", float(e))
    # ---- stage by stage -----------------------------------------------------------------------------------
    orc_mod.build()
    orc = orc_mod.Oracle("f32")
    orc.lib.orc_substep_stages.argtypes = [C.c_void_p] + [np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")] * 4
    for prefix in ("", "contact_"):  # the substep from state_init, then the one from a settled state (contacts active)
        if f"{prefix}substep_in_x_pos" not in g:
            continue
        if prefix:
            n_touch = int((np.asarray(g[f"{prefix}stage_4_contact_position_contact_dist"]) < 0).sum()) \
                if f"{prefix}stage_4_contact_position_contact_dist" in g else -1
            lines.append(f"---- substep from the settled state ({n_touch if n_touch >= 0 else '?'} penetrating contacts)")
        first = _compare_substep(g, prefix, m, ms, L, orc, tol, lines, first)
    if "bounce_vz" in g:  # record (C): does a ball with elasticity e come back up?
        vz = np.asarray(g["bounce_vz"], np.float64)
        hit = int(np.argmin(vz[:400]))
        ratio = float(vz[hit:hit + 30].max() / -vz[hit]) if vz[hit] < 0 else 0.0
        e = float(g["bounce_elasticity"])
        verdict = ("max(-e vn, 0): this repo's default" if abs(ratio - e) < 0.1 else
                   ("min(-e vn, 0): MBD_FLAG_RESTITUTION_MIN (32)" if ratio < 0.1 else "neither form"))
        lines.append(f"restitution: rebound / impact speed = {ratio:.3f} at elasticity {e:g} -> {verdict}")
    if first is not None:
        st = first[0]
        lines.append(f"FIRST MISMATCH: stage {st}, link {first[1]} ({m.link_names[first[1]] if 0 <= first[1] < L else '?'}), "
                     f"{first[2]}: err {first[3]:.3g} > {tol:g}")
        lines.append(f"  suspects: {SUSPECTS.get(st.replace('contact:', ''), 'see the stage-by-stage lines above')}")
    else:
        lines.append(f"all stages within {tol:g}")
    return lines, first


def _compare_substep(g, prefix, m, ms, L, orc, tol, lines, first):
    import ctypes as C
    tag = "contact:" if prefix else ""
    P = prefix
    s_in = _state(g, f"{P}substep_in")[:L]
    act = np.ascontiguousarray(g[f"{P}substep_action"], np.float32)
    out = np.zeros((L, 13), np.float32)
    stages = np.zeros((6, L, 13), np.float32)
    orc.lib.orc_substep_stages(C.addressof(ms), np.ascontiguousarray(s_in).reshape(-1), act, out.reshape(-1), stages.reshape(-1))
    staged = bool(g[f"{P}stage_composition_matches_pipeline_step"]) if f"{P}stage_composition_matches_pipeline_step" in g else False
    cols = {"pos": slice(0, 3), "rot": slice(3, 7), "vel": slice(7, 10), "ang": slice(10, 13)}
    if staged:
        for k, st in enumerate(STAGES):
            if st == "1_acceleration":  # accelerations (gravity excluded here, included there): compare them minus gravity
                ref_v = np.asarray(g[f"{P}stage_{st}_xdd_vel"], np.float32)[:L] - np.asarray(m.fields["gravity"], np.float32)
                ref_w = np.asarray(g[f"{P}stage_{st}_xdd_ang"], np.float32)[:L]
                pairs = {"lin. accel": (stages[k][:, 7:10], ref_v), "ang. accel": (stages[k][:, 10:13], ref_w)}
            else:
                ref = _state(g, f"{P}stage_{st}")[:L]
                pairs = {q: (stages[k][:, c], ref[:, c]) for q, c in cols.items()}
                # q and -q are the same rotation
                a, b = pairs["rot"]
                sign = np.sign(np.sum(a * b, axis=1, keepdims=True))
                pairs["rot"] = (a * np.where(sign == 0, 1, sign), b)
            for q, (a, b) in pairs.items():
                e = np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))
                lines.append(f"stage {st:20s} {q:10s} max err {e.max():.3g} (link {int(e.argmax())})")
                if e.max() > tol and first is None:
                    first = (tag + st, int(e.argmax()), q, float(e.max()))
    else:
        lines.append("stage records absent or flagged (this Brax composes its step differently): end-of-substep only")
    ref = _state(g, f"{P}substep_out")[:L]
    for q, c in cols.items():
        a, b = out[:, c], ref[:, c]
        if q == "rot":
            sign = np.sign(np.sum(a * b, axis=1, keepdims=True))
            a = a * np.where(sign == 0, 1, sign)
        e = np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))
        lines.append(f"end of substep        {q:10s} max err {e.max():.3g} (link {int(e.argmax())})")
        if e.max() > tol and first is None:
            first = (tag + "end_of_substep", int(e.argmax()), q, float(e.max()))
    return first


def search(path, tol=1e-5):
    """compare() under every combination of the specification switches; best first (see the module docstring)."""
    from mbd_hip.model import SPEC_FLAGS, spec_names
    bits = sorted(SPEC_FLAGS.values())
    order = ["sys"] + [p + s for p in ("", "contact:") for s in STAGES + ["end_of_substep"]]
    rows = []
    for mask in range(1 << len(bits)):
        flags = sum(b for k, b in enumerate(bits) if mask >> k & 1)
        lines, first = compare(path, tol, flags)
        if first is not None and first[0] == "none":
            return [(0, [], first, 0, 0.0)]
        ok = len(order) if first is None else (order.index(first[0]) if first[0] in order else 0)
        rows.append((flags, spec_names(flags), first, ok, 0.0 if first is None else first[3]))
    # (ties: the word closest to the default one — fewest switches flipped against model.DEFAULT_SPEC)
    from mbd_hip.model import DEFAULT_SPEC
    rows.sort(key=lambda r: (-r[3], r[4], bin(r[0] ^ DEFAULT_SPEC).count("1"), r[0]))
    return rows


if __name__ == "__main__":
    tol = float(sys.argv[sys.argv.index("--tol") + 1]) if "--tol" in sys.argv else 1e-5
    if "--search" in sys.argv:
        rows = search(sys.argv[1], tol)
        for flags, names, first, ok, err in rows[:12]:
            where = "all stages within tolerance" if first is None else f"first mismatch {first[0]} link {first[1]} {first[2]} err {err:.3g}"
            print(f"flags {flags:3d} [{', '.join(names) or 'none'}]{' = the default word' if flags == 4 else ''}: {where}")
        best = rows[0]
        print(f"BEST: flags {best[0]} ({', '.join(best[1]) or 'no switch set'}{': the default specification' if best[0] == 4 else ''})")
        sys.exit(0 if best[2] is None else (2 if best[2][0] == "none" else 1))
    flags = int(sys.argv[sys.argv.index("--flags") + 1]) if "--flags" in sys.argv else None
    lines, first = compare(sys.argv[1], tol, flags)
    print("\n".join(lines))
    sys.exit(0 if first is None else (2 if first[0] == "none" else 1))
