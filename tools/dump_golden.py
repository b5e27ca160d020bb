"""Golden-vector escape hatch: run THIS under a real `pip install jax brax` next to the
reference checkout to dump what the reference itself computes; commit the .npz under tests/golden/ and
tests/test_golden.py will hold this repo to it.  It only *consumes* the reference (imports `mbd`).

    python tools/dump_golden.py /path/to/model-based-diffusion humanoidrun 64 50 3

Two kinds of records go into golden_<env>_N<N>_H<H>.npz:

 (A) per diffusion step, the inputs and outputs of reverse_once (mbd_planner.py:97-135): key, eps, Y0s, rewss,
     weights, Ybar, tracked positions — the teacher-forced END-TO-END comparison;
 (B) ONE physics substep from state_init under a fixed action, dumped STAGE BY STAGE: Brax's x_i / xd_i after
     joints.acceleration_update (+ the accelerations), integrator.integrate_xdd, joints.position_update,
     collisions.resolve_position, integrator.project_xd and collisions.resolve_velocity, plus the compiled `sys`
     (link masses, inertias, joint frames, actuator gears, the <custom> scalars).  tools/compare_golden.py replays
     the same substep through this repo's oracle stage by stage and names the FIRST stage and link that differ by
     more than 1e-5 — so a mismatch says which of DESIGN.md §9's guesses is wrong; `--search` replays it under every
     combination of the specification switches (mbd_model_flags) and names the combination that fits;
 (C) a ball with elasticity 0.5 dropped onto the floor (z, vz per substep): the restitution clamp's sign convention.

Nothing here runs in the build container (jax/brax are absent) and no golden is ever fabricated.
"""
import sys

import numpy as np


def _np(tree):
    import jax
    return jax.tree_util.tree_map(lambda a: np.asarray(a), tree)


def dump_substep_stages(env, state_init, action, out, prefix=""):
    """(B): re-composes brax/positional/pipeline.py::step from Brax's OWN stage functions, records x_i / xd_i after
    each, and checks the composition against pipeline.step itself — if the installed Brax composes its step
    differently the stage records are dropped (with a note) and only the end-of-substep state is kept.
    `prefix`: "" for the substep from state_init (usually in the air: stages 4 and 6 do nothing), "contact_" for the
    substep from a settled state (feet on the floor; links with two colliders have both in contact) — the record that
    decides DESIGN.md §9's guesses about friction bounds and several contacts on one link."""
    import jax
    from jax import numpy as jnp
    from brax import actuator, com, kinematics  # noqa: F401
    from brax.base import Motion
    from brax.positional import collisions, integrator, joints
    from brax.positional import pipeline
    sys_, st = env.sys, state_init.pipeline_state
    rec = {}

    def put(name, x_i, xd_i, extra=None):
        rec[f"stage_{name}_x_pos"], rec[f"stage_{name}_x_rot"] = np.asarray(x_i.pos), np.asarray(x_i.rot)
        rec[f"stage_{name}_xd_vel"], rec[f"stage_{name}_xd_ang"] = np.asarray(xd_i.vel), np.asarray(xd_i.ang)
        for k, v in (extra or {}).items():
            rec[f"stage_{name}_{k}"] = np.asarray(v)

    def call(fn, *candidates):
        """Brax's stage functions changed their argument lists between releases: try the known forms in order."""
        err = None
        for args in candidates:
            try:
                return fn(*args)
            except TypeError as e:
                err = e
        raise err

    def tadd(a, b):  # brax.base.Base.__add__: leaf-wise
        return jax.tree_util.tree_map(jnp.add, a, b)

    with jax.disable_jit():
        ref_next = pipeline.step(sys_, st, action)
        try:
            from brax import geometry
            tau = actuator.to_tau(sys_, action, st.q, st.qd)
            xdd_i = tadd(joints.acceleration_update(sys_, st, tau), Motion.create(vel=sys_.gravity))
            put("1_acceleration", st.x_i, st.xd_i, dict(xdd_vel=xdd_i.vel, xdd_ang=xdd_i.ang, tau=tau))
            x_i_prev = st.x_i
            x_i, xd_i = integrator.integrate_xdd(sys_, st.x_i, st.xd_i, xdd_i)
            put("2_integrate", x_i, xd_i)
            p_j = call(joints.position_update, (sys_, x_i), (sys_, st.replace(x_i=x_i)))
            x_i = tadd(x_i, p_j)
            put("3_joint_position", x_i, xd_i)
            x_w = x_i.vmap().do(sys_.link.inertia.transform.inv())  # com.to_world: contacts are found in world frames
            contact = geometry.contact(sys_, x_w)
            p_i, dlambda = call(collisions.resolve_position, (sys_, x_i, x_i_prev, contact),
                                (sys_, st.replace(x_i=x_i), x_i_prev, contact))
            x_i = tadd(x_i, p_i)
            extra4 = dict(dlambda=dlambda) if dlambda is not None else {}
            if contact is not None:  # which links touch, and how deep: how many contacts act on ONE link
                for k_, name_ in (("dist", "contact_dist"), ("link_idx", "contact_link_idx")):
                    if hasattr(contact, k_):
                        extra4[name_] = np.asarray(getattr(contact, k_))
            put("4_contact_position", x_i, xd_i, extra4 or None)
            xd_i_prev = xd_i
            xd_i = integrator.project_xd(sys_, x_i, x_i_prev)
            put("5_project", x_i, xd_i)
            xdv_i = call(collisions.resolve_velocity, (sys_, xd_i, xd_i_prev, contact, dlambda),
                         (sys_, x_i, xd_i, xd_i_prev, contact, dlambda))
            xd_i = tadd(xd_i, xdv_i)
            put("6_contact_velocity", x_i, xd_i)
            ok = bool(jnp.allclose(x_i.pos, ref_next.x_i.pos, atol=1e-6) and jnp.allclose(x_i.rot, ref_next.x_i.rot, atol=1e-6)
                      and jnp.allclose(xd_i.vel, ref_next.xd_i.vel, atol=1e-5) and jnp.allclose(xd_i.ang, ref_next.xd_i.ang, atol=1e-5))
            rec["stage_composition_matches_pipeline_step"] = np.asarray(ok)
            if not ok:
                print("NOTE: this Brax composes positional.pipeline.step differently from tools/dump_golden.py; the "
                      "stage records are kept for inspection but flagged, compare_golden.py then only uses the "
                      "end-of-substep state")
        except Exception as e:  # a different Brax version: keep the end-of-substep record
            print(f"NOTE: stage-by-stage dump failed on this Brax ({type(e).__name__}: {e}); only the end-of-substep "
                  "state is recorded")
            rec["stage_composition_matches_pipeline_step"] = np.asarray(False)
    rec["substep_action"] = np.asarray(action)
    rec["substep_in_x_pos"], rec["substep_in_x_rot"] = np.asarray(st.x_i.pos), np.asarray(st.x_i.rot)
    rec["substep_in_xd_vel"], rec["substep_in_xd_ang"] = np.asarray(st.xd_i.vel), np.asarray(st.xd_i.ang)
    rec["substep_out_x_pos"], rec["substep_out_x_rot"] = np.asarray(ref_next.x_i.pos), np.asarray(ref_next.x_i.rot)
    rec["substep_out_xd_vel"], rec["substep_out_xd_ang"] = np.asarray(ref_next.xd_i.vel), np.asarray(ref_next.xd_i.ang)
    # the compiled system: what DESIGN.md §9's "MJCF compile" guesses have to reproduce
    rec["sys_link_mass"] = np.asarray(sys_.link.inertia.mass)
    rec["sys_link_inertia"] = np.asarray(sys_.link.inertia.i)
    rec["sys_link_com_pos"] = np.asarray(sys_.link.inertia.transform.pos)
    rec["sys_link_com_rot"] = np.asarray(sys_.link.inertia.transform.rot)
    rec["sys_link_joint_pos"] = np.asarray(sys_.link.joint.pos)
    rec["sys_link_joint_rot"] = np.asarray(sys_.link.joint.rot)
    rec["sys_link_transform_pos"] = np.asarray(sys_.link.transform.pos)
    rec["sys_link_transform_rot"] = np.asarray(sys_.link.transform.rot)
    rec["sys_link_parents"] = np.asarray(sys_.link_parents)
    rec["sys_link_types"] = np.asarray([ord(c) for c in sys_.link_types])
    rec["sys_actuator_gear"] = np.asarray(sys_.actuator.gear)
    rec["sys_actuator_ctrl_range"] = np.asarray(sys_.actuator.ctrl_range)
    rec["sys_dt"] = np.asarray(sys_.dt)
    for name in ("joint_scale_pos", "joint_scale_ang", "collide_scale", "vel_damping", "ang_damping", "elasticity"):
        if hasattr(sys_, name):
            rec[f"sys_{name}"] = np.asarray(getattr(sys_, name))
    for name in ("stiffness", "damping", "limit"):
        if hasattr(sys_.dof, name):
            rec[f"sys_dof_{name}"] = np.asarray(getattr(sys_.dof, name))
    # (the compiled system is recorded once, with the unprefixed substep)
    out.update({(k if k.startswith("sys_") else prefix + k): v for k, v in rec.items() if not (prefix and k.startswith("sys_"))})


BOUNCE_XML = """<mujoco><compiler angle="degree" inertiafromgeom="true"/>
<default><geom conaffinity="0" contype="0"/></default><option timestep="0.002"/>
<custom><numeric name="elasticity" data="0.5"/></custom>
<worldbody><geom conaffinity="1" type="plane" size="5 5 1"/>
<body name="ball" pos="0 0 0.6"><joint type="free" name="root"/>
<geom type="sphere" size="0.1" contype="1"/></body></worldbody></mujoco>"""


def dump_bounce(out):
    """(C) restitution: a ball with elasticity 0.5 dropped 0.5 m onto the floor through Brax's positional pipeline — the
    only record with elasticity != 0 (every env of the reference has 0).  z(t) and vz(t) of 600 substeps: a rebound at
    about half the impact speed means the velocity solve keeps max(-e vn, 0) (this repo's default since round 3); no
    rebound means the literal min(..) of eq. 34 (MBD_FLAG_RESTITUTION_MIN).  tools/compare_golden.py reads the ratio.
    Best effort: a Brax whose loader rejects the file leaves a note instead of a record."""
    try:
        import jax
        from jax import numpy as jnp
        from brax.io import mjcf
        from brax.positional import pipeline
        sys_ = mjcf.loads(BOUNCE_XML)
        st = jax.jit(pipeline.init)(sys_, sys_.init_q, jnp.zeros(sys_.qd_size()))
        step = jax.jit(pipeline.step)
        zs, vz = [], []
        for _ in range(600):
            st = step(sys_, st, jnp.zeros(sys_.act_size()))
            zs.append(float(st.x.pos[0, 2])); vz.append(float(st.xd.vel[0, 2]))
        out["bounce_z"], out["bounce_vz"], out["bounce_elasticity"] = np.asarray(zs), np.asarray(vz), np.asarray(0.5)
    except Exception as e:  # noqa: BLE001
        print(f"NOTE: the restitution record was not produced on this Brax ({type(e).__name__}: {e})")


def dump(ref, env_name, N, H, steps, out_dir=".", records="ABC"):
    """Run the reference under the jax / brax that are importable HERE and write golden_<env>_N<N>_H<H>.npz into out_dir;
    returns its path.  records: which of (A) the reverse_once steps, (B) the stage-by-stage substeps, (C) the bounce to
    produce — (B) and (C) reach into Brax's internals and leave a note instead of a record when the installed Brax differs.
    Importable: bench.py's reference leg calls it when `import jax, brax` succeeds (its `parity_jax` object), and
    tests/test_dump_golden.py executes record (A) under tools/make_ref_golden.py's numpy stand-in so that this code path
    has run before it ever meets a real Brax."""
    import os
    sys.dont_write_bytecode = True  # importing the reference must not leave __pycache__ in its tree
    if ref not in sys.path:
        sys.path.insert(0, ref)
    import functools
    import brax
    import jax
    from jax import numpy as jnp
    import mbd

    env = mbd.envs.get_env(env_name)
    Nu = env.action_size
    step_env = jax.jit(env.step)
    rollout_us = jax.jit(functools.partial(mbd.utils.rollout_us, step_env))
    rng = jax.random.PRNGKey(0)
    rng, rng_reset = jax.random.split(rng)
    state_init = jax.jit(env.reset)(rng_reset)
    Nd = 100
    betas = jnp.linspace(1e-4, 1e-2, Nd)
    alphas = 1.0 - betas
    alphas_bar = jnp.cumprod(alphas)
    sigmas = jnp.sqrt(1 - alphas_bar)
    rng_exp, rng = jax.random.split(rng)
    ps = state_init.pipeline_state
    out = dict(jax_version=str(getattr(jax, "__version__", "?")), brax_version=str(getattr(brax, "__version__", "?")),
               threefry_partitionable=bool(getattr(getattr(jax, "config", None), "jax_threefry_partitionable", False)),
               alphas_bar=np.asarray(alphas_bar), sigmas=np.asarray(sigmas),
               q0=np.asarray(ps.q), qd0=np.asarray(ps.qd), x0_pos=np.asarray(ps.x.pos), x0_rot=np.asarray(ps.x.rot))
    if hasattr(env, "sys") and "B" in records:  # (B) one substep, stage by stage; fixed action 0.3 on every actuator
        try:
            dump_substep_stages(env, state_init, jnp.full((Nu,), 0.3), out)
            # (B') the same from a SETTLED state — 12 control steps under that action: the feet are on the floor, links with
            # two colliders (hopper / walker2d / halfcheetah feet, ant legs) have both in contact
            st = state_init
            for _ in range(12):
                st = step_env(st, jnp.full((Nu,), 0.3))
            dump_substep_stages(env, st, jnp.full((Nu,), 0.3), out, prefix="contact_")
        except Exception as e:  # noqa: BLE001 — a Brax without these internals: the file then holds (A) only
            print(f"NOTE: the stage-by-stage records were not produced on this Brax ({type(e).__name__}: {e})")
    if hasattr(env, "sys") and "C" in records:
        dump_bounce(out)  # (C) the one record with elasticity != 0: settles the restitution clamp's sign convention
    Ybar = jnp.zeros([H, Nu])
    r = rng_exp
    for k, i in enumerate(range(Nd - 1, Nd - 1 - steps, -1)):
        if "A" not in records:
            break
        r, ks = jax.random.split(r)
        eps = jax.random.normal(ks, (N, H, Nu))
        Y0s = jnp.clip(eps * sigmas[i] + Ybar, -1.0, 1.0)
        rewss, qs = jax.vmap(rollout_us, in_axes=(None, 0))(state_init, Y0s)
        rews = rewss.mean(axis=-1)
        std = jnp.where(rews.std() < 1e-4, 1.0, rews.std())
        w = jax.nn.softmax((rews - rews.mean()) / std / 0.1)
        Ybar = jnp.einsum("n,nij->ij", w, Y0s)
        out.update({f"key_{k}": np.asarray(ks), f"eps_{k}": np.asarray(eps), f"Y0s_{k}": np.asarray(Y0s),
                    f"rewss_{k}": np.asarray(rewss), f"weights_{k}": np.asarray(w), f"Ybar_{k}": np.asarray(Ybar),
                    f"xpos_{k}": np.asarray(qs.x.pos)})
    path = os.path.join(out_dir, f"golden_{env_name}_N{N}_H{H}.npz")
    np.savez_compressed(path, **out)
    return path


def main():
    ref, env_name, N, H, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    print("wrote", dump(ref, env_name, N, H, steps))


if __name__ == "__main__":
    main()
