"""The specification switches (include/mbd_hip.h mbd_model_flags; round-3 verdict item 3): the code-level
guesses about Brax's positional pipeline as flag bits of the model, honoured by the checker (here) and the kernels
(tests/test_gpu_parity.py) alike.  The default word (model.DEFAULT_SPEC = contact_avg since round 6) is the specification the shipped tuned kernels compile in (rounds 1-4's, with stage (6) Jacobi per
link since round 5: the committed self_* goldens hold it bit for bit); every bit selects a named alternative whose effect is pinned here to what it is supposed to be, so that
tools/compare_golden.py --search can tell the alternatives apart when a real golden arrives."""

import numpy as np
import pytest

from conftest import load_model
from mbd_hip.model import SPEC_FLAGS, spec_bits, spec_names
from test_oracle_invariants import BALL, BALL3, SLED, _qaxis, _qmul
from test_oracle_physics import _compile, _rot


def _roll(orc, m, steps=40, seed=0, scale=0.5):
    ms = m.to_struct()
    g = np.random.default_rng(seed)
    st = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    out = []
    for _ in range(steps):
        a = np.clip(g.normal(size=m.act_size()) * scale, -1, 1).astype(np.float32)
        st, r = orc.env_step(ms, st, a)
        out.append(st.copy())
    return np.stack(out)


def test_flag_words():
    assert spec_bits("contact_avg", "contact6_gauss_seidel") == 12 and spec_names(16 | 128) == ["friction_vel_bound", "gyroscopic"]
    assert sorted(SPEC_FLAGS.values()) == [4, 8, 16, 32, 64, 128]


@pytest.mark.parametrize("name", ["humanoidrun", "hopper", "walker2d", "humanoidstandup"])
def test_contact_switches_only_touch_links_with_several_contacts(orc, name):
    """contact_avg / contact6_gauss_seidel (stage (6) one contact after the other instead of the default, all from the same
    velocities) act on links with two or more ACTIVE contacts: the humanoids' feet carry one sphere
    each — not a bit changes — while hopper's and walker2d's feet (two spheres: the planar restatement) and
    humanoidstandup's torso (five: the 3-D one) do once they stand / lie on both."""
    m = load_model(name)
    assert int(m.fields["flags"]) & 252 == spec_bits("contact_avg")   # (the default word since round 6: model.DEFAULT_SPEC)
    base = _roll(orc, m.with_spec(0), 100, scale=0.1)  # (the summed Jacobi form; gentle actions: the models settle onto their feet / their back)
    for bits in (spec_bits("contact_avg"), spec_bits("contact6_gauss_seidel"), spec_bits("contact_avg", "contact6_gauss_seidel")):
        got = _roll(orc, m.with_spec(bits), 100, scale=0.1)
        assert np.isfinite(got).all()
        if name == "humanoidrun":
            assert np.array_equal(got, base), spec_names(bits)
        else:
            assert not np.array_equal(got, base), spec_names(bits)


def test_contact_avg_halves_the_correction_of_two_equal_contacts(orc):
    """A sled on four runners dropped flat: with the sum every runner's full correction is added (the position solve of a
    link with n equal contacts overshoots n-fold, damped by collide_scale); with the average the link moves by ONE
    contact's correction.  After the first substep in contact the summed correction is 4x the averaged one."""
    m = _compile(SLED.format(gx=0.0, gz=-9.81, mu=1.0))
    ms0, ms1 = m.with_spec(0).to_struct(), m.with_spec(spec_bits("contact_avg")).to_struct()
    st = orc.forward(ms0, m.init_q, np.zeros(6, np.float32))
    st[0, 2] -= 0.02   # all four spheres 2 cm into the floor, at rest
    g = -9.81 * 0.002 * 0.002  # (the fall of one substep)
    z0 = float(st[0, 2])
    z_sum = float(orc.substep(ms0, st, np.zeros(0, np.float32))[0, 2]) - z0 - g
    z_avg = float(orc.substep(ms1, st, np.zeros(0, np.float32))[0, 2]) - z0 - g
    assert z_avg > 1e-3 and abs(z_sum / z_avg - 4.0) < 0.02, (z_sum, z_avg)


def test_jacobi_velocity_stage_sees_one_velocity(orc):
    """The default (Jacobi): all runners of the sled compute their friction impulse from the same slip velocity, so a sled
    sliding along x is braked by 4x one runner's impulse, while contact6_gauss_seidel gives every later runner a slower sled
    and a smaller impulse (the bound binds first): the Jacobi sled loses at least as much speed in the step."""
    m = _compile(SLED.format(gx=0.0, gz=-9.81, mu=1.0))
    st = None
    lost = {}
    for tag, bits in (("jacobi", 0), ("gs", spec_bits("contact6_gauss_seidel"))):
        ms = m.with_spec(bits).to_struct()
        st = orc.forward(ms, m.init_q, np.zeros(6, np.float32))
        for _ in range(300):  # settle
            st = orc.substep(ms, st, np.zeros(0, np.float32))
        st[0, 7] = 1.0        # shove it along x
        nxt = orc.substep(ms, st, np.zeros(0, np.float32))
        lost[tag] = 1.0 - float(nxt[0, 7])
    assert lost["gs"] > 1e-3 and lost["jacobi"] >= lost["gs"] * 0.999, lost


def test_friction_velocity_bound_is_the_literal_eq30(orc):
    """friction_vel_bound: |dv_t| = min(mu lambda_n / h, |v_t|) — the bound is a VELOCITY.  Against the default (an
    impulse, times the tangential inverse mass w_t) the deceleration of a sliding sled changes by the factor 1 / w_t
    ~ its mass: the flagged sled (60 kg-ish: w_t << 1) brakes far harder, and stops."""
    m = _compile(SLED.format(gx=0.0, gz=-9.81, mu=0.5))
    dec = {}
    for tag, bits in (("impulse", 0), ("velocity", spec_bits("friction_vel_bound"))):
        ms = m.with_spec(bits).to_struct()
        st = orc.forward(ms, m.init_q, np.zeros(6, np.float32))
        for _ in range(300):
            st = orc.substep(ms, st, np.zeros(0, np.float32))
        st[0, 7] = 2.0
        v = []
        for _ in range(50):
            st = orc.substep(ms, st, np.zeros(0, np.float32))
            v.append(float(st[0, 7]))
        dec[tag] = (2.0 - v[-1]) / (50 * 0.002)
    assert 0.8 * 0.5 * 9.81 < dec["impulse"] < 1.4 * 0.5 * 9.81, dec     # Coulomb: mu g
    assert dec["velocity"] > 3.0 * dec["impulse"], dec


@pytest.mark.parametrize("e", [0.0, 0.5])
def test_restitution_min_is_the_literal_clamp(orc, e):
    """restitution_min: min(-e vn_prev, 0) — with the floor's +z normal the term is never positive, the ball does not
    come back up whatever e says (the form of rounds 1-2); e = 0: not a bit of difference."""
    m = _compile(BALL.format(el=e))
    outs = []
    for bits in (0, spec_bits("restitution_min")):
        ms = m.with_spec(bits).to_struct()
        st = orc.forward(ms, m.init_q, np.zeros(6, np.float32))
        zs = []
        for _ in range(600):
            st = orc.substep(ms, st, np.zeros(0, np.float32))
            zs.append(float(st[0, 2]))
        outs.append(np.array(zs))
    hit = int(np.argmin(outs[0][:400]))
    if e == 0.0:
        assert np.array_equal(outs[0], outs[1])
    else:
        assert outs[0][hit:].max() > 0.1 + 0.2 * e * e * 0.5 and outs[1][hit + 5:].max() < 0.1 + 2e-3


@pytest.mark.parametrize("axes", [("1 0 0", "0 1 0", "0 0 1"), ("1 0 0", "0 0 1", "0 1 0")])
def test_euler_extrinsic_composes_about_the_fixed_axes(orc, axes):
    """euler_extrinsic: the joint's rotation is R(a3, q3) R(a2, q2) R(a1, q1) — each hinge about its axis in the PARENT's
    joint frame — instead of MuJoCo's listed-order composition about the moving axes.  Forward kinematics must produce
    that orientation, the solver's angles must read q back, and the pose must be a fixed point of the position solver.
    Single-hinge models do not change by a bit."""
    m = _compile(BALL3.format(a1=axes[0], a2=axes[1], a3=axes[2])).with_spec(spec_bits("euler_extrinsic"))
    ms = m.to_struct()
    ang = np.array([0.4, -0.7, 0.9])
    q = np.array(m.init_q, np.float32)
    qb = _qaxis([1, 2, 3], 0.8)
    q[3:7], q[7:10] = qb, ang
    st = orc.forward(ms, q, np.zeros(m.qd_size(), np.float32))
    rel = np.array([1.0, 0, 0, 0])
    for a, th in zip(axes, ang):
        rel = _qmul(_qaxis([float(t) for t in a.split()], th), rel)   # later hinges on the LEFT: fixed axes
    want = _qmul(qb, rel)
    got = st[1, 3:7].astype(float)
    assert np.abs(got - want * np.sign(np.dot(got, want))).max() < 1e-6
    sign = np.asarray(m.fields["rot_sign"][1], float)
    assert np.abs(orc.joint_angles(ms, st)[1] * sign - ang).max() < 1e-5
    s2 = st.copy()
    for _ in range(50):
        s2 = orc.substep(ms, s2, np.zeros(0, np.float32))
    assert np.abs(s2 - st).max() < 1e-3 and np.abs(orc.joint_angles(ms, s2)[1] * sign - ang).max() < 1e-3
    hop = load_model("hopper")
    hop3 = hop.with_spec(0)
    hop3.fields["flags"] = int(hop3.fields["flags"]) & ~2   # (the 3-D restatement: the planar one has no Euler angles)
    ext = hop3.with_spec(spec_bits("euler_extrinsic"))
    assert np.array_equal(_roll(orc, hop3, 10), _roll(orc, ext, 10))


def test_euler_extrinsic_velocities_match_finite_differences(orc):
    """the joint velocities of the extrinsic convention (forward kinematics at reset): the angular velocity the state
    carries equals the finite difference of the orientations at q and q + qd dt."""
    m = _compile(BALL3.format(a1="1 0 0", a2="0 1 0", a3="0 0 1")).with_spec(spec_bits("euler_extrinsic"))
    ms = m.to_struct()
    q = np.array(m.init_q, np.float32)
    q[7:10] = [0.3, -0.5, 0.7]
    qd = np.zeros(m.qd_size(), np.float32)
    qd[6:9] = [0.9, -1.1, 0.6]
    h = 1e-3
    s0 = orc.forward(ms, q, qd).astype(float)
    q1 = q.copy(); q1[7:10] += h * qd[6:9]
    s1 = orc.forward(ms, q1, qd).astype(float)
    dq = _qmul(s1[1, 3:7], s0[1, 3:7] * np.array([1, -1, -1, -1]))
    w_fd = 2.0 * dq[1:] / h * np.sign(dq[0])
    assert np.abs(w_fd - s0[1, 10:13]).max() < 5e-3, (w_fd, s0[1, 10:13])


BRICK = """<mujoco><compiler angle="degree" inertiafromgeom="true"/>
<default><geom conaffinity="0" contype="0"/></default><option timestep="{dt}" gravity="0 0 0"/>
<custom><numeric name="spring_inertia_scale" data="0"/><numeric name="ang_damping" data="0"/><numeric name="vel_damping" data="0"/></custom>
<worldbody><body name="brick" pos="0 0 1"><joint type="free"/>
<geom type="capsule" fromto="-0.3 0 0 0.3 0 0" size="0.05"/><geom type="capsule" fromto="0 -0.15 0 0 0.15 0" size="0.05"/></body></worldbody></mujoco>"""


def test_gyroscopic_term_conserves_angular_momentum(orc):
    """gyroscopic: a torque-free cross of two capsules (three distinct principal moments) tumbling.  Without the term the world-frame
    omega is constant and the angular momentum R I R^T omega wanders with the body; with it L stays put (first-order
    integrator: to a few percent over a revolution at dt = 1 ms).  Isotropic tensors and planar models ignore the bit."""
    m = _compile(BRICK.format(dt=0.001))
    assert int(m.fields["iso_inertia"]) == 0
    ii = np.asarray(m.fields["inv_inertia"][0], float)
    I = np.linalg.inv(np.array([[ii[0], ii[3], ii[4]], [ii[3], ii[1], ii[5]], [ii[4], ii[5], ii[2]]]))
    drift = {}
    for tag, bits in (("off", 0), ("on", spec_bits("gyroscopic"))):
        ms = m.with_spec(bits).to_struct()
        qd = np.array([0, 0, 0, 3.0, 0.4, 5.0], np.float32)
        st = orc.forward(ms, m.init_q, qd)

        def Lw(s):
            R = np.array([_rot(s[0, 3:7], e) for e in np.eye(3)]).T
            return R @ I @ R.T @ s[0, 10:13].astype(float)
        L0 = Lw(st)
        worst = 0.0
        for _ in range(1500):
            st = orc.substep(ms, st, np.zeros(0, np.float32))
            worst = max(worst, np.linalg.norm(Lw(st) - L0) / np.linalg.norm(L0))
        drift[tag] = worst
    assert drift["off"] > 0.2 and drift["on"] < 0.06, drift
    hum = load_model("humanoidrun")
    assert np.array_equal(_roll(orc, hum, 5), _roll(orc, hum.with_spec(spec_bits("gyroscopic")), 5))
    hop = load_model("hopper")
    assert np.array_equal(_roll(orc, hop, 5), _roll(orc, hop.with_spec(spec_bits("gyroscopic")), 5))
