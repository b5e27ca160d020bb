"""Physics oracle (oracle/mbd_oracle_physics.c): invariants that pin the restated positional PBD step
without Brax, plus the chaos measurement that motivates the bit-exact contract."""
import math
import os
import tempfile

import numpy as np
import pytest

from conftest import load_model

FREE_BALL = """<mujoco><compiler angle="degree" inertiafromgeom="true"/>
<default><geom conaffinity="0" contype="0"/></default><option timestep="0.005"/>
<worldbody><geom conaffinity="1" type="plane" size="5 5 1"/>
<body name="ball" pos="0 0 {z}"><joint type="free" name="root"/>
<geom type="sphere" size="0.1" contype="{ct}"/></body></worldbody></mujoco>"""

PENDULUM = """<mujoco><compiler angle="degree" inertiafromgeom="true"/>
<default><geom conaffinity="0" contype="0"/><joint damping="0" limited="false"/></default>
<option timestep="0.002"/>
<custom><numeric name="joint_scale_pos" data="1.0"/><numeric name="joint_scale_ang" data="1.0"/>
<numeric name="spring_inertia_scale" data="0"/></custom>
<worldbody><body name="base" pos="0 0 2"><joint type="free"/><geom type="sphere" size="0.3" density="100000"/>
<body name="arm" pos="0 0 0"><joint type="hinge" axis="0 1 0" pos="0 0 0" name="h" {lim}/>
<geom type="capsule" fromto="0 0 0 0 0 -0.5" size="0.05"/></body></body></worldbody>
<actuator><motor joint="h" gear="1" ctrllimited="false"/></actuator></mujoco>"""


def _compile(xml, **kw):
    from mbd_hip import mjcf
    with tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False) as f:
        f.write(xml)
    kw.setdefault("warn_unstable", False)   # (some test models are violent on purpose; tests/test_random_models.py covers the report)
    try:
        return mjcf.load(f.name, **kw)
    finally:
        os.unlink(f.name)


def test_free_fall_matches_semi_implicit_euler(orc):
    m = _compile(FREE_BALL.format(z=5.0, ct=0))
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(6, np.float32))
    dt, g = 0.005, -9.81
    z, vz = 5.0, 0.0
    for _ in range(100):
        st = orc.substep(ms, st, np.zeros(0, np.float32))
        vz += g * dt
        z += vz * dt
    # PBD re-derives v = (p - p_prev)/dt every substep: f32 round-off of p (ulp 5e-7 at z~5) times 1/dt
    # random-walks the velocity, so the match is to ~1e-3, not to round-off
    assert abs(st[0, 2] - z) < 3e-3 and abs(st[0, 9] - vz) < 2e-2
    assert np.allclose(st[0, 3:7], [1, 0, 0, 0]) and np.allclose(st[0, :2], 0)


def test_sphere_rests_on_the_plane(orc):
    m = _compile(FREE_BALL.format(z=0.3, ct=1))
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(6, np.float32))
    for _ in range(400):
        st = orc.substep(ms, st, np.zeros(0, np.float32))
    assert abs(st[0, 2] - 0.1) < 2e-3          # centre at z = radius
    assert np.abs(st[0, 7:]).max() < 0.06       # at rest up to one substep of gravity


def test_hinge_keeps_its_anchor_and_axis(orc):
    """A hinged rod on a heavy free base, both in free fall (no gravity in the falling frame): the rod
    keeps swinging about the hinge axis only, and its anchor stays on the base."""
    m = _compile(PENDULUM.format(lim=""))
    ms = m.to_struct()
    qd = np.zeros(7, np.float32)
    qd[6] = 2.0  # hinge angular velocity
    st = orc.forward(ms, m.init_q, qd)
    com = np.asarray(m.fields["com"][1], float)
    for _ in range(500):
        st = orc.substep(ms, st, np.zeros(1, np.float32))
        anchor_c = st[1, :3] - _rot(st[1, 3:7], com)      # hinge anchor = link-frame origin of the rod
        assert np.linalg.norm(anchor_c - st[0, :3]) < 2e-3
        ang = orc.joint_angles(ms, st)[1]
        assert abs(ang[1]) < 2e-3 and abs(ang[2]) < 2e-3   # no rotation out of the hinge axis
    assert orc.joint_angles(ms, st)[1, 0] > 1.0            # ~2 rad/s for 1 s, damped a little by the solver


def _rot(q, v):
    w, x, y, z = [float(t) for t in q]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, float)


def test_joint_limit_clamps(orc):
    m = _compile(PENDULUM.format(lim='limited="true" range="-20 20"'))
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(7, np.float32))
    worst = 0.0
    for _ in range(600):
        st = orc.substep(ms, st, np.array([3.0], np.float32))  # constant torque pushes into the limit
        worst = max(worst, abs(orc.joint_angles(ms, st)[1, 0]))
    assert worst < math.radians(20) + 0.08      # soft (scaled) projection keeps it near the limit
    assert worst > math.radians(15)              # the torque does drive it to the limit


def test_humanoid_zero_action_feet_rest_on_floor(orc):
    m = load_model("humanoidrun")
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    pos0 = orc.link_positions(ms, st)
    assert np.allclose(pos0[0], [0, 0, 1.4]) and abs(pos0[4, 2] - 0.532) < 1e-3
    zero = np.zeros(17, np.float32)
    for _ in range(12):
        st, r = orc.env_step(ms, st, zero)
    pos = orc.link_positions(ms, st)
    # shin origin is 0.35 above the foot sphere centre (radius 0.075): rests at 0.425
    assert abs(pos[4, 2] - 0.425) < 0.01 and abs(pos[6, 2] - 0.425) < 0.01
    assert np.isfinite(st).all() and abs(pos[0, 1]) < 1e-3       # left/right symmetric fall
    # reward = x - clip(|z - 1.3|) - 0.1 |y| of the torso origin (humanoidrun.py:46-51)
    assert abs(r - (pos[0, 0] - min(abs(pos[0, 2] - 1.3), 1.0) - 0.1 * abs(pos[0, 1]))) < 1e-6


@pytest.mark.parametrize("name", ["humanoidrun", "humanoidtrack", "hopper", "halfcheetah", "walker2d",
                                  "humanoidstandup", "cartpole", "ant"])
def test_rollouts_stay_finite_and_actions_saturate(orc, name):
    m = load_model(name)
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    g = np.random.default_rng(0)
    us = np.clip(g.normal(size=(6, 50, m.act_size())), -1, 1).astype(np.float32)
    rew, fin = orc.rollout(ms, st, us, want_final=True)
    assert np.isfinite(rew).all() and np.isfinite(fin).all()
    assert np.allclose(np.linalg.norm(fin[:, :, 3:7], axis=-1), 1.0, atol=1e-5)
    if name.startswith("humanoid"):
        # ctrlrange +-0.4 (humanoidrun.xml:6): actions beyond it saturate inside the actuator model
        big = (np.sign(us) * np.maximum(np.abs(us), 0.4)).astype(np.float32)
        sat = np.clip(big, -0.4, 0.4)
        assert np.array_equal(orc.rollout(ms, st, big), orc.rollout(ms, st, sat))


def test_humanoidtrack_reward_is_lagged(orc):
    """humanoidtrack.py:78 passes the INCOMING state to _get_reward: reward[0] is the reset state's."""
    m = load_model("humanoidtrack")
    ms = m.to_struct()
    st = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
    g = np.random.default_rng(1)
    us = g.uniform(-1, 1, (4, 5, 17)).astype(np.float32)
    rew = orc.rollout(ms, st, us)
    # torso at z=1.4, v=0: 1 - |0 - 1.6| - |1.4 - 1.3| - 0 = -0.7
    assert np.allclose(rew[:, 0], -0.7, atol=1e-6) and len(set(rew[:, 0].tolist())) == 1
    assert len(set(rew[:, 2].tolist())) > 1


def test_chaos_amplification(orc, orc64):
    """Why parity is bit-exact and not 'within 1e-5': the same code in f32 and f64 drifts by more than
    1e-5 (relative) in reward over a 50-step contact-rich rollout, i.e. no two correct-but-differently-
    rounded implementations can be compared through a tolerance at full horizon."""
    m = load_model("humanoidrun")
    ms = m.to_struct()
    g = np.random.default_rng(0)
    st = orc.forward(ms, m.init_q + g.uniform(-0.01, 0.01, 24).astype(np.float32),
                     g.uniform(-0.01, 0.01, 23).astype(np.float32))
    us = np.clip(g.normal(size=(48, 50, 17)) * 0.6, -1, 1).astype(np.float32)
    r32, r64 = orc.rollout(ms, st, us), orc64.rollout(ms, st, us)
    err = np.abs(r32 - r64)
    assert err[:, 0].max() < 5e-6            # one control step: round-off only
    assert err[:, -1].max() > 2e-5           # 350 substeps later: amplified beyond the 1e-5 the north star asks for


def test_cartpole_weld_limits_and_reward(orc):
    """The reference's own 2-link model (mbd/assets/cartpole.xml, mbd/envs/cartpole.py): the slide-only cart keeps
    its orientation (weld alignment), respects the +-1 m rail limit, and the reward is cos(q1) - |qd0|."""
    m = load_model("cartpole")
    ms = m.to_struct()
    assert m.fields["n_rot"].tolist() == [0, 1] and m.fields["n_slide"].tolist() == [1, 0]
    assert np.allclose(m.init_q, [0.0, np.pi]) and abs(m.fields["dt"] - 0.005) < 1e-9 and m.fields["n_frames"] == 4
    st = orc.forward(ms, m.init_q, np.zeros(2, np.float32))
    st, r = orc.env_step(ms, st, np.zeros(1, np.float32))
    assert abs(r - (-1.0)) < 1e-3                      # pole hanging down, cart at rest
    push = np.array([3.0], np.float32)                 # ctrlrange +-3, gear 100
    xs = []
    for _ in range(150):
        st, r = orc.env_step(ms, st, push)
        xs.append(st[0, 0])
        ang = orc.joint_angles(ms, st)[1, 0]
        assert abs(r - (np.cos(ang) - abs(st[0, 7]))) < 2e-3
        assert np.allclose(st[0, 3:7], [1, 0, 0, 0], atol=2e-3)   # the cart does not rotate
        assert abs(st[0, 1]) < 1e-3 and abs(st[0, 2]) < 8e-3      # and stays on its rail (soft PBD constraint)
    assert max(xs) < 1.0 + 0.15 and max(xs) > 0.9                # pressed against the soft +1 m limit


# ---- the planar restatement (oracle/mbd_oracle_planar.h; MBD_FLAG_PLANAR) ---------------------------------------
@pytest.mark.parametrize("name", ["hopper", "walker2d", "halfcheetah", "cartpole"])
def test_planar_restatement_agrees_with_the_3d_one_per_control_step(orc, name):
    """The planar models are simulated on their in-plane coordinates only (a specification of its own: the 3-D float
    arithmetic is not exactly planar).  Teacher-forced from the SAME state, one control step (16-20 substeps) of the
    planar restatement and of the general 3-D one must agree to float round-off — rewards to 2e-5, in-plane poses and
    velocities to 1e-3 — while a free-running 3-D rollout leaks out of the plane and the planar one cannot."""
    from oracle.planner import OracleEnv
    m = load_model(name)
    ms_pl, ms_3d = m.to_struct(), m.to_struct()
    assert ms_pl.flags & 2, "the MJCF compiler marks this model planar"
    ms_3d.flags = ms_3d.flags & ~2
    env = OracleEnv(orc, name, ms_pl, init_q=m.init_q)
    s = env.reset(orc.prng_key(3), 1)
    inpl = [0, 2, 3, 5, 7, 9, 11]
    outpl = [1, 4, 6, 8, 10, 12]
    g = np.random.default_rng(1)
    s3 = s.copy()
    worst_state, worst_rew, leak = 0.0, 0.0, 0.0
    for t in range(40):
        a = np.clip(g.normal(size=m.act_size()) * 0.7, -1, 1).astype(np.float32)
        n3, r3 = orc.env_step(ms_3d, s, a)
        npl, rpl = orc.env_step(ms_pl, s, a)
        worst_state = max(worst_state, float(np.abs(n3[:, inpl] - npl[:, inpl]).max()))
        worst_rew = max(worst_rew, abs(float(r3) - float(rpl)))
        assert np.all(npl[:, outpl] == 0.0), "the planar restatement never leaves the plane"
        s3, _ = orc.env_step(ms_3d, s3, a)
        leak = max(leak, float(np.abs(s3[:, outpl]).max()))
        s = npl
    assert worst_state < 1e-3 and worst_rew < 2e-5, (worst_state, worst_rew)
    assert leak > 0.0, "the 3-D arithmetic does leak out of the plane (which is why planar is a spec of its own)"


def test_planar_classification():
    """mbd_hip.mjcf.is_planar: the four planar models qualify, the humanoids / ant (free root, 3-D joints) do not."""
    from mbd_hip import mjcf
    for name, want in (("hopper", True), ("walker2d", True), ("halfcheetah", True), ("cartpole", True),
                       ("humanoidrun", False), ("ant", False), ("humanoidstandup", False), ("humanoidtrack", False)):
        m = load_model(name)
        assert mjcf.is_planar(m.fields) == want, name
        assert bool(int(m.fields["flags"]) & 2) == want, name


_PENDULUM_XML = """<mujoco model="pendulum">
  <compiler inertiafromgeom="true"/>
  <default><joint armature="0" damping="0" limited="false"/><geom contype="0"/></default>
  <option gravity="0 0 -9.81" timestep="0.002"/>
  <custom><numeric data="1" name="spring_inertia_scale"/></custom>
  <worldbody>
    <body name="pole" pos="0 0 2">
      <joint axis="0 1 0" name="hinge" pos="0 0 0" type="hinge"/>
      <geom fromto="0 0 0 0 0 -1.0" name="rod" size="0.02 0.5" type="capsule"/>
    </body>
  </worldbody>
  <actuator><motor gear="1" joint="hinge" name="m"/></actuator>
</mujoco>"""


@pytest.mark.parametrize("planar", [True, False])
def test_physical_pendulum_period_and_energy(orc, tmp_path, planar):
    """An analytic check of the restated solver that owes nothing to the kernels: a rod on a friction-less hinge is a
    physical pendulum with period 2 pi sqrt(I_pivot / (m g d)).  Released at 0.2 rad, the simulated period must match
    the closed form (finite-amplitude correction included) to 1 %, and the swing amplitude must not grow.  Run on the
    planar restatement (the compiler marks the model planar) and on the general 3-D one."""
    from mbd_hip import mjcf
    path = tmp_path / "pendulum.xml"
    path.write_text(_PENDULUM_XML)
    m = mjcf.load(str(path), env_name="hopper", n_frames=1, planar=planar)
    assert bool(int(m.fields["flags"]) & 2) == planar
    ms = m.to_struct()
    mass = 1.0 / float(m.fields["inv_mass"][0])
    I_com = 1.0 / float(m.fields["inv_inertia"][0][1])                 # about y, through the COM
    d = abs(float(m.fields["com"][0][2]))                              # pivot -> COM
    T_small = 2 * np.pi * np.sqrt((I_com + mass * d * d) / (mass * 9.81 * d))
    th0 = 0.2
    T_ref = T_small * (1 + th0 ** 2 / 16)                              # first finite-amplitude term
    s = orc.forward(ms, np.array([th0], np.float32), np.zeros(1, np.float32))
    a = np.zeros(1, np.float32)
    dt = float(m.fields["dt"])
    ang, t = [], []
    for k in range(int(3.2 * T_ref / dt)):
        s, _ = orc.env_step(ms, s, a)
        ang.append(float(orc.joint_angles(ms, s)[0, 0]))
        t.append((k + 1) * dt)
    ang, t = np.array(ang), np.array(t)
    # downward zero crossings (theta: + -> -), linearly interpolated
    idx = np.where((ang[:-1] > 0) & (ang[1:] <= 0))[0]
    cross = t[idx] + dt * ang[idx] / (ang[idx] - ang[idx + 1])
    assert len(cross) >= 3
    period = float(np.mean(np.diff(cross)))
    assert abs(period - T_ref) / T_ref < 0.01, (period, T_ref)
    first, last = np.abs(ang[: int(T_ref / dt)]).max(), np.abs(ang[-int(T_ref / dt):]).max()
    assert last <= first * 1.001 and last > 0.5 * first, (first, last)  # no energy gain; position-based damping is mild


def test_dependency_depth_of_a_substep(orc):
    """oracle/count_ops.cc carries, beside every value, the length of the dependency chain that produced it (round 5: the latency
    floor of a substep, measured).  The counting build computes the same values as the plain one; the depth grows by the same
    amount every substep (the recurrence's critical path), is the same for every link (the tree couples them within a substep),
    and is several times smaller than the operations per link — the width a wider layout could harvest at best."""
    from conftest import load_model
    from oracle import oracle as orc_mod
    for name, lo, hi in (("humanoidrun", 150, 220), ("hopper", 90, 130)):
        m = load_model(name)
        ms = m.to_struct()
        s = orc.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
        a = np.full(m.act_size(), 0.3, np.float32)
        for _ in range(12):
            s, _ = orc.env_step(ms, s, a)
        d = orc_mod.depth_substeps(ms, s, a, 10).astype(np.int64)
        growth = np.diff(d[:, -1])
        assert (growth[4:] == growth[-1]).all() and lo <= growth[-1] <= hi, (name, growth)
        assert (d[-1, :-1] - d[-2, :-1] == growth[-1]).all()
        counts, out = orc_mod.count_substep(ms, s, a)
        assert np.array_equal(out.reshape(-1), orc.substep(ms, s, a).reshape(-1))   # (counting does not change a value)
        ops = counts["add"] + counts["mul"] + counts["fma"] + counts["div"] + counts["sqrt"] + counts["cmp"]
        assert ops / m.n_links > 2.0 * growth[-1]
