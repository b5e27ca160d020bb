"""Host-side kinematics.inverse (observations, mbd_model_observe of the C ABI — pure host arithmetic, no device):
q/qd recovered from the COM-frame state must reproduce the generalized coordinates the state was built from
(oracle forward kinematics)."""
import numpy as np
import pytest

from conftest import load_model


class _FakeEnv:
    """RigidBodyEnv without a device: only the host-side methods are exercised."""

    def __init__(self, name, lib):
        from mbd_hip.envs.base import RigidBodyEnv
        self.__class__ = type("HostOnly", (RigidBodyEnv,), {"__del__": lambda self: None})
        self.env_name = name
        self.sys = load_model(name)
        self._lib = lib
        self._struct = self.sys.to_struct()


@pytest.mark.parametrize("name", ["humanoidrun", "hopper", "halfcheetah", "walker2d", "ant", "cartpole"])
def test_generalized_coordinates_round_trip(orc, lib, name):
    env = _FakeEnv(name, lib)
    m = env.sys
    g = np.random.default_rng(0)
    q = m.init_q.copy()
    lo, hi = np.asarray(m.fields["rot_lo"]), np.asarray(m.fields["rot_hi"])
    for l in range(m.n_links):
        if m.fields["n_rot"][l] < 0:
            continue
        qi, ns = int(m.fields["q_idx"][l]), int(m.fields["n_slide"][l])
        for k in range(int(m.fields["n_rot"][l])):
            sg = float(m.fields["rot_sign"][l][k])
            a, b = sorted((sg * max(lo[l, k], -0.6), sg * min(hi[l, k], 0.6)))
            q[qi + ns + k] = g.uniform(a, b) * 0.5
        for k in range(ns):
            q[qi + k] += g.uniform(-0.2, 0.2)
    qd = g.uniform(-0.5, 0.5, m.qd_size()).astype(np.float32)
    st = orc.forward(m.to_struct(), q.astype(np.float32), qd)
    q2, qd2 = env.generalized(st)
    assert np.allclose(q2, q, atol=2e-5), np.abs(q2 - q).max()
    # velocities: exact for 1-dof joints, first-order for multi-dof gimbals
    one = [int(m.fields["qd_idx"][l]) + int(m.fields["n_slide"][l]) for l in range(m.n_links)
           if m.fields["n_rot"][l] == 1]
    assert np.allclose(qd2[one], qd[one], atol=2e-4)
    obs = env._get_obs(st)
    assert obs.shape == (env.observation_size,)


def test_observation_layouts(orc, lib):
    """hopper.py:49-55: q[1] is replaced by the torso height, qd clipped to +-10; brax half_cheetah / ant drop the
    root x (x, y); everything else is concat(q, qd) (humanoidrun.py:43-44)."""
    for name, skip in (("humanoidrun", 0), ("halfcheetah", 1), ("ant", 2)):
        env = _FakeEnv(name, lib)
        m = env.sys
        st = orc.forward(m.to_struct(), m.init_q, np.full(m.qd_size(), 0.25, np.float32))
        q, qd = env.generalized(st)
        assert np.array_equal(env._get_obs(st), np.concatenate([q[skip:], qd]))
    env = _FakeEnv("hopper", lib)
    m = env.sys
    st = orc.forward(m.to_struct(), m.init_q, np.full(m.qd_size(), 20.0, np.float32))
    q, qd = env.generalized(st)
    obs = env._get_obs(st)
    assert obs[1] == env.link_positions(st)[0, 2] and np.array_equal(np.delete(obs[:6], 1), np.delete(q, 1))
    assert np.abs(obs[6:]).max() <= 10.0 and np.abs(qd).max() > 10.0
