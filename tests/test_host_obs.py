"""Host-side kinematics.inverse (observations): q/qd recovered from the COM-frame state must reproduce the
generalized coordinates the state was built from (oracle forward kinematics)."""
import numpy as np
import pytest

from conftest import load_model


class _FakeEnv:
    """RigidBodyEnv without a device: only the host-side methods are exercised."""

    def __init__(self, name):
        from mbd_hip.envs.base import RigidBodyEnv
        self.__class__ = type("HostOnly", (RigidBodyEnv,), {"__del__": lambda self: None})
        self.env_name = name
        self.sys = load_model(name)


@pytest.mark.parametrize("name", ["humanoidrun", "hopper", "halfcheetah", "walker2d"])
def test_generalized_coordinates_round_trip(orc, name):
    env = _FakeEnv(name)
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
