"""Oracle pinning: update rules of mbd/planners/path_integral.py:33-52,122-125 against numpy (float64)."""
import numpy as np


def _inputs(N=64, H=5, Nu=3, seed=0):
    g = np.random.default_rng(seed)
    rews = g.normal(size=N).astype(np.float32)
    mu = (g.normal(size=(H, Nu)) * 0.1).astype(np.float32)
    Y0s = np.clip(mu + g.normal(size=(N, H, Nu)) * 0.7, -1, 1).astype(np.float32)
    return rews, Y0s, mu


def _weights(rews, temp):
    r = rews.astype(np.float64)
    l = (r - r.mean()) / r.std() / temp
    w = np.exp(l - l.max())
    return w / w.sum()


def test_mppi_is_the_weighted_mean_and_keeps_sigma(orc):
    rews, Y0s, mu = _inputs()
    out, sigma, w, m = orc.pi_update(1, rews, Y0s, mu, 0.8, 0.1)
    wr = _weights(rews, 0.1)
    assert np.allclose(w, wr, rtol=3e-5, atol=1e-10) and abs(m - rews.astype(np.float64).mean()) < 1e-6
    assert np.abs(out - np.einsum("n,nij->ij", wr, Y0s.astype(np.float64))).max() < 2e-6
    assert sigma == np.float32(0.8)


def test_cma_es_sigma_update(orc):
    rews, Y0s, mu = _inputs(seed=1)
    out, sigma, w, _ = orc.pi_update(2, rews, Y0s, mu, 0.5, 0.2)
    wr = _weights(rews, 0.2)
    err2 = (Y0s.astype(np.float64) - mu) ** 2
    ref = np.sqrt(np.einsum("n,nij->ij", wr, err2)).mean() * 0.5
    assert abs(sigma - max(ref, 1e-3)) < 1e-6
    assert np.abs(out - np.einsum("n,nij->ij", wr, Y0s.astype(np.float64))).max() < 2e-6
    # the floor (:44)
    _, s2, _, _ = orc.pi_update(2, rews, np.repeat(mu[None], 64, 0), mu, 0.5, 0.2)
    assert s2 == np.float32(1e-3)


def test_cem_takes_the_ten_best(orc):
    rews, Y0s, mu = _inputs(seed=2)
    out, sigma, w, _ = orc.pi_update(3, rews, Y0s, mu, 1.0, 0.1)
    idx = np.argsort(w)[::-1][:10]
    assert np.abs(out - Y0s[idx].astype(np.float64).mean(0)).max() < 1e-6 and sigma == 1.0
    # ties: argsort()[::-1] prefers the HIGHER index among equal weights
    rews[:] = 0.0
    rews[5] = 1.0
    out, _, w, _ = orc.pi_update(3, rews, Y0s, mu, 1.0, 0.1)
    idx = [5] + list(range(63, 54, -1))
    assert np.abs(out - Y0s[idx].astype(np.float64).mean(0)).max() < 1e-6


def test_no_std_guard(orc):
    """path_integral.py:123 divides by rews.std() unguarded: constant rewards give NaN weights (as in JAX)."""
    rews, Y0s, mu = _inputs(seed=3)
    rews[:] = 0.3
    out, _, w, _ = orc.pi_update(1, rews, Y0s, mu, 1.0, 0.1)
    assert np.isnan(w).all() and np.isnan(out).all()
