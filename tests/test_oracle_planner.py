"""Oracle pinning: planner algebra of mbd_planner.py:84-135."""
import numpy as np


def test_schedule_known_answers(orc):
    # SURVEY.md §8(a) row A0, float32
    a, ab, s = orc.schedule(1e-4, 1e-2, 100)
    assert np.float32(ab[-1]) == np.float32(0.6024805) and np.float32(s[-1]) == np.float32(0.6304915)
    assert np.float32(s[0]) == np.float32(0.01000083)
    a, ab, s = orc.schedule(1e-4, 1e-2, 300)
    assert np.float32(ab[-1]) == np.float32(0.2186933) and np.float32(s[-1]) == np.float32(0.88391554)
    assert np.float32(s[0]) == np.float32(0.01000083)
    assert np.all(np.diff(ab) < 0) and np.allclose(a, 1 - np.linspace(1e-4, 1e-2, 300), atol=1e-7)


def _inputs(N=96, H=7, Nu=3, seed=0):
    g = np.random.default_rng(seed)
    rews = g.normal(size=N).astype(np.float32)
    Y0s = np.clip(g.normal(size=(N, H, Nu)) * 0.5, -1, 1).astype(np.float32)
    Ybar = (g.normal(size=(H, Nu)) * 0.1).astype(np.float32)
    return rews, Y0s, Ybar


def test_score_update_matches_numpy_float64(orc):
    rews, Y0s, Ybar = _inputs()
    a, ab, _ = orc.schedule(1e-4, 1e-2, 100)
    out, w, m = orc.score_update(rews, Y0s, Ybar, a[50], ab[50], ab[49], 0.1)
    r = rews.astype(np.float64)
    lp = (r - r.mean()) / r.std() / 0.1
    wr = np.exp(lp - lp.max())
    wr /= wr.sum()
    assert np.allclose(w, wr, rtol=2e-5, atol=1e-9) and abs(w.sum() - 1) < 1e-6
    assert abs(m - r.mean()) < 1e-6
    ref = np.einsum("n,nij->ij", wr, Y0s.astype(np.float64))
    assert np.abs(out - ref).max() < 2e-6


def test_g7_identity_literal_vs_weighted_mean(orc):
    """the literal score/Yim1/Ybar_im1 chain (:130-133) collapses to the weighted mean."""
    rews, Y0s, Ybar = _inputs(seed=1)
    a, ab, _ = orc.schedule(1e-4, 1e-2, 100)
    for i in (1, 37, 99):
        lit, _, _ = orc.score_update(rews, Y0s, Ybar, a[i], ab[i], ab[i - 1], 0.1, literal=True)
        idn, _, _ = orc.score_update(rews, Y0s, Ybar, a[i], ab[i], ab[i - 1], 0.1, literal=False)
        assert np.abs(lit - idn).max() <= 3e-7


def test_softmax_invariants_and_edges(orc):
    rews, Y0s, Ybar = _inputs(seed=2)
    a, ab, _ = orc.schedule(1e-4, 1e-2, 100)
    _, w0, _ = orc.score_update(rews, Y0s, Ybar, a[10], ab[10], ab[9], 0.1)
    _, w1, _ = orc.score_update(rews + np.float32(8.0), Y0s, Ybar, a[10], ab[10], ab[9], 0.1)
    assert np.allclose(w0, w1, rtol=1e-4, atol=1e-8)  # shift invariance
    # rew_std < 1e-4 -> 1.0 guard (:112): constant rewards give uniform weights
    const = np.full(96, 0.25, np.float32)
    out, w, m = orc.score_update(const, Y0s, Ybar, a[10], ab[10], ab[9], 0.1, literal=False)
    assert np.allclose(w, 1 / 96) and np.allclose(out, Y0s.mean(0), atol=1e-6) and m == np.float32(0.25)
    # N = 1
    out, w, m = orc.score_update(rews[:1], Y0s[:1], Ybar, a[10], ab[10], ab[9], 0.1, literal=False)
    assert w.tolist() == [1.0] and np.array_equal(out, Y0s[0])


def test_demo_blend_semantics(orc):
    """:117-125 — demo log-density replaces logp0 where larger, then re-standardise and /temp AGAIN."""
    rews, Y0s, Ybar = _inputs(seed=3)
    lp = -np.abs(np.random.default_rng(4).normal(size=96)).astype(np.float32)
    a, ab, _ = orc.schedule(1e-4, 1e-2, 100)
    temp, rx = 0.1, 1.0
    _, w, _ = orc.score_update(rews, Y0s, Ybar, a[10], ab[10], ab[9], temp, lp_demo=lp, rew_xref=rx)
    r = rews.astype(np.float64)
    l0 = (r - r.mean()) / r.std() / temp
    ld = ((lp.astype(np.float64) - lp.max()) + rx - r.mean()) / r.std() / temp
    l = np.where(ld > l0, ld, l0)
    l = (l - l.mean()) / l.std() / temp
    wr = np.exp(l - l.max())
    wr /= wr.sum()
    assert np.allclose(w, wr, rtol=5e-4, atol=1e-9)


def test_mean_h_is_sequential_sum_over_horizon(orc):
    from oracle.planner import mean_h
    x = np.random.default_rng(5).normal(size=(4, 50)).astype(np.float32)
    got = mean_h(orc, x)
    for b in range(4):
        s = np.float32(0)
        for t in range(50):
            s = np.float32(s + x[b, t])
        assert got[b] == np.float32(s / np.float32(50))
