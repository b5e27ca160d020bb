"""The numerical-contract primitives (oracle/spec_math.h) against numpy in float64."""
import ctypes as C

import numpy as np


def test_atan2_and_asin_accuracy(orc):
    rng = np.random.default_rng(0)
    y = rng.normal(size=5000).astype(np.float32)
    x = rng.normal(size=5000).astype(np.float32)
    got = np.array([orc.lib.orc_sp_atan2(float(a), float(b)) for a, b in zip(y, x)])
    assert np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 4e-7
    assert orc.lib.orc_sp_atan2(0.0, 0.0) == 0.0
    assert abs(orc.lib.orc_sp_atan2(1.0, 0.0) - np.pi / 2) < 2e-7
    assert abs(orc.lib.orc_sp_atan2(0.0, -1.0) - np.pi) < 3e-7
    v = np.linspace(-1, 1, 2001).astype(np.float32)
    got = np.array([orc.lib.orc_sp_asin(float(a)) for a in v])
    assert np.abs(got - np.arcsin(v.astype(np.float64))).max() < 5e-7


def test_sincos_accuracy(orc):
    xs = np.linspace(-40, 40, 8001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    err = 0.0
    for x in xs:
        orc.lib.orc_sp_sincos(float(x), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert err < 3e-7


def test_exp_log1p_accuracy(orc):
    xs = np.linspace(-80, 0, 4001).astype(np.float32)
    got = np.array([orc.lib.orc_sp_exp(float(x)) for x in xs])
    ref = np.exp(xs.astype(np.float64))
    assert (np.abs(got - ref) / ref).max() < 3e-7
    assert orc.lib.orc_sp_exp(-100.0) == 0.0
    ts = -np.linspace(0, 0.999999, 4001).astype(np.float32)
    got = np.array([orc.lib.orc_sp_log1p(float(t)) for t in ts])
    ref = np.log1p(ts.astype(np.float64))
    assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)).max() < 5e-7


def test_rotation_primitives(orc):
    rng = np.random.default_rng(1)
    for _ in range(200):
        q = rng.normal(size=4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        v = rng.normal(size=3).astype(np.float32)
        out = np.zeros(3, np.float32)
        orc.lib.orc_sp_rot(v, q, out)
        w, x, y, z = q.astype(np.float64)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.abs(out - R @ v).max() < 2e-6
        # asymmetric check of the quaternion product: (a*b) rotates like a after b
        b = rng.normal(size=4)
        b = (b / np.linalg.norm(b)).astype(np.float32)
        ab = np.zeros(4, np.float32)
        orc.lib.orc_sp_qmul(q, b, ab)
        o1, o2, o3 = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        orc.lib.orc_sp_rot(v, b, o1)
        orc.lib.orc_sp_rot(o1, q, o2)
        orc.lib.orc_sp_rot(v, ab, o3)
        assert np.abs(o2 - o3).max() < 3e-6


def test_canonical_sum_is_strided_then_butterfly(orc):
    x = np.random.default_rng(2).normal(size=1000).astype(np.float32)
    part = np.zeros(64, np.float32)
    for i, v in enumerate(x):
        part[i & 63] = np.float32(part[i & 63] + v)
    a = part.copy()
    off = 32
    while off >= 1:
        a = np.array([np.float32(a[j] + a[j ^ off]) for j in range(64)], np.float32)
        off >>= 1
    assert orc.lib.orc_sp_sum(x, 1000) == a[0]


def test_car2d_collision_threshold_is_the_sqrt_comparison():
    """csrc/mbd_kernels.h (car2d_rollout_kernel) tests `dx*dx + dy*dy < T` where car2d.py:84 and the checker compare
    `sqrt(dx*dx + dy*dy) < 0.3f`: T = 0x3db851ec is the smallest float32 whose correctly rounded square root is >= 0.3f,
    and the correctly rounded square root is monotonic — checked here on every float32 in [0.0899, 0.0901]."""
    c = np.float32(0.3)
    lo, hi = np.float32(0.0899), np.float32(0.0901)
    bits = np.arange(lo.view(np.uint32), hi.view(np.uint32) + 1, dtype=np.uint32)
    xs = bits.view(np.float32)
    s = np.sqrt(xs)  # float32: IEEE, correctly rounded
    T = np.uint32(0x3DB851EC).view(np.float32)
    assert float(T) == 0.09000000357627869
    assert np.array_equal(s < c, xs < T)
    assert np.all(np.diff(s) >= 0)
