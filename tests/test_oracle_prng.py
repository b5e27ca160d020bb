"""Oracle pinning: JAX PRNG restatement."""
import numpy as np
from scipy import special


def test_threefry_random123_known_answers(orc):
    # Random123 threefry2x32-20 KATs (also asserted by JAX's own test-suite)
    assert orc.threefry2x32(0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert orc.threefry2x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert orc.threefry2x32(0x13198A2E, 0x03707344, 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)


def test_prng_key_layout(orc):
    assert orc.prng_key(0).tolist() == [0, 0]
    assert orc.prng_key(42).tolist() == [0, 42]
    assert orc.prng_key((7 << 32) | 5).tolist() == [7, 5]


def test_split_layouts(orc):
    key = orc.prng_key(0)
    # partitionable: key_j = threefry(key, 0, j)
    ks = orc.split(key, 3, impl=1)
    for j in range(3):
        assert tuple(ks[j]) == orc.threefry2x32(0, 0, 0, j)
    # legacy: threefry_2x32(key, iota(2*num)) -> x0 = first half, x1 = second half, concat, reshape (num,2)
    ks = orc.split(key, 2, impl=0)
    a, b = orc.threefry2x32(0, 0, 0, 2), orc.threefry2x32(0, 0, 1, 3)
    assert ks.tolist() == [[a[0], b[0]], [a[1], b[1]]]
    ks3 = orc.split(key, 3, impl=0)
    t = [orc.threefry2x32(0, 0, j, j + 3) for j in range(3)]
    flat = [t[0][0], t[1][0], t[2][0], t[0][1], t[1][1], t[2][1]]
    assert ks3.reshape(-1).tolist() == flat


def test_random_bits_legacy_odd_size_padding(orc):
    key = np.array([1, 2], np.uint32)
    size = 5  # padded to 6: counts [0,1,2 | 3,4,0]
    bits = [orc.lib.orc_random_bits32(key, 0, j, size) for j in range(size)]
    blocks = [orc.threefry2x32(1, 2, 0, 3), orc.threefry2x32(1, 2, 1, 4), orc.threefry2x32(1, 2, 2, 0)]
    assert bits == [blocks[0][0], blocks[1][0], blocks[2][0], blocks[0][1], blocks[1][1]]


def test_uniform_range_and_bit_trick(orc):
    key = orc.prng_key(3)
    for impl in (0, 1):
        u = orc.uniform(key, 4096, -0.01, 0.01, impl)
        assert u.min() >= -0.01 and u.max() < 0.01
        assert abs(u.mean()) < 1e-3
    # the bit trick: bits>>9 | 0x3f800000 -> [1,2) - 1
    bits = orc.lib.orc_random_bits32(key, 1, 0, 16)
    f = np.array([(bits >> 9) | 0x3F800000], np.uint32).view(np.float32)[0] - np.float32(1.0)
    assert orc.uniform(key, 16, 0.0, 1.0, 1)[0] == f


def test_erfinv_matches_giles_accuracy(orc):
    xs = np.linspace(-0.999999, 0.999999, 4001).astype(np.float32)
    got = np.array([orc.erfinv(float(x)) for x in xs])
    ref = special.erfinv(xs.astype(np.float64))
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    assert rel.max() < 1e-5  # the single-precision polynomial itself is ~6e-6 accurate
    assert orc.erfinv(1.0) == np.inf and orc.erfinv(-1.0) == -np.inf and orc.erfinv(0.0) == 0.0


def test_normal_statistics(orc):
    key = orc.prng_key(11)
    for impl in (0, 1):
        z = orc.normal(key, (64, 50, 17), impl)
        assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
        assert np.isfinite(z).all()
    # the two layouts are different streams
    assert not np.array_equal(orc.normal(key, (8, 8), 0), orc.normal(key, (8, 8), 1))


def test_sample_rows_are_slices_of_the_global_tensor(orc):
    """Sharding contract: rows [begin, begin+count) generated on their own equal the same rows of the
    full [N, H*Nu] tensor (both PRNG layouts) — what lets every rank regenerate any candidate."""
    key = orc.prng_key(5)
    Ybar = np.linspace(-0.2, 0.2, 6 * 3).astype(np.float32).reshape(6, 3)
    for impl in (0, 1):
        full = orc.sample(key, impl, 16, 6, 3, 0, 16, 0.3, Ybar)
        part = orc.sample(key, impl, 16, 6, 3, 5, 7, 0.3, Ybar)
        assert np.array_equal(full[5:12], part)
        assert np.abs(full).max() <= 1.0


# ---- outputs of the real JAX, as printed in its public documentation ---------------------------------------
# These are the only reference-side numbers for this path that exist outside a JAX install, and they pin
# the whole noise chain of mbd_planner.py:97,103 (PRNGKey -> split -> random_bits -> uniform -> erf_inv):
#   * "JAX - The Sharp Bits", section "Random numbers" (jax_threefry_partitionable=False, the default up to
#     JAX 0.4.x): PRNGKey(0); normal(key, (1,)) = [-0.20584226]; split -> key [4146024105 967050713],
#     subkey [2718843009 1272950319]; normal(subkey, (1,)) = [-1.2515389]; uniform(PRNGKey(0)) = 0.41845703
#   * "Pseudorandom numbers" tutorial, old edition (PRNGKey(42), legacy layout): normal(key) = -0.18471177,
#     normal(key, (3,)) "all at once" = [0.18693547 -1.2806505 -1.5593132]
#   * the same tutorial, JAX >= 0.5 edition (key(42), jax_threefry_partitionable=True by default):
#     normal(key) = -0.028304616; split -> new_key [1832780943 270669613]; normal(subkey) = 0.60576403
JAX_DOC_LEGACY, JAX_DOC_PARTITIONABLE = 0, 1


def test_published_jax_outputs_legacy_layout(orc):
    k0 = orc.prng_key(0)
    ks = orc.split(k0, 2, JAX_DOC_LEGACY)
    assert ks.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert orc.normal(k0, (1,), JAX_DOC_LEGACY)[0] == np.float32(-0.20584226)
    assert orc.normal(ks[1], (1,), JAX_DOC_LEGACY)[0] == np.float32(-1.2515389)
    assert orc.uniform(k0, 1, 0.0, 1.0, JAX_DOC_LEGACY)[0] == np.float32(0.41845703)
    k42 = orc.prng_key(42)
    assert orc.normal(k42, (1,), JAX_DOC_LEGACY)[0] == np.float32(-0.18471177)
    assert np.array_equal(orc.normal(k42, (3,), JAX_DOC_LEGACY),
                          np.array([0.18693547, -1.2806505, -1.5593132], np.float32))


def test_published_jax_outputs_partitionable_layout(orc):
    k42 = orc.prng_key(42)
    assert orc.normal(k42, (1,), JAX_DOC_PARTITIONABLE)[0] == np.float32(-0.028304616)
    ks = orc.split(k42, 2, JAX_DOC_PARTITIONABLE)
    assert ks[0].tolist() == [1832780943, 270669613]
    assert orc.normal(ks[1], (1,), JAX_DOC_PARTITIONABLE)[0] == np.float32(0.60576403)
