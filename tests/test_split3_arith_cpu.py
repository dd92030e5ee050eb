"""CPU: the arithmetic claim behind cfg.pass_arith = EKF_ARITH_SPLIT3 (ekf_slam_amd/csrc/flush32_split.h), restated in NumPy and checked without
a GPU: (1) a float IS three bfloat16 pieces -- b1 = bf16(v), b2 = bf16(v - b1), b3 = bf16(v - b1 - b2), round-to-nearest-even on the bits as
k_split_pairs does it, sum to v EXACTLY, every residual exact in float; (2) every partial product of two pieces is exact in float (8 x 8
significant bits); (3) the six partial products the pass keeps (a3 b1 + a2 b2 + a1 b3 + a2 b1 + a1 b2 + a1 b1) differ from a b by the dropped
a2 b3 + a3 b2 + a3 b3: at most 2^-23 |a b| (|a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|), and far less on average -- measured here: root mean
square below 0.1 x 2^-24 |a b|, where ONE float rounding of the product has 0.29 x 2^-24 (its unit roundoff is 2^-24)."""
import numpy as np


def bf16_rn(v):
    """float32 array -> the float32 value of its round-to-nearest-even bfloat16 (integer arithmetic on the bits, as bf16_rn_bits)."""
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(v):
    v = np.asarray(v, dtype=np.float32)
    b1 = bf16_rn(v)
    r1 = (v - b1).astype(np.float32)
    b2 = bf16_rn(r1)
    r2 = (r1 - b2).astype(np.float32)
    b3 = bf16_rn(r2)
    return b1, b2, b3, r1, r2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-40, 40, n)
    sign = rng.choice([-1.0, 1.0], n)
    v = (sign * mant * 2.0 ** expo).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 3.0, 255.5, 256.0 - 2 ** -15, 1.9999999, 1e-30, -7.0e20,
                     np.float32(np.pi), np.float32(1) / np.float32(3), 0.1, 1.00390625, 1.005859375], dtype=np.float32)   # ties of the first cut among them
    return np.concatenate([v, edge])


def test_three_bf16_pieces_sum_to_the_float_exactly():
    v = _samples(400000, 1)
    b1, b2, b3, r1, r2 = split3(v)
    f64 = np.float64
    assert np.array_equal(f64(r1), f64(v) - f64(b1))                 # the residuals are exact in float
    assert np.array_equal(f64(r2), f64(r1) - f64(b2))
    assert np.array_equal(f64(b1) + f64(b2) + f64(b3), f64(v))       # nothing is left after the third piece
    for b in (b1, b2, b3):                                           # every piece IS a bfloat16: its low 16 bits are zero
        assert not np.any(b.view(np.uint32) & 0xFFFF)
    nz = v != 0
    assert np.all(np.abs(f64(b2[nz])) <= 2.0 ** -8 * np.abs(f64(v[nz])))     # |b2| <= half a bf16 ulp of v (<= 2^-8 |v|), |b3| <= 2^-16 |v|
    assert np.all(np.abs(f64(b3[nz])) <= 2.0 ** -16 * np.abs(f64(v[nz])))


def test_partial_products_are_exact_and_the_six_kept_ones_are_within_a_quarter_rounding():
    a, b = _samples(300000, 2), _samples(300000, 3)[::-1].copy()
    n = min(a.size, b.size)
    a, b = a[:n], b[:n]
    A, B = split3(a)[:3], split3(b)[:3]
    f64 = np.float64
    for p in range(3):
        for q in range(3):
            exact = f64(A[p]) * f64(B[q])
            normal = (np.abs(exact) >= 2.0 ** -126) | (exact == 0)    # (a product below the float's normal range is subnormal: not exact, and 1e-38)
            assert np.array_equal(f64(np.float32(exact[normal])), exact[normal])     # 8 x 8 significant bits: exact in float
    kept = sum(f64(A[p]) * f64(B[q]) for p, q in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)))
    ab = f64(a) * f64(b)
    nz = ab != 0
    rel = np.abs(kept[nz] - ab[nz]) / np.abs(ab[nz])
    rms = float(np.sqrt(np.mean(rel ** 2)))
    print("six kept partial products against a b, in units of 2^-24 |a b|: max %.3f  rms %.4f  mean %.4f" % (rel.max() * 2 ** 24, rms * 2 ** 24, rel.mean() * 2 ** 24))
    assert rel.max() <= 2.0 ** -23 * (1 + 2.0 ** -8)                  # the bound: 2^-8 2^-16 + 2^-16 2^-8 + 2^-16 2^-16
    assert rms <= 0.1 * 2.0 ** -24
