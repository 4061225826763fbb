"""Shared primitives (csrc/shared/*.h), evaluated on the host: accuracy of the deterministic math against
80-bit long double, Philox4x32-10 against the published Random123 known-answer vectors, fixed-point
conversions against Python integers."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

import oracle_binding as ob

LD = np.longdouble


def _ulp_err(got, ref_ld):
    ulp = np.spacing(np.abs(got)).astype(LD)
    return np.max(np.abs(got.astype(LD) - ref_ld) / ulp)


def test_exp_accuracy_and_specials():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-708, 5, 400000), rng.uniform(-1, 0, 100000), -10.0 ** rng.uniform(-20, 2, 100000)])
    e = ob.math_vec(0, x)
    assert _ulp_err(e, np.exp(x.astype(LD))) < 1.0
    sp = ob.math_vec(0, np.array([0.0, -0.0, -1e-300, -745.0, -746.0, 710.0, np.nan, -np.inf]))
    assert sp[0] == 1.0 and sp[1] == 1.0 and sp[2] == 1.0
    assert sp[3] == 5e-324 and sp[4] == 0.0 and np.isinf(sp[5]) and np.isnan(sp[6]) and sp[7] == 0.0
    xn = np.concatenate([x[x <= 0], [-745.0, -745.2, -746.0, -800.0, -1e6, -np.inf, 0.0]])
    assert np.array_equal(ob.math_vec(8, xn), ob.math_vec(0, xn))      # exp_le0 == exp on its domain
    assert np.isnan(ob.math_vec(8, np.array([np.nan]))[0])
    # monotone on a fine grid around 0 (the exp-weights of near-maximal particles)
    g = np.linspace(-1e-3, 0, 100001)
    assert np.all(np.diff(ob.math_vec(0, g)) >= 0)


def test_log_and_log1p_accuracy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 300000), 10.0 ** rng.uniform(-300, 300, 200000), [1.0, 2.0, 0.5, 5e-324]])
    l = ob.math_vec(1, x)
    assert _ulp_err(l[l != 0], np.log(x.astype(LD))[l != 0]) < 1.0
    assert ob.math_vec(1, np.array([1.0]))[0] == 0.0
    assert np.isneginf(ob.math_vec(1, np.array([0.0]))[0]) and np.isnan(ob.math_vec(1, np.array([-1.0]))[0])
    s = np.concatenate([10.0 ** rng.uniform(-25, 8, 300000), [0.0, 1e-300]])
    l1 = ob.math_vec(2, s)
    nz = l1 != 0
    assert _ulp_err(l1[nz], np.log1p(s.astype(LD))[nz]) < 1.5        # scalar-only path (once per filter step)


def test_sincos2pi_accuracy():
    rng = np.random.default_rng(2)
    u = np.concatenate([rng.uniform(0, 1, 400000), [0.0, 0.25, 0.5, 0.75, 0.125]])
    s, c = ob.math_vec(3, u), ob.math_vec(4, u)
    two_pi = LD(2) * LD("3.14159265358979323846264338327950288")
    a = u.astype(LD) * two_pi
    # absolute error below half an ulp of 1 everywhere, relative error < 1 ulp away from the zeros
    assert np.max(np.abs(s.astype(LD) - np.sin(a))) < 1.2e-16
    assert np.max(np.abs(c.astype(LD) - np.cos(a))) < 1.2e-16
    big = np.abs(s) > 0.3
    assert _ulp_err(s[big], np.sin(a)[big]) < 1.0
    big = np.abs(c) > 0.3
    assert _ulp_err(c[big], np.cos(a)[big]) < 1.0
    assert s[-5] == 0.0 and c[-5] == 1.0 and s[-4] == 1.0 and c[-3] == -1.0 and s[-2] == -1.0
    np.testing.assert_allclose(s * s + c * c, 1.0, rtol=0, atol=4e-16)


def test_rng_fast_log_and_sincos():
    """Table-driven forms used only inside the Box-Muller transform (llpf_rngmath.h)."""
    rng = np.random.default_rng(7)
    u = np.concatenate([(rng.integers(0, 2 ** 53, 400000) + 1) * 2.0 ** -53, 1 - rng.uniform(0, 1e-6, 50000),
                        rng.uniform(0.7, 1, 200000), [1.0, 2.0 ** -53, 0.75, 0.5, 1 - 2.0 ** -53]])
    l = ob.math_vec(9, u)
    ref = np.log(u.astype(LD))
    assert np.all(l <= 0) and ob.math_vec(9, np.array([1.0]))[0] == 0.0
    assert np.all(np.abs(l.astype(LD) - ref) <= LD(1.2e-16) + 2 * np.spacing(np.abs(l)).astype(LD))
    near1 = np.abs(u - 1) < 1 / 130
    assert _ulp_err(l[near1 & (l != 0)], ref[near1 & (l != 0)]) < 2.0          # exact-relative around u = 1
    uu = np.concatenate([rng.uniform(0, 1, 400000), [0, 0.25, 0.5, 0.75, 1 / 64, 0.5 - 2 ** -54]])
    s, c = ob.math_vec(10, uu), ob.math_vec(11, uu)
    a = uu.astype(LD) * LD(2) * LD("3.14159265358979323846264338327950288")
    assert np.max(np.abs(s.astype(LD) - np.sin(a))) < 2e-16 and np.max(np.abs(c.astype(LD) - np.cos(a))) < 2e-16
    assert s[-6] == 0.0 and c[-6] == 1.0 and s[-5] == 1.0 and c[-4] == -1.0 and s[-3] == -1.0


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    L = ob.lib()

    def ph(c, k):
        out = (C.c_uint32 * 4)()
        L.orc_philox_block(c[0], c[1], c[2], c[3], k[0], k[1], out)
        return [x for x in out]
    assert ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_known_answers_of_the_shipped_generator():
    """The generator every draw of the engine goes through is Philox4x32 stopped after LLPF_PHILOX_ROUNDS rounds (7 since round 4).
    Random123's kat_vectors hold known answers for that instance too ("philox4x32 7 ..."): the same three inputs as the 10-round test.
    (The expected words were cross-checked with a literal Python evaluation of the published round function, which shares nothing
    with the C header.)"""
    L = ob.lib()

    def ph(c, k):
        out = (C.c_uint32 * 4)()
        rounds = L.orc_philox_block_engine(c[0], c[1], c[2], c[3], k[0], k[1], out)
        return rounds, [x for x in out]
    kat = {7: ([0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48], [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662],
               [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
           10: ([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd],
                [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])}
    rounds, a = ph([0] * 4, [0] * 2)
    assert rounds in kat, "no Random123 vector for a %d-round build" % rounds
    assert a == kat[rounds][0]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2)[1] == kat[rounds][1]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])[1] == kat[rounds][2]

    def py_philox(c, k, R):            # the published round function, literally
        c, k = list(c), list(k)
        for _ in range(R):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return c
    rng = np.random.default_rng(5)
    for _ in range(200):
        c, k = [int(v) for v in rng.integers(0, 2 ** 32, 4)], [int(v) for v in rng.integers(0, 2 ** 32, 2)]
        assert ph(c, k)[1] == py_philox(c, k, rounds)


def test_normals_are_standard_normal():
    from scipy import stats
    z = ob.normals(2024, 3, 1, 4, 500000)
    assert abs(z.mean()) < 4e-3 and abs(z.var() - 1) < 5e-3
    assert np.max(np.abs(np.corrcoef(z.T) - np.eye(4))) < 5e-3
    assert stats.kstest(z[:200000, 0], "norm").pvalue > 1e-3
    assert stats.kstest(z[:200000, 3], "norm").pvalue > 1e-3
    assert np.abs(z).max() < 8.5 and np.abs(z).max() > 4.5
    # streams, steps and seeds decorrelate; the same key reproduces
    assert np.array_equal(z, ob.normals(2024, 3, 1, 4, 500000))
    assert abs(np.corrcoef(z[:, 0], ob.normals(2024, 4, 1, 4, 500000)[:, 0])[0, 1]) < 5e-3
    assert abs(np.corrcoef(z[:, 0], ob.normals(2025, 3, 1, 4, 500000)[:, 0])[0, 1]) < 5e-3
    # sub-blocks: dims (0,1) of an nd=2 draw are dims (0,1) of the nd=4 draw
    assert np.array_equal(ob.normals(2024, 3, 1, 2, 1000), z[:1000, :2])


def test_fixed_point_conversions_exact():
    L = ob.lib()
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.uniform(0, 1, 2000), 10.0 ** rng.uniform(-40, 0, 2000), [1.0, 0.5, 2.0 ** -96, 2.0 ** -97, 0.0, 5e-324]])
    for e in vals:
        lo_hi = (C.c_uint64 * 2)()
        L.orc_fix96(float(e), lo_hi)
        got = lo_hi[0] + (lo_hi[1] << 64)
        exp = int(Fraction(float(e)) * (1 << 96)) if e >= 2.0 ** -1022 else 0      # floor for positive values
        assert got == exp, e
        L.orc_fix96_unit(float(e), lo_hi)
        assert lo_hi[0] + (lo_hi[1] << 64) == exp, e            # branch-free form for [0,1]
        for K in (31, 42, 62):
            exp_q = int(Fraction(float(e)) * (1 << K)) if e >= 2.0 ** -1022 else 0
            assert L.orc_q64(float(e), K) == exp_q
            assert L.orc_q64_unit(float(e), K) == exp_q
    for bad in (-0.5, 2.5, np.inf, np.nan, -np.inf):            # outside [0,2) -> 0 in the unit forms
        lo_hi = (C.c_uint64 * 2)()
        L.orc_fix96_unit(float(bad), lo_hi)
        assert lo_hi[0] == 0 and lo_hi[1] == 0 and L.orc_q64_unit(float(bad), 40) == 0
    # 128-bit -> double conversion is round-to-nearest-even
    for _ in range(3000):
        bits = int(rng.integers(1, 128))
        v = int(rng.integers(0, 2 ** 63)) * int(rng.integers(0, 2 ** 63)) >> (126 - bits) if bits < 126 else int(rng.integers(0, 2 ** 63)) * int(rng.integers(0, 2 ** 63))
        v &= (1 << 128) - 1
        assert L.orc_u128_to_double(v & (2 ** 64 - 1), v >> 64) == float(v)
    for v in (2 ** 53 + 1, 2 ** 64 + 2 ** 11, 2 ** 64 + 2 ** 11 + 1, 2 ** 100 + 2 ** 47, 2 ** 100 + 2 ** 47 + 1, 2 ** 128 - 1):
        assert L.orc_u128_to_double(v & (2 ** 64 - 1), v >> 64) == float(v)


def test_device_order_sums_are_order_independent():
    """The point of the fixed-point formulation: permuting the weights does not change ll or the bins total."""
    rng = np.random.default_rng(6)
    w = rng.standard_normal(5000) * 10
    ll0, _, we0, _ = ob.logsumexp(w, ob.ORDER_DEVICE)
    for _ in range(5):
        p = rng.permutation(5000)
        ll, _, we, _ = ob.logsumexp(w[p], ob.ORDER_DEVICE)
        assert ll == ll0
        assert np.array_equal(we, we0[p])
    # whereas the reference order (pairwise fp64 sum) moves by rounding, within the stated tolerance
    llr, _, wer, _ = ob.logsumexp(w, ob.ORDER_REFERENCE)
    assert abs(llr - ll0) < 1e-12 and np.max(np.abs(wer - we0) / wer) < 1e-12
