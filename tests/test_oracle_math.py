"""Ties the oracle's two arithmetic flavours together: the portable IEEE sequences (which the HIP
kernels reproduce bit for bit) against the host libm the reference calls."""
import numpy as np

from oracle import orc


def _ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)))


def _both(fn, x):
    orc.set_math_mode(orc.MATH_LIBM)
    a = fn(x)
    orc.set_math_mode(orc.MATH_PORTABLE)
    b = fn(x)
    return a, b


def test_log_within_one_ulp_of_libm():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(400000), 2.0 ** -rng.uniform(0, 60, 100000),
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 2.0 ** -0.5, 1e-300]])
    x = x[x > 0]
    a, b = _both(orc.math_log, x)
    assert _ulp_diff(a, np.log(x)).max() <= 1.0     # numpy ships its own log; both are < 1 ulp
    assert _ulp_diff(a, b).max() <= 1.0
    assert (a == b).mean() > 0.9      # identical in 94 % of the cases, 1 ulp apart in the rest


def test_sincos_within_two_ulp_of_libm():
    rng = np.random.default_rng(1)
    phi = 2.0 * np.pi * np.concatenate([rng.random(400000), np.arange(9) / 8.0 * (1 - 2.0 ** -53)])
    (s0, c0), (s1, c1) = _both(orc.math_sincos, phi)
    # away from the zeros of sin / cos the two agree to an ulp; near a zero the error is measured
    # against 1 (absolute), like any argument-reduced implementation
    for a, b in ((s0, s1), (c0, c1)):
        big = np.abs(a) > 1e-3
        assert _ulp_diff(a[big], b[big]).max() <= 2.0
        assert np.abs(a - b).max() < 3e-16


def test_sincos2pi_against_libm_and_exact():
    """sin / cos of 2 pi u as the step functions form it from the uniform u (azimuth of every
    direction sample).  The libm flavour is the reference's `phi = 2.0 * M_PI * u; sin(phi)`:
    phi carries the rounding of the product (<= 4.4e-16 at phi ~ 2 pi), so it is within 7e-16
    (absolute) of the exact value; the portable flavour reduces on u itself and is within
    1.8e-16.  The two flavours therefore differ by <= 9e-16 absolute."""
    rng = np.random.default_rng(3)
    k = np.concatenate([rng.integers(0, 2 ** 52, 400000), np.arange(257) * 2 ** 44,
                        np.arange(1, 257) * 2 ** 44 - 1, [0, 2 ** 52 - 1]])
    u = (k.astype(np.float64) + 0.5) * 2.0 ** -52          # what the generator delivers
    (s0, c0), (s1, c1) = _both(orc.math_sincos2pi, u)
    ld = np.longdouble
    two_pi = ld(8) * np.arctan(ld(1))
    ref_s, ref_c = np.sin(two_pi * u.astype(ld)), np.cos(two_pi * u.astype(ld))
    for a, b, r in ((s0, s1, ref_s), (c0, c1, ref_c)):
        assert np.abs(b - r).astype(np.float64).max() < 1.8e-16     # portable vs exact
        assert np.abs(a - r).astype(np.float64).max() < 7.5e-16     # libm flavour vs exact
        assert np.abs(a - b).max() < 9e-16
    assert np.abs(s1 * s1 + c1 * c1 - 1.0).max() < 5e-16


def test_acos_within_one_ulp_of_libm():
    rng = np.random.default_rng(2)
    x = np.concatenate([2.0 * rng.random(400000) - 1.0, [-1.0, 1.0, 0.0, 0.5, -0.5, 1 - 2.0 ** -53]])
    a, b = _both(orc.math_acos, x)
    assert _ulp_diff(a, b).max() <= 1.0


def test_exact_values():
    orc.set_math_mode(orc.MATH_PORTABLE)
    assert orc.math_log(np.array([1.0]))[0] == 0.0
    s, c = orc.math_sincos(np.array([0.0]))
    assert s[0] == 0.0 and c[0] == 1.0
    s, c = orc.math_sincos2pi(np.array([0.25, 0.5, 0.75]))
    assert list(c) == [0.0, -1.0, 0.0] and list(s) == [1.0, 0.0, -1.0]
    assert orc.math_acos(np.array([1.0]))[0] == 0.0
    assert orc.math_acos(np.array([-1.0]))[0] == np.pi
