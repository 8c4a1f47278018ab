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


def test_one_minus_exp_neg_close_to_libm():
    """1 - exp(-x) (EPBremss stimulated-emission factor): the specified sequence is within 4 ulp
    of -expm1(-x) and exact at its edges."""
    rng = np.random.default_rng(5)
    x = np.concatenate([10 ** rng.uniform(-300, 2, 200000), rng.uniform(0.0, 45.0, 200000),
                        np.linspace(0.24, 0.26, 20001)])
    a, b = _both(orc.math_one_minus_exp_neg, x)
    assert _ulp_diff(a, -np.expm1(-x)).max() <= 1.0     # (numpy's expm1 is not glibc's)
    assert _ulp_diff(a, b).max() <= 4.0
    orc.set_math_mode(orc.MATH_PORTABLE)
    assert list(orc.math_one_minus_exp_neg(np.array([0.0, 40.0, 1e300, np.inf]))) == [0.0, 1.0, 1.0, 1.0]


def test_epbremss_model_is_consistent():
    """The stand-in for singularity-opac's EPBremss: Kirchhoff's law ties the absorption
    coefficient to the frequency-integrated emissivity (4 pi int alpha_nu B_nu dnu = j), the CGS
    coefficients are Rybicki & Lightman's 3.692e8 / 1.426e-27, and the code -> CGS scales commute
    with the evaluation."""
    h, kb, cl, mp = 6.62607015e-27, 1.380649e-16, 2.99792458e10, 1.67262192369e-24
    co = orc.model_coefficients()
    assert abs(co["ep_A"] * mp * mp / 3.692e8 - 1.0) < 2e-4
    assert abs(co["ep_E"] * mp * mp / 1.426e-27 - 1.0) < 4e-4
    assert abs(co["kappa_s_thomson"] / 6.6524587e-25 - 1.0) < 1e-8
    P = dict(opac_model=1, ep_A=co["ep_A"], ep_B=co["ep_B"], ep_E=co["ep_E"], kappa_s=0.0, apm=1.0)
    orc.set_math_mode(orc.MATH_PORTABLE)
    for rho, T in ((1e-3, 1e6), (2.0, 3e4)):
        nu = np.logspace(6, 19, 200001)
        al = orc.model_eval(P, 0, np.full_like(nu, rho), np.full_like(nu, T), nu)
        B = 2 * h * nu ** 3 / cl ** 2 / np.expm1(np.minimum(h * nu / (kb * T), 700.0))
        f = al * B
        j = 4 * np.pi * np.sum(0.5 * (f[1:] + f[:-1]) * np.diff(nu))
        assert abs(j / orc.model_eval(P, 1, [rho], [T], [1.0])[0] - 1.0) < 1e-6
    tau, mu, lam, th = 3e-9, 2e-7, 0.5, 1.1e3
    cs = orc.model_coefficients(tau, mu, lam, th)
    Ps = dict(opac_model=1, ep_A=cs["ep_A"], ep_B=cs["ep_B"], ep_E=cs["ep_E"], kappa_s=0.0, apm=1.0)
    rho, T, nu = 3.0, 40.0, 2.5e4
    a_code = orc.model_eval(Ps, 0, [rho], [T], [nu])[0]
    a_cgs = orc.model_eval(P, 0, [rho * mu / lam ** 3], [T * th], [nu / tau])[0]
    assert abs(a_code / (a_cgs * lam) - 1.0) < 1e-13
    j_code = orc.model_eval(Ps, 1, [rho], [T], [nu])[0]
    j_cgs = orc.model_eval(P, 1, [rho * mu / lam ** 3], [T * th], [1.0])[0]
    assert abs(j_code / (j_cgs * tau ** 3 * lam / mu) - 1.0) < 1e-13
    assert abs(cs["kappa_s_thomson"] * lam ** 2 / co["kappa_s_thomson"] - 1.0) < 1e-15
