/* oracle/orc_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Transcendentals used by the history loop, in two selectable flavours:
 *
 *   ORC_MATH_LIBM      the reference's own arithmetic: std::log / std::sin / std::cos /
 *                      std::acos / std::pow from the host libm (what the reference CPU build
 *                      calls at transport_utils.hpp:31-38,118-119,185,270-275,
 *                      scattering.hpp:23-28, planck.hpp:30-49, sourcing.cpp:93,180-185).
 *   ORC_MATH_PORTABLE  a fully specified IEEE-754 sequence (only +,-,*,/,sqrt,fma, integer bit
 *                      manipulation and two generated lookup tables, orc_tables.h): table-driven
 *                      log and sincos, fdlibm's e_acos.c (Sun Microsystems 1993).  The HIP
 *                      device code implements the same specification, which makes CPU-oracle
 *                      and GPU results BIT-IDENTICAL event for event.  tests/test_oracle_math.py
 *                      bounds the difference between the two flavours (log, acos <= 1 ulp;
 *                      sin, cos <= 2.5e-16 absolute).
 *
 * All other arithmetic (+,-,*,/,sqrt) is IEEE correctly rounded on both sides; the oracle is
 * compiled with -ffp-contract=off so no multiply-add is fused unless written as fma().
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc_tables.h"

#define ORC_MATH_LIBM 0
#define ORC_MATH_PORTABLE 1

extern int orc_math_mode; /* defined in orc.c */

static inline uint64_t orc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double orc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ---- portable log: x positive, finite, normal (table-driven, no division; see the description
 * in jaybenne_amd/csrc/jb_math.hpp -- this is the same sequence, written independently) ------- */
static inline double orc_pm_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const uint64_t ix = orc_d2u(x);
  const uint64_t tmp = ix - JB_LOG_OFF;
  const int i = (int)((tmp >> 45) & (JB_LOG_N - 1));
  const int k = (int)((int64_t)tmp >> 52);
  const double z = orc_u2d(ix - (tmp & 0xfff0000000000000ull));
  const double r = fma(z, jb_log_tab[i][0], -1.0);
  const double kd = (double)k;
  const double w = fma(kd, ln2_hi, jb_log_tab[i][1]);
  const double hi = w + r;
  const double lo = ((w - hi) + r) + fma(kd, ln2_lo, jb_log_tab[i][2]);
  const double r2 = r * r;
  double p = fma(r, -0.125, 1.0 / 7.0);
  p = fma(r, p, -1.0 / 6.0);
  p = fma(r, p, 0.2);
  p = fma(r, p, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  return fma(r2, p, lo) + hi;
}

/* ---- portable sincos: 0 <= x <= 2 pi (64-point table + short polynomials) ----------------- */
static inline void orc_pm_sincos(double x, double *sn, double *cs) {
  const int i = (int)(x * jb_sc_inv_step + 0.5);
  const double fi = (double)i;
  const double r = (x - fi * jb_sc_step_hi) - fi * jb_sc_step_lo;
  const double si = jb_sc_tab[i][0], ci = jb_sc_tab[i][1];
  const double r2 = r * r;
  const double sr = fma(r * r2,
                        fma(r2, fma(r2, fma(r2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0),
                            -1.0 / 6.0),
                        r);
  const double cm1 =
      r2 * fma(r2, fma(r2, fma(r2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5);
  *sn = si + fma(si, cm1, ci * sr);
  *cs = ci + fma(ci, cm1, -(si * sr));
}

/* ---- portable sin / cos of 2 pi u, u a uniform in (0,1): the grid point (256 per turn) is
 * found on u itself (i = (int)(256 u + 0.5), u - i/256 exact), the remainder is multiplied by
 * 2 pi once; angle addition with degree-5 / degree-6 polynomials ------------------------------ */
static inline void orc_pm_sincos2pi(double u, double *sn, double *cs) {
  const int i = (int)fma(u, 256.0, 0.5);
  const double r = fma((double)i, -0.00390625, u) * 6.283185307179586476925286766559;
  const double si = jb_sc2_tab[i][0], ci = jb_sc2_tab[i][1];
  const double r2 = r * r; /* |r| <= pi / 256: r^7 / 5040 < 1e-17, r^8 / 40320 < 2e-20 */
  const double sr = fma(r * r2, fma(r2, 1.0 / 120.0, -1.0 / 6.0), r);
  const double cm1 = r2 * fma(r2, fma(r2, -1.0 / 720.0, 1.0 / 24.0), -0.5);
  *sn = fma(si, cm1, fma(ci, sr, si));
  *cs = fma(ci, cm1, fma(-si, sr, ci));
}

/* ---- portable acos: |x| <= 1 ------------------------------------------------------------ */
static inline double orc_pm_acos_R(double z) {
  static const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                      pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                      pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                      qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                      qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const double p = z * fma(z, fma(z, fma(z, fma(z, fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
  const double q = fma(z, fma(z, fma(z, fma(z, qS4, qS3), qS2), qS1), 1.0);
  return p / q;
}
static inline double orc_pm_acos(double x) {
  static const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                      pi = 3.14159265358979311600e+00;
  const double ax = fabs(x);
  if (ax >= 1.0) return x > 0.0 ? 0.0 : pi; /* fdlibm: acos(1) = 0, acos(-1) = pi */
  if (ax < 0.5) {
    const double r = orc_pm_acos_R(x * x);
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (x < 0.0) {
    const double z = (1.0 + x) * 0.5;
    const double s = sqrt(z);
    const double w = orc_pm_acos_R(z) * s - pio2_lo;
    return pi - 2.0 * (s + w);
  } else {
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt(z);
    const double df = orc_u2d(orc_d2u(s) & 0xffffffff00000000ull);
    const double c = (z - df * df) / (s + df);
    const double w = orc_pm_acos_R(z) * s + c;
    return 2.0 * (df + w);
  }
}

/* 1 - exp(-x), x >= 0: the sequence of jb_math.hpp m_one_minus_exp_neg */
static inline double orc_pm_one_minus_exp_neg(double x) {
  static const double c[17] = {0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p-1,
                               0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7,
                               0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16,
                               0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
                               0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33, 0x1.93974a8c07c9dp-37,
                               0x1.ae7f3e733b81fp-41, 0x1.ae7f3e733b81fp-45};
  if (!(x < 40.0)) return 1.0;
  if (x < 0.25) {
    const double z = -x;
    double q = c[14];
    for (int n = 13; n >= 1; --n) q = fma(q, z, c[n]);
    return x * q;
  }
  const double kf = floor(fma(x, 0x1.71547652b82fep+0, 0.5));
  double r = fma(kf, -0x1.62e42fee00000p-1, x);
  r = fma(kf, -0x1.a39ef35793c76p-33, r);
  const double z = -r;
  double p = c[16];
  for (int n = 15; n >= 0; --n) p = fma(p, z, c[n]);
  const double scale = orc_u2d((uint64_t)(1023 - (int)kf) << 52);
  return 1.0 - p * scale;
}

/* ---- dispatch ---------------------------------------------------------------------------- */
static inline double orc_log(double x) {
  return orc_math_mode == ORC_MATH_LIBM ? log(x) : orc_pm_log(x);
}
static inline void orc_sincos(double x, double *sn, double *cs) {
  if (orc_math_mode == ORC_MATH_LIBM) {
    *sn = sin(x);
    *cs = cos(x);
  } else {
    orc_pm_sincos(x, sn, cs);
  }
}
/* sin, cos of phi = 2 pi u (reference: `phi = 2.0 * M_PI * drand(); cos(phi), sin(phi)`,
 * scattering.hpp:24-27, transport_utils.hpp:33-38,271-275, sourcing.cpp:181-185) */
static inline void orc_sincos2pi(double u, double *sn, double *cs) {
  if (orc_math_mode == ORC_MATH_LIBM) {
    const double phi = (2.0 * M_PI) * u;
    *sn = sin(phi);
    *cs = cos(phi);
  } else {
    orc_pm_sincos2pi(u, sn, cs);
  }
}
static inline double orc_acos(double x) {
  return orc_math_mode == ORC_MATH_LIBM ? acos(x) : orc_pm_acos(x);
}
static inline double orc_one_minus_exp_neg(double x) {
  return orc_math_mode == ORC_MATH_LIBM ? -expm1(-x) : orc_pm_one_minus_exp_neg(x);
}
/* T^4: std::pow(T, 4.0) in the reference (sourcing.cpp:93); portable flavour = (T*T)*(T*T) */
static inline double orc_pow4(double t) {
  if (orc_math_mode == ORC_MATH_LIBM) return pow(t, 4.0);
  const double t2 = t * t;
  return t2 * t2;
}

#endif /* ORC_MATH_H_ */
