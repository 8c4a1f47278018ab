// jb_math.hpp -- transcendentals of the history loop as fully specified IEEE-754 sequences.
//
// The reference calls std::log / std::sin / std::cos / std::acos / std::pow
// (transport_utils.hpp:31-38,118-119,185,270-275; scattering.hpp:23-28; planck.hpp:30-49;
// sourcing.cpp:93,180-185).  Device libm and host libm differ in the last bit, which after ~1e3
// events per history flips branch decisions and makes CPU and GPU histories diverge.  These
// versions use only + - * / sqrt fma and integer bit manipulation, following the published fdlibm
// algorithms (e_log.c, k_sin.c, k_cos.c with the first Cody-Waite step of e_rem_pio2.c,
// e_acos.c), so a CPU that evaluates the same sequence gets the same bits.  Accuracy: <= 1 ulp
// against correctly rounded results on the argument ranges used (tests/test_oracle_math.py).
//
// Compile with -ffp-contract=off: every fused multiply-add below is written as fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jb {

__device__ __forceinline__ double m_log(double x) {  // x positive, finite, normal
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                   Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                   Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                   Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  const uint64_t ix = (uint64_t)__double_as_longlong(x);
  int k = (int)(ix >> 52) - 1023;
  const uint32_t hx = (uint32_t)(ix >> 32) & 0x000fffffu;
  const uint32_t i = (hx + 0x95f64u) & 0x100000u;
  const uint64_t mbits = (ix & 0x000fffffffffffffull) | ((uint64_t)(0x3ff00000u ^ i) << 32);
  k += (int)(i >> 20);
  const double f = __longlong_as_double((long long)mbits) - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __forceinline__ void m_sincos(double x, double &sn, double &cs) {  // 0 <= x <~ 7
  constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                   pio2_1t = 6.07710050650619224932e-11;
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                   S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                   S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                   C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                   C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const int n = (int)(x * invpio2 + 0.5);
  const double fn = (double)n;
  const double r = x - fn * pio2_1;
  const double wt = fn * pio2_1t;
  const double y0 = r - wt;
  const double y1 = (r - y0) - wt;
  const double z = y0 * y0;
  const double v = z * y0;
  const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
  const double w = z * z;
  const double rc = z * fma(z, fma(z, C3, C2), C1) + (w * w) * fma(z, fma(z, C6, C5), C4);
  const double hz = 0.5 * z;
  const double w1 = 1.0 - hz;
  const double kc = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
  const int q = n & 3;
  const double a = (q & 1) ? kc : ks;
  const double b = (q & 1) ? ks : kc;
  sn = (q & 2) ? -a : a;
  cs = (q == 1 || q == 2) ? -b : b;
}

__device__ __forceinline__ double m_acos_R(double z) {
  constexpr double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                   pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                   pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                   qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                   qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const double p = z * fma(z, fma(z, fma(z, fma(z, fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
  const double q = fma(z, fma(z, fma(z, fma(z, qS4, qS3), qS2), qS1), 1.0);
  return p / q;
}

__device__ __forceinline__ double m_acos(double x) {  // |x| <= 1
  constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                   pi = 3.14159265358979311600e+00;
  const double ax = fabs(x);
  if (ax >= 1.0) return x > 0.0 ? 0.0 : pi;  // fdlibm: acos(1) = 0, acos(-1) = pi
  if (ax < 0.5) {
    const double r = m_acos_R(x * x);
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (x < 0.0) {
    const double z = (1.0 + x) * 0.5;
    const double s = sqrt(z);
    const double w = m_acos_R(z) * s - pio2_lo;
    return pi - 2.0 * (s + w);
  } else {
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt(z);
    const double df =
        __longlong_as_double((long long)((uint64_t)__double_as_longlong(s) & 0xffffffff00000000ull));
    const double c = (z - df * df) / (s + df);
    const double w = m_acos_R(z) * s + c;
    return 2.0 * (df + w);
  }
}

}  // namespace jb
