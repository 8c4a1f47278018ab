// jb_math.hpp -- transcendentals of the history loop as fully specified IEEE-754 sequences.
//
// The reference calls std::log / std::sin / std::cos / std::acos / std::pow
// (transport_utils.hpp:31-38,118-119,185,270-275; scattering.hpp:23-28; planck.hpp:30-49;
// sourcing.cpp:93,180-185).  Device libm and host libm differ in the last bit, which after ~1e3
// events per history flips branch decisions and makes CPU and GPU histories diverge.  These
// versions use only + - * / sqrt fma, integer bit manipulation and two small lookup tables
// (jb_tables.hpp, generated with 70-digit arithmetic by tools/gen_math_tables.py), so a CPU that
// evaluates the same sequence gets the same bits (the CPU oracle of the test suite does).
//
//   log     x = 2^k z, z in [0.707, 1.414) split in 128 intervals; r = z / c - 1 by one fma with
//           the tabulated 1/c (c = 1 exactly in the interval around 1), |r| <= 2^-7;
//           log x = k ln2 + log c + log1p(r), degree-8 Taylor polynomial, hi/lo accumulation.
//           No division.  <= 1 ulp.  133 SIMD-cycles per wave call on MI355X (fdlibm-style with a
//           division: 199; the device libm's log: 378).
//   sincos  phi = i (2 pi / 64) + r, |r| <= pi / 64; tabulated sin / cos of the grid points,
//           degree-9 / degree-8 polynomials for sin r and cos r - 1, angle-addition.  Absolute
//           error <= 1.2e-16.  129 cycles (fdlibm kernels: 276; device libm: 309).
//   acos    fdlibm e_acos.c (sourcing only).
//
// The tables live in LDS: every kernel that calls m_log / m_sincos runs load_math_tables() first.
// Compile with -ffp-contract=off: every fused multiply-add below is written as fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jb_tables.hpp"

namespace jb {

__shared__ double lds_log_tab[JB_LOG_N][3];
__shared__ double lds_sc_tab[JB_SC_N + 1][2];

// Copies the two tables into this workgroup's LDS (4.1 KB); ends with a barrier.
__device__ __forceinline__ void load_math_tables() {
  for (int q = threadIdx.x; q < JB_LOG_N * 3; q += blockDim.x)
    (&lds_log_tab[0][0])[q] = (&jb_log_tab[0][0])[q];
  for (int q = threadIdx.x; q < (JB_SC_N + 1) * 2; q += blockDim.x)
    (&lds_sc_tab[0][0])[q] = (&jb_sc_tab[0][0])[q];
  __syncthreads();
}

// Correctly rounded sqrt for 2^-700 < x < 2^700 (here: 1 - mu^2 and xi, both in (0, 1]): the
// refinement the compiler emits for sqrt(double) (v_rsq_f64 seed, Goldschmidt step, two fused
// residual corrections), without its scaling of tiny / huge inputs and its 0 / inf special cases.
// Bit-identical to IEEE sqrt on that range (tests/test_gpu_parity.py::test_device_math_bit_exact).
__device__ __forceinline__ double m_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d0 = fma(-g, g, x);
  g = fma(d0, h, g);
  const double d1 = fma(-g, g, x);
  return fma(d1, h, g);
}

// IEEE division a / b without the range handling: the sequence the compiler emits for `a / b`
// (v_rcp_f64 seed, two Newton steps on the reciprocal, quotient, one fused residual correction)
// minus v_div_scale / v_div_fmas / v_div_fixup, which only act when an operand or the quotient
// is denormal, the exponents are more than 2^768 apart, or b is 0 / inf / NaN.  Distances,
// speeds and opacities are nowhere near that, so the result is the correctly rounded quotient,
// bit-identical to `/` (tests/test_gpu_parity.py::test_device_math_bit_exact).  b = 0 gives NaN,
// not inf: callers that can see a zero divisor discard the quotient by a select.
// With a divisor that is constant over a kernel the refined reciprocal is computed once.
__device__ __forceinline__ double m_rcp_refined(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = fma(-b, y, 1.0);
  y = fma(y, e, y);
  e = fma(-b, y, 1.0);
  return fma(y, e, y);
}
__device__ __forceinline__ double m_div_r(double a, double b, double y) {  // y = m_rcp_refined(b)
  const double q = a * y;
  const double r = fma(-b, q, a);
  return fma(r, y, q);
}
__device__ __forceinline__ double m_div(double a, double b) { return m_div_r(a, b, m_rcp_refined(b)); }

__device__ __forceinline__ double m_log(double x) {  // x positive, finite, normal
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const uint64_t ix = (uint64_t)__double_as_longlong(x);
  const uint64_t tmp = ix - JB_LOG_OFF;
  const int i = (int)((tmp >> 45) & (JB_LOG_N - 1));
  const int k = (int)(uint32_t)(tmp >> 32) >> 20;  // = (int64)tmp >> 52, from the high word
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  const double z = __longlong_as_double((long long)iz);
  const double invc = lds_log_tab[i][0], lc_hi = lds_log_tab[i][1], lc_lo = lds_log_tab[i][2];
  const double r = fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = fma(kd, ln2_hi, lc_hi);  // exact: both terms are short
  const double hi = w + r;
  const double lo = ((w - hi) + r) + fma(kd, ln2_lo, lc_lo);
  const double r2 = r * r;
  double p = fma(r, -0.125, 1.0 / 7.0);
  p = fma(r, p, -1.0 / 6.0);
  p = fma(r, p, 0.2);
  p = fma(r, p, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  return fma(r2, p, lo) + hi;
}

__device__ __forceinline__ void m_sincos(double x, double &sn, double &cs) {  // 0 <= x <= 2 pi
  const int i = (int)(x * jb_sc_inv_step + 0.5);
  const double fi = (double)i;
  const double r = (x - fi * jb_sc_step_hi) - fi * jb_sc_step_lo;
  const double si = lds_sc_tab[i][0], ci = lds_sc_tab[i][1];
  const double r2 = r * r;
  const double sr = fma(r * r2,
                        fma(r2, fma(r2, fma(r2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0),
                            -1.0 / 6.0),
                        r);
  const double cm1 =
      r2 * fma(r2, fma(r2, fma(r2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5);
  sn = si + fma(si, cm1, ci * sr);
  cs = ci + fma(ci, cm1, -(si * sr));
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
