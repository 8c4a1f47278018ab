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


// fma through the 3-address VOP3 form.  Left to itself the compiler turns a Horner step
// p = fma(r, p, C) with a loop-invariant constant C into "v_mov_b64 acc, C; v_fmac_f64 acc, r, p"
// (two instructions; the kernels are VALU-issue bound).  Same operation, same rounding.
__device__ __forceinline__ double m_fma(double a, double b, double c) {
#ifdef JB_NO_ASM_FMA
  return fma(a, b, c);
#else
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#endif
}

// ... with the addend a loop-invariant constant held in a SCALAR register pair (a VOP3 instruction
// reads one): the "v" form above parks every polynomial coefficient in two vector registers for
// the life of the kernel; k_imc_cell, whose mesh view no longer occupies the scalar file, can afford
// the scalar ones and needs the vector ones for a fourth wave per SIMD
__device__ __forceinline__ double m_fma_s(double a, double b, double c_uniform) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
template <bool SC>
__device__ __forceinline__ double m_fma_k(double a, double b, double c_const) {
  if constexpr (SC) return m_fma_s(a, b, c_const);
  else return m_fma(a, b, c_const);
}

// fma(-a, b, c) with the negation as a source modifier (an asm operand `-a` would cost a v_xor)
__device__ __forceinline__ double m_fnma(double a, double b, double c) {
#ifdef JB_NO_ASM_FMA
  return fma(-a, b, c);
#else
  double d;
  asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#endif
}

// min of two non-NaN numbers / minNum when one is NaN: one v_min_f64 (std::min written as
// `(b < a) ? b : a` costs a compare and two selects).  Inputs are results of arithmetic, i.e.
// already canonical, which is all the IEEE-mode v_min_f64 asks for.
__device__ __forceinline__ double m_min(double a, double b) {
  double d;
  asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// cell index inside one block's [nk][nj][ni] array.  jb_mesh_create checks ni < 2^23 and
// nj nk < 2^23, so both products are 24-bit multiplications (v_mad_i32_i24, full rate; a
// general 32-bit multiply-add is a quarter-rate 64-bit one on gfx950).
// (through inline asm: left to itself the compiler forms k nj + j with v_mad_u64_u32, a
// quarter-rate instruction, in the tracking loops)
__device__ __forceinline__ int mad24(int a, int b_uniform, int c) {
#ifdef JB_NO_MAD24_ASM
  return __mul24(a, b_uniform) + c;
#else
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
  return d;
#endif
}
__shared__ double lds_log_tab[JB_LOG_N][4];  // {1/c, log c hi, log c lo, -}: 32-byte rows, one address serves both reads
__shared__ double lds_log2_tab[JB_LOG_N][2];  // {1/c, log c as one double}: the lean logarithm's row
__shared__ double lds_logl_tab[JB_LOGL_N][2]; // the same with 1024 rows (16 KB): m_log_lean<SC, true>, k_imc_cell
__shared__ double lds_sc_tab[JB_SC_N + 1][2];
__shared__ double lds_sc2_tab[JB_SC2_N + 1][2];

// Copies the tables into this workgroup's LDS (11.2 KB; a kernel that names the ones it reads
// -- logarithm, lean logarithm, sincos, sincos of 2 pi u -- gets only those allocated); ends with a
// barrier.
template <bool LOG = true, bool LOG2 = true, bool SC = true, bool SC2 = true, bool LOGL = false>
__device__ __forceinline__ void load_math_tables() {
  if constexpr (LOGL)
    for (int q = threadIdx.x; q < JB_LOGL_N * 2; q += blockDim.x)
      (&lds_logl_tab[0][0])[q] = (&jb_logl_tab[0][0])[q];
  if constexpr (LOG)
    for (int q = threadIdx.x; q < JB_LOG_N * 3; q += blockDim.x)
      lds_log_tab[q / 3][q % 3] = (&jb_log_tab[0][0])[q];
  if constexpr (LOG2)
    for (int q = threadIdx.x; q < JB_LOG_N; q += blockDim.x) {
      lds_log2_tab[q][0] = jb_log_tab[q][0];
      lds_log2_tab[q][1] = jb_log_tab[q][1] + jb_log_tab[q][2];
    }
  if constexpr (SC)
    for (int q = threadIdx.x; q < (JB_SC_N + 1) * 2; q += blockDim.x)
      (&lds_sc_tab[0][0])[q] = (&jb_sc_tab[0][0])[q];
  if constexpr (SC2)
    for (int q = threadIdx.x; q < (JB_SC2_N + 1) * 2; q += blockDim.x)
      (&lds_sc2_tab[0][0])[q] = (&jb_sc2_tab[0][0])[q];
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
// Lean arithmetic (the opt-out-able default of the gray IMC kernels):
// a reciprocal with ONE Newton step (relative error <= 2^-48: a quotient formed with it is up to
// ~20 ulp from the correctly rounded one, measured 19; tests/test_gpu_lean.py) ...
__device__ __forceinline__ double m_rcp_once(double b) {
  const double y = __builtin_amdgcn_rcp(b);
  return fma(y, fma(-b, y, 1.0), y);
}
__device__ __forceinline__ double m_div_r(double a, double b, double y) {  // y = m_rcp_refined(b)
  const double q = a * y;
  const double r = fma(-b, q, a);
  return fma(r, y, q);
}
__device__ __forceinline__ double m_div(double a, double b) { return m_div_r(a, b, m_rcp_refined(b)); }

__device__ __forceinline__ double m_log(double x) {  // x positive, finite, normal
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  // tmp = bits(x) - JB_LOG_OFF; i = (tmp >> 45) & 127; k = (int64)tmp >> 52;
  // z = bits(x) - (tmp & 0xfff0000000000000): the low word of JB_LOG_OFF is zero, so all of it
  // happens in the high word
  static_assert((JB_LOG_OFF & 0xffffffffull) == 0ull, "m_log works on the high word only");
  const uint32_t hx = (uint32_t)__double2hiint(x);
  const uint32_t th = hx - (uint32_t)(JB_LOG_OFF >> 32);
  const int i = (int)((th >> 13) & (JB_LOG_N - 1));
  const int k = (int)th >> 20;
  const double z = __hiloint2double((int)(hx - (th & 0xfff00000u)), __double2loint(x));
  const double invc = lds_log_tab[i][0], lc_hi = lds_log_tab[i][1], lc_lo = lds_log_tab[i][2];
  const double r = fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = fma(kd, ln2_hi, lc_hi);  // exact: both terms are short
  const double hi = w + r;
  const double lo = ((w - hi) + r) + fma(kd, ln2_lo, lc_lo);
  const double r2 = r * r;
  double p = m_fma(r, -0.125, 1.0 / 7.0);
  p = m_fma(r, p, -1.0 / 6.0);
  p = m_fma(r, p, 0.2);
  p = m_fma(r, p, -0.25);
  p = m_fma(r, p, 1.0 / 3.0);
  p = m_fma(r, p, -0.5);
  return fma(r2, p, lo) + hi;
}

// ... the logarithm without the compensated low-order sum: k ln2 + log c in one fused
// multiply-add on the rounded constants, plus r + r^2 P(r), added in plain double (<= 3 ulp of the
// result for the arguments in (0, 1) the kernels feed it, measured <= 2; next to x = 1 the table
// row is {1, 0} and the result is r + r^2 P(r) itself) ...
// WIDE: 1024 table rows instead of 128 -- |r| <= 2^-11 (2^-10 in the row that holds 1.0, where the
// result is r + r^2 P(r) itself), so that the series ends with r^5 / 5 (the next term is below 5e-18 of
// the result): two Horner steps fewer for 14 KB more LDS.  The kernel that has the room asks for it.
template <bool SC = false, bool WIDE = false>
__device__ __forceinline__ double m_log_lean(double x) {
  constexpr double ln2 = 6.93147180559945286227e-01;
  const uint32_t hx = (uint32_t)__double2hiint(x);
  const uint32_t th = hx - (uint32_t)(JB_LOG_OFF >> 32);
  if constexpr (WIDE) {
    const int i = (int)((th >> 10) & (JB_LOGL_N - 1));
    const int k = (int)th >> 20;
    const double z = __hiloint2double((int)(hx - (th & 0xfff00000u)), __double2loint(x));
    const double invc = lds_logl_tab[i][0], lc = lds_logl_tab[i][1];
    const double r = fma(z, invc, -1.0);
    const double w = fma((double)k, ln2, lc);
    const double r2 = r * r;
    double p = m_fma_k<SC>(r, 0.2, -0.25);
    p = m_fma_k<SC>(r, p, 1.0 / 3.0);
    p = m_fma_k<SC>(r, p, -0.5);
    return w + fma(r2, p, r);
  }
  const int i = (int)((th >> 13) & (JB_LOG_N - 1));
  const int k = (int)th >> 20;
  const double z = __hiloint2double((int)(hx - (th & 0xfff00000u)), __double2loint(x));
  const double invc = lds_log2_tab[i][0], lc = lds_log2_tab[i][1];
  const double r = fma(z, invc, -1.0);
  const double w = fma((double)k, ln2, lc);
  const double r2 = r * r;
  // (|r| <= 5.5e-3: the r^8 / 8 term of the series is below 2e-17 of the result, 0.2 ulp)
  double p = m_fma_k<SC>(r, 1.0 / 7.0, -1.0 / 6.0);
  p = m_fma_k<SC>(r, p, 0.2);
  p = m_fma_k<SC>(r, p, -0.25);
  p = m_fma_k<SC>(r, p, 1.0 / 3.0);
  p = m_fma_k<SC>(r, p, -0.5);
  return w + fma(r2, p, r);
}

// ... and the square root of 1 - mu^2 (in [2^-52, 1]) from the hardware's reciprocal square root
// with one coupled refinement step and one residual correction (m_sqrt: two, and a refined
// half-reciprocal); <= 2 ulp, measured in tests/test_gpu_lean.py.
__device__ __forceinline__ double m_sqrt_lean(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  const double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return fma(fma(-g, g, x), h, g);
}

__device__ __forceinline__ void m_sincos(double x, double &sn, double &cs) {  // 0 <= x <= 2 pi
  const int i = (int)(x * jb_sc_inv_step + 0.5);
  const double fi = (double)i;
  const double r = (x - fi * jb_sc_step_hi) - fi * jb_sc_step_lo;
  const double si = lds_sc_tab[i][0], ci = lds_sc_tab[i][1];
  const double r2 = r * r;
  const double sr = fma(r * r2,
                        m_fma(r2, m_fma(r2, m_fma(r2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0),
                              -1.0 / 6.0),
                        r);
  const double cm1 =
      r2 * m_fma(r2, m_fma(r2, m_fma(r2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5);
  sn = si + fma(si, cm1, ci * sr);
  cs = ci + fma(ci, cm1, -(si * sr));
}

// sin and cos of 2 pi u for a uniform u in (0,1) (the azimuth of every direction sample:
// scattering.hpp:24-27, transport_utils.hpp:33-38,271-275, sourcing.cpp:181): the grid point
// (256 per turn) is found on u itself -- i = (int)(256 u + 0.5), u - i/256 is exact -- so the
// reduction costs one multiplication by 2 pi of a remainder |u - i/256| <= 1/512 instead of an
// extended-precision subtraction, and |r| <= pi/256 needs only r - r^3/6 + r^5/120 and
// -r^2/2 + r^4/24 - r^6/720 (next terms < 1e-17).  Absolute error <= 1.7e-16.
constexpr double kTwoPiM = 6.283185307179586476925286766559;
template <bool SC = false>
__device__ __forceinline__ void m_sincos2pi(double u, double &sn, double &cs) {
  const int i = (int)fma(u, 256.0, 0.5);
  const double r = fma((double)i, -0.00390625, u) * kTwoPiM;
  const double si = lds_sc2_tab[i][0], ci = lds_sc2_tab[i][1];
  const double r2 = r * r;
  const double sr = m_fma(r * r2, m_fma_k<SC>(r2, 1.0 / 120.0, -1.0 / 6.0), r);
  const double cm1 = r2 * m_fma_k<SC>(r2, m_fma_k<SC>(r2, -1.0 / 720.0, 1.0 / 24.0), -0.5);
  sn = m_fma(si, cm1, m_fma(ci, sr, si));
  cs = m_fma(ci, cm1, m_fnma(si, sr, ci));
}

// 1 - exp(-x) for x >= 0 (the stimulated-emission factor of a thermal absorption coefficient),
// fully specified like the functions above: Taylor coefficients 1/n! (correctly rounded), Horner
// steps as fma, argument reduction x = k ln2 + r with k ln2_hi exact.  ~1 ulp of 1 - e^-x.
__device__ __forceinline__ double m_one_minus_exp_neg(double x) {
  constexpr double c[17] = {0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p-1,
                            0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7,
                            0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16,
                            0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
                            0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33, 0x1.93974a8c07c9dp-37,
                            0x1.ae7f3e733b81fp-41, 0x1.ae7f3e733b81fp-45};
  if (!(x < 40.0)) return 1.0;  // e^-40 < 2^-54
  if (x < 0.25) {               // x (1 - x/2! + x^2/3! - ...): no cancellation
    const double z = -x;
    double q = c[14];
    for (int n = 13; n >= 1; --n) q = fma(q, z, c[n]);
    return x * q;
  }
  const double kf = floor(fma(x, 0x1.71547652b82fep+0, 0.5));
  double r = fma(kf, -0x1.62e42fee00000p-1, x);
  r = fma(kf, -0x1.a39ef35793c76p-33, r);
  const double z = -r;  // |r| <= ln2 / 2 (+ rounding of k)
  double p = c[16];
  for (int n = 15; n >= 0; --n) p = fma(p, z, c[n]);
  const double scale = __longlong_as_double((long long)(1023 - (int)kf) << 52);  // 2^-k, k <= 58
  return 1.0 - p * scale;
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
