// jb_physics.hpp -- per-event device functions of the IMC / DDMC history loop.
//
// What each function computes, the order of its floating-point operations and the order of its
// random draws follow the reference (file:line given per function); they are templated on the
// dimensionality so that the multi_d / three_d gates of the reference fold at compile time, and
// on the random source (PhiloxRng in the product kernels, TapeRng in the debug entry points).
#pragma once

#include <float.h>
#include <hip/hip_runtime.h>

#include "jb_math.hpp"
#include "jb_rng.hpp"

namespace jb {

// parthenon::robust::EPS() (un-vendored; 10 * machine epsilon assumed)
constexpr double kEps = 10.0 * DBL_EPSILON;
// reference transport_utils.hpp:24-25
constexpr double kEpsImc = 1.0e6 * kEps;
constexpr double kEpsDdmc = 1.0e8 * kEps;
// reference transport_utils.hpp:282, jaybenne.cpp:326 (Habetler & Matkowski 1975)
constexpr double kLamExt = 0.7104;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }  // std::min

// reference jaybenne_utils.hpp:44-49
__device__ __forceinline__ bool fuzzy_equal(double a, double b, double c, double eps) {
  return fabs(a - b) < c * eps;
}

// reference transport_utils.hpp:27-39 -- the arithmetic on its two uniforms
__device__ __forceinline__ void face_iso_dir(double vv, double xi1, double xi2, double &v1,
                                             double &v2, double &v3) {
  const double mu = m_sqrt(xi1);
  const double om = 1.0 - mu * mu;  // exactly 0 when mu rounds to 1, else >= 2^-53
  const double nu = (om > 0.0) ? m_sqrt(om > 0.0 ? om : 1.0) : 0.0;
  double sn, cs;
  m_sincos2pi(xi2, sn, cs);
  v1 = vv * mu;
  v2 = vv * nu * cs;
  v3 = vv * nu * sn;
}

// reference transport_utils.hpp:27-39 -- 2 draws
template <class Rng>
__device__ __forceinline__ void sample_face_iso_dir(double vv, Rng &rng, double &v1, double &v2,
                                                    double &v3) {
  const double xi1 = rng.drand();
  const double xi2 = rng.drand();
  face_iso_dir(vv, xi1, xi2, v1, v2, v3);
}

// reference scattering.hpp:21-29 -- 2 draws
template <class Rng>
__device__ __forceinline__ void scatter(Rng &rng, double vv, double &vx, double &vy, double &vz) {
  double xi1, xi2;
  rng.drand2(xi1, xi2);
  const double mu = fma(2.0, xi1, -1.0);  // (2 xi is exact: same value as 2 xi - 1)
  const double st = m_sqrt(1.0 - mu * mu);  // |mu| <= 1 - 2^-52, so 1 - mu^2 >= 2^-52
  double sn, cs;
  m_sincos2pi(xi2, sn, cs);
  vx = vv * st * cs;
  vy = vv * st * sn;
  vz = vv * mu;
}

// reference planck.hpp:26-50 (Everett & Cashwell 1972) -- 5 draws.  `sb` really is the
// Stefan-Boltzmann constant there.
template <class Rng>
__device__ __forceinline__ double sample_planck_energy(Rng &rng, double sb, double temp) {
  constexpr double kPi = 3.14159265358979323846;
  const double xi0 = rng.drand();
  const double rhs = xi0 * ((kPi * kPi) * (kPi * kPi)) / 90.0;
  double ll = 1.0;
  for (int l = 1; l < 100; ++l) {
    double lhs = 0.0;
    for (int j = 1; j <= l; ++j) {
      const double dj = (double)j;
      lhs += 1.0 / ((dj * dj) * (dj * dj));
    }
    if (lhs >= rhs) {
      ll = (double)l;
      break;
    }
  }
  const double xi1 = rng.drand();
  const double xi2 = rng.drand();
  const double xi3 = rng.drand();
  const double xi4 = rng.drand();
  return -(1.0 / ll) * m_log(xi1 * xi2 * xi3 * xi4) * sb * temp;
}

// What one step reads and updates for one particle in one cell (kept in registers).
struct Step {
  double t_start, dt;
  double ff, aa, ss, vv, dx_push;
  double rvv;        // m_rcp_refined(vv): set by the caller (a per-context constant)
  double ffaa, sig;  // ff * aa and aa + ss (DDMC functions read these; set by the caller)
  double xl, yl, zl, xu, yu, zu;
  double Px_l, Py_l, Pz_l, Px_u, Py_u, Pz_u;
  double t, x, y, z, vx, vy, vz;
  int ip, jp, kp;
  bool is_absorbed, is_scattered, is_rejected;
  // Deferred direction of a DDMC leak (LAZY): channel 0..5 (x-, x+, y-, y+, z-, z+) or -1, and
  // the two uniforms sample_face_iso_dir drew for it.  vx, vy, vz are stale while pend >= 0.
  int pend;
  double pz1, pz2;
};

// reference transport_utils.hpp:115-117: mean free paths of a cell.  They depend on the cell
// only, so the tracking kernel keeps them while a particle stays in its cell (two FP64 divisions
// saved per scatter); same operations, same bits.
__device__ __forceinline__ void imc_cell_mfp(double ff, double aa, double ss, double &lam_abs,
                                             double &lam_sc) {
  const double rmin = DBL_MIN;
  lam_abs = 1.0 / (ff * aa + rmin);
  lam_sc = 1.0 / (ss + (1.0 - ff) * aa + rmin);
}

// reference transport_utils.hpp:118-159 -- the rest of one IMC tracking step, 2 draws
// NOABS: the material has no absorption opacity anywhere (opacity_model = none: sigma_a = 0, so
// lam_abs = 1 / DBL_MIN = 4.5e307).  Then dx_abs = -lam_abs ln(xi) >= 4.5e307 * 2^-53 = 5e291
// for every possible draw, never the smallest distance: the draw is consumed, its logarithm is
// not evaluated, and is_absorbed is false -- the same results as the general path, bit for bit.
template <int NDIM, bool NOABS = false, class Rng>
__device__ __forceinline__ void imc_step_core(Step &s, double lam_abs, double lam_sc, Rng &rng) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  double dx_abs = 0.0;
  if constexpr (NOABS) rng.skip();
  else dx_abs = -lam_abs * m_log(rng.drand());
  const double dx_sc = -lam_sc * m_log(rng.drand());
  const double dx_end = s.vv * ((s.t_start + s.dt) - s.t);
  double dx_push = dmin(s.dx_push, dx_end);
  // distance to the face the velocity points at (transport_utils.hpp:123-133): one division per
  // axis, selected operands instead of two divergent branches
  {
    const double d = m_div(s.vv * ((s.vx > 0.0 ? s.xu : s.xl) - s.x), s.vx);
    dx_push = (s.vx != 0.0) ? dmin(dx_push, d) : dx_push;
  }
  if (multi_d) {
    const double d = m_div(s.vv * ((s.vy > 0.0 ? s.yu : s.yl) - s.y), s.vy);
    dx_push = (s.vy != 0.0) ? dmin(dx_push, d) : dx_push;
  }
  if (three_d) {
    const double d = m_div(s.vv * ((s.vz > 0.0 ? s.zu : s.zl) - s.z), s.vz);
    dx_push = (s.vz != 0.0) ? dmin(dx_push, d) : dx_push;
  }

  s.is_absorbed = NOABS ? false : (dx_abs < dx_push) && (dx_abs < dx_sc);
  s.is_scattered = !s.is_absorbed && (dx_sc < dx_push);

  // (the refined reciprocal of the speed of light is computed once per context)
  const double dt_push = m_div_r(s.is_absorbed ? dx_abs : (s.is_scattered ? dx_sc : dx_push), s.vv,
                                 s.rvv);

  s.t += dt_push;
  s.x += s.vx * dt_push;
  s.y += (multi_d ? 1.0 : 0.0) * s.vy * dt_push;
  s.z += (three_d ? 1.0 : 0.0) * s.vz * dt_push;

  const double fdx = kEpsImc * (s.xu - s.xl);
  const double fdy = kEpsImc * (s.yu - s.yl);
  const double fdz = kEpsImc * (s.zu - s.zl);
  if (fabs(s.x - s.xl) < fdx) s.x = s.xl - fdx;
  if (fabs(s.x - s.xu) < fdx) s.x = s.xu + fdx;
  if (multi_d && fabs(s.y - s.yl) < fdy) s.y = s.yl - fdy;
  if (multi_d && fabs(s.y - s.yu) < fdy) s.y = s.yu + fdy;
  if (three_d && fabs(s.z - s.zl) < fdz) s.z = s.zl - fdz;
  if (three_d && fabs(s.z - s.zu) < fdz) s.z = s.zu + fdz;
}

// The same step as imc_step_core for the gray-opacity tracking kernels, fused with what the
// kernel does around it (transport.cpp:114-119,146): the caller hands in the cell's faces and the
// three nudge widths eps_imc (upper - lower), and gets the cell index of the new position back.
//  * The two face tests of an axis are made on the position before either nudge: the cell is
//    ~4e8 nudge widths across, so they cannot both hold, and a particle put eps below the lower
//    face is not within eps of the upper one -- the same result as the reference's sequence.
//  * Xtoijk after the step (transport.cpp:146): a particle is either >= eps inside its cell or
//    has just been put eps beyond a face, so floor((x - xmin) / dx) changes exactly when a nudge
//    fired, by one, in that direction.
__device__ __forceinline__ void idx_step(int &i, bool hi, bool lo) {
  i += (int)hi;
  i -= (int)lo;
}

struct ImcCell {
  double xl, xu, yl, yu, zl, zu;  // faces
  double fdx, fdy, fdz;           // eps_imc_offset * (upper - lower)
};
//  * LEAN: the same step in lean arithmetic -- distance to a face as (c (face - x)) times a
//    once-refined reciprocal of the velocity component, time step as distance times 1/c, position
//    update as one fused multiply-add per axis, logarithm without its compensated sum.  The
//    quotient is within 2^-48 (relative; ~20 ulp) of the exact variant's correctly rounded one,
//    the logarithm within 1 ulp, the fused update the more accurate of the two forms; a history parts ways with its exact twin only where
//    such a difference flips a comparison.  A zero velocity component still yields NaN in the
//    reciprocal's Newton step and v_min_f64 ignores it.
template <int NDIM, bool NOABS, bool LEAN = false, class Rng>
__device__ __forceinline__ void imc_step_fast(const ImcCell &c, double vv, double rvv, double t_end,
                                              double dx_push0, double lam_abs, double lam_sc,
                                              Rng &rng, double &t, double &x, double &y, double &z,
                                              double vx, double vy, double vz, int &ip, int &jp,
                                              int &kp, bool &is_absorbed, bool &is_scattered) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  double dx_abs = 0.0;
  if constexpr (NOABS) rng.skip();
  else dx_abs = -lam_abs * (LEAN ? m_log_lean(rng.drand()) : m_log(rng.drand()));
  const double dx_sc = -lam_sc * (LEAN ? m_log_lean(rng.drand()) : m_log(rng.drand()));
  const double dx_end = vv * (t_end - t);
  // std::min of two numbers is one v_min_f64.  A velocity component that is exactly zero makes
  // m_div return NaN (0 x inf in its first step), and minNum(x, NaN) = x: the distance stays as
  // it is, which is what the reference's `(v > 0) ? ... : (v < 0) ? ... : dx_push` says.
  double dx_push = m_min(dx_push0, dx_end);
  if constexpr (LEAN) {
    dx_push = m_min(dx_push, (vv * ((vx > 0.0 ? c.xu : c.xl) - x)) * m_rcp_once(vx));
    if (multi_d) dx_push = m_min(dx_push, (vv * ((vy > 0.0 ? c.yu : c.yl) - y)) * m_rcp_once(vy));
    if (three_d) dx_push = m_min(dx_push, (vv * ((vz > 0.0 ? c.zu : c.zl) - z)) * m_rcp_once(vz));
  } else {
    dx_push = m_min(dx_push, m_div(vv * ((vx > 0.0 ? c.xu : c.xl) - x), vx));
    if (multi_d) dx_push = m_min(dx_push, m_div(vv * ((vy > 0.0 ? c.yu : c.yl) - y), vy));
    if (three_d) dx_push = m_min(dx_push, m_div(vv * ((vz > 0.0 ? c.zu : c.zl) - z), vz));
  }
  is_absorbed = NOABS ? false : (dx_abs < dx_push) && (dx_abs < dx_sc);
  is_scattered = !is_absorbed && (dx_sc < dx_push);
  const double dx_move =
      NOABS ? m_min(dx_push, dx_sc) : (is_absorbed ? dx_abs : (is_scattered ? dx_sc : dx_push));
  if constexpr (LEAN) {
    const double dt_push = dx_move * rvv;
    t += dt_push;
    x = fma(vx, dt_push, x);
    if (multi_d) y = fma(vy, dt_push, y);
    if (three_d) z = fma(vz, dt_push, z);
  } else {
    const double dt_push = m_div_r(dx_move, vv, rvv);
    t += dt_push;
    x += vx * dt_push;
    y += (multi_d ? 1.0 : 0.0) * vy * dt_push;
    z += (three_d ? 1.0 : 0.0) * vz * dt_push;
  }
  {
    const bool lo = fabs(x - c.xl) < c.fdx, hi = fabs(x - c.xu) < c.fdx;
    x = lo ? c.xl - c.fdx : (hi ? c.xu + c.fdx : x);
    idx_step(ip, hi, lo);
  }
  if (multi_d) {
    const bool lo = fabs(y - c.yl) < c.fdy, hi = fabs(y - c.yu) < c.fdy;
    y = lo ? c.yl - c.fdy : (hi ? c.yu + c.fdy : y);
    idx_step(jp, hi, lo);
  }
  if (three_d) {
    const bool lo = fabs(z - c.zl) < c.fdz, hi = fabs(z - c.zu) < c.fdz;
    z = lo ? c.zl - c.fdz : (hi ? c.zu + c.fdz : z);
    idx_step(kp, hi, lo);
  }
}

// The lean tracking step of the gray IMC kernels (jb_set_arithmetic: the library's default), in
// "direction space": while a lane follows a photon it carries the unit direction omega = v / c
// and the distance left to census d_rem = c (t_end - t) instead of v and t (converted when the
// photon is loaded and written back).  The step of transport_utils.hpp:118-159 then reads
//   d_face = (face - x) / omega   (the reference: c (face - x) / v -- the same number),
//   x += omega d   (x += v (d / c)),   d_rem -= d   (t += d / c),
// with the quotient formed by a once-refined reciprocal (m_rcp_once: <= 2^-48 relative), the
// position update fused, and the logarithm without its compensated sum (m_log_lean: <= 2 ulp) --
// every operation within 4e-15 (relative) of the exact variant's.  One more difference, of another
// kind: per axis only the face the photon is MOVING TOWARDS is tested for the nudge of
// transport_utils.hpp:151-159.  The face behind it is at least eps_imc dx away -- the photon was
// put that far beyond it when it entered the cell -- unless the photon has moved less than the
// rounding error of its own position since (a flight of < 1e-16 cm: probability ~1e-13 per event)
// or was sourced within eps_imc dx = 2e-9 dx of a face; the exact variant tests both faces as the
// reference does.  A history parts ways with its exact twin only where such a difference flips a
// comparison (include/jaybenne_amd.h states the tolerance, tests/test_gpu_lean.py tests it).
// The block geometry the lean step reads: coordinate of cell index 0 and cell width per axis,
// nudge widths eps_imc dx.  EXACTG: power-of-two cell widths and block corners that are whole
// numbers of them (jb_mesh_exact_geometry), so that x0 + i dx is exact for every face i.
struct DirGeom {
  double x0[3], dx[3], fd[3];
};
// "moving up": the sign bit of the direction component is clear (+0.0 counts as up, -0.0 as down;
// either way a zero component reaches no face: see below)
__device__ __forceinline__ bool dir_up(double o) { return __double2hiint(o) >= 0; }
// the face the photon moves towards is face idx + up of its block (iu); EXACTG: one exact fma
template <bool EXACTG>
__device__ __forceinline__ double dir_face(const DirGeom &g, int d, int idx, bool up, int &iu) {
  iu = idx + (int)up;
  if constexpr (EXACTG) {
    return m_fma((double)iu, g.dx[d], g.x0[d]);
  } else {  // transport.cpp:114-119
    const double xcd = g.x0[d] + ((double)idx + 0.5) * g.dx[d];
    return up ? xcd + 0.5 * g.dx[d] : xcd - 0.5 * g.dx[d];
  }
}
// nudge (transport_utils.hpp:151-159) at the face ahead, and Xtoijk after it (transport.cpp:146):
// the cell index moves by one exactly when the nudge fires -- to iu when moving up, to iu - 1
// (= idx - 1) when moving down -- and is idx = iu - up otherwise: iu - (hit != up) in all cases
__device__ __forceinline__ void dir_nudge(double &x, int &idx, double o, double f, double fd, int iu,
                                          bool up) {
  const bool hit = fabs(x - f) < fd;
  // fd with the sign of o (one v_bfi_b32 on the high word, into a register of its own)
  int shi;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(shi) : "s"(0x7fffffff), "v"(__double2hiint(fd)), "v"(__double2hiint(o)));
  x = hit ? f + __hiloint2double(shi, __double2loint(fd)) : x;
  idx = iu - (int)(hit != up);
}
// ... with the nudge width carried from pass to pass WITH the sign it last had: only the sign bit
// is rewritten (v_bfi_b32 on the high word, in place), so no pass has to rebuild the register
// pair from the unsigned width (one v_mov_b32 per axis); the comparison reads |sfd| (a source
// modifier, free)
__device__ __forceinline__ void dir_nudge_carried(double &x, int &idx, double o, double f, double &sfd,
                                                  int iu, bool up) {
  int shi;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(shi) : "s"(0x7fffffff), "v"(__double2hiint(sfd)), "v"(__double2hiint(o)));
  sfd = __hiloint2double(shi, __double2loint(sfd));
  const bool hit = fabs(x - f) < fabs(sfd);
  x = hit ? f + sfd : x;
  idx = iu - (int)(hit != up);
}
// ... and, BOTH, at the face behind as well -- pushed outwards through it as the reference does
// whatever the direction (transport_utils.hpp:151-159).  The hybrid kernel on a general geometry
// needs it: a DDMC leak across a block face leaves the photon with ZERO velocity at the centre of
// the coarse cell it left -- a face of the fine cells -- and SampleDDMCBlockFace resamples it only if
// its fuzzy test (tolerance 2e-16 dx) recognises the position, which on cell widths that are not
// powers of two it does not always do; the photon then sits on that face until it scatters.
__device__ __forceinline__ void dir_nudge_behind(double &x, int &idx, double fb, double fd, bool up) {
  const bool hit = fabs(x - fb) < fd;
  x = hit ? (up ? fb - fd : fb + fd) : x;
  idx += hit ? (up ? -1 : 1) : 0;
}
template <int NDIM, bool NOABS, bool EXACTG, bool BOTH = false, class Rng>
__device__ __forceinline__ void imc_step_dir(DirGeom &g, double dx_push0, double lam_abs,
                                             double lam_sc, Rng &rng, double &d_rem, double &x,
                                             double &y, double &z, double ox, double oy, double oz,
                                             int &ip, int &jp, int &kp, bool &is_absorbed,
                                             bool &is_scattered) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  double dx_abs = 0.0;
  if constexpr (NOABS) rng.skip();
  else dx_abs = -lam_abs * m_log_lean(rng.drand());
  const double dx_sc = -lam_sc * m_log_lean(rng.drand());
  double dx_push = m_min(dx_push0, d_rem);
  // (a direction component that is exactly zero: 0 x inf = NaN in the reciprocal's Newton step,
  // and minNum ignores it -- the reference's third branch, as in imc_step_fast)
  const bool upx = dir_up(ox), upy = dir_up(oy), upz = dir_up(oz);
  int iux = 0, iuy = 0, iuz = 0;
  const double fx = dir_face<EXACTG>(g, 0, ip, upx, iux);
  const double fy = multi_d ? dir_face<EXACTG>(g, 1, jp, upy, iuy) : 0.0;
  const double fz = three_d ? dir_face<EXACTG>(g, 2, kp, upz, iuz) : 0.0;
#ifndef JB_NO_PRODUCT_RCP
  if constexpr (three_d) {
    // The three reciprocals from ONE hardware reciprocal, of the product of the components: 1 / ox =
    // (oy oz) / (ox oy oz) etc. -- one quarter-rate instruction (16 cycles) instead of three, at
    // 10 instructions instead of 9; each quotient within ~4e-15 of the exact one, as before.  A
    // component that is zero (or a product that underflows) takes the three separate reciprocals,
    // whose NaN for a zero component is what minNum then ignores (see above).  (In 2-D the same
    // trick, one reciprocal instead of two, was measured slower: +0.9 % on C4, +3 % on C5.)
    const double pxy = ox * oy, q = pxy * oz;
    double rx, ry, rz;
    if (fabs(q) > 1.0e-250) {
      const double r = m_rcp_once(q);
      const double roz = r * oz;
      rz = r * pxy; rx = roz * oy; ry = roz * ox;
    } else {
      rx = m_rcp_once(ox); ry = m_rcp_once(oy); rz = m_rcp_once(oz);
    }
    dx_push = m_min(dx_push, (fx - x) * rx);
    dx_push = m_min(dx_push, (fy - y) * ry);
    dx_push = m_min(dx_push, (fz - z) * rz);
  } else
#endif
  {
    dx_push = m_min(dx_push, (fx - x) * m_rcp_once(ox));
    if (multi_d) dx_push = m_min(dx_push, (fy - y) * m_rcp_once(oy));
    if (three_d) dx_push = m_min(dx_push, (fz - z) * m_rcp_once(oz));
  }
  is_absorbed = NOABS ? false : (dx_abs < dx_push) && (dx_abs < dx_sc);
  is_scattered = !is_absorbed && (dx_sc < dx_push);
  const double dx_move =
      NOABS ? m_min(dx_push, dx_sc) : (is_absorbed ? dx_abs : (is_scattered ? dx_sc : dx_push));
  d_rem -= dx_move;  // (exactly zero when the step ends at census: dx_move = d_rem)
  x = fma(ox, dx_move, x);
  if (multi_d) y = fma(oy, dx_move, y);
  if (three_d) z = fma(oz, dx_move, z);
  if constexpr (EXACTG) {  // (g.fd: eps_imc dx with the sign the direction component last had)
    dir_nudge_carried(x, ip, ox, fx, g.fd[0], iux, upx);
    if (multi_d) dir_nudge_carried(y, jp, oy, fy, g.fd[1], iuy, upy);
    if (three_d) dir_nudge_carried(z, kp, oz, fz, g.fd[2], iuz, upz);
  } else {
    dir_nudge(x, ip, ox, fx, kEpsImc * g.dx[0], iux, upx);
    if (multi_d) dir_nudge(y, jp, oy, fy, kEpsImc * g.dx[1], iuy, upy);
    if (three_d) dir_nudge(z, kp, oz, fz, kEpsImc * g.dx[2], iuz, upz);
  }
  if constexpr (BOTH) {
    // (the other face of the cell the step started in; never both faces of one axis: they are dx
    // apart, and a photon nudged through the face ahead ends fd beyond that one)
    int unused;
    dir_nudge_behind(x, ip, dir_face<EXACTG>(g, 0, iux - (int)upx, !upx, unused), kEpsImc * g.dx[0], upx);
    if (multi_d)
      dir_nudge_behind(y, jp, dir_face<EXACTG>(g, 1, iuy - (int)upy, !upy, unused), kEpsImc * g.dx[1], upy);
    if (three_d)
      dir_nudge_behind(z, kp, dir_face<EXACTG>(g, 2, iuz - (int)upz, !upz, unused), kEpsImc * g.dx[2], upz);
  }
}

// p = hit ? -copysign(m, p) : p -- the photon is put eps_imc dx inside the cell beyond the face it
// reached: one v_bfi_b32 on the high word and two selects, the negation as a source modifier
template <bool M_UNIFORM = false>
__device__ __forceinline__ double nudged(double p, double m, bool hit) {
#ifdef JB_NO_ASM_NUDGE
  return hit ? -copysign(m, p) : p;
#else
#ifdef JB_UNIFORM_NUDGE_CXX
  if constexpr (M_UNIFORM) return hit ? -copysign(m, p) : p;
#endif
  const unsigned long long mask = __ballot(hit);
  int chi, hi, lo;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(chi) : "s"(0x7fffffff), "v"(__double2hiint(m)), "v"(__double2hiint(p)));
  asm("v_cndmask_b32_e64 %0, %1, -%2, %3" : "=v"(hi) : "v"(__double2hiint(p)), "v"(chi), "s"(mask));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(p)), "v"(__double2loint(m)), "s"(mask));
  return __hiloint2double(hi, lo);
#endif
}

// The lean tracking step of transport_utils.hpp:118-159 + Xtoijk (transport.cpp:146) in CELL-LOCAL
// coordinates (jb_kernel_imc.hpp states the representation: p = x - cell centre per axis, unit
// direction omega, distance left to census, byte offset of the cell with byte strides 8 / sy / sz per
// axis; geometry: half cell widths h, h - eps_imc dx, smallest cell width).  2 draws.
//   distance to the face ahead   (h - sgn(omega) p) / |omega| = fma(-p, 1/omega, h |1/omega|)
//   move                         p = fma(omega, d, p)
//   nudge + Xtoijk               |p| > h - eps_imc dx  ->  p = -+(h - eps_imc dx), offset +- stride
// hit_any: the photon was put through a cell face (the caller then knows that a collision or the
// census of this step may have happened next to a BLOCK face).
struct CellGeom {
  double hx, hy, hz, mx, my, mz, dxp;
};
template <int NDIM, bool NOABS, bool UNIFORM = false, bool WIDELOG = false, class Rng>
__device__ __forceinline__ void imc_step_cell(const CellGeom &g, int sy, int sz, double lam_a, double lam_s,
                                              Rng &rng, double &drem, double &px, double &py, double &pz,
                                              double ox, double oy, double oz, unsigned &qoff,
                                              bool &is_absorbed, bool &is_scattered, bool &hit_any) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  constexpr int sx = 8;
  // ---- transport_utils.hpp:118-134: distances to collision, census, cell faces
  double dx_abs = 0.0;
  if constexpr (NOABS) rng.skip();
  else dx_abs = -lam_a * m_log_lean<true, WIDELOG>(rng.drand());
  const double dx_sc = -lam_s * m_log_lean<true, WIDELOG>(rng.drand());
  double dx_push = m_min(g.dxp, drem);
  double rx, ry = 0.0, rz = 0.0;
  if constexpr (three_d) {
    // the three reciprocals from ONE hardware reciprocal, of the product (imc_step_dir)
    const double pxy = ox * oy, q = pxy * oz;
    if (fabs(q) > 1.0e-250) {
      const double r = m_rcp_once(q);
      const double roz = r * oz;
      rz = r * pxy; rx = roz * oy; ry = roz * ox;
    } else {
      rx = m_rcp_once(ox); ry = m_rcp_once(oy); rz = m_rcp_once(oz);
    }
  } else {
    rx = m_rcp_once(ox);
    if (multi_d) ry = m_rcp_once(oy);
  }
  // (a direction component that is exactly zero: NaN, which minNum ignores -- the reference's
  // third branch)
  dx_push = m_min(dx_push, m_fnma(px, rx, g.hx * fabs(rx)));
  if (multi_d) dx_push = m_min(dx_push, m_fnma(py, ry, g.hy * fabs(ry)));
  if (three_d) dx_push = m_min(dx_push, m_fnma(pz, rz, g.hz * fabs(rz)));
  is_absorbed = NOABS ? false : (dx_abs < dx_push) && (dx_abs < dx_sc);
  is_scattered = !is_absorbed && (dx_sc < dx_push);
  const double dx_move =
      NOABS ? m_min(dx_push, dx_sc) : (is_absorbed ? dx_abs : (is_scattered ? dx_sc : dx_push));
  // ---- :136-159 move, nudge; transport.cpp:146 Xtoijk
  drem -= dx_move;  // (exactly zero when the step ends at census)
  px = fma(ox, dx_move, px);
  if (multi_d) py = fma(oy, dx_move, py);
  if (three_d) pz = fma(oz, dx_move, pz);
  const bool hit_x = fabs(px) > g.mx;
  const bool hit_y = multi_d && fabs(py) > g.my;
  const bool hit_z = three_d && fabs(pz) > g.mz;
  // (+-1 by the side of the cell the photon left through -- sign word of p, shifted down and ORed
  // with 1: two plain 32-bit operations -- times the stride: one multiply-add per axis onto the
  // offset, the strides as scalar operands)
  auto side = [](double p) { return (__double2hiint(p) >> 31) | 1; };
  qoff = (unsigned)mad24(hit_x ? side(px) : 0, sx, (int)qoff);
  px = nudged<UNIFORM>(px, g.mx, hit_x);
  if (multi_d) {
    qoff = (unsigned)mad24(hit_y ? side(py) : 0, sy, (int)qoff);
    py = nudged<UNIFORM>(py, g.my, hit_y);
  }
  if (three_d) {
    qoff = (unsigned)mad24(hit_z ? side(pz) : 0, sz, (int)qoff);
    pz = nudged<UNIFORM>(pz, g.mz, hit_z);
  }
  hit_any = hit_x || hit_y || hit_z;
}

// A crossing into a resident block one level coarser or finer, in cell-local coordinates
// (jb_device.hpp: kGhostCoarser / kGhostFiner; `code` = high word of the ghost cell's datum, `dst` =
// its low word).  Returns the byte offset of the photon's cell; position and geometry are updated.
template <int NDIM>
__device__ __forceinline__ unsigned cross_level(int code, unsigned dst, int sy, int sz, CellGeom &g, double &px,
                                                double &py, double &pz) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  if (code & 0x100) {  // to the coarser block: its cell widths are twice this block's
    px += (code & 0x400) ? -g.hx : g.hx;
    if (multi_d) py += (code & 0x800) ? -g.hy : g.hy;
    if (three_d) pz += (code & 0x1000) ? -g.hz : g.hz;
    g.hx *= 2.0; g.mx *= 2.0;   // (active axes only: an inactive one keeps the extent of the domain)
    if (multi_d) { g.hy *= 2.0; g.my *= 2.0; }
    if (three_d) { g.hz *= 2.0; g.mz *= 2.0; }
    g.dxp = dmin(2.0 * g.hx, dmin(2.0 * g.hy, 2.0 * g.hz));
    return dst;
  }
  // to a finer block: half the cell widths; which of the fine cells behind the coarse ghost cell
  g.hx *= 0.5; g.mx *= 0.5;
  if (multi_d) { g.hy *= 0.5; g.my *= 0.5; }
  if (three_d) { g.hz *= 0.5; g.mz *= 0.5; }
  g.dxp = dmin(2.0 * g.hx, dmin(2.0 * g.hy, 2.0 * g.hz));
  const int naxis = (code >> 10) & 3;
  const bool lower = (code & 0x1000) != 0;  // left through the lower face: the last fine layer
  {
    const bool up = !(px < 0.0);
    if (naxis == 0) px += lower ? -g.hx : g.hx;
    else { px += up ? -g.hx : g.hx; dst += up ? 8u : 0u; }
  }
  if (multi_d) {
    const bool up = !(py < 0.0);
    if (naxis == 1) py += lower ? -g.hy : g.hy;
    else { py += up ? -g.hy : g.hy; dst += up ? (unsigned)sy : 0u; }
  }
  if (three_d) {
    const bool up = !(pz < 0.0);
    if (naxis == 2) pz += lower ? -g.hz : g.hz;
    else { pz += up ? -g.hz : g.hz; dst += up ? (unsigned)sz : 0u; }
  }
  return dst;
}

// scattering.hpp:21-29 in direction space: the new unit direction (2 draws)
template <bool SC = false, class Rng>
__device__ __forceinline__ void scatter_dir(Rng &rng, double &ox, double &oy, double &oz) {
  double xi1, xi2;
  rng.drand2(xi1, xi2);
  const double mu = fma(2.0, xi1, -1.0);
  const double st = m_sqrt_lean(1.0 - mu * mu);
  double sn, cs;
  m_sincos2pi<SC>(xi2, sn, cs);
  ox = st * cs;
  oy = st * sn;
  oz = mu;
}

// reference transport_utils.hpp:111-160 -- one IMC tracking step, 2 draws
template <int NDIM, class Rng>
__device__ __forceinline__ void ptcl_transport_step(Step &s, Rng &rng) {
  double lam_abs, lam_sc;
  imc_cell_mfp(s.ff, s.aa, s.ss, lam_abs, lam_sc);
  imc_step_core<NDIM, false>(s, lam_abs, lam_sc, rng);
}

// Cyclic assignment (axis, axis+1, axis+2) <- (v1, v2, v3): the component order the reference
// passes to sample_face_iso_dir for an x-, y- or z-face (transport_utils.hpp:217,235,253).
__device__ __forceinline__ void assign_cyclic(int axis, double v1, double v2, double v3, double &vx,
                                              double &vy, double &vz) {
  vx = (axis == 0) ? v1 : (axis == 1 ? v3 : v2);
  vy = (axis == 0) ? v2 : (axis == 1 ? v1 : v3);
  vz = (axis == 0) ? v3 : (axis == 1 ? v2 : v1);
}

// The direction a DDMC leak through channel s.pend gives the particle
// (transport_utils.hpp:217,235,253), from the uniforms drawn at the time of the leak.
__device__ __forceinline__ void materialise_dir(Step &s) {
  const int axis = s.pend >> 1;
  const bool up = (s.pend & 1) != 0;
  double v1, v2, v3;
  face_iso_dir(up ? s.vv : -s.vv, s.pz1, s.pz2, v1, v2, v3);
  assign_cyclic(axis, v1, v2, v3, s.vx, s.vy, s.vz);
  s.pend = -1;
}

// reference transport_utils.hpp:163-277 -- one DDMC step
// draws: 1 (time); event: +1 (channel), leak: +2 (direction); census: +5
// The reference's six leak branches differ only in which face / axis they act on; here the
// channel is selected as data and the shared arithmetic (one half-isotropic direction sample)
// is executed once, so a wave whose lanes leak through different faces does not serialise six
// copies of it.  Operations and operands per lane are unchanged.
// Returns true when the particle reached census without an event (the caller then resamples it).
// LEAK_READY: s.P*_* already hold the leak opacities P_face / dx_d (lines 175-181), formed per
// cell by k_ddmc_pack from the same operands, instead of the face probabilities.
// LAZY: a leak draws the two direction uniforms (same order) but only records them (s.pend,
// s.pz1, s.pz2).  While a particle stays in DDMC cells its direction is never read -- the next
// leak, the census resampling or a block crossing (zero-velocity flag) overwrites it -- so the
// square roots and the sincos are evaluated only where a consumer appears (materialise_dir).
// PRELOG: the caller has made the step's first draw and passes -ln(u) in nlog (to have the
// logarithm evaluated while the cell record is still on its way).
template <int NDIM, bool LEAK_READY = false, bool LAZY = false, bool PRELOG = false, class Rng>
__device__ __forceinline__ bool ddmc_step_event(Step &s, Rng &rng, double nlog = 0.0) {
  constexpr int multi_d = NDIM >= 2 ? 1 : 0, three_d = NDIM == 3 ? 1 : 0;
  const double rmin = DBL_MIN;
  const double eps = kEpsDdmc;
  const double dx = s.xu - s.xl;
  const double dy = s.yu - s.yl;
  const double dz = s.zu - s.zl;

  const double leakx_l = LEAK_READY ? s.Px_l : s.Px_l / dx;
  const double leakx_u = LEAK_READY ? s.Px_u : s.Px_u / dx;
  const double leaky_l = LEAK_READY ? s.Py_l : s.Py_l / dy;
  const double leaky_u = LEAK_READY ? s.Py_u : s.Py_u / dy;
  const double leakz_l = LEAK_READY ? s.Pz_l : s.Pz_l / dz;
  const double leakz_u = LEAK_READY ? s.Pz_u : s.Pz_u / dz;
  const double leak_tot = leakx_l + leakx_u + leaky_l + leaky_u + leakz_l + leakz_u;

  const double cdf_ddmc = s.ffaa + leak_tot + rmin;
  // (-ln u in [1e-16, 37], c cdf in [c DBL_MIN, ~1e30]: inside the range where the lean division
  // sequence of jb_math.hpp gives the IEEE quotient -- no scaling or fix-up step needed)
  if constexpr (!PRELOG) nlog = -m_log(rng.drand());
  const double dt_ddmc = m_div(nlog, s.vv * cdf_ddmc);
  const double dt_end = (s.t_start + s.dt) - s.t;
  const bool is_ddmc_event = dt_ddmc < dt_end;

  s.t += dmin(dt_ddmc, dt_end);

  if (is_ddmc_event) {
    const double xi = cdf_ddmc * rng.drand();
    if (xi < s.ffaa) {
      s.is_absorbed = true;
    } else if (xi < s.ffaa + leak_tot) {
      const double xim = xi - s.ffaa;
      // cumulative thresholds, summed left to right exactly as written at lines 218-254
      const double c1 = leakx_l;
      const double c2 = leakx_l + leakx_u;
      const double c3 = c2 + leaky_l;
      const double c4 = c3 + leaky_u;
      const double c5 = c4 + leakz_l;
      const int ch = (xim < c1) ? 0 : (xim < c2) ? 1 : (xim < c3) ? 2 : (xim < c4) ? 3
                   : (xim < c5) ? 4 : (xim <= leak_tot) ? 5 : -1;
      if (ch >= 0) {
        const int axis = ch >> 1;
        const bool up = (ch & 1) != 0;
        const int step = up ? 1 : -1;
        s.ip += (axis == 0) ? step : 0;
        s.jp += (axis == 1) ? step * multi_d : 0;
        s.kp += (axis == 2) ? step * three_d : 0;
        // eps beyond the leak face along the leak axis, cell centre across it
        s.x = (axis == 0) ? (up ? s.xu + eps * dx : s.xl - eps * dx) : s.xl + 0.5 * dx;
        s.y = (axis == 1) ? (up ? s.yu + eps * dy : s.yl - eps * dy) : s.yl + 0.5 * dy;
        s.z = (axis == 2) ? (up ? s.zu + eps * dz : s.zl - eps * dz) : s.zl + 0.5 * dz;
        if constexpr (LAZY) {
          s.pz1 = rng.drand();
          s.pz2 = rng.drand();
          s.pend = ch;
        } else {
          double v1, v2, v3;
          sample_face_iso_dir(up ? s.vv : -s.vv, rng, v1, v2, v3);
          assign_cyclic(axis, v1, v2, v3, s.vx, s.vy, s.vz);
        }
      }
    }
  }
  return !is_ddmc_event;
}

// The same step for a particle in the virtual state (k_ddmc_all's event loop: LEAK_READY, LAZY,
// position never formed), reading the cell's STEP record: what ddmc_step_event derives from the
// cell record on every step -- the running sums of the leak opacities it compares the channel
// draw against, their total, and the refined reciprocal of c (f sigma_a + leak_tot + DBL_MIN) its
// division starts from -- formed once per cell and cycle by k_ddmc_pack, by the same operations
// on the same operands in the same order (transport_utils.hpp:175-254).  Saves the loop 5 of 7
// dependent additions and 5 of the 8 instructions of the division, right behind the gather.
//   rec = {f sigma_a, c1 .. c5, leak_tot, rcp}
struct DdmcStepRec {
  double ffaa, c1, c2, c3, c4, c5, leak_tot, rcp;
};
template <int NDIM, class Rng>
__device__ __forceinline__ bool ddmc_step_rec(const DdmcStepRec &r, double vv, double dt_end, double nlog,
                                              Rng &rng, double &t, int &ip, int &jp, int &kp, int &pend,
                                              unsigned long long &pzs, bool &is_absorbed) {
  constexpr int multi_d = NDIM >= 2 ? 1 : 0, three_d = NDIM == 3 ? 1 : 0;
  const double a2 = r.ffaa + r.leak_tot;
  const double cdf_ddmc = a2 + DBL_MIN;
  const double dt_ddmc = m_div_r(nlog, vv * cdf_ddmc, r.rcp);
  const bool is_ddmc_event = dt_ddmc < dt_end;
  t += dmin(dt_ddmc, dt_end);
  if (is_ddmc_event) {
    const double xi = cdf_ddmc * rng.drand();
    if (xi < r.ffaa) {
      is_absorbed = true;
    } else if (xi < a2) {
      const double xim = xi - r.ffaa;
      // the first threshold above xim, as the chain of transport_utils.hpp:218-254 finds it
      // (the leak opacities of an inactive axis are exactly zero -- k_ddmc_pack -- so its two
      // thresholds repeat the one before them and their comparisons can never be the first to hold)
      int ch = (xim <= r.leak_tot) ? 5 : -1;
      if constexpr (three_d) ch = (xim < r.c5) ? 4 : ch;   // (2-D: c5 = c4; 1-D: c5 = c4 = c3 = c2)
      if constexpr (multi_d) {
        ch = (xim < r.c4) ? 3 : ch;
        ch = (xim < r.c3) ? 2 : ch;
      }
      ch = (xim < r.c2) ? 1 : ch;
      ch = (xim < r.c1) ? 0 : ch;
      if (ch >= 0) {
        const int axis = ch >> 1;
        const int step = (ch & 1) != 0 ? 1 : -1;
        ip += (axis == 0) ? step : 0;
        jp += (axis == 1) ? step * multi_d : 0;
        kp += (axis == 2) ? step * three_d : 0;
        // the leak's two direction uniforms (transport_utils.hpp:217,235,253) are the next two
        // draws of the stream: remember where they start, step past them
        pzs = rng.s;
        rng.skip2();
        pend = ch;
      }
    }
  }
  return !is_ddmc_event;
}

// reference transport_utils.hpp:265-276: a DDMC particle that reaches census gets a uniform
// position in its cell (draw order z, x, y) and an isotropic direction with the polar axis along
// z.  5 draws.  Kept apart from the event part so that the tracking kernel can run it once per
// history, outside its event loop.
template <class Rng>
__device__ __forceinline__ void ddmc_census_resample(Step &s, Rng &rng) {
  const double dx = s.xu - s.xl;
  const double dy = s.yu - s.yl;
  const double dz = s.zu - s.zl;
  s.z = s.zl + rng.drand() * dz;
  s.x = s.xl + rng.drand() * dx;
  s.y = s.yl + rng.drand() * dy;
  const double mu = 1.0 - 2.0 * rng.drand();
  const double nu = sqrt(1.0 - mu * mu);
  const double xi2 = rng.drand();
  double sn, cs;
  m_sincos2pi(xi2, sn, cs);
  s.vz = s.vv * mu;
  s.vx = s.vv * nu * cs;
  s.vy = s.vv * nu * sn;
}

// reference transport_utils.hpp:163-277 -- one DDMC step (event part + census resampling)
template <int NDIM, class Rng>
__device__ __forceinline__ void ptcl_ddmc_step(Step &s, Rng &rng) {
  if (ddmc_step_event<NDIM, false, false>(s, rng)) ddmc_census_resample(s, rng);
}

// reference transport_utils.hpp:279-397 -- 0..3 draws.  The six face branches of the reference
// (x-, x+, y-, y+, z-, z+, in that order, y / z gated by dimensionality) are folded into "which
// face, if any" followed by one copy of the albedo arithmetic.
template <int NDIM, bool LAZY = false, class Rng>
__device__ __forceinline__ void ptcl_ddmc_albedo(Step &s, Rng &rng) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  const double dx = s.xu - s.xl;
  const double dy = s.yu - s.yl;
  const double dz = s.zu - s.zl;
  const double tol = 2.5 * kEpsImc;

  int face = -1;  // 0 x-, 1 x+, 2 y-, 3 y+, 4 z-, 5 z+
  if (fuzzy_equal(s.x, s.xl, dx, tol)) face = 0;
  else if (fuzzy_equal(s.x, s.xu, dx, tol)) face = 1;
  else if (multi_d && fuzzy_equal(s.y, s.yl, dy, tol)) face = 2;
  else if (multi_d && fuzzy_equal(s.y, s.yu, dy, tol)) face = 3;
  else if (three_d && fuzzy_equal(s.z, s.zl, dz, tol)) face = 4;
  else if (three_d && fuzzy_equal(s.z, s.zu, dz, tol)) face = 5;

  if (face >= 0) {
    if constexpr (LAZY) {
      if (s.pend >= 0) materialise_dir(s);  // the albedo reads the normal velocity component
    }
    const int axis = face >> 1;
    const double sgn = (face & 1) ? -1.0 : 1.0;  // +1 lower face, -1 upper face
    // (copies first: `c ? s.a : s.b` on members is an lvalue conditional, i.e. a select of
    // ADDRESSES followed by a load, which pins the whole Step record in scratch memory)
    const double svx = s.vx, svy = s.vy, svz = s.vz;
    const double sxl = s.xl, sxu = s.xu, syl = s.yl, syu = s.yu, szl = s.zl, szu = s.zu;
    const bool upper = (face & 1) != 0;
    const double dcell = (axis == 0) ? dx : (axis == 1 ? dy : dz);
    const double vn = (axis == 0) ? svx : (axis == 1 ? svy : svz);
    const double fx = upper ? sxu : sxl, fy = upper ? syu : syl, fz = upper ? szu : szl;
    const double fpos = (axis == 0) ? fx : (axis == 1 ? fy : fz);
    const double Pf = (2.0 / 3.0) / (s.sig * dcell + 2.0 * kLamExt);
    const double P = 2.0 * Pf * (1.0 + sgn * 1.5 * vn / s.vv);
    if (rng.drand() > P) {
      double v1, v2, v3;
      sample_face_iso_dir(-sgn * s.vv, rng, v1, v2, v3);
      assign_cyclic(axis, v1, v2, v3, s.vx, s.vy, s.vz);
      const double xn = fpos - sgn * kEpsImc * dcell;
      if (axis == 0) s.x = xn;
      else if (axis == 1) s.y = xn;
      else s.z = xn;
      s.is_rejected = true;
    }
  }

  if (!s.is_rejected) {
    s.x = 0.5 * (s.xl + s.xu);
    s.y = 0.5 * (s.yl + s.yu);
    s.z = 0.5 * (s.zl + s.zu);
  }
}

// Would ptcl_ddmc_albedo find the particle at a face of its cell (transport_utils.hpp:288-373)?
// The same six tests in the same order; when none holds the albedo step draws nothing and only
// moves the particle to the cell centre (:392-396).
template <int NDIM>
__device__ __forceinline__ bool at_cell_face(const Step &s) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  const double dx = s.xu - s.xl, dy = s.yu - s.yl, dz = s.zu - s.zl;
  const double tol = 2.5 * kEpsImc;
  return fuzzy_equal(s.x, s.xl, dx, tol) || fuzzy_equal(s.x, s.xu, dx, tol) ||
         (multi_d && (fuzzy_equal(s.y, s.yl, dy, tol) || fuzzy_equal(s.y, s.yu, dy, tol))) ||
         (three_d && (fuzzy_equal(s.z, s.zl, dz, tol) || fuzzy_equal(s.z, s.zu, dz, tol)));
}

// reference sample_ddmc_bface.cpp:24-41 -- 2 draws
template <class Rng>
__device__ __forceinline__ void sample_face_2d(int i_l, double dx, double P_l, double P_u, Rng &rng,
                                               int &i, double &x) {
  const double xi = (P_l + P_u) * rng.drand();
  if (xi < P_l) {
    x -= dx * rng.drand();
    i = i_l;
  } else {
    x += dx * rng.drand();
    i = i_l + 1;
  }
}

// reference sample_ddmc_bface.cpp:43-78 -- 3 draws
template <class Rng>
__device__ __forceinline__ void sample_face_3d(int i1_l, int i2_l, double dx1, double dx2,
                                               double P_ll, double P_lu, double P_ul, double P_uu,
                                               Rng &rng, int &i1, int &i2, double &x1, double &x2) {
  const double xi = (P_ll + P_lu + P_ul + P_uu) * rng.drand();
  if (xi < P_ll) {
    x1 -= dx1 * rng.drand(); i1 = i1_l;
    x2 -= dx2 * rng.drand(); i2 = i2_l;
  } else if (xi < P_ll + P_lu) {
    x1 += dx1 * rng.drand(); i1 = i1_l + 1;
    x2 -= dx2 * rng.drand(); i2 = i2_l;
  } else if (xi < P_ll + P_lu + P_ul) {
    x1 -= dx1 * rng.drand(); i1 = i1_l;
    x2 += dx2 * rng.drand(); i2 = i2_l + 1;
  } else {
    x1 += dx1 * rng.drand(); i1 = i1_l + 1;
    x2 += dx2 * rng.drand(); i2 = i2_l + 1;
  }
}

}  // namespace jb
