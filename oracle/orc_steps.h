/* oracle/orc_steps.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the reference's per-event arithmetic.  Each function names the reference
 * lines it follows; the order of floating-point operations and of random draws is kept exactly
 * (SURVEY.md Appendix A), because event-for-event parity depends on it.
 */
#ifndef ORC_STEPS_H_
#define ORC_STEPS_H_

#include <float.h>
#include <math.h>

#include "orc_math.h"
#include "orc_rng.h"

/* parthenon::robust::EPS() = 10 * machine epsilon (un-vendored; assumed) */
#define ORC_EPS (10.0 * DBL_EPSILON)
/* reference src/jaybenne/transport_utils.hpp:24-25 */
#define ORC_EPS_IMC (1.0e6 * ORC_EPS)
#define ORC_EPS_DDMC (1.0e8 * ORC_EPS)
#define ORC_LAM_EXT 0.7104 /* transport_utils.hpp:282, jaybenne.cpp:326 */
#define ORC_TWO_PI (2.0 * M_PI)

/* reference src/jaybenne/jaybenne_utils.hpp:44-49 */
static inline int orc_fuzzy_equal(double a, double b, double c, double eps) {
  return fabs(a - b) < c * eps;
}

/* reference src/jaybenne/transport_utils.hpp:27-39: half-isotropic direction about a face
 * normal; v1 is the component along the normal (sign carried by vv). 2 draws. */
static inline void orc_sample_face_iso_dir(double vv, orc_rng *rng, double *v1, double *v2,
                                           double *v3) {
  const double mu = sqrt(orc_drand(rng));
  const double nu = sqrt(1.0 - mu * mu);
  const double xi2 = orc_drand(rng);
  double sn, cs;
  orc_sincos2pi(xi2, &sn, &cs);
  *v1 = vv * mu;
  *v2 = vv * nu * cs;
  *v3 = vv * nu * sn;
}

/* reference src/jaybenne/scattering.hpp:21-29: isotropic scatter. 2 draws. */
static inline void orc_scatter(orc_rng *rng, double vv, double *vx, double *vy, double *vz) {
  const double mu = 2.0 * orc_drand(rng) - 1.0;
  const double xi2 = orc_drand(rng);
  const double st = sqrt(1.0 - mu * mu);
  double sn, cs;
  orc_sincos2pi(xi2, &sn, &cs);
  *vx = vv * st * cs;
  *vy = vv * st * sn;
  *vz = vv * mu;
}

/* reference src/jaybenne/planck.hpp:26-50 (Everett & Cashwell 1972). 5 draws.  `sb` really is
 * the Stefan-Boltzmann constant. */
static inline double orc_sample_planck_energy(orc_rng *rng, double sb, double temp) {
  const double xi0 = orc_drand(rng);
  /* pow(M_PI, 4.0) / 90.0 : libm pow in the reference; (pi^2)^2 is the portable flavour */
  const double pi4 = (orc_math_mode == ORC_MATH_LIBM) ? pow(M_PI, 4.0) : (M_PI * M_PI) * (M_PI * M_PI);
  const double rhs = xi0 * pi4 / 90.0;
  double ll = 1.0;
  for (int l = 1; l < 100; ++l) {
    double lhs = 0.0;
    for (int j = 1; j <= l; ++j) {
      const double dj = (double)j;
      /* pow(j, -4.0): j^4 is exact in double for j < 100, so 1/(j^4) is correctly rounded */
      lhs += (orc_math_mode == ORC_MATH_LIBM) ? pow(dj, -4.0) : 1.0 / ((dj * dj) * (dj * dj));
    }
    if (lhs >= rhs) {
      ll = (double)l;
      break;
    }
  }
  const double xi1 = orc_drand(rng);
  const double xi2 = orc_drand(rng);
  const double xi3 = orc_drand(rng);
  const double xi4 = orc_drand(rng);
  return -(1.0 / ll) * orc_log(xi1 * xi2 * xi3 * xi4) * sb * temp;
}

/* Everything a step function reads or updates for one particle in one cell. */
typedef struct orc_step {
  /* constants for the step */
  double t_start, dt;
  double ff, aa, ss; /* Fleck factor, absorption and scattering opacity (1/length) */
  double vv;         /* speed of light */
  double dx_push;    /* min cell extent of the block */
  int multi_d, three_d;
  double xl, yl, zl, xu, yu, zu;             /* cell faces */
  double Px_l, Py_l, Pz_l, Px_u, Py_u, Pz_u; /* DDMC face probabilities */
  /* updated */
  double t, x, y, z, vx, vy, vz;
  int ip, jp, kp;
  int is_absorbed, is_scattered, is_rejected;
} orc_step;

static inline double orc_min(double a, double b) { return (b < a) ? b : a; } /* std::min */

/* reference src/jaybenne/transport_utils.hpp:111-160: one IMC tracking step. 2 draws. */
static inline void orc_ptcl_transport_step(orc_step *s, orc_rng *rng) {
  const double rmin = DBL_MIN;
  const double lam_abs = 1.0 / (s->ff * s->aa + rmin);
  const double lam_sc = 1.0 / (s->ss + (1.0 - s->ff) * s->aa + rmin);
  const double dx_abs = -lam_abs * orc_log(orc_drand(rng));
  const double dx_sc = -lam_sc * orc_log(orc_drand(rng));
  const double dx_end = s->vv * ((s->t_start + s->dt) - s->t);
  double dx_push = orc_min(s->dx_push, dx_end);
  if (s->vx > 0.0)
    dx_push = orc_min(dx_push, s->vv * (s->xu - s->x) / s->vx);
  else if (s->vx < 0.0)
    dx_push = orc_min(dx_push, s->vv * (s->xl - s->x) / s->vx);
  if (s->multi_d) {
    if (s->vy > 0.0)
      dx_push = orc_min(dx_push, s->vv * (s->yu - s->y) / s->vy);
    else if (s->vy < 0.0)
      dx_push = orc_min(dx_push, s->vv * (s->yl - s->y) / s->vy);
  }
  if (s->three_d) {
    if (s->vz > 0.0)
      dx_push = orc_min(dx_push, s->vv * (s->zu - s->z) / s->vz);
    else if (s->vz < 0.0)
      dx_push = orc_min(dx_push, s->vv * (s->zl - s->z) / s->vz);
  }

  s->is_absorbed = (dx_abs < dx_push) && (dx_abs < dx_sc);
  s->is_scattered = !s->is_absorbed && (dx_sc < dx_push);

  const double dt_push = (s->is_absorbed ? dx_abs : (s->is_scattered ? dx_sc : dx_push)) / s->vv;

  s->t += dt_push;
  s->x += s->vx * dt_push;
  s->y += (double)s->multi_d * s->vy * dt_push;
  s->z += (double)s->three_d * s->vz * dt_push;

  /* a particle that lands within eps of a face is put eps beyond it (lines 151-159) */
  const double fdx = ORC_EPS_IMC * (s->xu - s->xl);
  const double fdy = ORC_EPS_IMC * (s->yu - s->yl);
  const double fdz = ORC_EPS_IMC * (s->zu - s->zl);
  if (fabs(s->x - s->xl) < fdx) s->x = s->xl - fdx;
  if (fabs(s->x - s->xu) < fdx) s->x = s->xu + fdx;
  if (s->multi_d && fabs(s->y - s->yl) < fdy) s->y = s->yl - fdy;
  if (s->multi_d && fabs(s->y - s->yu) < fdy) s->y = s->yu + fdy;
  if (s->three_d && fabs(s->z - s->zl) < fdz) s->z = s->zl - fdz;
  if (s->three_d && fabs(s->z - s->zu) < fdz) s->z = s->zu + fdz;
}

/* reference src/jaybenne/transport_utils.hpp:163-277: one DDMC step.
 * draws: 1 (time); event: +1 (channel), leak: +2 (direction); census: +5. */
static inline void orc_ptcl_ddmc_step(orc_step *s, orc_rng *rng) {
  const double rmin = DBL_MIN;
  const double eps = ORC_EPS_DDMC;
  const double dx = s->xu - s->xl;
  const double dy = s->yu - s->yl;
  const double dz = s->zu - s->zl;

  const double leakx_l = s->Px_l / dx;
  const double leakx_u = s->Px_u / dx;
  const double leaky_l = s->Py_l / dy;
  const double leaky_u = s->Py_u / dy;
  const double leakz_l = s->Pz_l / dz;
  const double leakz_u = s->Pz_u / dz;
  const double leak_tot = leakx_l + leakx_u + leaky_l + leaky_u + leakz_l + leakz_u;

  const double cdf_ddmc = s->ff * s->aa + leak_tot + rmin;
  const double dt_ddmc = -orc_log(orc_drand(rng)) / (s->vv * cdf_ddmc);
  const double dt_end = (s->t_start + s->dt) - s->t;
  const int is_ddmc_event = dt_ddmc < dt_end;

  s->t += orc_min(dt_ddmc, dt_end);

  if (is_ddmc_event) {
    const double xi = cdf_ddmc * orc_drand(rng);
    if (xi < s->ff * s->aa) {
      s->is_absorbed = 1;
    } else if (xi < s->ff * s->aa + leak_tot) {
      const double xim = xi - s->ff * s->aa;
      if (xim < leakx_l) { /* -x */
        s->ip -= 1;
        s->x = s->xl - eps * dx;
        s->y = s->yl + 0.5 * dy;
        s->z = s->zl + 0.5 * dz;
        orc_sample_face_iso_dir(-s->vv, rng, &s->vx, &s->vy, &s->vz);
      } else if (xim < leakx_l + leakx_u) { /* +x */
        s->ip += 1;
        s->x = s->xu + eps * dx;
        s->y = s->yl + 0.5 * dy;
        s->z = s->zl + 0.5 * dz;
        orc_sample_face_iso_dir(s->vv, rng, &s->vx, &s->vy, &s->vz);
      } else if (xim < leakx_l + leakx_u + leaky_l) { /* -y */
        s->jp -= s->multi_d;
        s->y = s->yl - eps * dy;
        s->z = s->zl + 0.5 * dz;
        s->x = s->xl + 0.5 * dx;
        orc_sample_face_iso_dir(-s->vv, rng, &s->vy, &s->vz, &s->vx);
      } else if (xim < leakx_l + leakx_u + leaky_l + leaky_u) { /* +y */
        s->jp += s->multi_d;
        s->y = s->yu + eps * dy;
        s->z = s->zl + 0.5 * dz;
        s->x = s->xl + 0.5 * dx;
        orc_sample_face_iso_dir(s->vv, rng, &s->vy, &s->vz, &s->vx);
      } else if (xim < leakx_l + leakx_u + leaky_l + leaky_u + leakz_l) { /* -z */
        s->kp -= s->three_d;
        s->z = s->zl - eps * dz;
        s->x = s->xl + 0.5 * dx;
        s->y = s->yl + 0.5 * dy;
        orc_sample_face_iso_dir(-s->vv, rng, &s->vz, &s->vx, &s->vy);
      } else if (xim <= leak_tot) { /* +z */
        s->kp += s->three_d;
        s->z = s->zu + eps * dz;
        s->x = s->xl + 0.5 * dx;
        s->y = s->yl + 0.5 * dy;
        orc_sample_face_iso_dir(s->vv, rng, &s->vz, &s->vx, &s->vy);
      }
    }
  } else {
    /* census: uniform position in the cell (draw order z, x, y), isotropic direction with
     * the polar axis along z (lines 267-275) */
    s->z = s->zl + orc_drand(rng) * dz;
    s->x = s->xl + orc_drand(rng) * dx;
    s->y = s->yl + orc_drand(rng) * dy;
    const double mu = 1.0 - 2.0 * orc_drand(rng);
    const double nu = sqrt(1.0 - mu * mu);
    const double xi2 = orc_drand(rng);
    double sn, cs;
    orc_sincos2pi(xi2, &sn, &cs);
    s->vz = s->vv * mu;
    s->vx = s->vv * nu * cs;
    s->vy = s->vv * nu * sn;
  }
}

/* One face of the IMC->DDMC albedo test (the six near-identical branches of
 * transport_utils.hpp:288-389).  sgn = +1 for a lower face, -1 for an upper face. */
static inline void orc_albedo_face(orc_step *s, orc_rng *rng, double dcell, double sgn,
                                   double *vn, double *va, double *vb, double *xn, double face) {
  const double Pf = (2.0 / 3.0) / ((s->aa + s->ss) * dcell + 2.0 * ORC_LAM_EXT);
  const double P = 2.0 * Pf * (1.0 + sgn * 1.5 * *vn / s->vv);
  if (orc_drand(rng) > P) {
    /* rejected: send it back out of the DDMC cell through the face it came in by */
    orc_sample_face_iso_dir(-sgn * s->vv, rng, vn, va, vb);
    *xn = face - sgn * ORC_EPS_IMC * dcell;
    s->is_rejected = 1;
  }
}

/* reference src/jaybenne/transport_utils.hpp:279-397. 0-3 draws. */
static inline void orc_ptcl_ddmc_albedo(orc_step *s, orc_rng *rng) {
  const double dx = s->xu - s->xl;
  const double dy = s->yu - s->yl;
  const double dz = s->zu - s->zl;
  const double tol = 2.5 * ORC_EPS_IMC;

  if (orc_fuzzy_equal(s->x, s->xl, dx, tol)) {
    orc_albedo_face(s, rng, dx, 1.0, &s->vx, &s->vy, &s->vz, &s->x, s->xl);
  } else if (orc_fuzzy_equal(s->x, s->xu, dx, tol)) {
    orc_albedo_face(s, rng, dx, -1.0, &s->vx, &s->vy, &s->vz, &s->x, s->xu);
  } else if (orc_fuzzy_equal(s->y, s->yl, dy, tol) && s->multi_d) {
    orc_albedo_face(s, rng, dy, 1.0, &s->vy, &s->vz, &s->vx, &s->y, s->yl);
  } else if (orc_fuzzy_equal(s->y, s->yu, dy, tol) && s->multi_d) {
    orc_albedo_face(s, rng, dy, -1.0, &s->vy, &s->vz, &s->vx, &s->y, s->yu);
  } else if (orc_fuzzy_equal(s->z, s->zl, dz, tol) && s->three_d) {
    orc_albedo_face(s, rng, dz, 1.0, &s->vz, &s->vx, &s->vy, &s->z, s->zl);
  } else if (orc_fuzzy_equal(s->z, s->zu, dz, tol) && s->three_d) {
    orc_albedo_face(s, rng, dz, -1.0, &s->vz, &s->vx, &s->vy, &s->z, s->zu);
  }

  if (!s->is_rejected) { /* admitted (or not at a face): DDMC particles live at cell centres */
    s->x = 0.5 * (s->xl + s->xu);
    s->y = 0.5 * (s->yl + s->yu);
    s->z = 0.5 * (s->zl + s->zu);
  }
}

/* reference src/jaybenne/sample_ddmc_bface.cpp:24-41. 2 draws. */
static inline void orc_sample_face_2d(int i_l, double dx, double P_l, double P_u, orc_rng *rng,
                                      int *i, double *x) {
  const double xi = (P_l + P_u) * orc_drand(rng);
  if (xi < P_l) {
    *x -= dx * orc_drand(rng);
    *i = i_l;
  } else {
    *x += dx * orc_drand(rng);
    *i = i_l + 1;
  }
}

/* reference src/jaybenne/sample_ddmc_bface.cpp:43-78. 3 draws. */
static inline void orc_sample_face_3d(int i1_l, int i2_l, double dx1, double dx2, double P_ll,
                                      double P_lu, double P_ul, double P_uu, orc_rng *rng,
                                      int *i1, int *i2, double *x1, double *x2) {
  const double xi = (P_ll + P_lu + P_ul + P_uu) * orc_drand(rng);
  if (xi < P_ll) {
    *x1 -= dx1 * orc_drand(rng); *i1 = i1_l;
    *x2 -= dx2 * orc_drand(rng); *i2 = i2_l;
  } else if (xi < P_ll + P_lu) {
    *x1 += dx1 * orc_drand(rng); *i1 = i1_l + 1;
    *x2 -= dx2 * orc_drand(rng); *i2 = i2_l;
  } else if (xi < P_ll + P_lu + P_ul) {
    *x1 -= dx1 * orc_drand(rng); *i1 = i1_l;
    *x2 += dx2 * orc_drand(rng); *i2 = i2_l + 1;
  } else {
    *x1 += dx1 * orc_drand(rng); *i1 = i1_l + 1;
    *x2 += dx2 * orc_drand(rng); *i2 = i2_l + 1;
  }
}

#endif /* ORC_STEPS_H_ */
