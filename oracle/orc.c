/* oracle/orc.c -- TEST INFRASTRUCTURE ONLY (see orc.h for what pins it).
 *
 * Task-level CPU restatement of the reference's history loop.  The reference's iterate-sublist
 * (transport -> swarm send/receive with boundary conditions -> SampleDDMCBlockFace ->
 * CheckCompletion, jaybenne.cpp:113-131) is flattened per particle: when a particle leaves its
 * block the "comm phase" operations are applied to it at once and its history continues.  With
 * one independent random stream per particle this yields the same particle states as running
 * the phases in lock step.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "orc_steps.h"

int orc_math_mode = ORC_MATH_LIBM;
static int orc_threads = 1;

void orc_set_math_mode(int mode) { orc_math_mode = mode ? ORC_MATH_PORTABLE : ORC_MATH_LIBM; }
int orc_get_math_mode(void) { return orc_math_mode; }
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return orc_threads; }

/* ---------------------------------------------------------------------------------------- */
/* vectorised helpers for tests                                                              */
/* ---------------------------------------------------------------------------------------- */
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc_philox4x32_10(ctr, key, out);
}
uint64_t orc_seed_state(uint32_t seed, uint32_t domain, uint64_t id) {
  return orc_rng_seed_state(seed, domain, id);
}
uint64_t orc_stream_start(uint32_t seed, uint64_t id) { return orc_rng_stream_start(seed, id); }
uint64_t orc_draw_stream(uint64_t state, int n, double *out) {
  orc_rng r = orc_rng_from_state(state);
  for (int i = 0; i < n; ++i) out[i] = orc_drand(&r);
  return r.s;
}
void orc_math_log(const double *x, int n, double *out) {
  for (int i = 0; i < n; ++i) out[i] = orc_log(x[i]);
}
void orc_math_sincos(const double *x, int n, double *sn, double *cs) {
  for (int i = 0; i < n; ++i) orc_sincos(x[i], &sn[i], &cs[i]);
}
void orc_math_sincos2pi(const double *u, int n, double *sn, double *cs) {
  for (int i = 0; i < n; ++i) orc_sincos2pi(u[i], &sn[i], &cs[i]);
}
void orc_math_one_minus_exp_neg(const double *x, int n, double *out) {
  for (int i = 0; i < n; ++i) out[i] = orc_one_minus_exp_neg(x[i]);
}
void orc_math_acos(const double *x, int n, double *out) {
  for (int i = 0; i < n; ++i) out[i] = orc_acos(x[i]);
}

static orc_rng tape_rng(const double *tape, int ntape) {
  orc_rng r = {0ull, 0u, tape, ntape};
  return r;
}
int orc_sizeof_step(void) { return (int)sizeof(orc_step); }
int orc_call_transport_step(void *st, const double *tape, int ntape) {
  orc_rng r = tape_rng(tape, ntape);
  orc_ptcl_transport_step((orc_step *)st, &r);
  return (int)r.ctr;
}
int orc_call_ddmc_step(void *st, const double *tape, int ntape) {
  orc_rng r = tape_rng(tape, ntape);
  orc_ptcl_ddmc_step((orc_step *)st, &r);
  return (int)r.ctr;
}
int orc_call_ddmc_albedo(void *st, const double *tape, int ntape) {
  orc_rng r = tape_rng(tape, ntape);
  orc_ptcl_ddmc_albedo((orc_step *)st, &r);
  return (int)r.ctr;
}
int orc_call_scatter(double vv, const double *tape, int ntape, double v[3]) {
  orc_rng r = tape_rng(tape, ntape);
  orc_scatter(&r, vv, &v[0], &v[1], &v[2]);
  return (int)r.ctr;
}
int orc_call_face_iso_dir(double vv, const double *tape, int ntape, double v[3]) {
  orc_rng r = tape_rng(tape, ntape);
  orc_sample_face_iso_dir(vv, &r, &v[0], &v[1], &v[2]);
  return (int)r.ctr;
}
int orc_call_planck(double sb, double temp, const double *tape, int ntape, double *e) {
  orc_rng r = tape_rng(tape, ntape);
  *e = orc_sample_planck_energy(&r, sb, temp);
  return (int)r.ctr;
}
int orc_call_face_2d(int i_l, double dx, double P_l, double P_u, const double *tape, int ntape,
                     int *i, double *x) {
  orc_rng r = tape_rng(tape, ntape);
  orc_sample_face_2d(i_l, dx, P_l, P_u, &r, i, x);
  return (int)r.ctr;
}
int orc_call_face_3d(int i1_l, int i2_l, double dx1, double dx2, const double P[4],
                     const double *tape, int ntape, int ij[2], double x12[2]) {
  orc_rng r = tape_rng(tape, ntape);
  orc_sample_face_3d(i1_l, i2_l, dx1, dx2, P[0], P[1], P[2], P[3], &r, &ij[0], &ij[1], &x12[0],
                     &x12[1]);
  return (int)r.ctr;
}

/* ---------------------------------------------------------------------------------------- */
/* mesh geometry (Parthenon semantics restated from call sites)            */
/* ---------------------------------------------------------------------------------------- */
typedef struct geom {
  int ni, nj, nk;       /* array extents incl. ghosts */
  int is, js, ks;       /* first interior index */
  int ie, je, ke;       /* last interior index */
  int64_t ntot;         /* cells per block incl. ghosts */
  int ncell;            /* interior cells per block */
} geom;

static geom make_geom(const orc_mesh *M) {
  geom g;
  g.is = M->ng;
  g.js = (M->ndim >= 2) ? M->ng : 0;
  g.ks = (M->ndim >= 3) ? M->ng : 0;
  g.ni = M->nx[0] + 2 * g.is;
  g.nj = M->nx[1] + 2 * g.js;
  g.nk = M->nx[2] + 2 * g.ks;
  g.ie = g.is + M->nx[0] - 1;
  g.je = g.js + M->nx[1] - 1;
  g.ke = g.ks + M->nx[2] - 1;
  g.ntot = (int64_t)g.ni * g.nj * g.nk;
  g.ncell = M->nx[0] * M->nx[1] * M->nx[2];
  return g;
}
static inline int64_t cidx(const geom *g, int b, int k, int j, int i) {
  return (int64_t)b * g->ntot + ((int64_t)k * g->nj + j) * g->ni + i;
}
/* cell-centre coordinate: Parthenon UniformCartesian Xc(idx) = x0 + (idx + 0.5) dx with x0 the
 * coordinate of the first (ghost) index */
static inline double xc(const orc_mesh *M, int b, int d, int first_interior, int idx) {
  const double dx = M->blk_dx[3 * b + d];
  const double x0 = M->blk_xmin[3 * b + d] - (double)first_interior * dx;
  return x0 + ((double)idx + 0.5) * dx;
}
/* SwarmDeviceContext::Xtoijk: inactive dimensions keep their single index */
static inline void xtoijk(const orc_mesh *M, const geom *g, int b, double x, double y, double z,
                          int *i, int *j, int *k) {
  /* (x - x_min) * (1 / dx): the reciprocal is a per-block constant.  Parthenon's own Xtoijk is
   * un-vendored; the step functions keep particles >= 2e-9 dx away from cell faces, so the
   * index does not depend on how the quotient is rounded. */
  *i = (int)floor((x - M->blk_xmin[3 * b + 0]) * (1.0 / M->blk_dx[3 * b + 0])) + g->is;
  *j = (M->ndim >= 2)
           ? (int)floor((y - M->blk_xmin[3 * b + 1]) * (1.0 / M->blk_dx[3 * b + 1])) + g->js
           : g->js;
  *k = (M->ndim >= 3)
           ? (int)floor((z - M->blk_xmin[3 * b + 2]) * (1.0 / M->blk_dx[3 * b + 2])) + g->ks
           : g->ks;
}
static inline int on_block(const geom *g, int i, int j, int k) {
  return i >= g->is && i <= g->ie && j >= g->js && j <= g->je && k >= g->ks && k <= g->ke;
}

/* EOS / opacity models of the host (singularity IdealGas, Gray, GrayS) */
static inline double eos_temperature(const orc_params *P, double rho, double sie) {
  (void)rho;
  const double t = sie / P->cv;
  return t > 0.0 ? t : 0.0;
}
static inline double opac_absorption(const orc_params *P, double rho, double temp, double nu) {
  if (P->opac_model == 1) { /* EPBremss, orc.h */
    if (!(temp > 0.0)) return 0.0;
    const double g = orc_one_minus_exp_neg((P->ep_B * nu) / temp);
    return ((P->ep_A * (rho * rho)) / sqrt(temp)) * (g / ((nu * nu) * nu));
  }
  return rho * P->kappa_a;
}
static inline double opac_emissivity(const orc_params *P, double rho, double temp) {
  if (P->opac_model == 1) return (P->ep_E * (rho * rho)) * sqrt(temp > 0.0 ? temp : 0.0);
  const double t2 = temp * temp;
  return (rho * P->kappa_a) * ((4.0 * P->sb) * (t2 * t2));
}
static inline double opac_scattering(const orc_params *P, double rho, double temp, double nu) {
  (void)temp; (void)nu;
  return (rho / P->apm) * P->kappa_s;
}
void orc_model_coefficients(const double scales[4], double out[4]) {
  /* CGS, CODATA 2018: e (esu), m_e, m_p, h, k_B, c, sigma_Thomson */
  const double qe = 4.803204712570263e-10, me = 9.1093837015e-28, mp = 1.67262192369e-24,
               hp = 6.62607015e-27, kb = 1.380649e-16, cl = 2.99792458e10,
               sigma_t = 6.6524587321e-25, pi = 3.14159265358979323846;
  const double tau = scales[0], mu = scales[1], lam = scales[2], th = scales[3];
  const double e6 = (qe * qe) * (qe * qe) * (qe * qe);
  /* Rybicki & Lightman 5.18a and 5.15a, Z = 1, Gaunt factor 1 */
  const double k_abs = (4.0 * e6) / (3.0 * me * hp * cl) * sqrt((2.0 * pi) / (3.0 * kb * me));
  const double k_em = sqrt((2.0 * pi * kb) / (3.0 * me)) *
                      ((32.0 * pi * e6) / (3.0 * hp * me * (cl * cl * cl)));
  const double n_per_rho = (mu / (lam * lam * lam)) / mp; /* n_e = n_i per unit code density */
  const double n2 = n_per_rho * n_per_rho, tau3 = tau * tau * tau;
  out[0] = lam * k_abs * n2 / sqrt(th) * tau3;    /* alpha: 1/cm -> 1/length; nu^-3 -> tau^3 */
  out[1] = hp / (kb * th * tau);                  /* x = h nu / k T */
  out[2] = k_em * sqrt(th) * n2 * (tau3 * lam / mu); /* erg s^-1 cm^-3 -> code */
  out[3] = sigma_t / (lam * lam);                 /* n_e = rho / apm, apm in code mass units */
}
void orc_model_eval(const orc_params *P, int which, const double *x, int n, double *out) {
  for (int i = 0; i < n; ++i) {
    const double rho = x[3 * i], temp = x[3 * i + 1], nu = x[3 * i + 2];
    out[i] = which == 0 ? opac_absorption(P, rho, temp, nu)
             : which == 1 ? opac_emissivity(P, rho, temp) : opac_scattering(P, rho, temp, nu);
  }
}

/* ---------------------------------------------------------------------------------------- */
/* a3 / a4: UpdateDerivedTransportFields (reference jaybenne.cpp:285-492)                    */
/* ---------------------------------------------------------------------------------------- */
static void face_probs_dir(const orc_mesh *M, const orc_params *P, const geom *g, int d,
                           double *F) {
  const int e[3] = {d == 0, d == 1, d == 2};
  for (int b = 0; b < M->nblocks; ++b) {
    const double dxd = M->blk_dx[3 * b + d];
    const double rlev = (double)M->blk_level[b];
    const double rlev_l = (double)M->blk_nbr_lev[6 * b + 2 * d];
    const double rlev_u = (double)M->blk_nbr_lev[6 * b + 2 * d + 1];
    const int fs = (d == 0) ? g->is : (d == 1 ? g->js : g->ks);
    const int fu = ((d == 0) ? g->ie : (d == 1 ? g->je : g->ke)) + 1;
    for (int k = g->ks; k <= g->ke + e[2]; ++k)
      for (int j = g->js; j <= g->je + e[1]; ++j)
        for (int i = g->is; i <= g->ie + e[0]; ++i) {
          const int f = (d == 0) ? i : (d == 1 ? j : k);
          const double dx_l = (f == fs) ? pow(2.0, rlev - rlev_l) * dxd : dxd;
          const double dx_u = (f == fu) ? pow(2.0, rlev - rlev_u) * dxd : dxd;
          const int64_t cl = cidx(g, b, k - e[2], j - e[1], i - e[0]);
          const int64_t cu = cidx(g, b, k, j, i);
          const double rho_l = M->rho[cl], rho_u = M->rho[cu];
          const double temp_l = eos_temperature(P, rho_l, M->sie[cl]);
          const double temp_u = eos_temperature(P, rho_u, M->sie[cu]);
          const double ss_l = opac_scattering(P, rho_l, temp_l, 1.0);
          const double aa_l = opac_absorption(P, rho_l, temp_l, 1.0);
          const double ss_u = opac_scattering(P, rho_u, temp_u, 1.0);
          const double aa_u = opac_absorption(P, rho_u, temp_u, 1.0);
          double tau_l = dx_l * (ss_l + aa_l);
          double tau_u = dx_u * (ss_u + aa_u);
          tau_l = tau_l > P->tau_ddmc ? tau_l : 2.0 * ORC_LAM_EXT;
          tau_u = tau_u > P->tau_ddmc ? tau_u : 2.0 * ORC_LAM_EXT;
          F[cu] = 2.0 / (3.0 * (tau_l + tau_u));
        }
  }
}

void orc_update_derived_transport_fields(const orc_mesh *M, const orc_params *P, double dt) {
  const geom g = make_geom(M);
  for (int b = 0; b < M->nblocks; ++b)
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          const int64_t c = cidx(&g, b, k, j, i);
          const double rho = M->rho[c];
          const double temp = eos_temperature(P, rho, M->sie[c]);
          const double emis = opac_emissivity(P, rho, temp);
          M->fleck[c] = 1.0 / (1.0 + (4.0 * emis / (rho * P->cv * temp)) * dt);
        }
  if (P->use_ddmc) {
    face_probs_dir(M, P, &g, 0, M->P1);
    if (M->ndim > 1) face_probs_dir(M, P, &g, 1, M->P2);
    if (M->ndim > 2) face_probs_dir(M, P, &g, 2, M->P3);
  }
}

/* ---------------------------------------------------------------------------------------- */
/* a5 / a6: SourcePhotons (reference sourcing.cpp:25-208)                                    */
/* ---------------------------------------------------------------------------------------- */
static inline uint64_t cell_stream_id(uint32_t epoch, int b, int cell) {
  return ((uint64_t)epoch << 44) | ((uint64_t)(uint32_t)b << 24) | (uint64_t)(uint32_t)cell;
}

void orc_source_count(const orc_mesh *M, const orc_params *P, int source_type, double dt,
                      int blocks_in_call, uint32_t epoch, int32_t *nper_block, int32_t *prefix) {
  const geom g = make_geom(M);
  const double npc =
      (double)P->num_particles / (double)g.ncell / (double)(blocks_in_call * M->nblocks);
  for (int b = 0; b < M->nblocks; ++b) {
    const double dv = M->blk_dx[3 * b] * M->blk_dx[3 * b + 1] * M->blk_dx[3 * b + 2];
    int run = 0, cell = 0;
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i, ++cell) {
          const int64_t c = cidx(&g, b, k, j, i);
          orc_rng rng = orc_rng_from_state(orc_rng_seed_state(
              (uint32_t)P->seed, ORC_RNG_DOMAIN_CELL, cell_stream_id(epoch, b, cell)));
          const double rho = M->rho[c];
          const double temp = eos_temperature(P, rho, M->sie[c]);
          double erad;
          if (source_type == ORC_SRC_THERMAL) {
            erad = (4.0 * P->sb / P->c) * orc_pow4(temp) * dv;
          } else {
            erad = M->fleck[c] * opac_emissivity(P, rho, temp) * dv * dt;
          }
          double snpc = floor(npc);
          snpc += (double)((npc - snpc) > orc_drand(&rng));
          M->src_num[c] = snpc;
          M->src_ew[c] = erad / snpc;
          prefix[(int64_t)b * g.ncell + cell] = run;
          run += (int)round(snpc);
        }
    nper_block[b] = run;
  }
}

void orc_source_fill(const orc_mesh *M, const orc_params *P, orc_swarm *S, int source_type,
                     double t_start, double dt, const int32_t *prefix, const int64_t *slot_base,
                     const uint64_t *id_base) {
  const geom g = make_geom(M);
  for (int b = 0; b < M->nblocks; ++b) {
    const double dx_i = M->blk_dx[3 * b], dx_j = M->blk_dx[3 * b + 1], dx_k = M->blk_dx[3 * b + 2];
    int cell = 0;
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i, ++cell) {
          const int64_t c = cidx(&g, b, k, j, i);
          const double xi = xc(M, b, 0, g.is, i);
          const double yi = xc(M, b, 1, g.js, j);
          const double zi = xc(M, b, 2, g.ks, k);
          const int pstart = prefix[(int64_t)b * g.ncell + cell];
          const int npart = (int)round(M->src_num[c]);
          const double rho = M->rho[c];
          const double temp = eos_temperature(P, rho, M->sie[c]);
          double dej = 0.0;
          for (int np = pstart; np < pstart + npart; ++np) {
            const int64_t n = slot_base[b] + np;
            orc_rng rng = orc_rng_from_state(
                orc_rng_stream_start((uint32_t)P->seed, id_base[b] + (uint64_t)np));
            S->ip[n] = i; S->jp[n] = j; S->kp[n] = k;
            S->blk[n] = b;
            S->status[n] = ORC_ST_ACTIVE;
            S->x[n] = xi + dx_i * (orc_drand(&rng) - 0.5);
            S->y[n] = yi + dx_j * (orc_drand(&rng) - 0.5);
            S->z[n] = zi + dx_k * (orc_drand(&rng) - 0.5);
            const double theta = orc_acos(2.0 * orc_drand(&rng) - 1.0);
            const double xi_phi = orc_drand(&rng);
            double sth, cth, sph, cph;
            orc_sincos(theta, &sth, &cth);
            orc_sincos2pi(xi_phi, &sph, &cph);
            S->vx[n] = P->c * sth * cph;
            S->vy[n] = P->c * sth * sph;
            S->vz[n] = P->c * cth;
            S->e[n] = orc_sample_planck_energy(&rng, P->sb, temp);
            S->w[n] = M->src_ew[c];
            if (source_type == ORC_SRC_EMISSION) {
              dej -= S->w[n];
              S->t[n] = t_start + orc_drand(&rng) * dt;
            } else {
              S->t[n] = 0.0;
            }
            S->id[n] = id_base[b] + (uint64_t)np;
            S->rng[n] = rng.s;
          }
          M->edelta[c] = dej;
        }
  }
}

/* ---------------------------------------------------------------------------------------- */
/* comm-phase pieces applied to one particle                                                 */
/* ---------------------------------------------------------------------------------------- */
/* a19 PhotonReflectBC (reference boundaries.hpp:46-82) + Parthenon's periodic / outflow swarm
 * boundaries; returns 0 if the particle left through an outflow face */
static int apply_swarm_bcs(const orc_mesh *M, double p[3], double v[3]) {
  for (int d = 0; d < M->ndim; ++d) {
    if (p[d] < M->gmin[d]) {
      const int bc = M->bc[2 * d];
      if (bc == ORC_BC_REFLECT) {
        p[d] = M->gmin[d] + (M->gmin[d] - p[d]);
        v[d] = -v[d];
      } else if (bc == ORC_BC_PERIODIC) {
        p[d] = M->gmax[d] - (M->gmin[d] - p[d]);
      } else {
        return 0;
      }
    }
    if (p[d] > M->gmax[d]) {
      const int bc = M->bc[2 * d + 1];
      if (bc == ORC_BC_REFLECT) {
        p[d] = M->gmax[d] - (p[d] - M->gmax[d]);
        v[d] = -v[d];
      } else if (bc == ORC_BC_PERIODIC) {
        p[d] = M->gmin[d] + (p[d] - M->gmax[d]);
      } else {
        return 0;
      }
    }
  }
  return 1;
}

/* destination block = the leaf containing the point (what GetNeighborBlockIndex + Send resolve) */
static int find_block(const orc_mesh *M, const double p[3]) {
  int l[3] = {0, 0, 0};
  for (int d = 0; d < M->ndim; ++d) {
    const double len = (M->gmax[d] - M->gmin[d]) / (double)M->nleaf[d];
    int q = (int)floor((p[d] - M->gmin[d]) * (1.0 / len));
    if (q < 0) q = 0;
    if (q > M->nleaf[d] - 1) q = M->nleaf[d] - 1;
    l[d] = q;
  }
  return M->leaf_map[((int64_t)l[2] * M->nleaf[1] + l[1]) * M->nleaf[0] + l[0]];
}

/* a14 SampleDDMCBlockFace for one particle (reference sample_ddmc_bface.cpp:119-424).
 * Axis triple (a, a1, a2) is cyclic: face normal a, transverse a1, a2. */
static void sample_block_face(const orc_mesh *M, const orc_params *P, const geom *g, int b,
                              orc_rng *rng, double p[3], double v[3], int ijk[3]) {
  const double eps = ORC_EPS;
  const double vv = P->c;
  if (!(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] < eps * vv * vv)) return;

  xtoijk(M, g, b, p[0], p[1], p[2], &ijk[0], &ijk[1], &ijk[2]);
  const int first[3] = {g->is, g->js, g->ks};
  double dxd[3], lo[3];
  for (int d = 0; d < 3; ++d) {
    dxd[d] = M->blk_dx[3 * b + d];
    lo[d] = xc(M, b, d, first[d], ijk[d]) - 0.5 * dxd[d];
  }
  const double *F[3] = {M->P1, M->P2, M->P3};

  for (int a = 0; a < M->ndim; ++a) {
    const int at_min = orc_fuzzy_equal(
        p[a], M->blk_xmin[3 * b + a] + 2.0 * ORC_EPS_DDMC * dxd[a], dxd[a], eps);
    const int at_max = orc_fuzzy_equal(
        p[a], M->blk_xmax[3 * b + a] - 2.0 * ORC_EPS_DDMC * dxd[a], dxd[a], eps);
    if (!(at_min || at_max)) continue;

    const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    const double dir_sgn = at_min ? 1.0 : -1.0;
    orc_sample_face_iso_dir(dir_sgn * vv, rng, &v[a], &v[a1], &v[a2]);
    int fidx[3] = {ijk[0], ijk[1], ijk[2]};
    fidx[a] = at_min ? ijk[a] : ijk[a] + 1;

    if (M->ndim == 2) {
      const int t = (a == 0) ? 1 : 0; /* the transverse in-plane axis */
      const int edge_u = orc_fuzzy_equal(p[t], lo[t], dxd[t], eps);
      const int edge_l = orc_fuzzy_equal(p[t], lo[t] + dxd[t], dxd[t], eps);
      if (edge_u || edge_l) {
        const int t_u = edge_u ? ijk[t] : ijk[t] + 1;
        const int t_l = edge_u ? ijk[t] - 1 : ijk[t];
        int iu[3] = {fidx[0], fidx[1], fidx[2]}, il[3] = {fidx[0], fidx[1], fidx[2]};
        iu[t] = t_u;
        il[t] = t_l;
        const double P_u = F[a][cidx(g, b, iu[2], iu[1], iu[0])];
        const double P_l = F[a][cidx(g, b, il[2], il[1], il[0])];
        orc_sample_face_2d(t_l, dxd[t], P_l, P_u, rng, &ijk[t], &p[t]);
      }
    } else {
      /* reference order of the two transverse axes: x-face (y,z), y-face (x,z) with the
       * probabilities indexed [z][x], z-face (x,y) -- see lines 318-331, 357-370, 396-409 */
      const int t1 = (a == 0) ? 1 : 0;
      const int t2 = (a == 2) ? 1 : 2;
      const int e1u = orc_fuzzy_equal(p[t1], lo[t1], dxd[t1], eps);
      const int e1l = orc_fuzzy_equal(p[t1], lo[t1] + dxd[t1], dxd[t1], eps);
      const int e2u = orc_fuzzy_equal(p[t2], lo[t2], dxd[t2], eps);
      const int e2l = orc_fuzzy_equal(p[t2], lo[t2] + dxd[t2], dxd[t2], eps);
      if ((e1u || e1l) && (e2u || e2l)) {
        const int t1_u = e1u ? ijk[t1] : ijk[t1] + 1, t1_l = e1u ? ijk[t1] - 1 : ijk[t1];
        const int t2_u = e2u ? ijk[t2] : ijk[t2] + 1, t2_l = e2u ? ijk[t2] - 1 : ijk[t2];
        double Pq[2][2]; /* [t2 lower/upper][t1 lower/upper] */
        for (int q2 = 0; q2 < 2; ++q2)
          for (int q1 = 0; q1 < 2; ++q1) {
            int id[3] = {fidx[0], fidx[1], fidx[2]};
            id[t1] = q1 ? t1_u : t1_l;
            id[t2] = q2 ? t2_u : t2_l;
            Pq[q2][q1] = F[a][cidx(g, b, id[2], id[1], id[0])];
          }
        /* SampleFace3D(i1_l, i2_l, d1, d2, P_ll, P_lu, P_ul, P_uu, ...) where the reference
         * names P_<t2><t1>: P_lu = (t2 lower, t1 upper) */
        orc_sample_face_3d(t1_l, t2_l, dxd[t1], dxd[t2], Pq[0][0], Pq[0][1], Pq[1][0], Pq[1][1],
                           rng, &ijk[t1], &ijk[t2], &p[t1], &p[t2]);
      }
    }
    break; /* reference: if / else-if chain over x, y, z faces */
  }
}

/* ---------------------------------------------------------------------------------------- */
/* a8 / a11: the history loop                                                                */
/* ---------------------------------------------------------------------------------------- */
static uint64_t history(const orc_mesh *M, const orc_params *P, const geom *g, orc_swarm *S,
                        int64_t n, double t_start, double dt, int64_t *abs_cell) {
  const int multi_d = (M->ndim >= 2), three_d = (M->ndim == 3);
  const double vv = P->c;
  orc_rng rng = orc_rng_from_state(S->rng[n]);
  int b = S->blk[n];
  double t = S->t[n];
  double p[3] = {S->x[n], S->y[n], S->z[n]};
  double v[3] = {S->vx[n], S->vy[n], S->vz[n]};
  const double ee = S->e[n];
  int ijk[3];
  int status = ORC_ST_ACTIVE;
  uint64_t nev = 0;
  *abs_cell = -1;

  xtoijk(M, g, b, p[0], p[1], p[2], &ijk[0], &ijk[1], &ijk[2]);

  while (t < t_start + dt) {
    ++nev;
    const double dx_i = M->blk_dx[3 * b], dx_j = M->blk_dx[3 * b + 1], dx_k = M->blk_dx[3 * b + 2];
    const double dx_push = orc_min(dx_i, orc_min(dx_j, dx_k));
    orc_step s;
    memset(&s, 0, sizeof s);
    s.t_start = t_start; s.dt = dt; s.vv = vv; s.dx_push = dx_push;
    s.multi_d = multi_d; s.three_d = three_d;
    s.xl = xc(M, b, 0, g->is, ijk[0]) - 0.5 * dx_i;
    s.xu = xc(M, b, 0, g->is, ijk[0]) + 0.5 * dx_i;
    s.yl = xc(M, b, 1, g->js, ijk[1]) - 0.5 * dx_j;
    s.yu = xc(M, b, 1, g->js, ijk[1]) + 0.5 * dx_j;
    s.zl = xc(M, b, 2, g->ks, ijk[2]) - 0.5 * dx_k;
    s.zu = xc(M, b, 2, g->ks, ijk[2]) + 0.5 * dx_k;

    const int64_t c = cidx(g, b, ijk[2], ijk[1], ijk[0]);
    const double rho = M->rho[c];
    const double temp = eos_temperature(P, rho, M->sie[c]);
    s.ff = M->fleck[c];
    s.ss = opac_scattering(P, rho, temp, ee);
    s.aa = opac_absorption(P, rho, temp, ee);
    s.t = t; s.x = p[0]; s.y = p[1]; s.z = p[2];
    s.vx = v[0]; s.vy = v[1]; s.vz = v[2];
    s.ip = ijk[0]; s.jp = ijk[1]; s.kp = ijk[2];

    const int is_ddmc_step = P->use_ddmc && (dx_push * (s.ss + s.aa) > P->tau_ddmc);
    if (is_ddmc_step) {
      /* reference transport_ddmc.cpp:137-179 (the repeated Xtoijk / bounds give the same cell) */
      s.Px_l = M->P1[c];
      s.Px_u = M->P1[cidx(g, b, ijk[2], ijk[1], ijk[0] + 1)];
      s.Py_l = multi_d ? M->P2[c] : 0.0;
      s.Py_u = multi_d ? M->P2[cidx(g, b, ijk[2], ijk[1] + 1, ijk[0])] : 0.0;
      s.Pz_l = three_d ? M->P3[c] : 0.0;
      s.Pz_u = three_d ? M->P3[cidx(g, b, ijk[2] + 1, ijk[1], ijk[0])] : 0.0;
      orc_ptcl_ddmc_albedo(&s, &rng);
      if (!s.is_rejected) orc_ptcl_ddmc_step(&s, &rng);
    } else {
      orc_ptcl_transport_step(&s, &rng);
    }
    t = s.t; p[0] = s.x; p[1] = s.y; p[2] = s.z;
    v[0] = s.vx; v[1] = s.vy; v[2] = s.vz;

    xtoijk(M, g, b, p[0], p[1], p[2], &ijk[0], &ijk[1], &ijk[2]);

    if (!on_block(g, ijk[0], ijk[1], ijk[2])) {
      if (P->use_ddmc) { /* transport_ddmc.cpp:203-211: zero velocity flags a DDMC leak */
        const double vmask = (double)!(is_ddmc_step && multi_d && !s.is_rejected);
        v[0] *= vmask; v[1] *= vmask; v[2] *= vmask;
      }
      /* --- comm phase, inline --- */
      if (!apply_swarm_bcs(M, p, v)) {
        status = ORC_ST_ESCAPED;
        break;
      }
      b = find_block(M, p);
      if (P->use_ddmc && multi_d) sample_block_face(M, P, g, b, &rng, p, v, ijk);
      /* next TransportPhotons launch starts with Xtoijk (transport.cpp:96) */
      xtoijk(M, g, b, p[0], p[1], p[2], &ijk[0], &ijk[1], &ijk[2]);
      continue;
    }
    if (s.is_absorbed) {
      *abs_cell = cidx(g, b, ijk[2], ijk[1], ijk[0]);
      status = ORC_ST_ABSORBED;
      break;
    }
    if (s.is_scattered) orc_scatter(&rng, vv, &v[0], &v[1], &v[2]);
  }

  S->blk[n] = b;
  S->t[n] = t;
  S->x[n] = p[0]; S->y[n] = p[1]; S->z[n] = p[2];
  S->vx[n] = v[0]; S->vy[n] = v[1]; S->vz[n] = v[2];
  S->ip[n] = ijk[0]; S->jp[n] = ijk[1]; S->kp[n] = ijk[2];
  S->status[n] = status;
  S->rng[n] = rng.s;
  return nev;
}

uint64_t orc_transport_photons(const orc_mesh *M, const orc_params *P, orc_swarm *S,
                               double t_start, double dt, int64_t first, int64_t last) {
  const geom g = make_geom(M);
  const int64_t cnt = last - first;
  if (cnt <= 0) return 0;
  int64_t *abs_cell = (int64_t *)malloc(sizeof(int64_t) * (size_t)cnt);
  uint64_t nev = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nev) num_threads(orc_threads)
#endif
  for (int64_t n = first; n < last; ++n) {
    abs_cell[n - first] = -1;
    if (S->status[n] != ORC_ST_ACTIVE) continue;
    nev += history(M, P, &g, S, n, t_start, dt, &abs_cell[n - first]);
  }
  /* Kokkos::atomic_add(&energy_delta, w) (transport.cpp:159-160), in particle order */
  for (int64_t n = first; n < last; ++n)
    if (abs_cell[n - first] >= 0) M->edelta[abs_cell[n - first]] += S->w[n];
  free(abs_cell);
  return nev;
}

/* a16 CheckCompletion (reference transport.cpp:187-216) */
int64_t orc_check_completion(const orc_swarm *S, double t_end) {
  int64_t unfinished = 0;
  for (int64_t n = 0; n < S->n; ++n)
    if (S->status[n] == ORC_ST_ACTIVE && S->t[n] < t_end) ++unfinished;
  return unfinished;
}

/* a17 EvaluateRadiationEnergy (reference jaybenne.cpp:514-564) */
void orc_evaluate_radiation_energy(const orc_mesh *M, const orc_swarm *S) {
  const geom g = make_geom(M);
  for (int b = 0; b < M->nblocks; ++b)
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) M->tally[cidx(&g, b, k, j, i)] = 0.0;
  for (int64_t n = 0; n < S->n; ++n) {
    if (S->status[n] != ORC_ST_ACTIVE) continue;
    const int b = S->blk[n];
    const double dv = M->blk_dx[3 * b] * M->blk_dx[3 * b + 1] * M->blk_dx[3 * b + 2];
    M->tally[cidx(&g, b, S->kp[n], S->jp[n], S->ip[n])] += S->w[n] / dv;
  }
}

/* a18 UpdateFluid (reference jaybenne.cpp:583-615) */
void orc_update_fluid(const orc_mesh *M, const orc_params *P) {
  if (!P->do_feedback) return;
  const geom g = make_geom(M);
  for (int b = 0; b < M->nblocks; ++b) {
    const double dv = M->blk_dx[3 * b] * M->blk_dx[3 * b + 1] * M->blk_dx[3 * b + 2];
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          const int64_t c = cidx(&g, b, k, j, i);
          const double delta = M->edelta[c] / dv;
          M->u[c] += delta;
        }
  }
}

/* a19 PhotonReflectBC<face> as a stand-alone task (reference boundaries.hpp:24-84) */
void orc_photon_reflect_bc(const orc_mesh *M, orc_swarm *S, int face) {
  const geom g = make_geom(M);
  const int d = face / 2, outer = face & 1;
  double *pos[3] = {S->x, S->y, S->z};
  double *vel[3] = {S->vx, S->vy, S->vz};
  for (int64_t n = 0; n < S->n; ++n) {
    if (S->status[n] != ORC_ST_ACTIVE) continue;
    double *q = &pos[d][n];
    int hit = 0;
    if (!outer && *q < M->gmin[d]) {
      *q = M->gmin[d] + (M->gmin[d] - *q);
      hit = 1;
    } else if (outer && *q > M->gmax[d]) {
      *q = M->gmax[d] - (*q - M->gmax[d]);
      hit = 1;
    }
    if (hit) {
      vel[d][n] = -vel[d][n];
      xtoijk(M, &g, S->blk[n], S->x[n], S->y[n], S->z[n], &S->ip[n], &S->jp[n], &S->kp[n]);
    }
  }
}

/* a14 as a stand-alone task over all active particles */
void orc_sample_ddmc_block_face(const orc_mesh *M, const orc_params *P, orc_swarm *S) {
  if (!(M->ndim > 1)) return;
  const geom g = make_geom(M);
  for (int64_t n = 0; n < S->n; ++n) {
    if (S->status[n] != ORC_ST_ACTIVE) continue;
    orc_rng rng = orc_rng_from_state(S->rng[n]);
    double p[3] = {S->x[n], S->y[n], S->z[n]};
    double v[3] = {S->vx[n], S->vy[n], S->vz[n]};
    int ijk[3] = {S->ip[n], S->jp[n], S->kp[n]};
    sample_block_face(M, P, &g, S->blk[n], &rng, p, v, ijk);
    S->x[n] = p[0]; S->y[n] = p[1]; S->z[n] = p[2];
    S->vx[n] = v[0]; S->vy[n] = v[1]; S->vz[n] = v[2];
    S->ip[n] = ijk[0]; S->jp[n] = ijk[1]; S->kp[n] = ijk[2];
    S->rng[n] = rng.s;
  }
}

/* RemoveMarkedParticles (reference transport.cpp:176-178): stable compaction */
int64_t orc_remove_marked(orc_swarm *S) {
  int64_t m = 0;
  for (int64_t n = 0; n < S->n; ++n) {
    if (S->status[n] != ORC_ST_ACTIVE) continue;
    if (m != n) {
      S->x[m] = S->x[n]; S->y[m] = S->y[n]; S->z[m] = S->z[n];
      S->vx[m] = S->vx[n]; S->vy[m] = S->vy[n]; S->vz[m] = S->vz[n];
      S->t[m] = S->t[n]; S->w[m] = S->w[n]; S->e[m] = S->e[n];
      S->ip[m] = S->ip[n]; S->jp[m] = S->jp[n]; S->kp[m] = S->kp[n];
      S->blk[m] = S->blk[n]; S->status[m] = S->status[n];
      S->id[m] = S->id[n]; S->rng[m] = S->rng[n];
    }
    ++m;
  }
  S->n = m;
  return m;
}
