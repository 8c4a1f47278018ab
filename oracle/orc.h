/* oracle/orc.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle for the Implicit-Monte-Carlo history loop of lanl/jaybenne: a plain-C restatement
 * of the reference's algorithm (the .cpp / .hpp files under src/jaybenne).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (jaybenne_amd/) never does.
 *
 * PINNING.  The reference holds no golden vectors and no unit tests for this path; its only
 * known-answer tests are tst/stepdiff.py (erf profile, weighted mean fractional error <= 0.05)
 * and tst/stepdiff_smr.py (<= 0.3).  The oracle is pinned against exactly those
 * (tests/test_oracle_physics.py), and its generator against the Random123 Philox known-answer
 * vectors.  The reference cannot be compiled here: every translation unit, including the
 * header-only step functions, includes <parthenon/...> and Kokkos headers, which are empty
 * submodules in /root/reference, and building it against hand-written stand-ins for those
 * headers is not a reference build.  Individual uniforms (Kokkos XorShift64 pool) and Parthenon
 * geometry helpers (Xtoijk, robust::EPS, neighbour lookup) are therefore restated from their call
 * sites and are "parity unpinned" at the bit level.
 */
#ifndef ORC_H_
#define ORC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BC_PERIODIC = 0, ORC_BC_REFLECT = 1, ORC_BC_OUTFLOW = 2 };
enum { ORC_SRC_THERMAL = 0, ORC_SRC_EMISSION = 1 };
enum { ORC_ST_ACTIVE = 0, ORC_ST_ABSORBED = 1, ORC_ST_ESCAPED = 2 };

/* <jaybenne> parameters (reference jaybenne.cpp:158-266) + the host's EOS / opacity models
 * (reference mcblock.cpp:78-145: IdealGas, Gray, GrayS) */
typedef struct orc_params {
  int64_t num_particles;
  double dt;
  double tau_ddmc;
  double c, sb;        /* speed of light, Stefan-Boltzmann (opacity.GetRuntimePhysicalConstants) */
  double cv;           /* IdealGas: T = sie / cv */
  double kappa_a;      /* Gray: sigma_a = rho * kappa_a ; j = sigma_a * 4 sb T^4 */
  double kappa_s, apm; /* GrayS: sigma_s = (rho / apm) * kappa_s */
  int32_t seed;
  int32_t use_ddmc;
  int32_t do_emission;
  int32_t do_feedback;
  /* absorption model: 0 Gray (kappa_a), 1 electron-proton bremsstrahlung in code units,
   * sigma_a = A rho^2 T^-1/2 (1 - e^(-B nu / T)) nu^-3, j = E rho^2 T^1/2 (mcblock.cpp:108-113
   * selects singularity-opac's EPBremss, which is not vendored: the published free-free formulas
   * of Rybicki & Lightman 5.18a / 5.15a with Gaunt factor 1 stand in; see orc_model_coefficients) */
  int32_t opac_model, pad_model;
  double ep_A, ep_B, ep_E;
} orc_params;

/* Whole mesh (the oracle is single-process).  All cell/face fields are block-major arrays
 * [nblocks][nk][nj][ni] including ng ghost layers in active dimensions. */
typedef struct orc_mesh {
  int32_t ndim, ng, nblocks, pad0;
  int32_t nx[3];    /* interior cells per block */
  int32_t nleaf[3]; /* leaf-map extent (blocks at the finest level) */
  int32_t bc[6];    /* swarm BC per face: ix1, ox1, ix2, ox2, ix3, ox3 */
  double gmin[3], gmax[3];
  const int32_t *leaf_map;    /* [nleaf2][nleaf1][nleaf0] -> block */
  const double *blk_xmin;     /* [nblocks][3] */
  const double *blk_xmax;     /* [nblocks][3] */
  const double *blk_dx;       /* [nblocks][3]; inactive dims hold the full extent */
  const int32_t *blk_level;   /* [nblocks] */
  const int32_t *blk_nbr_lev; /* [nblocks][6]; own level at physical boundaries */
  double *rho, *sie, *u, *fleck, *tally, *edelta, *src_ew, *src_num, *P1, *P2, *P3;
} orc_mesh;

typedef struct orc_swarm {
  int64_t n, cap;
  double *x, *y, *z, *vx, *vy, *vz, *t, *w, *e;
  int32_t *ip, *jp, *kp, *blk, *status;
  uint64_t *id;  /* creation index (diagnostic key) */
  uint64_t *rng; /* LCG state of the particle's stream */
} orc_swarm;

void orc_set_math_mode(int mode); /* 0 = libm (reference arithmetic), 1 = portable spec */
int orc_get_math_mode(void);
void orc_set_threads(int n);
int orc_get_threads(void);

/* generator + math, vectorised for tests */
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint64_t orc_seed_state(uint32_t seed, uint32_t domain, uint64_t id);
uint64_t orc_stream_start(uint32_t seed, uint64_t id); /* first state of particle id's stream */
uint64_t orc_draw_stream(uint64_t state, int n, double *out); /* returns the final state */
void orc_math_log(const double *x, int n, double *out);
void orc_math_sincos(const double *x, int n, double *sn, double *cs);
void orc_math_sincos2pi(const double *u, int n, double *sn, double *cs); /* of 2 pi u */
void orc_math_acos(const double *x, int n, double *out);
void orc_math_one_minus_exp_neg(const double *x, int n, double *out);

/* step functions driven by a tape of uniforms; `st` is an orc_step (orc_steps.h) laid out as
 * doubles/ints exactly as declared there.  Returns the number of uniforms consumed. */
int orc_call_transport_step(void *st, const double *tape, int ntape);
int orc_call_ddmc_step(void *st, const double *tape, int ntape);
int orc_call_ddmc_albedo(void *st, const double *tape, int ntape);
/* EPBremss A, B, E and the ThomsonS-as-GrayS kappa_s for the code -> CGS scales {time, mass,
 * length, temperature} (mcblock.cpp:84-91) */
void orc_model_coefficients(const double scales[4], double out[4]);
/* which: 0 absorption(rho, T, nu), 1 emissivity(rho, T), 2 scattering(rho, T, nu); x = n triples */
void orc_model_eval(const orc_params *P, int which, const double *x, int n, double *out);
int orc_call_scatter(double vv, const double *tape, int ntape, double v[3]);
int orc_call_face_iso_dir(double vv, const double *tape, int ntape, double v[3]);
int orc_call_planck(double sb, double temp, const double *tape, int ntape, double *e);
int orc_call_face_2d(int i_l, double dx, double P_l, double P_u, const double *tape, int ntape,
                     int *i, double *x);
int orc_call_face_3d(int i1_l, int i2_l, double dx1, double dx2, const double P[4],
                     const double *tape, int ntape, int ij[2], double x12[2]);
int orc_sizeof_step(void);

/* task-level restatements (names follow reference jaybenne.hpp:59-76) */
void orc_update_derived_transport_fields(const orc_mesh *M, const orc_params *P, double dt);
/* phase 1 of SourcePhotons: per-cell counts/weights, per-block totals and per-cell exclusive
 * prefix (prefix: [nblocks][ncells_interior]).  blocks_in_call reproduces the reference's
 * npc = N / cells / (nblocks * nbtotal). */
void orc_source_count(const orc_mesh *M, const orc_params *P, int source_type, double dt,
                      int blocks_in_call, uint32_t epoch, int32_t *nper_block, int32_t *prefix);
/* phase 2: fill particles; slot_base/id_base per block */
void orc_source_fill(const orc_mesh *M, const orc_params *P, orc_swarm *S, int source_type,
                     double t_start, double dt, const int32_t *prefix, const int64_t *slot_base,
                     const uint64_t *id_base);
/* full histories for particles [first,last): TransportPhotons / TransportPhotons_DDMC including
 * the comm phase (boundary conditions, neighbour block, SampleDDMCBlockFace) applied inline.
 * Returns the number of loop passes (events). */
uint64_t orc_transport_photons(const orc_mesh *M, const orc_params *P, orc_swarm *S,
                               double t_start, double dt, int64_t first, int64_t last);
int64_t orc_check_completion(const orc_swarm *S, double t_end);
void orc_evaluate_radiation_energy(const orc_mesh *M, const orc_swarm *S);
void orc_update_fluid(const orc_mesh *M, const orc_params *P);
void orc_photon_reflect_bc(const orc_mesh *M, orc_swarm *S, int face);
void orc_sample_ddmc_block_face(const orc_mesh *M, const orc_params *P, orc_swarm *S);
int64_t orc_remove_marked(orc_swarm *S);

#ifdef __cplusplus
}
#endif
#endif /* ORC_H_ */
