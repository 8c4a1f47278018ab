/* jaybenne_amd.h -- C ABI of the MI355X-native Implicit Monte Carlo history loop.
 *
 * This is the drop-in boundary for the hot path of lanl/jaybenne: one entry point per task the
 * reference package exposes to its host application (reference src/jaybenne/jaybenne.hpp:48-78),
 * over plain pointers and sizes.  The host framework owns every field and particle array (as
 * Parthenon does for the reference); the library borrows DEVICE pointers per call and owns only
 * its parameters, small lookup tables and scratch.  Nothing here throws; every function returns
 * a jb_status and jb_last_error() describes the most recent failure on the calling thread.
 *
 * Threading: calls on one jb_context are not re-entrant (the reference's tasks keep
 * function-local static pack descriptors, transport.cpp:44-52, and require one partition per
 * rank, jaybenne.cpp:92-95).  Work is enqueued on the context's HIP stream (jb_set_stream);
 * functions that return counts to the host synchronise that stream, the others do not.
 *
 * Index conventions (Parthenon's): a block's cell arrays are [nk][nj][ni]
 * with i fastest, ni = nx[0] + 2 ng etc. in active dimensions and a single index in inactive
 * ones; interior cells are ng .. ng+nx-1.  Face field F_d shares the cell layout: face i is
 * the lower-x_d face of cell i (reference transport_ddmc.cpp:150-159), so ng >= 1 is required.
 */
#ifndef JAYBENNE_AMD_H_
#define JAYBENNE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TaskStatus of the reference tasks (jaybenne.cpp:34,55-56; transport.cpp:211-215) + errors */
typedef enum jb_status {
  JB_COMPLETE = 0,
  JB_ITERATE = 1,
  JB_INCOMPLETE = 2,
  JB_ERR_INVALID = -1,   /* PARTHENON_REQUIRE / PARTHENON_FAIL conditions of the reference */
  JB_ERR_HIP = -2,
  JB_ERR_CAPACITY = -3,
  JB_ERR_UNSUPPORTED = -4
} jb_status;

enum { JB_BC_PERIODIC = 0, JB_BC_REFLECT = 1, JB_BC_OUTFLOW = 2 }; /* <parthenon/swarm> i/ox?_bc */
enum { JB_SOURCE_THERMAL = 0, JB_SOURCE_EMISSION = 1 };  /* SourceType, jaybenne.hpp:56 */
enum { JB_STRATEGY_UNIFORM = 0, JB_STRATEGY_ENERGY = 1 }; /* SourceStrategy, jaybenne.hpp:55 */
enum { JB_EOS_IDEAL_GAS = 0 };
enum { JB_OPAC_GRAY = 0, JB_OPAC_EPBREMSS = 1 };
enum { JB_SCAT_GRAY = 0, JB_SCAT_THOMSON = 1 };
/* particle status written by the transport tasks */
enum {
  JB_ST_ACTIVE = 0,   /* resident on this device (in flight before, at census after transport) */
  JB_ST_ABSORBED = 1, /* MarkParticleForRemoval after an absorption (transport.cpp:157-163) */
  JB_ST_ESCAPED = 2,  /* left through an outflow swarm boundary */
  JB_ST_OUTGOING = 3, /* to be handed to the rank that owns block `blk` (a GLOBAL id): the
                         particle left the resident blocks, or reached census in a halo copy */
  JB_ST_OUTGOING_ABSORBED = 4 /* absorbed inside a halo copy: the owner deposits its weight */
};

/* <jaybenne> input block, keys and defaults of reference jaybenne.cpp:163-223 */
typedef struct jb_params {
  int64_t num_particles; /* required */
  double dt;             /* default DBL_MAX */
  double min_swarm_occupancy;
  double numin, numax;   /* parsed, unused */
  double tau_ddmc;       /* default 5.0 */
  int32_t unique_rank_seeds;
  int32_t seed;          /* default 123 */
  int32_t max_transport_iterations; /* default 10000 */
  int32_t use_ddmc;
  int32_t source_strategy;
  int32_t do_emission;
  int32_t do_feedback;
  int32_t rank;          /* Globals::my_rank, used with unique_rank_seeds */
} jb_params;

/* the host's EOS / opacity / scattering objects (reference jaybenne.hpp:50-52,
 * jaybenne_config.hpp.in:19-26; mcblock.cpp:78-145), as tagged POD evaluated device-side */
typedef struct jb_eos {
  int32_t model, pad;
  double gm1, cv;        /* IdealGas(gm1, cv): T = sie / cv */
} jb_eos;
typedef struct jb_opacity {
  int32_t model, pad;
  double kappa;          /* Gray: sigma_a = rho kappa, j = sigma_a 4 sb T^4 */
  double c, sb;          /* GetRuntimePhysicalConstants(): speed of light, Stefan-Boltzmann,
                            in code units */
  /* NonCGSUnits<...>(opac, time, mass, length, temperature): code -> CGS conversion factors
   * (mcblock.cpp:84-91); read by JB_OPAC_EPBREMSS only (Gray takes kappa in code units):
   * electron-proton bremsstrahlung of fully ionised hydrogen, n_e = n_i = rho / m_p,
   *   alpha_nu = 4 e^6 / (3 m_e h c) sqrt(2 pi / (3 k m_e)) T^-1/2 n_e n_i nu^-3 (1 - e^(-h nu / k T))
   *   j        = sqrt(2 pi k T / (3 m_e)) 2^5 pi e^6 / (3 h m_e c^3) n_e n_i
   * (Rybicki & Lightman 5.18a, 5.15a, Gaunt factor 1) -- singularity-opac is not vendored in the
   * reference tree, so its constants could not be compared: parity unpinned for this model. */
  double time_scale, mass_scale, length_scale, temperature_scale;
} jb_opacity;
typedef struct jb_scattering {
  int32_t model, pad;
  double kappa_s, apm;   /* GrayS: sigma_s = (rho / apm) kappa_s; apm also read by ThomsonS */
  /* JB_SCAT_THOMSON: sigma_s = n_e sigma_T with n_e = rho / apm (apm in code mass units, as
   * mcblock.cpp:124 passes it), i.e. GrayS with kappa_s = sigma_T / length_scale^2 */
  double time_scale, mass_scale, length_scale, temperature_scale;
} jb_scattering;

/* The slice of the mesh this rank owns (the MeshData / SparsePack role).  All pointers in this
 * struct are HOST pointers; rho..P3 are host arrays of per-block DEVICE pointers. */
typedef struct jb_mesh_view {
  int32_t ndim, ng;
  int32_t nblocks;        /* blocks resident on this rank: the ones it owns + halo copies */
  int32_t nblocks_total;  /* Mesh::nbtotal */
  int32_t nx[3];          /* interior cells per block */
  int32_t nleaf[3];       /* extent of leaf_map (blocks of the finest level) */
  int32_t bc[6];          /* swarm boundary per face: ix1, ox1, ix2, ox2, ix3, ox3 */
  int32_t rank, pad;
  double gmin[3], gmax[3];
  const int32_t *leaf_map;     /* [nleaf2][nleaf1][nleaf0] -> global block id */
  const int32_t *owner;        /* [nblocks_total] rank that owns each global block */
  const int32_t *local_index;  /* [nblocks_total] index in this rank's resident list, or -1 */
  const int32_t *gid;          /* [nblocks] global id of each resident block */
  const int32_t *owned;        /* [nblocks] 1 = owned by this rank, 0 = halo copy: a read-only
                                  mirror of a neighbour rank's block (fields filled by the host)
                                  that lets particles be tracked across the rank boundary; may be
                                  NULL = every resident block is owned */
  const double *blk_xmin;      /* [nblocks][3] */
  const double *blk_xmax;      /* [nblocks][3] */
  const double *blk_dx;        /* [nblocks][3] (inactive dimensions: full extent) */
  const int32_t *blk_level;    /* [nblocks] */
  const int32_t *blk_nbr_lev;  /* [nblocks][6] neighbour level per face; own level at a
                                  physical boundary (jaybenne.cpp:341-351) */
  /* HOST_DENSITY, HOST_SPECIFIC_INTERNAL_ENERGY, HOST_UPDATE_ENERGY (jaybenne_config.hpp.in:28-30) */
  double *const *rho, *const *sie, *const *u;
  /* field.jaybenne.* (jaybenne_variables.hpp:35-40) */
  double *const *fleck, *const *tally, *const *edelta, *const *src_ew, *const *src_num;
  double *const *P1, *const *P2, *const *P3; /* ddmc_face_prob F1/F2/F3; may be NULL w/o DDMC */
} jb_mesh_view;

/* The photons swarm (jaybenne.cpp:236-245): structure of arrays, DEVICE pointers, particles
 * 0..n-1 valid.  rng is the state of each particle's own random stream (csrc/jb_rng.hpp); id is
 * the particle's creation index, which seeded that stream. */
typedef struct jb_swarm_view {
  int64_t n, capacity;
  double *x, *y, *z;      /* swarm_position::x,y,z */
  double *vx, *vy, *vz;   /* particle.photons.v[3] */
  double *t, *w, *e;      /* time, weight, energy */
  int32_t *ip, *jp, *kp;  /* particle.photons.ijk[3] */
  int32_t *blk;           /* local block index (global id when status == JB_ST_OUTGOING) */
  int32_t *status;
  uint64_t *id;
  uint64_t *rng;
} jb_swarm_view;

typedef struct jb_transport_stats {
  int64_t n_census, n_absorbed, n_escaped, n_outgoing;
  int64_t n_events;  /* passes through the while loop of transport.cpp:98-171 */
  int64_t n_wave_passes;    /* diagnostics: 64-lane passes through the kernel's event loop ... */
  int64_t n_wave_services;  /* ... and through its service phase, summed over waves */
} jb_transport_stats;

typedef struct jb_context jb_context;
typedef struct jb_mesh jb_mesh;

const char *jb_last_error(void);
const char *jb_version(void);

/* jaybenne::Initialize(pin, opacity, scattering, eos) -- jaybenne.hpp:50-52, jaybenne.cpp:158-266 */
jb_status jb_initialize(const jb_params *params, const jb_eos *eos, const jb_opacity *opacity,
                        const jb_scattering *scattering, int device, jb_context **ctx);
jb_status jb_finalize(jb_context *ctx);
jb_status jb_set_stream(jb_context *ctx, void *hip_stream);
jb_status jb_synchronize(jb_context *ctx);
/* Param<int>("seed"): seed + rank when unique_rank_seeds (jaybenne.cpp:187-190).  The random
 * streams are keyed with the UNADJUSTED seed, as the reference's pool is (quirk 1). */
int32_t jb_param_seed(const jb_context *ctx);

/* uploads the lookup tables of a mesh view; the view's host arrays may be freed afterwards */
jb_status jb_mesh_create(jb_context *ctx, const jb_mesh_view *view, jb_mesh **mesh);
jb_status jb_mesh_destroy(jb_mesh *mesh);

/* UpdateDerivedTransportFields(md, dt) -- jaybenne.hpp:66, jaybenne.cpp:285-492 */
jb_status jb_update_derived_transport_fields(jb_context *ctx, jb_mesh *mesh, double dt);

/* SourcePhotons<T,ST>(md, t_start, dt) -- jaybenne.hpp:63-64, sourcing.cpp:25-208, split at the
 * host round-trip of sourcing.cpp:120-131:
 *   _count  "SourcePhotons1": per-cell number / energy weight, per-block totals (to the host)
 *           and per-cell exclusive prefix (device int32 [nblocks][ncells]); blocks_in_call is
 *           the `nblocks` of sourcing.cpp:68-69 (1 on the MeshBlockData path);
 *   _fill   "SourcePhotons2": attributes of the nper_block[b] new particles of each block;
 *           block b writes slots slot_base[b].. and stream ids id_base[b]..  (host arrays
 *           [nblocks]; the slot assignment is Parthenon's AddEmptyParticles step).
 * Returns JB_COMPLETE without doing anything for emission when do_emission is false
 * (sourcing.cpp:41-43); JB_ERR_INVALID for the `energy` strategy (sourcing.cpp:38) and for
 * epoch >= 2^20 (the per-cell rounding streams are keyed by 20 bits of it). */
jb_status jb_source_photons_count(jb_context *ctx, jb_mesh *mesh, int source_type, double dt,
                                  int blocks_in_call, uint32_t epoch, int32_t *nper_block_host,
                                  int32_t *prefix_dev);
jb_status jb_source_photons_fill(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                 int source_type, double t_start, double dt,
                                 const int32_t *nper_block_host, const int32_t *prefix_dev,
                                 const int64_t *slot_base_host, const uint64_t *id_base_host);
/* ... and the same for a SHARE of every block's new particles: of the photons _count found for block b
 * (numbered 0.. in cell order, the order the stream ids follow) this call creates nper_block[b] of
 * them starting at number first_in_block[b]; id_base[b] stays the id of the block's photon 0.  This is
 * what a rank of a replicated-mesh run calls (every rank holds every block and sources its share of
 * each: SURVEY 8e "replicated mesh, split particles"; the reference has no such mode -- its blocks
 * are partitioned, jaybenne.cpp:92-95): histories are then dealt to ranks by stream id, whatever the
 * blocks cost.  set_energy_delta = 0 makes the emission source leave energy_delta at zero instead of
 * minus the emitted energy (sourcing.cpp:165-166,196): all ranks but one, so that the sum of the
 * ranks' energy_delta carries the emission once.  first_in_block_host = NULL: all of them, from 0. */
jb_status jb_source_photons_fill_range(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                       int source_type, double t_start, double dt,
                                       const int32_t *nper_block_host, const int32_t *prefix_dev,
                                       const int64_t *slot_base_host, const uint64_t *id_base_host,
                                       const int32_t *first_in_block_host, int set_energy_delta);

/* TransportPhotons / TransportPhotons_DDMC(md, t_start, dt) -- jaybenne.hpp:59-60,
 * transport.cpp:28-181, transport_ddmc.cpp:28-237.  Particles [first,last) with status ACTIVE
 * are followed until census, absorption, escape, or until they enter a block that is not
 * resident on this rank (status OUTGOING).  Crossing into a resident block -- owned or a halo
 * copy -- does not end the launch: the swarm
 * boundary conditions (boundaries.hpp:46-82, periodic, outflow), the destination-block lookup
 * and SampleDDMCBlockFace are applied to the particle in flight.  With fuse_census_tally != 0
 * a particle reaching census in an OWNED block adds weight / cell volume to energy_tally
 * (EvaluateRadiationEnergy fused; the caller zeroes the tally first with jb_zero_energy_tally).
 * A particle that reaches census or is absorbed in a HALO copy is marked OUTGOING /
 * OUTGOING_ABSORBED: its owner tallies it after the hand-off.
 * The calls return when the kernel is launched (jb_synchronize / jb_get_transport_stats wait
 * for it), with one exception: jb_transport_photons_ddmc with gray opacities reads one flag back
 * first (does every cell take DDMC steps?) and, on a mesh that mixes IMC and DDMC cells, runs
 * three launches with the size of a work list read back before the second and the third. */
jb_status jb_transport_photons(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                               double t_start, double dt, int64_t first, int64_t last,
                               int fuse_census_tally);
jb_status jb_transport_photons_ddmc(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                    double t_start, double dt, int64_t first, int64_t last,
                                    int fuse_census_tally);
/* Arithmetic of the tracking step of the gray IMC kernels (the headline path).
 *   JB_ARITH_EXACT: every operation is the one the CPU oracle's portable flavour performs --
 *     correctly rounded quotients, the reference's unfused position update -- and the particles
 *     come out bit-identical to it.
 *   JB_ARITH_LEAN (default): while a lane follows a photon it carries the unit direction v / c and
 *     the distance left to census c (t_end - t) instead of v and t, and the position RELATIVE TO
 *     THE CENTRE OF ITS CELL instead of x, with the cell as one byte offset into the per-cell arrays
 *     (k_imc_cell; any cell widths: only the conversion from and to the
 *     swarm's coordinates, when a photon is loaded and written back, depends on them -- exact where
 *     they are powers of two, jb_mesh_exact_geometry, an ulp of the position elsewhere; the
 *     per-cell arrays of the resident blocks must lie within 4 GiB); distance to a face as
 *     (h - sgn(omega) p) / |omega| with a once-refined reciprocal of the direction component (within
 *     2^-48, ~20 ulp, of the correctly rounded quotient; in 3-D the three reciprocals come from one,
 *     of the product of the components, to the same accuracy), position update as one fused
 *     multiply-add per axis, logarithm without its compensated sum (<= 3 ulp), square root of
 *     1 - mu^2 with one residual correction (<= 2 ulp) -- every operation within 4e-15 (relative) of
 *     the exact variant's, the cell-local position carrying ~8 bits MORE than the absolute one the
 *     exact variant (and the reference) round at every event.  Both faces of an axis are tested for
 *     the nudge of transport_utils.hpp:151-159, as in the reference.  (More than 4 GiB of per-cell
 *     arrays, or JB_NO_IMC_CELL=1: the x-space form of round 3, which tests only the face ahead.)
 *     ~40 % fewer instructions per event than the exact variant.
 *     In one line: 1e-9 PER CYCLE; measured growth about one decade per cycle, 1e-4 of the domain and
 *     <= 1e-3 of the histories re-sequenced after ten cycles (the same as the CPU path's libm vs portable
 *     arithmetic); the tally within 6 sigma per cell at any length.  bench.py measures these in every
 *     default run (accuracy.lean_vs_exact_after_10_cycles).
 *     Stated tolerance (tests/test_gpu_lean.py, tests/test_gpu_accuracy.py):
 *       - after ONE full cycle every floating-point attribute of every photon is within 1e-9 of the
 *         exact variant's and of the oracle's (positions relative to the domain size, velocities to
 *         c, times to dt), after TWO within 1e-8; integer attributes and stream states equal;
 *       - beyond that the two roundings of a history separate like nearby trajectories of any
 *         chaotic system -- measured: largest position difference over 1e5 photons 5e-12 of the
 *         domain after one cycle, 1.3e-9 after two, then about a decade per cycle, 1e-4 after ten
 *         (the oracle's own libm / portable flavours, <= 1 ulp apart in log and sincos, separate
 *         at the same rate; what the lean variant differs by from the exact one is mostly the exact
 *         variant's own rounding of x to the absolute grid) -- and a photon changes its sequence of
 *         events only where a difference flips a comparison: 20 of 1e5 photons in ten cycles;
 *       - at any length: the energy tally within 6 sigma of each cell's Monte Carlo noise of the
 *         libm-arithmetic CPU path on the same streams (measured 3e-13 sigma on BASELINE
 *         configs[0], 0.016 sigma per x-plane on configs[1]'s geometry) and the reference's error
 *         metric against the analytic profile not worse than that path's + 0.01.
 * The IMC steps of a hybrid (IMC / DDMC) deck follow the same switch -- on a mesh without the
 * exact geometry (jb_mesh_exact_geometry) testing BOTH faces per axis, as the reference does: there
 * a photon that leaked from a coarse DDMC cell into a finer block can be left, unresampled and
 * with zero velocity, on a face of the fine cells (sample_ddmc_bface.cpp's fuzzy test, tolerance
 * 2e-16 dx, missing it) -- ; DDMC steps and the per-event-opacity kernels have the exact arithmetic
 * only.  JB_EXACT_ARITH=1
 * in the environment makes exact the default of jb_initialize. */
enum { JB_ARITH_EXACT = 0, JB_ARITH_LEAN = 1 };
jb_status jb_set_arithmetic(jb_context *ctx, int mode);
int jb_get_arithmetic(const jb_context *ctx);
/* the kernel the last transport call on this mesh launched: "k_transport<3, true, 2, true, true>" =
 * <NDIM, TALLY (census tally fused), GRAY (0 per-event opacities, 1 gray, 2 gray without
 * absorption), EXACT (exact cell-face arithmetic, 32-bit cell offsets), LEAN (lean arithmetic)>;
 * "k_ddmc_all<3, true>" = <NDIM, TALLY> on a mesh whose every cell takes DDMC steps (", quad gather"
 * appended when its cell records, more than 1 MiB of them, are fetched quad-cooperatively,
 * ", records in LDS" when the mesh has at most 256 cells and the kernel keeps them in LDS,
 * ", cell codes" when the mesh has at most 256 DISTINCT step records: jb_mesh_ddmc_classes -- and
 * ", cell codes, queues" when, with at most 64 resident blocks, the wave's photons are also staged
 * through queues in LDS: k_ddmc_q, the default on such meshes; ", codes in LDS" behind that when the
 * mesh has at most 1024 cells and 64 distinct records and the codes sit in LDS too);
 * "k_hybrid<2, lean, exact geometry>" on a mesh that mixes IMC and DDMC cells (three launches: IMC
 * phase, DDMC phase, remainder); "" before the first launch.  jb_mesh_exact_geometry: 1 if every
 * resident block has power-of-two cell widths and a lower corner that is a whole number of them
 * (and the per-cell arrays of the resident blocks span less than 4 GiB). */
const char *jb_last_transport_variant(const jb_mesh *mesh);
int jb_mesh_exact_geometry(const jb_mesh *mesh);
/* All-DDMC meshes: the number of DISTINCT step records UpdateDerivedTransportFields found among the
 * resident cells this cycle, as the last jb_transport_photons_ddmc call read it back (0 before the
 * first; gray decks: one per level x face-neighbour pattern).  Up to 256 of them the tracking kernel
 * keeps the records in LDS and gathers a 4-byte cell code per step (variant "..., cell codes");
 * beyond (e.g. a material whose Fleck factor differs from cell to cell) it gathers the 64-byte step
 * record of the cell.  Same results either way. */
int jb_mesh_ddmc_classes(const jb_mesh *mesh);
/* counters accumulated by the transport tasks since the last reset (synchronises) */
jb_status jb_get_transport_stats(jb_context *ctx, jb_transport_stats *stats, int reset);

/* SampleDDMCBlockFace(md) -- jaybenne.hpp:61, sample_ddmc_bface.cpp:81-427; for particles
 * that arrived from another rank */
jb_status jb_sample_ddmc_block_face(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                    int64_t first, int64_t last);

/* CheckCompletion(md, t_end) -- jaybenne.hpp:62, transport.cpp:187-216: JB_ITERATE if any
 * ACTIVE particle has t < t_end, else JB_COMPLETE; the count goes to *unfinished */
jb_status jb_check_completion(jb_context *ctx, const jb_swarm_view *swarm, double t_end,
                              int64_t *unfinished);

/* EvaluateRadiationEnergy<T>(md) -- jaybenne.hpp:67-68, jaybenne.cpp:514-564 */
jb_status jb_zero_energy_tally(jb_context *ctx, jb_mesh *mesh);
jb_status jb_evaluate_radiation_energy(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm);

/* UpdateFluid(md) -- jaybenne.hpp:69, jaybenne.cpp:583-615 */
jb_status jb_update_fluid(jb_context *ctx, jb_mesh *mesh);

/* PhotonReflectBC<BFACE>(swarm) -- boundaries.hpp:24-84; face = 0..5 (ix1, ox1, ix2, ...) */
jb_status jb_photon_reflect_bc(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm, int face);

/* Swarm::RemoveMarkedParticles (transport.cpp:176-178) / DefragParticles (jaybenne.cpp:499-509):
 * keeps the ACTIVE particles and the ones still waiting for their hand-off (OUTGOING, not packed yet --
 * jb_pack_outgoing / jb_exchange turn those slots into holes), closes the holes, updates swarm->n */
jb_status jb_remove_marked_particles(jb_context *ctx, jb_swarm_view *swarm);
/* jaybenne::DefragParticles (reference jaybenne.cpp:499-509: Swarm::Defrag; scheduled by no task
 * list of the reference).  The swarm is compact after jb_remove_marked_particles; this call restores
 * its ORDER: the first swarm->n particles are sorted, in place, by (resident block, cell of their
 * position) -- the order photons are sourced in, which keeps the cell data a wave gathers in the L2
 * of its XCD and which diffusion loosens from cycle to cycle (all-DDMC 3-D, 1e8 photons: 26.5 ms per
 * cycle at cycle 2, 39.9 at cycle 16).  Nothing a particle carries changes; particles of one cell
 * come out in arbitrary order.  Works through 128 bytes of library scratch memory per particle.
 * Asynchronous on the context's stream. */
jb_status jb_defrag_particles(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm);
/* The default schedule of DefragParticles ("defrag_interval = -1", what RadiationStep of the hosts
 * calls at the end of a cycle, after it has read the cycle's event count): the library times the
 * tracking kernels of every jb_transport_photons* call and every sort with HIP events on the
 * context's stream.  It keeps the lowest time per event seen since the last sort and adds up what
 * the cycles since have cost above it (the loss L); the swarm is sorted (*sorted = 1) when one more
 * cycle in the present order would cost more than a cycle has cost on average since the last sort,
 * the sort included -- e(p + 1) >= (S + L) / p with p cycles behind the sort, S the cost of a sort
 * (measured; 15 ms per 1e8 photons until one has been) and e(p + 1) this cycle's loss extrapolated
 * linearly: the period with the lowest cost per cycle for a loss that keeps growing -- and the current
 * cycle is at least 1.5 % slower than the best one, at least 2 cycles after the last sort and with
 * at least 2^20 photons in the swarm.  A sort after which the next cycle is not at least 1 % faster
 * was not what the kernels needed: the minimum distance between sorts doubles (2, 4, .. 256 cycles)
 * until one pays
 * again.  The sort's scratch records (128 bytes per photon) are allocated when the cycles first slow down
 * by 0.5 % (at least a cycle before a sort can be decided, so that the allocation -- ~30 ms per GB -- does
 * not land in the sorting cycle; a run whose swarm keeps its order never allocates them);
 * jb_release_scratch returns them.
 * Slot order only affects speed, never results -- but under this schedule WHEN the swarm is sorted
 * depends on measured times, so the order of the slots (not what any photon carries: compare photons by
 * their creation id) differs from run to run; defrag_interval = 0 in the hosts keeps the order of the
 * reference (never sorted, reproducible slot for slot), k > 0 sorts after every k-th cycle.
 * The caller must have synchronised the stream since the cycle's last transport call.
 * mode: JB_DEFRAG_DECIDE_AND_SORT for a host that holds the whole swarm.  Several ranks sort
 * TOGETHER (a cycle is as long as its slowest rank: a sort on one rank per cycle, in turn, would be
 * paid every cycle): every rank calls with JB_DEFRAG_DECIDE (*sorted = 1: this rank would sort),
 * the host reduces that over the ranks (max) and, if any rank asked, every rank calls again with
 * JB_DEFRAG_SORT_NOW. */
enum { JB_DEFRAG_DECIDE_AND_SORT = 0, JB_DEFRAG_DECIDE = 1, JB_DEFRAG_SORT_NOW = 2 };
jb_status jb_defrag_policy(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                           int64_t events_this_cycle, int32_t mode, int32_t *sorted);
/* Gives the library's scratch memory back (synchronises the stream first).  The scratch buffer
 * grows on demand and is otherwise kept for the life of the context: 128 bytes per photon after a
 * DefragParticles (sized exactly), a few bytes per photon for the hole / hand-off lists.  If it cannot
 * be allocated, jb_defrag_particles returns JB_ERR_HIP and leaves the swarm as it was; jb_defrag_policy
 * then skips the sort with a message on stderr and stops asking. */
jb_status jb_release_scratch(jb_context *ctx);

/* MeshSend / MeshReceive (jaybenne.cpp:36-61) for the inter-rank part: OUTGOING particles are
 * among [first,last) are copied into fixed-size records (JB_RECORD_WORDS x 8 bytes), ordered by
 * destination rank, and their slots are re-marked JB_ST_ABSORBED-like holes (status
 * JB_ST_ESCAPED is kept for escapes; packed particles become JB_ST_ABSORBED so that a later
 * pack does not send them twice).  counts_host[r] = records for rank r.  Unpack appends records
 * to the swarm as ACTIVE particles of this rank's blocks. */
#define JB_RECORD_WORDS 13
jb_status jb_pack_outgoing(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                           int64_t first, int64_t last, int nranks, int64_t *records_dev,
                           int64_t record_capacity, int64_t *counts_host);
jb_status jb_unpack_incoming(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm,
                             const int64_t *records_dev, int64_t nrecords);

/* MeshResetCommunication -> MeshSend -> MeshReceive (jaybenne.cpp:26-61) in ONE call, the records never
 * leaving the device: the OUTGOING particles among [first,last) are counted per destination rank on the
 * device; the transport gathers every rank's counts into the rank x rank matrix straight from that buffer;
 * the matrix is read back once (the only host read-back of the call: nranks (nranks + 3) words -- it carries this
 * rank's receive sizes and *moved_anywhere, the answer to the completion question of jaybenne.cpp:130-131);
 * if anything moved anywhere the records are packed into send_dev, exchanged into recv_dev and appended to
 * the swarm, everything on the context's stream (the call returns when the unpack kernel is launched).
 * *nsent / *nreceived: records this rank handed over / took in.
 * JB_ERR_CAPACITY (nothing has been packed or changed yet; *nsent / *nreceived hold what THIS rank
 * needs; jb_last_error names the rank that lacks room): some rank's buffer or swarm is too small.  Every
 * rank's room travels with its counts in the all-gather, so EVERY rank returns this status in the same
 * call -- none is left waiting in the payload exchange -- and every rank calls again: after growing what
 * was too small, or closing its swarm's holes with jb_remove_marked_particles, with [first,last) =
 * [0, swarm->n) (what still has to go is found by its status).
 *
 * A jb_exchange_transport is the two collectives of the exchange on DEVICE buffers, enqueued on the given
 * HIP stream (0 = success): all_gather_u64 -- count words of every rank, in rank order; all_to_all_v --
 * send_counts[r] records of `words` 8-byte words at send_offsets[r] (in records) to rank r, recv_counts[r]
 * from rank r to recv_offsets[r].  jb_transport_rccl makes the production one from an RCCL communicator
 * (ncclComm_t, one rank per GPU): ncclAllGather and one grouped ncclSend / ncclRecv per peer -- xGMI is a
 * full point-to-point mesh, every rank pair moves its records over its own link.  RCCL is loaded with
 * dlopen when the first transport is made; the library does not link against it.  A host with another
 * fabric fills the struct with its own functions (examples/handoff_mpi.cpp: MPI with host staging). */
typedef struct jb_exchange_transport {
  void *handle;
  int (*all_gather_u64)(void *handle, const uint64_t *in_dev, uint64_t *out_dev, int count, void *hip_stream);
  int (*all_to_all_v)(void *handle, const int64_t *send_dev, const int64_t *send_counts,
                      const int64_t *send_offsets, int64_t *recv_dev, const int64_t *recv_counts,
                      const int64_t *recv_offsets, int words, void *hip_stream);
} jb_exchange_transport;
jb_status jb_transport_rccl(void *nccl_comm, int rank, int nranks, jb_exchange_transport *out);
jb_status jb_transport_release(jb_exchange_transport *transport);
jb_status jb_exchange(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm, int64_t first, int64_t last,
                      int rank, int nranks, const jb_exchange_transport *transport,
                      int64_t *send_dev, int64_t send_capacity, int64_t *recv_dev, int64_t recv_capacity,
                      int64_t *nsent, int64_t *nreceived, int64_t *moved_anywhere);

/* Ghost-zone / halo refresh of one host field -- the role of Parthenon's boundary exchange on
 * the host's FillGhost fields (mcblock.cpp:66-70: density, internal_energy; driven from
 * McblockDriver::HostUpdateTasks, mcblock_driver.cpp:58-74, after UpdateFluid changed
 * internal_energy).  `field` names a member of jb_mesh_view (enum jb_field).
 *   jb_gather_cells: out_dev[i] = field[blk_dev[i]][cell_dev[i]] -- packs the cells another rank
 *     asked for (cell = flat index into the block's [nk][nj][ni] array, ghosts included).
 *   jb_fill_cells: field[dst_blk_dev[i]][dst_cell_dev[i]] = mean of nsamples (a power of two)
 *     source values, summed pairwise (exact when the samples are equal); source s of destination
 *     i is field[src_blk_dev[i*nsamples+s]][src_cell_dev[i*nsamples+s]], or, when that src_blk is
 *     -1, remote_dev[src_cell_dev[i*nsamples+s]] (values received from other ranks).
 * Sources must not be destinations of the same call. */
typedef enum jb_field {
  JB_FIELD_RHO = 0, JB_FIELD_SIE = 1, JB_FIELD_U = 2, JB_FIELD_FLECK = 3, JB_FIELD_TALLY = 4,
  JB_FIELD_EDELTA = 5
} jb_field;
jb_status jb_gather_cells(jb_context *ctx, jb_mesh *mesh, int field, int64_t n,
                          const int32_t *blk_dev, const int32_t *cell_dev, double *out_dev);
jb_status jb_fill_cells(jb_context *ctx, jb_mesh *mesh, int field, int64_t n, int nsamples,
                        const int32_t *dst_blk_dev, const int32_t *dst_cell_dev,
                        const int32_t *src_blk_dev, const int32_t *src_cell_dev,
                        const double *remote_dev);

/* EstimateTimestepMesh(md) -- jaybenne.hpp:75, jaybenne.cpp:271-275 */
double jb_estimate_timestep(const jb_context *ctx);

/* RadiationStep(pmesh, t_start, dt) for a mesh held by ONE rank -- jaybenne.hpp:72,
 * jaybenne.cpp:68-151: derived fields, emission source, transport to completion, census tally,
 * fluid update.  next_id: first unused stream id (updated); cycle: the number of radiation cycles
 * taken so far (0 after initialisation; incremented at ENTRY) -- the emission source of cycle k keys
 * its per-cell rounding streams with epoch k, as every host does (jaybenne_amd.hpp: SourceEpoch).
 * (Changed in round 4: the counter used to be a source-call counter incremented AFTER the source; a
 * caller that still passes 1 for the first cycle gets epoch 2 and different -- equally valid -- rounding
 * streams than the other hosts.  JB_ERR_INVALID for *cycle >= 2^19 - 1: beyond that the keys of the
 * emission and the in-cycle thermal source, (1 << 19) | cycle, would meet.) */
jb_status jb_radiation_step(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm, double t_start,
                            double dt, uint64_t *next_id, uint32_t *cycle, int32_t *prefix_dev);

/* ---- trace ranges -- the reference's Kokkos::Profiling::pushRegion("Jaybenne::Timestep") ...
 * popRegion() and "Jaybenne::TransportLoop" (jaybenne.cpp:87,115,127,145).  Every task entry point
 * above opens a ROCTx range named after its reference task ("Jaybenne::TransportPhotons_DDMC", ...)
 * for its own duration and jb_radiation_step opens the reference's two; a host that drives the
 * tasks itself brackets its cycle and its iterate-sublist with these (jaybenne_amd/jaybenne.py,
 * include/jaybenne_amd.hpp do).  `rocprofv3 --marker-trace --kernel-trace` shows them beside the
 * kernels.  The marker library (librocprofiler-sdk-roctx / libroctx64) is dlopen()ed on first use:
 * no link dependency; absent, or with JB_NO_ROCTX=1, the calls do nothing and return -1
 * (jb_ranges_enabled() = 0).  push returns the nesting level as roctxRangePushA does. */
int jb_range_push(const char *name);
int jb_range_pop(void);
int jb_ranges_enabled(void);

/* ---- debug entry points (parity tests drive the device functions directly) ---------------- */
jb_status jb_debug_philox(jb_context *ctx, const uint32_t ctr[4], const uint32_t key[2],
                          uint32_t out[4]);
jb_status jb_debug_rocrand_philox(jb_context *ctx, uint64_t seed, uint64_t subsequence,
                                  uint32_t out[8]);
jb_status jb_debug_seed_state(jb_context *ctx, uint32_t seed, uint32_t domain, uint64_t id,
                              uint64_t *state);
/* first state of the random stream of the particle with creation index id (csrc/jb_rng.hpp) */
jb_status jb_debug_stream_start(jb_context *ctx, uint32_t seed, uint64_t id, uint64_t *state);
jb_status jb_debug_draw_stream(jb_context *ctx, uint64_t state, int n, double *out_host,
                               uint64_t *final_state);
/* which: 0 log, 1 sin, 2 cos, 3 acos, 4 sqrt, 5 reciprocal, 6 lean sqrt, 7 lean x[i] / x[i+1],
 * 8 lean x[i] / c, 9 sin(2 pi x), 10 cos(2 pi x), 11 1 - exp(-x), 12 x[i] / x[i+1] as the lean
 * arithmetic forms it (numerator times once-refined reciprocal), 13 lean log, 14 lean sqrt, 15 lean log
 * on its 1024-row table (the cell-local IMC kernel's) */
jb_status jb_debug_math(jb_context *ctx, int which, const double *x_host, int n, double *out_host);
/* the opacity / scattering models as the kernels evaluate them: out[0..3] = EPBremss A, B, E
 * (sigma_a = A rho^2 T^-1/2 (1 - e^(-B nu / T)) nu^-3, j = E rho^2 T^1/2, code units) and the
 * effective GrayS kappa_s; and one evaluation: which 0 absorption(rho, T, nu), 1 emissivity(rho,
 * T), 2 scattering(rho, T, nu) for n triples x_host[3 i .. 3 i + 2] */
jb_status jb_debug_model_coefficients(jb_context *ctx, double out[4]);
jb_status jb_debug_model_eval(jb_context *ctx, int which, const double *x_host, int n,
                              double *out_host);
/* step functions on a tape of uniforms.  st: jb_debug_step record (see below); which:
 * 0 ptcl_transport_step, 1 ptcl_ddmc_step, 2 ptcl_ddmc_albedo */
typedef struct jb_debug_step {
  double t_start, dt, ff, aa, ss, vv, dx_push;
  int32_t multi_d, three_d;
  double xl, yl, zl, xu, yu, zu, Px_l, Py_l, Pz_l, Px_u, Py_u, Pz_u;
  double t, x, y, z, vx, vy, vz;
  int32_t ip, jp, kp, is_absorbed, is_scattered, is_rejected;
} jb_debug_step;
jb_status jb_debug_step_call(jb_context *ctx, int which, jb_debug_step *st, const double *tape,
                             int ntape, int *ndraws);
/* which: 0 scatter(vv), 1 sample_face_iso_dir(vv), 2 sample_Planck_energy(sb=a0, temp=a1),
 * 3 SampleFace2D(i_l=i0, dx=a0, P_l=a1, P_u=a2; i=i1, x=a3),
 * 4 SampleFace3D(i1_l=i0, i2_l=i1, dx1=a0, dx2=a1, P=a2..a5; i=(i2,i3), x=(a6,a7)) */
jb_status jb_debug_sample_call(jb_context *ctx, int which, const double *a, const int32_t *i,
                               const double *tape, int ntape, double out[4], int32_t iout[2],
                               int *ndraws);

#ifdef __cplusplus
}
#endif
#endif /* JAYBENNE_AMD_H_ */
