// jb_api.hip -- host side of the C ABI declared in include/jaybenne_amd.h.
//
// Mirrors the task functions of the reference package (src/jaybenne/jaybenne.hpp:48-78) over
// raw device pointers.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include <dlfcn.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/jaybenne_amd.h"
#include "jb_kernels.hpp"
#include "jb_kernel_hybrid.hpp"
#include "jb_kernel_ddmc_q.hpp"
#include "jb_kernel_imc.hpp"

using namespace jb;

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static jb_status fail(jb_status st, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return st;
}

#define JB_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(JB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                 \
  } while (0)

struct jb_context {
  int device = 0;
  hipStream_t stream = nullptr;
  jb_params params{};
  jb_eos eos{};
  jb_opacity opac{};
  jb_scattering scat{};
  DevParams dp{};
  int num_cu = 256;
  unsigned long long *counters_d = nullptr;   // CNT_N + 2 cursors + per-rank counters
  unsigned long long *counters_h = nullptr;   // pinned
  unsigned long long *xch_d = nullptr;        // jb_exchange: the gathered count matrix + pack offsets
  int xch_ranks = 0;                          // ... sized for this many ranks
  long long *scratch_d = nullptr;             // holes / movers / small tables
  bool scratch_alloc_failed = false;          // set by ensure_scratch when hipMalloc itself said no
  std::vector<unsigned long long> xch_matrix;  // jb_exchange: the rank x rank record counts (host copy)
  std::vector<long long> xch_tab;             // ... this rank's send / receive counts and offsets
  unsigned long long xch_room[3] = {0, 0, 0}; // ... and its room: send buffer, receive buffer, free swarm slots (records)
  size_t scratch_words = 0;
  // arithmetic of the gray IMC tracking step: lean (default) or exact (JB_EXACT_ARITH=1 in the
  // environment at jb_initialize, or jb_set_arithmetic)
  bool lean_arith = true;
  int blocks_per_cu_env = 0;  // JB_TRANSPORT_BLOCKS_PER_CU at jb_initialize (tuning aid), 0 = occupancy query
  bool no_ddmc_all = false;   // JB_NO_DDMC_ALL=1 at jb_initialize (tests: k_hybrid on all-DDMC meshes)
  int ddmc_lds_codes = 1;          // JB_DDMC_LDS_CODES=0: k_ddmc_q gathers the cell codes from device memory on any mesh (tests, A/B)
  int ddmc_queues = 1;             // JB_DDMC_QUEUES=0: k_ddmc_all instead of k_ddmc_q where both apply (tests, A/B)
  int max_classes = kMaxClasses;   // JB_DDMC_MAX_CLASSES: fewer (tests of the fall-back to the 64-byte gather)
  int coop_gather = -1;       // JB_COOP_GATHER=0 / 1 / 2: k_ddmc_all's quad-cooperative gather off / on / on with 64-bit addresses, whatever the table size
  bool no_imc_cell = false;   // JB_NO_IMC_CELL=1 at jb_initialize (tests, A/B): the lean step in x-space (k_transport<.., LEAN>) instead of k_imc_cell
  // jb_defrag_policy: HIP events around every transport call of the running cycle, and what the
  // policy remembers from cycle to cycle
  std::vector<hipEvent_t> tev;       // pairs (start, stop)
  int tev_used = 0;                  // events of tev recorded since the last policy call
  bool tev_overflow = false;
  double rate_ref = 0.0;             // lowest ms per event seen since the last sort (0: none yet)
  double rate_before_sort = 0.0;     // ... the rate of the cycle that triggered the last sort
  double excess_ms = 0.0;            // time spent above rate_ref since the last sort
  double last_rate = 0.0;            // ms per event of the cycle the policy looked at last
  double sort_ms_per_photon = 1.5e-7;  // cost of a sort: 15 ms per 1e8 photons until one has been timed
  hipEvent_t sort_ev[2] = {nullptr, nullptr};
  bool sort_timed = false;           // sort_ev brackets a sort whose time has not been read yet
  long long sort_n = 0;
  int cycles_since_sort = 0;
  int min_interval = 2;              // cycles between two sorts (doubles when a sort bought nothing)
  int policy_sorts = 0;
  bool sort_scratch_tried = false;   // the sort's scratch records have been asked for (first policy call)
};
constexpr int kTransportEventPairs = 64;
constexpr int kRankEnd = 1024;                               // CNT_N.. | 16..17 cursors | 32..1023 per-rank counts
constexpr int kCounterWords = kRankEnd + kQueues * kQueueStride;  // | 1024.. the queue heads, one line each (CNT_QUEUE)
static_assert(CNT_QUEUE == kRankEnd, "the queue heads follow the per-rank counts");
constexpr int kCursorBase = 16;
constexpr int kRankBase = 32;

struct jb_mesh {
  jb_context *ctx = nullptr;
  DevMesh dm{};
  std::vector<void *> owned;  // device allocations
  int nranks_seen = 1;
  // every resident block: power-of-two cell widths, lower corner a whole number of them (the
  // cell-face arithmetic is then exact; k_transport<..., EXACT>)
  bool exact_geom = false;
  // the mean-free-path arrays of all resident blocks lie within 4 GiB (32-bit byte offsets): what
  // the cell-local IMC kernel needs of a mesh, whatever its cell widths
  bool offsets32 = false;
  // the cell-local kernels (k_imc_cell, k_hybrid MODE 3) step the photon's byte offset with 24-bit
  // multiply-adds of the BYTE strides 8 ni and 8 ni nj: both have to stay below 2^23
  bool cell_ok = false;
  const char *last_variant = "";  // the k_transport instantiation launched last
  const char *last_pair = "";     // ... and the k_ddmc_all launched beside it (gray DDMC), or ""
  const DevMesh *dm_dev = nullptr;  // copy of dm in device memory (k_hybrid reads the view through it)
  const int *nbr_dq = nullptr;      // k_imc_cell: change of the cell's byte offset per (block, face) crossing
  bool uniform_geom = false;        // every resident block has the cell widths of block 0 (k_imc_cell<.., UNIFORM>)
  // "some cell takes IMC steps" (DevMesh::not_all_ddmc) as the host last read it: -1 = not since
  // UpdateDerivedTransportFields rewrote it (the first DDMC transport call of a cycle reads it back,
  // one synchronisation; the further transport iterations of a multi-rank cycle reuse the answer)
  int not_all_ddmc_host = -1;
  int nclass_host = 0;   // ... and the number of distinct step records (DevMesh::ddmc_code), read with it
};

__global__ void k_rcp_refined(double b, double *out) { *out = m_rcp_refined(b); }

// ------------------------------------------------------------------------------------------------
// Trace ranges.  The reference brackets the cycle and its transport loop with
// Kokkos::Profiling::pushRegion("Jaybenne::Timestep" / "Jaybenne::TransportLoop") (jaybenne.cpp:87,115,
// 127,145), which Kokkos Tools hand to whatever profiler is attached.  Here every task entry point of
// the C ABI opens a ROCTx range of its reference name ("Jaybenne::<task>"), jb_radiation_step opens the
// reference's two, and a host that drives the tasks itself opens them through jb_range_push / _pop;
// `rocprofv3 --marker-trace` shows them beside the kernels.  The marker library is dlopen()ed on first
// use -- no link dependency -- and everything is a no-op when it is absent or JB_NO_ROCTX=1.
struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
static const RoctxApi &roctx_api() {
  static const RoctxApi api = [] {
    RoctxApi a;
    const char *off = getenv("JB_NO_ROCTX");
    if (off && off[0] == '1') return a;
    // (rocprofv3 listens to the SDK's marker library; the roctracer one is what older tools attach to)
    for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) continue;
      a.push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
      a.pop = (int (*)())dlsym(lib, "roctxRangePop");
      if (a.push && a.pop) break;
      a = RoctxApi();
    }
    return a;
  }();
  return api;
}
struct TraceRange {
  bool open;
  explicit TraceRange(const char *name) : open(roctx_api().push != nullptr) { if (open) (void)roctx_api().push(name); }
  ~TraceRange() { if (open) (void)roctx_api().pop(); }
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
};
#define JB_RANGE(name) TraceRange jb_trace_range_(name)

extern "C" int jb_range_push(const char *name) {
  if (!name || !roctx_api().push) return -1;
  return roctx_api().push(name);
}
extern "C" int jb_range_pop(void) { return roctx_api().pop ? roctx_api().pop() : -1; }
extern "C" int jb_ranges_enabled(void) { return roctx_api().push != nullptr ? 1 : 0; }

extern "C" const char *jb_last_error(void) { return g_err; }
extern "C" const char *jb_version(void) { return "jaybenne_amd 0.1 (gfx950)"; }

static int grid_for(const jb_context *ctx, long long n, int per_cu = 8) {
  long long blocks = (n + kBlock - 1) / kBlock;
  const long long cap = (long long)ctx->num_cu * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// (slack: small tables grow with the run and are re-requested often; the sort's particle records --
// 128 bytes per photon -- are sized exactly: 25 % on top of 12.8 GB is memory a run near the card's
// capacity may not have)
static jb_status ensure_scratch(jb_context *ctx, size_t words, bool slack = true) {
  if (words <= ctx->scratch_words) return JB_COMPLETE;
  if (ctx->scratch_d) JB_HIP(hipFree(ctx->scratch_d));
  ctx->scratch_d = nullptr;
  ctx->scratch_words = 0;
  const size_t want = slack ? words + words / 4 + 1024 : words + 16;
  const hipError_t e = hipMalloc(&ctx->scratch_d, want * sizeof(long long));
  if (e != hipSuccess) {   // (told apart from every other failure: DefragParticles may go without its scratch)
    ctx->scratch_d = nullptr;
    ctx->scratch_alloc_failed = true;
    (void)hipGetLastError();
    return fail(JB_ERR_HIP, "hipMalloc of %zu bytes of scratch memory failed: %s", want * sizeof(long long),
                hipGetErrorString(e));
  }
  ctx->scratch_words = want;
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
extern "C" jb_status jb_initialize(const jb_params *params, const jb_eos *eos,
                                   const jb_opacity *opacity, const jb_scattering *scattering,
                                   int device, jb_context **out) {
  if (!params || !eos || !opacity || !scattering || !out)
    return fail(JB_ERR_INVALID, "jb_initialize: null argument");
  // PARTHENON_REQUIRE / PARTHENON_FAIL conditions of jaybenne.cpp:167-171,205-216
  if (!(params->min_swarm_occupancy >= 0.0 && params->min_swarm_occupancy < 1.0))
    return fail(JB_ERR_INVALID,
                "Minimum allowable swarm occupancy must be >= 0 and less than 1");
  if (params->source_strategy != JB_STRATEGY_UNIFORM && params->source_strategy != JB_STRATEGY_ENERGY)
    return fail(JB_ERR_INVALID, "Only uniform or energy source strategies supported!");
  if (params->num_particles < 0) return fail(JB_ERR_INVALID, "num_particles must be >= 0");
  if (eos->model != JB_EOS_IDEAL_GAS)
    return fail(JB_ERR_UNSUPPORTED, "only the IdealGas EOS is built");
  if (opacity->model != JB_OPAC_GRAY && opacity->model != JB_OPAC_EPBREMSS)
    return fail(JB_ERR_UNSUPPORTED, "unknown absorption opacity model (Gray, EPBremss)");
  if (scattering->model != JB_SCAT_GRAY && scattering->model != JB_SCAT_THOMSON)
    return fail(JB_ERR_UNSUPPORTED, "unknown scattering model (GrayS, ThomsonS)");
  if (opacity->model == JB_OPAC_EPBREMSS &&
      !(opacity->time_scale > 0.0 && opacity->mass_scale > 0.0 && opacity->length_scale > 0.0 &&
        opacity->temperature_scale > 0.0))
    return fail(JB_ERR_INVALID, "EPBremss needs positive time / mass / length / temperature scales");
  if (scattering->model == JB_SCAT_THOMSON && !(scattering->length_scale > 0.0))
    return fail(JB_ERR_INVALID, "ThomsonS needs a positive length scale");
  JB_HIP(hipSetDevice(device));
  jb_context *ctx = new (std::nothrow) jb_context();
  if (!ctx) return fail(JB_ERR_HIP, "out of host memory");
  ctx->device = device;
  ctx->params = *params;
  ctx->eos = *eos;
  ctx->opac = *opacity;
  ctx->scat = *scattering;
  {
    const char *ex = getenv("JB_EXACT_ARITH");
    ctx->lean_arith = !(ex && ex[0] == '1');
  }
  if (const char *e = getenv("JB_TRANSPORT_BLOCKS_PER_CU")) ctx->blocks_per_cu_env = atoi(e);
  if (const char *e = getenv("JB_NO_DDMC_ALL")) ctx->no_ddmc_all = e[0] == '1';
  if (const char *e = getenv("JB_NO_IMC_CELL")) ctx->no_imc_cell = e[0] == '1';
  if (const char *e = getenv("JB_COOP_GATHER")) ctx->coop_gather = e[0] == '1' ? 1 : (e[0] == '2' ? 2 : (e[0] == '4' ? 4 : 0));  // (tests, A/B runs)
  if (const char *e = getenv("JB_DDMC_QUEUES")) ctx->ddmc_queues = e[0] != '0';
  if (const char *e = getenv("JB_DDMC_LDS_CODES")) ctx->ddmc_lds_codes = e[0] != '0';
  if (const char *e = getenv("JB_DDMC_MAX_CLASSES")) {   // (tests: the fall-back when a mesh has more distinct step records)
    const int v = atoi(e);
    ctx->max_classes = v < 0 ? 0 : (v > kMaxClasses ? kMaxClasses : v);
  }
  ctx->dp.key0 = (uint32_t)params->seed;  // RngPool rng_pool(seed): unadjusted (quirk 1)
  ctx->dp.use_ddmc = params->use_ddmc;
  ctx->dp.do_feedback = params->do_feedback;
  ctx->dp.tau_ddmc = params->tau_ddmc;
  ctx->dp.c = opacity->c;
  ctx->dp.sb = opacity->sb;
  ctx->dp.cv = eos->cv;
  ctx->dp.kappa_a = opacity->kappa;
  ctx->dp.kappa_s = scattering->kappa_s;
  ctx->dp.apm = scattering->apm;
  ctx->dp.opac_model = opacity->model;
  ctx->dp.lean = ctx->lean_arith ? 1 : 0;
  ctx->dp.hyb_imc_budget = JB_HYBRID_IMC_BUDGET;    // (tuning aids: environment overrides)
  ctx->dp.hyb_park_budget = JB_HYBRID_PARK_BUDGET;
  if (const char *e = getenv("JB_HYBRID_IMC_BUDGET")) ctx->dp.hyb_imc_budget = atoi(e) > 0 ? atoi(e) : JB_HYBRID_IMC_BUDGET;
  if (const char *e = getenv("JB_HYBRID_PARK_BUDGET")) ctx->dp.hyb_park_budget = atoi(e) > 0 ? atoi(e) : JB_HYBRID_PARK_BUDGET;
  ctx->dp.ep_A = ctx->dp.ep_B = ctx->dp.ep_E = 0.0;
  {
    // CGS constants (CODATA 2018): electron charge (esu), electron / proton mass, Planck,
    // Boltzmann, speed of light, Thomson cross-section
    const double qe = 4.803204712570263e-10, me = 9.1093837015e-28, mp = 1.67262192369e-24,
                 hp = 6.62607015e-27, kb = 1.380649e-16, cl = 2.99792458e10,
                 sigma_t = 6.6524587321e-25, pi = 3.14159265358979323846;
    if (opacity->model == JB_OPAC_EPBREMSS) {
      const double tau = opacity->time_scale, mu = opacity->mass_scale, lam = opacity->length_scale,
                   th = opacity->temperature_scale;
      const double e6 = (qe * qe) * (qe * qe) * (qe * qe);
      const double k_abs = (4.0 * e6) / (3.0 * me * hp * cl) * std::sqrt((2.0 * pi) / (3.0 * kb * me));
      const double k_em = std::sqrt((2.0 * pi * kb) / (3.0 * me)) *
                          ((32.0 * pi * e6) / (3.0 * hp * me * (cl * cl * cl)));
      const double n_per_rho = (mu / (lam * lam * lam)) / mp;  // n_e = n_i per unit code density
      const double n2 = n_per_rho * n_per_rho, tau3 = tau * tau * tau;
      ctx->dp.ep_A = lam * k_abs * n2 / std::sqrt(th) * tau3;
      ctx->dp.ep_B = hp / (kb * th * tau);
      ctx->dp.ep_E = k_em * std::sqrt(th) * n2 * (tau3 * lam / mu);
      ctx->dp.kappa_a = 0.0;
    }
    if (scattering->model == JB_SCAT_THOMSON)
      ctx->dp.kappa_s = sigma_t / (scattering->length_scale * scattering->length_scale);
  }
  // (a failure below must not leak the context: jb_finalize releases whatever exists)
  auto init_device_state = [&]() -> jb_status {
    hipDeviceProp_t prop;
    JB_HIP(hipGetDeviceProperties(&prop, device));
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    JB_HIP(hipMalloc(&ctx->counters_d, kCounterWords * sizeof(unsigned long long)));
    JB_HIP(hipMemset(ctx->counters_d, 0, kCounterWords * sizeof(unsigned long long)));
    JB_HIP(hipHostMalloc(&ctx->counters_h, kCounterWords * sizeof(unsigned long long)));
    // the refined reciprocal of c that the step functions divide with (jb_math.hpp, m_div_r): one
    // device evaluation, so that it is the v_rcp_f64-seeded value the kernels would compute
    hipLaunchKernelGGL(k_rcp_refined, dim3(1), dim3(1), 0, 0, ctx->dp.c, (double *)ctx->counters_d);
    JB_HIP(hipMemcpy(&ctx->dp.rc, ctx->counters_d, sizeof(double), hipMemcpyDeviceToHost));
    JB_HIP(hipMemset(ctx->counters_d, 0, sizeof(double)));
    return JB_COMPLETE;
  };
  const jb_status st = init_device_state();
  if (st != JB_COMPLETE) {
    jb_finalize(ctx);
    return st;
  }
  *out = ctx;
  return JB_COMPLETE;
}

extern "C" jb_status jb_finalize(jb_context *ctx) {
  if (!ctx) return JB_COMPLETE;
  (void)hipSetDevice(ctx->device);
  if (ctx->counters_d) (void)hipFree(ctx->counters_d);
  if (ctx->counters_h) (void)hipHostFree(ctx->counters_h);
  if (ctx->scratch_d) (void)hipFree(ctx->scratch_d);
  if (ctx->xch_d) (void)hipFree(ctx->xch_d);
  for (hipEvent_t e : ctx->tev) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->sort_ev) if (e) (void)hipEventDestroy(e);
  delete ctx;
  return JB_COMPLETE;
}

extern "C" jb_status jb_set_stream(jb_context *ctx, void *hip_stream) {
  if (!ctx) return fail(JB_ERR_INVALID, "null context");
  ctx->stream = (hipStream_t)hip_stream;
  return JB_COMPLETE;
}

extern "C" jb_status jb_synchronize(jb_context *ctx) {
  if (!ctx) return fail(JB_ERR_INVALID, "null context");
  JB_HIP(hipSetDevice(ctx->device));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  return JB_COMPLETE;
}

extern "C" int32_t jb_param_seed(const jb_context *ctx) {
  return ctx->params.unique_rank_seeds ? ctx->params.seed + ctx->params.rank : ctx->params.seed;
}

extern "C" double jb_estimate_timestep(const jb_context *ctx) { return ctx->params.dt; }

// ------------------------------------------------------------------------------------------------
template <class T>
static jb_status upload(jb_mesh *m, const T *host, size_t count, const T **dev) {
  void *p = nullptr;
  if (count == 0) count = 1;
  JB_HIP(hipMalloc(&p, count * sizeof(T)));
  m->owned.push_back(p);
  if (host) JB_HIP(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev = (const T *)p;
  return JB_COMPLETE;
}

extern "C" jb_status jb_mesh_create(jb_context *ctx, const jb_mesh_view *v, jb_mesh **out) {
  if (!ctx || !v || !out) return fail(JB_ERR_INVALID, "jb_mesh_create: null argument");
  if (v->ndim < 1 || v->ndim > 3) return fail(JB_ERR_INVALID, "ndim must be 1, 2 or 3");
  if (v->ng < 1) return fail(JB_ERR_INVALID, "ng >= 1 required (face fields share the cell layout)");
  if (v->nblocks < 1 || v->nblocks_total < v->nblocks)
    return fail(JB_ERR_INVALID, "bad block counts");
  for (int d = 0; d < 3; ++d) {
    if (v->nx[d] < 1 || v->nleaf[d] < 1) return fail(JB_ERR_INVALID, "bad nx / nleaf");
    if (d >= v->ndim && v->nx[d] != 1) return fail(JB_ERR_INVALID, "inactive dimension with nx != 1");
    if (!(v->gmax[d] > v->gmin[d])) return fail(JB_ERR_INVALID, "empty domain");
  }
  if (!v->leaf_map || !v->owner || !v->local_index || !v->gid || !v->blk_xmin || !v->blk_xmax ||
      !v->blk_dx || !v->blk_level || !v->blk_nbr_lev)
    return fail(JB_ERR_INVALID, "jb_mesh_create: missing table");
  if (!v->rho || !v->sie || !v->u || !v->fleck || !v->tally || !v->edelta || !v->src_ew ||
      !v->src_num)
    return fail(JB_ERR_INVALID, "jb_mesh_create: missing field pointer table");
  if (ctx->params.use_ddmc && (!v->P1 || (v->ndim > 1 && !v->P2) || (v->ndim > 2 && !v->P3)))
    return fail(JB_ERR_INVALID, "use_ddmc requires the ddmc_face_prob arrays");
  // every table index the kernels will form is checked here, once
  const long long nleaf = (long long)v->nleaf[0] * v->nleaf[1] * v->nleaf[2];
  for (long long q = 0; q < nleaf; ++q)
    if (v->leaf_map[q] < 0 || v->leaf_map[q] >= v->nblocks_total)
      return fail(JB_ERR_INVALID, "leaf_map entry out of range");
  for (int g = 0; g < v->nblocks_total; ++g) {
    const int li = v->local_index[g];
    if (li >= v->nblocks || li < -1 || (li >= 0 && v->gid[li] != g))
      return fail(JB_ERR_INVALID, "local_index / gid inconsistent for block %d", g);
    if (v->owner[g] == v->rank) {
      if (li < 0) return fail(JB_ERR_INVALID, "owned block %d is not resident", g);
      if (v->owned && !v->owned[li]) return fail(JB_ERR_INVALID, "block %d owned but flagged as halo", g);
    } else if (li >= 0 && (!v->owned || v->owned[li])) {
      return fail(JB_ERR_INVALID, "resident block %d of another rank must be flagged as a halo copy", g);
    }
    if (v->owner[g] < 0) return fail(JB_ERR_INVALID, "negative owner rank");
  }
  JB_HIP(hipSetDevice(ctx->device));
  jb_mesh *m = new (std::nothrow) jb_mesh();
  if (!m) return fail(JB_ERR_HIP, "out of host memory");
  m->ctx = ctx;
  DevMesh &D = m->dm;
  D.ndim = v->ndim; D.ng = v->ng; D.nblocks = v->nblocks; D.nblocks_total = v->nblocks_total;
  D.rank = v->rank;
  for (int d = 0; d < 3; ++d) { D.nx[d] = v->nx[d]; D.nleaf[d] = v->nleaf[d]; D.gmin[d] = v->gmin[d]; D.gmax[d] = v->gmax[d]; }
  for (int f = 0; f < 6; ++f) {
    if (v->bc[f] < 0 || v->bc[f] > 2) { delete m; return fail(JB_ERR_INVALID, "unknown swarm boundary"); }
    D.bc[f] = v->bc[f];
  }
  D.is = v->ng; D.js = v->ndim >= 2 ? v->ng : 0; D.ks = v->ndim >= 3 ? v->ng : 0;
  D.ni = v->nx[0] + 2 * D.is; D.nj = v->nx[1] + 2 * D.js; D.nk = v->nx[2] + 2 * D.ks;
  D.ie = D.is + v->nx[0] - 1; D.je = D.js + v->nx[1] - 1; D.ke = D.ks + v->nx[2] - 1;
  D.ncell = v->nx[0] * v->nx[1] * v->nx[2];
  D.ntot = (long long)D.ni * D.nj * D.nk;
  D.inv_ntot = 1.0 / (double)D.ntot;
  D.inv_nij = 1.0 / ((double)D.ni * (double)D.nj);
  D.inv_ni = 1.0 / (double)D.ni;
  if (D.ntot * 8 >= (1ll << 31)) { delete m; return fail(JB_ERR_INVALID, "block too large: cell indices are 32-bit"); }
  if (D.ni >= (1 << 23) || (long long)D.nj * D.nk >= (1ll << 23)) {
    delete m;
    return fail(JB_ERR_INVALID, "block too large: ni and nj * nk must stay below 2^23 (24-bit cell index arithmetic)");
  }
  if (v->nblocks_total >= (1 << 20)) {  // cell_stream_id packs the global block id in 20 bits
    delete m;
    return fail(JB_ERR_INVALID, "more than 2^20 blocks: the per-cell source streams would alias");
  }
  if (D.ncell >= (1 << 24)) {           // ... and the cell in 24
    delete m;
    return fail(JB_ERR_INVALID, "more than 2^24 cells per block: the per-cell source streams would alias");
  }
  {
    const char *off = getenv("JB_NO_EXACT_GEOM");  // tests: run the general kernels on an exact mesh
    bool exact = !(off && off[0] == '1');
    for (int b = 0; exact && b < v->nblocks; ++b)
      for (int d = 0; exact && d < v->ndim; ++d) {
        const double dx = v->blk_dx[3 * b + d], x0 = v->blk_xmin[3 * b + d];
        int e;
        const double q = x0 / dx;  // exact when dx is a power of two
        exact = dx > 0.0 && std::frexp(dx, &e) == 0.5 && q == std::nearbyint(q) &&
                std::fabs(q) < 1099511627776.0;  // 2^40: (q + i + 0.5) dx is exact
      }
    D.exact = exact ? 1 : 0;
    // (the EXACT tracking kernels also address the mean-free-path arrays of all resident blocks
    // with 32-bit byte offsets: 16 bytes per cell)
    m->offsets32 = 16ull * (unsigned long long)D.ntot * (unsigned long long)v->nblocks < (1ull << 32);
    m->exact_geom = exact && m->offsets32;
    m->cell_ok = m->offsets32 && 8ll * D.ni < (1ll << 23) && (v->ndim < 3 || 8ll * D.ni * D.nj < (1ll << 23));
    m->uniform_geom = true;
    for (int b = 1; b < v->nblocks; ++b)
      for (int d = 0; d < 3; ++d) m->uniform_geom = m->uniform_geom && v->blk_dx[3 * b + d] == v->blk_dx[d];
  }
  int maxrank = 0;
  for (int g = 0; g < v->nblocks_total; ++g) maxrank = v->owner[g] > maxrank ? v->owner[g] : maxrank;
  m->nranks_seen = maxrank + 1;
  jb_status st;
#define UP(field, count)                                                         \
  if ((st = upload(m, v->field, (size_t)(count), &D.field)) != JB_COMPLETE) {     \
    jb_mesh_destroy(m);                                                          \
    return st;                                                                   \
  }
  UP(leaf_map, nleaf);
  UP(owner, v->nblocks_total);
  UP(local_index, v->nblocks_total);
  UP(gid, v->nblocks);
  {
    std::vector<int32_t> ow(v->nblocks, 1);
    if (v->owned) ow.assign(v->owned, v->owned + v->nblocks);
    if ((st = upload(m, ow.data(), ow.size(), &D.owned)) != JB_COMPLETE) {
      jb_mesh_destroy(m);
      return st;
    }
  }
  UP(blk_xmin, 3 * v->nblocks);
  UP(blk_xmax, 3 * v->nblocks);
  UP(blk_dx, 3 * v->nblocks);
  {
    std::vector<double> inv(3 * (size_t)v->nblocks);
    for (size_t q = 0; q < inv.size(); ++q) inv[q] = 1.0 / v->blk_dx[q];
    if ((st = upload(m, inv.data(), inv.size(), &D.blk_inv_dx)) != JB_COMPLETE) {
      jb_mesh_destroy(m);
      return st;
    }
    for (int d = 0; d < 3; ++d)
      D.inv_leaf_len[d] = 1.0 / ((v->gmax[d] - v->gmin[d]) / (double)v->nleaf[d]);
  }
  UP(blk_level, v->nblocks);
  UP(blk_nbr_lev, 6 * v->nblocks);
#undef UP
#define UPF(field)                                                                           \
  if (v->field) {                                                                            \
    for (int b = 0; b < v->nblocks; ++b)                                                     \
      if (!v->field[b]) { jb_mesh_destroy(m); return fail(JB_ERR_INVALID, "null block pointer in " #field); } \
    const double *const *tmp = nullptr;                                                      \
    if ((st = upload(m, (const double *const *)v->field, (size_t)v->nblocks, &tmp)) != JB_COMPLETE) { \
      jb_mesh_destroy(m);                                                                    \
      return st;                                                                             \
    }                                                                                        \
    D.field = (double *const *)tmp;                                                          \
  } else {                                                                                   \
    D.field = nullptr;                                                                       \
  }
  UPF(rho) UPF(sie) UPF(u) UPF(fleck) UPF(tally) UPF(edelta) UPF(src_ew) UPF(src_num)
  UPF(P1) UPF(P2) UPF(P3)
#undef UPF
  // face-crossing table of the tracking kernels (DevMesh::nbr_ent / nbr_x0)
  {
    std::vector<int32_t> ent(6 * (size_t)v->nblocks, -1);
    std::vector<double> x0(6 * (size_t)v->nblocks, 0.0);
    // k_imc_cell: a photon that has stepped through face f of block b sits in b's ghost cell behind
    // that face; dq moves the byte offset of that cell (16 ntot bytes per block, 8 per cell) to the
    // cell it is in after the crossing -- the first / last interior cell along the axis of the
    // destination block, or, at a reflecting wall, the interior cell it came from
    std::vector<int32_t> dq(6 * (size_t)v->nblocks, 0);
    const long long stride8[3] = {8, 8ll * D.ni, 8ll * D.ni * D.nj};
    const int first[3] = {D.is, D.js, D.ks};
    for (int b = 0; b < v->nblocks; ++b) {
      long long l0[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
      bool ok = true;
      for (int d = 0; d < v->ndim; ++d) {
        const double leaf = (v->gmax[d] - v->gmin[d]) / (double)v->nleaf[d];
        const double q0 = (v->blk_xmin[3 * b + d] - v->gmin[d]) / leaf;
        const double qc = (v->blk_xmax[3 * b + d] - v->blk_xmin[3 * b + d]) / leaf;
        l0[d] = (long long)std::llround(q0);
        cnt[d] = (long long)std::llround(qc);
        ok = ok && std::fabs(q0 - (double)l0[d]) < 1e-6 && std::fabs(qc - (double)cnt[d]) < 1e-6 &&
             cnt[d] >= 1 && l0[d] >= 0 && l0[d] + cnt[d] <= v->nleaf[d];
      }
      if (!ok) continue;  // (a block that is not a union of leaves: general relocation only)
      for (int d = 0; d < v->ndim; ++d)
        for (int up = 0; up < 2; ++up) {
          const int f = 2 * d + up;
          long long l[3] = {l0[0], l0[1], l0[2]};
          l[d] = up ? l0[d] + cnt[d] : l0[d] - 1;
          int kind = 0;
          if (l[d] < 0 || l[d] >= v->nleaf[d]) {  // the face lies on the domain boundary
            const int bc = v->bc[f];
            if (bc == JB_BC_REFLECT) {
              ent[6 * (size_t)b + f] = (2 << 28) | b;
              x0[6 * (size_t)b + f] = v->blk_xmin[3 * b + d] - (double)first[d] * v->blk_dx[3 * b + d];
              dq[6 * (size_t)b + f] = (int32_t)(up ? -stride8[d] : stride8[d]);
              continue;
            }
            if (bc != JB_BC_PERIODIC) continue;  // outflow: the particle escapes
            kind = 1;
            l[d] = up ? 0 : v->nleaf[d] - 1;
          }
          const int g = v->leaf_map[(l[2] * v->nleaf[1] + l[1]) * v->nleaf[0] + l[0]];
          const int li = v->local_index[g];
          if (li < 0) continue;  // destination not resident: hand-off
          bool same = true;  // same size, aligned across the face
          for (int e = 0; e < 3; ++e) {
            same = same && v->blk_dx[3 * li + e] == v->blk_dx[3 * b + e];
            if (e != d && e < v->ndim) same = same && v->blk_xmin[3 * li + e] == v->blk_xmin[3 * b + e];
          }
          if (!same || li >= (1 << 28)) continue;
          ent[6 * (size_t)b + f] = (kind << 28) | li;
          x0[6 * (size_t)b + f] = v->blk_xmin[3 * li + d] - (double)first[d] * v->blk_dx[3 * li + d];
          if (m->offsets32)  // (offsets below 4 GiB: the difference fits 32 bits, as a wrapping sum)
            dq[6 * (size_t)b + f] = (int32_t)(uint32_t)(16ll * D.ntot * ((long long)li - b) +
                                                        (up ? -stride8[d] : stride8[d]) * v->nx[d]);
        }
    }
    if ((st = upload(m, dq.data(), dq.size(), &m->nbr_dq)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    if ((st = upload(m, ent.data(), ent.size(), &D.nbr_ent)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    if ((st = upload(m, x0.data(), x0.size(), &D.nbr_x0)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
  }
  {
    // [0] = 1 until UpdateDerivedTransportFields has looked at the cells; [1] = distinct step records (cell classes)
    const int init[2] = {1, 0};
    const int *flag = nullptr;
    if ((st = upload(m, init, 2, &flag)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    D.not_all_ddmc = (int *)flag;
  }
  D.ddmc_code = nullptr;
  D.ddmc_class = nullptr;
  D.ddmc_class_slot = nullptr;
  D.ddmc_base = nullptr;
  D.ddmc_step = nullptr;
  D.lam_hyb = nullptr;
  // gray (frequency-independent) opacities: library-owned per-cell mean-free-path arrays
  D.lam_base = nullptr;
  D.lam_abs = nullptr;
  D.lam_sc = nullptr;
  D.ddmc_cell = nullptr;
  // JB_PER_EVENT_OPACITY=1 keeps the general path -- EOS and opacities evaluated per event from
  // rho, sie and the photon energy, as the reference does (transport.cpp:122-127) -- also for gray
  // models; it is what a frequency-dependent opacity would run, and the tests exercise it
  const char *general = getenv("JB_PER_EVENT_OPACITY");
  const bool per_event = general && general[0] == '1';
  if (!per_event && ctx->opac.model == JB_OPAC_GRAY) {  // (ThomsonS has the GrayS form)
    const size_t per = (size_t)D.ntot;
    double *base = nullptr;
    hipError_t e = hipMalloc(&base, sizeof(double) * per * 2 * (size_t)v->nblocks);
    if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the mean-free-path arrays failed: %s", hipGetErrorString(e)); }
    m->owned.push_back(base);
    std::vector<const double *> pa(v->nblocks), ps(v->nblocks);
    for (int b = 0; b < v->nblocks; ++b) {
      pa[b] = base + (size_t)(2 * b) * per;
      ps[b] = base + (size_t)(2 * b + 1) * per;
    }
    D.lam_base = base;
    const double *const *tmp = nullptr;
    if ((st = upload(m, (const double *const *)pa.data(), (size_t)v->nblocks, &tmp)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    D.lam_abs = (double *const *)tmp;
    if ((st = upload(m, (const double *const *)ps.data(), (size_t)v->nblocks, &tmp)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    D.lam_sc = (double *const *)tmp;
    D.ddmc_cell = nullptr;
    if (ctx->params.use_ddmc) {
      double *pack = nullptr;
      e = hipMalloc(&pack, sizeof(double) * per * 8 * (size_t)v->nblocks);
      if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the DDMC cell records failed: %s", hipGetErrorString(e)); }
      m->owned.push_back(pack);
      std::vector<const double *> pp(v->nblocks);
      for (int b = 0; b < v->nblocks; ++b) pp[b] = pack + (size_t)b * per * 8;
      if ((st = upload(m, (const double *const *)pp.data(), (size_t)v->nblocks, &tmp)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
      D.ddmc_cell = (double *const *)tmp;
      D.ddmc_base = pack;
      double *step_rec = nullptr;
      e = hipMalloc(&step_rec, sizeof(double) * per * 8 * (size_t)v->nblocks);
      if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the DDMC step records failed: %s", hipGetErrorString(e)); }
      m->owned.push_back(step_rec);
      // (on the context's stream, like k_lam_ghost_codes below: a null-stream memset is not ordered against a
      // kernel on a non-blocking stream the host handed in with jb_set_stream, and would wipe the ghost codes)
      (void)hipMemsetAsync(step_rec, 0, sizeof(double) * per * 8 * (size_t)v->nblocks, ctx->stream);
      D.ddmc_step = step_rec;
      // cell codes + the table of distinct step records (k_ddmc_all<.., GATHER 4>); a code holds a record
      // number in 29 bits
      if ((unsigned long long)per * (unsigned long long)v->nblocks < (1ull << 29)) {
        unsigned *code = nullptr;
        e = hipMalloc(&code, sizeof(unsigned) * per * (size_t)v->nblocks);
        if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the DDMC cell codes failed: %s", hipGetErrorString(e)); }
        m->owned.push_back(code);
        // (ghost cells that k_lam_ghost_codes leaves alone -- there are none -- would read "class 0")
        (void)hipMemsetAsync(code, 0, sizeof(unsigned) * per * (size_t)v->nblocks, ctx->stream);
        double *cls = nullptr;
        e = hipMalloc(&cls, sizeof(double) * 8 * kMaxClasses + sizeof(int) * 2 * kClassSlots);
        if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the DDMC class table failed: %s", hipGetErrorString(e)); }
        m->owned.push_back(cls);
        D.ddmc_code = code;
        D.ddmc_class = cls;
        D.ddmc_class_slot = (int *)(cls + 8 * kMaxClasses);
      }
      double *hyb = nullptr;
      e = hipMalloc(&hyb, sizeof(double) * per * (size_t)v->nblocks);
      if (e != hipSuccess) { jb_mesh_destroy(m); return fail(JB_ERR_HIP, "hipMalloc of the hybrid mean-free-path array failed: %s", hipGetErrorString(e)); }
      m->owned.push_back(hyb);
      (void)hipMemsetAsync(hyb, 0, sizeof(double) * per * (size_t)v->nblocks, ctx->stream);
      D.lam_hyb = hyb;
    }
    // ghost cells of the scattering mean free path: which face of the block lies between them and
    // the interior (k_imc_cell reads "the photon has left its block" off the value it gathers);
    // UpdateDerivedTransportFields only ever writes interior cells
    hipLaunchKernelGGL(k_lam_ghost_codes, dim3(grid_for(ctx, (long long)v->nblocks * D.ntot)), dim3(kBlock), 0,
                       ctx->stream, D, m->nbr_dq);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      jb_mesh_destroy(m);
      return fail(JB_ERR_HIP, "writing the ghost-cell codes of the mean-free-path arrays failed");
    }
  }
  {
    const DevMesh *copy = nullptr;
    if ((st = upload(m, &D, 1, &copy)) != JB_COMPLETE) { jb_mesh_destroy(m); return st; }
    m->dm_dev = copy;
  }
  *out = m;
  return JB_COMPLETE;
}

extern "C" jb_status jb_mesh_destroy(jb_mesh *m) {
  if (!m) return JB_COMPLETE;
  for (void *p : m->owned) (void)hipFree(p);
  delete m;
  return JB_COMPLETE;
}

static DevSwarm dev_swarm(const jb_swarm_view *s) {
  DevSwarm S;
  S.x = s->x; S.y = s->y; S.z = s->z; S.vx = s->vx; S.vy = s->vy; S.vz = s->vz;
  S.t = s->t; S.w = s->w; S.e = s->e;
  S.ip = s->ip; S.jp = s->jp; S.kp = s->kp; S.blk = s->blk; S.status = s->status;
  S.id = (uint64_t *)s->id; S.rng = (uint64_t *)s->rng;
  return S;
}

static jb_status check_swarm(const jb_swarm_view *s, const char *who) {
  if (!s) return fail(JB_ERR_INVALID, "%s: null swarm", who);
  if (s->n < 0 || s->n > s->capacity) return fail(JB_ERR_INVALID, "%s: swarm n outside [0, capacity]", who);
  if (s->capacity > 0 && (!s->x || !s->y || !s->z || !s->vx || !s->vy || !s->vz || !s->t || !s->w ||
                          !s->e || !s->ip || !s->jp || !s->kp || !s->blk || !s->status || !s->id ||
                          !s->rng))
    return fail(JB_ERR_INVALID, "%s: null swarm array", who);
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
extern "C" jb_status jb_update_derived_transport_fields(jb_context *ctx, jb_mesh *mesh, double dt) {
  JB_RANGE("Jaybenne::UpdateDerivedTransportFields");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  const DevMesh &M = mesh->dm;
  const long long cells = (long long)M.nblocks * M.ncell;
  hipLaunchKernelGGL(k_fleck, dim3(grid_for(ctx, 64ll * M.nblocks * M.nx[1] * M.nx[2])), dim3(kBlock), 0,
                     ctx->stream, M, ctx->dp, dt);
  if (ctx->params.use_ddmc) {
    const int g = grid_for(ctx, cells * 2);
    hipLaunchKernelGGL(k_face_prob<0>, dim3(g), dim3(kBlock), 0, ctx->stream, M, ctx->dp);
    if (M.ndim > 1) hipLaunchKernelGGL(k_face_prob<1>, dim3(g), dim3(kBlock), 0, ctx->stream, M, ctx->dp);
    if (M.ndim > 2) hipLaunchKernelGGL(k_face_prob<2>, dim3(g), dim3(kBlock), 0, ctx->stream, M, ctx->dp);
    if (M.ddmc_cell) {
      // JB_NO_DDMC_ALL=1 keeps the general kernel also on all-DDMC meshes (tests, A/B)
      JB_HIP(hipMemsetAsync(M.not_all_ddmc, ctx->no_ddmc_all ? 1 : 0, sizeof(int), ctx->stream));
      JB_HIP(hipMemsetAsync(M.not_all_ddmc + 1, 0, sizeof(int), ctx->stream));   // the distinct step records are numbered afresh
      if (M.ddmc_class_slot) JB_HIP(hipMemsetAsync(M.ddmc_class_slot, 0, sizeof(int) * 2 * kClassSlots, ctx->stream));
      mesh->not_all_ddmc_host = -1;
      const int gp = grid_for(ctx, cells);
      const int mc = ctx->max_classes;
      if (M.ndim == 1) hipLaunchKernelGGL(k_ddmc_pack<1>, dim3(gp), dim3(kBlock), 0, ctx->stream, M, ctx->dp, mc);
      else if (M.ndim == 2) hipLaunchKernelGGL(k_ddmc_pack<2>, dim3(gp), dim3(kBlock), 0, ctx->stream, M, ctx->dp, mc);
      else hipLaunchKernelGGL(k_ddmc_pack<3>, dim3(gp), dim3(kBlock), 0, ctx->stream, M, ctx->dp, mc);
    }
  }
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_source_photons_count(jb_context *ctx, jb_mesh *mesh, int source_type,
                                             double dt, int blocks_in_call, uint32_t epoch,
                                             int32_t *nper_block_host, int32_t *prefix_dev) {
  JB_RANGE("Jaybenne::SourcePhotons1");
  if (!ctx || !mesh || !nper_block_host || !prefix_dev) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  if (ctx->params.source_strategy == JB_STRATEGY_ENERGY)
    return fail(JB_ERR_INVALID, "Energy source strategy not implemented!");  // sourcing.cpp:38
  if (epoch >= (1u << 20))  // cell_stream_id keeps the source-call counter in 20 bits
    return fail(JB_ERR_INVALID, "more than 2^20 source calls: the per-cell rounding streams would repeat");
  const DevMesh &M = mesh->dm;
  if (source_type == JB_SOURCE_EMISSION && !ctx->params.do_emission) {  // sourcing.cpp:41-43
    for (int b = 0; b < M.nblocks; ++b) nper_block_host[b] = 0;
    return JB_COMPLETE;
  }
  if (blocks_in_call < 1) return fail(JB_ERR_INVALID, "blocks_in_call must be >= 1");
  jb_status st = ensure_scratch(ctx, (size_t)M.nblocks);
  if (st != JB_COMPLETE) return st;
  // sourcing.cpp:68-69
  const double npc = (double)ctx->params.num_particles / (double)M.ncell /
                     (double)(blocks_in_call * M.nblocks_total);
  int *nper_d = (int *)ctx->scratch_d;
  hipLaunchKernelGGL(k_source_count, dim3(M.nblocks), dim3(kBlock), 0, ctx->stream, M, ctx->dp,
                     source_type, dt, npc, epoch, nper_d, prefix_dev);
  JB_HIP(hipGetLastError());
  JB_HIP(hipMemcpyAsync(nper_block_host, nper_d, sizeof(int) * M.nblocks, hipMemcpyDeviceToHost,
                        ctx->stream));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  return JB_COMPLETE;
}

extern "C" jb_status jb_source_photons_fill(jb_context *ctx, jb_mesh *mesh,
                                            const jb_swarm_view *swarm, int source_type,
                                            double t_start, double dt,
                                            const int32_t *nper_block_host,
                                            const int32_t *prefix_dev,
                                            const int64_t *slot_base_host,
                                            const uint64_t *id_base_host) {
  return jb_source_photons_fill_range(ctx, mesh, swarm, source_type, t_start, dt, nper_block_host, prefix_dev,
                                      slot_base_host, id_base_host, nullptr, 1);
}

extern "C" jb_status jb_source_photons_fill_range(jb_context *ctx, jb_mesh *mesh,
                                                  const jb_swarm_view *swarm, int source_type,
                                                  double t_start, double dt,
                                                  const int32_t *nper_block_host,
                                                  const int32_t *prefix_dev,
                                                  const int64_t *slot_base_host,
                                                  const uint64_t *id_base_host,
                                                  const int32_t *first_in_block_host,
                                                  int set_energy_delta) {
  JB_RANGE("Jaybenne::SourcePhotons2");
  if (!ctx || !mesh || !nper_block_host || !prefix_dev || !slot_base_host || !id_base_host)
    return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_source_photons_fill");
  if (st != JB_COMPLETE) return st;
  const DevMesh &M = mesh->dm;
  if (source_type == JB_SOURCE_EMISSION && !ctx->params.do_emission) return JB_COMPLETE;
  st = ensure_scratch(ctx, (size_t)M.nblocks * 5 + 8);
  if (st != JB_COMPLETE) return st;
  const int32_t *nper = nper_block_host;
  std::vector<long long> tab(4 * (size_t)M.nblocks);
  long long total = 0;
  for (int b = 0; b < M.nblocks; ++b) {
    tab[b] = total;                                   // blk_first
    tab[M.nblocks + b] = slot_base_host[b];           // slot_base
    tab[2 * M.nblocks + b] = (long long)id_base_host[b];
    tab[3 * M.nblocks + b] = first_in_block_host ? (long long)first_in_block_host[b] : 0ll;
    if (first_in_block_host && first_in_block_host[b] < 0)
      return fail(JB_ERR_INVALID, "negative first photon number for block %d", b);
    if (nper[b] < 0) return fail(JB_ERR_INVALID, "negative particle count for block %d", b);
    if (nper[b] > 0 && (slot_base_host[b] < 0 || slot_base_host[b] + nper[b] > swarm->capacity))
      return fail(JB_ERR_CAPACITY, "swarm capacity %lld too small for block %d (slots %lld..%lld)",
                  (long long)swarm->capacity, b, (long long)slot_base_host[b],
                  (long long)slot_base_host[b] + nper[b]);
    total += nper[b];
  }
  long long *tab_d = ctx->scratch_d + M.nblocks;  // after the int counts (nblocks ints fit in nblocks words)
  JB_HIP(hipMemcpyAsync(tab_d, tab.data(), sizeof(long long) * tab.size(), hipMemcpyHostToDevice,
                        ctx->stream));
  // (set_energy_delta = 0: energy_delta := 0 instead of minus the emitted energy -- every rank of a
  // replicated-mesh run but one, so that the sum over ranks carries the emission once)
  hipLaunchKernelGGL(k_source_edelta, dim3(grid_for(ctx, (long long)M.nblocks * M.ncell)),
                     dim3(kBlock), 0, ctx->stream, M, set_energy_delta ? source_type : 0);
  if (total > 0)
    hipLaunchKernelGGL(k_source_fill, dim3(grid_for(ctx, total)), dim3(kBlock), 0, ctx->stream, M,
                       ctx->dp, dev_swarm(swarm), source_type, t_start, dt, (const int *)prefix_dev,
                       (const long long *)tab_d, (const long long *)(tab_d + M.nblocks),
                       (const unsigned long long *)(tab_d + 2 * M.nblocks),
                       first_in_block_host ? (const long long *)(tab_d + 3 * M.nblocks) : (const long long *)nullptr,
                       total);
  JB_HIP(hipGetLastError());
  JB_HIP(hipStreamSynchronize(ctx->stream));  // tab lives on this stack frame
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
template <int NDIM, bool DDMC>
static jb_status launch_transport(jb_context *ctx, jb_mesh *mesh, const DevSwarm &S, double t_start,
                             double dt, long long first, long long last, bool tally) {
  const DevMesh &M = mesh->dm;
  // The kernel is persistent (waves draw particles from queues until they are empty), so the grid
  // is exactly what the chip holds at once: more workgroups would only start when the queues
  // are already drained.  JB_TRANSPORT_BLOCKS_PER_CU overrides the occupancy query (tuning aid).
  const int per_cu_env = ctx->blocks_per_cu_env;
  const bool gray = M.lam_abs != nullptr;
  (void)hipMemsetAsync(ctx->counters_d + CNT_QUEUE, 0, kQueues * kQueueStride * sizeof(unsigned long long), ctx->stream);
#define JB_LAUNCH_X(T, G, X, L)                                                                    \
  do {                                                                                             \
    static int occ = 0;  /* (one query per kernel instantiation and process) */                    \
    if (occ < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(                                  \
                        &occ, k_transport<NDIM, DDMC, T, G, X, L>, kBlock, 0) != hipSuccess || occ < 1)) \
      occ = 3;                                                                                     \
    const int g = grid_for(ctx, last - first, per_cu_env > 0 ? per_cu_env : occ);                  \
    const int *pair_flag = nullptr;                                                                \
    hipLaunchKernelGGL((k_transport<NDIM, DDMC, T, G, X, L>), dim3(g), dim3(kBlock), 0,            \
                       ctx->stream, M, ctx->dp, S, t_start, dt, first, last, ctx->counters_d,      \
                       pair_flag);                                                                 \
    mesh->last_variant = NDIM == 1 ? "k_transport<1, " #T ", " #G ", " #X ", " #L ">"              \
                         : NDIM == 2 ? "k_transport<2, " #T ", " #G ", " #X ", " #L ">"            \
                                     : "k_transport<3, " #T ", " #G ", " #X ", " #L ">";           \
  } while (0)
  // (variant string: NDIM, TALLY, GRAY, EXACT geometry, LEAN arithmetic; the DDMC flag is the
  // entry point that was called)
  // lean arithmetic on exact geometry: the step in cell-local coordinates (jb_kernel_imc.hpp)
  static const char *const imc_cell_names[3][2][2] = {
      {{"k_imc_cell<1, false, false, lean>", "k_imc_cell<1, false, true, lean>"},
       {"k_imc_cell<1, true, false, lean>", "k_imc_cell<1, true, true, lean>"}},
      {{"k_imc_cell<2, false, false, lean>", "k_imc_cell<2, false, true, lean>"},
       {"k_imc_cell<2, true, false, lean>", "k_imc_cell<2, true, true, lean>"}},
      {{"k_imc_cell<3, false, false, lean>", "k_imc_cell<3, false, true, lean>"},
       {"k_imc_cell<3, true, false, lean>", "k_imc_cell<3, true, true, lean>"}}};
  (void)imc_cell_names;
#define JB_LAUNCH_CELL_U(T, NA, U)                                                                 \
  do {                                                                                             \
    static int occ = 0;                                                                            \
    if (occ < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(                                  \
                        &occ, k_imc_cell<NDIM, T, NA, U>, kBlock, 0) != hipSuccess || occ < 1))    \
      occ = 3;                                                                                     \
    const int g = grid_for(ctx, last - first, per_cu_env > 0 ? per_cu_env : occ);                  \
    hipLaunchKernelGGL((k_imc_cell<NDIM, T, NA, U>), dim3(g), dim3(kBlock), 0, ctx->stream, mesh->dm_dev, \
                       ctx->dp, S, t_start, dt, first, last, ctx->counters_d, mesh->nbr_dq);       \
    /* (variant string: NDIM, TALLY, NOABS, and the arithmetic) */                                 \
    mesh->last_variant = imc_cell_names[NDIM - 1][(T) ? 1 : 0][(NA) ? 1 : 0];                      \
  } while (0)
#define JB_LAUNCH_CELL(T, NA)                                                                      \
  do {                                                                                             \
    if (mesh->uniform_geom) JB_LAUNCH_CELL_U(T, NA, true);                                         \
    else JB_LAUNCH_CELL_U(T, NA, false);                                                           \
  } while (0)
#define JB_LAUNCH(T, G)                                                                            \
  do {                                                                                             \
    if constexpr (!DDMC && G != 0) {                                                               \
      if (mesh->exact_geom) {                                                                      \
        if (ctx->lean_arith && !ctx->no_imc_cell && mesh->cell_ok) JB_LAUNCH_CELL(T, (G == 2));    \
        else if (ctx->lean_arith) JB_LAUNCH_X(T, G, true, true);                                   \
        else JB_LAUNCH_X(T, G, true, false);                                                       \
      } else {                                                                                     \
        /* (the cell-local step does not care what the cell widths are: only its conversions to and \
           from the swarm's coordinates round, by an ulp of the position) */                       \
        if (ctx->lean_arith && !ctx->no_imc_cell && mesh->cell_ok) JB_LAUNCH_CELL(T, (G == 2));    \
        else if (ctx->lean_arith) JB_LAUNCH_X(T, G, false, true);                                  \
        else JB_LAUNCH_X(T, G, false, false);                                                      \
      }                                                                                            \
    } else {                                                                                       \
      JB_LAUNCH_X(T, G, false, false);                                                             \
    }                                                                                              \
  } while (0)
  mesh->last_pair = "";
  if constexpr (DDMC) {
    // Gray opacities: UpdateDerivedTransportFields has packed the cell records and left a flag on
    // the device saying whether every cell takes DDMC steps.  Every cell: k_ddmc_all; a mix of IMC
    // and DDMC cells: k_hybrid.  Both keep the per-block tables in LDS, so meshes with more resident
    // blocks than fit there stay with the general kernel below.
    if (gray && M.ddmc_cell && M.nblocks <= kLdsBlocks) {
      if (mesh->not_all_ddmc_host < 0) {
        int *flag_h = (int *)(ctx->counters_h + kCounterWords - 1);
        flag_h[0] = 1; flag_h[1] = 0;
        (void)hipMemcpyAsync(flag_h, M.not_all_ddmc, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        JB_HIP(hipStreamSynchronize(ctx->stream));
        mesh->not_all_ddmc_host = flag_h[0] != 0 ? 1 : 0;
        mesh->nclass_host = flag_h[1];
      }
      const bool noabs_h = ctx->dp.kappa_a == 0.0;
      if (mesh->not_all_ddmc_host == 0) {
        // The quad-cooperative gather (jb_kernel_ddmc.hpp) addresses the step records with 32-bit byte
        // offsets; it pays once the records no longer sit in the CU's vector L1 (measured, ms per
        // 1e8 histories: 128^3 cells 31.5 -> 27.9, 64^3 7.2 -> 6.9, 32^3 equal, 128 cells in 1-D
        // 11.6 -> 13.1: there every lookup hits L1 and the detour through LDS only adds latency).
        const unsigned long long rec_bytes = 64ull * (unsigned long long)M.ntot * (unsigned long long)M.nblocks;
        // (record numbers are 32-bit: fewer than 2^32 resident cells; 64-bit addresses when the records
        // span 4 GiB or more -- JB_COOP_GATHER=2 forces that form on any table, for the parity tests)
        const bool coop = rec_bytes < (64ull << 32) &&
                          (ctx->coop_gather >= 0 ? ctx->coop_gather >= 1 : rec_bytes >= (1ull << 20));
        const bool wide = coop && (rec_bytes >= (1ull << 32) || ctx->coop_gather == 2);
        // ... and a mesh of at most kLdsRecCells cells (the reference's 1-D decks) keeps its records in
        // LDS: 12.3 -> 10.3 ms per 1e8 histories on BASELINE configs[2] as shipped
        const bool in_lds = !coop && ctx->coop_gather < 0 && (long long)M.nblocks * M.ntot <= (long long)kLdsRecCells;
        // ... and everything between with at most kMaxClasses DISTINCT step records (k_ddmc_pack counts them
        // every cycle: the gray decks have a handful) gathers a 4-byte cell code per step, the records in LDS
        // (JB_COOP_GATHER=4 also on the smallest meshes; 0 / 1 / 2 keep the 64-byte forms, for tests and A/B)
        const bool codes_ok = M.ddmc_code != nullptr && mesh->nclass_host >= 1 && mesh->nclass_host <= ctx->max_classes;
        // ... and, with the codes, the wave's photons staged through queues in LDS (k_ddmc_q, jb_kernel_ddmc_q.hpp:
        // the event loop at full width, the service phase in whole batches) -- any mesh size, up to kQBlocks
        // resident blocks and 2^32 slots; JB_DDMC_QUEUES=0 keeps k_ddmc_all
        const bool queues = codes_ok && ctx->ddmc_queues && ctx->coop_gather < 0 && M.nblocks <= kQBlocks &&
                            last <= (1ll << 32);
        const bool codes = queues || (codes_ok && (ctx->coop_gather == 4 || (ctx->coop_gather < 0 && !in_lds)));
        const int gather = codes ? 4 : (coop ? (wide ? 3 : 1) : (in_lds ? 2 : 0));
        // ... the codes themselves in LDS on a mesh of at most kLdsCodeCells cells (JB_DDMC_LDS_CODES=0: not)
        // (and at most 64 classes: tally 8 KB + classes 4 KB + codes 4 KB + 37.7 KB static stay under the 64 KB a
        // workgroup may have; with up to kMaxClasses = 256 records, 16 KB, the codes would not fit beside them)
        const bool lcodes = queues && ctx->ddmc_lds_codes && (long long)M.nblocks * M.ntot <= (long long)kLdsCodeCells &&
                            mesh->nclass_host <= 64;
        // A particle that sits at a face of its cell when it is loaded or relocated (one in ~1e8) needs
        // the albedo step: k_ddmc_all lists it, and k_hybrid<.., both loops>, launched behind it on
        // that list (its length read on the device: no synchronisation), tracks it to the end.
        {
          const jb_status st_s = ensure_scratch(ctx, (size_t)(last - first) / 2 + 16);
          if (st_s != JB_COMPLETE) return st_s;
        }
        unsigned *handed = (unsigned *)ctx->scratch_d;
        unsigned long long *n_handed = ctx->counters_d + kCursorBase;
        // (dynamic shared memory: the tally of a mesh with <= kLdsTally cells, resident blocks' ghosts included)
        const size_t ncell_all = (size_t)M.nblocks * (size_t)M.ntot;
        const size_t lds_tally_bytes =
            (tally && (long long)ncell_all <= (long long)kLdsTally ? sizeof(double) * ((ncell_all + 1) / 2 * 2) : 0) +
            (codes ? 64 * (size_t)mesh->nclass_host : (in_lds ? 64 * ncell_all : 0)) +
            (lcodes ? sizeof(unsigned) * ((ncell_all + 1) / 2 * 2) : 0);
        (void)hipMemsetAsync(n_handed, 0, sizeof(unsigned long long), ctx->stream);
#define JB_LAUNCH_DDMC_ALL(TL, CO)                                                                          \
  do {                                                                                                      \
    /* (the dynamic LDS -- tally and record table of a small mesh -- changes with the mesh: ask per launch) */ \
    int oc = 0;                                                                                             \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, k_ddmc_all<NDIM, TL, CO>, kBlock, lds_tally_bytes) != hipSuccess || oc < 1) oc = 3; \
    /* (four 16-byte loads per lane on a table that does not sit in L1 -- only a table of >= 4 GiB, or  \
       JB_COOP_GATHER=0, gets here -- saturate the vector L1's look-ups: a fourth wave per SIMD then  \
       costs time, 38.2 against 31.5 ms per 1e8 histories on the 160 MB table) */                      \
    if ((CO) == 0 && rec_bytes >= (1ull << 20) && oc > 3) oc = 3;                                           \
    const int g = grid_for(ctx, last - first, per_cu_env > 0 ? per_cu_env : oc);                            \
    hipLaunchKernelGGL((k_ddmc_all<NDIM, TL, CO>), dim3(g), dim3(kBlock), lds_tally_bytes, ctx->stream, mesh->dm_dev, ctx->dp, S, \
                       t_start, dt, first, last, ctx->counters_d, (const int *)M.not_all_ddmc, handed, n_handed); \
  } while (0)
#define JB_LAUNCH_DDMC_Q1(TL, LC)                                                                          \
  do {                                                                                                      \
    int oc = 0;                                                                                             \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&oc, k_ddmc_q<NDIM, TL, LC>, kBlock, lds_tally_bytes) != hipSuccess || oc < 1) oc = 3; \
    const int g = grid_for(ctx, last - first, per_cu_env > 0 ? per_cu_env : oc);                            \
    hipLaunchKernelGGL((k_ddmc_q<NDIM, TL, LC>), dim3(g), dim3(kBlock), lds_tally_bytes, ctx->stream, mesh->dm_dev, ctx->dp, S, \
                       t_start, dt, first, last, ctx->counters_d, (const int *)M.not_all_ddmc, handed, n_handed); \
  } while (0)
#define JB_LAUNCH_DDMC_Q(TL)                                                                               \
  do {                                                                                                      \
    if (lcodes) JB_LAUNCH_DDMC_Q1(TL, true);                                                                \
    else JB_LAUNCH_DDMC_Q1(TL, false);                                                                      \
  } while (0)
#define JB_LAUNCH_HANDED(TL, NA)                                                                            \
  do {                                                                                                      \
    (void)hipMemsetAsync(ctx->counters_d + CNT_QUEUE, 0, kQueues * kQueueStride * sizeof(unsigned long long), ctx->stream); \
    hipLaunchKernelGGL((k_hybrid<NDIM, TL, NA, 0, 0>), dim3(kQueues), dim3(kBlock), 0, ctx->stream, mesh->dm_dev, \
                       ctx->dp, S, t_start, dt, 0ll, (long long)(last - first), ctx->counters_d,            \
                       (const unsigned *)handed, (unsigned *)nullptr, (unsigned long long *)nullptr,         \
                       (const unsigned long long *)n_handed);                                               \
  } while (0)
        static const char *const names[3][2][4] = {
            {{"k_ddmc_all<1, false>", "k_ddmc_all<1, false, quad gather>", "k_ddmc_all<1, false, records in LDS>", "k_ddmc_all<1, false, cell codes>"},
             {"k_ddmc_all<1, true>", "k_ddmc_all<1, true, quad gather>", "k_ddmc_all<1, true, records in LDS>", "k_ddmc_all<1, true, cell codes>"}},
            {{"k_ddmc_all<2, false>", "k_ddmc_all<2, false, quad gather>", "k_ddmc_all<2, false, records in LDS>", "k_ddmc_all<2, false, cell codes>"},
             {"k_ddmc_all<2, true>", "k_ddmc_all<2, true, quad gather>", "k_ddmc_all<2, true, records in LDS>", "k_ddmc_all<2, true, cell codes>"}},
            {{"k_ddmc_all<3, false>", "k_ddmc_all<3, false, quad gather>", "k_ddmc_all<3, false, records in LDS>", "k_ddmc_all<3, false, cell codes>"},
             {"k_ddmc_all<3, true>", "k_ddmc_all<3, true, quad gather>", "k_ddmc_all<3, true, records in LDS>", "k_ddmc_all<3, true, cell codes>"}}};
        static const char *const qnames[2][3][2] = {
            {{"k_ddmc_all<1, false, cell codes, queues>", "k_ddmc_all<1, true, cell codes, queues>"},
             {"k_ddmc_all<2, false, cell codes, queues>", "k_ddmc_all<2, true, cell codes, queues>"},
             {"k_ddmc_all<3, false, cell codes, queues>", "k_ddmc_all<3, true, cell codes, queues>"}},
            {{"k_ddmc_all<1, false, cell codes, queues, codes in LDS>", "k_ddmc_all<1, true, cell codes, queues, codes in LDS>"},
             {"k_ddmc_all<2, false, cell codes, queues, codes in LDS>", "k_ddmc_all<2, true, cell codes, queues, codes in LDS>"},
             {"k_ddmc_all<3, false, cell codes, queues, codes in LDS>", "k_ddmc_all<3, true, cell codes, queues, codes in LDS>"}}};
        mesh->last_variant = queues ? qnames[lcodes ? 1 : 0][NDIM - 1][tally ? 1 : 0]
                                    : names[NDIM - 1][tally ? 1 : 0][gather == 4 ? 3 : (gather == 3 ? 1 : gather)];
        if (tally) {
          if (queues) JB_LAUNCH_DDMC_Q(true);
          else if (gather == 4) JB_LAUNCH_DDMC_ALL(true, 4);
          else if (gather == 1) JB_LAUNCH_DDMC_ALL(true, 1);
          else if (gather == 2) JB_LAUNCH_DDMC_ALL(true, 2);
          else if (gather == 3) JB_LAUNCH_DDMC_ALL(true, 3);
          else JB_LAUNCH_DDMC_ALL(true, 0);
          if (noabs_h) JB_LAUNCH_HANDED(true, true);
          else JB_LAUNCH_HANDED(true, false);
        } else {
          if (queues) JB_LAUNCH_DDMC_Q(false);
          else if (gather == 4) JB_LAUNCH_DDMC_ALL(false, 4);
          else if (gather == 1) JB_LAUNCH_DDMC_ALL(false, 1);
          else if (gather == 2) JB_LAUNCH_DDMC_ALL(false, 2);
          else if (gather == 3) JB_LAUNCH_DDMC_ALL(false, 3);
          else JB_LAUNCH_DDMC_ALL(false, 0);
          if (noabs_h) JB_LAUNCH_HANDED(false, true);
          else JB_LAUNCH_HANDED(false, false);
        }
#undef JB_LAUNCH_HANDED
#undef JB_LAUNCH_DDMC_Q
#undef JB_LAUNCH_DDMC_Q1
#undef JB_LAUNCH_DDMC_ALL
        return JB_COMPLETE;
      }
      // A mix of IMC and DDMC cells, three launches: k_hybrid<.., PHASE 1> follows the photons in
      // IMC cells (its service phase also takes the albedo step of a photon that enters a DDMC
      // cell) and parks those that settle in DDMC cells; <.., PHASE 2> follows these and parks the
      // ones that leak back into IMC cells; <.., PHASE 0>, which runs both event loops, finishes
      // that remainder -- the photons that keep changing regime at the interface (alternating
      // phases 1 and 2 until nothing is left costs one launch per change of regime of the most
      // persistent photon: measured ~80 rounds of ~0.5 ms on BASELINE configs[4]).  The lists of
      // parked photons (slot numbers, 4 bytes each) live in the context's scratch buffer.
      // (variant string: NDIM, TALLY, NOABS, MODE: 0 exact arithmetic, 1 lean, 2 lean on exact geometry)
      const long long nrange = last - first;
      {  // 2 lists x 4 bytes x n
        const jb_status st_s = ensure_scratch(ctx, (size_t)nrange + 16);
        if (st_s != JB_COMPLETE) return st_s;
      }
      unsigned *list_d = (unsigned *)ctx->scratch_d;          // parked by phase 1, read by phase 2
      unsigned *list_i = list_d + nrange;                     // parked by phase 2, read by phase 1
      unsigned long long *cnt = ctx->counters_d + kCursorBase;  // [0] |list_d|, [1] |list_i|
      volatile unsigned long long *cnt_h = ctx->counters_h + kCursorBase;
#define JB_LAUNCH_H(T, NA, MD, PH, F, L, LIN, LOUT, COUT)                                          \
  do {                                                                                             \
    static int occ = 0;                                                                            \
    if (occ < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_hybrid<NDIM, T, NA, MD, PH>, \
                                                                 kBlock, 0) != hipSuccess || occ < 1)) \
      occ = 3;                                                                                     \
    const int g = grid_for(ctx, (L) - (F), per_cu_env > 0 ? per_cu_env : occ);                     \
    (void)hipMemsetAsync(ctx->counters_d + CNT_QUEUE, 0, kQueues * kQueueStride * sizeof(unsigned long long), ctx->stream); \
    hipLaunchKernelGGL((k_hybrid<NDIM, T, NA, MD, PH>), dim3(g), dim3(kBlock), 0, ctx->stream,     \
                       mesh->dm_dev, ctx->dp, S, t_start, dt, (long long)(F), (long long)(L), ctx->counters_d,   \
                       (const unsigned *)(LIN), (unsigned *)(LOUT), (unsigned long long *)(COUT),  \
                       (const unsigned long long *)nullptr);                                       \
  } while (0)
#define JB_PHASE1(T, NA, F, L, LIN)                                                                \
  do {                                                                                             \
    if (!ctx->lean_arith) JB_LAUNCH_H(T, NA, 0, 1, F, L, LIN, list_d, cnt);                        \
    else if (mesh->exact_geom && mesh->cell_ok && !ctx->no_imc_cell) JB_LAUNCH_H(T, NA, 3, 1, F, L, LIN, list_d, cnt); \
    else if (mesh->exact_geom) JB_LAUNCH_H(T, NA, 2, 1, F, L, LIN, list_d, cnt);                            \
    else JB_LAUNCH_H(T, NA, 1, 1, F, L, LIN, list_d, cnt);                                         \
  } while (0)
      {
        const bool cell = mesh->exact_geom && mesh->cell_ok && !ctx->no_imc_cell;
        static const char *const hyb_names[3][4] = {
            {"k_hybrid<1, exact>", "k_hybrid<1, lean>", "k_hybrid<1, lean, exact geometry>", "k_hybrid<1, lean, cell-local>"},
            {"k_hybrid<2, exact>", "k_hybrid<2, lean>", "k_hybrid<2, lean, exact geometry>", "k_hybrid<2, lean, cell-local>"},
            {"k_hybrid<3, exact>", "k_hybrid<3, lean>", "k_hybrid<3, lean, exact geometry>", "k_hybrid<3, lean, cell-local>"}};
        mesh->last_variant = hyb_names[NDIM - 1][!ctx->lean_arith ? 0 : (cell ? 3 : (mesh->exact_geom ? 2 : 1))];
      }
#define JB_PHASE0(T, NA, L, LIN)                                                                   \
  do {                                                                                             \
    if (!ctx->lean_arith) JB_LAUNCH_H(T, NA, 0, 0, 0, L, LIN, nullptr, nullptr);                   \
    else if (mesh->exact_geom && mesh->cell_ok && !ctx->no_imc_cell) JB_LAUNCH_H(T, NA, 3, 0, 0, L, LIN, nullptr, nullptr); \
    else if (mesh->exact_geom) JB_LAUNCH_H(T, NA, 2, 0, 0, L, LIN, nullptr, nullptr);                       \
    else JB_LAUNCH_H(T, NA, 1, 0, 0, L, LIN, nullptr, nullptr);                                    \
  } while (0)
      (void)hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), ctx->stream);
      if (tally) { if (noabs_h) JB_PHASE1(true, true, first, last, nullptr); else JB_PHASE1(true, false, first, last, nullptr); }
      else { if (noabs_h) JB_PHASE1(false, true, first, last, nullptr); else JB_PHASE1(false, false, first, last, nullptr); }
      (void)hipMemcpyAsync((void *)cnt_h, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
      JB_HIP(hipStreamSynchronize(ctx->stream));
      const long long n_d = (long long)cnt_h[0];
      if (n_d == 0) return JB_COMPLETE;
      if (tally) JB_LAUNCH_H(true, true, 0, 2, 0, n_d, list_d, list_i, cnt + 1);
      else JB_LAUNCH_H(false, true, 0, 2, 0, n_d, list_d, list_i, cnt + 1);
      (void)hipMemcpyAsync((void *)(cnt_h + 1), cnt + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
      JB_HIP(hipStreamSynchronize(ctx->stream));
      const long long n_i = (long long)cnt_h[1];
      if (n_i == 0) return JB_COMPLETE;
      if (tally) { if (noabs_h) JB_PHASE0(true, true, n_i, list_i); else JB_PHASE0(true, false, n_i, list_i); }
      else { if (noabs_h) JB_PHASE0(false, true, n_i, list_i); else JB_PHASE0(false, false, n_i, list_i); }
#undef JB_PHASE0
#undef JB_PHASE1
#undef JB_LAUNCH_H
      return JB_COMPLETE;
    }
  }
  // gray opacity with kappa = 0 (opacity_model = none): sigma_a = rho * 0 in every cell
  const bool noabs = gray && ctx->dp.kappa_a == 0.0;
  if (noabs) {
    if (tally) JB_LAUNCH(true, 2);
    else JB_LAUNCH(false, 2);
    return JB_COMPLETE;
  }
  if (tally && gray) JB_LAUNCH(true, 1);
  else if (tally) JB_LAUNCH(true, 0);
  else if (gray) JB_LAUNCH(false, 1);
  else JB_LAUNCH(false, 0);
#undef JB_LAUNCH
#undef JB_LAUNCH_CELL
#undef JB_LAUNCH_CELL_U
#undef JB_LAUNCH_X
  return JB_COMPLETE;
}

static jb_status transport_impl(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                double t_start, double dt, int64_t first, int64_t last, int tally,
                                bool ddmc) {
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_transport_photons");
  if (st != JB_COMPLETE) return st;
  if (first < 0 || last > swarm->n || first > last)
    return fail(JB_ERR_INVALID, "particle range [%lld,%lld) outside the swarm (n = %lld)",
                (long long)first, (long long)last, (long long)swarm->n);
  const DevMesh &M = mesh->dm;
  if (ddmc && (!M.P1 || (M.ndim > 1 && !M.P2) || (M.ndim > 2 && !M.P3)))
    return fail(JB_ERR_INVALID, "DDMC transport needs the ddmc_face_prob arrays");
  if (first == last) return JB_COMPLETE;
  const DevSwarm S = dev_swarm(swarm);
  const bool tl = tally != 0;
  // (jb_defrag_policy: the time of the tracking kernels of this cycle, per event)
  hipEvent_t ev_stop = nullptr;
  if (ctx->tev_used + 2 <= 2 * kTransportEventPairs) {
    while ((int)ctx->tev.size() < ctx->tev_used + 2) {
      hipEvent_t e = nullptr;
      JB_HIP(hipEventCreate(&e));
      ctx->tev.push_back(e);
    }
    JB_HIP(hipEventRecord(ctx->tev[ctx->tev_used], ctx->stream));
    ev_stop = ctx->tev[ctx->tev_used + 1];
    ctx->tev_used += 2;
  } else {
    ctx->tev_overflow = true;
  }
  switch (M.ndim * 2 + (ddmc ? 1 : 0)) {
  case 2: st = launch_transport<1, false>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  case 3: st = launch_transport<1, true>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  case 4: st = launch_transport<2, false>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  case 5: st = launch_transport<2, true>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  case 6: st = launch_transport<3, false>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  default: st = launch_transport<3, true>(ctx, mesh, S, t_start, dt, first, last, tl); break;
  }
  if (ev_stop) (void)hipEventRecord(ev_stop, ctx->stream);
  if (st != JB_COMPLETE) return st;
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_transport_photons(jb_context *ctx, jb_mesh *mesh,
                                          const jb_swarm_view *swarm, double t_start, double dt,
                                          int64_t first, int64_t last, int fuse_census_tally) {
  JB_RANGE("Jaybenne::TransportPhotons");
  return transport_impl(ctx, mesh, swarm, t_start, dt, first, last, fuse_census_tally, false);
}
extern "C" jb_status jb_transport_photons_ddmc(jb_context *ctx, jb_mesh *mesh,
                                               const jb_swarm_view *swarm, double t_start, double dt,
                                               int64_t first, int64_t last, int fuse_census_tally) {
  JB_RANGE("Jaybenne::TransportPhotons_DDMC");
  return transport_impl(ctx, mesh, swarm, t_start, dt, first, last, fuse_census_tally, true);
}

extern "C" const char *jb_last_transport_variant(const jb_mesh *mesh) {
  return mesh ? mesh->last_variant : "";
}
extern "C" int jb_mesh_exact_geometry(const jb_mesh *mesh) { return mesh && mesh->exact_geom; }
extern "C" int jb_mesh_ddmc_classes(const jb_mesh *mesh) { return mesh ? mesh->nclass_host : 0; }
extern "C" jb_status jb_set_arithmetic(jb_context *ctx, int mode) {
  if (!ctx || (mode != JB_ARITH_EXACT && mode != JB_ARITH_LEAN)) return fail(JB_ERR_INVALID, "bad argument");
  ctx->lean_arith = mode == JB_ARITH_LEAN;
  ctx->dp.lean = ctx->lean_arith ? 1 : 0;
  return JB_COMPLETE;
}
extern "C" int jb_get_arithmetic(const jb_context *ctx) {
  return ctx && ctx->lean_arith ? JB_ARITH_LEAN : JB_ARITH_EXACT;
}

static jb_status fetch_counters(jb_context *ctx) {
  JB_HIP(hipMemcpyAsync(ctx->counters_h, ctx->counters_d, sizeof(unsigned long long) * kRankBase,
                        hipMemcpyDeviceToHost, ctx->stream));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  return JB_COMPLETE;
}

extern "C" jb_status jb_get_transport_stats(jb_context *ctx, jb_transport_stats *stats, int reset) {
  if (!ctx || !stats) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = fetch_counters(ctx);
  if (st != JB_COMPLETE) return st;
  stats->n_census = (int64_t)ctx->counters_h[CNT_CENSUS];
  stats->n_absorbed = (int64_t)ctx->counters_h[CNT_ABSORBED];
  stats->n_escaped = (int64_t)ctx->counters_h[CNT_ESCAPED];
  stats->n_outgoing = (int64_t)ctx->counters_h[CNT_OUTGOING];
  stats->n_events = (int64_t)ctx->counters_h[CNT_EVENTS];
  stats->n_wave_passes = (int64_t)ctx->counters_h[CNT_PASSES];
  stats->n_wave_services = (int64_t)ctx->counters_h[CNT_SERVICE];
#ifdef JB_TIMING  // (diagnostic build: wave-cycles / 1024 per sub-phase of k_ddmc_all's service phase)
  fprintf(stderr, "JB_TIMING phases reloc %llu claim %llu done %llu take %llu real %llu | episodes %llu passes %llu services %llu\n",
          ctx->counters_h[24], ctx->counters_h[25], ctx->counters_h[26], ctx->counters_h[27],
          ctx->counters_h[28], ctx->counters_h[29], ctx->counters_h[30], ctx->counters_h[31]);
#ifdef JB_TIMING_LOOP
  fprintf(stderr, "JB_TIMING_LOOP top %llu code wait %llu step %llu retire+tail %llu\n", ctx->counters_h[20],
          ctx->counters_h[21], ctx->counters_h[22], ctx->counters_h[23]);
#endif
#endif
#ifdef JB_HYB_STATS  // (diagnostic build of k_hybrid: see jb_kernel_hybrid.hpp)
  {
    unsigned long long h[48];
    (void)hipMemcpy(h, ctx->counters_d + 64, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[9] = {"idle", "imc", "virt", "real", "done", "reloc", "emerge", "new", "park"};
    for (int ph = 0; ph < 3; ++ph) {
      fprintf(stderr, "JB_HYB_STATS phase %d: services %llu imc passes %llu (lanes %llu) ddmc passes %llu | lanes at service:", ph,
              h[16 * ph + 12], h[16 * ph + 9], h[16 * ph + 11], h[16 * ph + 10]);
      for (int k = 0; k < 9; ++k) fprintf(stderr, " %s %llu", nm[k], h[16 * ph + k]);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (reset) JB_HIP(hipMemsetAsync(ctx->counters_d, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
  return JB_COMPLETE;
}

extern "C" jb_status jb_sample_ddmc_block_face(jb_context *ctx, jb_mesh *mesh,
                                               const jb_swarm_view *swarm, int64_t first,
                                               int64_t last) {
  JB_RANGE("Jaybenne::SampleDDMCBlockFace");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_sample_ddmc_block_face");
  if (st != JB_COMPLETE) return st;
  const DevMesh &M = mesh->dm;
  if (!(M.ndim > 1)) return JB_COMPLETE;  // sample_ddmc_bface.cpp:90
  if (first < 0 || last > swarm->n || first > last) return fail(JB_ERR_INVALID, "bad particle range");
  if (!M.P1 || !M.P2 || (M.ndim > 2 && !M.P3)) return fail(JB_ERR_INVALID, "needs ddmc_face_prob");
  if (first == last) return JB_COMPLETE;
  const int g = grid_for(ctx, last - first);
  if (M.ndim == 2)
    hipLaunchKernelGGL(k_block_face<2>, dim3(g), dim3(kBlock), 0, ctx->stream, M, ctx->dp, dev_swarm(swarm), first, last);
  else
    hipLaunchKernelGGL(k_block_face<3>, dim3(g), dim3(kBlock), 0, ctx->stream, M, ctx->dp, dev_swarm(swarm), first, last);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_check_completion(jb_context *ctx, const jb_swarm_view *swarm, double t_end,
                                         int64_t *unfinished) {
  JB_RANGE("Jaybenne::CheckCompletion");
  if (!ctx || !unfinished) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_check_completion");
  if (st != JB_COMPLETE) return st;
  JB_HIP(hipMemsetAsync(&ctx->counters_d[CNT_UNFINISHED], 0, sizeof(unsigned long long), ctx->stream));
  if (swarm->n > 0)
    hipLaunchKernelGGL(k_check_completion, dim3(grid_for(ctx, swarm->n)), dim3(kBlock), 0, ctx->stream,
                       dev_swarm(swarm), (long long)swarm->n, t_end, ctx->counters_d);
  JB_HIP(hipGetLastError());
  st = fetch_counters(ctx);
  if (st != JB_COMPLETE) return st;
  *unfinished = (int64_t)ctx->counters_h[CNT_UNFINISHED];
  return *unfinished > 0 ? JB_ITERATE : JB_COMPLETE;
}

extern "C" jb_status jb_zero_energy_tally(jb_context *ctx, jb_mesh *mesh) {
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  const DevMesh &M = mesh->dm;
  hipLaunchKernelGGL(k_zero_tally, dim3(grid_for(ctx, (long long)M.nblocks * M.ncell)), dim3(kBlock), 0,
                     ctx->stream, M);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_evaluate_radiation_energy(jb_context *ctx, jb_mesh *mesh,
                                                  const jb_swarm_view *swarm) {
  JB_RANGE("Jaybenne::EvaluateRadiationEnergy");
  jb_status st = jb_zero_energy_tally(ctx, mesh);
  if (st != JB_COMPLETE) return st;
  st = check_swarm(swarm, "jb_evaluate_radiation_energy");
  if (st != JB_COMPLETE) return st;
  if (swarm->n > 0)
    hipLaunchKernelGGL(k_tally, dim3(grid_for(ctx, swarm->n)), dim3(kBlock), 0, ctx->stream, mesh->dm,
                       dev_swarm(swarm), (long long)swarm->n);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_update_fluid(jb_context *ctx, jb_mesh *mesh) {
  JB_RANGE("Jaybenne::UpdateFluid");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  if (!ctx->params.do_feedback) return JB_COMPLETE;  // jaybenne.cpp:590
  const DevMesh &M = mesh->dm;
  hipLaunchKernelGGL(k_update_fluid, dim3(grid_for(ctx, (long long)M.nblocks * M.ncell)), dim3(kBlock), 0,
                     ctx->stream, M);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_photon_reflect_bc(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                          int face) {
  JB_RANGE("Jaybenne::PhotonReflectBC");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  if (face < 0 || face > 5) return fail(JB_ERR_INVALID, "face must be 0..5");
  jb_status st = check_swarm(swarm, "jb_photon_reflect_bc");
  if (st != JB_COMPLETE) return st;
  if (swarm->n > 0)
    hipLaunchKernelGGL(k_reflect_bc, dim3(grid_for(ctx, swarm->n)), dim3(kBlock), 0, ctx->stream, mesh->dm,
                       dev_swarm(swarm), (long long)swarm->n, face);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_remove_marked_particles(jb_context *ctx, jb_swarm_view *swarm) {
  JB_RANGE("Jaybenne::RemoveMarkedParticles");
  if (!ctx) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_remove_marked_particles");
  if (st != JB_COMPLETE) return st;
  const long long n = swarm->n;
  if (n == 0) return JB_COMPLETE;
  const DevSwarm S = dev_swarm(swarm);
  unsigned long long *cur = ctx->counters_d + kCursorBase;
  JB_HIP(hipMemsetAsync(cur, 0, 4 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_count_active, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, S, n, cur + 2);
  JB_HIP(hipGetLastError());
  st = fetch_counters(ctx);
  if (st != JB_COMPLETE) return st;
  const long long survivors = (long long)ctx->counters_h[kCursorBase + 2];
  const long long removed = n - survivors;
  if (removed > 0 && survivors > 0) {
    const long long maxlist = removed < survivors ? removed : survivors;
    st = ensure_scratch(ctx, (size_t)(2 * maxlist));
    if (st != JB_COMPLETE) return st;
    long long *holes = ctx->scratch_d, *movers = ctx->scratch_d + maxlist;
    hipLaunchKernelGGL(k_list_holes_movers, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, S, n,
                       survivors, holes, movers, cur);
    JB_HIP(hipGetLastError());
    st = fetch_counters(ctx);
    if (st != JB_COMPLETE) return st;
    const long long nh = (long long)ctx->counters_h[kCursorBase], nm = (long long)ctx->counters_h[kCursorBase + 1];
    if (nh != nm) return fail(JB_ERR_INVALID, "compaction lists disagree (%lld holes, %lld movers)", nh, nm);
    if (nh > 0)
      hipLaunchKernelGGL(k_fill_holes, dim3(grid_for(ctx, nh)), dim3(kBlock), 0, ctx->stream, S, holes,
                         movers, nh);
    JB_HIP(hipGetLastError());
  }
  swarm->n = survivors;
  return JB_COMPLETE;
}

extern "C" jb_status jb_defrag_particles(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm) {
  JB_RANGE("Jaybenne::DefragParticles");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_defrag_particles");
  if (st != JB_COMPLETE) return st;
  const long long n = swarm->n;
  if (n == 0) return JB_COMPLETE;
  if (n >= (1ll << 32)) return fail(JB_ERR_INVALID, "jb_defrag_particles: more than 2^32 - 1 particles");
  const DevMesh &M = mesh->dm;
  const unsigned long long nkeys64 = (unsigned long long)M.nblocks * (unsigned long long)M.ntot;
  if (nkeys64 >= (1ull << 32) - 1ull) return fail(JB_ERR_INVALID, "jb_defrag_particles: more than 2^32 - 2 cells");
  const unsigned nkeys = (unsigned)nkeys64;
  const long long nbins = (long long)nkeys + 1;                       // + the bin behind all cells
  const int ntiles = (int)((nbins + kScanTile - 1) / kScanTile);
  // scratch: the particle records (16 words each, on a 128-byte boundary), then histogram / offsets
  // (nbins), tile sums (ntiles) and keys (n), 4 bytes each
  const size_t rec_words = (size_t)kSortRecWords * (size_t)n;
  st = ensure_scratch(ctx, rec_words + (size_t)((nbins + ntiles + n) / 2 + 16), /*slack=*/false);
  if (st != JB_COMPLETE) return st;
  unsigned long long *rec = (unsigned long long *)ctx->scratch_d;
  unsigned *hist = (unsigned *)(rec + rec_words);
  unsigned *sums = hist + nbins;
  unsigned *key = sums + ntiles;
  JB_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)nbins, ctx->stream));
  const DevSwarm S = dev_swarm(swarm);
  hipLaunchKernelGGL(k_sort_count, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, M, S, n, nkeys, key, hist);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kBlock), 0, ctx->stream, hist, nbins, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(kBlock), 0, ctx->stream, hist, nbins, (const unsigned *)sums);
  hipLaunchKernelGGL(k_sort_pack, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, S, n, (const unsigned *)key,
                     hist, rec);
  hipLaunchKernelGGL(k_sort_unpack, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, S, n,
                     (const unsigned long long *)rec);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

// the sort of jb_defrag_policy, timed, and the policy's state behind it
static jb_status defrag_now(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm, double rate, int32_t *sorted) {
  if (!ctx->sort_ev[0]) {
    JB_HIP(hipEventCreate(&ctx->sort_ev[0]));
    JB_HIP(hipEventCreate(&ctx->sort_ev[1]));
  }
  (void)hipEventRecord(ctx->sort_ev[0], ctx->stream);
  ctx->scratch_alloc_failed = false;
  jb_status st = jb_defrag_particles(ctx, mesh, swarm);
  // (only "no room for the sort's scratch records" lets the run go on unsorted: any other failure -- a
  // launch or a kernel of the sort itself -- may have left the swarm partly permuted and is the caller's)
  if (st == JB_ERR_HIP && ctx->scratch_alloc_failed) {
    fprintf(stderr, "jaybenne_amd: DefragParticles skipped (%s)\n", g_err);
    (void)hipGetLastError();
    ctx->min_interval = 256;
    ctx->cycles_since_sort = 0;
    ctx->excess_ms = 0.0;
    return JB_COMPLETE;
  }
  if (st != JB_COMPLETE) return st;
  (void)hipEventRecord(ctx->sort_ev[1], ctx->stream);
  ctx->sort_timed = true;
  ctx->sort_n = swarm->n;
  ctx->rate_before_sort = rate;
  ctx->rate_ref = 0.0;
  ctx->excess_ms = 0.0;
  ctx->cycles_since_sort = 0;
  ++ctx->policy_sorts;
  *sorted = 1;
  return JB_COMPLETE;
}

extern "C" jb_status jb_release_scratch(jb_context *ctx) {
  if (!ctx) return fail(JB_ERR_INVALID, "null context");
  JB_HIP(hipSetDevice(ctx->device));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->scratch_d) JB_HIP(hipFree(ctx->scratch_d));
  ctx->scratch_d = nullptr;
  ctx->scratch_words = 0;
  return JB_COMPLETE;
}

extern "C" jb_status jb_defrag_policy(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                       int64_t events_this_cycle, int32_t mode, int32_t *sorted) {
  if (!ctx || !mesh || !sorted) return fail(JB_ERR_INVALID, "null argument");
  if (mode < JB_DEFRAG_DECIDE_AND_SORT || mode > JB_DEFRAG_SORT_NOW) return fail(JB_ERR_INVALID, "unknown mode");
  *sorted = 0;
  if (mode == JB_DEFRAG_SORT_NOW) {  // (the ranks have agreed: the cycle's times were taken by the DECIDE call)
    JB_HIP(hipSetDevice(ctx->device));
    jb_status st2 = check_swarm(swarm, "jb_defrag_policy");
    if (st2 != JB_COMPLETE) return st2;
    return defrag_now(ctx, mesh, swarm, ctx->last_rate, sorted);
  }
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_defrag_policy");
  if (st != JB_COMPLETE) return st;
  // time of this cycle's tracking kernels (the caller has synchronised: it knows the event count)
  double ms = 0.0;
  bool ok = !ctx->tev_overflow && ctx->tev_used > 0;
  for (int q = 0; ok && q + 1 < ctx->tev_used; q += 2) {
    float t = 0.0f;
    if (hipEventElapsedTime(&t, ctx->tev[q], ctx->tev[q + 1]) != hipSuccess) ok = false;
    ms += (double)t;
  }
  (void)hipGetLastError();  // (an event that has not completed: not an error of the library's)
  ctx->tev_used = 0;
  ctx->tev_overflow = false;
  if (ctx->sort_timed) {  // what the last sort cost (it was queued behind the previous cycle)
    float t = 0.0f;
    if (hipEventElapsedTime(&t, ctx->sort_ev[0], ctx->sort_ev[1]) == hipSuccess && ctx->sort_n > 0)
      ctx->sort_ms_per_photon = (double)t / (double)ctx->sort_n;
    (void)hipGetLastError();
    ctx->sort_timed = false;
  }
  if (!ok || events_this_cycle <= 0 || swarm->n < (1ll << 20)) return JB_COMPLETE;
  const double rate = ms / (double)events_this_cycle;
  ++ctx->cycles_since_sort;
  if (ctx->cycles_since_sort == 1 && ctx->rate_before_sort > 0.0) {
    // the first cycle behind a sort: did it pay?  If the rate came down by less than 1 % (timing
    // noise between cycles is ~0.3 %), what made the kernels slower was not the order of the swarm
    // -- wait twice as long before the next one
    if (rate > 0.99 * ctx->rate_before_sort) ctx->min_interval = ctx->min_interval < 256 ? 2 * ctx->min_interval : 256;
    else ctx->min_interval = 2;
    ctx->rate_before_sort = 0.0;
  }
  // The sort's scratch records (128 bytes per photon; a fresh allocation costs ~30 ms per GB, which must not
  // land in the cycle that first sorts: 400 ms on BASELINE configs[2]) are asked for as soon as the cycles
  // START to slow down (0.5 %: at least one cycle before the 1.5 % a sort needs) -- a run whose swarm keeps
  // its order never allocates them (12.8 GB at 1e8 photons); one that has no room learns so here.
  // Where memory is plentiful (the records would take less than a quarter of what is free: 12.8 GB of an
  // MI355X's 288) they are taken at the FIRST cycle the policy sees -- a host's warm-up -- so that no later
  // cycle of the run pays for the allocation at all.  And a cycle that meets the sort's own condition without
  // them (a slow-down from under 0.5 % to over 1.5 % within one cycle) takes the allocation, not the sort:
  // the sort follows a cycle later (below).
  bool scratch_now = !ctx->sort_scratch_tried && ctx->rate_ref > 0.0 && rate > 1.005 * ctx->rate_ref;
  if (!ctx->sort_scratch_tried && !scratch_now && ctx->rate_ref == 0.0) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
        (double)free_b > 4.0 * 8.0 * (double)kSortRecWords * (double)swarm->n)
      scratch_now = true;
    (void)hipGetLastError();
  }
  bool allocated_this_cycle = false;
  if (scratch_now) {
    allocated_this_cycle = true;
    ctx->sort_scratch_tried = true;
    const DevMesh &M0 = mesh->dm;
    const unsigned long long nbins0 = (unsigned long long)M0.nblocks * (unsigned long long)M0.ntot + 1ull;
    const unsigned long long ntiles0 = (nbins0 + kScanTile - 1) / kScanTile;
    if (swarm->n < (1ll << 32) && nbins0 < (1ull << 32)) {
      const size_t words = (size_t)kSortRecWords * (size_t)swarm->n + (size_t)((nbins0 + ntiles0 + swarm->n) / 2 + 16);
      if (ensure_scratch(ctx, words, /*slack=*/false) != JB_COMPLETE) {
        fprintf(stderr, "jaybenne_amd: no room for the scratch records of DefragParticles (%s): the swarm stays unsorted\n", g_err);
        (void)hipGetLastError();
        ctx->min_interval = 256;
      }
    }
  }
  if (ctx->rate_ref == 0.0 || rate < ctx->rate_ref) ctx->rate_ref = rate;
  const double excess_now = (rate - ctx->rate_ref) * (double)events_this_cycle;
  ctx->excess_ms += excess_now;
  // Sort when one more cycle in this order would cost more than a cycle has cost on average since
  // the last sort, the sort included: with p cycles behind the sort, loss L so far and a sort that
  // costs S, that is  e(p + 1) >= (S + L) / p  -- for a loss per cycle that keeps growing, the period
  // with the lowest (S + L) / p.  e(p + 1) is extrapolated from this cycle's loss, e(p) p / (p - 1)
  // (linear growth from zero in the first cycle).  Only on a slow-down that is no timing noise (1.5 %).
  ctx->last_rate = rate;
  const double sort_ms = ctx->sort_ms_per_photon * (double)swarm->n;
  const int p = ctx->cycles_since_sort;
  const double excess_next = p > 1 ? excess_now * (double)p / (double)(p - 1) : excess_now;
  if (p >= ctx->min_interval && rate > 1.015 * ctx->rate_ref &&
      excess_next * (double)p >= sort_ms + ctx->excess_ms) {
    if (!ctx->sort_scratch_tried) {   // (never asked for: this cycle takes the allocation, the next one the sort)
      ctx->sort_scratch_tried = true;
      const DevMesh &M1 = mesh->dm;
      const unsigned long long nbins1 = (unsigned long long)M1.nblocks * (unsigned long long)M1.ntot + 1ull;
      const unsigned long long ntiles1 = (nbins1 + kScanTile - 1) / kScanTile;
      if (swarm->n < (1ll << 32) && nbins1 < (1ull << 32)) {
        const size_t words = (size_t)kSortRecWords * (size_t)swarm->n + (size_t)((nbins1 + ntiles1 + swarm->n) / 2 + 16);
        if (ensure_scratch(ctx, words, /*slack=*/false) != JB_COMPLETE) {
          fprintf(stderr, "jaybenne_amd: no room for the scratch records of DefragParticles (%s): the swarm stays unsorted\n", g_err);
          (void)hipGetLastError();
          ctx->min_interval = 256;
        }
      }
      return JB_COMPLETE;
    }
    if (allocated_this_cycle && mode != JB_DEFRAG_SORT_NOW) return JB_COMPLETE;
    if (mode == JB_DEFRAG_DECIDE) {
      *sorted = 1;  // (this rank would sort: the host asks the others, then calls again with SORT_NOW)
      return JB_COMPLETE;
    }
    return defrag_now(ctx, mesh, swarm, rate, sorted);
  }
  return JB_COMPLETE;
}

extern "C" jb_status jb_pack_outgoing(jb_context *ctx, jb_mesh *mesh, const jb_swarm_view *swarm,
                                      int64_t first, int64_t last, int nranks, int64_t *records_dev,
                                      int64_t record_capacity, int64_t *counts_host) {
  JB_RANGE("Jaybenne::MeshSend");
  if (!ctx || !mesh || !counts_host) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_pack_outgoing");
  if (st != JB_COMPLETE) return st;
  if (first < 0 || last > swarm->n || first > last) return fail(JB_ERR_INVALID, "bad particle range");
  if (nranks < mesh->nranks_seen || nranks > kRankEnd - kRankBase)
    return fail(JB_ERR_INVALID, "nranks = %d does not cover the owners in the mesh view", nranks);
  unsigned long long *per_rank = ctx->counters_d + kRankBase;
  JB_HIP(hipMemsetAsync(per_rank, 0, sizeof(unsigned long long) * nranks, ctx->stream));
  const DevSwarm S = dev_swarm(swarm);
  if (last > first)
    hipLaunchKernelGGL(k_count_outgoing, dim3(grid_for(ctx, last - first)), dim3(kBlock), 0, ctx->stream,
                       mesh->dm, S, (long long)first, (long long)last, per_rank);
  JB_HIP(hipGetLastError());
  JB_HIP(hipMemcpyAsync(ctx->counters_h + kRankBase, per_rank, sizeof(unsigned long long) * nranks,
                        hipMemcpyDeviceToHost, ctx->stream));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  long long total = 0;
  std::vector<long long> firsts(nranks);
  for (int r = 0; r < nranks; ++r) {
    counts_host[r] = (int64_t)ctx->counters_h[kRankBase + r];
    firsts[r] = total;
    total += counts_host[r];
  }
  if (total == 0) return JB_COMPLETE;
  if (!records_dev || total > record_capacity)
    return fail(JB_ERR_CAPACITY, "hand-off buffer holds %lld records, %lld needed",
                (long long)record_capacity, total);
  st = ensure_scratch(ctx, (size_t)nranks);
  if (st != JB_COMPLETE) return st;
  JB_HIP(hipMemcpyAsync(ctx->scratch_d, firsts.data(), sizeof(long long) * nranks, hipMemcpyHostToDevice,
                        ctx->stream));
  JB_HIP(hipMemsetAsync(per_rank, 0, sizeof(unsigned long long) * nranks, ctx->stream));
  hipLaunchKernelGGL(k_pack_outgoing, dim3(grid_for(ctx, last - first)), dim3(kBlock), 0, ctx->stream,
                     mesh->dm, S, (long long)first, (long long)last, (const long long *)ctx->scratch_d,
                     per_rank, (long long *)records_dev);
  JB_HIP(hipGetLastError());
  JB_HIP(hipStreamSynchronize(ctx->stream));  // firsts lives on this stack frame
  return JB_COMPLETE;
}

extern "C" jb_status jb_unpack_incoming(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm,
                                        const int64_t *records_dev, int64_t nrecords) {
  JB_RANGE("Jaybenne::MeshReceive");
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_unpack_incoming");
  if (st != JB_COMPLETE) return st;
  if (nrecords < 0) return fail(JB_ERR_INVALID, "negative record count");
  if (nrecords == 0) return JB_COMPLETE;
  if (!records_dev) return fail(JB_ERR_INVALID, "null record buffer");
  if (swarm->n + nrecords > swarm->capacity)
    return fail(JB_ERR_CAPACITY, "swarm capacity %lld too small for %lld arrivals",
                (long long)swarm->capacity, (long long)nrecords);
  hipLaunchKernelGGL(k_unpack_incoming, dim3(grid_for(ctx, nrecords)), dim3(kBlock), 0, ctx->stream,
                     mesh->dm, dev_swarm(swarm), (long long)swarm->n, (const long long *)records_dev,
                     (long long)nrecords);
  JB_HIP(hipGetLastError());
  swarm->n += nrecords;
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
// jb_exchange: MeshResetCommunication -> MeshSend -> MeshReceive (jaybenne.cpp:26-61) as ONE call on the
// context's stream, the records never leaving the device.  What moves the bytes is a jb_exchange_transport
// (two collectives on device buffers): jb_transport_rccl below is the production one.
extern "C" jb_status jb_exchange(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm, int64_t first,
                                 int64_t last, int rank, int nranks, const jb_exchange_transport *tr,
                                 int64_t *send_dev, int64_t send_capacity, int64_t *recv_dev,
                                 int64_t recv_capacity, int64_t *nsent, int64_t *nreceived,
                                 int64_t *moved_anywhere) {
  JB_RANGE("Jaybenne::MeshSendReceive");
  if (!ctx || !mesh || !tr || !tr->all_gather_u64 || !tr->all_to_all_v || !nsent || !nreceived || !moved_anywhere)
    return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  jb_status st = check_swarm(swarm, "jb_exchange");
  if (st != JB_COMPLETE) return st;
  if (first < 0 || last > swarm->n || first > last) return fail(JB_ERR_INVALID, "bad particle range");
  if (rank < 0 || rank >= nranks) return fail(JB_ERR_INVALID, "rank outside [0, nranks)");
  if (nranks < mesh->nranks_seen || nranks + 3 > kRankEnd - kRankBase)
    return fail(JB_ERR_INVALID, "nranks = %d does not cover the owners in the mesh view", nranks);
  *nsent = 0; *nreceived = 0; *moved_anywhere = 0;
  const int row = nranks + 3;   // what a rank contributes to the all-gather: counts | send, receive, swarm room
  // 1. records per destination rank, counted on the device ...
  unsigned long long *per_rank = ctx->counters_d + kRankBase;
  JB_HIP(hipMemsetAsync(per_rank, 0, sizeof(unsigned long long) * nranks, ctx->stream));
  const DevSwarm S = dev_swarm(swarm);
  if (last > first)
    hipLaunchKernelGGL(k_count_outgoing, dim3(grid_for(ctx, last - first)), dim3(kBlock), 0, ctx->stream,
                       mesh->dm, S, (long long)first, (long long)last, per_rank);
  JB_HIP(hipGetLastError());
  // ... followed, in the same buffer, by what this rank has room for: the capacity decision below must
  // come out the same on EVERY rank (a rank that returned early would leave the others hanging in the
  // payload exchange), so every rank gets to see every rank's room
  ctx->xch_room[0] = (unsigned long long)(send_dev ? send_capacity : 0);
  ctx->xch_room[1] = (unsigned long long)(recv_dev ? recv_capacity : 0);
  ctx->xch_room[2] = (unsigned long long)(swarm->capacity - swarm->n);
  JB_HIP(hipMemcpyAsync(per_rank + nranks, ctx->xch_room, 3 * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  // 2. ... gathered from every rank straight from that buffer (no read-back in front of the collective):
  // the rank x rank matrix carries this rank's receive sizes AND the answer to "did anything move
  // anywhere" (the completion test of jaybenne.cpp:130-131 needs no collective of its own) ...
  // (the count matrix lives in a buffer of its own, taken on the FIRST call of a context -- before any rank has
  // entered a collective of the run's hot loop; a rank that cannot get its few KB fails there, and says so
  // in the gathered row of every later call: a rank-local failure must not leave the others in a collective)
  if (!ctx->xch_d || ctx->xch_ranks < nranks) {
    if (ctx->xch_d) (void)hipFree(ctx->xch_d);
    const size_t R = nranks > 64 ? (size_t)nranks : 64;
    if (hipMalloc(&ctx->xch_d, R * (R + 8) * sizeof(unsigned long long)) == hipSuccess) {
      ctx->xch_ranks = (int)R;
    } else {
      ctx->xch_d = nullptr;
      ctx->xch_ranks = 0;
      (void)hipGetLastError();
      return fail(JB_ERR_HIP, "jb_exchange: hipMalloc of the count matrix failed");
    }
  }
  unsigned long long *matrix_d = ctx->xch_d;
  if (tr->all_gather_u64(tr->handle, (const uint64_t *)per_rank, (uint64_t *)matrix_d, row, (void *)ctx->stream) != 0)
    return fail(JB_ERR_HIP, "jb_exchange: the transport's all-gather of the record counts failed");
  // 3. ... and read back ONCE per call: nranks (nranks + 3) words
  ctx->xch_matrix.resize((size_t)nranks * (size_t)row);
  JB_HIP(hipMemcpyAsync(ctx->xch_matrix.data(), matrix_d, sizeof(unsigned long long) * ctx->xch_matrix.size(),
                        hipMemcpyDeviceToHost, ctx->stream));
  JB_HIP(hipStreamSynchronize(ctx->stream));
  auto cnt = [&](int from, int to) { return (long long)ctx->xch_matrix[(size_t)from * row + to]; };
  long long total = 0, mine_out = 0, mine_in = 0;
  ctx->xch_tab.assign(4 * (size_t)nranks, 0);   // send counts | send offsets | recv counts | recv offsets (records)
  long long *sc = ctx->xch_tab.data(), *so = sc + nranks, *rc = so + nranks, *ro = rc + nranks;
  for (int s_ = 0; s_ < nranks; ++s_)
    for (int r = 0; r < nranks; ++r) total += cnt(s_, r);
  for (int r = 0; r < nranks; ++r) {
    sc[r] = cnt(rank, r);
    rc[r] = cnt(r, rank);
    so[r] = mine_out; ro[r] = mine_in;
    mine_out += sc[r]; mine_in += rc[r];
  }
  *moved_anywhere = total;
  *nsent = mine_out; *nreceived = mine_in;
  // (every rank looks at every rank's diagonal: the same verdict everywhere)
  for (int q = 0; q < nranks; ++q)
    if (cnt(q, q) != 0) return fail(JB_ERR_INVALID, "jb_exchange: rank %d hands particles to itself", q);
  if (total == 0) return JB_COMPLETE;
  // the same verdict on every rank: does EVERY rank have room for what it sends and takes in?
  for (int q = 0; q < nranks; ++q) {
    long long out_q = 0, in_q = 0;
    for (int r = 0; r < nranks; ++r) { out_q += cnt(q, r); in_q += cnt(r, q); }
    const long long send_room = (long long)ctx->xch_matrix[(size_t)q * row + nranks];
    const long long recv_room = (long long)ctx->xch_matrix[(size_t)q * row + nranks + 1];
    const long long swarm_room = (long long)ctx->xch_matrix[(size_t)q * row + nranks + 2];
    if (out_q > send_room)
      return fail(JB_ERR_CAPACITY, "jb_exchange: rank %d's send buffer holds %lld records, %lld needed", q, send_room, out_q);
    if (in_q > recv_room)
      return fail(JB_ERR_CAPACITY, "jb_exchange: rank %d's receive buffer holds %lld records, %lld needed", q, recv_room, in_q);
    if (in_q > swarm_room)
      return fail(JB_ERR_CAPACITY, "jb_exchange: rank %d's swarm has room for %lld arrivals, %lld needed", q, swarm_room, in_q);
  }
  // 4. pack (the packed slots become holes), 5. the payload, 6. unpack behind it on the same stream
  if (mine_out > 0) {
    long long *firsts_d = (long long *)(matrix_d + (size_t)nranks * row);
    JB_HIP(hipMemcpyAsync(firsts_d, so, sizeof(long long) * nranks, hipMemcpyHostToDevice, ctx->stream));
    JB_HIP(hipMemsetAsync(per_rank, 0, sizeof(unsigned long long) * nranks, ctx->stream));
    hipLaunchKernelGGL(k_pack_outgoing, dim3(grid_for(ctx, last - first)), dim3(kBlock), 0, ctx->stream,
                       mesh->dm, S, (long long)first, (long long)last, (const long long *)firsts_d,
                       per_rank, (long long *)send_dev);
    JB_HIP(hipGetLastError());
  }
  if (tr->all_to_all_v(tr->handle, send_dev, (const int64_t *)sc, (const int64_t *)so, recv_dev,
                       (const int64_t *)rc, (const int64_t *)ro, kRecWords, (void *)ctx->stream) != 0)
    return fail(JB_ERR_HIP, "jb_exchange: the transport's record exchange failed");
  if (mine_in > 0) {
    hipLaunchKernelGGL(k_unpack_incoming, dim3(grid_for(ctx, mine_in)), dim3(kBlock), 0, ctx->stream,
                       mesh->dm, dev_swarm(swarm), (long long)swarm->n, (const long long *)recv_dev,
                       (long long)mine_in);
    JB_HIP(hipGetLastError());
    swarm->n += mine_in;
  }
  return JB_COMPLETE;
}

// ---- the RCCL transport: all-gather + grouped send / recv on the caller's communicator.  RCCL is
// loaded when the first transport is made (dlopen: the library itself does not link against it, so a
// host without RCCL -- or with its own copy already in the process, as PyTorch has -- loads fine).
namespace {
struct RcclApi {
  void *lib = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
constexpr int kNcclInt64 = 4, kNcclUint64 = 5;   // ncclDataType_t (rccl.h)
struct RcclHandle { void *comm; int rank, nranks; };

bool load_rccl() {
  if (g_rccl.lib) return true;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) return false;
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
  g_rccl.Send = (decltype(g_rccl.Send))dlsym(g_rccl.lib, "ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(g_rccl.lib, "ncclRecv");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.lib, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.lib, "ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
  return g_rccl.AllGather && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
}

int rccl_all_gather(void *h, const uint64_t *in_dev, uint64_t *out_dev, int count, void *stream) {
  const RcclHandle *H = (const RcclHandle *)h;
  return g_rccl.AllGather(in_dev, out_dev, (size_t)count, kNcclUint64, H->comm, (hipStream_t)stream);
}
// every rank pair uses its own xGMI link: one grouped send / recv per peer, no ring through the node
int rccl_all_to_all_v(void *h, const int64_t *send_dev, const int64_t *sc, const int64_t *so, int64_t *recv_dev,
                      const int64_t *rc, const int64_t *ro, int words, void *stream) {
  const RcclHandle *H = (const RcclHandle *)h;
  int err = g_rccl.GroupStart();
  for (int r = 0; r < H->nranks && err == 0; ++r) {
    if (sc[r] > 0) err = g_rccl.Send(send_dev + so[r] * words, (size_t)(sc[r] * words), kNcclInt64, r, H->comm, (hipStream_t)stream);
    if (err == 0 && rc[r] > 0)
      err = g_rccl.Recv(recv_dev + ro[r] * words, (size_t)(rc[r] * words), kNcclInt64, r, H->comm, (hipStream_t)stream);
  }
  const int end = g_rccl.GroupEnd();
  return err != 0 ? err : end;
}
}  // namespace

extern "C" jb_status jb_transport_rccl(void *nccl_comm, int rank, int nranks, jb_exchange_transport *out) {
  if (!nccl_comm || !out) return fail(JB_ERR_INVALID, "null argument");
  if (rank < 0 || rank >= nranks) return fail(JB_ERR_INVALID, "rank outside [0, nranks)");
  if (!load_rccl()) return fail(JB_ERR_INVALID, "jb_transport_rccl: librccl.so could not be loaded (%s)", dlerror());
  RcclHandle *H = new (std::nothrow) RcclHandle{nccl_comm, rank, nranks};
  if (!H) return fail(JB_ERR_INVALID, "out of memory");
  out->handle = H;
  out->all_gather_u64 = rccl_all_gather;
  out->all_to_all_v = rccl_all_to_all_v;
  return JB_COMPLETE;
}

extern "C" jb_status jb_transport_release(jb_exchange_transport *tr) {
  if (tr && tr->all_gather_u64 == rccl_all_gather) delete (RcclHandle *)tr->handle;
  if (tr) { tr->handle = nullptr; tr->all_gather_u64 = nullptr; tr->all_to_all_v = nullptr; }
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
static double *const *field_table(const DevMesh &M, int field) {
  switch (field) {
    case JB_FIELD_RHO: return M.rho;
    case JB_FIELD_SIE: return M.sie;
    case JB_FIELD_U: return M.u;
    case JB_FIELD_FLECK: return M.fleck;
    case JB_FIELD_TALLY: return M.tally;
    case JB_FIELD_EDELTA: return M.edelta;
    default: return nullptr;
  }
}

extern "C" jb_status jb_gather_cells(jb_context *ctx, jb_mesh *mesh, int field, int64_t n,
                                     const int32_t *blk_dev, const int32_t *cell_dev,
                                     double *out_dev) {
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  double *const *F = field_table(mesh->dm, field);
  if (!F) return fail(JB_ERR_INVALID, "jb_gather_cells: unknown field %d", field);
  if (n < 0) return fail(JB_ERR_INVALID, "negative cell count");
  if (n == 0) return JB_COMPLETE;
  if (!blk_dev || !cell_dev || !out_dev) return fail(JB_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(k_gather_cells, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, F,
                     (long long)n, blk_dev, cell_dev, out_dev);
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

extern "C" jb_status jb_fill_cells(jb_context *ctx, jb_mesh *mesh, int field, int64_t n,
                                   int nsamples, const int32_t *dst_blk_dev,
                                   const int32_t *dst_cell_dev, const int32_t *src_blk_dev,
                                   const int32_t *src_cell_dev, const double *remote_dev) {
  if (!ctx || !mesh) return fail(JB_ERR_INVALID, "null argument");
  JB_HIP(hipSetDevice(ctx->device));
  double *const *F = field_table(mesh->dm, field);
  if (!F) return fail(JB_ERR_INVALID, "jb_fill_cells: unknown field %d", field);
  if (n < 0) return fail(JB_ERR_INVALID, "negative cell count");
  if (n == 0) return JB_COMPLETE;
  if (!dst_blk_dev || !dst_cell_dev || !src_blk_dev || !src_cell_dev)
    return fail(JB_ERR_INVALID, "null argument");
#define JB_FILL(NS)                                                                              \
  hipLaunchKernelGGL(k_fill_cells<NS>, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, F,   \
                     (long long)n, dst_blk_dev, dst_cell_dev, src_blk_dev, src_cell_dev, remote_dev)
  switch (nsamples) {
    case 1: JB_FILL(1); break;
    case 2: JB_FILL(2); break;
    case 4: JB_FILL(4); break;
    case 8: JB_FILL(8); break;
    default: return fail(JB_ERR_INVALID, "jb_fill_cells: nsamples must be 1, 2, 4 or 8");
  }
#undef JB_FILL
  JB_HIP(hipGetLastError());
  return JB_COMPLETE;
}

// ------------------------------------------------------------------------------------------------
// RadiationStep for a mesh held entirely by this rank: the task list of jaybenne.cpp:104-138 with
// the iterate-sublist collapsed to one launch (every block crossing is resolved in flight).
extern "C" jb_status jb_radiation_step(jb_context *ctx, jb_mesh *mesh, jb_swarm_view *swarm,
                                       double t_start, double dt, uint64_t *next_id, uint32_t *cycle,
                                       int32_t *prefix_dev) {
  if (!ctx || !mesh || !swarm || !next_id || !cycle) return fail(JB_ERR_INVALID, "null argument");
  const DevMesh &M = mesh->dm;
  if (M.nblocks != M.nblocks_total || mesh->nranks_seen != 1)
    return fail(JB_ERR_INVALID, "jb_radiation_step needs the whole mesh on one rank");
  if (*cycle >= (1u << 19) - 1u)   // (SourceEpoch: emission keys k and in-cycle thermal keys (1 << 19) | k must not meet)
    return fail(JB_ERR_INVALID, "jb_radiation_step: cycle counter %u at the limit of the source epochs (2^19 - 1)", *cycle);
  *cycle += 1;   // (keys the per-cell rounding streams of this cycle's emission source: SourceEpoch)
  JB_RANGE("Jaybenne::Timestep");               // jaybenne.cpp:87 ... :145
  jb_status st = jb_update_derived_transport_fields(ctx, mesh, dt);
  if (st != JB_COMPLETE) return st;
  if (ctx->params.do_emission) {
    if (!prefix_dev) return fail(JB_ERR_INVALID, "emission source needs the prefix workspace");
    std::vector<int32_t> nper(M.nblocks);
    st = jb_source_photons_count(ctx, mesh, JB_SOURCE_EMISSION, dt, M.nblocks, *cycle, nper.data(),
                                 prefix_dev);
    if (st != JB_COMPLETE) return st;
    std::vector<int64_t> slot(M.nblocks);
    std::vector<uint64_t> ids(M.nblocks);
    int64_t tot = 0;
    for (int b = 0; b < M.nblocks; ++b) {
      slot[b] = swarm->n + tot;
      ids[b] = *next_id + (uint64_t)tot;
      tot += nper[b];
    }
    if (swarm->n + tot > swarm->capacity)
      return fail(JB_ERR_CAPACITY, "swarm capacity %lld too small for %lld new particles",
                  (long long)swarm->capacity, (long long)tot);
    st = jb_source_photons_fill(ctx, mesh, swarm, JB_SOURCE_EMISSION, t_start, dt, nper.data(),
                                prefix_dev, slot.data(), ids.data());
    if (st != JB_COMPLETE) return st;
    swarm->n += tot;
    *next_id += (uint64_t)tot;
  }
  st = jb_zero_energy_tally(ctx, mesh);
  if (st != JB_COMPLETE) return st;
  jb_transport_stats before;
  st = jb_get_transport_stats(ctx, &before, 0);
  if (st != JB_COMPLETE) return st;
  {
    JB_RANGE("Jaybenne::TransportLoop");        // jaybenne.cpp:115 ... :127 (one pass: every crossing is resolved in flight)
    st = transport_impl(ctx, mesh, swarm, t_start, dt, 0, swarm->n, 1, ctx->params.use_ddmc != 0);
  }
  if (st != JB_COMPLETE) return st;
  jb_transport_stats after;
  st = jb_get_transport_stats(ctx, &after, 0);
  if (st != JB_COMPLETE) return st;
  if (after.n_outgoing != before.n_outgoing)
    return fail(JB_ERR_INVALID, "particles left for another rank in a single-rank step");
  if (after.n_absorbed != before.n_absorbed || after.n_escaped != before.n_escaped) {
    st = jb_remove_marked_particles(ctx, swarm);
    if (st != JB_COMPLETE) return st;
  }
  return jb_update_fluid(ctx, mesh);
}

// ------------------------------------------------------------------------------------------------
// debug entry points
__global__ void k_dbg_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1, uint32_t *out) {
  const PhiloxBlock b = philox4x32_10(c0, c1, c2, c3, k0, k1);
  out[0] = b.w0; out[1] = b.w1; out[2] = b.w2; out[3] = b.w3;
}
__global__ void k_dbg_rocrand(unsigned long long seed, unsigned long long subseq, uint32_t *out) {
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, subseq, 0ull, &st);
  const uint4 a = rocrand4(&st);
  const uint4 b = rocrand4(&st);
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
  out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
}
__global__ void k_dbg_draw(unsigned long long state, int n, double *out, unsigned long long *fin) {
  LcgRng rng(state);
  for (int i = 0; i < n; ++i) out[i] = rng.drand();
  *fin = rng.s;
}
__global__ void k_dbg_seed(uint32_t seed, uint32_t domain, unsigned long long id,
                           unsigned long long *out) {
  *out = rng_seed_state(seed, domain, id);
}
__global__ void k_dbg_stream_start(uint32_t seed, unsigned long long id, unsigned long long *out) {
  *out = rng_stream_start(seed, id);
}
__global__ void k_dbg_math(int which, const double *x, int n, double *out) {
  load_math_tables<true, true, true, true, true>();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double s, c;
    switch (which) {
    case 0: out[i] = m_log(x[i]); break;
    case 1: m_sincos(x[i], s, c); out[i] = s; break;
    case 2: m_sincos(x[i], s, c); out[i] = c; break;
    case 3: out[i] = m_acos(x[i]); break;
    case 4: out[i] = sqrt(x[i]); break;
    case 5: out[i] = 1.0 / x[i]; break;
    case 6: out[i] = m_sqrt(x[i]); break;
    case 7: out[i] = m_div(x[i], x[(i + 1) % n]); break;
    case 8: out[i] = m_div_r(x[i], 2.99792458e10, m_rcp_refined(2.99792458e10)); break;
    case 9: m_sincos2pi(x[i], s, c); out[i] = s; break;
    case 10: m_sincos2pi(x[i], s, c); out[i] = c; break;
    case 11: out[i] = m_one_minus_exp_neg(x[i]); break;
    case 12: out[i] = x[i] * m_rcp_once(x[(i + 1) % n]); break;   // lean quotient
    case 13: out[i] = m_log_lean(x[i]); break;
    case 15: out[i] = m_log_lean<false, true>(x[i]); break;
    default: out[i] = m_sqrt_lean(x[i]); break;
    }
  }
}
__global__ void k_dbg_model(DevParams P, int which, const double *x, int n, double *out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double rho = x[3 * i], temp = x[3 * i + 1], nu = x[3 * i + 2];
    out[i] = which == 0 ? opac_absorption(P, rho, temp, nu)
             : which == 1 ? opac_emissivity(P, rho, temp) : opac_scattering(P, rho, temp, nu);
  }
}
__global__ void k_dbg_step(int which, jb_debug_step *d, const double *tape, int ntape, int *ndraws) {
  load_math_tables();
  TapeRng rng(tape, ntape);
  Step s;
  s.t_start = d->t_start; s.dt = d->dt; s.ff = d->ff; s.aa = d->aa; s.ss = d->ss; s.vv = d->vv; s.rvv = m_rcp_refined(d->vv);
  s.dx_push = d->dx_push;
  s.ffaa = s.ff * s.aa;
  s.sig = s.aa + s.ss;
  s.xl = d->xl; s.yl = d->yl; s.zl = d->zl; s.xu = d->xu; s.yu = d->yu; s.zu = d->zu;
  s.Px_l = d->Px_l; s.Py_l = d->Py_l; s.Pz_l = d->Pz_l; s.Px_u = d->Px_u; s.Py_u = d->Py_u; s.Pz_u = d->Pz_u;
  s.t = d->t; s.x = d->x; s.y = d->y; s.z = d->z; s.vx = d->vx; s.vy = d->vy; s.vz = d->vz;
  s.ip = d->ip; s.jp = d->jp; s.kp = d->kp;
  s.is_absorbed = d->is_absorbed != 0; s.is_scattered = d->is_scattered != 0;
  s.is_rejected = d->is_rejected != 0;
  const int nd = d->three_d ? 3 : (d->multi_d ? 2 : 1);
#define DISPATCH(fn)                                   \
  if (nd == 1) fn<1>(s, rng);                          \
  else if (nd == 2) fn<2>(s, rng);                     \
  else fn<3>(s, rng);
  if (which == 0) { DISPATCH(ptcl_transport_step) }
  else if (which == 1) { DISPATCH(ptcl_ddmc_step) }
  else { DISPATCH(ptcl_ddmc_albedo) }
#undef DISPATCH
  d->t = s.t; d->x = s.x; d->y = s.y; d->z = s.z; d->vx = s.vx; d->vy = s.vy; d->vz = s.vz;
  d->ip = s.ip; d->jp = s.jp; d->kp = s.kp;
  d->is_absorbed = s.is_absorbed; d->is_scattered = s.is_scattered; d->is_rejected = s.is_rejected;
  *ndraws = (int)rng.ctr;
}
__global__ void k_dbg_sample(int which, const double *a, const int *iv, const double *tape, int ntape,
                             double *out, int *iout, int *ndraws) {
  load_math_tables();
  TapeRng rng(tape, ntape);
  if (which == 0) {
    scatter(rng, a[0], out[0], out[1], out[2]);
  } else if (which == 1) {
    sample_face_iso_dir(a[0], rng, out[0], out[1], out[2]);
  } else if (which == 2) {
    out[0] = sample_planck_energy(rng, a[0], a[1]);
  } else if (which == 3) {
    int i = iv[1];
    double x = a[3];
    sample_face_2d(iv[0], a[0], a[1], a[2], rng, i, x);
    iout[0] = i; out[0] = x;
  } else {
    int i1 = iv[2], i2 = iv[3];
    double x1 = a[6], x2 = a[7];
    sample_face_3d(iv[0], iv[1], a[0], a[1], a[2], a[3], a[4], a[5], rng, i1, i2, x1, x2);
    iout[0] = i1; iout[1] = i2; out[0] = x1; out[1] = x2;
  }
  *ndraws = (int)rng.ctr;
}

// small helper: run a debug kernel with device copies of host buffers
struct DbgBuf {
  void *d = nullptr;
  size_t bytes = 0;
  ~DbgBuf() { if (d) (void)hipFree(d); }
  hipError_t put(const void *h, size_t n) {
    bytes = n ? n : 8;
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) return e;
    if (h && n) e = hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    return e;
  }
  hipError_t get(void *h, size_t n) { return hipMemcpy(h, d, n, hipMemcpyDeviceToHost); }
};

extern "C" jb_status jb_debug_philox(jb_context *ctx, const uint32_t ctr[4], const uint32_t key[2],
                                     uint32_t out[4]) {
  if (!ctx) return fail(JB_ERR_INVALID, "null context");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf o;
  JB_HIP(o.put(nullptr, 16));
  hipLaunchKernelGGL(k_dbg_philox, dim3(1), dim3(1), 0, ctx->stream, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], (uint32_t *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out, 16));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_rocrand_philox(jb_context *ctx, uint64_t seed, uint64_t subsequence,
                                             uint32_t out[8]) {
  if (!ctx) return fail(JB_ERR_INVALID, "null context");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf o;
  JB_HIP(o.put(nullptr, 32));
  hipLaunchKernelGGL(k_dbg_rocrand, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long)seed,
                     (unsigned long long)subsequence, (uint32_t *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out, 32));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_seed_state(jb_context *ctx, uint32_t seed, uint32_t domain,
                                         uint64_t id, uint64_t *state) {
  if (!ctx || !state) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf o;
  JB_HIP(o.put(nullptr, 8));
  hipLaunchKernelGGL(k_dbg_seed, dim3(1), dim3(1), 0, ctx->stream, seed, domain, (unsigned long long)id, (unsigned long long *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(state, 8));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_stream_start(jb_context *ctx, uint32_t seed, uint64_t id,
                                           uint64_t *state) {
  if (!ctx || !state) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf o;
  JB_HIP(o.put(nullptr, 8));
  hipLaunchKernelGGL(k_dbg_stream_start, dim3(1), dim3(1), 0, ctx->stream, seed, (unsigned long long)id, (unsigned long long *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(state, 8));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_draw_stream(jb_context *ctx, uint64_t state, int n, double *out_host,
                                          uint64_t *final_state) {
  if (!ctx || n < 0 || !final_state) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf o, f;
  JB_HIP(o.put(nullptr, sizeof(double) * n));
  JB_HIP(f.put(nullptr, 8));
  hipLaunchKernelGGL(k_dbg_draw, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long)state, n, (double *)o.d, (unsigned long long *)f.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out_host, sizeof(double) * n));
  JB_HIP(f.get(final_state, 8));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_model_coefficients(jb_context *ctx, double out[4]) {
  if (!ctx || !out) return fail(JB_ERR_INVALID, "bad argument");
  out[0] = ctx->dp.ep_A; out[1] = ctx->dp.ep_B; out[2] = ctx->dp.ep_E; out[3] = ctx->dp.kappa_s;
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_model_eval(jb_context *ctx, int which, const double *x_host, int n,
                                         double *out_host) {
  if (!ctx || n < 0 || which < 0 || which > 2) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf x, o;
  JB_HIP(x.put(x_host, sizeof(double) * 3 * n));
  JB_HIP(o.put(nullptr, sizeof(double) * n));
  hipLaunchKernelGGL(k_dbg_model, dim3(64), dim3(256), 0, ctx->stream, ctx->dp, which, (const double *)x.d, n, (double *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out_host, sizeof(double) * n));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_math(jb_context *ctx, int which, const double *x_host, int n,
                                   double *out_host) {
  if (!ctx || n < 0) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf x, o;
  JB_HIP(x.put(x_host, sizeof(double) * n));
  JB_HIP(o.put(nullptr, sizeof(double) * n));
  hipLaunchKernelGGL(k_dbg_math, dim3(64), dim3(256), 0, ctx->stream, which, (const double *)x.d, n, (double *)o.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out_host, sizeof(double) * n));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_step_call(jb_context *ctx, int which, jb_debug_step *st,
                                        const double *tape, int ntape, int *ndraws) {
  if (!ctx || !st || !tape || ntape < 1 || !ndraws) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf s, t, n;
  JB_HIP(s.put(st, sizeof(*st)));
  JB_HIP(t.put(tape, sizeof(double) * ntape));
  JB_HIP(n.put(nullptr, sizeof(int)));
  hipLaunchKernelGGL(k_dbg_step, dim3(1), dim3(1), 0, ctx->stream, which, (jb_debug_step *)s.d, (const double *)t.d, ntape, (int *)n.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(s.get(st, sizeof(*st)));
  JB_HIP(n.get(ndraws, sizeof(int)));
  return JB_COMPLETE;
}
extern "C" jb_status jb_debug_sample_call(jb_context *ctx, int which, const double *a,
                                          const int32_t *i, const double *tape, int ntape,
                                          double out[4], int32_t iout[2], int *ndraws) {
  if (!ctx || !a || !i || !tape || ntape < 1 || !ndraws) return fail(JB_ERR_INVALID, "bad argument");
  JB_HIP(hipSetDevice(ctx->device));
  DbgBuf da, di, t, o, io, n;
  JB_HIP(da.put(a, sizeof(double) * 8));
  JB_HIP(di.put(i, sizeof(int) * 4));
  JB_HIP(t.put(tape, sizeof(double) * ntape));
  JB_HIP(o.put(nullptr, sizeof(double) * 4));
  JB_HIP(io.put(nullptr, sizeof(int) * 2));
  JB_HIP(n.put(nullptr, sizeof(int)));
  JB_HIP(hipMemset(o.d, 0, sizeof(double) * 4));
  JB_HIP(hipMemset(io.d, 0, sizeof(int) * 2));
  hipLaunchKernelGGL(k_dbg_sample, dim3(1), dim3(1), 0, ctx->stream, which, (const double *)da.d,
                     (const int *)di.d, (const double *)t.d, ntape, (double *)o.d, (int *)io.d, (int *)n.d);
  JB_HIP(hipStreamSynchronize(ctx->stream));
  JB_HIP(o.get(out, sizeof(double) * 4));
  JB_HIP(io.get(iout, sizeof(int) * 2));
  JB_HIP(n.get(ndraws, sizeof(int)));
  return JB_COMPLETE;
}
