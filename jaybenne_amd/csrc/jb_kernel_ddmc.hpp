// jb_kernel_ddmc.hpp -- TransportPhotons_DDMC for meshes whose every cell takes the DDMC branch
// (transport_ddmc.cpp:135: dx_push (sigma_s + sigma_a) > tau_ddmc everywhere; gray opacities):
// the reference's stepdiff_ddmc / stepdiff_smr_ddmc / inf_stiff decks and BASELINE configs[2].
//
// Why a kernel of its own.  A DDMC history is short (~34 steps in 3-D) and each step waits for
// one random 64-byte cell record: the loop is bound by memory latency, so what counts is how many
// waves a SIMD can keep in flight, i.e. how few registers a lane needs inside the loop.  In an
// all-DDMC mesh almost nothing of a particle is live between two steps:
//   * after a leak the position is "eps_ddmc dx beyond the face it left through, cell centre
//     across" (transport_utils.hpp:209-263) -- a function of the new cell and the leak channel;
//     the albedo test of the next step (transport_utils.hpp:288-389) compares 1e8 eps against
//     2.5e6 eps and cannot fire, and then moves the particle to the cell centre (:392-396);
//   * the direction drawn at the leak is read by nobody until the particle leaves DDMC cells, is
//     absorbed or crosses a block (deferred: channel + its two uniforms, as in k_transport).
// So a lane in the event loop carries its cell -- since round 5 as ONE 32-bit record number, block and
// cell in one, which is what the step's gather is addressed with and what a leak moves by +-1, +-ni,
// +-ni nj --, time, random-stream state and the pending leak ("virtual" state: 10 registers); block, cell
// indices, position and direction exist only in the service phase.  Whether a leak left the block is read
// off the record the next pass gathers anyway (ghost codes: kStepGhostTable below).
// It loads a particle and evaluates the albedo step's six face tests on its real position: if
// none holds (a particle lies within 5.5e-9 dx of a face of its cell once in ~1e8) the step
// starts from the cell centre like every other and the lane goes straight to the loop; otherwise
// the particle is handed over: its slot goes on a list that k_hybrid -- which has the general
// step functions, albedo step included -- works through right after this kernel (until round 3
// that step was taken here: code that ran for a handful of particles per launch and set the
// kernel's register count, 137 instead of 116).  A direction that no step
// has changed is not carried either: it stays where it was loaded from (S.vx, vy, vz).  Position
// and direction are rebuilt from the virtual state where a consumer appears (block crossing,
// census resampling, absorption, write-back).  Same draws, same
// arithmetic, same bits as k_transport<NDIM, true, TALLY, 1 or 2> -- tests/test_gpu_parity.py
// holds both to the oracle.
#pragma once

#include "jb_device.hpp"

namespace jb {

#ifndef JB_DDMC_ALL_WAVES_PER_SIMD
#define JB_DDMC_ALL_WAVES_PER_SIMD 4   // (128 registers; 39 KB of LDS per workgroup without the LDS tally)
#endif
#ifndef JB_DDMC_ALL_BUDGET   // idle lane-passes that buy a service phase
#define JB_DDMC_ALL_BUDGET 256
#endif
#ifndef JB_DDMC_ALL_CHUNK
#define JB_DDMC_ALL_CHUNK 128
#endif
#ifndef JB_DDMC_ALL_WINDOW   // 1: GATHER == 2 keeps the next 64 slots of the swarm requested ahead (see PF below)
#define JB_DDMC_ALL_WINDOW 1
#endif

// (DS_ABS / DS_CENSUS: how a lane left the event loop -- absorbed / at census without an event in its last
// step: turned into DS_DONE, with the bookkeeping that goes with it, right behind the loop)
enum { DS_IDLE = 0, DS_VIRT = 1, DS_PARK = 2, DS_DONE = 3, DS_RELOC = 4, DS_ABS = 5, DS_CENSUS = 6 };
// Ghost cells of the STEP records (DevMesh::ddmc_step; written once per mesh by k_lam_ghost_codes, the
// per-cycle k_ddmc_pack only writes interior cells): the last double -- in a real cell the positive
// reciprocal of c (f sigma_a + leak_tot + DBL_MIN) -- is negative there, its high word
// 0x80000000 | flags, and with kStepGhostTable its low word is the RECORD NUMBER of the cell a particle
// that leaks into this ghost cell really is in: the first / last interior cell along the axis of the
// same-size resident neighbour or of the block across a periodic boundary, or the cell it came from at
// a reflecting wall (in 1-D with kStepGhostMirror: the direction travels there and is mirrored).
// Without the flag (level changes, destinations that are not resident, outflow): the general relocation.
constexpr int kStepGhostTable = 1;
// ... and kStepGhostMirror (1-D only): the face is a reflecting wall -- the particle is back in the cell it
// came from and the direction of its pending leak is mirrored (boundaries.hpp:46-82: v_x = -v_x; in 1-D a
// leak's direction travels with the particle, transport_ddmc.cpp:206) -- so that the 1-D decks' wall leaks
// (one in a hundred steps on BASELINE configs[2] as shipped: a lane in nearly every service phase) stay
// in the event loop like every other same-size crossing
constexpr int kStepGhostMirror = 2;
constexpr int kPdZero = 0x40000000;  // pd: the zero-velocity flag of a multi-D leak across a block face

// The swarm is a stream: every particle attribute is read once and written once per launch.  Where the
// event loop gathers its records through L2 (GATHER 0 / 1 / 3) these accesses carry the non-temporal
// hint, so that they do not push the cell records -- which ARE re-read, by every step -- out of the XCD's
// L2: BASELINE configs[2] in 3-D 28.1 -> 26.9 ms (A/B on one box; 5 % of the record requests miss L2 and
// nearly every 64-lane pass waits for one of them).  With the records in LDS (GATHER 2) there is
// nothing to protect and the hint costs 11 %.  JB_DDMC_NT=0 switches it off.
#ifndef JB_DDMC_NT
#define JB_DDMC_NT 1
#endif
template <bool NT, class T>
__device__ __forceinline__ T swarm_ld(const T *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, class T, class V>
__device__ __forceinline__ void swarm_st(T *p, V v) {
#ifdef JB_DDMC_EXP_NOSTORE   // (timing experiment, tools/dev/ab1.sh: what the write-back costs; results are wrong)
  if (v == (V)-12345) *p = (T)v;
  return;
#endif
  if constexpr (NT) __builtin_nontemporal_store((T)v, p);
  else *p = (T)v;
}

// the kernel's argument list as the kernel-argument segment holds it (natural alignment, in order)
struct DdmcAllArgs {
  MeshConstPtr Mp;
  DevParams P;
  DevSwarm S;
  double t_start, dt;
  long long first, last;
  unsigned long long *counters;
  const int *not_all_ddmc;
  unsigned *park_list;              // slots handed over to k_hybrid ...
  unsigned long long *park_count;   // ... and how many
};

// quad_bcast_add<K>(v, add): v of lane K of the caller's quad (lanes 4q .. 4q+3), plus the caller's add
template <int K>
__device__ __forceinline__ unsigned quad_bcast_add(unsigned v, unsigned add) {
  constexpr int ctrl = K * 0x55;  // quad_perm:[K, K, K, K]
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xf, 0xf, true) + add;
}

// COOP (the cell records of all resident blocks span < 4 GiB: jb_transport_photons_ddmc): the four
// lanes of a quad fetch the four 16-byte pieces of ONE record with one LDS-direct load, instruction
// k serving the record of quad-lane k -- the vector L1 then looks up one line per record instead of
// four (one per 16-byte piece of every lane), which is what bounds the loop: 4 x 64 lookups per wave
// pass x 12 waves per CU = the ~3000 cycles a round of passes takes.
// GATHER: how the event loop fetches the step record of a lane's cell -- 0 four 16-byte loads per
// lane; 1 = COOP above; 2 the records of ALL resident blocks' cells copied into LDS at kernel
// start (<= kLdsRecCells cells, 64 B each: the reference's 1-D decks, BASELINE configs[2] as
// shipped), after which the loop issues no vector-memory instruction at all.  Measured on that
// deck the vector L1 was busy 95 % of the kernel's time serving 4.8e9 record look-ups.
// GATHER 4 (round 6; the gather of k_ddmc_q, jb_kernel_ddmc_q.hpp -- the default on meshes with <= 64 resident
// blocks -- and of this kernel where that one does not apply): the loop gathers a 4-byte CELL CODE per step
// (DevMesh::ddmc_code) -- the class of the cell's step record, whose <= kMaxClasses distinct values
// (k_ddmc_pack numbers them every cycle: gray decks have one per level x face-neighbour pattern) sit in
// LDS, or for a ghost cell where a particle that leaked there really is -- instead of the 64-byte record:
// the table of BASELINE configs[2] in 3-D shrinks from 161 MB to 10 MB (what one XCD's photons touch:
// from 1.8 MB to 110 KB -- L1 / L2 resident), the code for the NEXT pass is requested at the end of a pass,
// and the workgroup's 16 KB landing buffer of the quad gather is not needed.
constexpr int kLdsRecCells = 256;
// (NDIM < 3) pd of a leak through the "z+" arm of transport_utils.hpp:254-263 on a mesh without a z axis
// (xim rounds onto leak_tot: one event in ~1e16): the channel is z+, the cell does not change (:256,
// kp += three_d)
constexpr int kPdStay = 0x20000000;
template <int NDIM, bool TALLY, int GATHER>
#ifndef JB_DDMC_ALL_ATTR
#define JB_DDMC_ALL_ATTR
#endif
__global__ void __launch_bounds__(kBlock, JB_DDMC_ALL_WAVES_PER_SIMD) JB_DDMC_ALL_ATTR
    k_ddmc_all(const DevMesh *__restrict__, DevParams, DevSwarm, double, double, long long, long long,
               unsigned long long *, const int *, unsigned *, unsigned long long *) {
  // The arguments are read where they are used, straight from the kernel-argument segment (scalar
  // loads from constant memory), not through the parameters: those are fetched at kernel entry and
  // then live in ~80 scalar registers across both loops -- the registers the loop's constants
  // (logarithm coefficients, stream multipliers) then do not get, so that they sit in vector
  // registers instead.  The mesh view (~120 dwords) is behind a pointer for the same reason.
  const DdmcAllArgs &A = *(const DdmcAllArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  if (*A.not_all_ddmc != 0) return;  // (uniform) some cell takes IMC steps: k_hybrid runs instead
  const DevMesh &M = *(const DevMesh *)A.Mp;
  const DevParams &P = A.P;
  const DevSwarm &S = A.S;
  const double t_start = A.t_start, dt = A.dt;
  const long long first = A.first, last = A.last;
  unsigned long long *const counters = g1(A.counters);
  constexpr bool COOP = GATHER == 1 || GATHER == 3;   // (3: COOP with 64-bit addresses, records >= 4 GiB)
  constexpr bool NT = GATHER != 2 && JB_DDMC_NT != 0;  // non-temporal swarm accesses (see swarm_ld)
  constexpr bool CODES = GATHER == 4;
  // PF (records in LDS: the event loop issues no vector-memory load, so one requested before it is
  // not waited for until it is needed -- the counter of outstanding loads completes in order):
  // the wave keeps a WINDOW of the next 64 slots of the swarm it will hand to its lanes, requested
  // one service phase ahead.  Lane l holds the window's entry at position (l - win_head) mod 64;
  // lanes that need a particle take the first entries through the lane crossbar (ds_bpermute) and
  // exactly those entries are requested again for the slots behind the window -- consecutive lanes,
  // consecutive slots, every particle read once.  Without it a service phase waited for its own
  // requests: 54 % of the waves' time on BASELINE configs[2] as shipped, SIMDs 57 % busy.
  // (1-D: with the index registers of a second and third axis the window does not fit 128 registers)
  constexpr bool PF = GATHER == 2 && NDIM == 1 && JB_DDMC_ALL_WINDOW != 0;
  // (the LDS tally of a small mesh -- all resident blocks' cells, <= kLdsTally -- and the LDS copy
  // of its step records are dynamic shared memory, sized by the launch: a mesh that uses neither
  // leaves the room to a fourth workgroup per CU)
  // (16-byte aligned: the record table carved out of it is read as 32-byte vectors, ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
  double *const lds_tally = lds_dyn;
  const bool tally_in_lds = TALLY && (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  const int ncell_all = M.nblocks * (int)M.ntot;
  // (behind the tally, on a 16-byte boundary)
  const double *const lds_rec_tab = lds_dyn + (tally_in_lds ? (ncell_all + 1) / 2 * 2 : 0);
  if constexpr (TALLY) {
    if (tally_in_lds)
      for (int q = threadIdx.x; q < ncell_all; q += blockDim.x) lds_tally[q] = 0.0;
  }
  if constexpr (GATHER == 2) {
    double *const dst = lds_dyn + (tally_in_lds ? (ncell_all + 1) / 2 * 2 : 0);
    for (int q = threadIdx.x; q < 8 * ncell_all; q += blockDim.x) dst[q] = ((gcptr)M.ddmc_step)[q];
  }
  if constexpr (CODES) {   // the distinct step records of this cycle (the launch sized the room for them)
    double *const dst = lds_dyn + (tally_in_lds ? (ncell_all + 1) / 2 * 2 : 0);
    const int ncls = ((gcptr_i)M.not_all_ddmc)[1];
    for (int q = threadIdx.x; q < 8 * ncls; q += blockDim.x) dst[q] = ((gcptr)M.ddmc_class)[q];
  }
  __shared__ LdsBlockTableT<false> lds_blocks;
  // [wave][quad-lane k][quad q]: the record of lane 4 q + k of the wave (LDS-direct loads deposit
  // lane l's 16 bytes at base + 16 l: the four pieces of a quad's record land side by side)
  __shared__ __attribute__((aligned(16))) char lds_rec[COOP ? kBlock / 64 : 1][COOP ? 4 : 1][COOP ? 1024 : 16];
  fill_block_table(M, lds_blocks);
  load_math_tables<true, false, false, true>();  // logarithm, sincos of 2 pi u (ends with a barrier)
  constexpr bool multi_d = NDIM >= 2;
  constexpr bool coop_wide = GATHER == 3;
  constexpr int kBudget = JB_DDMC_ALL_BUDGET;
  constexpr long long kChunk = JB_DDMC_ALL_CHUNK;
  const double vv = P.c;
  const double t_end = t_start + dt;
  const int lane = threadIdx.x & 63;
  unsigned long long *queue = counters + CNT_QUEUE;
  const long long per_q = (last - first + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues, tried = 0;
  bool more = true;
  long long chunk_pos = 0, chunk_end = 0;

  // wave-level counters (scalar registers; updated outside divergent branches): finished histories
  // by outcome; events = running lanes summed over the event-loop passes + service-phase steps
  unsigned int c_census = 0, c_abs = 0, c_esc = 0, c_out = 0;
  unsigned long long c_ev = 0;
  unsigned int c_pass = 0, c_service = 0;
  unsigned long long c_ghost = 0;   // lane-passes spent on a ghost record (counted in c_ev, taken off at the end)

  // ---- lane state that lives across the event loop ("virtual" particle): slot, stream state, time,
  //      the RECORD NUMBER of its cell (block * cells per block + cell: what the step gathers with, and
  //      what a leak moves by +-1, +-ni, +-ni nj), the pending leak and the stream state in front of it
  int ls = DS_IDLE;
  long long n = 0;
  LcgRng rng(0);
  unsigned rec = 0u;
  double t = 0.0;
  // pd: the record-number step of the last leak while its direction is deferred (+-1 / +-ni / +-ni nj:
  // channel x-, x+, y-, y+, z-, z+); 0: none -- the direction is the one in S.vx, vy, vz [n];
  // kPdZero: zero, the flag a multi-D DDMC leak across a block face leaves behind
  // (transport_ddmc.cpp:203-211)
  int pd = 0;
  bool mir = false;   // (1-D) the pending leak's direction has been mirrored by a reflecting wall an odd number of times
  // ---- ... and what exists between two event loops only (decoded from rec / pd behind the loop):
  // block, cell indices, and the leak as channel 0..5 / -1 none / -2 zero-velocity flag
  int b = 0, ip = 0, jp = 0, kp = 0;
  int pend = -1;
  // ... kept as the stream state right before them: the loop only steps the stream past the two
  // draws (one multiply-add by the two-step constants instead of two draws with their conversions)
  unsigned long long pzs = 0ull;
  auto pending_uniforms = [&](double &u1, double &u2) {
    LcgRng at_leak(pzs);
    u1 = at_leak.drand();
    u2 = at_leak.drand();
  };
  double wgt_l = 0.0;     // (PF) the weight, requested when the particle is taken (else read at the tally)
  // (PF) the window: per lane one requested slot of the swarm and what the kernel reads of it
  int win_head = 0, win_count = 0;
  long long pf_slot = -1;
  int pf_st = ST_ABSORBED, pf_b = 0;
  unsigned long long pf_rng = 0ull;
  double pf_t = 0.0, pf_x = 0.0, pf_y = 0.0, pf_z = 0.0, pf_vx = 0.0, pf_vy = 0.0, pf_vz = 0.0;
  bool resample = false;  // reached census in a DDMC step: position / direction to be resampled
  bool fresh = false;     // loaded and not touched since: handing it over needs no write-back
  // ---- state that exists only between two points of one service phase
  int status = ST_ACTIVE;
  double x = 0.0, y = 0.0, z = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
  bool real_pos = false;  // x, y, z, v hold the particle's position / direction (DS_RELOC lanes)

  auto faces_of = [&](Step &s, const Blk &Bq, int i, int j, int k) {  // transport.cpp:114-119
    s.xl = xc(Bq, 0, i) - 0.5 * Bq.dx[0]; s.xu = xc(Bq, 0, i) + 0.5 * Bq.dx[0];
    s.yl = xc(Bq, 1, j) - 0.5 * Bq.dx[1]; s.yu = xc(Bq, 1, j) + 0.5 * Bq.dx[1];
    s.zl = xc(Bq, 2, k) - 0.5 * Bq.dx[2]; s.zu = xc(Bq, 2, k) + 0.5 * Bq.dx[2];
  };
  // Uniform values the event loop reads, as scalar registers of their own: the kernel arguments
  // arrive in 16-dword groups, and a group the compiler parks in VGPR lanes comes back whole
  // (16 v_readlane per pass for one dword of it).
  const unsigned ntot_u = sgpr_copy((unsigned)M.ntot);
  // step records, or (CODES) the cell codes: what the event loop gathers from
  const double *step_base = sgpr_copy_ptr(CODES ? (const double *)M.ddmc_code : M.ddmc_step);
  typedef const unsigned __attribute__((address_space(1))) *gcptr_u;
  unsigned code = 0u;   // (CODES) the code of the lane's cell, requested at the end of the pass before
  const int l_ni = (int)sgpr_copy((unsigned)M.ni), l_nj = (int)sgpr_copy((unsigned)M.nj);
  const int l_is = (int)sgpr_copy((unsigned)M.is), l_ie = (int)sgpr_copy((unsigned)M.ie);
  const int l_js = (int)sgpr_copy((unsigned)M.js), l_je = (int)sgpr_copy((unsigned)M.je);
  const int l_ks = (int)sgpr_copy((unsigned)M.ks), l_ke = (int)sgpr_copy((unsigned)M.ke);
  auto cidx_l = [&](int k, int j, int i) { return __mul24(__mul24(k, l_nj) + j, l_ni) + i; };
  // record-number strides of the three axes (and their reciprocals, for the decode behind the loop)
  const int l_nij = (int)sgpr_copy((unsigned)(M.ni * M.nj));
  const double inv_ntot = M.inv_ntot, inv_nij = M.inv_nij, inv_ni = M.inv_ni;   // (jb_mesh_create)
  // floor(x / d) for x < 2^32 with inv = 1.0 / d: the product is within 2^-20 of the quotient, so the
  // truncation is the floor or one below it (an exact multiple of d seen from below)
  auto udiv = [](unsigned x, unsigned d, double inv, unsigned &rem) {
    unsigned q = (unsigned)((double)x * inv);
    rem = x - q * d;
    if (rem >= d) { ++q; rem -= d; }
    return q;
  };
  // (the index along an inactive axis is the block's first there and never moves: Xtoijk, the step)
  auto on_block_l = [&](int i, int j, int k) {
    bool on = i >= l_is && i <= l_ie;
    if constexpr (NDIM >= 2) on = on && j >= l_js && j <= l_je;
    if constexpr (NDIM == 3) on = on && k >= l_ks && k <= l_ke;
    return on;
  };
  typedef double v4d __attribute__((ext_vector_type(4)));
  const unsigned sub16 = 16u * (unsigned)(lane & 3);
  const v4d *my_rec = (const v4d *)&lds_rec[COOP ? threadIdx.x >> 6 : 0][COOP ? lane & 3 : 0][COOP ? 64 * (lane >> 2) : 0];
  typedef __attribute__((address_space(3))) char *lchar;
  lchar const wave_buf = (lchar)(unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(lchar)&lds_rec[COOP ? threadIdx.x >> 6 : 0][0][0]);
  // a particle with a real position (just loaded / just relocated) enters the loop unless the
  // albedo step would find it at a face of its cell
  auto enter = [&](const Blk &Bq) {
    Step s;
    faces_of(s, Bq, ip, jp, kp);
    s.x = x; s.y = y; s.z = z;
    // (selects: stores to different variables in the two arms of a branch are merged into one store through
    // a selected address, and those variables then live in scratch memory)
    const bool face = at_cell_face<NDIM>(s);
    ls = face ? DS_PARK : DS_VIRT;
    real_pos = face && real_pos;
    rec = face ? rec : (unsigned)b * ntot_u + (unsigned)cidx_l(kp, jp, ip);
    pd = face ? pd : 0;   // (pend is -1 wherever a particle enters: just loaded, or relocated)
    mir = face && mir;
  };

#ifdef JB_TIMING
  unsigned c_epi = 0;   // event-loop episodes (entries of the loop): services per episode = c_service / c_epi
  unsigned long long cyc_ev = 0, cyc_sv = 0, cyc_mark = __builtin_readcyclecounter();
  unsigned long long cyc_ph[6] = {0, 0, 0, 0, 0, 0}, ph_mark = 0;  // reloc, claim, done, take, real, tail
#define JB_PH(k) { const unsigned long long now_ = __builtin_readcyclecounter(); cyc_ph[k] += now_ - ph_mark; ph_mark = now_; }
#else
#define JB_PH(k)
#endif
  for (;;) {
    // ================================ SERVICE ================================
    ++c_service;
#ifdef JB_TIMING
    { const unsigned long long now = __builtin_readcyclecounter(); cyc_ev += now - cyc_mark; cyc_mark = now; }
#endif
#ifdef JB_TIMING
    ph_mark = __builtin_readcyclecounter();
#endif
    // (PF: the window's requests are a whole event loop old; saying so here, before this phase
    // issues stores, keeps the wait for them from being placed behind those stores)
    if constexpr (PF) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    // -- 1. block crossings: the comm phase of the reference for one particle in flight
    if (ls == DS_RELOC) {
      fresh = false;
      Blk Bo;
      load_block_lds(M, lds_blocks, b, Bo);
      Step s;
      s.vv = vv; s.pend = pend; s.pz1 = 0.0; s.pz2 = 0.0;
      if (pend >= 0) pending_uniforms(s.pz1, s.pz2);
      if (!real_pos) {
        // leak out of the block from the virtual state: the position the step function gave the
        // particle (transport_utils.hpp:209-263), from the cell it left and the channel
        const int axis = pend >> 1;
        const bool up = (pend & 1) != 0;
        const int step = up ? 1 : -1;
        faces_of(s, Bo, ip - (axis == 0 ? step : 0), jp - (axis == 1 ? step : 0),
                 kp - (axis == 2 ? step : 0));
        const double dx = s.xu - s.xl, dy = s.yu - s.yl, dz = s.zu - s.zl;
        const double eps = kEpsDdmc;
        x = (axis == 0) ? (up ? s.xu + eps * dx : s.xl - eps * dx) : s.xl + 0.5 * dx;
        y = (axis == 1) ? (up ? s.yu + eps * dy : s.yl - eps * dy) : s.yl + 0.5 * dy;
        z = (axis == 2) ? (up ? s.zu + eps * dz : s.zl - eps * dz) : s.zl + 0.5 * dz;
        vx = 0.0; vy = 0.0; vz = 0.0;  // (overwritten below or flagged as a DDMC leak)
      }
      // transport_ddmc.cpp:203-211: zero velocity flags a DDMC leak for SampleDDMCBlockFace
      // (multi-D); in 1-D the direction travels with the particle
      if (pend >= 0) {
        if constexpr (multi_d) {
          vx = 0.0; vy = 0.0; vz = 0.0;
        } else {
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = mir ? -s.vx : s.vx; vy = s.vy; vz = s.vz;
          mir = false;
        }
        pend = -1;
      }
      if (!apply_swarm_bcs<NDIM>(M, x, y, z, vx, vy, vz)) {
        status = ST_ESCAPED;
        ls = DS_DONE;
      } else {
        const int g = find_block<NDIM>(M, x, y, z);
        const int li = M.local_index[g];
        if (li < 0) {  // not resident here: hand the particle to the block's owner
          status = ST_OUTGOING;
          b = g;  // global id travels in blk
          ls = DS_DONE;
        } else {
          b = li;
          Blk Bn;
          load_block_lds(M, lds_blocks, b, Bn);
          if constexpr (multi_d)
            sample_block_face<NDIM>(M, P, Bn, b, rng, x, y, z, vx, vy, vz, ip, jp, kp);
          xtoijk<NDIM>(M, Bn, x, y, z, ip, jp, kp);
          real_pos = true;
          if (t < t_end) {
            enter(Bn);
            if (ls == DS_VIRT) {  // (the loop does not carry the direction: park it)
              swarm_st<NT>(&g1(S.vx)[n], vx); swarm_st<NT>(&g1(S.vy)[n], vy); swarm_st<NT>(&g1(S.vz)[n], vz);
            }
          } else {
            ls = DS_DONE;
          }
        }
      }
      if (ls != DS_VIRT) real_pos = true;
    }
    JB_PH(0)
    // -- 2a. every lane that is idle, or will be once its finished particle is written back,
    //        claims the next slot of the wave's chunk (chunks of consecutive slots, one atomic
    //        per chunk) and REQUESTS that particle now: vector-memory operations complete in
    //        issue order, so loads issued after the write-back below would wait for its stores
    //        and its tally atomic
    long long cand = -1;
    int st_in = ST_ABSORBED, b_in = 0;
    unsigned long long rng_in = 0ull;
    double t_in = 0.0, x_in = 0.0, y_in = 0.0, z_in = 0.0, vx_in = 0.0, vy_in = 0.0, vz_in = 0.0;
    // (PF: who takes which entry of the window is settled here, the entries move in 3b -- behind the
    // write-back, so that through it a lane holds one set of particle registers, not two)
    bool pf_mine = false;
    int pf_src = 0, pf_give = 0;
    if constexpr (PF) {
      const unsigned long long need = __ballot(ls == DS_IDLE || ls == DS_DONE);
      const int want = __popcll(need);
      pf_give = want < win_count ? want : win_count;
      const int rank = __popcll(need & ((1ull << lane) - 1ull));
      pf_mine = ((need >> lane) & 1ull) != 0ull && rank < pf_give;
      pf_src = (win_head + rank) & 63;
    } else {
      unsigned long long need = __ballot(ls == DS_IDLE || ls == DS_DONE);
      while (need != 0ull && more) {
        if (chunk_pos >= chunk_end) {
          const long long q_first = first + (long long)cur * per_q;
          long long q_last = q_first + per_q;
          if (q_last > last) q_last = last;
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&queue[cur * kQueueStride], (unsigned long long)kChunk);
          chunk_pos = q_first + (long long)uniform_u64(base);
          chunk_end = chunk_pos + kChunk < q_last ? chunk_pos + kChunk : q_last;
          if (chunk_pos >= q_last) {  // this queue is drained: move on
            chunk_pos = chunk_end = 0;
            cur = (cur + 1) % kQueues;
            if (++tried == kQueues) more = false;
            continue;
          }
        }
        const int want = __popcll(need);
        const long long avail = chunk_end - chunk_pos;
        const int give = (long long)want < avail ? want : (int)avail;
        const int rank = __popcll(need & ((1ull << lane) - 1ull));
        const bool mine = ((need >> lane) & 1ull) != 0ull && rank < give;
        if (mine) {
          cand = chunk_pos + rank;
          st_in = swarm_ld<NT>(&g1(S.status)[cand]);
          rng_in = swarm_ld<NT>(&g1(S.rng)[cand]);
          b_in = swarm_ld<NT>(&g1(S.blk)[cand]);
          t_in = swarm_ld<NT>(&g1(S.t)[cand]); x_in = swarm_ld<NT>(&g1(S.x)[cand]); y_in = swarm_ld<NT>(&g1(S.y)[cand]); z_in = swarm_ld<NT>(&g1(S.z)[cand]);
          vx_in = swarm_ld<NT>(&g1(S.vx)[cand]); vy_in = swarm_ld<NT>(&g1(S.vy)[cand]); vz_in = swarm_ld<NT>(&g1(S.vz)[cand]);
        }
        chunk_pos += give;
        need &= ~__ballot(mine);
      }
    }
    JB_PH(1)
    // -- 2b. finished particles: census resampling, write-back, tally
    const bool was_done = ls == DS_DONE;
    if (ls == DS_DONE) {
      bool write_v = real_pos;  // else: unchanged since it was loaded (or parked), or set below
      if (status != ST_OUTGOING && status != ST_ESCAPED) {
        Blk Bd;
        load_block_lds(M, lds_blocks, b, Bd);
        Step s;
        s.vv = vv;
        faces_of(s, Bd, ip, jp, kp);
        if (resample) {  // transport_utils.hpp:265-276, once per history
          ddmc_census_resample(s, rng);
          x = s.x; y = s.y; z = s.z; vx = s.vx; vy = s.vy; vz = s.vz;
          pend = -1;
          write_v = true;
        } else if (!real_pos) {
          // absorbed (or already at census when it was loaded) in the virtual state: the albedo
          // step left it at the cell centre (transport_utils.hpp:392-396), with the direction of
          // its last leak
          x = 0.5 * (s.xl + s.xu); y = 0.5 * (s.yl + s.yu); z = 0.5 * (s.zl + s.zu);
        }
        if (pend >= 0) {
          s.pend = pend;
          pending_uniforms(s.pz1, s.pz2);
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = s.vx; vy = s.vy; vz = s.vz;
          if constexpr (NDIM == 1) { vx = mir ? -vx : vx; mir = false; }
          pend = -1;
          write_v = true;
        } else if (pend == -2) {
          vx = 0.0; vy = 0.0; vz = 0.0;
          pend = -1;
          write_v = true;
        }
        if ((status == ST_ACTIVE || status == ST_OUTGOING_ABSORBED) && lds_blocks.owned[b] == 0) {
          if (status == ST_ACTIVE) status = ST_OUTGOING;  // the owner of the block tallies it
          b = M.gid[b];
        } else if (status == ST_ACTIVE) {
          if constexpr (TALLY) {  // jaybenne.cpp:547-561
            const double dv = Bd.dx[0] * Bd.dx[1] * Bd.dx[2];
            // (the weight is read here, once per history, rather than carried through the event loop)
            const double wgt = PF ? wgt_l : swarm_ld<NT>(&g1(S.w)[n]);
            if (tally_in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + cidx(M, kp, jp, ip)], wgt / dv);
            else atomicAdd(&lds_blocks.tally[b][cidx(M, kp, jp, ip)], wgt / dv);
          }
        }
      }
      swarm_st<NT>(&g1(S.blk)[n], b);
      swarm_st<NT>(&g1(S.t)[n], t);
      swarm_st<NT>(&g1(S.x)[n], x); swarm_st<NT>(&g1(S.y)[n], y); swarm_st<NT>(&g1(S.z)[n], z);
      if (write_v) {
        swarm_st<NT>(&g1(S.vx)[n], vx); swarm_st<NT>(&g1(S.vy)[n], vy); swarm_st<NT>(&g1(S.vz)[n], vz);
      }
      swarm_st<NT>(&g1(S.ip)[n], ip); swarm_st<NT>(&g1(S.jp)[n], jp); swarm_st<NT>(&g1(S.kp)[n], kp);
      swarm_st<NT>(&g1(S.status)[n], status);
      swarm_st<NT>(&g1(S.rng)[n], rng.s);
      resample = false;
      ls = DS_IDLE;
    }
    {
      const int n_done = __popcll(__ballot(was_done));
      const int n_census = __popcll(__ballot(was_done && status == ST_ACTIVE));
      const int n_abs = __popcll(__ballot(was_done && status == ST_ABSORBED));
      const int n_esc = __popcll(__ballot(was_done && status == ST_ESCAPED));
      c_census += n_census; c_abs += n_abs; c_esc += n_esc;
      c_out += n_done - n_census - n_abs - n_esc;
    }
    JB_PH(2)
    // -- 3b. the lanes that claimed a slot in 2a take their new particle
    if constexpr (PF) {
      // (every lane takes part in the exchange: a lane outside the execution mask supplies nothing)
      const long long c = __shfl(pf_slot, pf_src, 64);
      st_in = __shfl(pf_st, pf_src, 64); b_in = __shfl(pf_b, pf_src, 64);
      rng_in = __shfl(pf_rng, pf_src, 64);
      t_in = __shfl(pf_t, pf_src, 64);
      x_in = __shfl(pf_x, pf_src, 64); y_in = __shfl(pf_y, pf_src, 64); z_in = __shfl(pf_z, pf_src, 64);
      vx_in = __shfl(pf_vx, pf_src, 64); vy_in = __shfl(pf_vy, pf_src, 64); vz_in = __shfl(pf_vz, pf_src, 64);
      if (pf_mine) {
        cand = c;
        // (the weight is not part of the window: asked for now, read when the history ends)
        wgt_l = swarm_ld<NT>(&g1(S.w)[c]);
      }
      win_head = (win_head + pf_give) & 63;
      win_count -= pf_give;
      // ... and the window is filled up again: the entries just taken (and, at the start, all of
      // them) are requested for the next slots of the wave's chunk -- needed one event loop from now.
      // The free entries take what is left of the current chunk, then the head of the next one (a
      // chunk holds two windows; a queue's short tail may leave entries free until the next phase);
      // the requests themselves are one straight-line block.
      const int free_n = 64 - win_count;
      long long seg_first[2] = {0, 0};
      int seg_n[2] = {0, 0};
      int got = 0;
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        while (got < free_n && more && chunk_pos >= chunk_end) {
          const long long q_first = first + (long long)cur * per_q;
          long long q_last = q_first + per_q;
          if (q_last > last) q_last = last;
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&queue[cur * kQueueStride], (unsigned long long)kChunk);
          chunk_pos = q_first + (long long)uniform_u64(base);
          chunk_end = chunk_pos + kChunk < q_last ? chunk_pos + kChunk : q_last;
          if (chunk_pos >= q_last) {  // this queue is drained: move on
            chunk_pos = chunk_end = 0;
            cur = (cur + 1) % kQueues;
            if (++tried == kQueues) more = false;
          }
        }
        if (got < free_n && chunk_pos < chunk_end) {
          const long long avail = chunk_end - chunk_pos;
          const int give = (long long)(free_n - got) < avail ? free_n - got : (int)avail;
          seg_first[sg] = chunk_pos;
          seg_n[sg] = give;
          chunk_pos += give;
          got += give;
        }
      }
      {
        const int pos = (lane - win_head - win_count) & 63;  // this lane's place among the free entries
        if (pos < got) {
          const long long q = pos < seg_n[0] ? seg_first[0] + pos : seg_first[1] + (pos - seg_n[0]);
          pf_slot = q;
          pf_st = swarm_ld<NT>(&g1(S.status)[q]);
          pf_rng = swarm_ld<NT>(&g1(S.rng)[q]);
          pf_b = swarm_ld<NT>(&g1(S.blk)[q]);
          pf_t = swarm_ld<NT>(&g1(S.t)[q]);
          pf_x = swarm_ld<NT>(&g1(S.x)[q]); pf_y = swarm_ld<NT>(&g1(S.y)[q]); pf_z = swarm_ld<NT>(&g1(S.z)[q]);
          pf_vx = swarm_ld<NT>(&g1(S.vx)[q]); pf_vy = swarm_ld<NT>(&g1(S.vy)[q]); pf_vz = swarm_ld<NT>(&g1(S.vz)[q]);
        }
        win_count += got;
      }
    }
    if (ls == DS_IDLE && cand >= 0 && st_in == ST_ACTIVE) {
      n = cand;
      rng.s = rng_in;
      b = b_in;
      t = t_in;
      x = x_in; y = y_in; z = z_in; vx = vx_in; vy = vy_in; vz = vz_in;
      status = ST_ACTIVE;
      resample = false;
      pend = -1;
      real_pos = true;
      fresh = true;
      Blk Bn;
      load_block_lds(M, lds_blocks, b, Bn);
      xtoijk<NDIM>(M, Bn, x, y, z, ip, jp, kp);  // transport.cpp:96
      if (t < t_end) enter(Bn);
      else ls = DS_DONE;  // already at census: nothing to track
    }
    JB_PH(3)
    // -- 4. a particle that sits at a face of its cell (just loaded, or just relocated): handed
    //       over to k_hybrid as it stands
    if (ls == DS_PARK) {
      if (!fresh) {
        swarm_st<NT>(&g1(S.blk)[n], b);
        swarm_st<NT>(&g1(S.t)[n], t);
        swarm_st<NT>(&g1(S.x)[n], x); swarm_st<NT>(&g1(S.y)[n], y); swarm_st<NT>(&g1(S.z)[n], z);
        swarm_st<NT>(&g1(S.vx)[n], vx); swarm_st<NT>(&g1(S.vy)[n], vy); swarm_st<NT>(&g1(S.vz)[n], vz);
        swarm_st<NT>(&g1(S.rng)[n], rng.s);
      }
      const unsigned long long pm = __ballot(true);
      const int leader = __ffsll((long long)pm) - 1;
      unsigned long long base = 0ull;
      if (lane == leader) base = atomicAdd(g1(A.park_count), (unsigned long long)__popcll(pm));
      base = __shfl(base, leader, 64);
      g1(A.park_list)[base + (unsigned long long)__popcll(pm & ((1ull << lane) - 1ull))] = (unsigned)n;
      ls = DS_IDLE;
    }
    JB_PH(4)
    // Lanes that still hold a real position (a step at a face, a general relocation, a particle
    // to write back) are served before the loop is entered again: position and direction are
    // then dead across the event loop and cost it no registers.
    if (__ballot(ls == DS_DONE || ls == DS_RELOC) != 0ull) continue;
    const int running = __popcll(__ballot(ls == DS_VIRT));
    if (running == 0) {
      if (more || win_count > 0) continue;   // (PF: the window still holds requested slots)
      break;
    }
    int waste = 0;

    // ================================ EVENTS =================================
#ifdef JB_TIMING
    { const unsigned long long now = __builtin_readcyclecounter(); cyc_sv += now - cyc_mark; cyc_mark = now; }
#endif
#ifdef JB_TIMING
    ++c_epi;
#endif
    int thresh = 1;
    int nrun = running;
    // One DDMC step per running lane and pass (transport_utils.hpp:163-263 on the virtual state), as
    // straight-line code: the three stream states a step can end in (no event: one draw; an event that
    // is no leak: two; a leak: four -- its two direction uniforms stay deferred) are formed side by
    // side from the state the step starts in, the channel walk of :218-254 selects the record-number
    // step of the leak, and the outcome is committed through selects.  "Has the particle left its
    // block?" is read off the record the next pass gathers anyway (kStepGhostTable above).
    constexpr unsigned long long kMul2 = kLcgMul * kLcgMul, kInc2 = (kLcgMul + 1ull) * kLcgInc;
    constexpr unsigned long long kMul4 = kMul2 * kMul2, kInc4 = (kMul2 + 1ull) * kInc2;
    if constexpr (CODES) {
      // (nothing of the service phase is left in flight when the loop starts: the compiler places a full wait
      // at the loop's top otherwise -- for registers the service phase's loads were headed for -- and that wait
      // would then be taken in every pass, in front of the code's first use)
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      code = *(gcptr_u)((const char *)step_base + ((ls == DS_VIRT ? rec : 0u) << 2));
    }
    while (nrun >= thresh) {
      ++c_pass;
      c_ev += (unsigned int)nrun;
      if constexpr (CODES) {
        // The same step (transport_utils.hpp:184-263 on the virtual state: see the general form below) arranged
        // around ONE 4-byte gather whose result is not needed until a third of the way into the NEXT pass: what
        // does not depend on the cell -- the three stream states, the two uniforms, the logarithm -- comes first;
        // then the code that was requested at the end of the pass before is read (its register is dead from
        // there on: the request for the next pass can land in it without a copy, i.e. without a wait at the
        // loop's tail); everything is committed through selects, no branch but the ghost lanes' block.
        const bool run = ls == DS_VIRT;
        const unsigned long long s0 = rng.s;
        const unsigned long long s1 = s0 * kLcgMul + kLcgInc;
        const unsigned long long s2 = s0 * kMul2 + kInc2;
        const unsigned long long s4 = s0 * kMul4 + kInc4;
        const double u2 = u52_to_double(s2 >> 12);
        const double nlog = -m_log(u52_to_double(s1 >> 12));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned cd = code;
        const bool ghost = (int)cd < 0;
        // (a ghost cell has no record: any row will do for the arithmetic nobody commits)
        const v4d *rp = (const v4d *)(lds_rec_tab + 8u * (ghost ? 0u : cd));
        const v4d r0 = rp[0];
        const v4d r1 = rp[1];
        DdmcStepRec r;
        r.ffaa = r0.x; r.c1 = r0.y; r.c2 = r0.z; r.c3 = r0.w;
        r.c4 = r1.x; r.c5 = r1.y; r.leak_tot = r1.z; r.rcp = r1.w;
        const bool gl = run && ghost;
        const unsigned long long gm = __ballot(gl);
        c_ghost += (unsigned int)__popcll(gm);   // (no step taken: not an event)
        if (gm != 0ull) {   // a leak through a block face: see the general form below
          const bool tab = gl && (cd & kCodeTable) != 0u;
          rec = tab ? cd & kCodeRecMask : rec;
          if constexpr (multi_d) pd = tab ? kPdZero : pd;
          if constexpr (NDIM == 1) mir = (tab && (cd & kCodeMirror) != 0u) ? !mir : mir;
          ls = (gl && !tab) ? DS_RELOC : ls;
        }
        const bool live = run && !ghost;
        // transport_utils.hpp:184-191
        const double a2 = r.ffaa + r.leak_tot;
        const double cdf_ddmc = a2 + DBL_MIN;
        const double dt_ddmc = m_div_r(nlog, vv * cdf_ddmc, r.rcp);
        const double dt_end = t_end - t;
        const bool ev = dt_ddmc < dt_end;
        const double t_new = t + dmin(dt_ddmc, dt_end);
        // :196-254
        const double xi = cdf_ddmc * u2;
        const bool absorbed = xi < r.ffaa;
        const double xim = xi - r.ffaa;
        int delta = (xim <= r.leak_tot) ? (NDIM == 3 ? l_nij : kPdStay) : 0;
        if constexpr (NDIM == 3) delta = (xim < r.c5) ? -l_nij : delta;
        if constexpr (multi_d) {
          delta = (xim < r.c4) ? l_ni : delta;
          delta = (xim < r.c3) ? -l_ni : delta;
        }
        delta = (xim < r.c2) ? 1 : delta;
        delta = (xim < r.c1) ? -1 : delta;
        const bool leak = live && ev && !absorbed && xi < a2 && delta != 0;
        const bool done = !(t_new < t_end);
        t = live ? t_new : t;
        rng.s = live ? (leak ? s4 : (ev ? s2 : s1)) : rng.s;
        pzs = leak ? s2 : pzs;
        pd = leak ? delta : pd;
        if constexpr (NDIM == 1) mir = mir && !leak;   // (a new leak: a new direction)
        if constexpr (NDIM == 3) rec = leak ? rec + (unsigned)delta : rec;
        else rec = (leak && delta != kPdStay) ? rec + (unsigned)delta : rec;
        ls = live ? ((ev && absorbed) ? DS_ABS : (done ? (ev ? DS_DONE : DS_CENSUS) : DS_VIRT)) : ls;
        // the code of the cell the lane is in now, for the next pass (a lane that has left the loop asks for word 0)
        code = *(gcptr_u)((const char *)step_base + ((ls == DS_VIRT ? rec : 0u) << 2));   // (32-bit byte offset: < 2 GiB of codes)
        // (keeps the request HERE: left to itself the compiler merges it with the one in front of the loop into
        // the loop's header, where its result is needed a dozen instructions later)
        __builtin_amdgcn_sched_barrier(0);
        nrun = __popcll(__ballot(ls == DS_VIRT));
        waste += running - nrun;
        if (waste >= kBudget) thresh = 65;
        continue;
      }
      const bool run = ls == DS_VIRT;
      // (every lane takes part in the gather; one without a particle in the loop asks for record 0)
      const unsigned rq = run ? rec : 0u;
      if constexpr (COOP) {
        typedef const __attribute__((address_space(1))) void *gvoid;
        if constexpr (!coop_wide) {  // the records span < 4 GiB: 32-bit byte offsets from a scalar base
          const unsigned off = rq << 6;
          __builtin_amdgcn_global_load_lds((gvoid)((const char *)step_base + quad_bcast_add<0>(off, sub16)), wave_buf, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)((const char *)step_base + quad_bcast_add<1>(off, sub16)), wave_buf + 1024, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)((const char *)step_base + quad_bcast_add<2>(off, sub16)), wave_buf + 2048, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)((const char *)step_base + quad_bcast_add<3>(off, sub16)), wave_buf + 3072, 16, 0, 0);
        } else {           // ... or more (> 67e6 resident cells): the record number travels, 64-bit addresses
          const char *const mine = (const char *)step_base + sub16;
          __builtin_amdgcn_global_load_lds((gvoid)(mine + ((unsigned long long)quad_bcast_add<0>(rq, 0u) << 6)), wave_buf, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)(mine + ((unsigned long long)quad_bcast_add<1>(rq, 0u) << 6)), wave_buf + 1024, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)(mine + ((unsigned long long)quad_bcast_add<2>(rq, 0u) << 6)), wave_buf + 2048, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gvoid)(mine + ((unsigned long long)quad_bcast_add<3>(rq, 0u) << 6)), wave_buf + 3072, 16, 0, 0);
        }
      }
      const unsigned long long s0 = rng.s;
      const unsigned long long s1 = s0 * kLcgMul + kLcgInc;
      DdmcStepRec r;
      double nlog;
      if constexpr (COOP) {
        nlog = -m_log(u52_to_double(s1 >> 12));  // (the step's first draw, while the record is on its way)
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the four pieces have landed
        const v4d r0 = my_rec[0];
        const v4d r1 = my_rec[1];
        r.ffaa = r0.x; r.c1 = r0.y; r.c2 = r0.z; r.c3 = r0.w;
        r.c4 = r1.x; r.c5 = r1.y; r.leak_tot = r1.z; r.rcp = r1.w;
      } else if constexpr (GATHER == 2) {
        const v4d *rp = (const v4d *)(lds_rec_tab + 8u * rq);
        const v4d r0 = rp[0];
        const v4d r1 = rp[1];
        nlog = -m_log(u52_to_double(s1 >> 12));
        r.ffaa = r0.x; r.c1 = r0.y; r.c2 = r0.z; r.c3 = r0.w;
        r.c4 = r1.x; r.c5 = r1.y; r.leak_tot = r1.z; r.rcp = r1.w;
      } else {
        typedef const v4d __attribute__((address_space(1))) *grec;
        const grec rp = (grec)((gcptr)step_base + 8ull * (unsigned long long)rq);
        const v4d r0 = rp[0];
        const v4d r1 = rp[1];
        nlog = -m_log(u52_to_double(s1 >> 12));
        r.ffaa = r0.x; r.c1 = r0.y; r.c2 = r0.z; r.c3 = r0.w;
        r.c4 = r1.x; r.c5 = r1.y; r.leak_tot = r1.z; r.rcp = r1.w;
      }
      const int rcp_hi = __double2hiint(r.rcp);
      const bool ghost = rcp_hi < 0;
      // transport_utils.hpp:184-191
      const double a2 = r.ffaa + r.leak_tot;
      const double cdf_ddmc = a2 + DBL_MIN;
      const double dt_ddmc = m_div_r(nlog, vv * cdf_ddmc, r.rcp);
      const double dt_end = t_end - t;
      const bool ev = dt_ddmc < dt_end;
      const double t_new = t + dmin(dt_ddmc, dt_end);
      // :196-254 (what an event does; committed below only if there was one)
      const unsigned long long s2 = s0 * kMul2 + kInc2;
      const double xi = cdf_ddmc * u52_to_double(s2 >> 12);
      const bool absorbed = xi < r.ffaa;
      const double xim = xi - r.ffaa;
      // the first threshold above xim, as the chain of :218-254 finds it (the leak opacities of an
      // inactive axis are exactly zero -- k_ddmc_pack -- so its two thresholds repeat the one before
      // them and their comparisons can never be the first to hold); 0: none (xim beyond leak_tot)
      // (without a z axis the last arm, "z+", leaves the cell where it is -- kp += three_d, :256 --: kPdStay)
      int delta = (xim <= r.leak_tot) ? (NDIM == 3 ? l_nij : kPdStay) : 0;
      if constexpr (NDIM == 3) delta = (xim < r.c5) ? -l_nij : delta;   // (2-D: c5 = c4; 1-D: c5 = c4 = c3 = c2)
      if constexpr (multi_d) {
        delta = (xim < r.c4) ? l_ni : delta;
        delta = (xim < r.c3) ? -l_ni : delta;
      }
      delta = (xim < r.c2) ? 1 : delta;
      delta = (xim < r.c1) ? -1 : delta;
      const bool leak = ev && !absorbed && xi < a2 && delta != 0;
      const unsigned long long s4 = s0 * kMul4 + kInc4;
      if (run && !ghost) {
        t = t_new;
        rng.s = leak ? s4 : (ev ? s2 : s1);
        // the leak's two direction uniforms (:217,235,253) are the two draws behind s2: remember where
        // they start; the particle is in the neighbouring cell (ip, jp, kp = Xtoijk of the position the
        // step gives it: see the header)
        pzs = leak ? s2 : pzs;
        pd = leak ? delta : pd;
        if constexpr (NDIM == 1) mir = mir && !leak;   // (a new leak: a new direction)
        if constexpr (NDIM == 3) rec = leak ? rec + (unsigned)delta : rec;
        else rec = (leak && delta != kPdStay) ? rec + (unsigned)delta : rec;
        const bool done = !(t_new < t_end);
        ls = (ev && absorbed) ? DS_ABS : (done ? (ev ? DS_DONE : DS_CENSUS) : DS_VIRT);
      }
      const unsigned long long gm = __ballot(run && ghost);
      c_ghost += (unsigned int)__popcll(gm);   // (no step taken: not an event)
      if (gm != 0ull) {
        // A leak through a block face (0.4 per history on BASELINE configs[2]): the particle sat this
        // pass out.  Into a resident block of the same size -- directly or through a periodic boundary
        // -- nothing happens to it beyond the new cell: SampleDDMCBlockFace only acts on an arrival
        // from a COARSER block (it looks for x_min + 2 eps_ddmc dx, a same-size leak lands at x_min +
        // eps_ddmc dx: sample_ddmc_bface.cpp:158-165), Xtoijk gives the first or last cell along the
        // axis, and in more than one dimension the direction is zeroed (transport_ddmc.cpp:203-211).
        // Everything else goes through the service phase.
        // (selects, not branches with stores to one of two variables: the compiler merges such stores into
        // one store through a selected ADDRESS, which pins both variables in scratch memory)
        const bool gl = run && ghost;
        const bool tab = gl && (rcp_hi & kStepGhostTable) != 0;
        rec = tab ? (unsigned)__double2loint(r.rcp) : rec;
        if constexpr (multi_d) pd = tab ? kPdZero : pd;
        if constexpr (NDIM == 1) mir = (tab && (rcp_hi & kStepGhostMirror) != 0) ? !mir : mir;
        ls = (gl && !tab) ? DS_RELOC : ls;
      }
      nrun = __popcll(__ballot(ls == DS_VIRT));
      waste += running - nrun;
      if (waste >= kBudget) thresh = 65;
    }
    // ---- behind the loop: block, cell indices and leak channel of every lane from its record number
    //      and record-number step (block and indices are dead across the loop: five registers fewer)
    {
      unsigned q, rr, ii;
      const unsigned bb = udiv(rec, ntot_u, inv_ntot, q);
      const unsigned kk = udiv(q, (unsigned)l_nij, inv_nij, rr);
      const unsigned jj = udiv(rr, (unsigned)l_ni, inv_ni, ii);
      b = (int)bb; kp = (int)kk; jp = (int)jj; ip = (int)ii;
      const int ad = pd < 0 ? -pd : pd;
      const int axis = ad == 1 ? 0 : ((NDIM >= 2 && ad == l_ni) ? 1 : 2);   // (kPdStay: z+)
      pend = pd == 0 ? -1 : (pd == kPdZero ? -2 : 2 * axis + (pd > 0 ? 1 : 0));
      status = ST_ACTIVE;
      real_pos = false;
      fresh = false;
      resample = ls == DS_CENSUS;
      if (ls == DS_ABS) {  // transport.cpp:157-163
        if (lds_blocks.owned[b] != 0) {
          atomicAdd(&M.edelta[b][cidx_l(kp, jp, ip)], PF ? wgt_l : swarm_ld<NT>(&g1(S.w)[n]));
          status = ST_ABSORBED;
        } else {
          status = ST_OUTGOING_ABSORBED;
        }
      }
      if (ls == DS_ABS || ls == DS_CENSUS) ls = DS_DONE;
      // (a leak whose event time rounds onto the census time -- one step in ~1e16 -- may end the
      // history in a ghost cell: the general relocation puts it where it belongs)
      if (ls == DS_DONE && !on_block_l(ip, jp, kp)) ls = DS_RELOC;
    }
  }

  if constexpr (TALLY) {
    if (tally_in_lds) {
      __syncthreads();
      for (int q = threadIdx.x; q < M.nblocks * (int)M.ntot; q += blockDim.x) {
        const double v = lds_tally[q];
        if (v != 0.0) atomicAdd(&M.tally[q / (int)M.ntot][q % (int)M.ntot], v);
      }
    }
  }
  const unsigned long long r_census = c_census, r_abs = c_abs, r_esc = c_esc, r_out = c_out, r_ev = c_ev - c_ghost;
  if (lane == 0) {
    if (r_census) atomicAdd(&counters[CNT_CENSUS], r_census);
    if (r_abs) atomicAdd(&counters[CNT_ABSORBED], r_abs);
    if (r_esc) atomicAdd(&counters[CNT_ESCAPED], r_esc);
    if (r_out) atomicAdd(&counters[CNT_OUTGOING], r_out);
    if (r_ev) atomicAdd(&counters[CNT_EVENTS], r_ev);
#ifdef JB_TIMING
    atomicAdd(&counters[CNT_PASSES], cyc_ev >> 10);
    atomicAdd(&counters[CNT_SERVICE], cyc_sv >> 10);
    for (int k = 0; k < 5; ++k) atomicAdd(&counters[24 + k], cyc_ph[k] >> 10);  // (scratch words)
    atomicAdd(&counters[29], (unsigned long long)c_epi);
    atomicAdd(&counters[30], (unsigned long long)c_pass);
    atomicAdd(&counters[31], (unsigned long long)c_service);
#else
    atomicAdd(&counters[CNT_PASSES], (unsigned long long)c_pass);
    atomicAdd(&counters[CNT_SERVICE], (unsigned long long)c_service);
#endif
  }
}

}  // namespace jb
