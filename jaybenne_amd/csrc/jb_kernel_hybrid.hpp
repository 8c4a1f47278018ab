// jb_kernel_hybrid.hpp -- TransportPhotons_DDMC for gray opacities on meshes where some cells take
// IMC steps and others DDMC steps (transport_ddmc.cpp:135: dx_push (sigma_s + sigma_a) > tau_ddmc
// decides per cell and per step): the reference's stepdiff_smr_hybrid deck, BASELINE configs[4].
//
// Why a kernel of its own.  On such a deck ~99 % of the events are IMC steps of photons in the
// optically thin (refined) cells, and about half of the photons live in DDMC cells and take a few
// tens of steps each.  One loop body that can do either step (k_transport<.., DDMC = true>, which
// still serves per-event opacities) makes every IMC pass carry the DDMC step's registers: it
// spilled, and ran 50 % slower than the pure-IMC kernel on the same IMC work.  Here a wave runs TWO
// event loops and a lane is in one of them at a time:
//   * the IMC loop of the gray IMC kernels (imc_step_dir / imc_step_fast: cell index from the
//     nudge predicates, block crossings from the face table, next cell's mean free path requested
//     ahead), for lanes whose photon is in an IMC cell;
//   * the DDMC loop of k_ddmc_all ("virtual" state: cell, time, stream, pending leak), for lanes
//     whose photon is in a DDMC cell;
//   * a service phase for everything that happens once per history or once per change of regime:
//     loading / writing back, general relocation (level changes, other ranks), the albedo step
//     of a photon that arrives in a DDMC cell with a real position (from an IMC cell, from
//     another block, or freshly loaded next to a face: transport_utils.hpp:279-397), and the
//     re-materialisation of a DDMC photon that leaked into an IMC cell.
// Neither loop carries the other's temporaries; what a lane keeps across both is the IMC state
// (position, direction, time, stream, cell, block geometry) plus the pending leak.  A lane learns
// the regime of a cell it has just entered from the data it gathers for the next step anyway: the
// sign of lam_hyb (IMC loop) or of the record's sigma (DDMC loop), both written by k_ddmc_pack.
// Lanes of the other regime wait (parked) while a loop runs; the IMC loop is left when parked
// DDMC lanes have waited long enough (kParkBudget lane-passes) and the DDMC loop is cut short when
// IMC lanes wait (kDdmcMaxPasses).
//
// Same draws, same arithmetic, same bits as the reference's per-step choice (tests/test_gpu_parity.py
// holds it to the oracle on the 2-D / 3-D SMR hybrid decks, one rank and several); MODE picks the
// arithmetic of the IMC steps exactly as in the gray IMC kernels: 0 = exact (bit-identical to the
// oracle), 1 = lean, 2 = lean on exact geometry, 3 = lean on exact geometry in CELL-LOCAL coordinates (the
// step of k_imc_cell, jb_kernel_imc.hpp: imc_step_cell; block crossings read off the ghost cells of
// lam_hyb; the default).  DDMC steps have the exact arithmetic only.
#pragma once

#include "jb_device.hpp"

namespace jb {

#ifndef JB_HYBRID_WAVES_PER_SIMD
#define JB_HYBRID_WAVES_PER_SIMD 3
#endif
#ifndef JB_HYBRID_CELL_WAVES_PER_SIMD   // the cell-local IMC phase (MODE 3, PHASE 1) in 1-D / 2-D, see below
#define JB_HYBRID_CELL_WAVES_PER_SIMD 4
#endif
#ifndef JB_HYBRID_CELL3D_WAVES_PER_SIMD   // ... in 3-D
#define JB_HYBRID_CELL3D_WAVES_PER_SIMD 3
#endif
#ifndef JB_HYBRID_XLEAN_WAVES_PER_SIMD    // the x-space lean IMC phase (MODE 1 / 2) in 1-D / 2-D
#define JB_HYBRID_XLEAN_WAVES_PER_SIMD 4
#endif
#ifndef JB_HYBRID_REMAINDER_WAVES_PER_SIMD
#define JB_HYBRID_REMAINDER_WAVES_PER_SIMD 3
#endif
#ifndef JB_HYBRID_IMC_BUDGET      // idle lane-passes (lanes that left the IMC loop) that buy a service phase
#define JB_HYBRID_IMC_BUDGET 96
#endif
#ifndef JB_HYBRID_PARK_BUDGET     // lane-passes parked DDMC lanes wait before the DDMC loop is run
#define JB_HYBRID_PARK_BUDGET 256
#endif
#ifndef JB_HYBRID_DDMC_BUDGET     // idle lane-passes in the DDMC loop that buy a service phase
#define JB_HYBRID_DDMC_BUDGET 256
#endif
#ifndef JB_HYBRID_DDMC_MAX_PASSES  // ... and its length while IMC lanes wait
#define JB_HYBRID_DDMC_MAX_PASSES 24
#endif
#ifndef JB_HYBRID_CHUNK
#define JB_HYBRID_CHUNK 128
#endif

enum { HS_IDLE = 0, HS_IMC = 1, HS_VIRT = 2, HS_REAL = 3, HS_DONE = 4, HS_RELOC = 5, HS_EMERGE = 6,
       HS_NEW = 7, HS_PARK = 8 };

// PHASE 0: both event loops in one launch (lanes of the other regime wait in their registers).
// PHASE 1 / 2: the IMC loop only / the DDMC loop only -- a photon that turns out to belong to the
// other regime is PARKED: written back as it stands (position and direction materialised: the
// state the reference's swarm holds between two transport iterations) and its slot appended to
// park_list, which the other phase's launch then works through (list_in), until both lists stay
// empty (launch_hybrid in jb_api.hip).  Each phase is a kernel of its own to the register
// allocator, and every loop runs with full waves.
// the kernel's argument list as the kernel-argument segment holds it (natural alignment, in order)
struct HybridArgs {
  const DevMesh *Mp;
  DevParams P;
  DevSwarm S;
  double t_start, dt;
  long long first, last;
  unsigned long long *counters;
  const unsigned *list_in;
  unsigned *park_list;
  unsigned long long *park_count;
  const unsigned long long *list_count;
};

template <int NDIM, bool TALLY, bool NOABS, int MODE, int PHASE>
// (PHASE 0 keeps the state of both loops: two waves per SIMD, 256 registers; it only ever sees the
// remainder.  PHASES 1 and 2 are held to three waves per SIMD, 168 registers, without spills.)
// The exact-arithmetic IMC phase in 3-D or with an absorption opacity, and the lean one in 3-D with
// an absorption opacity, do not fit 168 registers either (parity-test and absorbing-material
// configurations; the stepdiff decks run none of them).
// Round 4: the cell-local IMC phase in 1-D / 2-D runs FOUR waves per SIMD: 128 registers cost it
// 4 - 18 dwords of scratch, all of them stored and reloaded around the event loops (none inside:
// tests/test_cabi.py), and the loop is bound by instruction issue -- BASELINE configs[4] 60.1 -> 57.0 ms;
// the x-space lean IMC phase (general geometry, MODE 1; JB_NO_IMC_CELL=1, MODE 2) likewise: 79.5 -> 73.1 ms.
__global__ void __launch_bounds__(kBlock, PHASE == 0 ? JB_HYBRID_REMAINDER_WAVES_PER_SIMD
                                          : ((PHASE == 1 && MODE == 0 && (NDIM == 3 || !NOABS)) ||
                                             (PHASE == 1 && NDIM == 3 && !NOABS)) ? 2
                                          : (PHASE == 1 && MODE == 3 && NDIM < 3) ? JB_HYBRID_CELL_WAVES_PER_SIMD
                                          : (PHASE == 1 && MODE == 3) ? JB_HYBRID_CELL3D_WAVES_PER_SIMD
                                          : (PHASE == 1 && NDIM < 3 && MODE != 0) ? JB_HYBRID_XLEAN_WAVES_PER_SIMD
                                                                      : JB_HYBRID_WAVES_PER_SIMD)
    k_hybrid(const DevMesh *__restrict__, DevParams, DevSwarm, double, double, long long, long long,
             unsigned long long *, const unsigned *, unsigned *, unsigned long long *,
             const unsigned long long *) {
  // The arguments are read where they are used, from the kernel-argument segment (scalar loads),
  // not through the parameters: fetched at kernel entry they occupy ~80 scalar registers across
  // the loops, and what spills from there is fetched back with VALU instructions (v_readlane) in
  // a loop that is bound by VALU issue (k_ddmc_all, section 4.2 of DESIGN.md, does the same).  The
  // mesh view (~120 dwords) is behind a pointer for the same reason.
  const HybridArgs &A = *(const HybridArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const DevMesh &M = *A.Mp;
  const DevParams &P = A.P;
  const DevSwarm &S = A.S;
  const double t_start = A.t_start, dt = A.dt;
  const long long first = A.first;
  unsigned long long *const counters = g1(A.counters);
  const unsigned *const list_in = g1(A.list_in);
  unsigned *const park_list = g1(A.park_list);
  unsigned long long *const park_count = g1(A.park_count);
  // (list_count: the length of list_in where only the device knows it -- the photons k_ddmc_all
  // hands over -- ; the launch then names the list's capacity)
  long long last = A.last;
  if (A.list_count != nullptr) {
    const long long have = first + (long long)*g1(A.list_count);
    last = have < last ? have : last;
    // (nothing handed over -- all but one cycle in a hundred: the whole launch leaves before it fills its tables;
    // ~45 us of every all-DDMC cycle otherwise)
    if (last <= first) return;
  }
  __shared__ double lds_tally[TALLY ? kLdsTally : 1];
  const bool tally_in_lds = TALLY && (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  if constexpr (TALLY) {
    if (tally_in_lds)
      for (int q = threadIdx.x; q < M.nblocks * (int)M.ntot; q += blockDim.x) lds_tally[q] = 0.0;
  }
  __shared__ LdsBlockTable lds_blocks;
  fill_block_table(M, lds_blocks);
  load_math_tables();  // (ends with a barrier)
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  constexpr bool kLean = MODE != 0, kExactG = MODE >= 2, kCell = MODE == 3;
  constexpr long long kChunk = JB_HYBRID_CHUNK;
  const double vv = P.c;
  const double t_end = t_start + dt;
  const int lane = threadIdx.x & 63;
  unsigned long long *queue = counters + CNT_QUEUE;
  const long long per_q = (last - first + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues, tried = 0;
  bool more = true;
  long long chunk_pos = 0, chunk_end = 0;

  // wave-level counters (scalar registers): finished histories by outcome; events = stepping lanes
  // summed over the passes of both loops + steps taken in the service phase
  unsigned int c_census = 0, c_abs = 0, c_esc = 0, c_out = 0;
  unsigned long long c_ev = 0;
  unsigned int c_pass = 0, c_service = 0;

  // ---- lane state
  int ls = HS_IDLE;
  long long n = 0;
  LcgRng rng(0);
  int b = 0, ip = 0, jp = 0, kp = 0, status = ST_ACTIVE;
  // time; for an HS_IMC lane of a lean kernel: the distance left to census c (t_end - t), and
  // (vx, vy, vz) the unit direction (in_dir)
  double t = 0.0;
  double x = 0.0, y = 0.0, z = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
  bool in_dir = false;
  bool real_pos = false;  // x, y, z hold the photon's position (not so for a virtual DDMC lane)
  // channel of the last DDMC leak (0..5) while its direction is deferred: its two uniforms then
  // live in (vx, vy) -- the direction they stand for replaces a stale one; -1: the direction is
  // the one in vx, vy, vz
  int pend = -1;
  bool resample = false;  // reached census in a DDMC step: position / direction to be resampled
  bool fresh = false;     // loaded and not touched since: parking it needs no write-back
  // IMC lanes: geometry of the lane's block (index-0 coordinate, cell width, nudge width per axis)
  // and the mean free paths of its cell (lam_cur < 0: the cell takes DDMC steps)
  // (bound when the IMC loop is entered: nothing of it is live in the service phase or the DDMC loop)
  DirGeom g;
  double dxp = 0.0, lam_cur = 0.0, lam_a_cur = 0.0;
  // MODE 3: while in_local, (x, y, z) hold the position relative to the centre of the photon's cell,
  // qoff the byte offset of that cell in lam_hyb (8 (b ntot + cell)); ip, jp, kp are stale
  bool in_local = false;
  unsigned qoff = 0u;
  CellGeom cg{0, 0, 0, 0, 0, 0, 0};

  const unsigned ntot_u = sgpr_copy((unsigned)M.ntot);
  const double *rec_base = sgpr_copy_ptr(M.ddmc_base);
  const double *hyb_base = sgpr_copy_ptr(M.lam_hyb);
  const int l_ni = (int)sgpr_copy((unsigned)M.ni), l_nj = (int)sgpr_copy((unsigned)M.nj);
  const int l_sy = (int)sgpr_copy(8u * (unsigned)M.ni), l_sz = (int)sgpr_copy(8u * (unsigned)(M.ni * M.nj));
  const double inv_ni = 1.0 / (double)M.ni, inv_ninj = 1.0 / (double)(M.ni * M.nj);
  const int l_is = (int)sgpr_copy((unsigned)M.is), l_ie = (int)sgpr_copy((unsigned)M.ie);
  const int l_js = (int)sgpr_copy((unsigned)M.js), l_je = (int)sgpr_copy((unsigned)M.je);
  const int l_ks = (int)sgpr_copy((unsigned)M.ks), l_ke = (int)sgpr_copy((unsigned)M.ke);
  auto cidx_l = [&](int k, int j, int i) { return mad24(mad24(k, l_nj, j), l_ni, i); };
  auto on_block_l = [&](int i, int j, int k) {
    return i >= l_is && i <= l_ie && j >= l_js && j <= l_je && k >= l_ks && k <= l_ke;
  };
  auto cell_word = [&](int blk, int q) {  // index of cell q of block blk in the per-cell arrays
    return (unsigned long long)(unsigned)blk * ntot_u + (unsigned)q;
  };
  auto faces_of = [&](Step &s, const Blk &Bq, int i, int j, int k) {  // transport.cpp:114-119
    s.xl = xc(Bq, 0, i) - 0.5 * Bq.dx[0]; s.xu = xc(Bq, 0, i) + 0.5 * Bq.dx[0];
    s.yl = xc(Bq, 1, j) - 0.5 * Bq.dx[1]; s.yu = xc(Bq, 1, j) + 0.5 * Bq.dx[1];
    s.zl = xc(Bq, 2, k) - 0.5 * Bq.dx[2]; s.zu = xc(Bq, 2, k) + 0.5 * Bq.dx[2];
  };
  // the 64-byte record of a cell {f sigma_a, +-(sigma_a + sigma_s), six leak opacities}; the sign
  // of the second entry is negative in a cell that takes IMC steps (k_ddmc_pack)
  auto load_record = [&](Step &s, int blk, int q) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef const v4d __attribute__((address_space(1))) *grec;
    const grec rec = (grec)((gcptr)rec_base + 8 * cell_word(blk, q));
    const v4d r0 = rec[0];
    const v4d r1 = rec[1];
    s.ffaa = r0.x; s.sig = r0.y;
    s.Px_l = r0.z; s.Px_u = r0.w; s.Py_l = r1.x; s.Py_u = r1.y; s.Pz_l = r1.z; s.Pz_u = r1.w;
  };
  // (MODE 2: the per-cell arrays of all resident blocks span < 4 GiB -- jb_mesh_create -- and a
  // lane keeps the byte offset of its block's part, as the EXACT gray IMC kernels do)
  unsigned hyb_off = 0u;
  auto fetch_lam = [&]() {  // the IMC loop's gather for the lane's cell
    if constexpr (kCell) {
      lam_cur = *(gcptr)((const char *)hyb_base + qoff);
      if constexpr (!NOABS) lam_a_cur = *(gcptr)((const char *)M.lam_base + (qoff + hyb_off));
    } else if constexpr (kExactG) {
      const unsigned off = ((unsigned)cidx_l(kp, jp, ip) << 3) + hyb_off;
      lam_cur = *(gcptr)((const char *)hyb_base + off);
      if constexpr (!NOABS) lam_a_cur = *(gcptr)((const char *)M.lam_base + (off + hyb_off));
    } else {
      const unsigned long long w = cell_word(b, cidx_l(kp, jp, ip));
      lam_cur = ((gcptr)hyb_base)[w];
      if constexpr (!NOABS) lam_a_cur = ((gcptr)M.lam_base)[2ull * (unsigned)b * ntot_u + (unsigned)cidx_l(kp, jp, ip)];
    }
  };
  // a photon with a real position in a DDMC cell enters the DDMC loop unless the albedo step
  // would find it at a face of its cell (k_ddmc_all)
  auto enter_ddmc = [&](const Blk &Bq) {
    Step s;
    faces_of(s, Bq, ip, jp, kp);
    s.x = x; s.y = y; s.z = z;
    if (at_cell_face<NDIM>(s)) {
      ls = HS_REAL;
    } else {
      real_pos = false;
      ls = HS_VIRT;
    }
  };
  // ... and one in an IMC cell enters the IMC loop: lean units
  auto bind_geometry = [&]() {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      g.dx[d] = lds_blocks.dx[b][d];
      g.x0[d] = lds_x0(M, lds_blocks, b, d);
      g.fd[d] = kEpsImc * g.dx[d];
    }
    dxp = dmin(g.dx[0], dmin(g.dx[1], g.dx[2]));
    hyb_off = (unsigned)b * (ntot_u * 8u);
  };
  // MODE 3: into the IMC loop's coordinates and back (centre of cell idx: x0 + (idx + 0.5) dx, exact on
  // this geometry)
  auto to_local = [&]() {
    const double d0 = lds_blocks.dx[b][0], d1 = lds_blocks.dx[b][1], d2 = lds_blocks.dx[b][2];
    cg.hx = 0.5 * d0; cg.hy = 0.5 * d1; cg.hz = 0.5 * d2;
    cg.mx = cg.hx - kEpsImc * d0; cg.my = cg.hy - kEpsImc * d1; cg.mz = cg.hz - kEpsImc * d2;
    cg.dxp = dmin(d0, dmin(d1, d2));
    x -= fma((double)ip + 0.5, d0, lds_x0(M, lds_blocks, b, 0));
    if (multi_d) y -= fma((double)jp + 0.5, d1, lds_x0(M, lds_blocks, b, 1));
    if (three_d) z -= fma((double)kp + 0.5, d2, lds_x0(M, lds_blocks, b, 2));
    hyb_off = (unsigned)b * (ntot_u * 8u);
    qoff = hyb_off + ((unsigned)cidx_l(kp, jp, ip) << 3);
    in_local = true;
  };
  auto from_local = [&]() {
    const int q = (int)((qoff - hyb_off) >> 3);
    kp = three_d ? (int)(((double)q + 0.5) * inv_ninj) : 0;
    const int r = q - kp * (l_ni * l_nj);
    jp = multi_d ? (int)(((double)r + 0.5) * inv_ni) : 0;
    ip = r - jp * l_ni;
    x += fma((double)ip + 0.5, lds_blocks.dx[b][0], lds_x0(M, lds_blocks, b, 0));
    if (multi_d) y += fma((double)jp + 0.5, lds_blocks.dx[b][1], lds_x0(M, lds_blocks, b, 1));
    if (three_d) z += fma((double)kp + 0.5, lds_blocks.dx[b][2], lds_x0(M, lds_blocks, b, 2));
    in_local = false;
  };
  auto enter_imc = [&]() {
    fresh = false;
    if constexpr (kLean) {
      t = vv * (t_end - t);
      vx *= P.rc; vy *= P.rc; vz *= P.rc;
      in_dir = true;
    }
    ls = HS_IMC;
  };
  // one face of the lane's block crossed by an IMC step into a resident block of the same size, a
  // periodic wrap or a reflection: served from the face table (k_transport's cross_face)
  auto cross_face = [&](auto axis_c, bool up, double &pos, double &vel, int &idx, int first_i,
                        int last_i) -> bool {
    constexpr int AXIS = decltype(axis_c)::value;
    const int ent = lds_blocks.nbr_ent[b][2 * AXIS + (int)up];
    if (ent < 0) return false;
    const int kind = ent >> 28;
    bool at_first = up;
    if (kind != 0) {
      const double lo = M.gmin[AXIS], hi = M.gmax[AXIS];
      if (kind == 1) {  // periodic
        pos = up ? lo + (pos - hi) : hi - (lo - pos);
      } else {  // reflecting (boundaries.hpp:46-82): back into the same block
        pos = up ? hi - (pos - hi) : lo + (lo - pos);
        vel = -vel;
        at_first = !up;
      }
    }
    b = ent & 0x0fffffff;
    g.x0[AXIS] = lds_x0(M, lds_blocks, b, AXIS);
    hyb_off = (unsigned)b * (ntot_u * 8u);
    idx = at_first ? first_i : last_i;
    if (!((kLean ? t > 0.0 : t < t_end))) ls = HS_DONE;
    else fetch_lam();
    return true;
  };

  int park_waste = 0;  // wave-level: lane-passes parked DDMC lanes have waited since the last DDMC loop

#ifdef JB_HYB_STATS  // diagnostic build: lanes entering the service phase by state, loop passes by kind
  unsigned int st_cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st_imc_pass = 0, st_ddmc_pass = 0, st_imc_lanes = 0;
#endif
  for (;;) {
    // ================================ SERVICE ================================
    ++c_service;
#ifdef JB_HYB_STATS
    for (int k = 0; k < 9; ++k) st_cnt[k] += (unsigned)__popcll(__ballot(ls == k));
#endif
    // lanes that left the IMC loop of a lean kernel: back to time and velocity (census: the
    // distance left is exactly 0, t = t_end)
    if constexpr (kCell) {
      if (in_local && ls != HS_IMC) from_local();
    }
    if constexpr (kLean) {
      if (in_dir && ls != HS_IMC) {
        t = fma(-t, P.rc, t_end);
        vx *= vv; vy *= vv; vz *= vv;
        in_dir = false;
      }
    }
    // -- 1. block crossings: the comm phase of the reference for one particle in flight
    if (ls == HS_RELOC) {
      fresh = false;
      Blk Bo;
      load_block_lds(M, lds_blocks, b, Bo);
      Step s;
      s.vv = vv; s.pend = pend; s.pz1 = vx; s.pz2 = vy;
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
      }
      // transport_ddmc.cpp:203-211: zero velocity flags a DDMC leak for SampleDDMCBlockFace
      // (multi-D); in 1-D the direction travels with the particle
      if (pend >= 0) {
        if constexpr (multi_d) {
          vx = 0.0; vy = 0.0; vz = 0.0;
        } else {
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = s.vx; vy = s.vy; vz = s.vz;
        }
        pend = -1;
      }
      real_pos = true;
      if (!apply_swarm_bcs<NDIM>(M, x, y, z, vx, vy, vz)) {
        status = ST_ESCAPED;
        ls = HS_DONE;
      } else {
        const int gb = find_block<NDIM>(M, x, y, z);
        const int li = M.local_index[gb];
        if (li < 0) {  // not resident here: hand the particle to the block's owner
          status = ST_OUTGOING;
          b = gb;  // global id travels in blk
          ls = HS_DONE;
        } else {
          b = li;
          Blk Bn;
          load_block_lds(M, lds_blocks, b, Bn);
          if constexpr (multi_d)
            sample_block_face<NDIM>(M, P, Bn, b, rng, x, y, z, vx, vy, vz, ip, jp, kp);
          xtoijk<NDIM>(M, Bn, x, y, z, ip, jp, kp);
          ls = HS_NEW;
        }
      }
    }
    // -- 2. a DDMC photon that leaked into an IMC cell of its block: position and direction as
    //       the step function gave them (transport_utils.hpp:209-263), from the cell it left
    if (ls == HS_EMERGE) {
      fresh = false;
      Blk Bo;
      load_block_lds(M, lds_blocks, b, Bo);
      Step s;
      s.vv = vv; s.pend = pend; s.pz1 = vx; s.pz2 = vy;
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
      s.vx = vx; s.vy = vy; s.vz = vz;
      materialise_dir(s);  // (an IMC step reads the direction)
      vx = s.vx; vy = s.vy; vz = s.vz;
      pend = -1;
      real_pos = true;
      ls = HS_NEW;
    }
    // -- 3a. finished particles: census resampling, write-back, tally
    const bool was_done = ls == HS_DONE;
    if (ls == HS_DONE) {
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
        } else if (!real_pos) {
          // absorbed in the virtual state: the albedo step left it at the cell centre
          // (transport_utils.hpp:392-396), with the direction of its last leak
          x = 0.5 * (s.xl + s.xu); y = 0.5 * (s.yl + s.yu); z = 0.5 * (s.zl + s.zu);
        }
        if (pend >= 0) {
          s.pend = pend; s.pz1 = vx; s.pz2 = vy;
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = s.vx; vy = s.vy; vz = s.vz;
          pend = -1;
        }
        if ((status == ST_ACTIVE || status == ST_OUTGOING_ABSORBED) && lds_blocks.owned[b] == 0) {
          if (status == ST_ACTIVE) status = ST_OUTGOING;  // the owner of the block tallies it
          b = M.gid[b];
        } else if (status == ST_ACTIVE) {
          if constexpr (TALLY) {  // jaybenne.cpp:547-561
            const double dv = Bd.dx[0] * Bd.dx[1] * Bd.dx[2];
            if (tally_in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + cidx(M, kp, jp, ip)], g1(S.w)[n] / dv);
            else atomicAdd(&lds_blocks.tally[b][cidx(M, kp, jp, ip)], g1(S.w)[n] / dv);
          }
        }
      }
      g1(S.blk)[n] = b;
      g1(S.t)[n] = t;
      g1(S.x)[n] = x; g1(S.y)[n] = y; g1(S.z)[n] = z;
      g1(S.vx)[n] = vx; g1(S.vy)[n] = vy; g1(S.vz)[n] = vz;
      g1(S.ip)[n] = ip; g1(S.jp)[n] = jp; g1(S.kp)[n] = kp;
      g1(S.status)[n] = status;
      g1(S.rng)[n] = rng.s;
      resample = false;
      ls = HS_IDLE;
    }
    {  // (wave-level counters are updated outside divergent branches)
      const int n_done = __popcll(__ballot(was_done));
      const int n_census = __popcll(__ballot(was_done && status == ST_ACTIVE));
      const int n_abs = __popcll(__ballot(was_done && status == ST_ABSORBED));
      const int n_esc = __popcll(__ballot(was_done && status == ST_ESCAPED));
      c_census += n_census; c_abs += n_abs; c_esc += n_esc;
      c_out += n_done - n_census - n_abs - n_esc;
    }
    // -- 3b. every idle lane claims the next slot of the wave's chunk (chunks of consecutive slots,
    //        one atomic per chunk)
    long long cand = -1;
    int st_in = ST_ABSORBED, b_in = 0;
    unsigned long long rng_in = 0ull;
    double t_in = 0.0, x_in = 0.0, y_in = 0.0, z_in = 0.0, vx_in = 0.0, vy_in = 0.0, vz_in = 0.0;
    {
      unsigned long long need = __ballot(ls == HS_IDLE);
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
          if (list_in != nullptr) cand = (long long)list_in[cand];  // (uniform)
          st_in = g1(S.status)[cand];
          rng_in = g1(S.rng)[cand];
          b_in = g1(S.blk)[cand];
          t_in = g1(S.t)[cand]; x_in = g1(S.x)[cand]; y_in = g1(S.y)[cand]; z_in = g1(S.z)[cand];
          vx_in = g1(S.vx)[cand]; vy_in = g1(S.vy)[cand]; vz_in = g1(S.vz)[cand];
        }
        chunk_pos += give;
        need &= ~__ballot(mine);
      }
    }
    // -- 3c. the lanes that claimed a slot in 3a take their new particle
    if (ls == HS_IDLE && cand >= 0 && st_in == ST_ACTIVE) {
      n = cand;
      rng.s = rng_in;
      b = b_in;
      t = t_in;
      x = x_in; y = y_in; z = z_in; vx = vx_in; vy = vy_in; vz = vz_in;
      status = ST_ACTIVE;
      resample = false;
      pend = -1;
      real_pos = true;
      in_dir = false;
      fresh = true;
      Blk Bn;
      load_block_lds(M, lds_blocks, b, Bn);
      xtoijk<NDIM>(M, Bn, x, y, z, ip, jp, kp);  // transport.cpp:96
      ls = HS_NEW;
    }
    // -- 4. one step with the real position in a DDMC cell (arrival from an IMC cell or another
    //       block, a particle loaded next to a face, the steps after an albedo rejection): the
    //       general step functions (transport_ddmc.cpp:137-179)
    c_ev += (unsigned)__popcll(__ballot(ls == HS_REAL));
    if (ls == HS_REAL) {
      fresh = false;
      Blk Br;
      load_block_lds(M, lds_blocks, b, Br);
      Step s;
      s.t_start = t_start; s.dt = dt; s.vv = vv; s.rvv = P.rc; s.dx_push = Br.dx_push;
      faces_of(s, Br, ip, jp, kp);
      s.t = t; s.x = x; s.y = y; s.z = z; s.vx = vx; s.vy = vy; s.vz = vz;
      s.ip = ip; s.jp = jp; s.kp = kp;
      s.is_absorbed = false; s.is_scattered = false; s.is_rejected = false;
      s.pend = pend; s.pz1 = vx; s.pz2 = vy;
      load_record(s, b, cidx_l(kp, jp, ip));
      ptcl_ddmc_albedo<NDIM, true>(s, rng);
      bool census = false;
      if (!s.is_rejected) census = ddmc_step_event<NDIM, true, true>(s, rng);
      t = s.t; x = s.x; y = s.y; z = s.z; vz = s.vz;
      pend = s.pend;
      vx = pend >= 0 ? s.pz1 : s.vx;  // (a deferred leak: its uniforms instead of a direction)
      vy = pend >= 0 ? s.pz2 : s.vy;
      xtoijk<NDIM>(M, Br, x, y, z, ip, jp, kp);  // transport.cpp:146
      if (!s.is_rejected) resample = census;  // (the flag of the last DDMC step)
      real_pos = true;
      if (!on_block_l(ip, jp, kp)) {
        // (a rejected particle keeps its direction: pend < 0; a leak is flagged in step 1)
        ls = HS_RELOC;
      } else if (s.is_absorbed) {  // transport.cpp:157-163
        if (lds_blocks.owned[b] != 0) {
          atomicAdd(&M.edelta[b][cidx_l(kp, jp, ip)], g1(S.w)[n]);
          status = ST_ABSORBED;
        } else {
          status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
        }
        ls = HS_DONE;
      } else if (!(t < t_end)) {  // census
        ls = HS_DONE;
      } else if (!s.is_rejected) {
        if constexpr (PHASE == 1) {
          ls = HS_NEW;       // leaked into a neighbouring cell of this block (x, y, z: where to)
        } else {
          real_pos = false;  // ... : virtual from here on
          ls = HS_VIRT;      // (the DDMC loop finds out whether that cell takes DDMC steps)
        }
      } else {
        ls = HS_NEW;       // rejected: a real position in the cell on the other side of the face
      }
    }
    // -- 5. a real position in a cell of a resident block: which loop?
    if (ls == HS_NEW) {
      if (!(t < t_end)) {
        ls = HS_DONE;  // already at census: nothing to track
      } else {
        const double lam = ((gcptr)hyb_base)[cell_word(b, cidx_l(kp, jp, ip))];
        if (__double2hiint(lam) < 0) {
          if constexpr (PHASE == 1) {
            ls = HS_PARK;
          } else {
            Blk Bn;
            load_block_lds(M, lds_blocks, b, Bn);
            enter_ddmc(Bn);
          }
        } else {
          if constexpr (PHASE == 2) {
            ls = HS_PARK;
          } else {
            if (pend >= 0) {  // (a DDMC step of the service phase leaked straight into this cell)
              Step s;
              s.vv = vv; s.pend = pend; s.pz1 = vx; s.pz2 = vy;
              s.vx = vx; s.vy = vy; s.vz = vz;
              materialise_dir(s);
              vx = s.vx; vy = s.vy; vz = s.vz;
              pend = -1;
            }
            enter_imc();
          }
        }
      }
    }
    // -- 6. a photon of the other phase's regime: written back as it stands, its slot appended
    //       to the list the other phase works through
    if constexpr (PHASE != 0) {
      if (ls == HS_PARK) {
        if (pend >= 0) {
          Step s;
          s.vv = vv; s.pend = pend; s.pz1 = vx; s.pz2 = vy;
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = s.vx; vy = s.vy; vz = s.vz;
          pend = -1;
        }
        if (!fresh) {  // (a photon parked as it was loaded -- e.g. one that starts the cycle in a DDMC
                       // cell, seen by the IMC phase -- is only listed)
          g1(S.blk)[n] = b;
          g1(S.t)[n] = t;
          g1(S.x)[n] = x; g1(S.y)[n] = y; g1(S.z)[n] = z;
          g1(S.vx)[n] = vx; g1(S.vy)[n] = vy; g1(S.vz)[n] = vz;
          g1(S.ip)[n] = ip; g1(S.jp)[n] = jp; g1(S.kp)[n] = kp;
          g1(S.rng)[n] = rng.s;
        }
        const unsigned long long pm = __ballot(true);
        const int leader = __ffsll((long long)pm) - 1;
        unsigned long long base = 0ull;
        if (lane == leader) base = atomicAdd(park_count, (unsigned long long)__popcll(pm));
        base = __shfl(base, leader, 64);
        park_list[base + (unsigned long long)__popcll(pm & ((1ull << lane) - 1ull))] = (unsigned)n;
        ls = HS_IDLE;
      }
    }
    // lanes that still need the service phase are served before a loop is entered
    if (__ballot(ls == HS_REAL || ls == HS_DONE || ls == HS_RELOC || ls == HS_EMERGE || ls == HS_NEW) != 0ull)
      continue;
    const int n_imc = __popcll(__ballot(ls == HS_IMC));
    const int n_virt = __popcll(__ballot(ls == HS_VIRT));
    if (n_imc == 0 && n_virt == 0) {
      if (more) continue;
      break;
    }

    // ================================ DDMC EVENTS =============================
    // (run when the parked lanes have waited long enough, or nothing else can run)
    if (PHASE != 1 && n_virt > 0 && (n_imc == 0 || park_waste >= P.hyb_park_budget)) {
      park_waste = 0;
      int waste = 0, passes = 0;
      int nrun = n_virt;
      int thresh = 1;
      while (nrun >= thresh) {
        Step s;
        if (ls == HS_VIRT) {
          s.t_start = t_start; s.dt = dt; s.vv = vv;
          s.t = t;
          s.ip = ip; s.jp = jp; s.kp = kp;
          s.is_absorbed = false;
          s.pend = pend; s.pz1 = vx; s.pz2 = vy;
          s.xl = s.yl = s.zl = 0.0; s.xu = s.yu = s.zu = 1.0;
          load_record(s, b, cidx_l(kp, jp, ip));
          // the cell the last leak led into takes IMC steps: out of this loop
          if (__double2hiint(s.sig) < 0) ls = HS_EMERGE;
        }
        const int stepping = __popcll(__ballot(ls == HS_VIRT));
#ifdef JB_HYB_STATS
        ++st_ddmc_pass;
#endif
        ++c_pass;
        c_ev += (unsigned int)stepping;
        if (ls == HS_VIRT) {
          const bool census = ddmc_step_event<NDIM, true, true>(s, rng);
          t = s.t;
          ip = s.ip; jp = s.jp; kp = s.kp;  // = Xtoijk of the position the step gives
          pend = s.pend;
          if (pend >= 0) { vx = s.pz1; vy = s.pz2; }
          resample = census;
          if (!on_block_l(ip, jp, kp)) {
            ls = HS_RELOC;  // a leak through a block face: the service phase
          } else if (s.is_absorbed) {  // transport.cpp:157-163
            if (lds_blocks.owned[b] != 0) {
              atomicAdd(&M.edelta[b][cidx_l(kp, jp, ip)], g1(S.w)[n]);
              status = ST_ABSORBED;
            } else {
              status = ST_OUTGOING_ABSORBED;
            }
            ls = HS_DONE;
          } else if (!(t < t_end)) {  // census
            ls = HS_DONE;
          }
        }
        nrun = __popcll(__ballot(ls == HS_VIRT));
        waste += n_virt - nrun;
        ++passes;
        if (waste >= JB_HYBRID_DDMC_BUDGET || (n_imc > 0 && passes >= JB_HYBRID_DDMC_MAX_PASSES)) thresh = 65;
      }
      continue;  // (service: the lanes that left the loop)
    }

    // ================================ IMC EVENTS ==============================
    if constexpr (PHASE != 2) {
      if (ls == HS_IMC) {
        if constexpr (kCell) {
          if (!in_local) to_local();
        } else {
          bind_geometry();
        }
        fetch_lam();
      }
      int waste = 0;
      int nrun = n_imc;
      int thresh = 1;
      while (nrun >= thresh) {
       if constexpr (kCell) {
        // ---- the step of k_imc_cell.  A negative datum: the cell entered in the last pass takes DDMC
        // steps (out of this loop; albedo: service), or it is a ghost cell of the block (k_lam_ghost_codes:
        // the crossing from its code, or the general relocation of the service phase); the lane sits
        // this pass out either way
        const unsigned long long neg_m =
            __builtin_amdgcn_ballot_w64(ls == HS_IMC) & __builtin_amdgcn_ballot_w64(__double2hiint(lam_cur) < 0);
        const bool neg = ls == HS_IMC && __double2hiint(lam_cur) < 0;
        const bool stepping = ls == HS_IMC && !neg;
        ++c_pass;
        c_ev += (unsigned int)(nrun - __popcll(neg_m));
#ifdef JB_HYB_STATS
        ++st_imc_pass; st_imc_lanes += nrun - __popcll(neg_m);
#endif
        if (neg_m != 0ull) {
          if (neg) {
            const int code = __double2hiint(lam_cur);
            if ((code & 0x7ff00000) != (kGhostHi & 0x7ff00000)) {
              ls = HS_REAL;                      // a DDMC cell of this block
            } else if ((code & kGhostTable) == 0) {
              ls = HS_RELOC;                     // level change, another rank's block, outflow, a corner
            } else {
              const int fx = (code << 15) & (int)0x80000000u, fy = (code << 14) & (int)0x80000000u,
                        fz = (code << 13) & (int)0x80000000u;
              x = __hiloint2double(__double2hiint(x) ^ fx, __double2loint(x));
              vx = __hiloint2double(__double2hiint(vx) ^ fx, __double2loint(vx));
              if (multi_d) {
                y = __hiloint2double(__double2hiint(y) ^ fy, __double2loint(y));
                vy = __hiloint2double(__double2hiint(vy) ^ fy, __double2loint(vy));
              }
              if (three_d) {
                z = __hiloint2double(__double2hiint(z) ^ fz, __double2loint(z));
                vz = __hiloint2double(__double2hiint(vz) ^ fz, __double2loint(vz));
              }
              qoff = (unsigned)__double2loint(lam_cur);
              if (code & (kGhostCoarser | kGhostFiner))  // a resident block one level up or down
                qoff = cross_level<NDIM>(code, qoff, l_sy, l_sz, cg, x, y, z);
              b = code & 0xff;
              hyb_off = (unsigned)b * (ntot_u * 8u);
              if (!(t > 0.0)) ls = HS_DONE;      // (reached census and a block face in one step)
              else fetch_lam();
            }
          }
        }
        if (stepping) {
          bool is_absorbed, is_scattered, hit_any;
          imc_step_cell<NDIM, NOABS>(cg, l_sy, l_sz, lam_a_cur, lam_cur, rng, t, x, y, z, vx, vy, vz, qoff,
                                     is_absorbed, is_scattered, hit_any);
          fetch_lam();  // (for the next pass, ahead of the scatter)
          const bool census = !(t > 0.0);
          bool collide = is_absorbed || is_scattered;
          bool off = false;
          const bool at_face = hit_any && (collide || census);
          if (__builtin_amdgcn_ballot_w64(at_face) != 0ull) {
            // a collision or the census within eps of a cell face: if that face is a block face the
            // reference relocates the photon first and forgets the collision (transport.cpp:149-155)
            asm volatile("; collision or census next to a cell face" ::: "memory");
            if (at_face)
              off = __double2hiint(lam_cur) < 0 && (__double2hiint(lam_cur) & 0x7ff00000) == (kGhostHi & 0x7ff00000);
          }
          collide = collide && !off;
          if (!NOABS && is_absorbed && collide) {  // transport.cpp:157-163
            if (lds_blocks.owned[b] != 0) {
              atomicAdd(&M.edelta[b][(qoff - hyb_off) >> 3], g1(S.w)[n]);
              status = ST_ABSORBED;
            } else {
              status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
            }
            ls = HS_DONE;
          } else {
            if (is_scattered && collide) scatter_dir<true>(rng, vx, vy, vz);  // transport.cpp:165-170
            if (census && !off) ls = HS_DONE;
          }
        }
       } else {
        // the cell entered in the last pass takes DDMC steps: out of this loop (albedo: service)
        if (ls == HS_IMC && __double2hiint(lam_cur) < 0) ls = HS_REAL;
        const int stepping = __popcll(__ballot(ls == HS_IMC));
        ++c_pass;
        c_ev += (unsigned int)stepping;
#ifdef JB_HYB_STATS
        ++st_imc_pass; st_imc_lanes += stepping;
#endif
        bool crossing = false;  // left its block in this pass
        if (ls == HS_IMC) {
          bool is_absorbed, is_scattered;
          if constexpr (kLean) {
            imc_step_dir<NDIM, NOABS, kExactG, !kExactG>(g, dxp, lam_a_cur, lam_cur, rng, t, x, y, z, vx, vy, vz,
                                               ip, jp, kp, is_absorbed, is_scattered);
          } else {
            ImcCell c;
            if (M.exact) {  // (uniform) the same doubles in 3 instead of 8 operations per axis
              c.xl = m_fma((double)ip, g.dx[0], g.x0[0]); c.xu = c.xl + g.dx[0];
              c.yl = m_fma((double)jp, g.dx[1], g.x0[1]); c.yu = c.yl + g.dx[1];
              c.zl = m_fma((double)kp, g.dx[2], g.x0[2]); c.zu = c.zl + g.dx[2];
              c.fdx = g.fd[0]; c.fdy = g.fd[1]; c.fdz = g.fd[2];
            } else {  // transport.cpp:114-119, transport_utils.hpp:151-153
              const double xcx = g.x0[0] + ((double)ip + 0.5) * g.dx[0];
              const double xcy = g.x0[1] + ((double)jp + 0.5) * g.dx[1];
              const double xcz = g.x0[2] + ((double)kp + 0.5) * g.dx[2];
              c.xl = xcx - 0.5 * g.dx[0]; c.xu = xcx + 0.5 * g.dx[0];
              c.yl = xcy - 0.5 * g.dx[1]; c.yu = xcy + 0.5 * g.dx[1];
              c.zl = xcz - 0.5 * g.dx[2]; c.zu = xcz + 0.5 * g.dx[2];
              c.fdx = kEpsImc * (c.xu - c.xl); c.fdy = kEpsImc * (c.yu - c.yl);
              c.fdz = kEpsImc * (c.zu - c.zl);
            }
            imc_step_fast<NDIM, NOABS, false>(c, vv, P.rc, t_end, dxp, lam_a_cur, lam_cur, rng, t, x, y,
                                              z, vx, vy, vz, ip, jp, kp, is_absorbed, is_scattered);
          }
          if (!on_block_l(ip, jp, kp)) {
            crossing = true;
          } else if (is_absorbed) {  // transport.cpp:157-163
            if (lds_blocks.owned[b] != 0) {
              atomicAdd(&M.edelta[b][cidx_l(kp, jp, ip)], g1(S.w)[n]);
              status = ST_ABSORBED;
            } else {
              status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
            }
            ls = HS_DONE;
          } else {
            fetch_lam();  // (for the next pass, ahead of the scatter)
            if constexpr (kLean) {
              if (is_scattered) scatter_dir(rng, vx, vy, vz);
              if (!(t > 0.0)) ls = HS_DONE;
            } else {
              if (is_scattered) scatter(rng, vv, vx, vy, vz);  // transport.cpp:165-170
              if (!(t < t_end)) ls = HS_DONE;                  // census
            }
          }
        }
        // one face crossed into a resident block of the same size, a periodic wrap, a reflection:
        // from the face table, at once; everything else: the service phase (HS_RELOC stays)
        if (__ballot(crossing) != 0ull) {
          if (crossing) {
            const bool xo = ip < l_is || ip > l_ie;
            const bool yo = multi_d && (jp < l_js || jp > l_je);
            const bool zo = three_d && (kp < l_ks || kp > l_ke);
            bool done = false;
            if (xo && !yo && !zo) done = cross_face(std::integral_constant<int, 0>{}, ip > l_ie, x, vx, ip, l_is, l_ie);
            else if (yo && !xo && !zo) done = cross_face(std::integral_constant<int, 1>{}, jp > l_je, y, vy, jp, l_js, l_je);
            else if (zo && !xo && !yo) done = cross_face(std::integral_constant<int, 2>{}, kp > l_ke, z, vz, kp, l_ks, l_ke);
            if (!done) ls = HS_RELOC;
          }
        }
       }
        nrun = __popcll(__ballot(ls == HS_IMC));
        waste += n_imc - nrun;
        park_waste += n_virt;
        if (waste >= P.hyb_imc_budget || (n_virt > 0 && park_waste >= P.hyb_park_budget)) thresh = 65;
      }
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
  const unsigned long long r_census = c_census, r_abs = c_abs, r_esc = c_esc, r_out = c_out, r_ev = c_ev;
  if (lane == 0) {
    if (r_census) atomicAdd(&counters[CNT_CENSUS], r_census);
    if (r_abs) atomicAdd(&counters[CNT_ABSORBED], r_abs);
    if (r_esc) atomicAdd(&counters[CNT_ESCAPED], r_esc);
    if (r_out) atomicAdd(&counters[CNT_OUTGOING], r_out);
    if (r_ev) atomicAdd(&counters[CNT_EVENTS], r_ev);
    atomicAdd(&counters[CNT_PASSES], (unsigned long long)c_pass);
    atomicAdd(&counters[CNT_SERVICE], (unsigned long long)c_service);
#ifdef JB_HYB_STATS
    for (int k = 0; k < 9; ++k) atomicAdd(&counters[64 + 16 * PHASE + k], (unsigned long long)st_cnt[k]);
    atomicAdd(&counters[64 + 16 * PHASE + 9], (unsigned long long)st_imc_pass);
    atomicAdd(&counters[64 + 16 * PHASE + 10], (unsigned long long)st_ddmc_pass);
    atomicAdd(&counters[64 + 16 * PHASE + 11], (unsigned long long)st_imc_lanes);
    atomicAdd(&counters[64 + 16 * PHASE + 12], (unsigned long long)c_service);
#endif
  }
}

}  // namespace jb
