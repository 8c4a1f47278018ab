// jb_kernel_imc.hpp -- TransportPhotons (transport.cpp:28-181) for gray opacities, lean arithmetic
// and exact block geometry (power-of-two cell widths: every stepdiff deck): the headline kernel.
//
// The tracking step of transport_utils.hpp:111-160 in CELL-LOCAL coordinates.  While a lane follows
// a photon it carries
//   p = x - (centre of the photon's cell), per axis, |p| <= h = dx / 2;
//   omega = v / c and d_rem = c (t_end - t)                     (as imc_step_dir, jb_physics.hpp);
//   the byte offset of the cell's entry in the mean-free-path arrays of the resident blocks
// and nothing else of the geometry: no cell index per axis, no block corner.  Per axis the step is
//   distance to the face ahead   (h - sgn(omega) p) / |omega| = fma(-p, 1/omega, h |1/omega|)
//   move                         p = fma(omega, d, p)
//   nudge + Xtoijk               |p| > h - eps_imc dx  ->  p = -+(h - eps_imc dx), offset +- stride
// -- the photon is eps_imc dx beyond the face it reached, i.e. that far inside the next cell, which
// is where transport_utils.hpp:151-159 puts it and what Xtoijk (transport.cpp:146) then finds; both
// faces of an axis are tested, as in the reference.  What the x-space form spends per axis and pass
// on the face coordinate (index -> double, fma), on the difference face - x, on the signed nudge
// width, on the index arithmetic and on the six "still on this block?" comparisons is gone: 8
// instead of 18 vector instructions per axis.
// "Has the photon left its block?" is read off the datum the lane gathers anyway: a ghost cell of
// the scattering mean-free-path array holds a negative number whose low word is the byte offset of
// the cell a photon stepping into it really is in (k_lam_ghost_codes, once per mesh: the first
// interior cell of the same-level resident neighbour, of the block across a periodic boundary, or,
// at a reflecting wall, the cell it came from, position and direction mirrored).  The value
// requested at the end of a pass is looked at in the next one; a lane that finds such a number
// takes the offset, requests that cell's datum and sits the pass out (0.36 % of the events of
// BASELINE configs[1]) -- p does not change across a face between blocks of one size, and the block
// index is not part of the loop's state at all (offset / bytes per block, where a consumer appears).
// Level changes, destinations that are not resident, outflow, edges and corners materialise x,
// (i, j, k) and take the general relocation.
// A scatter or an absorption that ends within eps_imc dx of a BLOCK face (the reference: relocation
// first, the collision is dropped -- transport.cpp:149-155) waits for the gathered value.
//
// Arithmetic: positions relative to the cell centre carry ~8 more bits than absolute ones, every
// operation is within 4e-15 (relative) of the exact variant's, draws and their order are the
// reference's.  The stated tolerance of the lean variant (include/jaybenne_amd.h) is unchanged and
// is what tests/test_gpu_lean.py / test_gpu_accuracy.py hold this kernel to; bit-for-bit parity with
// the oracle is the exact variant's (k_transport<.., EXACT, !LEAN>).
#pragma once

#include "jb_kernels.hpp"

namespace jb {

// Waves per SIMD the register allocator is held to.  The 2-D and 1-D kernels need < 128 registers
// anyway (four waves).  The 3-D kernel takes 140 with the block geometry per lane (blocks of several
// sizes: SMR) and stays at three waves; on a mesh whose resident blocks all have ONE cell size
// (UNIFORM: BASELINE configs[1]) the geometry is wave-uniform -- seven scalar register pairs instead
// of fourteen vector registers -- and the kernel fits 128 registers without spilling: four waves,
// 2.4 % (A/B on one box, twice).  (Held to 128 with per-lane geometry it parks 33 service-phase
// values in scratch memory around the loop: the same 2.4 %, and 8 GB of scratch traffic per launch.)
#ifndef JB_IMC_WAVES_PER_SIMD_UNIFORM
#define JB_IMC_WAVES_PER_SIMD_UNIFORM 4
#endif
// Blocks of several sizes in 3-D (per-lane geometry, 140 registers): four waves as well, 14 dwords
// stored and reloaded around the event loop -- stepdiff_smr in 3-D (64 x 32 x 32 cells, 1e7 photons)
// 50.5 -> 48.0 ms.
#ifndef JB_IMC_WAVES_PER_SIMD
#define JB_IMC_WAVES_PER_SIMD 4
#endif
// 1-D / 2-D: four waves as well (122 / 108 registers: no scratch).  Rounds 4 and 5 ran five (96 registers, the
// rest stored and reloaded around the event loop) for 43.6 -> 43.0 ms on BASELINE configs[3]; measured again in
// round 6 on one box (tools/dev/c4_waves.sh): 43.10 (five) against 43.02 ms (four) -- and the five-wave form's
// scratch traffic is 2.9 GB read + 12.9 GB written per 1e7 histories against 0.44 + 1.96 GB: not worth a
// memory system that eight ranks and RCCL share.  (The 3-D kernel at five waves loses: 57.3 -> 59.9 ms.)
#ifndef JB_IMC_WAVES_PER_SIMD_LOWD
#define JB_IMC_WAVES_PER_SIMD_LOWD 4
#endif
#ifndef JB_IMC_NT_STORES
#define JB_IMC_NT_STORES 0
#endif
#ifndef JB_IMC_SERVICE_BUDGET
#define JB_IMC_SERVICE_BUDGET 96
#endif

// the kernel's argument list as the kernel-argument segment holds it
struct ImcArgs {
  const DevMesh *Mp;
  DevParams P;
  DevSwarm S;
  double t_start, dt;
  long long first, last;
  unsigned long long *counters;
  const int *nbr_dq;
};

enum { IS_IDLE = 0, IS_RUN = 1, IS_DONE = 2, IS_DONE_RAW = 3 };
// the lean logarithm on its 1024-row table (m_log_lean<SC, true>): 16 KB of LDS per workgroup instead of 2
#ifndef JB_IMC_WIDE_LOG
#define JB_IMC_WIDE_LOG 1
#endif
constexpr bool kWideLog = JB_IMC_WIDE_LOG != 0;


template <int NDIM, bool TALLY, bool NOABS, bool UNIFORM>
__global__ void __launch_bounds__(kBlock, NDIM < 3 ? JB_IMC_WAVES_PER_SIMD_LOWD
                                          : UNIFORM ? JB_IMC_WAVES_PER_SIMD_UNIFORM : JB_IMC_WAVES_PER_SIMD)
    k_imc_cell(const DevMesh *__restrict__, DevParams, DevSwarm, double, double, long long, long long,
               unsigned long long *, const int *) {
  // The arguments are read where they are used, from the kernel-argument segment (scalar loads), and
  // the mesh view through a pointer to its copy in device memory: the event loop needs four scalars
  // of it (base of the scattering mean free paths, two strides), everything else is service-phase
  // material that would otherwise sit in ~120 scalar registers across the loop (k_hybrid).
  const ImcArgs &A = *(const ImcArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const DevMesh &M = *A.Mp;
  const DevParams &P = A.P;
  const DevSwarm &S = A.S;
  const double t_start = A.t_start, dt = A.dt;
  const long long first = A.first, last = A.last;
  unsigned long long *const counters = g1(A.counters);
  (void)A.nbr_dq;   // (the face table's byte-offset changes live in the ghost codes of lam_sc since round 4)
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  __shared__ double lds_tally[TALLY ? kLdsTally : 1];
  const bool tally_in_lds = TALLY && (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  if constexpr (TALLY) {
    if (tally_in_lds)
      for (int q = threadIdx.x; q < M.nblocks * (int)M.ntot; q += blockDim.x) lds_tally[q] = 0.0;
  }
  // (a non-absorbing material only: with the absorption logarithm beside it the 3-D form spills registers)
  constexpr bool kWide = kWideLog && NOABS;
  load_math_tables<false, !kWide, false, true, kWide>();  // lean logarithm, sincos of 2 pi u (ends with a barrier)

  constexpr int kServiceBudget = JB_IMC_SERVICE_BUDGET;
  const double vv = P.c;
  const double t_end = t_start + dt;
  const int lane = threadIdx.x & 63;
  unsigned long long *queue = counters + CNT_QUEUE;
  const long long per_q = (last - first + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues;
  int tried = 0;
  bool more = true;

  unsigned int c_census = 0, c_abs = 0, c_esc = 0, c_out = 0;
  // wave-level (scalar): events in the low 40 bits, passes above them (one 64-bit add per pass)
  unsigned long long c_evp = 0;
  unsigned int c_service = 0;

  // byte strides of the [nk][nj][ni] cell arrays; the two arrays of block b start at 16 b ntot
  // bytes behind lam_abs0 / lam_sc0 (jb_mesh_create: everything below 4 GiB)
  const int sy = (int)sgpr_copy(8u * (unsigned)M.ni), sz = (int)sgpr_copy(8u * (unsigned)(M.ni * M.nj));
  const unsigned blk_bytes = 16u * (unsigned)M.ntot;
  const char *const lam_abs0 = (const char *)sgpr_copy_ptr(M.lam_base);
  const char *const lam_sc0 = (const char *)sgpr_copy_ptr(M.lam_base + M.ntot);
  const double inv_ni = 1.0 / (double)M.ni, inv_ninj = 1.0 / (double)(M.ni * M.nj);
  const double inv_blk_bytes = 1.0 / (double)blk_bytes;

  // lane state
  int ls = IS_IDLE;
  long long n = 0;
  LcgRng rng(0);
  int b = 0, status = ST_ACTIVE;
  unsigned qoff = 0u;                       // byte offset of the photon's cell (block included)
  double px = 0, py = 0, pz = 0;            // position relative to the cell centre
  double ox = 0, oy = 0, oz = 0;            // unit direction (IS_DONE_RAW: the velocity as loaded)
  double drem = 0;                          // distance left to census (IS_DONE_RAW: the time as loaded)
  // half cell widths of the block, h - eps_imc dx, min cell width of the block (transport.cpp:75-78):
  // per lane, or (UNIFORM: every resident block has the cell size of block 0) once per wave
  CellGeom cg{0, 0, 0, 0, 0, 0, 0};
  if constexpr (UNIFORM) {
    const double d0 = uniform_f64(M.blk_dx[0]), d1 = uniform_f64(M.blk_dx[1]), d2 = uniform_f64(M.blk_dx[2]);
    cg.hx = 0.5 * d0; cg.hy = 0.5 * d1; cg.hz = 0.5 * d2;
    cg.mx = cg.hx - kEpsImc * d0; cg.my = cg.hy - kEpsImc * d1; cg.mz = cg.hz - kEpsImc * d2;
    cg.dxp = dmin(d0, dmin(d1, d2));
  }
  double lam_a = 0.0, lam_s = 0.0;          // mean free paths of the photon's cell, or the ghost code

  auto fetch_lam = [&]() {
    if constexpr (!NOABS) lam_a = *(gcptr)(lam_abs0 + qoff);
    lam_s = *(gcptr)(lam_sc0 + qoff);
  };
  auto bind_block = [&](int blk) {
    if constexpr (UNIFORM) return;
    const double d0 = ((gcptr)M.blk_dx)[3 * blk], d1 = ((gcptr)M.blk_dx)[3 * blk + 1],
                 d2 = ((gcptr)M.blk_dx)[3 * blk + 2];
    cg.hx = 0.5 * d0; cg.hy = 0.5 * d1; cg.hz = 0.5 * d2;
    cg.mx = cg.hx - kEpsImc * d0; cg.my = cg.hy - kEpsImc * d1; cg.mz = cg.hz - kEpsImc * d2;
    cg.dxp = dmin(d0, dmin(d1, d2));
  };
  // the resident block the byte offset lies in (formed where a consumer appears: the loop does not
  // carry it) -- offsets are multiples of 8, so (qoff + 4) / blk_bytes is >= 4 / blk_bytes away from
  // an integer and the rounded product truncates to the quotient
  auto block_of = [&]() { return (int)(((double)qoff + 4.0) * inv_blk_bytes); };
  // cell (i, j, k) of the byte offset (ghost layers included)
  auto cell_of = [&](int &i, int &j, int &k) {
    const int q = (int)((qoff - (unsigned)b * blk_bytes) >> 3);
    k = three_d ? (int)(((double)q + 0.5) * inv_ninj) : 0;
    const int r = q - k * (M.ni * M.nj);
    j = multi_d ? (int)(((double)r + 0.5) * inv_ni) : 0;
    i = r - j * M.ni;
  };
  // centre of cell index idx along axis d of block blk: x0 + (idx + 0.5) dx, exact on this geometry
  auto centre = [&](int blk, int d, int idx) {
    const double dxd = ((gcptr)M.blk_dx)[3 * blk + d];
    const int first_d = d == 0 ? M.is : (d == 1 ? M.js : M.ks);
    const double x0 = ((gcptr)M.blk_xmin)[3 * blk + d] - (double)first_d * dxd;
    return fma((double)idx + 0.5, dxd, x0);
  };
  // the photon in the swarm's terms: position, cell
  auto materialise = [&](double &x, double &y, double &z, int &i, int &j, int &k) {
    cell_of(i, j, k);
    x = centre(b, 0, i) + px;
    y = multi_d ? centre(b, 1, j) + py : py;
    z = three_d ? centre(b, 2, k) + pz : pz;
  };
  // ... and back (the photon sits in cell (i, j, k) of resident block b)
  auto localise = [&](double x, double y, double z, int i, int j, int k) {
    px = x - centre(b, 0, i);
    py = multi_d ? y - centre(b, 1, j) : y;
    pz = three_d ? z - centre(b, 2, k) : z;
    qoff = (unsigned)b * blk_bytes + ((unsigned)cidx(M, k, j, i) << 3);
  };
  // the comm phase of the reference for one photon in flight, in general (boundary conditions,
  // destination block by the leaf map: level changes, destinations that are not resident, outflow)
  // (a finished lane holds: false = local coordinates (p, offset); true = the absolute position in
  // p and no cell -- escaped, or bound for a block that is not resident here)
  bool abs_pos = false;
  auto relocate = [&]() {
    double x, y, z;
    int i, j, k;
    materialise(x, y, z, i, j, k);
    // (inactive axes: p holds the coordinate itself)
    if (!apply_swarm_bcs<NDIM>(M, x, y, z, ox, oy, oz)) {
      status = ST_ESCAPED;
      px = x; py = y; pz = z;  // written back as they are
      abs_pos = true;
      ls = IS_DONE;
      return;
    }
    const int g = find_block<NDIM>(M, x, y, z);
    const int li = M.local_index[g];
    if (li < 0) {  // not resident here: hand the photon to the block's owner
      status = ST_OUTGOING;
      b = g;  // global id travels in blk
      px = x; py = y; pz = z;
      abs_pos = true;
      ls = IS_DONE;
      return;
    }
    b = li;
    bind_block(b);
    Blk B;
    load_block(M, b, B);
    xtoijk<NDIM>(M, B, x, y, z, i, j, k);
    localise(x, y, z, i, j, k);
    ls = (drem > 0.0) ? IS_RUN : IS_DONE;
    fetch_lam();
  };

  for (;;) {
    // ================================ SERVICE ================================
    ++c_service;
    if (ls == IS_DONE || ls == IS_DONE_RAW) {
      double x = px, y = py, z = pz;
      int ip = 0, jp = 0, kp = 0;
      if (!abs_pos) {
        b = block_of();
        materialise(x, y, z, ip, jp, kp);
      }
      double t = drem, vx = ox, vy = oy, vz = oz;
      if (ls == IS_DONE) {  // back to time and velocity (census: d_rem = 0 exactly, t = t_end)
        t = fma(-drem, P.rc, t_end);
        vx *= vv; vy *= vv; vz *= vv;
      }
      int bw = b;
      if (!abs_pos && (status == ST_ACTIVE || status == ST_OUTGOING_ABSORBED) && !M.owned[b]) {
        // finished inside a halo copy: the owner of the block takes it from here
        if (status == ST_ACTIVE) status = ST_OUTGOING;
        bw = M.gid[b];
      }
      // (-DJB_IMC_NT_STORES=1: the write-back with the non-temporal hint.  Measured in round 6, counters of one launch,
      // read + written: BASELINE configs[1] (two 161 MB per-cell arrays going through L2) 1.91 + 5.03 GB plain,
      // 1.35 + 4.61 GB with the hint; configs[3] (a 0.4 MB mesh: L2 keeps a line until its neighbours' stores have
      // arrived) 0.44 + 1.96 GB plain, 0.41 + 5.28 GB with it; the time the same in all four (tools/dev/c2_stores.sh).
      // Plain stores it is.  What is left of the write side is 13 scattered 4- and 8-byte stores per finished
      // history at 32 bytes of memory traffic each where L2 cannot merge them: the price of a structure-of-arrays
      // swarm whose photons finish one at a time.)
#if JB_IMC_NT_STORES
#define JB_WST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define JB_WST(p, v) (*(p) = (v))
#endif
      JB_WST(&g1(S.blk)[n], bw);
      JB_WST(&g1(S.t)[n], t);
      JB_WST(&g1(S.x)[n], x); JB_WST(&g1(S.y)[n], y); JB_WST(&g1(S.z)[n], z);
      JB_WST(&g1(S.vx)[n], vx); JB_WST(&g1(S.vy)[n], vy); JB_WST(&g1(S.vz)[n], vz);
      JB_WST(&g1(S.ip)[n], ip); JB_WST(&g1(S.jp)[n], jp); JB_WST(&g1(S.kp)[n], kp);
      JB_WST(&g1(S.status)[n], status);
      JB_WST(&g1(S.rng)[n], (uint64_t)rng.s);
#undef JB_WST
      if (status == ST_ACTIVE) {
        ++c_census;
        if constexpr (TALLY) {  // jaybenne.cpp:547-561
          const double dv = (8.0 * cg.hx) * cg.hy * cg.hz;  // dx dy dz: powers of two, exact in any order
          const int q = (int)((qoff - (unsigned)b * blk_bytes) >> 3);
          if (tally_in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + q], g1(S.w)[n] / dv);
          else atomicAdd(&M.tally[b][q], g1(S.w)[n] / dv);
        }
      } else if (status == ST_ABSORBED) {
        ++c_abs;
      } else if (status == ST_ESCAPED) {
        ++c_esc;
      } else {
        ++c_out;
      }
      ls = IS_IDLE;
      abs_pos = false;
    }
    {
      // hand new particles to idle lanes: one claim of exactly what they need per service phase
      const unsigned long long idle = __ballot(ls == IS_IDLE);
      if (idle != 0ull && more) {
        const int leader = __ffsll((long long)idle) - 1;
        const int want = __popcll(idle);
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&queue[cur * kQueueStride], (unsigned long long)want);
        base = __shfl(base, leader, 64);
        const long long q_first = first + (long long)cur * per_q;
        long long q_last = q_first + per_q;
        if (q_last > last) q_last = last;
        const long long cand = q_first + (long long)base + __popcll(idle & ((1ull << lane) - 1ull));
        if (q_first + (long long)base + want >= q_last) {  // this queue is drained: move on
          cur = (cur + 1) % kQueues;
          if (++tried == kQueues) more = false;
        }
        if (ls == IS_IDLE && cand < q_last && g1(S.status)[cand] == ST_ACTIVE) {
          n = cand;
          rng.s = g1(S.rng)[n];
          b = g1(S.blk)[n];
          bind_block(b);
          const double t = g1(S.t)[n];
          const double x = g1(S.x)[n], y = g1(S.y)[n], z = g1(S.z)[n];
          ox = g1(S.vx)[n]; oy = g1(S.vy)[n]; oz = g1(S.vz)[n];
          status = ST_ACTIVE;
          Blk B;
          load_block(M, b, B);
          int i, j, k;
          xtoijk<NDIM>(M, B, x, y, z, i, j, k);  // transport.cpp:96
          localise(x, y, z, i, j, k);
          if (t < t_end) {
            drem = vv * (t_end - t);
            ox *= P.rc; oy *= P.rc; oz *= P.rc;
            ls = IS_RUN;
          } else {  // already at census: nothing to track, nothing to convert
            drem = t;
            ls = IS_DONE_RAW;
          }
          fetch_lam();
        }
      }
    }
    const int running = __popcll(__ballot(ls == IS_RUN));
    if (running == 0) {
      if (__ballot(ls != IS_IDLE) != 0ull || more) continue;
      break;
    }
    int waste = 0;

    // ================================ EVENTS =================================
    int thresh = 1;
    int nrun = running;
    while (nrun >= thresh) {
      const bool run = ls == IS_RUN;
      // ---- a photon in a ghost cell: it has left its block (transport.cpp:149-155)
      const bool cross = run && lam_s < 0.0;
      const bool stepping = run && !cross;  // (a crossing lane sits this pass out: the new cell's datum is on its way)
      // (the ballots of the two comparisons, ANDed as scalars: the ballot of a conjunction goes through
      // a vector register)
      const unsigned long long cross_m =
          __builtin_amdgcn_ballot_w64(ls == IS_RUN) & __builtin_amdgcn_ballot_w64(lam_s < 0.0);
      c_evp += (1ull << 40) + (unsigned long long)(nrun - __popcll(cross_m));
      if (cross_m != 0ull) {
        if (cross) {
          const int code = __double2hiint(lam_s);
          if ((code & kGhostTable) == 0) {
            b = block_of();
            relocate();
          } else {
            // (a reflecting wall normal to x / y / z: position and direction mirrored -- the sign bits
            // flipped by bits 16 / 17 / 18 of the code, moved to bit 31)
            const int fx = (code << 15) & (int)0x80000000u, fy = (code << 14) & (int)0x80000000u,
                      fz = (code << 13) & (int)0x80000000u;
            px = __hiloint2double(__double2hiint(px) ^ fx, __double2loint(px));
            ox = __hiloint2double(__double2hiint(ox) ^ fx, __double2loint(ox));
            if (multi_d) {
              py = __hiloint2double(__double2hiint(py) ^ fy, __double2loint(py));
              oy = __hiloint2double(__double2hiint(oy) ^ fy, __double2loint(oy));
            }
            if (three_d) {
              pz = __hiloint2double(__double2hiint(pz) ^ fz, __double2loint(pz));
              oz = __hiloint2double(__double2hiint(oz) ^ fz, __double2loint(oz));
            }
            qoff = (unsigned)__double2loint(lam_s);
            if constexpr (!UNIFORM) {
              if (code & (kGhostCoarser | kGhostFiner))  // a resident block one level up or down
                qoff = cross_level<NDIM>(code, qoff, sy, sz, cg, px, py, pz);
            }
            fetch_lam();
            if (!(drem > 0.0)) ls = IS_DONE;  // (reached census and a block face in one step)
          }
        }
      }
      if (stepping) {
        bool is_absorbed, is_scattered, hit_any;
        imc_step_cell<NDIM, NOABS, UNIFORM, kWide>(cg, sy, sz, lam_a, lam_s, rng, drem, px, py, pz, ox, oy, oz, qoff,
                                            is_absorbed, is_scattered, hit_any);
        fetch_lam();  // (for the next pass, ahead of the scatter)
        const bool census = !(drem > 0.0);
        bool collide = is_absorbed || is_scattered;
        bool off = false;
        const bool at_face = hit_any && (collide || census);
        if (__builtin_amdgcn_ballot_w64(at_face) != 0ull) {
          // a collision or the census within eps of a cell face (one event in ~1e8): if that face is
          // a block face the reference relocates the photon first (and forgets the collision,
          // transport.cpp:149-155).  (A branch of its own: the test waits for the value just requested.)
          asm volatile("; collision or census next to a cell face" ::: "memory");
          if (at_face) off = lam_s < 0.0;
        }
        collide = collide && !off;
        if (!NOABS && is_absorbed && collide) {  // transport.cpp:157-163
          b = block_of();
          if (M.owned[b]) {
            atomicAdd(&M.edelta[b][(qoff - (unsigned)b * blk_bytes) >> 3], g1(S.w)[n]);
            status = ST_ABSORBED;
          } else {
            status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
          }
          ls = IS_DONE;
        } else {
          if (is_scattered && collide) scatter_dir<true>(rng, ox, oy, oz);  // transport.cpp:165-170
          if (census && !off) ls = IS_DONE;
        }
      }
      nrun = __popcll(__ballot(ls == IS_RUN));
      waste += running - nrun;
      if (waste >= kServiceBudget) thresh = 65;
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
  unsigned long long r_census = wave_sum(c_census), r_abs = wave_sum(c_abs), r_esc = wave_sum(c_esc),
                     r_out = wave_sum(c_out);
  if (lane == 0) {
    if (r_census) atomicAdd(&counters[CNT_CENSUS], r_census);
    if (r_abs) atomicAdd(&counters[CNT_ABSORBED], r_abs);
    if (r_esc) atomicAdd(&counters[CNT_ESCAPED], r_esc);
    if (r_out) atomicAdd(&counters[CNT_OUTGOING], r_out);
    atomicAdd(&counters[CNT_EVENTS], c_evp & ((1ull << 40) - 1ull));
    atomicAdd(&counters[CNT_PASSES], c_evp >> 40);
    atomicAdd(&counters[CNT_SERVICE], (unsigned long long)c_service);
  }
}

}  // namespace jb
