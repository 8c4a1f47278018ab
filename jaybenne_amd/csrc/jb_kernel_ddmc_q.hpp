// jb_kernel_ddmc_q.hpp -- k_ddmc_all<.., cell codes> with the wave's particles staged through two small
// queues in LDS (round 6).  Same task, same arithmetic, same bits as k_ddmc_all (jb_kernel_ddmc.hpp:
// TransportPhotons_DDMC on a mesh whose every cell takes the DDMC branch, transport_ddmc.cpp:69-230 with
// transport_utils.hpp:163-397); what changes is WHEN a lane does what.
//
// What bound k_ddmc_all once its per-step gather was 4 bytes (profiles/r06_*): instruction issue -- ~155
// instructions per 64-lane pass of the event loop at 51 lanes in use, plus a service phase of ~1100
// instructions every ~20 passes that handles the ~31 lanes whose history has ended (write-back, census
// resampling, tally) and loads their successors, i.e. runs at half width; on the reference's 1-D deck
// (12-step histories) that phase served 11 lanes at a time and took half of the kernel.  Here
//   * a lane whose history ends inside the event loop puts what is left of it -- slot, record number, time,
//     stream state, pending leak: 40 bytes -- on the wave's DONE queue and takes its next photon from the
//     wave's READY queue (slot, record number, time, stream state: 24 bytes, prepared ahead) in the same
//     pass: ~40 instructions in the passes that end a history, and the loop runs with (nearly) all 64 lanes;
//   * the service phase works on whole queues: 64 finished histories at a time (decode, census resampling
//     transport_utils.hpp:265-276, write-back, tally jaybenne.cpp:547-561, block crossings that the loop
//     does not resolve) and 64 new photons at a time (load, Xtoijk transport.cpp:96, the albedo step's face
//     tests) -- every lane busy in both.
// The queues are private to a wave (no barriers); a history's draws and operations do not depend on the lane
// or the pass it is followed in, so the particles come out bit-identical (tests/test_gpu_parity.py holds this
// kernel, k_ddmc_all and the general kernel to the oracle).
#pragma once

#include "jb_kernel_ddmc.hpp"

namespace jb {

constexpr int kQReady = 128;   // entries: up to 64 left over + 64 photons per fill (or re-entering after a relocation)
constexpr int kQDone = 64;     // one full-width batch for the service phase
constexpr int kQBlocks = 64;   // per-block tables in LDS (k_ddmc_all: 128): with the queues the workgroup stays under
                               // 40 KB of LDS, i.e. four workgroups per CU
#ifndef JB_DDMC_Q_RETIRE_MIN   // lanes without a running photon before the loop exchanges them (3-D / 2-D; 1-D: 1)
#define JB_DDMC_Q_RETIRE_MIN 1
#endif
#ifndef JB_DDMC_Q_BUDGET       // idle lane-passes the loop's tail (no photons left to load) spends before it is left
#define JB_DDMC_Q_BUDGET 256
#endif

struct WaveQueues {
  // READY: photons prepared for the event loop -- slot | record number << 32, stream state, time
  unsigned long long rd_nrec[kQReady], rd_rng[kQReady];
  double rd_t[kQReady];
  // DONE: what is left of a history that has left the loop -- ... and the stream state in front of its pending
  // leak's two deferred draws, the leak's record-number step | (how it left the loop, mirror flag) << 32
  unsigned long long dn_nrec[kQDone], dn_rng[kQDone], dn_pzs[kQDone], dn_pdfl[kQDone];
  double dn_t[kQDone];
};

// LCODES: the cell codes of a mesh of at most kLdsCodeCells cells (the reference's 1-D decks) are copied to LDS
// as well, and the event loop has no vector-memory instruction left: its gather no longer queues in the CU's
// vector L1 behind the scattered photon loads and stores of the other waves' service phases (measured on
// BASELINE configs[2] as shipped: 910 of a pass's 2560 wave-cycles were that wait; the table is 544 bytes).
constexpr int kLdsCodeCells = 1024;

template <int NDIM, bool TALLY, bool LCODES = false>
__global__ void __launch_bounds__(kBlock, JB_DDMC_ALL_WAVES_PER_SIMD)
    k_ddmc_q(const DevMesh *__restrict__, DevParams, DevSwarm, double, double, long long, long long,
             unsigned long long *, const int *, unsigned *, unsigned long long *) {
  // (arguments read where they are used, from the kernel-argument segment: see k_ddmc_all)
  const DdmcAllArgs &A = *(const DdmcAllArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  if (*A.not_all_ddmc != 0) return;  // (uniform) some cell takes IMC steps: k_hybrid runs instead
  const DevMesh &M = *(const DevMesh *)A.Mp;
  const DevParams &P = A.P;
  const DevSwarm &S = A.S;
  const double t_start = A.t_start, dt = A.dt;
  const long long first = A.first, last = A.last;
  unsigned long long *const counters = g1(A.counters);
  // non-temporal swarm accesses (see swarm_ld): on a large mesh they keep the particle stream from pushing the
  // cell codes out of L2 (C3: 17.4 against 18.2 ms); on a mesh whose codes sit in LDS nothing competes with the
  // stream, and the hint costs time at the same traffic (the 1-D deck: 7.38 against 7.06 ms, 4.4 + 13.8 GB either way)
#ifndef JB_DDMC_Q_NT_LD_SMALL   // (A/B: loads / stores apart)
#define JB_DDMC_Q_NT_LD_SMALL 0
#endif
#ifndef JB_DDMC_Q_NT_ST_SMALL
#define JB_DDMC_Q_NT_ST_SMALL 0
#endif
  constexpr bool NT_LD = JB_DDMC_NT != 0 && (!LCODES || JB_DDMC_Q_NT_LD_SMALL != 0);
  constexpr bool NT_ST = JB_DDMC_NT != 0 && (!LCODES || JB_DDMC_Q_NT_ST_SMALL != 0);
  constexpr bool multi_d = NDIM >= 2;
  constexpr int kRetireMin = NDIM == 1 ? 1 : JB_DDMC_Q_RETIRE_MIN;
  typedef double v4d __attribute__((ext_vector_type(4)));
  typedef const unsigned __attribute__((address_space(1))) *gcptr_u;

  // dynamic shared memory: the tally of a small mesh, then the distinct step records of this cycle
  extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
  double *const lds_tally = lds_dyn;
  const bool tally_in_lds = TALLY && (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  const int ncell_all = M.nblocks * (int)M.ntot;
  double *const lds_cls = lds_dyn + (tally_in_lds ? (ncell_all + 1) / 2 * 2 : 0);
  if constexpr (TALLY) {
    if (tally_in_lds)
      for (int q = threadIdx.x; q < ncell_all; q += blockDim.x) lds_tally[q] = 0.0;
  }
  const int ncls = ((gcptr_i)M.not_all_ddmc)[1];
  for (int q = threadIdx.x; q < 8 * ncls; q += blockDim.x) lds_cls[q] = ((gcptr)M.ddmc_class)[q];
  unsigned *const lds_code = (unsigned *)(lds_cls + 8 * ncls);
  if constexpr (LCODES) {
    for (int q = threadIdx.x; q < ncell_all; q += blockDim.x) lds_code[q] = ((gcptr_u)M.ddmc_code)[q];
  }
  __shared__ LdsBlockTableT<false, kQBlocks> lds_blocks;   // (the host launches this kernel on <= kQBlocks resident blocks)
  // the wave's queues (structure of arrays: consecutive lanes, consecutive 8-byte words; one block per wave, so
  // that an entry's words are one address register and constant offsets apart)
  __shared__ WaveQueues lds_queues[kBlock / 64];
  fill_block_table(M, lds_blocks);
  load_math_tables<true, false, false, true>();  // logarithm, sincos of 2 pi u (ends with a barrier)
  WaveQueues &Q = lds_queues[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  unsigned long long *const rd_nrec = Q.rd_nrec, *const rd_rng = Q.rd_rng;
  double *const rd_t = Q.rd_t;
  unsigned long long *const dn_nrec = Q.dn_nrec, *const dn_rng = Q.dn_rng, *const dn_pzs = Q.dn_pzs,
                     *const dn_pdfl = Q.dn_pdfl;
  double *const dn_t = Q.dn_t;

  // (slots per claim: with 12-step histories the claims come three times as often per unit of time -- 256 against
  // 128 slots: 7.58 against 7.79 ms on the 1-D deck, 17.75 against 17.62 on the 3-D one)
  constexpr long long kChunk = NDIM == 1 ? 2 * JB_DDMC_ALL_CHUNK : JB_DDMC_ALL_CHUNK;
  const double vv = P.c;
  const double t_end = t_start + dt;
  const int lane = threadIdx.x & 63;
  unsigned long long *queue = counters + CNT_QUEUE;
  const long long per_q = (last - first + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues, tried = 0;
  bool more = true;
  long long chunk_pos = 0, chunk_end = 0;
  int ready_cnt = 0, done_cnt = 0;   // (wave-uniform)
#ifdef JB_DDMC_EXP_SEQSTORE
  long long exp_wr_cur = 0, exp_wr_end = 0;
#endif

  unsigned int c_census = 0, c_abs = 0, c_esc = 0, c_out = 0;
  unsigned long long c_ev = 0;
  unsigned int c_pass = 0, c_service = 0;

  // ---- a lane's photon in the event loop ("virtual" state: see k_ddmc_all)
  int r_ls = DS_IDLE;
  unsigned r_n = 0u, r_rec = 0u;
  unsigned long long r_rng = 0ull, r_pzs = 0ull;
  double r_t = 0.0;
  int r_pd = 0;
  unsigned r_mir = 0u;   // (0 / 1: an integer, so that it stays in a vector register -- as a lane predicate it was
                         // carried through the loop in a scalar pair and rebuilt from a register every pass)
  unsigned code = 0u;

  auto faces_of = [&](Step &s, const Blk &Bq, int i, int j, int k) {  // transport.cpp:114-119
    s.xl = xc(Bq, 0, i) - 0.5 * Bq.dx[0]; s.xu = xc(Bq, 0, i) + 0.5 * Bq.dx[0];
    s.yl = xc(Bq, 1, j) - 0.5 * Bq.dx[1]; s.yu = xc(Bq, 1, j) + 0.5 * Bq.dx[1];
    s.zl = xc(Bq, 2, k) - 0.5 * Bq.dx[2]; s.zu = xc(Bq, 2, k) + 0.5 * Bq.dx[2];
  };
  const unsigned ntot_u = sgpr_copy((unsigned)M.ntot);
  const unsigned *const code_base = (const unsigned *)sgpr_copy_ptr((const double *)M.ddmc_code);
  const int l_ni = (int)sgpr_copy((unsigned)M.ni), l_nj = (int)sgpr_copy((unsigned)M.nj);
  const int l_is = (int)sgpr_copy((unsigned)M.is), l_ie = (int)sgpr_copy((unsigned)M.ie);
  const int l_js = (int)sgpr_copy((unsigned)M.js), l_je = (int)sgpr_copy((unsigned)M.je);
  const int l_ks = (int)sgpr_copy((unsigned)M.ks), l_ke = (int)sgpr_copy((unsigned)M.ke);
  auto cidx_l = [&](int k, int j, int i) { return __mul24(__mul24(k, l_nj) + j, l_ni) + i; };
  const int l_nij = (int)sgpr_copy((unsigned)(M.ni * M.nj));
  const double inv_ntot = M.inv_ntot, inv_nij = M.inv_nij, inv_ni = M.inv_ni;
  auto udiv = [](unsigned x, unsigned d, double inv, unsigned &rem) {   // floor(x / d): see k_ddmc_all
    unsigned q = (unsigned)((double)x * inv);
    rem = x - q * d;
    if (rem >= d) { ++q; rem -= d; }
    return q;
  };
  auto on_block_l = [&](int i, int j, int k) {
    bool on = i >= l_is && i <= l_ie;
    if constexpr (NDIM >= 2) on = on && j >= l_js && j <= l_je;
    if constexpr (NDIM == 3) on = on && k >= l_ks && k <= l_ke;
    return on;
  };
  auto load_code = [&](unsigned recno) {
    if constexpr (LCODES) return lds_code[recno];
    else return *(gcptr_u)((const char *)code_base + (recno << 2));
  };
  auto below = [&](unsigned long long m) {   // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  };

  // A lane whose history has left the loop puts it on the DONE queue (as long as there is room: a lane that
  // finds none keeps it, and the loop is left); a lane without a photon takes the next one off the READY queue.
  // (wave-uniform: some lane stands idle because the READY queue had nothing for it)
  bool have_idle = true;
  auto retire_refill = [&]() {
    const unsigned long long fm = __ballot(r_ls >= DS_DONE);
    const int nf = __popcll(fm);
    if (!have_idle && nf != 0 && nf <= kQDone - done_cnt && nf <= ready_cnt) {
      // the common case: every finished history finds room AND a successor, and no other lane is waiting for one --
      // ONE exchange: entry out, entry in
      if (r_ls >= DS_DONE) {
        const int rk = below(fm);
        const int q = done_cnt + rk;
        dn_nrec[q] = (unsigned long long)r_n | ((unsigned long long)r_rec << 32);
        dn_t[q] = r_t;
        dn_rng[q] = r_rng;
        dn_pzs[q] = r_pzs;
        dn_pdfl[q] = (unsigned long long)(unsigned)r_pd | ((unsigned long long)((unsigned)r_ls | (r_mir << 3)) << 32);
        const int p = ready_cnt - 1 - rk;   // (off the top)
        const unsigned long long nr = rd_nrec[p];
        r_n = (unsigned)nr;
        r_rec = (unsigned)(nr >> 32);
        r_t = rd_t[p];
        r_rng = rd_rng[p];
        r_pd = 0;
        r_mir = 0u;
        r_ls = DS_VIRT;
      }
      done_cnt += nf;
      ready_cnt -= nf;
      return;
    }
    if (fm != 0ull) {
      const int room = kQDone - done_cnt;
      const int r = below(fm);
      if (r_ls >= DS_DONE && r < room) {
        const int q = done_cnt + r;
        dn_nrec[q] = (unsigned long long)r_n | ((unsigned long long)r_rec << 32);
        dn_t[q] = r_t;
        dn_rng[q] = r_rng;
        dn_pzs[q] = r_pzs;
        dn_pdfl[q] = (unsigned long long)(unsigned)r_pd | ((unsigned long long)((unsigned)r_ls | (r_mir << 3)) << 32);
        r_ls = DS_IDLE;
      }
      done_cnt += nf < room ? nf : room;
    }
    const unsigned long long wm = __ballot(r_ls == DS_IDLE);
    const int want = __popcll(wm);
    int take = 0;
    if (ready_cnt > 0 && wm != 0ull) {
      const int r = below(wm);
      take = want < ready_cnt ? want : ready_cnt;
      if (r_ls == DS_IDLE && r < take) {
        const int q = ready_cnt - 1 - r;   // (off the top)
        const unsigned long long nr = rd_nrec[q];
        r_n = (unsigned)nr;
        r_rec = (unsigned)(nr >> 32);
        r_t = rd_t[q];
        r_rng = rd_rng[q];
        r_pd = 0;
        r_mir = 0u;
        r_ls = DS_VIRT;
      }
      ready_cnt -= take;
    }
    have_idle = take < want;
  };

  constexpr unsigned long long kMul2 = kLcgMul * kLcgMul, kInc2 = (kLcgMul + 1ull) * kLcgInc;
  constexpr unsigned long long kMul4 = kMul2 * kMul2, kInc4 = (kMul2 + 1ull) * kInc2;

#ifdef JB_TIMING   // (diagnostic build, tools/dev/timing.sh: wave-cycles in the DONE batches / the fills / the event loop)
  unsigned long long cyc_ph[4] = {0, 0, 0, 0}, cyc_mark = __builtin_readcyclecounter();
  unsigned c_epi = 0;
  unsigned long long cyc_lp[4] = {0, 0, 0, 0};
#define JB_QT(k) { const unsigned long long now_ = __builtin_readcyclecounter(); cyc_ph[k] += now_ - cyc_mark; cyc_mark = now_; }
#else
#define JB_QT(k)
#endif
  for (;;) {
    // ================================ SERVICE ================================
    // two batches, each at full width: the DONE queue (phase 0), then new photons for the READY queue (phase 1)
    for (int phase = 0; phase < 2; ++phase) {
      JB_QT(phase == 0 ? 2 : 0)
      if (phase == 0 ? done_cnt == 0 : !(ready_cnt < 64 && more)) continue;
      ++c_service;
      // ---- one item per lane: what k_ddmc_all's service phase calls the lane's own state
      int ls = DS_IDLE;
      long long n = 0;
      LcgRng rng(0);
      unsigned rec = 0u;
      double t = 0.0;
      unsigned long long pzs = 0ull;
      bool mir = false;
      int b = 0, ip = 0, jp = 0, kp = 0;
      int pend = -1;
      bool resample = false, fresh = false, real_pos = false;
      int status = ST_ACTIVE;
      double x = 0.0, y = 0.0, z = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
      auto pending_uniforms = [&](double &u1, double &u2) {
        LcgRng at_leak(pzs);
        u1 = at_leak.drand();
        u2 = at_leak.drand();
      };
      // a particle with a real position (just loaded / just relocated) goes on unless the albedo step would
      // find it at a face of its cell
      auto enter = [&](const Blk &Bq) {
        Step s;
        faces_of(s, Bq, ip, jp, kp);
        s.x = x; s.y = y; s.z = z;
        const bool face = at_cell_face<NDIM>(s);
        ls = face ? DS_PARK : DS_VIRT;
        real_pos = face && real_pos;
        rec = face ? rec : (unsigned)b * ntot_u + (unsigned)cidx_l(kp, jp, ip);
        mir = face && mir;
      };
      double wgt_done = 0.0;
      if (phase == 0) {
        // ---- the DONE queue: block, cell indices and leak channel of every entry from its record number and
        //      record-number step
        int pd = 0;
        if (lane < done_cnt) {
          const unsigned long long nr = dn_nrec[lane];
          const unsigned long long pf = dn_pdfl[lane];
          n = (long long)(unsigned)nr;
          rec = (unsigned)(nr >> 32);
          t = dn_t[lane];
          rng.s = dn_rng[lane];
          pzs = dn_pzs[lane];
          pd = (int)(unsigned)pf;
          const int fl = (int)(unsigned)(pf >> 32);
          ls = fl & 7;
          mir = (fl & 8) != 0;
          // (the weight, for the tally or the absorption: asked for now, needed after the decode and the resampling)
          wgt_done = swarm_ld<NT_LD>(&g1(S.w)[n]);
        }
        done_cnt = 0;
        if (ls != DS_IDLE) {
          unsigned q, rr, ii;
          const unsigned bb = udiv(rec, ntot_u, inv_ntot, q);
          const unsigned kk = udiv(q, (unsigned)l_nij, inv_nij, rr);
          const unsigned jj = udiv(rr, (unsigned)l_ni, inv_ni, ii);
          b = (int)bb; kp = (int)kk; jp = (int)jj; ip = (int)ii;
          const int ad = pd < 0 ? -pd : pd;
          const int axis = ad == 1 ? 0 : ((NDIM >= 2 && ad == l_ni) ? 1 : 2);   // (kPdStay: z+)
          pend = pd == 0 ? -1 : (pd == kPdZero ? -2 : 2 * axis + (pd > 0 ? 1 : 0));
          resample = ls == DS_CENSUS;
          if (ls == DS_ABS) {  // transport.cpp:157-163
            if (lds_blocks.owned[b] != 0) {
              atomicAdd(&M.edelta[b][cidx_l(kp, jp, ip)], wgt_done);
              status = ST_ABSORBED;
            } else {
              status = ST_OUTGOING_ABSORBED;
            }
          }
          if (ls == DS_ABS || ls == DS_CENSUS) ls = DS_DONE;
          // (a leak whose event time rounds onto the census time may end the history in a ghost cell)
          if (ls == DS_DONE && !on_block_l(ip, jp, kp)) ls = DS_RELOC;
        }
      } else {
        // ---- new photons: every lane claims the next slot of the wave's chunk (chunks of consecutive slots,
        //      one atomic per chunk) and loads it
        long long cand = -1;
        int st_in = ST_ABSORBED, b_in = 0;
        unsigned long long rng_in = 0ull;
        double t_in = 0.0, x_in = 0.0, y_in = 0.0, z_in = 0.0, vx_in = 0.0, vy_in = 0.0, vz_in = 0.0;
        unsigned long long need = ~0ull;
        while (need != 0ull && more) {
          if (chunk_pos >= chunk_end) {
            const long long q_first = first + (long long)cur * per_q;
            long long q_last = q_first + per_q;
            if (q_last > last) q_last = last;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&queue[cur * kQueueStride], (unsigned long long)kChunk);
            chunk_pos = q_first + (long long)uniform_u64(base);
            chunk_end = chunk_pos + kChunk < q_last ? chunk_pos + kChunk : q_last;
#ifdef JB_DDMC_EXP_SEQSTORE   // (timing experiment: results are wrong)
            if (exp_wr_cur >= exp_wr_end && chunk_pos < q_last) { exp_wr_cur = chunk_pos; exp_wr_end = chunk_end; }
#endif
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
          const int rank = below(need);
          const bool mine = ((need >> lane) & 1ull) != 0ull && rank < give;
          if (mine) {
            cand = chunk_pos + rank;
            st_in = swarm_ld<NT_LD>(&g1(S.status)[cand]);
            rng_in = swarm_ld<NT_LD>(&g1(S.rng)[cand]);
            b_in = swarm_ld<NT_LD>(&g1(S.blk)[cand]);
            t_in = swarm_ld<NT_LD>(&g1(S.t)[cand]); x_in = swarm_ld<NT_LD>(&g1(S.x)[cand]); y_in = swarm_ld<NT_LD>(&g1(S.y)[cand]); z_in = swarm_ld<NT_LD>(&g1(S.z)[cand]);
            vx_in = swarm_ld<NT_LD>(&g1(S.vx)[cand]); vy_in = swarm_ld<NT_LD>(&g1(S.vy)[cand]); vz_in = swarm_ld<NT_LD>(&g1(S.vz)[cand]);
          }
          chunk_pos += give;
          need &= ~__ballot(mine);
        }
        if (cand >= 0 && st_in == ST_ACTIVE) {
          n = cand;
          rng.s = rng_in;
          b = b_in;
          t = t_in;
          x = x_in; y = y_in; z = z_in; vx = vx_in; vy = vy_in; vz = vz_in;
          real_pos = true;
          fresh = true;
          Blk Bn;
          load_block_lds(M, lds_blocks, b, Bn);
          xtoijk<NDIM>(M, Bn, x, y, z, ip, jp, kp);  // transport.cpp:96
          if (t < t_end) enter(Bn);
          else ls = DS_DONE;  // already at census: nothing to track
        }
      }
      // ---- block crossings the loop does not resolve: the comm phase of the reference for one particle in flight
      if (ls == DS_RELOC) {
        fresh = false;
        Blk Bo;
        load_block_lds(M, lds_blocks, b, Bo);
        Step s;
        s.vv = vv; s.pend = pend; s.pz1 = 0.0; s.pz2 = 0.0;
        if (pend >= 0) pending_uniforms(s.pz1, s.pz2);
        {
          // leak out of the block from the virtual state: the position the step function gave the particle
          // (transport_utils.hpp:209-263), from the cell it left and the channel
          const int axis = pend >> 1;
          const bool up = (pend & 1) != 0;
          const int step = up ? 1 : -1;
          faces_of(s, Bo, ip - (axis == 0 ? step : 0), jp - (axis == 1 ? step : 0), kp - (axis == 2 ? step : 0));
          const double dx = s.xu - s.xl, dy = s.yu - s.yl, dz = s.zu - s.zl;
          const double eps = kEpsDdmc;
          x = (axis == 0) ? (up ? s.xu + eps * dx : s.xl - eps * dx) : s.xl + 0.5 * dx;
          y = (axis == 1) ? (up ? s.yu + eps * dy : s.yl - eps * dy) : s.yl + 0.5 * dy;
          z = (axis == 2) ? (up ? s.zu + eps * dz : s.zl - eps * dz) : s.zl + 0.5 * dz;
          vx = 0.0; vy = 0.0; vz = 0.0;  // (overwritten below or flagged as a DDMC leak)
        }
        // transport_ddmc.cpp:203-211: zero velocity flags a DDMC leak for SampleDDMCBlockFace (multi-D); in 1-D
        // the direction travels with the particle
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
                swarm_st<NT_ST>(&g1(S.vx)[n], vx); swarm_st<NT_ST>(&g1(S.vy)[n], vy); swarm_st<NT_ST>(&g1(S.vz)[n], vz);
              }
            } else {
              ls = DS_DONE;
            }
          }
        }
        if (ls != DS_VIRT) real_pos = true;
      }
      // ---- finished particles: census resampling, write-back, tally
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
#ifndef JB_DDMC_EXP_NORESAMPLE   // (timing experiment: results are wrong)
            ddmc_census_resample(s, rng);
#endif
            x = s.x; y = s.y; z = s.z; vx = s.vx; vy = s.vy; vz = s.vz;
            pend = -1;
            write_v = true;
          } else if (!real_pos) {
            // absorbed (or already at census when it was loaded) in the virtual state: the albedo step left it
            // at the cell centre (transport_utils.hpp:392-396), with the direction of its last leak
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
#ifndef JB_DDMC_EXP_NOTALLY      // (timing experiment: results are wrong)
              // (a photon that was at census when it was loaded comes through the fill, not the DONE queue)
              const double wgt = phase == 0 ? wgt_done : swarm_ld<NT_LD>(&g1(S.w)[n]);
              if (tally_in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + cidx(M, kp, jp, ip)], wgt / dv);
              else atomicAdd(&lds_blocks.tally[b][cidx(M, kp, jp, ip)], wgt / dv);
#else
              (void)dv;
#endif
            }
          }
        }
#ifdef JB_DDMC_EXP_SEQSTORE
        if (phase == 0) {
          if (exp_wr_cur + 64 > exp_wr_end) exp_wr_cur = exp_wr_end - 64 > first ? exp_wr_end - 64 : first;
          n = exp_wr_cur + lane;
        }
#endif
        swarm_st<NT_ST>(&g1(S.blk)[n], b);
        swarm_st<NT_ST>(&g1(S.t)[n], t);
        swarm_st<NT_ST>(&g1(S.x)[n], x); swarm_st<NT_ST>(&g1(S.y)[n], y); swarm_st<NT_ST>(&g1(S.z)[n], z);
        if (write_v) {
          swarm_st<NT_ST>(&g1(S.vx)[n], vx); swarm_st<NT_ST>(&g1(S.vy)[n], vy); swarm_st<NT_ST>(&g1(S.vz)[n], vz);
        }
        swarm_st<NT_ST>(&g1(S.ip)[n], ip); swarm_st<NT_ST>(&g1(S.jp)[n], jp); swarm_st<NT_ST>(&g1(S.kp)[n], kp);
        // (loaded as ST_ACTIVE -- nothing else is tracked --: a photon that reaches census keeps its status word)
        if (status != ST_ACTIVE) swarm_st<NT_ST>(&g1(S.status)[n], status);
        swarm_st<NT_ST>(&g1(S.rng)[n], rng.s);
        ls = DS_IDLE;
      }
#ifdef JB_DDMC_EXP_SEQSTORE
      if (phase == 0) exp_wr_cur += 64;
#endif
      {
        const int n_done = __popcll(__ballot(was_done));
        const int n_census = __popcll(__ballot(was_done && status == ST_ACTIVE));
        const int n_abs = __popcll(__ballot(was_done && status == ST_ABSORBED));
        const int n_esc = __popcll(__ballot(was_done && status == ST_ESCAPED));
        c_census += n_census; c_abs += n_abs; c_esc += n_esc;
        c_out += n_done - n_census - n_abs - n_esc;
      }
      // ---- a particle that sits at a face of its cell (just loaded, or just relocated): handed over to k_hybrid
      //      as it stands
      if (ls == DS_PARK) {
        if (!fresh) {
          swarm_st<NT_ST>(&g1(S.blk)[n], b);
          swarm_st<NT_ST>(&g1(S.t)[n], t);
          swarm_st<NT_ST>(&g1(S.x)[n], x); swarm_st<NT_ST>(&g1(S.y)[n], y); swarm_st<NT_ST>(&g1(S.z)[n], z);
          swarm_st<NT_ST>(&g1(S.vx)[n], vx); swarm_st<NT_ST>(&g1(S.vy)[n], vy); swarm_st<NT_ST>(&g1(S.vz)[n], vz);
          swarm_st<NT_ST>(&g1(S.rng)[n], rng.s);
        }
        const unsigned long long pm = __ballot(true);
        const int leader = __ffsll((long long)pm) - 1;
        unsigned long long base = 0ull;
        if (lane == leader) base = atomicAdd(g1(A.park_count), (unsigned long long)__popcll(pm));
        base = __shfl(base, leader, 64);
        g1(A.park_list)[base + (unsigned long long)__popcll(pm & ((1ull << lane) - 1ull))] = (unsigned)n;
        ls = DS_IDLE;
      }
      // ---- ... and the ones that go (back) into the event loop: onto the READY queue
      {
        const unsigned long long vm = __ballot(ls == DS_VIRT);
        if (vm != 0ull) {
          if (ls == DS_VIRT) {
            const int q = ready_cnt + below(vm);
            rd_nrec[q] = (unsigned long long)(unsigned)n | ((unsigned long long)rec << 32);
            rd_t[q] = t;
            rd_rng[q] = rng.s;
          }
          ready_cnt += __popcll(vm);
        }
      }
    }

    JB_QT(1)
#ifdef JB_TIMING
    ++c_epi;
#endif
    // ================================ EVENTS =================================
    retire_refill();
    if (__ballot(r_ls == DS_VIRT) == 0ull) {
      // (nothing to follow: finished histories that found no room, photons still to be loaded, or the end)
      if (__ballot(r_ls >= DS_DONE) != 0ull || done_cnt > 0 || ready_cnt > 0 || more) continue;
      break;
    }
    int waste = 0;
    // (nothing of the service phase is left in flight when the loop starts: see k_ddmc_all; with the codes in LDS
    // the loop never waits for the vector-memory counter, and the service phase's stores drain behind it)
    if constexpr (!LCODES) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    code = load_code(r_rec);
    for (;;) {
      ++c_pass;
#ifdef JB_TIMING_LOOP
      const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
      // One DDMC step per running lane (transport_utils.hpp:184-263 on the virtual state): k_ddmc_all's pass for
      // cell codes, word for word -- what does not depend on the cell first, then the code requested at the end of
      // the pass before, everything committed through selects.
      const bool run = r_ls == DS_VIRT;
      const unsigned long long s0 = r_rng;
      const unsigned long long s1 = s0 * kLcgMul + kLcgInc;
      const unsigned long long s2 = s0 * kMul2 + kInc2;
      const unsigned long long s4 = s0 * kMul4 + kInc4;
      const double u2 = u52_to_double(s2 >> 12);
      const double nlog = -m_log(u52_to_double(s1 >> 12));
      __builtin_amdgcn_sched_barrier(0);
#ifdef JB_TIMING_LOOP
      const unsigned long long tl1 = __builtin_readcyclecounter();
      __builtin_amdgcn_s_waitcnt(0x0f70);
      const unsigned long long tl2 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
      const unsigned cd = code;
      const bool ghost = (int)cd < 0;
      // (a ghost cell has no record: any row will do for the arithmetic nobody commits)
      const v4d *rp = (const v4d *)(lds_cls + 8u * (ghost ? 0u : cd));
      const v4d r0 = rp[0];
      const v4d r1 = rp[1];
      DdmcStepRec r;
      r.ffaa = r0.x; r.c1 = r0.y; r.c2 = r0.z; r.c3 = r0.w;
      r.c4 = r1.x; r.c5 = r1.y; r.leak_tot = r1.z; r.rcp = r1.w;
      const bool gl = run && ghost;
      // (events of the pass: the running lanes, less those that sit it out below)
      c_ev += (unsigned int)__popcll(__ballot(run));
      if (const unsigned long long gm = __ballot(gl); gm != 0ull) {   // a leak through a block face: the particle sits this pass out (k_ddmc_all)
        c_ev -= (unsigned int)__popcll(gm);
        const bool tab = gl && (cd & kCodeTable) != 0u;
        r_rec = tab ? cd & kCodeRecMask : r_rec;
        if constexpr (multi_d) r_pd = tab ? kPdZero : r_pd;
        if constexpr (NDIM == 1) r_mir ^= (tab && (cd & kCodeMirror) != 0u) ? 1u : 0u;
        r_ls = (gl && !tab) ? DS_RELOC : r_ls;
      }
      const bool live = run && !ghost;
      // transport_utils.hpp:184-191
      const double a2 = r.ffaa + r.leak_tot;
      const double cdf_ddmc = a2 + DBL_MIN;
      const double dt_ddmc = m_div_r(nlog, vv * cdf_ddmc, r.rcp);
      const double dt_end = t_end - r_t;
      const bool ev = dt_ddmc < dt_end;
      const double t_new = r_t + dmin(dt_ddmc, dt_end);
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
      r_t = live ? t_new : r_t;
      r_rng = live ? (leak ? s4 : (ev ? s2 : s1)) : r_rng;
      r_pzs = leak ? s2 : r_pzs;
      r_pd = leak ? delta : r_pd;
      if constexpr (NDIM == 1) r_mir = leak ? 0u : r_mir;   // (a new leak: a new direction)
      if constexpr (NDIM == 3) r_rec = leak ? r_rec + (unsigned)delta : r_rec;
      else r_rec = (leak && delta != kPdStay) ? r_rec + (unsigned)delta : r_rec;
      r_ls = live ? ((ev && absorbed) ? DS_ABS : (done ? (ev ? DS_DONE : DS_CENSUS) : DS_VIRT)) : r_ls;
      // ---- histories that have ended leave for the DONE queue and their lanes take the next photons -- once
      //      kRetireMin lanes stand still (the ~40 instructions of that exchange are the same for one lane and for
      //      eight: a lane that waits a pass for company costs 1/64 of a pass)
#ifdef JB_TIMING_LOOP
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tl3 = __builtin_readcyclecounter() + (r_ls == 99 ? 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
      unsigned long long vm = __ballot(r_ls == DS_VIRT);
      if (kRetireMin <= 1 ? vm != ~0ull : 64 - __popcll(vm) >= kRetireMin) {
        retire_refill();
        vm = __ballot(r_ls == DS_VIRT);
      }
      // the code of the cell every lane is in now, for the next pass (32-bit byte offset from a scalar base: the
      // codes of < 2^29 cells span < 2 GiB)
      // (a lane without a running photon reads the code of the cell its last one was in: nobody looks at it)
      code = load_code(r_rec);
      __builtin_amdgcn_sched_barrier(0);
      // leave the loop: a finished history found no room (the DONE queue is full: a whole batch for the service
      // phase), nothing runs, or lanes stand idle and there are photons to be loaded (at the tail of the launch:
      // once they have idled long enough).  The common case -- room, and all 64 lanes running -- first.
#ifdef JB_TIMING_LOOP
      {
        const unsigned long long tl4 = __builtin_readcyclecounter();
        cyc_lp[0] += tl1 - tl0; cyc_lp[1] += tl2 - tl1; cyc_lp[2] += tl3 - tl2; cyc_lp[3] += tl4 - tl3;
      }
#endif
      if (done_cnt == kQDone) break;
      if (vm == ~0ull) continue;
      if (vm == 0ull) break;
      if (ready_cnt == 0) {
        if (more) break;
        waste += 64 - __popcll(vm);
        if (waste >= JB_DDMC_Q_BUDGET && done_cnt > 0) break;
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
  // the launch's counters: summed over the workgroup in LDS first (every wave of the launch ends at about the same
  // time, and atomics on ONE line of device memory are served one after the other: 7 per wave were 28,000)
  __shared__ unsigned long long lds_cnt[8];
  __syncthreads();   // (the queues are dead: every wave is past its last pass)
  if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  if (lane == 0) {
    if (c_census) atomicAdd(&lds_cnt[CNT_CENSUS], (unsigned long long)c_census);
    if (c_abs) atomicAdd(&lds_cnt[CNT_ABSORBED], (unsigned long long)c_abs);
    if (c_esc) atomicAdd(&lds_cnt[CNT_ESCAPED], (unsigned long long)c_esc);
    if (c_out) atomicAdd(&lds_cnt[CNT_OUTGOING], (unsigned long long)c_out);
    if (c_ev) atomicAdd(&lds_cnt[CNT_EVENTS], c_ev);
    atomicAdd(&lds_cnt[CNT_PASSES], (unsigned long long)c_pass);
    atomicAdd(&lds_cnt[CNT_SERVICE], (unsigned long long)c_service);
  }
  __syncthreads();
  if (threadIdx.x < 8 && threadIdx.x != CNT_UNFINISHED && lds_cnt[threadIdx.x] != 0ull)
    atomicAdd(&counters[threadIdx.x], lds_cnt[threadIdx.x]);
  if (lane == 0) {
#ifdef JB_TIMING   // (the words jb_api.hip prints as "reloc claim done take real | episodes passes services": here
    // DONE batches, fills, event loop, - , - | loop entries, passes, batches)
    atomicAdd(&counters[24], cyc_ph[0] >> 10);
    atomicAdd(&counters[25], cyc_ph[1] >> 10);
    atomicAdd(&counters[26], cyc_ph[2] >> 10);
#ifdef JB_TIMING_LOOP   // (event loop: top -> code wanted | waiting for the code | step | retire / refill + tail)
    atomicAdd(&counters[20], cyc_lp[0] >> 10);
    atomicAdd(&counters[21], cyc_lp[1] >> 10);
    atomicAdd(&counters[22], cyc_lp[2] >> 10);
    atomicAdd(&counters[23], cyc_lp[3] >> 10);
#endif
    atomicAdd(&counters[29], (unsigned long long)c_epi);
    atomicAdd(&counters[30], (unsigned long long)c_pass);
    atomicAdd(&counters[31], (unsigned long long)c_service);
#endif
  }
#undef JB_QT
}

}  // namespace jb
