// jb_kernels.hpp -- HIP kernels of the history loop for gfx950 (wave64).
//
// Kernel              reference launch site it replaces
// k_fleck             "UpdateDerivedTransportFields::Fleck-Factor"   jaybenne.cpp:304-316
// k_face_prob<D>      "...::X{1,2,3}-DDMC-Prob"                      jaybenne.cpp:336-487
// k_source_count      "SourcePhotons1"                               sourcing.cpp:73-119
// k_source_fill       "SourcePhotons2"                               sourcing.cpp:141-205
// k_transport         "TransportPhotons" / "TransportPhotons_DDMC"   transport.cpp:67-174,
//                     + PhotonReflectBC, swarm send/receive to local  transport_ddmc.cpp:69-230,
//                     blocks, SampleDDMCBlockFace, CheckCompletion    boundaries.hpp:36-83,
//                     and (optionally) FillEnergyTally, fused         jaybenne.cpp:547-561
// k_block_face        "SampleDDMCBlockFace::2D/3D"                   sample_ddmc_bface.cpp:120-423
// k_check_completion  "CheckCompletion"                              transport.cpp:198-209
// k_zero_tally/k_tally "ZeroEnergyTally"/"FillEnergyTally"           jaybenne.cpp:540-561
// k_update_fluid      "UpdateFluid"                                  jaybenne.cpp:603-612
// k_reflect_bc        PhotonReflectBC<BFACE>                         boundaries.hpp:36-83
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "jb_device.hpp"

namespace jb {

constexpr int kBlock = 256;  // 4 waves per workgroup

// counters[]: 0 census, 1 absorbed, 2 escaped, 3 outgoing, 4 events, 5 unfinished
enum { CNT_CENSUS = 0, CNT_ABSORBED, CNT_ESCAPED, CNT_OUTGOING, CNT_EVENTS, CNT_UNFINISHED, CNT_PASSES, CNT_SERVICE, CNT_N };
constexpr int kLdsTally = 1024;  // cells (all resident blocks, ghosts included) tallied in LDS
// Heads of the 8 particle queues of the running transport launch, each in a 128-byte line of its own: the claims
// are returning atomics from every XCD, and atomics on one line are served one after the other -- with the eight
// heads in ONE line (rounds 1 - 5) and 128-slot claims, BASELINE configs[2] as shipped (12-step histories: 7.8e5
// claims per 1e8 photons) ran 10.03 ms whatever else was changed; see DESIGN.md 4.2.
constexpr int CNT_QUEUE = 1024;
constexpr int kQueueStride = 16;
constexpr int kQueues = 8;    // one per XCD (workgroups b and b + 8 share an XCD and its L2)

// lane 0's value in every lane, as a wave-uniform (scalar register) quantity
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// interior cell (b, k, j, i) of flat index c over nblocks * ncell
__device__ __forceinline__ void decode_cell(const DevMesh &M, long long c, int &b, int &k, int &j,
                                            int &i, int &cell) {
  b = (int)(c / M.ncell);
  cell = (int)(c - (long long)b * M.ncell);
  k = cell / (M.nx[0] * M.nx[1]);
  const int r = cell - k * (M.nx[0] * M.nx[1]);
  j = r / M.nx[0];
  i = r - j * M.nx[0];
  k += M.ks;
  j += M.js;
  i += M.is;
}

// -------------------------------------------------------------------------------------------
// (one wave per row of interior cells along x: the index arithmetic -- three integer divisions --
// is done per row by scalar code, the lanes stream contiguous cells)
__global__ void __launch_bounds__(kBlock) k_fleck(DevMesh M, DevParams P, double dt) {
  const long long rows = (long long)M.nblocks * M.nx[1] * M.nx[2];
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave0; r < rows; r += nwaves) {
    const long long ru = (long long)uniform_u64((unsigned long long)r);
    const int b = (int)(ru / (M.nx[1] * M.nx[2]));
    const int rem = (int)(ru - (long long)b * (M.nx[1] * M.nx[2]));
    const int k = rem / M.nx[1] + M.ks, j = rem % M.nx[1] + M.js;
    const double *rho_b = M.rho[b], *sie_b = M.sie[b];
    double *fleck_b = M.fleck[b];
    double *la_b = M.lam_abs ? M.lam_abs[b] : nullptr, *ls_b = M.lam_abs ? M.lam_sc[b] : nullptr;
    for (int i = M.is + lane; i <= M.ie; i += 64) {
      const long long q = cidx(M, k, j, i);
      const double rho = rho_b[q];
      const double temp = eos_temperature(P, rho, sie_b[q]);
      const double emis = opac_emissivity(P, rho, temp);
      const double ff = 1.0 / (1.0 + (4.0 * emis / (rho * P.cv * temp)) * dt);
      fleck_b[q] = ff;
      if (la_b) {  // gray opacities: the IMC mean free paths are per-cell constants
        double la, ls;
        imc_cell_mfp(ff, opac_absorption(P, rho, temp, 1.0), opac_scattering(P, rho, temp, 1.0), la,
                     ls);
        la_b[q] = la;
        ls_b[q] = ls;
      }
    }
  }
}

// ghost-cell codes of the mean-free-path arrays (jb_device.hpp: kGhostHi), once per mesh
__global__ void __launch_bounds__(kBlock) k_lam_ghost_codes(DevMesh M, const int *nbr_dq) {
  const long long total = (long long)M.nblocks * M.ntot;
  const int first[3] = {M.is, M.js, M.ks}, last[3] = {M.ie, M.je, M.ke};
  const unsigned stride8[3] = {8u, 8u * (unsigned)M.ni, 8u * (unsigned)(M.ni * M.nj)};
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(c / M.ntot);
    const int q = (int)(c - (long long)b * M.ntot);
    const int k = q / (M.ni * M.nj), r = q - k * (M.ni * M.nj), j = r / M.ni, i = r - j * M.ni;
    const int idx[3] = {i, j, k};
    const int lo[3] = {i < M.is, j < M.js, k < M.ks}, hi[3] = {i > M.ie, j > M.je, k > M.ke};
    const int nout = lo[0] + hi[0] + lo[1] + hi[1] + lo[2] + hi[2];
    if (nout == 0) continue;
    int bits = 0, dest = 0;            // bits: what goes into the high word next to kGhostHi
    unsigned dcell = 0u;               // cell (flat index) in block dest
    if (nout == 1) {
      int f = 0;
      for (int d = 0; d < 3; ++d) {
        if (lo[d]) f = 2 * d;
        if (hi[d]) f = 2 * d + 1;
      }
      const int n = f >> 1;
      // (only the ghost layer next to the interior is ever entered: a step moves one cell)
      const bool adjacent = (f & 1) ? idx[n] == last[n] + 1 : idx[n] == first[n] - 1;
      const int ent = M.nbr_ent[6 * b + f];
      if (adjacent && ent >= 0) {      // same-level neighbour, periodic image or reflecting wall
        bits = kGhostTable | ((ent >> 28) == 2 ? 0x10000 << n : 0);
        dest = ent & 0x0fffffff;
        // (nbr_dq: in lam_sc's layout, 16 ntot bytes per block)
        // (unsigned 32-bit arithmetic throughout: the offsets wrap modulo 2^32 by construction, jb_mesh_create)
        // the wrapped 32-bit difference, then back to a signed (small) byte count
        dcell = (unsigned)(q + (int)((unsigned)nbr_dq[6 * b + f] - 16u * (unsigned)M.ntot * (unsigned)(dest - b)) / 8);
      } else if (adjacent) {
        // a neighbour of another level?  (cell centres are half a cell from every face: the floors
        // below hold on any geometry)  centre of the ghost cell, through a periodic boundary if need be
        double xg[3];
        bool inside = true;
        for (int d = 0; d < 3; ++d) {
          const double dx = M.blk_dx[3 * b + d];
          xg[d] = M.blk_xmin[3 * b + d] + ((double)(idx[d] - first[d]) + 0.5) * dx;
          if (d >= M.ndim) continue;
          if (xg[d] < M.gmin[d]) { if (M.bc[2 * d] == BC_PERIODIC) xg[d] += M.gmax[d] - M.gmin[d]; else inside = false; }
          if (xg[d] > M.gmax[d]) { if (M.bc[2 * d + 1] == BC_PERIODIC) xg[d] -= M.gmax[d] - M.gmin[d]; else inside = false; }
        }
        int li = -1;
        if (inside) {
          long long l[3] = {0, 0, 0};
          for (int d = 0; d < M.ndim; ++d) {
            long long v = (long long)floor((xg[d] - M.gmin[d]) * M.inv_leaf_len[d]);
            l[d] = v < 0 ? 0 : (v > M.nleaf[d] - 1 ? M.nleaf[d] - 1 : v);
          }
          li = M.local_index[M.leaf_map[(l[2] * M.nleaf[1] + l[1]) * M.nleaf[0] + l[0]]];
        }
        if (li >= 0 && M.blk_level[li] == M.blk_level[b] - 1) {
          // COARSER: the coarse cell that contains the ghost cell; per axis: is the fine centre below the coarse one?
          bits = kGhostTable | kGhostCoarser;
          int cc[3] = {M.is, M.js, M.ks};
          for (int d = 0; d < M.ndim; ++d) {
            const double dxc = M.blk_dx[3 * li + d];
            cc[d] = (int)floor((xg[d] - M.blk_xmin[3 * li + d]) / dxc) + first[d];
            const double centre = M.blk_xmin[3 * li + d] + ((double)(cc[d] - first[d]) + 0.5) * dxc;
            if (xg[d] < centre) bits |= 0x400 << d;
          }
          dest = li;
          dcell = (unsigned)((cc[2] * M.nj + cc[1]) * M.ni + cc[0]);
        } else if (li >= 0 && M.blk_level[li] == M.blk_level[b] + 1) {
          // FINER: the lowest of the fine cells behind the ghost cell, in the layer next to the face
          bits = kGhostTable | kGhostFiner | (n << 10) | ((f & 1) ? 0 : 0x1000);
          int cc[3] = {M.is, M.js, M.ks};
          for (int d = 0; d < M.ndim; ++d) {
            const double dxf = M.blk_dx[3 * li + d];
            if (d == n) cc[d] = (f & 1) ? first[d] : last[d];
            else cc[d] = (int)floor((xg[d] - 0.5 * dxf - M.blk_xmin[3 * li + d]) / dxf) + first[d];
          }
          dest = li;
          dcell = (unsigned)((cc[2] * M.nj + cc[1]) * M.ni + cc[0]);
        }
      }
    }
    const unsigned ntot = (unsigned)M.ntot;
    if (M.ddmc_step != nullptr) {
      // ... and of the DDMC step records (jb_kernel_ddmc.hpp: kStepGhostTable): the record number of the
      // cell a particle that leaked into this ghost cell is in -- same-size resident neighbour, periodic
      // image, or (more than one dimension) the cell it came from at a reflecting wall
      int sflag = 0;
      unsigned srec = 0u;
      if (nout == 1) {
        int f = 0;
        for (int d = 0; d < 3; ++d) {
          if (lo[d]) f = 2 * d;
          if (hi[d]) f = 2 * d + 1;
        }
        const int n = f >> 1;
        const bool adjacent = (f & 1) ? idx[n] == last[n] + 1 : idx[n] == first[n] - 1;
        const int ent = M.nbr_ent[6 * b + f];
        const bool wall = (ent >> 28) == 2;
        if (adjacent && ent >= 0) {
          const int stride = n == 0 ? 1 : (n == 1 ? M.ni : M.ni * M.nj);
          const int back = wall ? stride : stride * M.nx[n];
          // kStepGhostTable; + kStepGhostMirror at a reflecting wall in 1-D (the direction travels there
          // and is mirrored; in more dimensions it is zeroed by the crossing anyway)
          sflag = (wall && M.ndim == 1) ? 3 : 1;
          srec = (unsigned)(ent & 0x0fffffff) * ntot + (unsigned)(q + ((f & 1) ? -back : back));
        }
      }
      M.ddmc_step[8 * ((long long)b * M.ntot + q) + 7] = __hiloint2double((int)(0x80000000u | (unsigned)sflag), (int)srec);
      // ... and the same as a cell code (DevMesh::ddmc_code)
      if (M.ddmc_code != nullptr)
        M.ddmc_code[(long long)b * M.ntot + q] =
            kCodeGhost | ((sflag & 1) ? kCodeTable : 0u) | ((sflag & 2) ? kCodeMirror : 0u) | (srec & kCodeRecMask);
    }
    M.lam_sc[b][q] = __hiloint2double(kGhostHi | bits, (int)(16u * ntot * (unsigned)dest + 8u * dcell));
    M.lam_abs[b][q] = 1.0;
    if (M.lam_hyb != nullptr)  // (the destination block in the low 8 bits: <= kLdsBlocks resident blocks there)
      M.lam_hyb[(long long)b * M.ntot + q] =
          dest < 256 ? __hiloint2double(kGhostHi | bits | dest, (int)(8u * ntot * (unsigned)dest + 8u * dcell))
                     : __hiloint2double(kGhostHi, 0);
    (void)stride8;
  }
}

template <int D>
__global__ void __launch_bounds__(kBlock) k_face_prob(DevMesh M, DevParams P) {
  // faces normal to D: interior cells plus one extra layer in D
  const int n0 = M.nx[0] + (D == 0), n1 = M.nx[1] + (D == 1), n2 = M.nx[2] + (D == 2);
  const long long per_block = (long long)n0 * n1 * n2;
  const long long total = per_block * M.nblocks;
  double *const *F = (D == 0) ? M.P1 : (D == 1 ? M.P2 : M.P3);
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(c / per_block);
    long long r = c - (long long)b * per_block;
    const int k = (int)(r / ((long long)n0 * n1)) + M.ks;
    r -= (long long)(k - M.ks) * n0 * n1;
    const int j = (int)(r / n0) + M.js;
    const int i = (int)(r - (long long)(j - M.js) * n0) + M.is;
    const double dxd = M.blk_dx[3 * b + D];
    const double rlev = (double)M.blk_level[b];
    const double rlev_l = (double)M.blk_nbr_lev[6 * b + 2 * D];
    const double rlev_u = (double)M.blk_nbr_lev[6 * b + 2 * D + 1];
    const int f = (D == 0) ? i : (D == 1 ? j : k);
    const int fs = (D == 0) ? M.is : (D == 1 ? M.js : M.ks);
    const int fu = ((D == 0) ? M.ie : (D == 1 ? M.je : M.ke)) + 1;
    // std::pow(2.0, integer difference) is exact: ldexp
    const double dx_l = (f == fs) ? ldexp(1.0, (int)(rlev - rlev_l)) * dxd : dxd;
    const double dx_u = (f == fu) ? ldexp(1.0, (int)(rlev - rlev_u)) * dxd : dxd;
    const long long cl = cidx(M, k - (D == 2), j - (D == 1), i - (D == 0));
    const long long cu = cidx(M, k, j, i);
    const double rho_l = M.rho[b][cl], rho_u = M.rho[b][cu];
    const double temp_l = eos_temperature(P, rho_l, M.sie[b][cl]);
    const double temp_u = eos_temperature(P, rho_u, M.sie[b][cu]);
    const double ss_l = opac_scattering(P, rho_l, temp_l, 1.0);
    const double aa_l = opac_absorption(P, rho_l, temp_l, 1.0);
    const double ss_u = opac_scattering(P, rho_u, temp_u, 1.0);
    const double aa_u = opac_absorption(P, rho_u, temp_u, 1.0);
    double tau_l = dx_l * (ss_l + aa_l);
    double tau_u = dx_u * (ss_u + aa_u);
    tau_l = tau_l > P.tau_ddmc ? tau_l : 2.0 * kLamExt;
    tau_u = tau_u > P.tau_ddmc ? tau_u : 2.0 * kLamExt;
    F[b][cu] = 2.0 / (3.0 * (tau_l + tau_u));
  }
}

// Gray opacities + DDMC: gather what a DDMC step reads for one cell into one 64-byte record
// {f sigma_a, sigma_a + sigma_s, leak opacities P_face / dx_d of the six faces} (instead of nine
// gathers from seven arrays and six divisions per step).  Same values: the products, sums and
// quotients are the ones the step functions form, from the same operands.
// The class of one cell's step record (DevMesh::ddmc_code): its number among the DISTINCT records of this
// cycle, found or entered in a small hash table in device memory.  The lanes of a wave that hold bit-equal
// records are served together by one of them (on the gray decks a wave holds one or two distinct records,
// and the table is read, not written, by all waves but the first few).  Entering a record: the slot is
// claimed by compare-and-swap, the record and its number are written, the slot is published; a wave that
// finds a claimed slot waits for its publication (the writer waits for nobody).  Numbers beyond
// max_classes are handed out but not stored: the host sees the count and keeps the 64-byte gather.
// The numbering depends on who comes first; nothing downstream does.
// Every look-up in the table is a chain of four reads that bypass the caches (count, slot, number, record) and two
// cache invalidations (the acquire loads): ~8 us.  A wave therefore remembers the record it found last
// (ClassLast, wave-uniform: scalar registers) and asks the table only about another one -- on a uniform mesh
// once per wave and then at the faces of the block (k_ddmc_pack on 128^3 cells: 434 -> see DESIGN.md 4.2).
struct ClassLast {
  int w[16];
  int id = -1;
};

__device__ __forceinline__ int classify_step_record(const DevMesh &M, const double (&r)[8], int max_classes, ClassLast &last) {
  int *const count = M.not_all_ddmc + 1;
  int cls = -1;
  unsigned long long todo = __ballot(true);
  const int lane = threadIdx.x & 63;
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    bool same = true;
    bool known = last.id >= 0;   // (wave-uniform)
    int w[16];
    unsigned h = 0x9E3779B9u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(r[q]), leader);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(r[q]), leader);
      same = same && lo == __double2loint(r[q]) && hi == __double2hiint(r[q]);
      known = known && lo == last.w[2 * q] && hi == last.w[2 * q + 1];
      w[2 * q] = lo;
      w[2 * q + 1] = hi;
      h = (h ^ (unsigned)lo) * 0x85EBCA6Bu;
      h = (h ^ (unsigned)hi) * 0xC2B2AE35u;
    }
    h ^= h >> 15;
    if (known) {
      cls = same ? last.id : cls;
      todo &= ~__ballot(same);
      continue;
    }
    int found = -1;
    if (lane == leader) {
      if (*(volatile int *)count > max_classes) {
        found = max_classes;    // (hopeless: every cell its own record, e.g. a material with feedback)
      } else {
        unsigned slot = h % (unsigned)kClassSlots;
        for (int probe = 0; probe < kClassSlots && found < 0; ++probe, slot = (slot + 1u) % (unsigned)kClassSlots) {
          int *const st = M.ddmc_class_slot + 2 * slot;
          // (relaxed, not acquire: an agent-scope acquire is a load + an invalidation of the XCD's whole L2 -- thousands
          // of them while every other wave of the launch streams the cell data through it.  What is read behind the
          // flag is read past the caches (volatile: sc0 sc1), and only after the flag has been seen -- the branch on
          // it stands between the two --, so the writer's release (write-back, then the flag) is all that is needed)
          int s = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (s == 0) s = atomicCAS(st, 0, 1);
          if (s == 0) {           // ours: number it, store it, publish it
            const int id = atomicAdd(count, 1);
            if (id < max_classes) {
#pragma unroll
              for (int q = 0; q < 8; ++q) M.ddmc_class[8 * id + q] = r[q];
            }
            st[1] = id;
            __hip_atomic_store(st, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            found = id < max_classes ? id : max_classes;
            break;
          }
          while (s == 1) s = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int id = ((volatile int *)st)[1];
          if (id >= max_classes) { found = max_classes; break; }
          bool eq = true;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const long long a = __double_as_longlong(((volatile double *)M.ddmc_class)[8 * id + q]);
            eq = eq && a == __double_as_longlong(r[q]);
          }
          if (eq) found = id;
        }
        if (found < 0) {          // (table full of other records: count it as an overflow)
          atomicAdd(count, max_classes + 1);
          found = max_classes;
        }
      }
    }
    found = __builtin_amdgcn_readlane(found, leader);
    if (found < max_classes) {
#pragma unroll
      for (int q = 0; q < 16; ++q) last.w[q] = w[q];
      last.id = found;
    }
    cls = same ? found : cls;
    todo &= ~__ballot(same);
  }
  return cls;
}

template <int NDIM>
__global__ void __launch_bounds__(kBlock) k_ddmc_pack(DevMesh M, DevParams P, int max_classes) {
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  const long long total = (long long)M.nblocks * M.ncell;
  ClassLast last;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    int b, k, j, i, cell;
    decode_cell(M, c, b, k, j, i, cell);
    const long long q = cidx(M, k, j, i);
    const double rho = M.rho[b][q];
    const double temp = eos_temperature(P, rho, M.sie[b][q]);
    const double ff = M.fleck[b][q];
    const double ss = opac_scattering(P, rho, temp, 1.0);
    const double aa = opac_absorption(P, rho, temp, 1.0);
    // cell widths exactly as the tracking kernel forms them (upper face - lower face, faces from
    // the cell centre, transport_ddmc.cpp:139-147), so P / dx has the same operands as
    // transport_utils.hpp:175-181
    Blk B;
    load_block(M, b, B);
    const double dx = (xc(B, 0, i) + 0.5 * B.dx[0]) - (xc(B, 0, i) - 0.5 * B.dx[0]);
    const double dy = (xc(B, 1, j) + 0.5 * B.dx[1]) - (xc(B, 1, j) - 0.5 * B.dx[1]);
    const double dz = (xc(B, 2, k) + 0.5 * B.dx[2]) - (xc(B, 2, k) - 0.5 * B.dx[2]);
    double *o = M.ddmc_cell[b] + 8 * q;
    o[0] = ff * aa;
    const double sig = aa + ss;
    const bool ddmc_cell = B.dx_push * sig > P.tau_ddmc;  // transport_ddmc.cpp:135
    // (the sign of the record's sigma tells k_hybrid's DDMC loop that a photon has leaked into a
    // cell that takes IMC steps; only DDMC cells' records are read for their values)
    o[1] = ddmc_cell ? sig : -sig;
    if (!ddmc_cell) atomicOr(M.not_all_ddmc, 1);
    const double ls = M.lam_sc[b][q];  // (k_fleck, this cycle)
    M.lam_hyb[(long long)b * M.ntot + q] = ddmc_cell ? -ls : ls;
    o[2] = M.P1[b][q] / dx;
    o[3] = M.P1[b][cidx(M, k, j, i + 1)] / dx;
    o[4] = multi_d ? M.P2[b][q] / dy : 0.0;
    o[5] = multi_d ? M.P2[b][cidx(M, k, j + 1, i)] / dy : 0.0;
    o[6] = three_d ? M.P3[b][q] / dz : 0.0;
    o[7] = three_d ? M.P3[b][cidx(M, k + 1, j, i)] / dz : 0.0;
    {  // the step record (DdmcStepRec, jb_physics.hpp): ddmc_step_event's per-step sums, formed here
      double *r = M.ddmc_step + 8 * ((long long)b * M.ntot + q);
      const double c2 = o[2] + o[3];
      const double c3 = c2 + o[4];
      const double c4 = c3 + o[5];
      const double c5 = c4 + o[6];
      const double leak_tot = c5 + o[7];
      const double cdf_ddmc = o[0] + leak_tot + DBL_MIN;
      r[0] = o[0]; r[1] = o[2]; r[2] = c2; r[3] = c3; r[4] = c4; r[5] = c5; r[6] = leak_tot;
      r[7] = m_rcp_refined(P.c * cdf_ddmc);
      if (M.ddmc_code != nullptr) {
        const double rec8[8] = {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]};
        M.ddmc_code[(long long)b * M.ntot + q] = (unsigned)classify_step_record(M, rec8, max_classes, last);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// SourcePhotons phase 1: one workgroup per block.  Per cell: energy to source, stochastically
// rounded particle count (1 draw from the cell's own stream), energy weight; then an exclusive
// scan over the block's cells in (k,j,i) order (wave scan + LDS across the 4 waves + running
// carry over chunks of 256 cells).
__device__ __forceinline__ uint64_t cell_stream_id(uint32_t epoch, int gblock, int cell) {
  return ((uint64_t)epoch << 44) | ((uint64_t)(uint32_t)gblock << 24) | (uint64_t)(uint32_t)cell;
}

__global__ void __launch_bounds__(kBlock)
    k_source_count(DevMesh M, DevParams P, int source_type, double dt, double npc, uint32_t epoch,
                   int *nper_block, int *prefix) {
  __shared__ int wave_tot[kBlock / 64];
  __shared__ int carry_s;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double dv = M.blk_dx[3 * b] * M.blk_dx[3 * b + 1] * M.blk_dx[3 * b + 2];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < M.ncell; base += kBlock) {
    const int cell = base + threadIdx.x;
    int cnt = 0;
    if (cell < M.ncell && M.owned[b]) {   // halo copies source nothing: their owner does
      int k = cell / (M.nx[0] * M.nx[1]);
      const int r = cell - k * (M.nx[0] * M.nx[1]);
      int j = r / M.nx[0];
      int i = r - j * M.nx[0];
      k += M.ks; j += M.js; i += M.is;
      const long long q = cidx(M, k, j, i);
      LcgRng rng(rng_seed_state(P.key0, kRngDomainCell, cell_stream_id(epoch, M.gid[b], cell)));
      const double rho = M.rho[b][q];
      const double temp = eos_temperature(P, rho, M.sie[b][q]);
      double erad;
      if (source_type == 0) {  // thermal: (4 sb / c) T^4 dV   (sourcing.cpp:93)
        const double t2 = temp * temp;
        erad = (4.0 * P.sb / P.c) * (t2 * t2) * dv;
      } else {  // emission: f j dV dt   (sourcing.cpp:95-96)
        erad = M.fleck[b][q] * opac_emissivity(P, rho, temp) * dv * dt;
      }
      double snpc = floor(npc);
      snpc += (double)((npc - snpc) > rng.drand());
      M.src_num[b][q] = snpc;
      M.src_ew[b][q] = erad / snpc;
      cnt = (int)rint(snpc);
    }
    // inclusive scan within the wave
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wv; ++w) wave_off += wave_tot[w];
    const int carry = carry_s;
    if (cell < M.ncell) prefix[(long long)b * M.ncell + cell] = carry + wave_off + incl - cnt;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) nper_block[b] = carry_s;
}

// SourcePhotons phase 2: one thread per NEW PARTICLE (the reference runs one thread per cell with
// a serial loop over that cell's particles, sourcing.cpp:167-202).  blk_first[b] = number of new
// particles in blocks < b; the cell is found by bisection in the block's prefix array.
__global__ void __launch_bounds__(kBlock)
    k_source_fill(DevMesh M, DevParams P, DevSwarm S, int source_type, double t_start, double dt,
                  const int *prefix, const long long *blk_first, const long long *slot_base,
                  const unsigned long long *id_base, const long long *ord_first, long long total) {
  load_math_tables();
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = M.nblocks - 1;  // last b with blk_first[b] <= g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (blk_first[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const int b = lo;
    // (ord_first: this call sources the block's photons number ord_first[b] ... only -- a rank of a
    // replicated-mesh run takes its share of every block, jb_source_photons_fill_range)
    const int nq = (int)(g - blk_first[b]);
    const int np = nq + (ord_first ? (int)ord_first[b] : 0);
    const int *pf = prefix + (long long)b * M.ncell;
    lo = 0; hi = M.ncell - 1;  // last cell with pf[cell] <= np (empty cells share a prefix value:
                               // the LAST of them that still satisfies <= is the non-empty one)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pf[mid] <= np) lo = mid; else hi = mid - 1;
    }
    const int cell = lo;
    int k = cell / (M.nx[0] * M.nx[1]);
    const int r = cell - k * (M.nx[0] * M.nx[1]);
    int j = r / M.nx[0];
    int i = r - j * M.nx[0];
    k += M.ks; j += M.js; i += M.is;
    const long long q = cidx(M, k, j, i);
    Blk B;
    load_block(M, b, B);
    const long long n = slot_base[b] + nq;
    const uint64_t id = id_base[b] + (uint64_t)np;
    LcgRng rng(rng_stream_start(P.key0, id));
    S.ip[n] = i; S.jp[n] = j; S.kp[n] = k;
    S.blk[n] = b;
    S.status[n] = ST_ACTIVE;
    // draw order of sourcing.cpp:175-198: x, y, z; theta, phi; Planck (5); emission: time
    S.x[n] = xc(B, 0, i) + B.dx[0] * (rng.drand() - 0.5);
    S.y[n] = xc(B, 1, j) + B.dx[1] * (rng.drand() - 0.5);
    S.z[n] = xc(B, 2, k) + B.dx[2] * (rng.drand() - 0.5);
    const double theta = m_acos(2.0 * rng.drand() - 1.0);
    const double xi_phi = rng.drand();  // phi = 2 pi xi
    double sth, cth, sph, cph;
    m_sincos(theta, sth, cth);
    m_sincos2pi(xi_phi, sph, cph);
    S.vx[n] = P.c * sth * cph;
    S.vy[n] = P.c * sth * sph;
    S.vz[n] = P.c * cth;
    const double rho = M.rho[b][q];
    const double temp = eos_temperature(P, rho, M.sie[b][q]);
    S.e[n] = sample_planck_energy(rng, P.sb, temp);
    S.w[n] = M.src_ew[b][q];
    if (source_type == 1) {
      S.t[n] = t_start + rng.drand() * dt;
    } else {
      S.t[n] = 0.0;
    }
    S.id[n] = id;
    S.rng[n] = rng.s;
  }
}

// energy_delta = -(sum of the cell's new weights), subtracting one weight at a time like the
// reference's serial loop (sourcing.cpp:165-166,196); zero for the thermal source.
__global__ void __launch_bounds__(kBlock) k_source_edelta(DevMesh M, int source_type) {
  const long long total = (long long)M.nblocks * M.ncell;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    int b, k, j, i, cell;
    decode_cell(M, c, b, k, j, i, cell);
    if (!M.owned[b]) continue;
    const long long q = cidx(M, k, j, i);
    double dej = 0.0;
    if (source_type == 1) {
      const int n = (int)rint(M.src_num[b][q]);
      const double ew = M.src_ew[b][q];
      for (int p = 0; p < n; ++p) dej -= ew;
    }
    M.edelta[b][q] = dej;
  }
}

// -------------------------------------------------------------------------------------------
// The history loop.  One lane follows one particle from its state at t_start to census /
// absorption / escape / departure to another rank.
//
// Structure.  A wave alternates between an EVENT loop -- every running lane takes one tracking
// step per pass; nothing else is in it -- and a SERVICE phase that does everything that happens
// once per history or once per block crossing: relocation of particles that left their block
// (swarm boundary conditions, destination-block lookup, DDMC block-face resampling), DDMC census
// resampling, write-back, tallies, and handing the next particles to idle lanes.  The event loop
// runs until enough lanes have left it (see kServiceBudget); keeping the rare, long code paths
// out of it keeps its 64 lanes on one instruction stream.
//
// Particles are dealt from 8 queues, each over one contiguous eighth of the (cell-ordered) swarm
// (ballot + popcount prefix over the idle mask, one returning atomic per wave).  A workgroup
// starts on queue blockIdx % 8 -- the workgroups that share an XCD, hence an L2 -- and moves on
// when its queue is drained, so the particles in flight on one XCD form one short contiguous
// window of the swarm and the cell data they gather stays in that XCD's 4 MiB L2.  Placement only
// affects speed, never results.
#ifndef JB_TRANSPORT_WAVES_PER_SIMD
#define JB_TRANSPORT_WAVES_PER_SIMD 1
#endif
#ifndef JB_LEAN_WAVES_PER_SIMD   // the lean kernels of a non-absorbing material (the stepdiff decks) are
                                 // held to three waves per SIMD: 168 registers, no spill (tests/test_cabi.py)
#define JB_LEAN_WAVES_PER_SIMD 3
#endif
#ifndef JB_SERVICE_BUDGET        // idle lane-passes that buy a service phase: IMC kernels
#define JB_SERVICE_BUDGET 96
#endif
#ifndef JB_SERVICE_BUDGET_DDMC   // ... DDMC / hybrid kernels (their service phase costs more)
#define JB_SERVICE_BUDGET_DDMC 128
#endif

// LS_DONE_RAW (lean kernels): finished without having been tracked -- t, v still hold what was loaded
enum { LS_IDLE = 0, LS_RUN = 1, LS_DONE = 2, LS_RELOC = 3, LS_DONE_RAW = 4 };

#ifndef JB_DDMC_WAVES_PER_SIMD
#define JB_DDMC_WAVES_PER_SIMD 3
#endif
// GRAY: 0 = opacities evaluated per event from rho, sie (frequency dependent in general);
// 1 = gray: per-cell mean free paths / DDMC records precomputed by UpdateDerivedTransportFields;
// 2 = gray and no absorption opacity at all (see imc_step_core; in a DDMC kernel it applies to
//     the IMC steps of a hybrid deck).
// EXACT (gray IMC kernels only): every resident block has power-of-two cell widths and a lower
// corner that is a whole number of them (checked by jb_mesh_create; true of all the stepdiff
// decks).  Then x0 + (i + 0.5) dx -+ 0.5 dx and upper - lower are exact, so the cell faces are
// formed as fma(i, dx, x0) and + dx, and the nudge width as eps dx: the same doubles as the
// general formulas (transport.cpp:114-119), in 3 instead of 8 operations per axis.
// LEAN (gray IMC kernels only): lean arithmetic in the tracking step (imc_step_fast): within 2^-48
// (relative) per operation of the exact variant, ~8 % fewer instructions; jb_set_arithmetic picks.
// the kernel's argument list as the kernel-argument segment holds it (natural alignment, in order)
struct TransportArgs {
  DevMesh M;
  DevParams P;
  DevSwarm S;
  double t_start, dt;
  long long first, last;
  unsigned long long *counters;
  const int *skip_unless;
};
template <int NDIM, bool DDMC, bool TALLY, int GRAY, bool EXACT = false, bool LEAN = false>
__global__ void
__launch_bounds__(kBlock, DDMC ? JB_DDMC_WAVES_PER_SIMD
                            : (LEAN && GRAY == 2 ? JB_LEAN_WAVES_PER_SIMD : JB_TRANSPORT_WAVES_PER_SIMD))
    k_transport(DevMesh M, DevParams, DevSwarm, double t_start, double dt, long long first,
                long long last, unsigned long long *counters, const int *skip_unless) {
  static_assert(!EXACT || (GRAY != 0 && !DDMC), "EXACT is a variant of the gray IMC kernels");
  static_assert(!LEAN || (GRAY != 0 && !DDMC), "LEAN is a variant of the gray IMC kernels");
  // The swarm view and the parameters are read where they are used (the service phase), from the
  // kernel-argument segment, not through the parameters (see k_ddmc_all); the mesh view, which the
  // event loop reads, stays in scalar registers.
  const TransportArgs &A = *(const TransportArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const DevParams &P = A.P;
  const DevSwarm &S = A.S;
  // (gray DDMC launches come in pairs: k_ddmc_all runs when every cell is a DDMC cell, this
  // kernel when *skip_unless says otherwise)
  if (skip_unless != nullptr && *skip_unless == 0) return;
  // Small meshes (the reference's 1-D decks: ~1e2 cells under 1e5..1e8 particles): the census
  // tally of every workgroup goes to LDS and is flushed once at the end, instead of 1e8 global
  // atomics contending for a handful of cache lines.
  __shared__ double lds_tally[TALLY ? kLdsTally : 1];
  const bool tally_in_lds = TALLY && (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  if constexpr (TALLY) {
    if (tally_in_lds)
      for (int q = threadIdx.x; q < M.nblocks * (int)M.ntot; q += blockDim.x) lds_tally[q] = 0.0;
  }
  // DDMC kernels re-read the block geometry at the top of every event pass (below): from LDS
  __shared__ std::conditional_t<DDMC, LdsBlockTable, int> blk_tab;
  if constexpr (DDMC) fill_block_table(M, blk_tab);
  // gray IMC kernels: the face-crossing table (destination block and the coordinate that changes
  // per face of a resident block) in LDS -- a crossing lane's two dependent look-ups then cost an
  // LDS round trip instead of two trips to L2 that the whole wave waits for
  // (256 resident blocks: BASELINE configs[1] on 8 GPUs holds 64 owned blocks + 80 halo copies per rank)
  constexpr int kLdsFaceBlocks = 256;
  struct FaceTable { int ent[kLdsFaceBlocks][6]; double x0[kLdsFaceBlocks][6]; };
  __shared__ std::conditional_t<(GRAY != 0 && !DDMC), FaceTable, int> face_tab;
  const bool faces_in_lds = M.nblocks <= kLdsFaceBlocks;
  if constexpr (GRAY != 0 && !DDMC) {
    if (faces_in_lds)
      for (int q = threadIdx.x; q < 6 * M.nblocks; q += blockDim.x) {
        (&face_tab.ent[0][0])[q] = M.nbr_ent[q];
        (&face_tab.x0[0][0])[q] = M.nbr_x0[q];
      }
  }
  load_math_tables();  // (ends with a barrier)
  constexpr bool multi_d = NDIM >= 2, three_d = NDIM == 3;
  constexpr bool kFastGray = GRAY != 0 && !DDMC;
  constexpr bool kNoAbs = GRAY == 2;
  // lean arithmetic: a lane carries the unit direction v / c in (vx, vy, vz) and the distance
  // left to census c (t_end - t) in t while it follows a photon (imc_step_dir)
  constexpr bool kDir = LEAN;
  constexpr bool kPackedDdmc = GRAY != 0 && DDMC;
  // DDMC kernels re-read the block geometry (10 cached doubles) at the top of every event pass
  // instead of carrying it in 26 registers per lane across the whole loop: that is what lets
  // three waves instead of two share a SIMD.
#ifndef JB_RELOAD_BLOCK
#define JB_RELOAD_BLOCK DDMC
#endif
  constexpr bool kReloadBlock = JB_RELOAD_BLOCK;
  constexpr int kServiceBudget = DDMC ? JB_SERVICE_BUDGET_DDMC : JB_SERVICE_BUDGET;
  const double vv = P.c;
  const double t_end = t_start + dt;  // the reference re-evaluates t_start + dt: same double
  const int lane = threadIdx.x & 63;
  unsigned long long *queue = counters + CNT_QUEUE;  // zeroed by the host before the launch
  const long long per_q = (last - first + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues;  // wave-uniform: queue this wave draws from
  int tried = 0;                   // queues found drained so far
  bool more = true;                // wave-uniform: some queue may still hold particles
  // DDMC: slots [chunk_pos, chunk_end) of the swarm are this wave's to deal out (wave-uniform);
  // short DDMC histories claim 128 at a time (one contended atomic per ~5 service phases); larger
  // chunks lengthen the tail of a hybrid launch whose IMC histories are ~1e3 events long
#ifndef JB_DDMC_CHUNK
#define JB_DDMC_CHUNK 128
#endif
  constexpr long long kChunk = JB_DDMC_CHUNK;
  long long chunk_pos = 0, chunk_end = 0;

  unsigned int c_census = 0, c_abs = 0, c_esc = 0, c_out = 0;
  // wave-level (scalar): events = running lanes summed over the passes, passes, service phases
  unsigned long long c_ev = 0;
  unsigned int c_pass = 0, c_service = 0;

  // lane state
  int ls = LS_IDLE;
  bool resample = false;  // DDMC particle reached census: position / direction to be resampled
  long long n = 0;
  LcgRng rng(0);
  int b = 0, ip = 0, jp = 0, kp = 0, status = ST_ACTIVE;
  double t = 0, x = 0, y = 0, z = 0, vx = 0, vy = 0, vz = 0, ee = 0;
  Blk B;
  gcptr f0 = nullptr, f1 = nullptr, f2 = nullptr;  // this block's cell arrays
  // EXACT gray IMC kernels: the mean-free-path arrays of all resident blocks span < 4 GiB
  // (jb_mesh_create), so a lane keeps the byte offset of its block's pair of arrays (32 bits)
  // instead of two pointers and gathers through scalar base + 32-bit vector offset
  constexpr bool kOff32 = GRAY != 0 && !DDMC && EXACT;
  unsigned lam_off = 0u;
  const char *const lam_abs0 = (const char *)M.lam_base;
  const char *const lam_sc0 = (const char *)(M.lam_base + M.ntot);
  double fd[3] = {0.0, 0.0, 0.0};                  // EXACT: the block's nudge widths
  // deferred direction of the last DDMC leak (packed-record DDMC kernels; see ddmc_step_event)
  int pend = -1;
  double pz1 = 0.0, pz2 = 0.0;
  // hybrid kernels: lam_hyb of the lane's cell (its sign picks the IMC or the DDMC step), requested
  // as soon as the cell is known -- at the end of the previous pass -- so that the choice at the
  // top of a pass does not wait for a gather
  double lam_cur = 0.0;

  auto bind_arrays = [&](int blk) {
    if constexpr (kOff32) {
      lam_off = (unsigned)(2 * blk) * ((unsigned)M.ntot * 8u);
    } else if constexpr (kFastGray) {  // (library-owned, contiguous: no pointer-table load)
      f0 = (gcptr)(M.lam_base + (long long)(2 * blk) * M.ntot);
      f1 = (gcptr)(M.lam_base + (long long)(2 * blk + 1) * M.ntot);
    } else if constexpr (kPackedDdmc) {
      f0 = (gcptr)(M.ddmc_base + (long long)(8 * blk) * M.ntot);
      f1 = (gcptr)(M.lam_base + (long long)(2 * blk) * M.ntot);
      f2 = (gcptr)(M.lam_hyb + (long long)blk * M.ntot);
    } else {
      f0 = (gcptr)M.rho[blk];
      f1 = (gcptr)M.sie[blk];
      f2 = (gcptr)M.fleck[blk];
    }
  };
  // gray IMC kernels: the mean free paths of the lane's cell, requested as soon as the cell is
  // known (when a photon is loaded or relocated, and at the end of a pass for the next one) so
  // that the step at the top of a pass does not wait for the gather
#ifndef JB_PREFETCH_LAM
#define JB_PREFETCH_LAM 1
#endif
  constexpr bool kPrefetch = kFastGray && JB_PREFETCH_LAM;
  double lam_a_cur = 0.0, lam_s_cur = 0.0;
  auto fetch_lam = [&]() {
    const int q = cidx(M, kp, jp, ip);
    if constexpr (kOff32) {
      const unsigned off = ((unsigned)q << 3) + lam_off;
      if constexpr (!kNoAbs) lam_a_cur = *(gcptr)(lam_abs0 + off);
      lam_s_cur = *(gcptr)(lam_sc0 + off);
    } else {
      if constexpr (!kNoAbs) lam_a_cur = f0[q];
      lam_s_cur = f1[q];
    }
  };
  auto bind_block = [&](int blk) {
    load_block(M, blk, B);
    if constexpr (EXACT) {  // nudge widths eps_imc (upper - lower) = eps_imc dx, exactly
      fd[0] = kEpsImc * B.dx[0]; fd[1] = kEpsImc * B.dx[1]; fd[2] = kEpsImc * B.dx[2];
    }
    bind_arrays(blk);
  };
  auto cell_faces = [&](Step &s, const Blk &Bq) {  // transport.cpp:114-119
    if constexpr (DDMC) {
      if (M.exact) {  // (uniform) the same doubles in 3 instead of 8 operations per axis
        s.xl = m_fma((double)ip, Bq.dx[0], Bq.x0[0]); s.xu = s.xl + Bq.dx[0];
        s.yl = m_fma((double)jp, Bq.dx[1], Bq.x0[1]); s.yu = s.yl + Bq.dx[1];
        s.zl = m_fma((double)kp, Bq.dx[2], Bq.x0[2]); s.zu = s.zl + Bq.dx[2];
        return;
      }
    }
    s.xl = xc(Bq, 0, ip) - 0.5 * Bq.dx[0];
    s.xu = xc(Bq, 0, ip) + 0.5 * Bq.dx[0];
    s.yl = xc(Bq, 1, jp) - 0.5 * Bq.dx[1];
    s.yu = xc(Bq, 1, jp) + 0.5 * Bq.dx[1];
    s.zl = xc(Bq, 2, kp) - 0.5 * Bq.dx[2];
    s.zu = xc(Bq, 2, kp) + 0.5 * Bq.dx[2];
  };

  // the comm phase of the reference, for one particle in flight: boundary conditions
  // (boundaries.hpp:46-82, periodic, outflow), destination block, SampleDDMCBlockFace
  auto relocate = [&]() {
    if (!apply_swarm_bcs<NDIM>(M, x, y, z, vx, vy, vz)) {
      status = ST_ESCAPED;
      ls = LS_DONE;
    } else {
      const int g = find_block<NDIM>(M, x, y, z);
      const int li = M.local_index[g];
      if (li < 0) {  // not resident here: hand the particle to the block's owner
        status = ST_OUTGOING;
        b = g;  // global id travels in blk
        ls = LS_DONE;
      } else {
        b = li;
        bind_block(b);
        if constexpr (DDMC && multi_d)
          sample_block_face<NDIM>(M, P, B, b, rng, x, y, z, vx, vy, vz, ip, jp, kp);
        xtoijk<NDIM>(M, B, x, y, z, ip, jp, kp);  // next launch's transport.cpp:96
        ls = (kDir ? t > 0.0 : t < t_end) ? LS_RUN : LS_DONE;
        if constexpr (kPackedDdmc) lam_cur = f2[cidx(M, kp, jp, ip)];
        if constexpr (kPrefetch) fetch_lam();
      }
    }
  };

  // One face of the lane's block crossed (gray kernels; in a DDMC kernel: by an IMC step, whose
  // particle SampleDDMCBlockFace leaves alone): destination and the one geometry
  // value that changes come from the per-(block, face) table; returns false when the general
  // relocation has to run.
  auto cross_face = [&](auto axis_c, bool up, double &pos, double &vel, int &idx, int first_i,
                        int last_i) -> bool {
    constexpr int AXIS = decltype(axis_c)::value;
    const int slot = 6 * b + 2 * AXIS + (int)up;
    int ent;
    double nx0 = 0.0;  // (the DDMC kernels re-read the block geometry every pass)
    if constexpr (kFastGray) {
      if (faces_in_lds) {  // (uniform)
        ent = (&face_tab.ent[0][0])[slot];
        nx0 = (&face_tab.x0[0][0])[slot];
      } else {
        ent = M.nbr_ent[slot];
        nx0 = ((gcptr)M.nbr_x0)[slot];
      }
    } else {
      ent = M.nbr_ent[slot];
    }
    if (ent < 0) return false;
    const int kind = ent >> 28;
    bool at_first = up;  // entered through the destination's lower face
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
    if constexpr (kFastGray) B.x0[AXIS] = nx0;
    idx = at_first ? first_i : last_i;
    bind_arrays(b);
    ls = (kDir ? t > 0.0 : t < t_end) ? LS_RUN : LS_DONE;
    if constexpr (kPackedDdmc) lam_cur = f2[cidx(M, kp, jp, ip)];
    if constexpr (kPrefetch) fetch_lam();
    return true;
  };

#ifdef JB_TIMING  // scratch diagnostics: CNT_PASSES / CNT_SERVICE carry cycles / 1024 instead
  unsigned long long cyc_ev = 0, cyc_sv = 0, cyc_mark = __builtin_readcyclecounter();
#endif
  for (;;) {
    // ================================ SERVICE ================================
    ++c_service;
#ifdef JB_TIMING
    { const unsigned long long now = __builtin_readcyclecounter(); cyc_ev += now - cyc_mark; cyc_mark = now; }
#endif
    if constexpr (!kFastGray) {  // (the gray IMC kernels relocate inside their event loop)
      if (ls == LS_RELOC) relocate();
    }
    if (ls == LS_DONE || (kDir && ls == LS_DONE_RAW)) {
      // (a particle relocated to a block that is not resident carries the GLOBAL id in b: the
      // geometry tables only cover resident blocks, and nothing below reads B for it)
      if constexpr (kReloadBlock) {
        if (status != ST_OUTGOING) {
          if constexpr (DDMC) load_block(M, blk_tab, b, B);
          else load_block(M, b, B);
        }
      }
      if constexpr (kPackedDdmc) {
        if (pend >= 0 && !resample) {  // (absorbed right after a leak: the record still gets it)
          Step s;
          s.vv = vv; s.pend = pend; s.pz1 = pz1; s.pz2 = pz2;
          s.vx = vx; s.vy = vy; s.vz = vz;
          materialise_dir(s);
          vx = s.vx; vy = s.vy; vz = s.vz;
        }
        pend = -1;
      }
      if constexpr (DDMC) {
        if (resample) {  // transport_utils.hpp:265-276, once per history
          Step s;
          s.vv = vv;
          cell_faces(s, B);
          ddmc_census_resample(s, rng);
          x = s.x; y = s.y; z = s.z; vx = s.vx; vy = s.vy; vz = s.vz;
          resample = false;
        }
      }
      if ((status == ST_ACTIVE || status == ST_OUTGOING_ABSORBED) && !M.owned[b]) {
        // finished inside a halo copy: the owner of the block takes it from here (census tally
        // or absorption deposit)
        if (status == ST_ACTIVE) status = ST_OUTGOING;
        b = M.gid[b];
      }
      if constexpr (kDir) {
        if (ls == LS_DONE) {  // back to time and velocity (census: d_rem = 0 exactly, t = t_end)
          t = fma(-t, P.rc, t_end);
          vx *= vv; vy *= vv; vz *= vv;
        }
      }
      g1(S.blk)[n] = b;
      g1(S.t)[n] = t;
      g1(S.x)[n] = x; g1(S.y)[n] = y; g1(S.z)[n] = z;
      g1(S.vx)[n] = vx; g1(S.vy)[n] = vy; g1(S.vz)[n] = vz;
      g1(S.ip)[n] = ip; g1(S.jp)[n] = jp; g1(S.kp)[n] = kp;
      g1(S.status)[n] = status;
      g1(S.rng)[n] = rng.s;
      if (status == ST_ACTIVE) {
        ++c_census;
        if constexpr (TALLY) {  // jaybenne.cpp:547-561
          const double dv = B.dx[0] * B.dx[1] * B.dx[2];
          if (tally_in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + cidx(M, kp, jp, ip)], g1(S.w)[n] / dv);
          else atomicAdd(&M.tally[b][cidx(M, kp, jp, ip)], g1(S.w)[n] / dv);
        }
      } else if (status == ST_ABSORBED) {
        ++c_abs;
      } else if (status == ST_ESCAPED) {
        ++c_esc;
      } else {
        ++c_out;
      }
      ls = LS_IDLE;
    }
    if constexpr (!DDMC) {
      // hand new particles to idle lanes: one claim of exactly what they need per service phase.
      // IMC histories are ~1e3 events long: this keeps the particles in flight on an XCD one tight
      // window of the swarm and the tail of the launch short.
      const unsigned long long idle = __ballot(ls == LS_IDLE);
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
        if (ls == LS_IDLE && cand < q_last && g1(S.status)[cand] == ST_ACTIVE) {
          n = cand;
          rng.s = g1(S.rng)[n];
          b = g1(S.blk)[n];
          bind_block(b);
          t = g1(S.t)[n];
          x = g1(S.x)[n]; y = g1(S.y)[n]; z = g1(S.z)[n];
          vx = g1(S.vx)[n]; vy = g1(S.vy)[n]; vz = g1(S.vz)[n];
          ee = g1(S.e)[n];
          status = ST_ACTIVE;
          resample = false;
          xtoijk<NDIM>(M, B, x, y, z, ip, jp, kp);  // transport.cpp:96
          if constexpr (kDir) {
            if (t < t_end) {
              t = vv * (t_end - t);
              vx *= P.rc; vy *= P.rc; vz *= P.rc;
              ls = LS_RUN;
            } else {
              ls = LS_DONE_RAW;
            }
          } else {
            ls = (t < t_end) ? LS_RUN : LS_DONE;    // already at census: nothing to track
          }
          if constexpr (kPrefetch) fetch_lam();
        }
      }
    } else {
      // hand new particles to idle lanes: the wave claims kChunk consecutive slots of its queue
      // with one atomic and deals them out over the following service phases (ballot + popcount
      // prefix over the idle lanes); slots that hold no live particle leave their lane idle for
      // the next round of the loop
      unsigned long long idle = __ballot(ls == LS_IDLE);
      while (idle != 0ull && more) {
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
        const int want = __popcll(idle);
        const long long avail = chunk_end - chunk_pos;
        const int give = (long long)want < avail ? want : (int)avail;
        const int rank = __popcll(idle & ((1ull << lane) - 1ull));
        if (ls == LS_IDLE && rank < give) {
          // every field is requested before the first one is looked at: one memory latency
          const long long cand = chunk_pos + rank;
          const int st_in = g1(S.status)[cand];
          const unsigned long long rng_in = g1(S.rng)[cand];
          const int b_in = g1(S.blk)[cand];
          const double t_in = g1(S.t)[cand], x_in = g1(S.x)[cand], y_in = g1(S.y)[cand], z_in = g1(S.z)[cand];
          const double vx_in = g1(S.vx)[cand], vy_in = g1(S.vy)[cand], vz_in = g1(S.vz)[cand], e_in = g1(S.e)[cand];
          if (st_in == ST_ACTIVE) {
            n = cand;
            rng.s = rng_in;
            b = b_in;
            bind_block(b);
            t = t_in;
            x = x_in; y = y_in; z = z_in;
            vx = vx_in; vy = vy_in; vz = vz_in;
            ee = e_in;
            status = ST_ACTIVE;
            resample = false;
            xtoijk<NDIM>(M, B, x, y, z, ip, jp, kp);  // transport.cpp:96
            ls = (t < t_end) ? LS_RUN : LS_DONE;      // already at census: nothing to track
            if constexpr (kPackedDdmc) lam_cur = f2[cidx(M, kp, jp, ip)];
          }
        }
        chunk_pos += give;
        idle = __ballot(ls == LS_IDLE);
      }
    }
    const int running = __popcll(__ballot(ls == LS_RUN));
    if (running == 0) {
      if (__ballot(ls != LS_IDLE) != 0ull || more) continue;  // pending service / more to load
      break;
    }
    // Leave the event loop when the lane-passes idled away by lanes that have dropped out of it
    // add up to the price of a service phase (kServiceBudget lane-passes): with long IMC histories
    // that is after ~4 lanes have left, with short DDMC histories after ~24, and a hybrid deck
    // finds its own balance.
    int waste = 0;

    // ================================ EVENTS =================================
    // (no `continue` / `break` below: every lane must reach the ballot of the loop condition)
#ifdef JB_TIMING
    { const unsigned long long now = __builtin_readcyclecounter(); cyc_sv += now - cyc_mark; cyc_mark = now; }
#endif
    int thresh = 1;
    int nrun = running;
    while (nrun >= thresh) {
      ++c_pass;
      c_ev += (unsigned int)nrun;
      if constexpr (kFastGray) {
        if (ls == LS_RUN) {
          // per-cell mean free paths precomputed by k_fleck: two gathers instead of three, and
          // no division (same values: same operations on the same operands)
          double lam_a = 0.0, lam_s;
          if constexpr (kPrefetch) {
            lam_a = lam_a_cur; lam_s = lam_s_cur;
          } else {
            const int q = cidx(M, kp, jp, ip);
            if constexpr (kOff32) {
              const unsigned off = ((unsigned)q << 3) + lam_off;
              if constexpr (!kNoAbs) lam_a = *(gcptr)(lam_abs0 + off);
              lam_s = *(gcptr)(lam_sc0 + off);
            } else {
              if constexpr (!kNoAbs) lam_a = f0[q];
              lam_s = f1[q];
            }
          }
          bool is_absorbed, is_scattered;
          if constexpr (kDir) {
            DirGeom g;
            g.x0[0] = B.x0[0]; g.x0[1] = B.x0[1]; g.x0[2] = B.x0[2];
            g.dx[0] = B.dx[0]; g.dx[1] = B.dx[1]; g.dx[2] = B.dx[2];
            g.fd[0] = fd[0]; g.fd[1] = fd[1]; g.fd[2] = fd[2];
            imc_step_dir<NDIM, kNoAbs, EXACT>(g, B.dx_push, lam_a, lam_s, rng, t, x, y, z,
                                              vx, vy, vz, ip, jp, kp, is_absorbed, is_scattered);
            // (the nudge widths come back with the sign the direction last gave them)
            if constexpr (EXACT) { fd[0] = g.fd[0]; fd[1] = g.fd[1]; fd[2] = g.fd[2]; }
          } else {
            ImcCell c;
            if constexpr (EXACT) {
              c.xl = m_fma((double)ip, B.dx[0], B.x0[0]); c.xu = c.xl + B.dx[0];
              c.yl = m_fma((double)jp, B.dx[1], B.x0[1]); c.yu = c.yl + B.dx[1];
              c.zl = m_fma((double)kp, B.dx[2], B.x0[2]); c.zu = c.zl + B.dx[2];
              c.fdx = fd[0]; c.fdy = fd[1]; c.fdz = fd[2];
            } else {  // transport.cpp:114-119, transport_utils.hpp:151-153
              c.xl = xc(B, 0, ip) - 0.5 * B.dx[0]; c.xu = xc(B, 0, ip) + 0.5 * B.dx[0];
              c.yl = xc(B, 1, jp) - 0.5 * B.dx[1]; c.yu = xc(B, 1, jp) + 0.5 * B.dx[1];
              c.zl = xc(B, 2, kp) - 0.5 * B.dx[2]; c.zu = xc(B, 2, kp) + 0.5 * B.dx[2];
              c.fdx = kEpsImc * (c.xu - c.xl); c.fdy = kEpsImc * (c.yu - c.yl);
              c.fdz = kEpsImc * (c.zu - c.zl);
            }
            imc_step_fast<NDIM, kNoAbs, false>(c, vv, P.rc, t_end, B.dx_push, lam_a,
                                               lam_s, rng, t, x, y, z, vx, vy, vz, ip, jp, kp,
                                               is_absorbed, is_scattered);
          }
          if (!on_block(M, ip, jp, kp)) {
            ls = LS_RELOC;  // comm phase: below, for the lanes that need it
          } else if (is_absorbed) {  // transport.cpp:157-163
            if (M.owned[b]) {
              atomicAdd(&M.edelta[b][cidx(M, kp, jp, ip)], g1(S.w)[n]);
              status = ST_ABSORBED;
            } else {
              status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
            }
            ls = LS_DONE;
          } else {
            if constexpr (kPrefetch) fetch_lam();  // (for the next pass, ahead of the scatter)
            if constexpr (kDir) {
              if (is_scattered) scatter_dir(rng, vx, vy, vz);
              if (!(t > 0.0)) ls = LS_DONE;
            } else {
              if (is_scattered) scatter(rng, vv, vx, vy, vz);  // transport.cpp:165-170
              if (!(t < t_end)) ls = LS_DONE;                  // census
            }
          }
        }
        // A particle that left its block is relocated at once (in the reference: end of the
        // launch, swarm send / receive, a new launch).  On the headline workload that is 0.36 %
        // of the events, i.e. one wave pass in five has a lane here, so the common cases -- one
        // face crossed into a resident block of the same size, a periodic wrap, a reflection --
        // are served from a per-(block, face) table: destination, and the one geometry value
        // that changes.  Same arithmetic on the position as apply_swarm_bcs; the new cell index
        // is what Xtoijk gives for a particle eps_imc dx inside the face it came through (first
        // or last cell along the axis, the others unchanged).  Everything else takes relocate().
        if (__ballot(ls == LS_RELOC) != 0ull) {
          if (ls == LS_RELOC) {
            const bool xo = ip < M.is || ip > M.ie;
            const bool yo = multi_d && (jp < M.js || jp > M.je);
            const bool zo = three_d && (kp < M.ks || kp > M.ke);
            // (one lane, rarely two, gets here in a pass: each axis has its own copy of the few
            // instructions instead of selecting position / index / geometry by axis number)
            bool done = false;
            if (xo && !yo && !zo) done = cross_face(std::integral_constant<int, 0>{}, ip > M.ie, x, vx, ip, M.is, M.ie);
            else if (yo && !xo && !zo) done = cross_face(std::integral_constant<int, 1>{}, jp > M.je, y, vy, jp, M.js, M.je);
            else if (zo && !xo && !yo) done = cross_face(std::integral_constant<int, 2>{}, kp > M.ke, z, vz, kp, M.ks, M.ke);
            if (!done) relocate();
          }
        }
      } else if (ls == LS_RUN) {
        Step s;
        Blk Bl;
        if constexpr (kReloadBlock) {
          if constexpr (DDMC) load_block(M, blk_tab, b, Bl);
          else load_block(M, b, Bl);
        }
        const Blk &Bp = kReloadBlock ? Bl : B;
        s.t_start = t_start; s.dt = dt; s.vv = vv; s.rvv = P.rc; s.dx_push = Bp.dx_push;
        cell_faces(s, Bp);
        const long long q = cidx(M, kp, jp, ip);
        s.t = t; s.x = x; s.y = y; s.z = z; s.vx = vx; s.vy = vy; s.vz = vz;
        s.ip = ip; s.jp = jp; s.kp = kp;
        s.is_absorbed = false; s.is_scattered = false; s.is_rejected = false;
        s.pend = pend; s.pz1 = pz1; s.pz2 = pz2;
        bool is_ddmc_step = false;
        if constexpr (kPackedDdmc) {
          typedef double v4d __attribute__((ext_vector_type(4)));
          typedef const v4d __attribute__((address_space(1))) *grec;
          const double lam = lam_cur;  // lam_sc, sign bit set in a DDMC cell (k_ddmc_pack)
          is_ddmc_step = lam < 0.0;    // transport_ddmc.cpp:135
          if (is_ddmc_step) {
            const grec rec = (grec)(f0 + 8 * q);
            const v4d r0 = rec[0];
            const v4d r1 = rec[1];  // (same cache line; requested before r0 is looked at)
            s.ffaa = r0.x; s.sig = r0.y;
            s.Px_l = r0.z; s.Px_u = r0.w; s.Py_l = r1.x; s.Py_u = r1.y; s.Pz_l = r1.z; s.Pz_u = r1.w;
            ptcl_ddmc_albedo<NDIM, true>(s, rng);
            if (!s.is_rejected) resample = ddmc_step_event<NDIM, true, true>(s, rng);
          } else {
            if (s.pend >= 0) materialise_dir(s);  // an IMC step reads the direction
            // the fused step of the gray IMC kernels: it hands back the cell index of the new
            // position (see imc_step_fast), so these lanes skip the Xtoijk below
            ImcCell c;
            c.xl = s.xl; c.xu = s.xu; c.yl = s.yl; c.yu = s.yu; c.zl = s.zl; c.zu = s.zu;
            if (M.exact) {
              c.fdx = kEpsImc * Bp.dx[0]; c.fdy = kEpsImc * Bp.dx[1]; c.fdz = kEpsImc * Bp.dx[2];
            } else {
              c.fdx = kEpsImc * (s.xu - s.xl); c.fdy = kEpsImc * (s.yu - s.yl);
              c.fdz = kEpsImc * (s.zu - s.zl);
            }
            if (P.lean)  // (uniform) jb_set_arithmetic
              imc_step_fast<NDIM, kNoAbs, true>(c, vv, P.rc, t_end, Bp.dx_push, kNoAbs ? 0.0 : f1[q],
                                                lam, rng, s.t, s.x, s.y, s.z, s.vx, s.vy, s.vz,
                                                s.ip, s.jp, s.kp, s.is_absorbed, s.is_scattered);
            else
              imc_step_fast<NDIM, kNoAbs, false>(c, vv, P.rc, t_end, Bp.dx_push, kNoAbs ? 0.0 : f1[q],
                                                 lam, rng, s.t, s.x, s.y, s.z, s.vx, s.vy, s.vz,
                                                 s.ip, s.jp, s.kp, s.is_absorbed, s.is_scattered);
          }
        } else {
          const double rho = f0[q];
          const double temp = eos_temperature(P, rho, f1[q]);
          s.ff = f2[q];
          s.ss = opac_scattering(P, rho, temp, ee);
          s.aa = opac_absorption(P, rho, temp, ee);
          s.ffaa = s.ff * s.aa;
          s.sig = s.aa + s.ss;
          if constexpr (DDMC) is_ddmc_step = Bp.dx_push * s.sig > P.tau_ddmc;
          if (DDMC && is_ddmc_step) {
            // transport_ddmc.cpp:137-179
            s.Px_l = M.P1[b][q];
            s.Px_u = M.P1[b][cidx(M, kp, jp, ip + 1)];
            s.Py_l = multi_d ? M.P2[b][q] : 0.0;
            s.Py_u = multi_d ? M.P2[b][cidx(M, kp, jp + 1, ip)] : 0.0;
            s.Pz_l = three_d ? M.P3[b][q] : 0.0;
            s.Pz_u = three_d ? M.P3[b][cidx(M, kp + 1, jp, ip)] : 0.0;
            ptcl_ddmc_albedo<NDIM>(s, rng);
            if (!s.is_rejected) resample = ddmc_step_event<NDIM, false, false>(s, rng);
          } else {
            double lam_abs, lam_sc;
            imc_cell_mfp(s.ff, s.aa, s.ss, lam_abs, lam_sc);
            imc_step_core<NDIM, false>(s, lam_abs, lam_sc, rng);
          }
        }
        t = s.t; x = s.x; y = s.y; z = s.z; vx = s.vx; vy = s.vy; vz = s.vz;
        if constexpr (kPackedDdmc) { pend = s.pend; pz1 = s.pz1; pz2 = s.pz2; }

        if (kPackedDdmc && !is_ddmc_step) {
          ip = s.ip; jp = s.jp; kp = s.kp;
        } else {
          xtoijk<NDIM>(M, Bp, x, y, z, ip, jp, kp);  // transport.cpp:146
        }

        if (!on_block(M, ip, jp, kp)) {
          bool crossed = false;
          if constexpr (kPackedDdmc) {
            // an IMC step of a hybrid deck through one face into a block of the same size, a
            // periodic wrap or a reflection: served from the face table at once, as in the gray
            // IMC kernels (the particle has a real velocity, which SampleDDMCBlockFace skips)
            if (!is_ddmc_step) {
              const bool xo = ip < M.is || ip > M.ie;
              const bool yo = multi_d && (jp < M.js || jp > M.je);
              const bool zo = three_d && (kp < M.ks || kp > M.ke);
              if (xo && !yo && !zo) crossed = cross_face(std::integral_constant<int, 0>{}, ip > M.ie, x, vx, ip, M.is, M.ie);
              else if (yo && !xo && !zo) crossed = cross_face(std::integral_constant<int, 1>{}, jp > M.je, y, vy, jp, M.js, M.je);
              else if (zo && !xo && !yo) crossed = cross_face(std::integral_constant<int, 2>{}, kp > M.ke, z, vz, kp, M.ks, M.ke);
            }
          }
          if (!crossed) {
            if constexpr (DDMC) {  // transport_ddmc.cpp:203-211: zero velocity flags a DDMC leak
              const bool flag = is_ddmc_step && multi_d && !s.is_rejected;
              if constexpr (kPackedDdmc) {
                if (pend >= 0 && !flag) {  // (1-D: the direction travels with the particle)
                  materialise_dir(s);
                  vx = s.vx; vy = s.vy; vz = s.vz;
                }
                pend = -1;
              }
              const double vmask = flag ? 0.0 : 1.0;
              vx *= vmask; vy *= vmask; vz *= vmask;
            }
            ls = LS_RELOC;  // comm phase: in the service phase
          }
        } else if (s.is_absorbed) {  // transport.cpp:157-163
          if (M.owned[b]) {
            atomicAdd(&M.edelta[b][cidx(M, kp, jp, ip)], g1(S.w)[n]);
            status = ST_ABSORBED;
          } else {
            status = ST_OUTGOING_ABSORBED;  // deposited by the block's owner
          }
          ls = LS_DONE;
        } else {
          if constexpr (kPackedDdmc) lam_cur = f2[cidx(M, kp, jp, ip)];  // (for the next pass)
          if (s.is_scattered) scatter(rng, vv, vx, vy, vz);  // transport.cpp:165-170
          if (!(t < t_end)) ls = LS_DONE;                    // census
        }
      }
      nrun = __popcll(__ballot(ls == LS_RUN));
      waste += running - nrun;
      if (waste >= kServiceBudget) thresh = 65;  // wave-uniform: ends the loop
    }
  }

  if constexpr (TALLY) {
    if (tally_in_lds) {
      __syncthreads();  // every wave of the workgroup leaves the loop above exactly once
      for (int q = threadIdx.x; q < M.nblocks * (int)M.ntot; q += blockDim.x) {
        const double v = lds_tally[q];
        if (v != 0.0) atomicAdd(&M.tally[q / (int)M.ntot][q % (int)M.ntot], v);
      }
    }
  }

  unsigned long long r_census = wave_sum(c_census), r_abs = wave_sum(c_abs), r_esc = wave_sum(c_esc),
                     r_out = wave_sum(c_out), r_ev = c_ev;
  if (lane == 0) {
    if (r_census) atomicAdd(&counters[CNT_CENSUS], r_census);
    if (r_abs) atomicAdd(&counters[CNT_ABSORBED], r_abs);
    if (r_esc) atomicAdd(&counters[CNT_ESCAPED], r_esc);
    if (r_out) atomicAdd(&counters[CNT_OUTGOING], r_out);
    if (r_ev) atomicAdd(&counters[CNT_EVENTS], r_ev);
#ifdef JB_TIMING
    atomicAdd(&counters[CNT_PASSES], cyc_ev >> 10);
    atomicAdd(&counters[CNT_SERVICE], cyc_sv >> 10);
#else
    atomicAdd(&counters[CNT_PASSES], (unsigned long long)c_pass);
    atomicAdd(&counters[CNT_SERVICE], (unsigned long long)c_service);
#endif
  }
}

// -------------------------------------------------------------------------------------------
template <int NDIM>
__global__ void __launch_bounds__(kBlock)
    k_block_face(DevMesh M, DevParams P, DevSwarm S, long long first, long long last) {
  load_math_tables();
  for (long long n = first + (long long)blockIdx.x * blockDim.x + threadIdx.x; n < last;
       n += (long long)gridDim.x * blockDim.x) {
    if (S.status[n] != ST_ACTIVE) continue;
    const int b = S.blk[n];
    Blk B;
    load_block(M, b, B);
    LcgRng rng(S.rng[n]);
    double x = S.x[n], y = S.y[n], z = S.z[n];
    double vx = S.vx[n], vy = S.vy[n], vz = S.vz[n];
    int ip = S.ip[n], jp = S.jp[n], kp = S.kp[n];
    sample_block_face<NDIM>(M, P, B, b, rng, x, y, z, vx, vy, vz, ip, jp, kp);
    S.x[n] = x; S.y[n] = y; S.z[n] = z;
    S.vx[n] = vx; S.vy[n] = vy; S.vz[n] = vz;
    S.ip[n] = ip; S.jp[n] = jp; S.kp[n] = kp;
    S.rng[n] = rng.s;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_check_completion(DevSwarm S, long long n_total, double t_end, unsigned long long *counters) {
  unsigned long long c = 0;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < n_total;
       n += (long long)gridDim.x * blockDim.x)
    if (S.status[n] == ST_ACTIVE && S.t[n] < t_end) ++c;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counters[CNT_UNFINISHED], c);
}

__global__ void __launch_bounds__(kBlock) k_zero_tally(DevMesh M) {
  const long long total = (long long)M.nblocks * M.ncell;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    int b, k, j, i, cell;
    decode_cell(M, c, b, k, j, i, cell);
    M.tally[b][cidx(M, k, j, i)] = 0.0;
  }
}

__global__ void __launch_bounds__(kBlock) k_tally(DevMesh M, DevSwarm S, long long n_total) {
  __shared__ double lds_tally[kLdsTally];  // small meshes: see k_transport
  const int ncells = M.nblocks * (int)M.ntot;
  const bool in_lds = (long long)M.nblocks * M.ntot <= (long long)kLdsTally;
  if (in_lds) {
    for (int q = threadIdx.x; q < ncells; q += blockDim.x) lds_tally[q] = 0.0;
    __syncthreads();
  }
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < n_total;
       n += (long long)gridDim.x * blockDim.x) {
    if (S.status[n] != ST_ACTIVE) continue;
    const int b = S.blk[n];
    const double dv = M.blk_dx[3 * b] * M.blk_dx[3 * b + 1] * M.blk_dx[3 * b + 2];
    const int q = cidx(M, S.kp[n], S.jp[n], S.ip[n]);
    if (in_lds) atomicAdd(&lds_tally[b * (int)M.ntot + q], S.w[n] / dv);
    else atomicAdd(&M.tally[b][q], S.w[n] / dv);
  }
  if (in_lds) {
    __syncthreads();
    for (int q = threadIdx.x; q < ncells; q += blockDim.x) {
      const double v = lds_tally[q];
      if (v != 0.0) atomicAdd(&M.tally[q / (int)M.ntot][q % (int)M.ntot], v);
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_update_fluid(DevMesh M) {
  const long long total = (long long)M.nblocks * M.ncell;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += (long long)gridDim.x * blockDim.x) {
    int b, k, j, i, cell;
    decode_cell(M, c, b, k, j, i, cell);
    const long long q = cidx(M, k, j, i);
    const double dv = M.blk_dx[3 * b] * M.blk_dx[3 * b + 1] * M.blk_dx[3 * b + 2];
    const double delta = M.edelta[b][q] / dv;
    M.u[b][q] += delta;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_reflect_bc(DevMesh M, DevSwarm S, long long n_total, int face) {
  const int d = face >> 1, outer = face & 1;
  double *pos = (d == 0) ? S.x : (d == 1 ? S.y : S.z);
  double *vel = (d == 0) ? S.vx : (d == 1 ? S.vy : S.vz);
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < n_total;
       n += (long long)gridDim.x * blockDim.x) {
    if (S.status[n] != ST_ACTIVE) continue;
    double q = pos[n];
    bool hit = false;
    if (!outer && q < M.gmin[d]) {
      q = M.gmin[d] + (M.gmin[d] - q);
      hit = true;
    } else if (outer && q > M.gmax[d]) {
      q = M.gmax[d] - (q - M.gmax[d]);
      hit = true;
    }
    if (hit) {
      pos[n] = q;
      vel[n] = -vel[n];
      Blk B;
      load_block(M, S.blk[n], B);
      int i, j, k;
      if (M.ndim == 1) xtoijk<1>(M, B, S.x[n], S.y[n], S.z[n], i, j, k);
      else if (M.ndim == 2) xtoijk<2>(M, B, S.x[n], S.y[n], S.z[n], i, j, k);
      else xtoijk<3>(M, B, S.x[n], S.y[n], S.z[n], i, j, k);
      S.ip[n] = i; S.jp[n] = j; S.kp[n] = k;
    }
  }
}

// -------------------------------------------------------------------------------------------
// RemoveMarkedParticles: survivors S = #ACTIVE in [0,n).  Holes (non-ACTIVE slots below S) are
// filled by movers (ACTIVE slots at or above S); both lists are built with wave-aggregated
// atomics (ballot + popcount prefix), so which mover fills which hole is unordered -- harmless,
// every particle carries its own random stream.
__device__ __forceinline__ long long wave_slot(bool pred, unsigned long long *cursor) {
  const unsigned long long mask = __ballot(pred);
  const int lane = threadIdx.x & 63;
  long long base = 0;
  if (mask) {
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = (long long)atomicAdd(cursor, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
  }
  return base + __popcll(mask & ((1ull << lane) - 1ull));
}

// What RemoveMarkedParticles keeps: the active photons AND the ones waiting for their hand-off (not packed
// yet -- k_pack_outgoing turns their slots into holes): a compaction between a transport pass and the
// exchange (jb_exchange's answer to JB_ERR_CAPACITY) must not drop them.
__device__ __forceinline__ bool swarm_slot_live(int status) {
  return status == ST_ACTIVE || status == ST_OUTGOING || status == ST_OUTGOING_ABSORBED;
}

__global__ void __launch_bounds__(kBlock)
    k_count_active(DevSwarm S, long long n_total, unsigned long long *cursor) {
  unsigned long long c = 0;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < n_total;
       n += (long long)gridDim.x * blockDim.x)
    if (swarm_slot_live(S.status[n])) ++c;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(cursor, c);
}

__global__ void __launch_bounds__(kBlock)
    k_list_holes_movers(DevSwarm S, long long n_total, long long survivors, long long *holes,
                        long long *movers, unsigned long long *cursors) {
  const long long span = ((n_total + kBlock - 1) / kBlock) * kBlock;  // keep waves converged
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < span;
       n += (long long)gridDim.x * blockDim.x) {
    const bool valid = n < n_total;
    const bool active = valid && swarm_slot_live(S.status[n]);
    const bool hole = valid && !active && n < survivors;
    const bool mover = active && n >= survivors;
    const long long hs = wave_slot(hole, &cursors[0]);
    if (hole) holes[hs] = n;
    const long long ms = wave_slot(mover, &cursors[1]);
    if (mover) movers[ms] = n;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_fill_holes(DevSwarm S, const long long *holes, const long long *movers, long long count) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < count;
       q += (long long)gridDim.x * blockDim.x) {
    const long long d = holes[q], s = movers[q];
    S.x[d] = S.x[s]; S.y[d] = S.y[s]; S.z[d] = S.z[s];
    S.vx[d] = S.vx[s]; S.vy[d] = S.vy[s]; S.vz[d] = S.vz[s];
    S.t[d] = S.t[s]; S.w[d] = S.w[s]; S.e[d] = S.e[s];
    S.ip[d] = S.ip[s]; S.jp[d] = S.jp[s]; S.kp[d] = S.kp[s];
    S.blk[d] = S.blk[s]; S.status[d] = S.status[s];
    S.id[d] = S.id[s]; S.rng[d] = S.rng[s];
  }
}

// -------------------------------------------------------------------------------------------
// DefragParticles (reference jaybenne.cpp:499-509: Swarm::Defrag, never scheduled there).  Here the
// swarm is compact after RemoveMarkedParticles; what degrades over the cycles is its ORDER: photons
// are sourced in (block, cell) order, so the photons a wave holds -- and the ones in flight on an
// XCD -- gather their cell data from a few neighbouring rows that stay in that XCD's 4 MiB L2, and
// every cycle of diffusion loosens that (measured on BASELINE configs[2] in 3-D: 26.5 ms per cycle
// at cycle 2, 39.9 at cycle 16).  The defragmentation is therefore a counting sort of the photons
// by (resident block, cell of their position): histogram, exclusive scan, move (through records in
// scratch memory, back into the same arrays).  Photons of one cell come out in arbitrary order
// (the move claims slots with atomics); nothing a photon carries changes.
// (Photons of one cell sit next to each other in a swarm that is still nearly in order -- ~50 per
// cell on the stepdiff decks --, so a wave's 64 keys are a few RUNS of equal keys: one atomic per run,
// by its first lane, instead of one per photon on the same address.)
// run_of: for the calling lane, the first lane of its run of equal keys among consecutive active
// lanes and the length of that run
__device__ __forceinline__ void run_of(unsigned key, bool active, int lane, int &head, int &len) {
  const unsigned prev = __shfl_up(key, 1, 64);
  const unsigned long long act = __ballot(active);
  const unsigned long long heads = __ballot(active && (lane == 0 || key != prev));
  const unsigned long long upto = heads & ((2ull << lane) - 1ull);          // heads at or below me
  head = 63 - __clzll((long long)upto);
  const unsigned long long above = heads & ~((2ull << head) - 1ull);        // heads above my run's
  const int nact = __popcll(act);                                            // (active lanes: 0 .. nact-1)
  len = (above != 0ull ? __ffsll((long long)above) - 1 : nact) - head;
}
__global__ void __launch_bounds__(kBlock)
    k_sort_count(DevMesh M, DevSwarm S, long long n, unsigned nkeys, unsigned *key, unsigned *hist) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) - lane;
  for (long long base = wave0; base < n; base += (long long)gridDim.x * blockDim.x) {
    const long long p = base + lane;
    const bool active = p < n;
    unsigned k = nkeys;  // (anything but an active photon in a resident block: behind all cells)
    if (active) {
      const int b = S.blk[p];
      if (S.status[p] == ST_ACTIVE && b >= 0 && b < M.nblocks) {
        Blk B;
        load_block(M, b, B);
        int i, j, kk;
        if (M.ndim == 1) xtoijk<1>(M, B, S.x[p], S.y[p], S.z[p], i, j, kk);
        else if (M.ndim == 2) xtoijk<2>(M, B, S.x[p], S.y[p], S.z[p], i, j, kk);
        else xtoijk<3>(M, B, S.x[p], S.y[p], S.z[p], i, j, kk);
        i = i < 0 ? 0 : (i >= M.ni ? M.ni - 1 : i);
        j = j < 0 ? 0 : (j >= M.nj ? M.nj - 1 : j);
        kk = kk < 0 ? 0 : (kk >= M.nk ? M.nk - 1 : kk);
        k = (unsigned)b * (unsigned)M.ntot + (unsigned)cidx(M, kk, j, i);
      }
      key[p] = k;
    }
    int head, len;
    run_of(k, active, lane, head, len);
    if (active && lane == head) atomicAdd(&hist[k], (unsigned)len);
  }
}

// In-place exclusive scan of `data` in tiles of kScanTile elements (one workgroup each); the tile
// totals go to `sums`, which k_scan_sums scans and k_scan_add adds back.
constexpr int kScanItems = 8, kScanTile = kBlock * kScanItems;
__global__ void __launch_bounds__(kBlock) k_scan_tiles(unsigned *data, long long n, unsigned *sums) {
  __shared__ unsigned wave_tot[kBlock / 64];
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  unsigned v[kScanItems], mine = 0u;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    v[q] = base + q < n ? data[base + q] : 0u;
    mine += v[q];
  }
  // exclusive scan of `mine` over the workgroup: wave shuffles, then the wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned before = 0u, total = 0u;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) before += wave_tot[w];
    total += wave_tot[w];
  }
  unsigned run = before + incl - mine;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    if (base + q < n) data[base + q] = run;
    run += v[q];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) k_scan_sums(unsigned *sums, int count) {
  // one workgroup, any count: a running offset carried from one stretch of 1024 entries to the next
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s0 = 0; s0 < count; s0 += 1024) {
    const int q = s0 + (int)threadIdx.x;
    const unsigned mine = q < count ? sums[q] : 0u;
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned before = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) before += wave_tot[w];
      total += wave_tot[w];
    }
    const unsigned off = carry;
    if (q < count) sums[q] = off + before + incl - mine;
    __syncthreads();
    if (threadIdx.x == 0) carry = off + total;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kBlock) k_scan_add(unsigned *data, long long n, const unsigned *sums) {
  const unsigned add = sums[blockIdx.x];
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q)
    if (base + q < n) data[base + q] += add;
}

// The move goes through an array of 128-byte particle records in device scratch memory: moving a
// photon then costs ONE out-of-order access of a full line instead of sixteen 8-byte ones (after a
// few cycles a photon's new slot is millions of slots from its old one: measured per 1e8 photons,
// sixteen scattered stores 47.6 ms, sixteen scattered loads 34.8 ms, the histogram 3.8 ms).
//   k_sort_pack:   slot s, read in order -> record at the photon's new position (8 x 16 B, one line)
//   k_sort_unpack: records in order -> the swarm arrays, written in order -- the SAME arrays
// record: {x, y} {z, vx} {vy, vz} {t, w} {e, id} {rng, ip | jp << 32} {kp | blk << 32, status} {-, -}
constexpr int kSortRecWords = 16;  // 8-byte words per record
// (each record is written by ONE store instruction, whole: the wave stages its 64 records in LDS and
// eight lanes then store the eight 16-byte pieces of a record together.  With every lane storing
// its own record piece by piece -- eight instructions that each touch 64 different lines -- the
// pieces of a line reached HBM separately: WRITE_SIZE 38.6 GB for 12.8 GB of records, 13.5 ms.)
constexpr int kSortRowWords = 18;  // 16 words of record + 2 of padding per LDS row (bank spread)
__global__ void __launch_bounds__(kBlock)
    k_sort_pack(DevSwarm S, long long n, const unsigned *key, unsigned *offs, unsigned long long *rec) {
  typedef unsigned long long u64;
  typedef u64 v2u __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) u64 stage[kBlock / 64][64][kSortRowWords];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) - lane;
  for (long long base = wave0; base < n; base += (long long)gridDim.x * blockDim.x) {
    const long long s = base + lane;
    const bool active = s < n;
    const unsigned k = active ? key[s] : 0u;
    int head, len;
    run_of(k, active, lane, head, len);
    unsigned first = 0u;
    if (active && lane == head) first = atomicAdd(&offs[k], (unsigned)len);   // (one claim per run)
    first = __shfl(first, head, 64);
    const unsigned dest = first + (unsigned)(lane - head);
    if (active) {
      v2u *o = (v2u *)&stage[wave][lane][0];
      auto bits = [](double v) { return (u64)__double_as_longlong(v); };
      o[0] = v2u{bits(S.x[s]), bits(S.y[s])};
      o[1] = v2u{bits(S.z[s]), bits(S.vx[s])};
      o[2] = v2u{bits(S.vy[s]), bits(S.vz[s])};
      o[3] = v2u{bits(S.t[s]), bits(S.w[s])};
      o[4] = v2u{bits(S.e[s]), (u64)S.id[s]};
      o[5] = v2u{(u64)S.rng[s], (u64)(unsigned)S.ip[s] | ((u64)(unsigned)S.jp[s] << 32)};
      o[6] = v2u{(u64)(unsigned)S.kp[s] | ((u64)(unsigned)S.blk[s] << 32), (u64)(unsigned)S.status[s]};
      o[7] = v2u{0ull, 0ull};
    }
    // (LDS operations of one wave complete in order; the fence keeps the compiler from moving the
    // reads of other lanes' rows above these writes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int piece = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 8 * j + (lane >> 3);                     // the record these eight lanes store
      const unsigned d = __shfl(dest, row, 64);
      if (base + row < n) {
        const v2u v = *(const v2u *)&stage[wave][row][2 * piece];
        *(v2u *)(rec + (size_t)kSortRecWords * (size_t)d + 2 * piece) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
__global__ void __launch_bounds__(kBlock)
    k_sort_unpack(DevSwarm D, long long n, const unsigned long long *rec) {
  typedef unsigned long long u64;
  typedef u64 v2u __attribute__((ext_vector_type(2)));
  for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < n;
       d += (long long)gridDim.x * blockDim.x) {
    const v2u *r = (const v2u *)(rec + (size_t)kSortRecWords * (size_t)d);
    const v2u a = r[0], b = r[1], c = r[2], e = r[3], f = r[4], g = r[5], h = r[6];
    auto dbl = [](u64 v) { return __longlong_as_double((long long)v); };
    D.x[d] = dbl(a.x); D.y[d] = dbl(a.y); D.z[d] = dbl(b.x);
    D.vx[d] = dbl(b.y); D.vy[d] = dbl(c.x); D.vz[d] = dbl(c.y);
    D.t[d] = dbl(e.x); D.w[d] = dbl(e.y); D.e[d] = dbl(f.x);
    D.id[d] = f.y; D.rng[d] = g.x;
    D.ip[d] = (int)(unsigned)g.y; D.jp[d] = (int)(unsigned)(g.y >> 32);
    D.kp[d] = (int)(unsigned)h.x; D.blk[d] = (int)(unsigned)(h.x >> 32);
    D.status[d] = (int)(unsigned)h.y;
  }
}

// -------------------------------------------------------------------------------------------
// Inter-rank hand-off records: 13 x 8 bytes
//   0..8  x y z vx vy vz t w e   9 id   10 (ip | jp << 32)   11 (kp | gblock << 32)   12 rng state
constexpr int kRecWords = 13;

// Both kernels address the per-rank counters once per wave and destination rank, not once per photon: atomics on
// one 128-byte line of device memory are served one after the other (~13 ns each with every XCD sending them:
// DESIGN.md 4.2), and a rank of a spatially decomposed BASELINE configs[4] run hands over ~1.3e6 photons per
// cycle.  for_each_destination: the lanes of the wave that hold an outgoing photon, grouped by the rank that owns
// its block; f(rank, lanes of the group as a mask, this lane's place in the group) runs with the group's lanes.
template <class F>
__device__ __forceinline__ void for_each_destination(bool out, int r, F f) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(out);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int rl = __builtin_amdgcn_readlane(r, leader);
    const unsigned long long grp = __ballot(out && r == rl);
    if (out && r == rl) f(rl, grp, __popcll(grp & ((1ull << lane) - 1ull)), lane == leader);
    todo &= ~grp;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_count_outgoing(DevMesh M, DevSwarm S, long long n_first, long long n_total,
                     unsigned long long *per_rank) {
  const int lane = threadIdx.x & 63;
  // (wave-uniform trip count: every lane of the wave takes part in the grouping)
  for (long long n0 = n_first + (long long)blockIdx.x * blockDim.x + (threadIdx.x - lane); n0 < n_total;
       n0 += (long long)gridDim.x * blockDim.x) {
    const long long n = n0 + lane;
    const int st = n < n_total ? S.status[n] : ST_ABSORBED;
    const bool out = st == ST_OUTGOING || st == ST_OUTGOING_ABSORBED;
    const int r = out ? M.owner[S.blk[n]] : 0;
    for_each_destination(out, r, [&](int rl, unsigned long long grp, int, bool first) {
      if (first) atomicAdd(&per_rank[rl], (unsigned long long)__popcll(grp));
    });
  }
}

__global__ void __launch_bounds__(kBlock)
    k_pack_outgoing(DevMesh M, DevSwarm S, long long n_first, long long n_total,
                    const long long *rank_first, unsigned long long *rank_cursor, long long *rec) {
  const int lane = threadIdx.x & 63;
  for (long long n0 = n_first + (long long)blockIdx.x * blockDim.x + (threadIdx.x - lane); n0 < n_total;
       n0 += (long long)gridDim.x * blockDim.x) {
    const long long n = n0 + lane;
    const int st = n < n_total ? S.status[n] : ST_ABSORBED;
    const bool out = st == ST_OUTGOING || st == ST_OUTGOING_ABSORBED;
    const int g = out ? S.blk[n] : 0;
    const int r = out ? M.owner[g] : 0;
    long long slot = -1;
    for_each_destination(out, r, [&](int rl, unsigned long long grp, int place, bool first) {
      unsigned long long base = 0ull;
      if (first) base = atomicAdd(&rank_cursor[rl], (unsigned long long)__popcll(grp));
      base = __shfl(base, __ffsll((long long)grp) - 1, 64);
      slot = rank_first[rl] + (long long)base + place;
    });
    if (!out) continue;
    S.status[n] = ST_ABSORBED;  // the slot is a hole from now on (removed by the next compaction)
    long long *o = rec + slot * kRecWords;
    o[0] = __double_as_longlong(S.x[n]); o[1] = __double_as_longlong(S.y[n]);
    o[2] = __double_as_longlong(S.z[n]); o[3] = __double_as_longlong(S.vx[n]);
    o[4] = __double_as_longlong(S.vy[n]); o[5] = __double_as_longlong(S.vz[n]);
    o[6] = __double_as_longlong(S.t[n]); o[7] = __double_as_longlong(S.w[n]);
    o[8] = __double_as_longlong(S.e[n]);
    // bit 63 of the id word: absorbed in a halo copy, the receiver only deposits the weight
    o[9] = (long long)(S.id[n] | (st == ST_OUTGOING_ABSORBED ? (1ull << 63) : 0ull));
    o[10] = (long long)(((unsigned long long)(unsigned)S.jp[n] << 32) | (unsigned)S.ip[n]);
    o[11] = (long long)(((unsigned long long)(unsigned)g << 32) | (unsigned)S.kp[n]);
    o[12] = (long long)S.rng[n];
  }
}

__global__ void __launch_bounds__(kBlock)
    k_unpack_incoming(DevMesh M, DevSwarm S, long long first_slot, const long long *rec,
                      long long nrec) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nrec;
       q += (long long)gridDim.x * blockDim.x) {
    const long long *o = rec + q * kRecWords;
    const long long n = first_slot + q;
    S.x[n] = __longlong_as_double(o[0]); S.y[n] = __longlong_as_double(o[1]);
    S.z[n] = __longlong_as_double(o[2]); S.vx[n] = __longlong_as_double(o[3]);
    S.vy[n] = __longlong_as_double(o[4]); S.vz[n] = __longlong_as_double(o[5]);
    S.t[n] = __longlong_as_double(o[6]); S.w[n] = __longlong_as_double(o[7]);
    S.e[n] = __longlong_as_double(o[8]);
    const bool absorbed = ((unsigned long long)o[9] >> 63) != 0ull;
    S.id[n] = (uint64_t)o[9] & ~(1ull << 63);
    S.ip[n] = (int)(unsigned)(o[10] & 0xffffffffll);
    S.jp[n] = (int)(unsigned)((unsigned long long)o[10] >> 32);
    S.kp[n] = (int)(unsigned)(o[11] & 0xffffffffll);
    const int g = (int)(unsigned)((unsigned long long)o[11] >> 32);
    const int li = M.local_index[g];
    S.blk[n] = li;
    S.rng[n] = (uint64_t)o[12];
    if (absorbed) {  // transport.cpp:159-161 on behalf of the rank that tracked the particle
      atomicAdd(&M.edelta[li][cidx(M, S.kp[n], S.jp[n], S.ip[n])], S.w[n]);
      S.status[n] = ST_ABSORBED;
    } else {
      S.status[n] = ST_ACTIVE;
    }
  }
}

// -------------------------------------------------------------------------------------------
// Ghost-zone / halo refresh of a host field (see jb_gather_cells / jb_fill_cells in the header).
__global__ void __launch_bounds__(kBlock)
    k_gather_cells(double *const *F, long long n, const int *blk, const int *cell, double *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = F[blk[i]][cell[i]];
}

template <int NS>
__global__ void __launch_bounds__(kBlock)
    k_fill_cells(double *const *F, long long n, const int *dst_blk, const int *dst_cell,
                 const int *src_blk, const int *src_cell, const double *remote) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    double v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int sb = src_blk[i * NS + s];
      const int sc = src_cell[i * NS + s];
      v[s] = sb >= 0 ? F[sb][sc] : remote[sc];
    }
#pragma unroll
    for (int w = NS; w > 1; w >>= 1)
#pragma unroll
      for (int s = 0; s < w / 2; ++s) v[s] = v[2 * s] + v[2 * s + 1];
    F[dst_blk[i]][dst_cell[i]] = v[0] / (double)NS;
  }
}

}  // namespace jb

#include "jb_kernel_ddmc.hpp"
