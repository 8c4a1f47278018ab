// jb_device.hpp -- device-side views of the mesh / swarm and the geometry helpers that stand in
// for Parthenon's SparsePack coordinates and SwarmDeviceContext (absent from the reference tree;
// semantics restated from the call sites).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jb_physics.hpp"

namespace jb {

// ---- by-value kernel arguments (land in SGPRs) -------------------------------------------------
struct DevMesh {
  int ndim, ng, nblocks, nblocks_total, rank;
  int nx[3], nleaf[3], bc[6];
  int is, js, ks, ie, je, ke, ni, nj, nk;
  int ncell;          // interior cells per block
  long long ntot;     // cells per block incl. ghosts
  double gmin[3], gmax[3];
  const int *leaf_map, *owner, *local_index, *gid;
  const int *owned;  // [nblocks] 1 = owned by this rank, 0 = halo copy
  const double *blk_xmin, *blk_xmax, *blk_dx;
  const double *blk_inv_dx;  // [nblocks][3]: 1.0 / dx, computed once on the host
  double inv_leaf_len[3];    // 1.0 / ((gmax - gmin) / nleaf)
  const int *blk_level, *blk_nbr_lev;
  double *const *rho, *const *sie, *const *u, *const *fleck, *const *tally, *const *edelta,
      *const *src_ew, *const *src_num, *const *P1, *const *P2, *const *P3;
  // library-owned per-cell mean free paths 1/(f sigma_a), 1/(sigma_s + (1-f) sigma_a), filled by
  // UpdateDerivedTransportFields for frequency-independent (gray) opacities
  double *const *lam_abs, *const *lam_sc;
  const double *lam_base;  // lam_abs[b] = lam_base + 2 b ntot, lam_sc[b] = lam_base + (2 b + 1) ntot
  // Crossing one face of a resident block b (face f = 2 axis + upper): nbr_ent[6 b + f] is -1 when
  // the general relocation has to run (level change, destination not resident, outflow, inactive
  // axis), else kind << 28 | destination block (local index) with kind 0 = the same-level
  // neighbour, 1 = the same-level block on the other side of a periodic boundary, 2 = reflecting
  // boundary (the block itself); nbr_x0[6 b + f] = coordinate of index 0 of the destination
  // along the face's axis (Blk::x0).  Built by jb_mesh_create.
  const int *nbr_ent;
  const double *nbr_x0;
  // library-owned, gray opacities with DDMC: 8 doubles per cell {f sigma_a, sigma_a + sigma_s,
  // leak opacities P/dx of the faces x-, x+, y-, y+, z-, z+} -- everything a DDMC step gathers,
  // in one 64-byte record
  double *const *ddmc_cell;
  const double *ddmc_base;  // ddmc_cell[b] = ddmc_base + 8 b ntot
  double *ddmc_step;        // step records of k_ddmc_all's event loop (DdmcStepRec), same indexing
  // Cell codes (k_ddmc_all<.., GATHER 4>): one 32-bit word per cell, same indexing.  Interior cells: the
  // CLASS of the cell's step record -- k_ddmc_pack numbers the DISTINCT step records of the resident
  // blocks each cycle (gray decks: one per level x face-neighbour pattern) and keeps them in ddmc_class
  // (8 doubles each), which the tracking kernel copies to LDS; the event loop then gathers 4 bytes per
  // step instead of 64.  Ghost cells (k_lam_ghost_codes, once per mesh): kCodeGhost | flags | the record
  // number of the cell a particle that leaks there really is in (jb_kernel_ddmc.hpp: kStepGhostTable).
  // nullptr: no codes on this mesh (>= 2^29 resident cells).
  unsigned *ddmc_code;
  double *ddmc_class;       // [kMaxClasses][8]
  int *ddmc_class_slot;     // [kClassSlots][2]: hash slots {state 0 empty / 1 being written / 2 valid, class id}
  // ... and one double per cell (block b at lam_hyb + b ntot) for the hybrid kernel: the cell's
  // scattering mean free path lam_sc, with the sign bit set when the cell takes DDMC steps
  // (dx_push (sigma_a + sigma_s) > tau_ddmc) -- a lane in an IMC cell gathers nothing else
  double *lam_hyb;
  // 1 = every resident block has power-of-two cell widths and a lower corner that is a whole
  // number of them (jb_mesh_exact_geometry): the DDMC kernels form cell faces as the EXACT gray
  // IMC kernels do
  int exact;
  // set to 1 by UpdateDerivedTransportFields when some interior cell of a resident block takes
  // IMC steps (dx_push (sigma_a + sigma_s) <= tau_ddmc, transport_ddmc.cpp:135); 0 = every step
  // of every particle is a DDMC step (k_ddmc_all)
  // ([1]: the number of distinct step records k_ddmc_pack has numbered this cycle; > kMaxClasses = the
  // mesh has more than the tracking kernel's LDS table holds and the codes are not valid)
  int *not_all_ddmc;
  // 1.0 / ntot, 1.0 / (ni nj), 1.0 / ni (host): k_ddmc_all turns a record number back into block and
  // cell indices with them (floor(x / d) as a product and one correction)
  double inv_ntot, inv_nij, inv_ni;
};

struct DevParams {
  uint32_t key0;      // Philox key word 0 = deck seed (unadjusted, quirk 1)
  int use_ddmc, do_feedback;
  double tau_ddmc;
  double c, sb;       // speed of light, Stefan-Boltzmann
  double rc;          // m_rcp_refined(c), evaluated on the device at jb_initialize
  double cv;          // IdealGas
  double kappa_a;     // Gray
  double kappa_s, apm;  // GrayS (ThomsonS: kappa_s = sigma_T / length_scale^2)
  // EPBremss (opac_model 1): sigma_a = A rho^2 T^-1/2 (1 - e^(-B nu / T)) nu^-3, j = E rho^2 T^1/2
  int opac_model;
  int lean;           // 1: lean arithmetic in the IMC steps of the hybrid kernel (jb_set_arithmetic)
  int hyb_imc_budget;   // k_hybrid: idle lane-passes that buy a service phase (JB_HYBRID_IMC_BUDGET)
  int hyb_park_budget;  // ... lane-passes parked DDMC lanes wait for their loop (JB_HYBRID_PARK_BUDGET)
  double ep_A, ep_B, ep_E;
};

struct DevSwarm {
  double *x, *y, *z, *vx, *vy, *vz, *t, *w, *e;
  int *ip, *jp, *kp, *blk, *status;
  uint64_t *id;   // creation index (diagnostic key, never read by the tracking kernel)
  uint64_t *rng;  // LCG state of the particle's stream
};

// read-only field data reached through a pointer that was itself loaded from memory: telling
// the compiler it is global memory makes the gathers global_load instead of flat_load
typedef const double __attribute__((address_space(1))) *gcptr;
typedef const int __attribute__((address_space(1))) *gcptr_i;

// (a pointer read from memory instead of passed as a kernel argument: say that it is global memory,
// or every access through it is a flat_ instruction)
template <class T>
__device__ __forceinline__ T *g1(T *p) {
  return (T *)(__attribute__((address_space(1))) T *)p;
}
// The mesh view as the kernels' argument structs name it: a pointer to CONSTANT memory -- written
// by the host before the launch, never by a kernel.  Read through this type (the struct is overlaid
// on the kernel-argument segment) its fields arrive by scalar loads, counted by lgkmcnt; as global
// loads each was a vector-memory instruction with its s_waitcnt vmcnt(0) behind it, i.e. a wait
// for every store and every particle request the wave had in flight -- six to ten times per
// service phase.  (A cast to this address space and back on a generic pointer is folded away.)
typedef const __attribute__((address_space(4))) struct DevMesh *MeshConstPtr;

// Ghost cells of lam_sc hold, instead of a mean free path, what becomes of a photon that steps into
// them -- a NEGATIVE double whose words are
//   high: 0xC330'0000 | flags << 16   (sign set, exponent of 2^52: an ordinary negative number)
//   low : byte offset of the cell the photon is in after the crossing
// flags bit 3 (kGhostTable): the low word is valid -- the first interior cell of the same-level
// resident neighbour behind that face or of the block across a periodic boundary, or, with bit
// 0 / 1 / 2 set, the cell the photon came from at a reflecting wall normal to x / y / z
// (boundaries.hpp:46-82; position and direction mirrored); bit 3 clear: everything else (edges and
// corners, destinations that are not resident, outflow) -- the general relocation.
// A face to a resident block one level COARSER or FINER (bits 8 / 9 of the high word, with bit 3):
//   coarser: the low word is the coarse cell that contains the ghost cell; the cell-local position
//     moves by -+ h_fine per axis (bits 10 / 11 / 12 set: minus), the lane's geometry doubles;
//   finer: the low word is the lowest of the 2 (2-D) or 4 (3-D) fine cells behind the coarse ghost
//     cell, in the layer next to the face; bits 10-11 name the face's axis, bit 12 its side (set: the
//     photon left through the lower face).  At run time every transverse axis adds its stride when
//     the local coordinate is >= 0 (Xtoijk's floor) and the coordinate moves by -+ h_fine; the normal
//     one moves by +- h_fine; the lane's geometry halves.
//   (in cell-local coordinates a level change is these few additions: no boundary conditions, no
//   leaf-map lookup, no Xtoijk -- SampleDDMCBlockFace does not act on a photon with a velocity).
// Built once per mesh (k_lam_ghost_codes) from the face table of jb_mesh_create (nbr_ent, nbr_dq)
// and the block geometry; ghost cells of lam_abs: 1 (never looked at).
// lam_hyb (the hybrid kernel's per-cell datum, 8 ntot bytes per block) carries the same codes with
// byte offsets of its own layout and, in the low 8 bits of the high word, the destination block
// (that kernel keeps the block index per lane, at most kLdsBlocks = 128 resident blocks); a DDMC
// cell's datum there is negative too, with an ordinary exponent.
constexpr int kGhostHi = (int)0xC3300000u, kGhostTable = 1 << 19;
// cell codes of the all-DDMC kernel (DevMesh::ddmc_code)
constexpr int kMaxClasses = 256;            // 16 KB of LDS per workgroup at most
constexpr int kClassSlots = 4 * kMaxClasses;
constexpr unsigned kCodeGhost = 0x80000000u, kCodeMirror = 0x40000000u, kCodeTable = 0x20000000u;
constexpr unsigned kCodeRecMask = 0x1fffffffu;
constexpr int kGhostCoarser = 1 << 8, kGhostFiner = 1 << 9;

enum { ST_ACTIVE = 0, ST_ABSORBED = 1, ST_ESCAPED = 2, ST_OUTGOING = 3, ST_OUTGOING_ABSORBED = 4 };
enum { BC_PERIODIC = 0, BC_REFLECT = 1, BC_OUTFLOW = 2 };

// ---- EOS / opacity evaluated per event, device side (singularity IdealGas / Gray / GrayS;
// reference mcblock.cpp:78-145, call sites transport.cpp:122-127) -----------------------------
__device__ __forceinline__ double eos_temperature(const DevParams &P, double rho, double sie) {
  (void)rho;
  const double t = sie / P.cv;
  return t > 0.0 ? t : 0.0;
}
__device__ __forceinline__ double opac_absorption(const DevParams &P, double rho, double temp,
                                                  double nu) {
  if (P.opac_model == 1) {  // (uniform) EPBremss, include/jaybenne_amd.h
    if (!(temp > 0.0)) return 0.0;
    const double g = m_one_minus_exp_neg((P.ep_B * nu) / temp);
    return ((P.ep_A * (rho * rho)) / sqrt(temp)) * (g / ((nu * nu) * nu));
  }
  return rho * P.kappa_a;
}
__device__ __forceinline__ double opac_emissivity(const DevParams &P, double rho, double temp) {
  if (P.opac_model == 1) return (P.ep_E * (rho * rho)) * sqrt(temp > 0.0 ? temp : 0.0);
  const double t2 = temp * temp;
  return (rho * P.kappa_a) * ((4.0 * P.sb) * (t2 * t2));
}
__device__ __forceinline__ double opac_scattering(const DevParams &P, double rho, double temp,
                                                  double nu) {
  (void)temp; (void)nu;
  return (rho / P.apm) * P.kappa_s;
}

// ---- per-lane copy of the current block's geometry ---------------------------------------------
struct Blk {
  double xmin[3], dx[3], x0[3];  // x0 = coordinate of index 0 (first ghost) per dimension
  double inv_dx[3];
  double dx_push;                // min(dx1, dx2, dx3)  (transport.cpp:75-78)
};

__device__ __forceinline__ void load_block(const DevMesh &M, int b, Blk &B) {
  const int first[3] = {M.is, M.js, M.ks};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    B.xmin[d] = ((gcptr)M.blk_xmin)[3 * b + d];
    B.dx[d] = ((gcptr)M.blk_dx)[3 * b + d];
    B.x0[d] = B.xmin[d] - (double)first[d] * B.dx[d];
    B.inv_dx[d] = ((gcptr)M.blk_inv_dx)[3 * b + d];
  }
  B.dx_push = dmin(B.dx[0], dmin(B.dx[1], B.dx[2]));
}

// The same from a copy of the per-block tables in LDS (kernels whose service phase would wait
// for these small dependent loads behind its own stores: vector-memory operations complete in
// issue order, LDS reads have their own counter).  Up to kLdsBlocks resident blocks.
#ifndef JB_LDS_BLOCKS
#define JB_LDS_BLOCKS 128
#endif
constexpr int kLdsBlocks = JB_LDS_BLOCKS;
// (X0: also the coordinate of cell index 0, xmin - first dx, per block and axis -- 3 KB that
// k_ddmc_all, which needs the room for a fourth workgroup per CU, forms where it reads it)
template <bool X0, int NB = kLdsBlocks>
struct LdsBlockTableT {
  static constexpr bool has_x0 = X0;
  static constexpr int capacity = NB;
  double xmin[NB][3], dx[NB][3], inv_dx[NB][3];
  double x0[X0 ? NB : 1][3];
  double *tally[NB];
  int owned[NB];
  int nbr_ent[NB][6];
};
using LdsBlockTable = LdsBlockTableT<true>;
template <class Tab>
__device__ __forceinline__ void fill_block_table(const DevMesh &M, Tab &T) {
  if (M.nblocks > Tab::capacity) return;
  for (int q = threadIdx.x; q < 3 * M.nblocks; q += blockDim.x) {
    (&T.xmin[0][0])[q] = M.blk_xmin[q];
    (&T.dx[0][0])[q] = M.blk_dx[q];
    (&T.inv_dx[0][0])[q] = M.blk_inv_dx[q];
    if constexpr (Tab::has_x0) {
      const int first = (q % 3 == 0) ? M.is : (q % 3 == 1 ? M.js : M.ks);
      (&T.x0[0][0])[q] = M.blk_xmin[q] - (double)first * M.blk_dx[q];
    }
  }
  for (int q = threadIdx.x; q < M.nblocks; q += blockDim.x) {
    T.tally[q] = M.tally[q];
    T.owned[q] = M.owned[q];
  }
  for (int q = threadIdx.x; q < 6 * M.nblocks; q += blockDim.x) (&T.nbr_ent[0][0])[q] = M.nbr_ent[q];
}
template <class Tab>
__device__ __forceinline__ double lds_x0(const DevMesh &M, const Tab &T, int b, int d) {
  if constexpr (Tab::has_x0) {
    return T.x0[b][d];
  } else {
    const int first = d == 0 ? M.is : (d == 1 ? M.js : M.ks);
    return T.xmin[b][d] - (double)first * T.dx[b][d];
  }
}
template <class Tab>
__device__ __forceinline__ int block_nbr_ent(const DevMesh &M, const Tab &T, int b, int face) {
  return M.nblocks > kLdsBlocks ? ((gcptr_i)M.nbr_ent)[6 * b + face] : T.nbr_ent[b][face];
}
template <class Tab>
__device__ __forceinline__ bool block_owned(const DevMesh &M, const Tab &T, int b) {
  return (M.nblocks > kLdsBlocks ? M.owned[b] : T.owned[b]) != 0;
}
template <class Tab>
__device__ __forceinline__ double *block_tally(const DevMesh &M, const Tab &T, int b) {
  return M.nblocks > kLdsBlocks ? M.tally[b] : T.tally[b];
}
template <class Tab>
__device__ __forceinline__ void load_block(const DevMesh &M, const Tab &T, int b, Blk &B) {
  if (M.nblocks > kLdsBlocks) {  // (uniform)
    load_block(M, b, B);
    return;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    B.xmin[d] = T.xmin[b][d];
    B.dx[d] = T.dx[b][d];
    B.x0[d] = lds_x0(M, T, b, d);
    B.inv_dx[d] = T.inv_dx[b][d];
  }
  B.dx_push = dmin(B.dx[0], dmin(B.dx[1], B.dx[2]));
}

// ... and without the fallback to the global tables (k_ddmc_all: the host only launches it when
// the resident blocks fit the LDS table)
template <class Tab>
__device__ __forceinline__ void load_block_lds(const DevMesh &M, const Tab &T, int b, Blk &B) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    B.xmin[d] = T.xmin[b][d];
    B.dx[d] = T.dx[b][d];
    B.x0[d] = lds_x0(M, T, b, d);
    B.inv_dx[d] = T.inv_dx[b][d];
  }
  B.dx_push = dmin(B.dx[0], dmin(B.dx[1], B.dx[2]));
}

// A uniform value in scalar registers of its own (see k_ddmc_all).
__device__ __forceinline__ unsigned sgpr_copy(unsigned v) {
  // (readfirstlane: a value loaded from memory may arrive in a vector register even though every
  // lane holds the same; on a value that is already scalar it folds away)
  unsigned r;
  asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"((unsigned)__builtin_amdgcn_readfirstlane((int)v)));
  return r;
}
// a wave-uniform double in a scalar register pair of its own
__device__ __forceinline__ double uniform_f64(double v) {
  // (two 32-bit copies: the 64-bit "s" operand of a single s_mov_b64 is sometimes handed over in a
  // vector register pair, which the assembler rejects)
  const unsigned long long q = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = sgpr_copy((unsigned)q);
  const unsigned hi = sgpr_copy((unsigned)(q >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ const double *sgpr_copy_ptr(const double *p) {
  const unsigned long long q = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32));
  unsigned long long r;
  asm volatile("s_mov_b64 %0, %1" : "=s"(r) : "s"(((unsigned long long)hi << 32) | lo));
  return (const double *)r;
}

// Parthenon UniformCartesian: Xc(idx) = x0 + (idx + 0.5) dx
__device__ __forceinline__ double xc(const Blk &B, int d, int idx) {
  return B.x0[d] + ((double)idx + 0.5) * B.dx[d];
}

// SwarmDeviceContext::Xtoijk (transport.cpp:96,146): (x - x_min) * (1 / dx) with the reciprocal a
// per-block constant.  Parthenon's own arithmetic is un-vendored; the step functions keep
// particles >= 2e-9 dx away from cell faces, so the index cannot depend on the rounding of the
// quotient.  Saves three FP64 divisions per event.
template <int NDIM>
__device__ __forceinline__ void xtoijk(const DevMesh &M, const Blk &B, double x, double y, double z,
                                       int &i, int &j, int &k) {
  i = (int)floor((x - B.xmin[0]) * B.inv_dx[0]) + M.is;
  j = (NDIM >= 2) ? (int)floor((y - B.xmin[1]) * B.inv_dx[1]) + M.js : M.js;
  k = (NDIM >= 3) ? (int)floor((z - B.xmin[2]) * B.inv_dx[2]) + M.ks : M.ks;
}

__device__ __forceinline__ bool on_block(const DevMesh &M, int i, int j, int k) {
  return i >= M.is && i <= M.ie && j >= M.js && j <= M.je && k >= M.ks && k <= M.ke;
}

// cell index inside one block's [nk][nj][ni] array (mad24: jb_math.hpp)
__device__ __forceinline__ int cidx(const DevMesh &M, int k, int j, int i) {
  return mad24(mad24(k, M.nj, j), M.ni, i);
}

// ---- comm phase applied to one particle in flight ----------------------------------------------
// PhotonReflectBC (boundaries.hpp:46-82) + Parthenon's periodic / outflow swarm boundaries.
// Returns false when the particle leaves through an outflow face.
template <int NDIM>
__device__ __forceinline__ bool apply_swarm_bcs(const DevMesh &M, double &x, double &y, double &z,
                                                double &vx, double &vy, double &vz) {
  double *p[3] = {&x, &y, &z};
  double *v[3] = {&vx, &vy, &vz};
#pragma unroll
  for (int d = 0; d < NDIM; ++d) {
    if (*p[d] < M.gmin[d]) {
      const int bc = M.bc[2 * d];
      if (bc == BC_REFLECT) {
        *p[d] = M.gmin[d] + (M.gmin[d] - *p[d]);
        *v[d] = -*v[d];
      } else if (bc == BC_PERIODIC) {
        *p[d] = M.gmax[d] - (M.gmin[d] - *p[d]);
      } else {
        return false;
      }
    }
    if (*p[d] > M.gmax[d]) {
      const int bc = M.bc[2 * d + 1];
      if (bc == BC_REFLECT) {
        *p[d] = M.gmax[d] - (*p[d] - M.gmax[d]);
        *v[d] = -*v[d];
      } else if (bc == BC_PERIODIC) {
        *p[d] = M.gmin[d] + (*p[d] - M.gmax[d]);
      } else {
        return false;
      }
    }
  }
  return true;
}

// destination block = leaf that contains the point (GetNeighborBlockIndex + Swarm::Send)
template <int NDIM>
__device__ __forceinline__ int find_block(const DevMesh &M, double x, double y, double z) {
  const double p[3] = {x, y, z};
  int l[3] = {0, 0, 0};
#pragma unroll
  for (int d = 0; d < NDIM; ++d) {
    int q = (int)floor((p[d] - M.gmin[d]) * M.inv_leaf_len[d]);
    q = q < 0 ? 0 : q;
    q = q > M.nleaf[d] - 1 ? M.nleaf[d] - 1 : q;
    l[d] = q;
  }
  return M.leaf_map[((long long)l[2] * M.nleaf[1] + l[1]) * M.nleaf[0] + l[0]];
}

template <int D>
__device__ __forceinline__ double &pick(double &a, double &b, double &c) {
  if constexpr (D == 0) return a;
  else if constexpr (D == 1) return b;
  else return c;
}
template <int D>
__device__ __forceinline__ int &picki(int &a, int &b, int &c) {
  if constexpr (D == 0) return a;
  else if constexpr (D == 1) return b;
  else return c;
}

// One face-normal axis A of SampleDDMCBlockFace (sample_ddmc_bface.cpp:167-222 in 2-D,
// :294-410 in 3-D).  Returns true if the particle sat at a block face normal to A.
template <int NDIM, int A, class Rng>
__device__ __forceinline__ bool bface_axis(const DevMesh &M, const Blk &B, int b,
                                           const double *F, Rng &rng, double vv, double &x,
                                           double &y, double &z, double &vx, double &vy, double &vz,
                                           int &ip, int &jp, int &kp) {
  constexpr int A1 = (A + 1) % 3, A2 = (A + 2) % 3;
  const double eps = kEps;
  const double da = B.dx[A];
  const double pa = pick<A>(x, y, z);
  const bool at_min = fuzzy_equal(pa, M.blk_xmin[3 * b + A] + 2.0 * kEpsDdmc * da, da, eps);
  const bool at_max = fuzzy_equal(pa, M.blk_xmax[3 * b + A] - 2.0 * kEpsDdmc * da, da, eps);
  if (!(at_min || at_max)) return false;

  const double dir_sgn = at_min ? 1.0 : -1.0;
  sample_face_iso_dir(dir_sgn * vv, rng, pick<A>(vx, vy, vz), pick<A1>(vx, vy, vz),
                      pick<A2>(vx, vy, vz));
  int f[3] = {ip, jp, kp};
  f[A] = at_min ? f[A] : f[A] + 1;

  if constexpr (NDIM == 2) {
    constexpr int T = (A == 0) ? 1 : 0;
    const double dt_ = B.dx[T];
    int &it = picki<T>(ip, jp, kp);
    double &pt = pick<T>(x, y, z);
    const double lo = xc(B, T, it) - 0.5 * dt_;
    const bool edge_u = fuzzy_equal(pt, lo, dt_, eps);
    const bool edge_l = fuzzy_equal(pt, lo + dt_, dt_, eps);
    if (edge_u || edge_l) {
      const int t_u = edge_u ? it : it + 1;
      const int t_l = edge_u ? it - 1 : it;
      int iu[3] = {f[0], f[1], f[2]}, il[3] = {f[0], f[1], f[2]};
      iu[T] = t_u;
      il[T] = t_l;
      const double P_u = F[cidx(M, iu[2], iu[1], iu[0])];
      const double P_l = F[cidx(M, il[2], il[1], il[0])];
      sample_face_2d(t_l, dt_, P_l, P_u, rng, it, pt);
    }
  } else if constexpr (NDIM == 3) {
    constexpr int T1 = (A == 0) ? 1 : 0;
    constexpr int T2 = (A == 2) ? 1 : 2;
    const double d1 = B.dx[T1], d2 = B.dx[T2];
    int &i1 = picki<T1>(ip, jp, kp);
    int &i2 = picki<T2>(ip, jp, kp);
    double &p1 = pick<T1>(x, y, z);
    double &p2 = pick<T2>(x, y, z);
    const double lo1 = xc(B, T1, i1) - 0.5 * d1;
    const double lo2 = xc(B, T2, i2) - 0.5 * d2;
    const bool e1u = fuzzy_equal(p1, lo1, d1, eps), e1l = fuzzy_equal(p1, lo1 + d1, d1, eps);
    const bool e2u = fuzzy_equal(p2, lo2, d2, eps), e2l = fuzzy_equal(p2, lo2 + d2, d2, eps);
    if ((e1u || e1l) && (e2u || e2l)) {
      const int t1_u = e1u ? i1 : i1 + 1, t1_l = e1u ? i1 - 1 : i1;
      const int t2_u = e2u ? i2 : i2 + 1, t2_l = e2u ? i2 - 1 : i2;
      double Pq[2][2];  // [T2 lower/upper][T1 lower/upper]
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int q1 = 0; q1 < 2; ++q1) {
          int id[3] = {f[0], f[1], f[2]};
          id[T1] = q1 ? t1_u : t1_l;
          id[T2] = q2 ? t2_u : t2_l;
          Pq[q2][q1] = F[cidx(M, id[2], id[1], id[0])];
        }
      sample_face_3d(t1_l, t2_l, d1, d2, Pq[0][0], Pq[0][1], Pq[1][0], Pq[1][1], rng, i1, i2, p1,
                     p2);
    }
  }
  return true;
}

// SampleDDMCBlockFace for one particle (sample_ddmc_bface.cpp:119-424).
template <int NDIM, class Rng>
__device__ __forceinline__ void sample_block_face(const DevMesh &M, const DevParams &P,
                                                  const Blk &B, int b, Rng &rng, double &x,
                                                  double &y, double &z, double &vx, double &vy,
                                                  double &vz, int &ip, int &jp, int &kp) {
  if constexpr (NDIM >= 2) {
    const double vv = P.c;
    if (!(vx * vx + vy * vy + vz * vz < kEps * vv * vv)) return;
    // Work on copies: the compiler merges the per-axis code paths below into stores through a
    // selected ADDRESS, which would pin the caller's own x, y, z, ip, jp, kp (live across its
    // whole event loop) in scratch memory instead of registers.
    double lx = x, ly = y, lz = z, lvx = vx, lvy = vy, lvz = vz;
    int li, lj, lk;
    xtoijk<NDIM>(M, B, lx, ly, lz, li, lj, lk);
    bool done = bface_axis<NDIM, 0>(M, B, b, M.P1[b], rng, vv, lx, ly, lz, lvx, lvy, lvz, li, lj, lk);
    if (!done)
      done = bface_axis<NDIM, 1>(M, B, b, M.P2[b], rng, vv, lx, ly, lz, lvx, lvy, lvz, li, lj, lk);
    if constexpr (NDIM == 3) {
      if (!done) bface_axis<NDIM, 2>(M, B, b, M.P3[b], rng, vv, lx, ly, lz, lvx, lvy, lvz, li, lj, lk);
    }
    x = lx; y = ly; z = lz; vx = lvx; vy = lvy; vz = lvz;
    ip = li; jp = lj; kp = lk;
  }
}

}  // namespace jb
