// jaybenne_amd.hpp -- C++ host-side mirror of the reference's package / task interface
// (reference src/jaybenne/jaybenne.hpp:48-78) over the C ABI of jaybenne_amd.h.
//
// The reference is a C++ package driven by a C++ host application (src/mcblock).  This header
// is what such a host includes: the task functions keep the reference's names, argument order
// and TaskStatus results; PARTHENON_REQUIRE / PARTHENON_FAIL conditions surface as
// jaybenne_amd::Error.  It needs nothing but the C ABI (no HIP headers, no Parthenon): the host
// owns every device buffer and hands raw device pointers over, exactly as Parthenon's packs do
// for the reference.  examples/mcblock_amd.cpp is a complete host application written on it;
// jaybenne_amd/jaybenne.py is the same mirror in Python (and adds the multi-rank hand-off).
//
// Single rank only: RadiationStep here is the task list of jaybenne.cpp:104-138 for a mesh held
// by one rank, where one transport launch resolves every block crossing in flight.
#ifndef JAYBENNE_AMD_HPP_
#define JAYBENNE_AMD_HPP_

#include <cstdint>
#include <cmath>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "jaybenne_amd.h"

namespace jaybenne_amd {

using Real = double;

// parthenon::TaskStatus as the reference's tasks return it (jaybenne.cpp:34,55-56)
enum class TaskStatus { complete, incomplete, iterate };
enum class SourceType { thermal, emission };     // jaybenne.hpp:56
enum class SourceStrategy { uniform, energy };   // jaybenne.hpp:55

struct Error : std::runtime_error {
  jb_status status;
  Error(jb_status st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline TaskStatus Check(jb_status st) {
  if (st < 0) throw Error(st, jb_last_error());
  return st == JB_ITERATE ? TaskStatus::iterate
                          : (st == JB_INCOMPLETE ? TaskStatus::incomplete : TaskStatus::complete);
}

// The StateDescriptor role (jaybenne.cpp:158-266): owns the package context; parameters by name.
class StateDescriptor {
 public:
  StateDescriptor(const jb_params &p, const jb_opacity &opacity, const jb_scattering &scattering,
                  const jb_eos &eos, int device = 0)
      : params_(p) {
    Check(jb_initialize(&p, &eos, &opacity, &scattering, device, &ctx_));
  }
  ~StateDescriptor() { jb_finalize(ctx_); }
  StateDescriptor(const StateDescriptor &) = delete;
  StateDescriptor &operator=(const StateDescriptor &) = delete;

  jb_context *ctx() const { return ctx_; }
  const jb_params &params() const { return params_; }
  int seed() const { return jb_param_seed(ctx_); }  // Param<int>("seed"), jaybenne.cpp:187-190
  // arithmetic of the gray IMC tracking step: JB_ARITH_LEAN (default) or JB_ARITH_EXACT
  void SetArithmetic(int mode) { Check(jb_set_arithmetic(ctx_, mode)); }
  int Arithmetic() const { return jb_get_arithmetic(ctx_); }

 private:
  jb_context *ctx_ = nullptr;
  jb_params params_;
};

// jaybenne::Initialize(pin, opacity, scattering, eos) -- jaybenne.hpp:50-52
inline std::shared_ptr<StateDescriptor> Initialize(const jb_params &p, const jb_opacity &opacity,
                                                   const jb_scattering &scattering,
                                                   const jb_eos &eos, int device = 0) {
  return std::make_shared<StateDescriptor>(p, opacity, scattering, eos, device);
}

// The MeshData<Real> + swarm role: what one rank's tasks operate on.  The host fills `view`
// (host arrays of per-block DEVICE pointers) and `swarm` (device arrays it allocated);
// `reserve(n)` is the host's pool growth (Swarm::AddEmptyParticles, sourcing.cpp:123-131): it
// must leave swarm.capacity >= n with the first swarm.n particles preserved.
class MeshData {
 public:
  MeshData(std::shared_ptr<StateDescriptor> pkg, const jb_mesh_view &view, int32_t *prefix_dev,
           std::function<void(jb_swarm_view &, int64_t)> reserve)
      : pkg_(std::move(pkg)), nblocks_(view.nblocks), prefix_dev_(prefix_dev),
        reserve_(std::move(reserve)) {
    Check(jb_mesh_create(pkg_->ctx(), &view, &mesh_));
  }
  ~MeshData() { jb_mesh_destroy(mesh_); }
  MeshData(const MeshData &) = delete;
  MeshData &operator=(const MeshData &) = delete;

  StateDescriptor &pkg() const { return *pkg_; }
  jb_context *ctx() const { return pkg_->ctx(); }
  jb_mesh *mesh() const { return mesh_; }
  int nblocks() const { return nblocks_; }
  int32_t *prefix_dev() const { return prefix_dev_; }
  void Reserve(int64_t n) {
    if (n > swarm.capacity) reserve_(swarm, n);
    if (n > swarm.capacity) throw Error(JB_ERR_CAPACITY, "swarm pool could not be grown");
  }

  jb_swarm_view swarm{};   // n, capacity and the device arrays
  uint64_t next_id = 0;    // first unused random-stream id
  uint64_t cycle = 0;      // RadiationStep counter (keys the per-cell rounding streams: SourceEpoch); 0 = initialisation
  int64_t events = 0;      // tracking events so far
  // DefragParticles: -1 (default) when the library's policy asks for it (jb_defrag_policy: a cycle
  // that costs 10 % more per event than the best one since the last sort); k > 0 after every k-th
  // RadiationStep; 0 never, as the reference (slot order then stays what the task list makes it)
  int defrag_interval = -1;
  int defrags = 0;
  int steps_since_defrag = 0;

 private:
  std::shared_ptr<StateDescriptor> pkg_;
  jb_mesh *mesh_ = nullptr;
  int nblocks_;
  int32_t *prefix_dev_;
  std::function<void(jb_swarm_view &, int64_t)> reserve_;
};

// ---- host-side arithmetic of SourcePhotons shared by every host (this header's single-rank tasks,
// examples/handoff_mpi.cpp across MPI ranks, adapters/parthenon/jaybenne_amd_tasks.cpp) ----------
// Random-stream ids are GLOBAL creation indices -- block-major over the global block ids, then the
// order within the block -- so that results do not depend on the block -> rank partition.
//   nper_local[b]   new photons of resident block b of this rank (jb_source_photons_count)
//   gid[b]          its global block id
//   all_counts[g]   new photons of every global block g (the sum of the ranks' contributions;
//                   one rank: its own counts scattered by gid)
// Result: slot_base[b] (first swarm slot of block b's new photons, appended after the n_now live
// ones), id_base[b] (stream id of its first new photon), total_local, and the first unused id
// after this source call (identical on every rank).
struct SourcePlan {
  std::vector<int64_t> slot_base;
  std::vector<uint64_t> id_base;
  int64_t total_local = 0;
  uint64_t next_id = 0;
};
inline SourcePlan PlanSource(const std::vector<int32_t> &nper_local, const std::vector<int32_t> &gid,
                             const std::vector<long long> &all_counts, uint64_t next_id,
                             int64_t n_now) {
  SourcePlan pl;
  const size_t nb = nper_local.size();
  std::vector<uint64_t> excl(all_counts.size());
  uint64_t run = 0;
  for (size_t g = 0; g < all_counts.size(); ++g) { excl[g] = run; run += (uint64_t)all_counts[g]; }
  pl.slot_base.resize(nb);
  pl.id_base.resize(nb);
  for (size_t b = 0; b < nb; ++b) {
    pl.slot_base[b] = n_now + pl.total_local;
    pl.id_base[b] = next_id + excl[(size_t)gid[b]];
    pl.total_local += nper_local[b];
  }
  pl.next_id = next_id + run;
  return pl;
}
// The key of the per-cell rounding streams (`epoch` of jb_source_photons_count) as a function of
// the cycle (0 = initialisation, k = the k-th RadiationStep) and the source type, so that every rank
// of every host -- whatever the number of blocks it calls the source for -- uses the same one: 0 for
// the initial thermal source, k for the emission source of cycle k -- the order in which a run makes
// its source calls (jaybenne.cpp:104-105, 570-578); a thermal source inside a cycle, which no task
// list of the reference makes, gets a key of its own.  (jb_source_photons_count accepts epochs below
// 2^20: half a million cycles.)
inline uint32_t SourceEpoch(uint64_t cycle, SourceType st) {
  if (st == SourceType::emission) return (uint32_t)cycle;
  return cycle == 0 ? 0u : (uint32_t)((1u << 19) | cycle);
}

// ---- block -> rank partition (several ranks; jaybenne_amd/mesh.py Mesh.partition is the same algorithm,
// operation for operation, so that a C++ host and the Python host deal the same blocks to the same rank) --
// Contiguous runs of the Z-ordered block list.  cost[b] = the tracking work of block b for one cycle
// (photons sourced there x events per history: BlockCost below); the runs are the split with the
// SMALLEST LARGEST load, found exactly by dynamic programming over the prefix sums (a cycle lasts as
// long as its slowest rank); then a boundary that separates the children of one parent (group[b] equal
// for consecutive blocks) moves to the nearer end of that family if the largest load stays within
// 1 + sibling_slack of the optimum.  The reference inherits Parthenon's balancer with unit cost per
// block (jaybenne.cpp:92-95): pass cost = all ones for that.
// Returns bounds[0 .. nranks]: rank r owns blocks bounds[r] .. bounds[r + 1] - 1.
inline std::vector<int32_t> PartitionBlocks(const std::vector<double> &cost, int nranks,
                                            const std::vector<int64_t> &group, double sibling_slack = 0.10) {
  const int nb = (int)cost.size();
  if (nranks < 1 || nranks > nb) throw Error(JB_ERR_INVALID, "cannot spread the blocks over that many ranks");
  std::vector<double> S((size_t)nb + 1, 0.0);
  for (int b = 0; b < nb; ++b) {
    if (!(cost[b] > 0.0) || !std::isfinite(cost[b])) throw Error(JB_ERR_INVALID, "block costs must be positive and finite");
    S[(size_t)b + 1] = S[(size_t)b] + cost[b];
  }
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<std::vector<double>> best((size_t)nranks, std::vector<double>((size_t)nb + 1, inf));
  std::vector<std::vector<int32_t>> cut((size_t)nranks, std::vector<int32_t>((size_t)nb + 1, 0));
  for (int i = 1; i <= nb; ++i) best[0][(size_t)i] = S[(size_t)i];
  for (int r = 1; r < nranks; ++r)
    for (int i = r + 1; i <= nb; ++i) {
      double bv = inf;
      int32_t bj = r;
      for (int j = r; j < i; ++j) {
        const double a = best[(size_t)r - 1][(size_t)j], c = S[(size_t)i] - S[(size_t)j];
        const double v = a > c ? a : c;
        if (v < bv) { bv = v; bj = j; }                    // (the first of equal minima)
      }
      best[(size_t)r][(size_t)i] = bv;
      cut[(size_t)r][(size_t)i] = bj;
    }
  std::vector<int32_t> bounds((size_t)nranks + 1, 0);
  bounds[(size_t)nranks] = nb;
  for (int r = nranks - 1; r >= 1; --r) bounds[(size_t)r] = cut[(size_t)r][(size_t)bounds[(size_t)r + 1]];
  const double optimum = best[(size_t)nranks - 1][(size_t)nb];
  auto largest = [&](const std::vector<int32_t> &bd) {
    double m = 0.0;
    for (int r = 0; r < nranks; ++r) {
      const double l = S[(size_t)bd[(size_t)r + 1]] - S[(size_t)bd[(size_t)r]];
      m = l > m ? l : m;
    }
    return m;
  };
  for (int r = 1; r < nranks; ++r) {
    const int q = bounds[(size_t)r];
    if (group[(size_t)q - 1] != group[(size_t)q]) continue;   // not inside a family
    int lo = q, hi = q;
    while (lo > 0 && group[(size_t)lo - 1] == group[(size_t)q]) --lo;
    while (hi < nb && group[(size_t)hi] == group[(size_t)q]) ++hi;
    bool have = false;
    double bl = 0.0;
    int bd_dist = 0, bc = q;
    for (int cand : {lo, hi}) {
      if (!(bounds[(size_t)r - 1] < cand && cand < bounds[(size_t)r + 1])) continue;
      std::vector<int32_t> bd = bounds;
      bd[(size_t)r] = cand;
      const double l = largest(bd);
      const int dist = cand > q ? cand - q : q - cand;
      // (order of the Python host's tuples: load, distance, position)
      if (!have || l < bl || (l == bl && (dist < bd_dist || (dist == bd_dist && cand < bc)))) {
        have = true; bl = l; bd_dist = dist; bc = cand;
      }
    }
    if (have && bl <= (1.0 + sibling_slack) * optimum) bounds[(size_t)r] = bc;
  }
  return bounds;
}

// group[] of PartitionBlocks from the Z-ordered leaf list: level[b] and the logical location lloc[3 b + d]
// at that level (Mesh.sibling_groups of the Python host)
inline std::vector<int64_t> SiblingGroups(int ndim, const std::vector<int32_t> &level, const std::vector<int32_t> &lloc) {
  std::vector<int64_t> group(level.size());
  int64_t g = -1;
  bool have_prev = false;
  int32_t pl = 0, pp[3] = {0, 0, 0};
  for (size_t b = 0; b < level.size(); ++b) {
    bool same = false;
    int32_t par[3] = {0, 0, 0};
    if (level[b] > 0) {
      for (int d = 0; d < 3; ++d) par[d] = d < ndim ? lloc[3 * b + d] >> 1 : 0;
      same = have_prev && pl == level[b] && pp[0] == par[0] && pp[1] == par[1] && pp[2] == par[2];
    }
    if (!same) ++g;
    group[b] = g;
    have_prev = level[b] > 0;
    pl = level[b];
    for (int d = 0; d < 3; ++d) pp[d] = par[d];
  }
  return group;
}

// The tracking work of a block for one cycle, in events per photon it starts with (mcblock.block_costs of
// the Python host, the same expressions): c dt (sigma_s + sigma_a + sum_d 1 / (2 dx_d)) in a block that
// takes IMC steps; c dt (sigma_a + sum_d 2 P / dx_d), P = 1 / (3 (sigma_a + sigma_s) dx_d), in one that takes
// DDMC steps (dx_min (sigma_a + sigma_s) > tau_ddmc, transport_ddmc.cpp:135).  With the `uniform` source
// strategy every block starts with the same number of photons (sourcing.cpp:68-69), so this IS the cost.
inline double BlockCost(int ndim, const double dx[3], double c_dt, double sig_a, double sig_s, bool use_ddmc,
                        double tau_ddmc) {
  double dmin = dx[0];
  for (int d = 1; d < ndim; ++d) dmin = dx[d] < dmin ? dx[d] : dmin;
  double per_history;
  if (use_ddmc && dmin * (sig_a + sig_s) > tau_ddmc) {
    double leak = 0.0;
    for (int d = 0; d < ndim; ++d) leak += 2.0 * (2.0 / (3.0 * 2.0 * (sig_a + sig_s) * dx[d])) / dx[d];
    per_history = c_dt * (sig_a + leak);
  } else {
    double cross = 0.0;
    for (int d = 0; d < ndim; ++d) cross += 0.5 / dx[d];
    per_history = c_dt * (sig_s + sig_a + cross);
  }
  return per_history > 1.0 ? per_history : 1.0;
}

// Replicated mesh, split particles (jb_source_photons_fill_range; jaybenne_amd/jaybenne.py rank_share): of
// the nper[b] new photons of block b, rank `rank` of `nranks` sources count[b] starting at first[b].
inline void RankShare(const std::vector<int32_t> &nper, int rank, int nranks, std::vector<int32_t> *first,
                      std::vector<int32_t> *count) {
  first->resize(nper.size());
  count->resize(nper.size());
  for (size_t b = 0; b < nper.size(); ++b) {
    const int64_t n = nper[b];
    const int64_t f = (n * rank) / nranks;
    (*first)[b] = (int32_t)f;
    (*count)[b] = (int32_t)((n * (rank + 1)) / nranks - f);
  }
}

// ---- halo copies (several ranks; shared by examples/handoff_mpi.cpp, the Parthenon adapter and,
// restated in numpy, the Python host: jaybenne_amd/mesh.py Mesh.neighbours, halo.py) -------------
// A rank keeps read-only copies of the other ranks' blocks that TOUCH one of its own (face, edge or
// corner, through periodic boundaries too).  A photon that wanders across the rank boundary keeps
// being tracked in flight and is handed to the owner of the block it ends in ONCE, when its history
// is over: two transport iterations per cycle instead of one per rank-boundary crossing of the most
// persistent photon (reference jaybenne.cpp:113-131 iterates ~80 times on BASELINE configs[1]).
//   Mesh description (what jb_mesh_view holds for ALL blocks of the mesh, by global id):
//   leaf_map over the finest-level leaf grid nleaf[3], block corners, mesh boundary kinds
//   (periodic[2 d + side]), owner[g].
// Result: resident_gids = this rank's blocks (ascending global id) followed by its halo copies
// (ascending), owned[] for jb_mesh_view::owned, local_index[g] (-1: not resident).
struct HaloPlan {
  std::vector<int32_t> resident_gids, owned, local_index;
  int nowned = 0;
};
struct MeshTopology {
  int ndim = 1;
  double gmin[3] = {0, 0, 0}, gmax[3] = {1, 1, 1};
  int nleaf[3] = {1, 1, 1};
  bool periodic[6] = {false, false, false, false, false, false};
  const int32_t *leaf_map = nullptr;   // [nleaf[2]][nleaf[1]][nleaf[0]] -> global block id
  int nblocks_total = 0;
  const double *blk_xmin = nullptr, *blk_xmax = nullptr;   // [nblocks_total][3]
  const int32_t *owner = nullptr;                           // [nblocks_total]
  const int32_t *level = nullptr;                           // [nblocks_total] (FaceNeighbourLevels only)
};
// jb_mesh_view::blk_nbr_lev of block g: the level of the block behind each face (2 axis + upper), the
// block's own level at a mesh boundary that is not periodic and on an inactive axis
// (jaybenne.cpp:341-351) -- what a host that keeps a HALO COPY of g has to supply for it
inline void FaceNeighbourLevels(const MeshTopology &T, int g, int32_t out[6]) {
  for (int f = 0; f < 6; ++f) out[f] = T.level[g];
  long long l0[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
  for (int d = 0; d < T.ndim; ++d) {
    const double fine = (T.gmax[d] - T.gmin[d]) / (double)T.nleaf[d];
    l0[d] = (long long)std::llround((T.blk_xmin[3 * g + d] - T.gmin[d]) / fine);
    cnt[d] = (long long)std::llround((T.blk_xmax[3 * g + d] - T.blk_xmin[3 * g + d]) / fine);
  }
  for (int d = 0; d < T.ndim; ++d)
    for (int up = 0; up < 2; ++up) {
      long long l[3] = {l0[0], l0[1], l0[2]};   // (any leaf behind the face: 2:1 balance, one level there)
      l[d] = up ? l0[d] + cnt[d] : l0[d] - 1;
      if (l[d] < 0) { if (!T.periodic[2 * d]) continue; l[d] += T.nleaf[d]; }
      else if (l[d] >= T.nleaf[d]) { if (!T.periodic[2 * d + 1]) continue; l[d] -= T.nleaf[d]; }
      out[2 * d + up] = T.level[T.leaf_map[(l[2] * T.nleaf[1] + l[1]) * T.nleaf[0] + l[0]]];
    }
}
// leaf blocks that touch block b: the blocks under the one-leaf-wide shell of leaves around it
inline void TouchingBlocks(const MeshTopology &T, int b, std::vector<int32_t> *out) {
  long long lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // the shell's leaf range per axis, inclusive
  for (int d = 0; d < 3; ++d) {
    if (d >= T.ndim) continue;
    const double fine = (T.gmax[d] - T.gmin[d]) / (double)T.nleaf[d];
    lo[d] = (long long)std::llround((T.blk_xmin[3 * b + d] - T.gmin[d]) / fine) - 1;
    hi[d] = (long long)std::llround((T.blk_xmax[3 * b + d] - T.gmin[d]) / fine);
  }
  for (long long k = lo[2]; k <= hi[2]; ++k)
    for (long long j = lo[1]; j <= hi[1]; ++j)
      for (long long i = lo[0]; i <= hi[0]; ++i) {
        long long l[3] = {i, j, k};
        bool inside = true;
        for (int d = 0; d < T.ndim; ++d) {
          if (l[d] < 0) { if (T.periodic[2 * d]) l[d] += T.nleaf[d]; else inside = false; }
          else if (l[d] >= T.nleaf[d]) { if (T.periodic[2 * d + 1]) l[d] -= T.nleaf[d]; else inside = false; }
        }
        if (!inside) continue;
        const int32_t g = T.leaf_map[(l[2] * T.nleaf[1] + l[1]) * T.nleaf[0] + l[0]];
        if (g != b) out->push_back(g);
      }
}
inline HaloPlan PlanHalo(const MeshTopology &T, int rank, int rings = 1) {
  HaloPlan pl;
  std::vector<char> have((size_t)T.nblocks_total, 0), is_halo((size_t)T.nblocks_total, 0);
  std::vector<int32_t> frontier;
  for (int g = 0; g < T.nblocks_total; ++g)
    if (T.owner[g] == rank) { have[g] = 1; frontier.push_back(g); pl.resident_gids.push_back(g); }
  pl.nowned = (int)pl.resident_gids.size();
  for (int ring = 0; ring < rings; ++ring) {
    std::vector<int32_t> touched, next;
    for (int32_t b : frontier) TouchingBlocks(T, b, &touched);
    for (int32_t g : touched)
      if (!have[g]) { have[g] = 1; is_halo[g] = 1; next.push_back(g); }
    frontier.swap(next);
  }
  for (int g = 0; g < T.nblocks_total; ++g)
    if (is_halo[g]) pl.resident_gids.push_back(g);
  pl.owned.assign(pl.resident_gids.size(), 0);
  for (int q = 0; q < pl.nowned; ++q) pl.owned[q] = 1;
  pl.local_index.assign((size_t)T.nblocks_total, -1);
  for (size_t q = 0; q < pl.resident_gids.size(); ++q) pl.local_index[(size_t)pl.resident_gids[q]] = (int32_t)q;
  return pl;
}
// The refresh of the halo copies' interior cells after the owner changed a field (UpdateFluid:
// internal_energy; mcblock_driver.cpp:58-74 runs Parthenon's boundary exchange there): every rank
// can work out, without talking to anybody, which of its blocks the others keep copies of.
//   serve_*      (local block, cell) of the cells this rank sends, grouped by destination rank in
//                rank order -- jb_gather_cells packs them; send_counts[r] cells go to rank r
//   dst_*, src_* arguments of jb_fill_cells with one sample per destination: interior cell c of halo
//                copy lb takes element src_cell of the receive buffer (src_blk = -1), which holds
//                recv_counts[r] values from rank r, in rank order
// Cells are the interior cells of a block in (k, j, i) order, as flat indices into [nk][nj][ni].
struct HaloRefreshPlan {
  std::vector<int32_t> serve_blk, serve_cell, dst_blk, dst_cell, src_blk, src_cell;
  std::vector<int64_t> send_counts, recv_counts;
};
inline HaloRefreshPlan PlanHaloRefresh(const MeshTopology &T, int rank, int nranks, const int nx[3], int ng,
                                       int rings = 1) {
  HaloRefreshPlan pl;
  pl.send_counts.assign((size_t)nranks, 0);
  pl.recv_counts.assign((size_t)nranks, 0);
  const int is = ng, js = T.ndim >= 2 ? ng : 0, ks = T.ndim >= 3 ? ng : 0;
  const int ni = nx[0] + 2 * is, nj = nx[1] + 2 * js;
  std::vector<int32_t> interior;
  for (int k = 0; k < nx[2]; ++k)
    for (int j = 0; j < nx[1]; ++j)
      for (int i = 0; i < nx[0]; ++i) interior.push_back(((k + ks) * nj + (j + js)) * ni + (i + is));
  const HaloPlan mine = PlanHalo(T, rank, rings);
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) continue;
    const HaloPlan theirs = PlanHalo(T, r, rings);
    for (size_t q = (size_t)theirs.nowned; q < theirs.resident_gids.size(); ++q) {
      const int32_t g = theirs.resident_gids[q];
      if (T.owner[g] != rank) continue;          // rank r keeps a copy of MY block g
      for (int32_t c : interior) { pl.serve_blk.push_back(mine.local_index[(size_t)g]); pl.serve_cell.push_back(c); }
      pl.send_counts[(size_t)r] += (int64_t)interior.size();
    }
  }
  int64_t off = 0;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) continue;
    for (size_t q = (size_t)mine.nowned; q < mine.resident_gids.size(); ++q) {
      const int32_t g = mine.resident_gids[q];
      if (T.owner[g] != r) continue;             // my copy of rank r's block g
      for (int32_t c : interior) {
        pl.dst_blk.push_back((int32_t)q); pl.dst_cell.push_back(c);
        pl.src_blk.push_back(-1); pl.src_cell.push_back((int32_t)off++);
      }
      pl.recv_counts[(size_t)r] += (int64_t)interior.size();
    }
  }
  return pl;
}

// ---- tasks (jaybenne.hpp:59-76) ----------------------------------------------------------------
inline TaskStatus UpdateDerivedTransportFields(MeshData *md, const Real dt) {
  return Check(jb_update_derived_transport_fields(md->ctx(), md->mesh(), dt));
}

// SourcePhotons<T, ST>(md, t_start, dt): per_block = the MeshBlockData instantiation used at
// initialisation (one call per block, sourcing.cpp:68-69)
inline TaskStatus SourcePhotons(MeshData *md, SourceType st, const Real t_start, const Real dt,
                                bool per_block = false) {
  const jb_params &p = md->pkg().params();
  if (p.source_strategy == JB_STRATEGY_ENERGY)  // sourcing.cpp:38
    throw Error(JB_ERR_INVALID, "Energy source strategy not implemented!");
  if (st == SourceType::emission && !p.do_emission) return TaskStatus::complete;
  const int type = st == SourceType::thermal ? JB_SOURCE_THERMAL : JB_SOURCE_EMISSION;
  const int nb = md->nblocks();
  std::vector<int32_t> nper(nb, 0);
  Check(jb_source_photons_count(md->ctx(), md->mesh(), type, dt, per_block ? 1 : nb,
                                SourceEpoch(md->cycle, st), nper.data(), md->prefix_dev()));
  // (this rank holds the whole mesh: global id = local index, global counts = local counts)
  std::vector<int32_t> gid(nb);
  std::vector<long long> all(nb);
  for (int b = 0; b < nb; ++b) { gid[b] = b; all[b] = nper[b]; }
  const SourcePlan pl = PlanSource(nper, gid, all, md->next_id, md->swarm.n);
  md->Reserve(md->swarm.n + pl.total_local);
  Check(jb_source_photons_fill(md->ctx(), md->mesh(), &md->swarm, type, t_start, dt, nper.data(),
                               md->prefix_dev(), pl.slot_base.data(), pl.id_base.data()));
  md->swarm.n += pl.total_local;
  md->next_id = pl.next_id;
  return TaskStatus::complete;
}

inline TaskStatus TransportPhotons(MeshData *md, const Real t_start, const Real dt,
                                   bool fuse_census_tally = false) {
  return Check(jb_transport_photons(md->ctx(), md->mesh(), &md->swarm, t_start, dt, 0,
                                    md->swarm.n, fuse_census_tally ? 1 : 0));
}
inline TaskStatus TransportPhotons_DDMC(MeshData *md, const Real t_start, const Real dt,
                                        bool fuse_census_tally = false) {
  return Check(jb_transport_photons_ddmc(md->ctx(), md->mesh(), &md->swarm, t_start, dt, 0,
                                         md->swarm.n, fuse_census_tally ? 1 : 0));
}
inline TaskStatus SampleDDMCBlockFace(MeshData *md) {
  return Check(jb_sample_ddmc_block_face(md->ctx(), md->mesh(), &md->swarm, 0, md->swarm.n));
}
inline TaskStatus CheckCompletion(MeshData *md, const Real t_end) {
  int64_t unfinished = 0;
  return Check(jb_check_completion(md->ctx(), &md->swarm, t_end, &unfinished));
}
inline TaskStatus EvaluateRadiationEnergy(MeshData *md) {
  return Check(jb_evaluate_radiation_energy(md->ctx(), md->mesh(), &md->swarm));
}
inline TaskStatus UpdateFluid(MeshData *md) { return Check(jb_update_fluid(md->ctx(), md->mesh())); }
// jaybenne::DefragParticles -- jaybenne.cpp:499-509 (scheduled by no task list of the reference):
// the swarm sorted by block and cell, in place (jaybenne_amd.h: jb_defrag_particles)
inline TaskStatus DefragParticles(MeshData *md) {
  return Check(jb_defrag_particles(md->ctx(), md->mesh(), &md->swarm));
}
inline Real EstimateTimestepMesh(MeshData *md) { return jb_estimate_timestep(md->ctx()); }

// jaybenne::InitializeRadiation(mbd, is_thermal) -- jaybenne.cpp:570-578
inline void InitializeRadiation(MeshData *md, bool is_thermal) {
  if (is_thermal) SourcePhotons(md, SourceType::thermal, 0.0, 0.0, /*per_block=*/true);
  EvaluateRadiationEnergy(md);
}

// A trace range for the lifetime of the object (jb_range_push / jb_range_pop: ROCTx, a no-op without the marker
// library) -- the reference's Kokkos::Profiling::pushRegion / popRegion, jaybenne.cpp:87,115,127,145
struct TraceRange {
  explicit TraceRange(const char *name) { jb_range_push(name); }
  ~TraceRange() { jb_range_pop(); }
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
};

// jaybenne::RadiationStep(pmesh, t_start, dt) for one rank -- jaybenne.cpp:68-151
inline TaskStatus RadiationStep(MeshData *md, const Real t_start, const Real dt) {
  TraceRange timestep("Jaybenne::Timestep");
  const jb_params &p = md->pkg().params();
  md->cycle += 1;
  UpdateDerivedTransportFields(md, dt);
  SourcePhotons(md, SourceType::emission, t_start, dt);
  Check(jb_zero_energy_tally(md->ctx(), md->mesh()));
  jb_transport_stats before{}, after{};
  Check(jb_get_transport_stats(md->ctx(), &before, 0));
  {
    TraceRange loop("Jaybenne::TransportLoop");   // (one pass: every block crossing is resolved in flight)
    if (p.use_ddmc) TransportPhotons_DDMC(md, t_start, dt, /*fuse_census_tally=*/true);
    else TransportPhotons(md, t_start, dt, /*fuse_census_tally=*/true);
  }
  Check(jb_get_transport_stats(md->ctx(), &after, 0));
  if (after.n_outgoing != before.n_outgoing)
    throw Error(JB_ERR_INVALID, "particles left for another rank in a single-rank step");
  if (after.n_absorbed != before.n_absorbed || after.n_escaped != before.n_escaped)
    Check(jb_remove_marked_particles(md->ctx(), &md->swarm));
  md->events += after.n_events - before.n_events;
  if (CheckCompletion(md, t_start + dt) != TaskStatus::complete) return TaskStatus::iterate;
  UpdateFluid(md);
  // (not in the reference's task list: DefragParticles on the library's schedule, or every k-th cycle)
  if (md->defrag_interval < 0) {
    int32_t sorted = 0;
    Check(jb_defrag_policy(md->ctx(), md->mesh(), &md->swarm, after.n_events - before.n_events,
                           JB_DEFRAG_DECIDE_AND_SORT, &sorted));
    md->defrags += sorted;
  } else if (md->defrag_interval > 0 && ++md->steps_since_defrag >= md->defrag_interval) {
    DefragParticles(md);
    md->steps_since_defrag = 0;
    ++md->defrags;
  }
  return TaskStatus::complete;
}

}  // namespace jaybenne_amd

#endif  // JAYBENNE_AMD_HPP_
