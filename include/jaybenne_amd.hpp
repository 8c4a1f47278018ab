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
#include <functional>
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
  uint32_t epoch = 0;      // source-call counter (keys the per-cell rounding streams)
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
// the cycle and the source type, so that every rank -- whatever the number of blocks it calls the
// source for -- uses the same one: 0 for the initial thermal source, then 2 cycle + type.
// (jb_source_photons_count accepts epochs below 2^20: half a million cycles.)
inline uint32_t SourceEpoch(uint64_t cycle, SourceType st) {
  return (uint32_t)(2u * cycle + (st == SourceType::emission ? 1u : 0u));
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
  Check(jb_source_photons_count(md->ctx(), md->mesh(), type, dt, per_block ? 1 : nb, md->epoch,
                                nper.data(), md->prefix_dev()));
  md->epoch += 1;
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

// jaybenne::RadiationStep(pmesh, t_start, dt) for one rank -- jaybenne.cpp:68-151
inline TaskStatus RadiationStep(MeshData *md, const Real t_start, const Real dt) {
  const jb_params &p = md->pkg().params();
  UpdateDerivedTransportFields(md, dt);
  SourcePhotons(md, SourceType::emission, t_start, dt);
  Check(jb_zero_energy_tally(md->ctx(), md->mesh()));
  jb_transport_stats before{}, after{};
  Check(jb_get_transport_stats(md->ctx(), &before, 0));
  if (p.use_ddmc) TransportPhotons_DDMC(md, t_start, dt, /*fuse_census_tally=*/true);
  else TransportPhotons(md, t_start, dt, /*fuse_census_tally=*/true);
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
    Check(jb_defrag_policy(md->ctx(), md->mesh(), &md->swarm, after.n_events - before.n_events, &sorted));
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
