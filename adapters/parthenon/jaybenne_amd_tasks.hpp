// jaybenne_amd_tasks.hpp -- the reference's package interface (src/jaybenne/jaybenne.hpp:48-78)
// implemented on libjaybenne_amd.so.  SOURCE ONLY: needs Parthenon + Kokkos(HIP) + singularity,
// which this repository's image does not have (see README.md next to this file).
#ifndef JAYBENNE_AMD_TASKS_HPP_
#define JAYBENNE_AMD_TASKS_HPP_

#include <memory>
#include <string>
#include <vector>

#include <parthenon/driver.hpp>
#include <parthenon/package.hpp>

#include "jaybenne_amd.h"
#include "jaybenne_amd.hpp"   // PlanSource, SourceEpoch: the host-side arithmetic shared with the tested hosts
#include "jaybenne_config.hpp"      // EOS / Opacity / Scattering aliases, HOST_* variables
#include "jaybenne_variables.hpp"   // field::jaybenne::*, particle::photons::*, photons_swarm_name

using namespace parthenon;
using namespace parthenon::driver::prelude;
using namespace parthenon::package::prelude;

namespace jaybenne {

enum class SourceStrategy { uniform, energy };   // jaybenne.hpp:55
enum class SourceType { thermal, emission };     // jaybenne.hpp:56

// What the package keeps between calls (stored as Param<std::shared_ptr<AmdState>>("amd_state")).
struct AmdState {
  jb_context *ctx = nullptr;
  jb_mesh *mesh = nullptr;           // rebuilt when the mesh changes (remesh / load balance)
  jb_swarm_view sw{};                // rank-wide photon pool
  std::vector<ParArray1D<Real>> pool_f64;      // x y z vx vy vz t w e
  std::vector<ParArray1D<int>> pool_i32;       // ip jp kp blk status
  ParArray1D<std::uint64_t> pool_id, pool_rng;
  ParArray1D<int> prefix;            // per-cell exclusive source counts (SourcePhotons phase 1)
  ParArray1D<std::int64_t> records;  // hand-off records, JB_RECORD_WORDS words each
  std::uint64_t next_id = 0;         // first unused stream id (kept in step on every rank)
  std::uint64_t cycle = 0;           // RadiationStep counter (keys the per-cell source streams:
                                     // jaybenne_amd::SourceEpoch); 0 = initialisation
  std::vector<int> pending_initial_blocks;   // local ids recorded by InitializeRadiation(mbd)
  bool initial_source_done = false;
  int mesh_generation = -1;          // Mesh::nbtotal / block list stamp the view was built for
  // halo copies (several ranks): read-only copies of the other ranks' blocks that touch this rank's
  // (jaybenne_amd::PlanHalo) -- eleven cell arrays each, owned by the adapter -- and the plan that
  // refreshes their material state from the owners (jaybenne_amd::PlanHaloRefresh)
  int nowned = 0, nhalo = 0;
  ParArray1D<Real> halo_fields;      // [nhalo][11][ntot]
  jaybenne_amd::HaloRefreshPlan refresh;
  ParArray1D<int> refresh_idx;       // serve_blk | serve_cell | dst_blk | dst_cell | src_blk | src_cell
  ParArray1D<Real> refresh_send, refresh_recv;
  ~AmdState();
};

std::shared_ptr<StateDescriptor> Initialize(ParameterInput *pin, Opacity &opacity,
                                            Scattering &scattering, EOS &eos);

// Tasks -- jaybenne.hpp:59-69
TaskStatus TransportPhotons(MeshData<Real> *md, const Real t_start, const Real dt);
TaskStatus TransportPhotons_DDMC(MeshData<Real> *md, const Real t_start, const Real dt);
TaskStatus SampleDDMCBlockFace(MeshData<Real> *md);
TaskStatus CheckCompletion(MeshData<Real> *md, const Real t_end);
template <typename T, SourceType ST>
TaskStatus SourcePhotons(T *md, const Real t_start, const Real dt);
TaskStatus DefragParticles(MeshBlock *pmb);
TaskStatus UpdateDerivedTransportFields(MeshData<Real> *md, const Real dt);
template <typename T>
TaskStatus EvaluateRadiationEnergy(T *md);
TaskStatus UpdateFluid(MeshData<Real> *md);
// The initial thermal source of every block recorded by InitializeRadiation(mbd, true), in ONE
// collective step (an MPI_Allreduce): call once on every rank after ParthenonInitPackagesAndMesh
// (reference main.cpp:42) -- ranks hold different numbers of blocks, so the per-block hook itself
// must not communicate.
TaskStatus FlushInitialSource(Mesh *pmesh);
// The halo copies' density / sie / internal energy from their owners (gather -> MPI_Alltoallv ->
// fill).  Run by the adapter when the mesh view is built; the HOST adds it behind its own update of
// the material state -- mcblock_driver.cpp:58-74: after the boundary exchange and sie = u / rho of
// HostUpdateTasks -- whenever do_feedback changes that state.
TaskStatus RefreshHaloCopies(MeshData<Real> *md);

TaskCollection RadiationStep(Mesh *pmesh, const Real t_start, const Real dt);   // jaybenne.hpp:72
Real EstimateTimestepMesh(MeshData<Real> *md);                                  // jaybenne.hpp:75
void InitializeRadiation(MeshBlockData<Real> *mbd, const bool is_thermal);      // jaybenne.hpp:76

// Copies the pool into the registered `photons` swarm (outputs, restarts).
TaskStatus ExportToParthenonSwarm(MeshData<Real> *md);

}  // namespace jaybenne

#endif  // JAYBENNE_AMD_TASKS_HPP_
