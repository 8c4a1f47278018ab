// jaybenne_amd_tasks.cpp -- the reference's tasks (src/jaybenne/jaybenne.hpp:59-76) as calls into
// libjaybenne_amd.so.  SOURCE ONLY: written against Parthenon's public API as the reference uses
// it (call sites cited per function); never compiled in this repository -- Parthenon, Kokkos and
// singularity are absent from its image.  The compiled, tested twin of this layer is
// include/jaybenne_amd.hpp + examples/mcblock_amd.cpp (stand-ins for MeshData / StateDescriptor)
// and examples/handoff_mpi.cpp (the MPI hand-off).
#include "jaybenne_amd_tasks.hpp"

#include <mpi.h>

#include <algorithm>
#include <cfloat>
#include <numeric>

namespace jaybenne {

namespace fj = field::jaybenne;
namespace fjh = field::jaybenne::host;
namespace ph = particle::photons;

#define JB_REQUIRE(call) PARTHENON_REQUIRE((call) >= 0, jb_last_error())

static TaskStatus Status(jb_status s) {
  PARTHENON_REQUIRE(s >= 0, jb_last_error());
  return s == JB_ITERATE ? TaskStatus::iterate
                         : (s == JB_INCOMPLETE ? TaskStatus::incomplete : TaskStatus::complete);
}

AmdState::~AmdState() {
  if (mesh) jb_mesh_destroy(mesh);
  if (ctx) jb_finalize(ctx);
}

static AmdState &State(Mesh *pm) {
  return *pm->packages.Get("jaybenne")->Param<std::shared_ptr<AmdState>>("amd_state");
}

//----------------------------------------------------------------------------------------
// jaybenne::Initialize -- jaybenne.cpp:158-266: same keys, defaults, requirements, fields and
// swarm registration; the random pool of jaybenne.cpp:192-197 is replaced by one stream per
// photon inside the library (keyed by the same, unadjusted, seed).
std::shared_ptr<StateDescriptor> Initialize(ParameterInput *pin, Opacity &opacity,
                                            Scattering &scattering, EOS &eos) {
  auto pkg = std::make_shared<StateDescriptor>("jaybenne");
  jb_params p{};
  p.num_particles = pin->GetInteger("jaybenne", "num_particles");
  p.dt = pin->GetOrAddReal("jaybenne", "dt", std::numeric_limits<Real>::max());
  p.min_swarm_occupancy = pin->GetOrAddReal("jaybenne", "min_swarm_occupancy", 0.0);
  PARTHENON_REQUIRE(p.min_swarm_occupancy >= 0.0 && p.min_swarm_occupancy < 1.0,
                    "Minimum allowable swarm occupancy must be >= 0 and less than 1");
  p.numin = pin->GetOrAddReal("jaybenne", "numin", std::numeric_limits<Real>::min());
  p.numax = pin->GetOrAddReal("jaybenne", "numax", std::numeric_limits<Real>::max());
  p.unique_rank_seeds = pin->GetOrAddBoolean("jaybenne", "unique_rank_seeds", true);
  p.seed = pin->GetOrAddInteger("jaybenne", "seed", 123);
  p.max_transport_iterations = pin->GetOrAddInteger("jaybenne", "max_transport_iterations", 10000);
  p.use_ddmc = pin->GetOrAddBoolean("jaybenne", "use_ddmc", false);
  p.tau_ddmc = pin->GetOrAddReal("jaybenne", "tau_ddmc", 5.0);
  const std::string strategy = pin->GetOrAddString("jaybenne", "source_strategy", "uniform");
  if (strategy == "uniform") {
    p.source_strategy = JB_STRATEGY_UNIFORM;
  } else if (strategy == "energy") {
    p.source_strategy = JB_STRATEGY_ENERGY;
  } else {
    PARTHENON_FAIL("Only uniform or energy source strategies supported!");
  }
  p.do_emission = pin->GetOrAddBoolean("jaybenne", "do_emission", true);
  p.do_feedback = pin->GetOrAddBoolean("jaybenne", "do_feedback", true);
  p.rank = Globals::my_rank;

  // the host's model objects as tagged POD (jaybenne_amd.h): singularity IdealGas, Gray or
  // EPBremss, GrayS or ThomsonS.  The variants (opacity.hpp:23-30) carry no type tag the adapter
  // could read back, so the host says which alternative it built and with which code -> CGS
  // scales: the optional <jaybenne> keys below default to the gray models in CGS, for which the
  // coefficients are read off the objects themselves (frequency independent, linear in rho).
  auto units = opacity.GetRuntimePhysicalConstants();
  jb_eos e{JB_EOS_IDEAL_GAS, 0, eos.GruneisenParamFromDensityTemperature(1.0, 1.0),
           eos.SpecificHeatFromDensityTemperature(1.0, 1.0)};
  const bool epbremss = pin->GetOrAddString("jaybenne", "amd_opacity_model", "gray") == "ep_bremss";
  const bool thomson = pin->GetOrAddString("jaybenne", "amd_scattering_model", "gray") == "thomson";
  const Real ts = pin->GetOrAddReal("mcblock", "time_scale", 1.), ms = pin->GetOrAddReal("mcblock", "mass_scale", 1.),
             ls = pin->GetOrAddReal("mcblock", "length_scale", 1.), tks = pin->GetOrAddReal("mcblock", "temperature_scale", 1.);
  const Real apm = pin->GetOrAddReal("mcblock", "apm", 1.);
  jb_opacity o{epbremss ? JB_OPAC_EPBREMSS : JB_OPAC_GRAY, 0,
               epbremss ? 0.0 : opacity.AbsorptionCoefficient(1.0, 1.0, 1.0), units.c, units.sb,
               ts, ms, ls, tks};
  jb_scattering s{thomson ? JB_SCAT_THOMSON : JB_SCAT_GRAY, 0,
                  thomson ? 0.0 : scattering.TotalScatteringCoefficient(1.0, 1.0, 1.0) * apm, apm,
                  ts, ms, ls, tks};

  auto st = std::make_shared<AmdState>();
  int device = 0;
  (void)hipGetDevice(&device);   // the device Kokkos::initialize selected for this rank
  JB_REQUIRE(jb_initialize(&p, &e, &o, &s, device, &st->ctx));
  pkg->AddParam<>("amd_state", st);
  pkg->AddParam<>("num_particles", (int)p.num_particles);
  pkg->AddParam<>("dt", p.dt);
  pkg->AddParam<>("min_swarm_occupancy", p.min_swarm_occupancy);
  pkg->AddParam<>("numin", p.numin);
  pkg->AddParam<>("numax", p.numax);
  pkg->AddParam<>("speed_of_light", units.c);
  pkg->AddParam<>("stefan_boltzmann", units.sb);
  pkg->AddParam<>("unique_rank_seeds", (bool)p.unique_rank_seeds);
  pkg->AddParam<>("seed", jb_param_seed(st->ctx));
  pkg->AddParam<>("max_transport_iterations", (int)p.max_transport_iterations);
  pkg->AddParam<>("use_ddmc", (bool)p.use_ddmc);
  pkg->AddParam<>("tau_ddmc", p.tau_ddmc);
  pkg->AddParam<>("do_emission", (bool)p.do_emission);
  pkg->AddParam<>("do_feedback", (bool)p.do_feedback);
  pkg->AddParam<>("eos_d", eos.GetOnDevice());
  pkg->AddParam<>("opacity_d", opacity.GetOnDevice());
  pkg->AddParam<>("scattering_d", scattering.GetOnDevice());

  // swarm + fields exactly as jaybenne.cpp:236-260 (the swarm is an output view of the pool)
  Metadata swarm_metadata({Metadata::Provides, Metadata::None});
  pkg->AddSwarm(photons_swarm_name, swarm_metadata);
  Metadata real_swarmvalue_metadata({Metadata::Real});
  pkg->AddSwarmValue(ph::weight::name(), photons_swarm_name, real_swarmvalue_metadata);
  pkg->AddSwarmValue(ph::energy::name(), photons_swarm_name, real_swarmvalue_metadata);
  pkg->AddSwarmValue(ph::time::name(), photons_swarm_name, real_swarmvalue_metadata);
  Metadata vec(std::vector<MetadataFlag>{Metadata::Real}, std::vector<int>{3});
  pkg->AddSwarmValue(ph::v::name(), photons_swarm_name, vec);
  Metadata ivec(std::vector<MetadataFlag>{Metadata::Integer}, std::vector<int>{3});
  pkg->AddSwarmValue(ph::ijk::name(), photons_swarm_name, ivec);
  Metadata m({Metadata::Cell, Metadata::Independent});
  pkg->AddField(fj::energy_tally::name(), m);
  pkg->AddField(fj::fleck_factor::name(), m);
  Metadata m_onecopy({Metadata::Cell, Metadata::OneCopy});
  pkg->AddField(fj::source_ew_per_cell::name(), m_onecopy);
  pkg->AddField(fj::source_num_per_cell::name(), m_onecopy);
  pkg->AddField(fj::energy_delta::name(), m_onecopy);
  Metadata m_face({Metadata::Face, Metadata::Derived, Metadata::FillGhost});
  pkg->AddField(fj::ddmc_face_prob::name(), m_face);
  pkg->EstimateTimestepMesh = EstimateTimestepMesh;
  return pkg;
}

//----------------------------------------------------------------------------------------
// MeshData -> jb_mesh_view.  Rebuilt when the block list changes.
static void EnsureMeshView(MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  AmdState &st = State(pm);
  if (st.mesh && st.mesh_generation == pm->nbtotal) return;
  if (st.mesh) jb_mesh_destroy(st.mesh);
  static auto desc =
      MakePackDescriptor<fjh::density, fjh::sie, fjh::update_energy, fj::fleck_factor,
                         fj::energy_tally, fj::energy_delta, fj::source_ew_per_cell,
                         fj::source_num_per_cell, fj::ddmc_face_prob>(pm->resolved_packages.get());
  auto vmesh = desc.GetPack(md);
  const int nb = md->NumBlocks();
  const int ndim = pm->ndim;
  const auto ib = md->GetBoundsI(IndexDomain::interior);
  const auto jb = md->GetBoundsJ(IndexDomain::interior);
  const auto kb = md->GetBoundsK(IndexDomain::interior);

  jb_mesh_view v{};
  v.ndim = ndim;
  v.ng = Globals::nghost;
  v.nblocks = nb;
  v.nblocks_total = pm->nbtotal;
  v.rank = Globals::my_rank;
  v.nx[0] = ib.e - ib.s + 1; v.nx[1] = jb.e - jb.s + 1; v.nx[2] = kb.e - kb.s + 1;
  const auto &ms = pm->mesh_size;
  for (int d = 0; d < 3; ++d) { v.gmin[d] = ms.xmin(static_cast<CoordinateDirection>(d + 1));
                                v.gmax[d] = ms.xmax(static_cast<CoordinateDirection>(d + 1)); }
  // <parthenon/swarm> boundaries: the names registered by mcblock::ProblemModifier (mcblock.cpp:271-282)
  auto bc_of = [&](const std::string &name) {
    return name == "jaybenne_reflecting" ? JB_BC_REFLECT : (name == "periodic" ? JB_BC_PERIODIC : JB_BC_OUTFLOW);
  };
  for (int f = 0; f < 6; ++f) v.bc[f] = bc_of(pm->mesh_swarm_bc_names[f]);   // ix1, ox1, ix2, ...

  // leaves of the block tree: every global block with its logical location, level and rank
  const int maxlev = pm->GetCurrentLevel() - pm->GetRootLevel();
  const auto nrb = pm->nrbx;                       // root-grid blocks per dimension
  std::vector<int32_t> owner(pm->nbtotal), local_index(pm->nbtotal, -1), gid(nb), level(nb),
      nbr_lev(6 * nb);
  for (int d = 0; d < 3; ++d) v.nleaf[d] = d < ndim ? nrb[d] << maxlev : 1;
  std::vector<int32_t> leaf_map((size_t)v.nleaf[0] * v.nleaf[1] * v.nleaf[2], 0);
  const auto &locs = pm->GetLocList();             // logical location of every global block
  const auto &ranks = pm->GetRankList();
  for (int g = 0; g < pm->nbtotal; ++g) {
    owner[g] = ranks[g];
    const int lev = locs[g].level() - pm->GetRootLevel();
    const int span = 1 << (maxlev - lev);
    const int l0[3] = {(int)locs[g].lx1() * span, ndim > 1 ? (int)locs[g].lx2() * span : 0,
                       ndim > 2 ? (int)locs[g].lx3() * span : 0};
    for (int k = 0; k < (ndim > 2 ? span : 1); ++k)
      for (int j = 0; j < (ndim > 1 ? span : 1); ++j)
        for (int i = 0; i < span; ++i)
          leaf_map[((size_t)(l0[2] + k) * v.nleaf[1] + (l0[1] + j)) * v.nleaf[0] + (l0[0] + i)] = g;
  }
  // every block of the mesh by global id: corners and level from its logical location
  std::vector<double> gxmin(3 * (size_t)pm->nbtotal), gxmax(3 * (size_t)pm->nbtotal);
  std::vector<int32_t> glevel(pm->nbtotal);
  for (int g = 0; g < pm->nbtotal; ++g) {
    glevel[g] = locs[g].level() - pm->GetRootLevel();
    const std::int64_t lx[3] = {locs[g].lx1(), locs[g].lx2(), locs[g].lx3()};
    for (int d = 0; d < 3; ++d) {
      const double ext = d < ndim ? (v.gmax[d] - v.gmin[d]) / (double)(nrb[d] << glevel[g]) : v.gmax[d] - v.gmin[d];
      gxmin[3 * g + d] = d < ndim ? v.gmin[d] + (double)lx[d] * ext : v.gmin[d];
      gxmax[3 * g + d] = d < ndim ? v.gmin[d] + (double)(lx[d] + 1) * ext : v.gmax[d];
    }
  }
  jaybenne_amd::MeshTopology topo;
  topo.ndim = ndim;
  for (int d = 0; d < 3; ++d) { topo.gmin[d] = v.gmin[d]; topo.gmax[d] = v.gmax[d]; topo.nleaf[d] = v.nleaf[d]; }
  for (int f = 0; f < 6; ++f) topo.periodic[f] = pm->mesh_bcs[f] == BoundaryFlag::periodic;
  topo.leaf_map = leaf_map.data(); topo.nblocks_total = pm->nbtotal;
  topo.blk_xmin = gxmin.data(); topo.blk_xmax = gxmax.data(); topo.owner = owner.data(); topo.level = glevel.data();
  // this rank's blocks (MeshData holds them in ascending global id) + halo copies of the blocks of
  // other ranks that touch them: a photon that wanders across the rank boundary is tracked on and
  // handed over once, when its history is over (2 transport iterations per cycle instead of ~80)
  const jaybenne_amd::HaloPlan halo = Globals::nranks > 1 ? jaybenne_amd::PlanHalo(topo, Globals::my_rank)
                                                          : jaybenne_amd::HaloPlan{};
  const int nres = Globals::nranks > 1 ? (int)halo.resident_gids.size() : nb;
  PARTHENON_REQUIRE(Globals::nranks == 1 || halo.nowned == nb, "MeshData does not hold every block of this rank");
  st.nowned = nb;
  st.nhalo = nres - nb;
  const size_t ntot = (size_t)(v.nx[0] + 2 * v.ng) * (ndim > 1 ? v.nx[1] + 2 * v.ng : 1) *
                      (ndim > 2 ? v.nx[2] + 2 * v.ng : 1);
  if (st.nhalo > 0) st.halo_fields = ParArray1D<Real>("jb halo copies", (size_t)st.nhalo * 11 * ntot);
  gid.resize(nres); level.resize(nres); nbr_lev.resize(6 * (size_t)nres);
  std::vector<int32_t> owned_flag(nres, 1);
  std::vector<double> xmin(3 * (size_t)nres), xmax(3 * (size_t)nres), dxs(3 * (size_t)nres);
  std::vector<double *> tab[11];
  for (auto &t : tab) t.resize(nres);
  for (int q = nb; q < nres; ++q) {        // the halo copies: geometry from the tree, arrays of the adapter
    const int g = halo.resident_gids[q];
    gid[q] = g; local_index[g] = q; level[q] = glevel[g]; owned_flag[q] = 0;
    for (int d = 0; d < 3; ++d) {
      xmin[3 * q + d] = gxmin[3 * g + d]; xmax[3 * q + d] = gxmax[3 * g + d];
      dxs[3 * q + d] = (gxmax[3 * g + d] - gxmin[3 * g + d]) / (double)v.nx[d];
    }
    jaybenne_amd::FaceNeighbourLevels(topo, g, &nbr_lev[6 * (size_t)q]);
    for (int f = 0; f < 11; ++f) tab[f][q] = st.halo_fields.data() + ((size_t)(q - nb) * 11 + f) * ntot;
  }
  auto vmesh_h = vmesh;   // (device pack: element addresses are taken on the host, not dereferenced)
  for (int b = 0; b < nb; ++b) {
    auto pmb = md->GetBlockData(b)->GetBlockPointer();
    gid[b] = pmb->gid;
    local_index[pmb->gid] = b;
    level[b] = pmb->loc.level() - pm->GetRootLevel();
    const auto &coords = pmb->coords;
    for (int d = 0; d < 3; ++d) {
      const auto dir = static_cast<CoordinateDirection>(d + 1);
      const auto &bs = pmb->block_size;
      xmin[3 * b + d] = bs.xmin(dir); xmax[3 * b + d] = bs.xmax(dir);
      dxs[3 * b + d] = coords.Dxc(dir);   // (inactive dimensions: the full extent, one cell)
    }
    // neighbour level per face; own level at a physical boundary (jaybenne.cpp:341-351)
    const int off[6][3] = {{0, 0, -1}, {0, 0, 1}, {0, -1, 0}, {0, 1, 0}, {-1, 0, 0}, {1, 0, 0}};
    for (int f = 0; f < 6; ++f)
      nbr_lev[6 * b + f] = vmesh.IsPhysicalBoundary(b, off[f][0], off[f][1], off[f][2])
                               ? level[b]
                               : vmesh.GetLevel(b, off[f][0], off[f][1], off[f][2]) - pm->GetRootLevel();
    tab[0][b] = &vmesh_h(b, fjh::density(), 0, 0, 0);
    tab[1][b] = &vmesh_h(b, fjh::sie(), 0, 0, 0);
    tab[2][b] = &vmesh_h(b, fjh::update_energy(), 0, 0, 0);
    tab[3][b] = &vmesh_h(b, fj::fleck_factor(), 0, 0, 0);
    tab[4][b] = &vmesh_h(b, fj::energy_tally(), 0, 0, 0);
    tab[5][b] = &vmesh_h(b, fj::energy_delta(), 0, 0, 0);
    tab[6][b] = &vmesh_h(b, fj::source_ew_per_cell(), 0, 0, 0);
    tab[7][b] = &vmesh_h(b, fj::source_num_per_cell(), 0, 0, 0);
    tab[8][b] = &vmesh_h(b, TopologicalElement::F1, fj::ddmc_face_prob(), 0, 0, 0);
    tab[9][b] = &vmesh_h(b, TopologicalElement::F2, fj::ddmc_face_prob(), 0, 0, 0);
    tab[10][b] = &vmesh_h(b, TopologicalElement::F3, fj::ddmc_face_prob(), 0, 0, 0);
  }
  v.leaf_map = leaf_map.data(); v.owner = owner.data(); v.local_index = local_index.data();
  v.nblocks = nres;
  v.gid = gid.data(); v.owned = owned_flag.data();
  v.blk_xmin = xmin.data(); v.blk_xmax = xmax.data(); v.blk_dx = dxs.data();
  v.blk_level = level.data(); v.blk_nbr_lev = nbr_lev.data();
  v.rho = tab[0].data(); v.sie = tab[1].data(); v.u = tab[2].data(); v.fleck = tab[3].data();
  v.tally = tab[4].data(); v.edelta = tab[5].data(); v.src_ew = tab[6].data(); v.src_num = tab[7].data();
  v.P1 = tab[8].data(); v.P2 = tab[9].data(); v.P3 = tab[10].data();
  JB_REQUIRE(jb_mesh_create(st.ctx, &v, &st.mesh));
  st.mesh_generation = pm->nbtotal;
  if (st.prefix.size() < (size_t)nres * v.nx[0] * v.nx[1] * v.nx[2])
    st.prefix = ParArray1D<int>("jb prefix", (size_t)nres * v.nx[0] * v.nx[1] * v.nx[2]);
  if (st.nhalo > 0 || Globals::nranks > 1) {
    // which cells travel when the halo copies are refreshed: worked out on every rank by itself
    st.refresh = jaybenne_amd::PlanHaloRefresh(topo, Globals::my_rank, Globals::nranks, v.nx, v.ng);
    const auto &rp = st.refresh;
    const size_t ns = rp.serve_blk.size(), nd = rp.dst_blk.size();
    st.refresh_idx = ParArray1D<int>("jb halo plan", std::max<size_t>(2 * ns + 4 * nd, 1));
    auto idx_h = Kokkos::create_mirror_view(st.refresh_idx);
    for (size_t q = 0; q < ns; ++q) { idx_h(q) = rp.serve_blk[q]; idx_h(ns + q) = rp.serve_cell[q]; }
    for (size_t q = 0; q < nd; ++q) {
      idx_h(2 * ns + q) = rp.dst_blk[q]; idx_h(2 * ns + nd + q) = rp.dst_cell[q];
      idx_h(2 * ns + 2 * nd + q) = rp.src_blk[q]; idx_h(2 * ns + 3 * nd + q) = rp.src_cell[q];
    }
    Kokkos::deep_copy(st.refresh_idx, idx_h);
    st.refresh_send = ParArray1D<Real>("jb halo send", std::max<size_t>(ns, 1));
    st.refresh_recv = ParArray1D<Real>("jb halo recv", std::max<size_t>(nd, 1));
    RefreshHaloCopies(md);
  }
}

// jaybenne_amd.h: jb_gather_cells -> MPI_Alltoallv -> jb_fill_cells, per field (the role Parthenon's
// boundary exchange plays for ghost zones, mcblock_driver.cpp:58-74, for the halo copies' interiors)
TaskStatus RefreshHaloCopies(MeshData<Real> *md) {
  AmdState &st = State(md->GetParentPointer());
  if (Globals::nranks == 1) return TaskStatus::complete;
  const auto &rp = st.refresh;
  const size_t ns = rp.serve_blk.size(), nd = rp.dst_blk.size();
  const int *idx = st.refresh_idx.data();
  std::vector<int> sc(Globals::nranks), sd(Globals::nranks), rc(Globals::nranks), rd(Globals::nranks);
  int so = 0, ro = 0;
  for (int r = 0; r < Globals::nranks; ++r) {
    sc[r] = (int)rp.send_counts[r]; sd[r] = so; so += sc[r];
    rc[r] = (int)rp.recv_counts[r]; rd[r] = ro; ro += rc[r];
  }
  for (int field : {JB_FIELD_RHO, JB_FIELD_SIE, JB_FIELD_U}) {
    JB_REQUIRE(jb_gather_cells(st.ctx, st.mesh, field, (std::int64_t)ns, idx, idx + ns, st.refresh_send.data()));
    JB_REQUIRE(jb_synchronize(st.ctx));
    // (device buffers: a GPU-aware MPI; otherwise stage through host mirrors as examples/handoff_mpi.cpp does)
    PARTHENON_MPI_CHECK(MPI_Alltoallv(st.refresh_send.data(), sc.data(), sd.data(), MPI_PARTHENON_REAL,
                                      st.refresh_recv.data(), rc.data(), rd.data(), MPI_PARTHENON_REAL,
                                      MPI_COMM_WORLD));
    JB_REQUIRE(jb_fill_cells(st.ctx, st.mesh, field, (std::int64_t)nd, 1, idx + 2 * ns, idx + 2 * ns + nd,
                             idx + 2 * ns + 2 * nd, idx + 2 * ns + 3 * nd, st.refresh_recv.data()));
  }
  return TaskStatus::complete;
}

// pool growth: the role of Swarm::AddEmptyParticles (sourcing.cpp:123-131)
static void Reserve(AmdState &st, std::int64_t nslots) {
  if (nslots <= st.sw.capacity) return;
  const std::int64_t cap = 2 * nslots;
  auto grow = [&](auto &view, const char *name) {
    using V = std::decay_t<decltype(view)>;
    V bigger(name, cap);
    if (st.sw.n > 0)
      Kokkos::deep_copy(Kokkos::subview(bigger, std::make_pair((std::int64_t)0, st.sw.n)),
                        Kokkos::subview(view, std::make_pair((std::int64_t)0, st.sw.n)));
    view = bigger;
  };
  if (st.pool_f64.empty()) { st.pool_f64.resize(9); st.pool_i32.resize(5); }
  for (auto &a : st.pool_f64) grow(a, "jb pool f64");
  for (auto &a : st.pool_i32) grow(a, "jb pool i32");
  grow(st.pool_id, "jb pool id");
  grow(st.pool_rng, "jb pool rng");
  double **f[9] = {&st.sw.x, &st.sw.y, &st.sw.z, &st.sw.vx, &st.sw.vy, &st.sw.vz, &st.sw.t, &st.sw.w, &st.sw.e};
  for (int q = 0; q < 9; ++q) *f[q] = st.pool_f64[q].data();
  int32_t **ii[5] = {&st.sw.ip, &st.sw.jp, &st.sw.kp, &st.sw.blk, &st.sw.status};
  for (int q = 0; q < 5; ++q) *ii[q] = st.pool_i32[q].data();
  st.sw.id = st.pool_id.data();
  st.sw.rng = st.pool_rng.data();
  st.sw.capacity = cap;
}

//----------------------------------------------------------------------------------------
TaskStatus UpdateDerivedTransportFields(MeshData<Real> *md, const Real dt) {   // jaybenne.cpp:285-492
  EnsureMeshView(md);
  AmdState &st = State(md->GetParentPointer());
  FlushInitialSource(md->GetParentPointer());   // (safeguard: normally done by the host after init)
  st.cycle += 1;   // first task of RadiationStep: once per cycle on every rank (keys SourceEpoch)
  return Status(jb_update_derived_transport_fields(st.ctx, st.mesh, dt));
}

// sourcing.cpp:25-208.  T = MeshData<Real> (cycle loop) or MeshBlockData<Real> (initialisation,
// one call per block: `nblocks` of sourcing.cpp:68-69 is then 1).
//
// Random-stream ids are global creation indices, so a source call needs every rank's per-block
// counts: one MPI_Allreduce.  The MeshData form runs once per rank and cycle: every rank makes the
// same collective calls.  The MeshBlockData form is called once per LOCAL block by the host's
// problem generator (mcblock.cpp:202), a number that differs between ranks (20 SMR blocks on 8
// ranks): it must not communicate.  It therefore only RECORDS the block; the source itself runs
// for all recorded blocks of this rank in one collective step, FlushInitialSource -- called by
// the host once after Parthenon has built the mesh and, as a
// safeguard, at the top of the first UpdateDerivedTransportFields.  The per-cell rounding streams
// are keyed by (cycle, source type) -- SourceEpoch of jaybenne_amd.hpp -- not by a per-call
// counter, so they do not depend on how many blocks a rank holds either.
static void SourceBlocks(Mesh *pm, AmdState &st, SourceType type, const Real t_start, const Real dt,
                         const std::vector<char> &selected, bool per_block) {
  auto *mesh_md = pm->mesh_data.Get().get();
  EnsureMeshView(mesh_md);
  const int nb = mesh_md->NumBlocks();
  const int nres = st.nowned + st.nhalo;   // (the library counts for every resident block: halo copies source nothing)
  std::vector<int32_t> nper(nres, 0), gid(nres, 0);
  const int src = type == SourceType::thermal ? JB_SOURCE_THERMAL : JB_SOURCE_EMISSION;
  JB_REQUIRE(jb_source_photons_count(st.ctx, st.mesh, src, dt, per_block ? 1 : nb,
                                     jaybenne_amd::SourceEpoch(st.cycle, type == SourceType::thermal
                                                                             ? jaybenne_amd::SourceType::thermal
                                                                             : jaybenne_amd::SourceType::emission),
                                     nper.data(), st.prefix.data()));
  std::vector<long long> counts(pm->nbtotal, 0), all(pm->nbtotal, 0);
  for (int b = 0; b < nb; ++b) {
    gid[b] = mesh_md->GetBlockData(b)->GetBlockPointer()->gid;
    if (!selected[b]) nper[b] = 0;
    counts[gid[b]] = nper[b];
  }
  for (int q = nb; q < nres; ++q) nper[q] = 0;   // (gid of a halo copy is not looked at: it sources nothing)
  MPI_Allreduce(counts.data(), all.data(), pm->nbtotal, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
  // (the index arithmetic is the tested one: include/jaybenne_amd.hpp, PlanSource)
  const jaybenne_amd::SourcePlan pl = jaybenne_amd::PlanSource(nper, gid, all, st.next_id, st.sw.n);
  Reserve(st, st.sw.n + pl.total_local);
  JB_REQUIRE(jb_source_photons_fill(st.ctx, st.mesh, &st.sw, src, t_start, dt, nper.data(),
                                    st.prefix.data(), pl.slot_base.data(), pl.id_base.data()));
  st.sw.n += pl.total_local;
  st.next_id = pl.next_id;
}

// One collective step for the blocks whose initial source was requested (every rank calls this
// exactly once, with or without requests of its own).
TaskStatus FlushInitialSource(Mesh *pm) {
  AmdState &st = State(pm);
  if (st.initial_source_done) return TaskStatus::complete;
  auto *mesh_md = pm->mesh_data.Get().get();
  std::vector<char> selected(mesh_md->NumBlocks(), 0);
  for (int lid : st.pending_initial_blocks) selected[lid] = 1;
  SourceBlocks(pm, st, SourceType::thermal, 0.0, 0.0, selected, /*per_block=*/true);
  st.pending_initial_blocks.clear();
  st.initial_source_done = true;
  return Status(jb_evaluate_radiation_energy(st.ctx, st.mesh, &st.sw));   // jaybenne.cpp:577
}

template <typename T, SourceType ST>
TaskStatus SourcePhotons(T *md, const Real t_start, const Real dt) {
  auto pm = md->GetParentPointer();
  AmdState &st = State(pm);
  auto &jbn = pm->packages.Get("jaybenne");
  if constexpr (ST == SourceType::emission) {
    if (!jbn->template Param<bool>("do_emission")) return TaskStatus::complete;   // sourcing.cpp:41-43
  }
  if constexpr (std::is_same_v<T, MeshBlockData<Real>>) {
    st.pending_initial_blocks.push_back(md->GetBlockPointer()->lid);   // no communication here
    return TaskStatus::complete;
  } else {
    std::vector<char> all_blocks(md->NumBlocks(), 1);
    SourceBlocks(pm, st, ST, t_start, dt, all_blocks, /*per_block=*/false);
    return TaskStatus::complete;
  }
}
template TaskStatus SourcePhotons<MeshBlockData<Real>, SourceType::thermal>(MeshBlockData<Real> *, const Real, const Real);
template TaskStatus SourcePhotons<MeshBlockData<Real>, SourceType::emission>(MeshBlockData<Real> *, const Real, const Real);
template TaskStatus SourcePhotons<MeshData<Real>, SourceType::thermal>(MeshData<Real> *, const Real, const Real);
template TaskStatus SourcePhotons<MeshData<Real>, SourceType::emission>(MeshData<Real> *, const Real, const Real);

// transport.cpp:28-181 / transport_ddmc.cpp:28-237.  `first` of the iterate sublist: the photons
// that arrived in the last MeshReceive (everything on the first pass).
static std::int64_t &FirstUntracked(AmdState &st) {
  static std::int64_t first = 0;
  return first;
}
TaskStatus TransportPhotons(MeshData<Real> *md, const Real t_start, const Real dt) {
  AmdState &st = State(md->GetParentPointer());
  return Status(jb_transport_photons(st.ctx, st.mesh, &st.sw, t_start, dt, FirstUntracked(st), st.sw.n, 0));
}
TaskStatus TransportPhotons_DDMC(MeshData<Real> *md, const Real t_start, const Real dt) {
  AmdState &st = State(md->GetParentPointer());
  return Status(jb_transport_photons_ddmc(st.ctx, st.mesh, &st.sw, t_start, dt, FirstUntracked(st), st.sw.n, 0));
}

// MeshResetCommunication / MeshSend / MeshReceive -- jaybenne.cpp:26-61 -- as one task: records
// per destination rank, counts by all-to-all, payload by all-to-all-v (device pointers: needs a
// GPU-aware MPI; otherwise stage through host buffers as examples/handoff_mpi.cpp does).
TaskStatus MeshSendReceive(MeshData<Real> *md) {
  AmdState &st = State(md->GetParentPointer());
  const int nranks = Globals::nranks;
  std::vector<std::int64_t> send(nranks), recv(nranks);
  const std::int64_t first = FirstUntracked(st), last = st.sw.n;
  if ((std::int64_t)st.records.size() < (last - first + 1) * JB_RECORD_WORDS)
    st.records = ParArray1D<std::int64_t>("jb records", (last - first + 1) * JB_RECORD_WORDS);
  JB_REQUIRE(jb_pack_outgoing(st.ctx, st.mesh, &st.sw, first, last, nranks, st.records.data(),
                              st.records.size() / JB_RECORD_WORDS, send.data()));
  MPI_Alltoall(send.data(), 1, MPI_INT64_T, recv.data(), 1, MPI_INT64_T, MPI_COMM_WORLD);
  std::vector<int> sc(nranks), sd(nranks), rc(nranks), rd(nranks);
  std::int64_t ns = 0, nr = 0;
  for (int r = 0; r < nranks; ++r) {
    sd[r] = ns * JB_RECORD_WORDS; sc[r] = send[r] * JB_RECORD_WORDS; ns += send[r];
    rd[r] = nr * JB_RECORD_WORDS; rc[r] = recv[r] * JB_RECORD_WORDS; nr += recv[r];
  }
  ParArray1D<std::int64_t> inbox("jb inbox", std::max<std::int64_t>(nr, 1) * JB_RECORD_WORDS);
  MPI_Alltoallv(st.records.data(), sc.data(), sd.data(), MPI_INT64_T, inbox.data(), rc.data(),
                rd.data(), MPI_INT64_T, MPI_COMM_WORLD);
  if (st.sw.n + nr > st.sw.capacity) JB_REQUIRE(jb_remove_marked_particles(st.ctx, &st.sw));
  Reserve(st, st.sw.n + nr);
  FirstUntracked(st) = st.sw.n;                      // the next transport pass covers the arrivals
  JB_REQUIRE(jb_unpack_incoming(st.ctx, st.mesh, &st.sw, inbox.data(), nr));
  return TaskStatus::complete;
}

TaskStatus SampleDDMCBlockFace(MeshData<Real> *md) {   // sample_ddmc_bface.cpp:81-427
  AmdState &st = State(md->GetParentPointer());
  return Status(jb_sample_ddmc_block_face(st.ctx, st.mesh, &st.sw, FirstUntracked(st), st.sw.n));
}

TaskStatus CheckCompletion(MeshData<Real> *md, const Real t_end) {   // transport.cpp:187-216
  AmdState &st = State(md->GetParentPointer());
  std::int64_t unfinished = 0;
  const jb_status s = jb_check_completion(st.ctx, &st.sw, t_end, &unfinished);
  if (s == JB_COMPLETE) FirstUntracked(st) = 0;       // the sublist is done: next cycle starts over
  return Status(s);
}

template <typename T>
TaskStatus EvaluateRadiationEnergy(T *md) {   // jaybenne.cpp:514-564
  auto pm = md->GetParentPointer();
  EnsureMeshView(pm->mesh_data.Get().get());
  AmdState &st = State(pm);
  return Status(jb_evaluate_radiation_energy(st.ctx, st.mesh, &st.sw));
}
template TaskStatus EvaluateRadiationEnergy<MeshBlockData<Real>>(MeshBlockData<Real> *);
template TaskStatus EvaluateRadiationEnergy<MeshData<Real>>(MeshData<Real> *);

TaskStatus UpdateFluid(MeshData<Real> *md) {   // jaybenne.cpp:583-615
  AmdState &st = State(md->GetParentPointer());
  JB_REQUIRE(jb_remove_marked_particles(st.ctx, &st.sw));   // (transport.cpp:176-178, once per cycle)
  return Status(jb_update_fluid(st.ctx, st.mesh));
}

// jaybenne.cpp:499-509 (Swarm::Defrag; no task list of the reference schedules it).  The resident
// swarm is one array set for all blocks of the rank and compact after UpdateFluid; the task sorts
// it by (block, cell) -- the locality the tracking kernels' cell gathers live on (jaybenne_amd.h) --
// once per call series: the first block of the rank does the work for all of them.
TaskStatus DefragParticles(MeshBlock *pmb) {
  AmdState &st = State(pmb->pmy_mesh);
  if (pmb->lid != 0) return TaskStatus::complete;
  return Status(jb_defrag_particles(st.ctx, st.mesh, &st.sw));
}

Real EstimateTimestepMesh(MeshData<Real> *md) {   // jaybenne.cpp:271-275
  return jb_estimate_timestep(State(md->GetParentPointer()).ctx);
}

void InitializeRadiation(MeshBlockData<Real> *mbd, const bool is_thermal) {   // jaybenne.cpp:570-578
  // (records the block; the source and the initial tally run in FlushInitialSource, see above)
  if (is_thermal) SourcePhotons<MeshBlockData<Real>, SourceType::thermal>(mbd, 0.0, 0.0);
}

//----------------------------------------------------------------------------------------
// jaybenne::RadiationStep -- jaybenne.cpp:68-151: the same regions, task list and iterate sublist
// with the same completion semantics; MeshResetCommunication / MeshSend / MeshReceive are one task.
TaskCollection RadiationStep(Mesh *pmesh, const Real t_start, const Real dt) {
  auto &jb_pkg = pmesh->packages.Get("jaybenne");
  const auto &max_transport_iterations = jb_pkg->Param<int>("max_transport_iterations");
  const bool &use_ddmc = jb_pkg->Param<bool>("use_ddmc");
  TaskCollection tc;
  TaskID none(0);
  const int num_partitions = pmesh->DefaultNumPartitions();
  PARTHENON_REQUIRE(num_partitions == 1,
                    "Iterative tasking may not support multiple partitions per rank as of 2024/5/14")
  auto &reg = tc.AddRegion(num_partitions);
  for (int i = 0; i < num_partitions; i++) {
    auto &tl = reg[i];
    auto &base = pmesh->mesh_data.GetOrAdd("base", i);
    auto derived = tl.AddTask(none, UpdateDerivedTransportFields, base.get(), dt);
    auto source = tl.AddTask(derived, SourcePhotons<MeshData<Real>, SourceType::emission>,
                             base.get(), t_start, dt);
    // (no ghost exchange of ddmc_face_prob: no kernel reads a ghost face)
    auto [itl, push] = tl.AddSublist(source, {1, max_transport_iterations});
    auto transport = use_ddmc ? itl.AddTask(none, TransportPhotons_DDMC, base.get(), t_start, dt)
                              : itl.AddTask(none, TransportPhotons, base.get(), t_start, dt);
    auto exchange = itl.AddTask(transport, MeshSendReceive, base.get());
    auto sample_ddmc_bface = use_ddmc ? itl.AddTask(exchange, SampleDDMCBlockFace, base.get()) : exchange;
    auto complete = itl.AddTask(TQ::once_per_region | TQ::global_sync | TQ::completion,
                                sample_ddmc_bface, CheckCompletion, base.get(), t_start + dt);
    auto eval_rad = tl.AddTask(push, EvaluateRadiationEnergy<MeshData<Real>>, base.get());
    auto update_fluid = tl.AddTask(eval_rad, UpdateFluid, base.get());
  }
  return tc;
}

//----------------------------------------------------------------------------------------
// Output view: copy the pool into the registered swarm, block by block (positions, the five
// variables of jaybenne.cpp:236-245); called before an output that lists `swarms = photons`.
TaskStatus ExportToParthenonSwarm(MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  AmdState &st = State(pm);
  const int nb = md->NumBlocks();
  // particles per block (host histogram of the blk attribute), then one gather kernel per block
  auto blk_h = Kokkos::create_mirror_view_and_copy(Kokkos::HostSpace(),
                                                   Kokkos::subview(st.pool_i32[3], std::make_pair((std::int64_t)0, st.sw.n)));
  std::vector<std::vector<std::int64_t>> members(nb);
  for (std::int64_t n = 0; n < st.sw.n; ++n) members[blk_h(n)].push_back(n);
  for (int b = 0; b < nb; ++b) {
    auto swarm = md->GetSwarmData(b)->Get(photons_swarm_name);
    swarm->RemoveMarkedParticles();
    auto ctx_new = swarm->AddEmptyParticles(members[b].size());
    ParArray1D<std::int64_t> idx("jb members", std::max<size_t>(members[b].size(), 1));
    auto idx_h = Kokkos::create_mirror_view(idx);
    for (size_t q = 0; q < members[b].size(); ++q) idx_h(q) = members[b][q];
    Kokkos::deep_copy(idx, idx_h);
    auto &x = swarm->Get<Real>(swarm_position::x::name()).Get();
    auto &y = swarm->Get<Real>(swarm_position::y::name()).Get();
    auto &z = swarm->Get<Real>(swarm_position::z::name()).Get();
    auto &w = swarm->Get<Real>(ph::weight::name()).Get();
    auto &e = swarm->Get<Real>(ph::energy::name()).Get();
    auto &t = swarm->Get<Real>(ph::time::name()).Get();
    auto &vel = swarm->Get<Real>(ph::v::name()).Get();
    const auto sw = st.sw;
    parthenon::par_for(
        DEFAULT_LOOP_PATTERN, "jaybenne_amd::ExportToParthenonSwarm", DevExecSpace(), 0,
        (int)members[b].size() - 1, KOKKOS_LAMBDA(const int q) {
          const int n = ctx_new.GetNewParticleIndex(q);
          const std::int64_t s = idx(q);
          x(n) = sw.x[s]; y(n) = sw.y[s]; z(n) = sw.z[s];
          w(n) = sw.w[s]; e(n) = sw.e[s]; t(n) = sw.t[s];
          vel(0, n) = sw.vx[s]; vel(1, n) = sw.vy[s]; vel(2, n) = sw.vz[s];
        });
  }
  return TaskStatus::complete;
}

}  // namespace jaybenne
