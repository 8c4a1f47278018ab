// Host-side logic of the Parthenon adapter (adapters/parthenon/jaybenne_amd_tasks.cpp) on a CPU: the adapter
// is COMPILED against the declared-interface headers of tests/parthenon_iface/ (not Parthenon: see the header
// there) and LINKED against the recording stand-in for the C ABI below, then driven as rank `rank` of `nranks`
// on a mesh read from stdin (written by tests/test_cabi.py from jaybenne_amd.mesh.Mesh).  It prints what the
// adapter hands to jb_mesh_create, jb_source_photons_count / _fill and jb_gather_cells / jb_fill_cells; the
// Python test holds that to what the Python host builds for the same rank.  A syntax / host-logic check of
// the adapter -- never parity evidence.
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "jaybenne_amd_tasks.hpp"

namespace parthenon { namespace Globals { int my_rank = 0, nranks = 1, nghost = 2; } }
extern "C" int hipGetDevice(int *d) { *d = 0; return 0; }

// ---- the recording stand-in for libjaybenne_amd.so ---------------------------------------------------
struct jb_context { int dummy; };
struct jb_mesh { int dummy; };
namespace rec {
jb_mesh_view view;
std::vector<int32_t> leaf_map, owner, local_index, gid, owned, level, nbr_lev;
std::vector<double> xmin, xmax, dx;
std::vector<const double *> rho;
int creates = 0, derived = 0, counts = 0, fills = 0, tallies = 0;
std::vector<int32_t> last_nper;
std::vector<int64_t> last_slot;
std::vector<uint64_t> last_id;
long long gathered = 0, filled = 0;
int blocks_in_call = -1;
uint32_t epoch = 99;
}  // namespace rec
extern "C" {
const char *jb_last_error(void) { return "recording stand-in"; }
jb_status jb_initialize(const jb_params *, const jb_eos *, const jb_opacity *, const jb_scattering *, int, jb_context **c) {
  *c = new jb_context{0};
  return JB_COMPLETE;
}
jb_status jb_finalize(jb_context *c) { delete c; return JB_COMPLETE; }
int32_t jb_param_seed(const jb_context *) { return 123; }
jb_status jb_synchronize(jb_context *) { return JB_COMPLETE; }
jb_status jb_mesh_create(jb_context *, const jb_mesh_view *v, jb_mesh **m) {
  using namespace rec;
  view = *v;
  const size_t nleaf = (size_t)v->nleaf[0] * v->nleaf[1] * v->nleaf[2];
  leaf_map.assign(v->leaf_map, v->leaf_map + nleaf);
  owner.assign(v->owner, v->owner + v->nblocks_total);
  local_index.assign(v->local_index, v->local_index + v->nblocks_total);
  gid.assign(v->gid, v->gid + v->nblocks);
  owned.assign(v->owned, v->owned + v->nblocks);
  level.assign(v->blk_level, v->blk_level + v->nblocks);
  nbr_lev.assign(v->blk_nbr_lev, v->blk_nbr_lev + 6 * (size_t)v->nblocks);
  xmin.assign(v->blk_xmin, v->blk_xmin + 3 * (size_t)v->nblocks);
  xmax.assign(v->blk_xmax, v->blk_xmax + 3 * (size_t)v->nblocks);
  dx.assign(v->blk_dx, v->blk_dx + 3 * (size_t)v->nblocks);
  rho.assign(v->rho, v->rho + v->nblocks);
  for (int b = 0; b < v->nblocks; ++b)
    if (!v->rho[b] || !v->sie[b] || !v->u[b] || !v->fleck[b] || !v->tally[b] || !v->edelta[b] || !v->src_ew[b] ||
        !v->src_num[b] || !v->P1[b] || !v->P2[b] || !v->P3[b]) return JB_ERR_INVALID;
  ++creates;
  *m = new jb_mesh{0};
  return JB_COMPLETE;
}
jb_status jb_mesh_destroy(jb_mesh *m) { delete m; return JB_COMPLETE; }
jb_status jb_update_derived_transport_fields(jb_context *, jb_mesh *, double) { ++rec::derived; return JB_COMPLETE; }
jb_status jb_source_photons_count(jb_context *, jb_mesh *, int, double, int blocks_in_call, uint32_t epoch,
                                  int32_t *nper, int32_t *) {
  ++rec::counts;
  rec::blocks_in_call = blocks_in_call;
  rec::epoch = epoch;
  for (int b = 0; b < rec::view.nblocks; ++b) nper[b] = rec::owned[(size_t)b] ? 100 + rec::gid[(size_t)b] : 0;
  return JB_COMPLETE;
}
jb_status jb_source_photons_fill(jb_context *, jb_mesh *, const jb_swarm_view *, int, double, double,
                                 const int32_t *nper, const int32_t *, const int64_t *slot, const uint64_t *id) {
  ++rec::fills;
  rec::last_nper.assign(nper, nper + rec::view.nblocks);
  rec::last_slot.assign(slot, slot + rec::view.nblocks);
  rec::last_id.assign(id, id + rec::view.nblocks);
  return JB_COMPLETE;
}
jb_status jb_evaluate_radiation_energy(jb_context *, jb_mesh *, const jb_swarm_view *) { ++rec::tallies; return JB_COMPLETE; }
jb_status jb_gather_cells(jb_context *, jb_mesh *, int, int64_t n, const int32_t *, const int32_t *, double *) { rec::gathered += n; return JB_COMPLETE; }
jb_status jb_fill_cells(jb_context *, jb_mesh *, int, int64_t n, int, const int32_t *, const int32_t *, const int32_t *,
                        const int32_t *, const double *) { rec::filled += n; return JB_COMPLETE; }
// (referenced by tasks this test does not run)
jb_status jb_transport_photons(jb_context *, jb_mesh *, const jb_swarm_view *, double, double, int64_t, int64_t, int) { return JB_COMPLETE; }
jb_status jb_transport_photons_ddmc(jb_context *, jb_mesh *, const jb_swarm_view *, double, double, int64_t, int64_t, int) { return JB_COMPLETE; }
jb_status jb_pack_outgoing(jb_context *, jb_mesh *, const jb_swarm_view *, int64_t, int64_t, int nranks, int64_t *, int64_t, int64_t *c) {
  for (int r = 0; r < nranks; ++r) c[r] = 0;
  return JB_COMPLETE;
}
jb_status jb_unpack_incoming(jb_context *, jb_mesh *, jb_swarm_view *, const int64_t *, int64_t) { return JB_COMPLETE; }
jb_status jb_remove_marked_particles(jb_context *, jb_swarm_view *) { return JB_COMPLETE; }
jb_status jb_sample_ddmc_block_face(jb_context *, jb_mesh *, const jb_swarm_view *, int64_t, int64_t) { return JB_COMPLETE; }
jb_status jb_check_completion(jb_context *, const jb_swarm_view *, double, int64_t *u) { *u = 0; return JB_COMPLETE; }
jb_status jb_update_fluid(jb_context *, jb_mesh *) { return JB_COMPLETE; }
jb_status jb_defrag_particles(jb_context *, jb_mesh *, const jb_swarm_view *) { return JB_COMPLETE; }
double jb_estimate_timestep(const jb_context *) { return 1.0; }
}

using namespace parthenon;

int main() {
  int ndim, nb_total, nranks, rank, nx[3], ng, nroot[3], maxlev, per[6];
  double gmin[3], gmax[3];
  if (std::scanf("%d %d %d %d %d %d", &ndim, &nb_total, &nranks, &rank, &ng, &maxlev) != 6) return 2;
  for (int d = 0; d < 3; ++d)
    if (std::scanf("%lf %lf %d %d", &gmin[d], &gmax[d], &nroot[d], &nx[d]) != 4) return 2;
  for (int f = 0; f < 6; ++f) if (std::scanf("%d", &per[f]) != 1) return 2;
  Globals::my_rank = rank; Globals::nranks = nranks; Globals::nghost = ng;
  Mesh mesh;
  mesh.nbtotal = nb_total; mesh.ndim = ndim; mesh.max_level = maxlev;
  for (int d = 0; d < 3; ++d) { mesh.mesh_size.lo[d] = gmin[d]; mesh.mesh_size.hi[d] = gmax[d]; mesh.nrbx[d] = nroot[d]; }
  for (int f = 0; f < 6; ++f) {
    mesh.mesh_bcs[f] = per[f] ? BoundaryFlag::periodic : BoundaryFlag::outflow;
    mesh.mesh_swarm_bc_names[f] = per[f] ? "periodic" : "jaybenne_reflecting";
  }
  mesh.locs.resize((size_t)nb_total); mesh.ranks.resize((size_t)nb_total);
  auto md = std::make_shared<MeshData<Real>>();
  md->pmesh = &mesh;
  md->ib = {ng, ng + nx[0] - 1};
  md->jb = ndim > 1 ? IndexRange{ng, ng + nx[1] - 1} : IndexRange{0, 0};
  md->kb = ndim > 2 ? IndexRange{ng, ng + nx[2] - 1} : IndexRange{0, 0};
  const size_t ntot = (size_t)(nx[0] + 2 * ng) * (ndim > 1 ? nx[1] + 2 * ng : 1) * (ndim > 2 ? nx[2] + 2 * ng : 1);
  const char *names[] = {"field.material.density", "field.material.sie", "field.material.internal_energy",
                         "field.jaybenne.fleck_factor", "field.jaybenne.energy_tally", "field.jaybenne.energy_delta",
                         "field.jaybenne.source_ew_per_cell", "field.jaybenne.source_num_per_cell",
                         "field.jaybenne.ddmc_face_prob/F1", "field.jaybenne.ddmc_face_prob/F2",
                         "field.jaybenne.ddmc_face_prob/F3"};
  int lid = 0;
  for (int g = 0; g < nb_total; ++g) {
    int own, lev, nl[6], pb[6];
    long long l[3];
    double lo[3], hi[3];
    if (std::scanf("%d %d %lld %lld %lld", &own, &lev, &l[0], &l[1], &l[2]) != 5) return 2;
    for (int d = 0; d < 3; ++d) if (std::scanf("%lf %lf", &lo[d], &hi[d]) != 2) return 2;
    for (int f = 0; f < 6; ++f) if (std::scanf("%d %d", &nl[f], &pb[f]) != 2) return 2;
    mesh.locs[(size_t)g].lev = lev;
    for (int d = 0; d < 3; ++d) mesh.locs[(size_t)g].l[d] = l[d];
    mesh.ranks[(size_t)g] = own;
    if (own != rank) continue;
    auto pmb = std::make_shared<MeshBlock>();
    pmb->gid = g; pmb->lid = lid++; pmb->loc = mesh.locs[(size_t)g]; pmb->pmy_mesh = &mesh;
    for (int d = 0; d < 3; ++d) {
      pmb->block_size.lo[d] = lo[d]; pmb->block_size.hi[d] = hi[d];
      pmb->coords.dx[d] = (hi[d] - lo[d]) / (double)nx[d];
    }
    for (int f = 0; f < 6; ++f) { pmb->nbr_level[f] = nl[f]; pmb->phys_bdry[f] = pb[f] != 0; }
    for (const char *n : names) pmb->vars[n].assign(ntot, 0.0);
    auto mbd = std::make_shared<MeshBlockData<Real>>();
    mbd->pmb = pmb;
    md->blocks.push_back(mbd);
  }
  mesh.mesh_data.Get() = md;
  mesh.mesh_data.GetOrAdd("base", 0) = md;

  ParameterInput pin;
  pin.kv["jaybenne/num_particles"] = "100000";
  pin.kv["jaybenne/use_ddmc"] = "true";
  EOS eos;
  Opacity opac;
  Scattering scat;
  scat.kappa_s = 1.0e3;
  auto pkg = jaybenne::Initialize(&pin, opac, scat, eos);
  mesh.packages.Add(pkg);
  if (pkg->fields.size() != 6 || pkg->swarms.size() != 1 || pkg->swarm_values.size() != 5) return 3;
  if (pkg->Param<int>("num_particles") != 100000 || !pkg->Param<bool>("use_ddmc") || pkg->Param<Real>("tau_ddmc") != 5.0) return 3;

  // the problem generator's per-block hook, then the one collective flush, then the first task of a cycle
  for (int b = 0; b < md->NumBlocks(); ++b) jaybenne::InitializeRadiation(md->GetBlockData(b).get(), true);
  if (rec::counts != 0) return 4;                          // the per-block hook must not source (no communication there)
  if (jaybenne::FlushInitialSource(&mesh) != TaskStatus::complete) return 4;
  if (rec::creates != 1 || rec::counts != 1 || rec::fills != 1 || rec::tallies != 1) return 4;
  if (rec::blocks_in_call != 1 || rec::epoch != 0u) return 5;   // MeshBlockData path: nblocks = 1, epoch 0
  std::printf("initial nper"); for (auto v : rec::last_nper) std::printf(" %d", v); std::printf("\n");
  std::printf("initial slot"); for (auto v : rec::last_slot) std::printf(" %lld", (long long)v); std::printf("\n");
  std::printf("initial id"); for (auto v : rec::last_id) std::printf(" %llu", (unsigned long long)v); std::printf("\n");
  if (jaybenne::UpdateDerivedTransportFields(md.get(), 1.0e-11) != TaskStatus::complete) return 6;
  if (rec::creates != 1 || rec::derived != 1) return 6;   // the mesh view is built once per block list
  if (jaybenne::SourcePhotons<MeshData<Real>, jaybenne::SourceType::emission>(md.get(), 0.0, 1.0e-11) != TaskStatus::complete) return 7;
  if (rec::blocks_in_call != md->NumBlocks() || rec::epoch != 1u) return 7;   // MeshData path: this rank's blocks, cycle 1

  const jb_mesh_view &v = rec::view;
  std::printf("view %d %d %d %d %d | %d %d %d | %d %d %d | %d %d %d %d %d %d\n", v.ndim, v.ng, v.nblocks, v.nblocks_total,
              v.rank, v.nx[0], v.nx[1], v.nx[2], v.nleaf[0], v.nleaf[1], v.nleaf[2], v.bc[0], v.bc[1], v.bc[2], v.bc[3],
              v.bc[4], v.bc[5]);
  std::printf("leaf_map"); for (auto q : rec::leaf_map) std::printf(" %d", q); std::printf("\n");
  std::printf("owner"); for (auto q : rec::owner) std::printf(" %d", q); std::printf("\n");
  std::printf("local_index"); for (auto q : rec::local_index) std::printf(" %d", q); std::printf("\n");
  std::printf("gid"); for (auto q : rec::gid) std::printf(" %d", q); std::printf("\n");
  std::printf("owned"); for (auto q : rec::owned) std::printf(" %d", q); std::printf("\n");
  std::printf("level"); for (auto q : rec::level) std::printf(" %d", q); std::printf("\n");
  std::printf("nbr_lev"); for (auto q : rec::nbr_lev) std::printf(" %d", q); std::printf("\n");
  std::printf("xmin"); for (auto q : rec::xmin) std::printf(" %.17g", q); std::printf("\n");
  std::printf("xmax"); for (auto q : rec::xmax) std::printf(" %.17g", q); std::printf("\n");
  std::printf("dx"); for (auto q : rec::dx) std::printf(" %.17g", q); std::printf("\n");
  // the block's own arrays for owned blocks (Parthenon's), the adapter's for halo copies: all distinct
  for (size_t a = 0; a < rec::rho.size(); ++a)
    for (size_t b = a + 1; b < rec::rho.size(); ++b) if (rec::rho[a] == rec::rho[b]) return 8;
  for (int b = 0; b < md->NumBlocks(); ++b)
    if (rec::rho[(size_t)b] != md->GetBlockData(b)->GetBlockPointer()->vars["field.material.density"].data()) return 8;
  std::printf("refresh gathered %lld filled %lld\n", rec::gathered, rec::filled);
  // the task graph of a cycle (jaybenne.cpp:68-151): one region, derived -> source -> iterate{transport,
  // exchange, block faces (DDMC), completion} -> radiation energy -> fluid update
  TaskCollection tc = jaybenne::RadiationStep(&mesh, 0.0, 1.0e-11);
  if (tc.regions.size() != 1 || tc.regions[0].size() != 1) return 9;
  TaskList &tl = tc.regions[0][0];
  if (tl.tasks.size() != 4 || tl.sublists.size() != 1 || tl.sublists[0]->tasks.size() != 4) return 9;
  const auto &it = tl.sublists[0]->tasks;
  const unsigned want = (unsigned)TQ::once_per_region | (unsigned)TQ::global_sync | (unsigned)TQ::completion;
  if (it[3].qual != want || tl.sublist_iters[0].first != 1 || tl.sublist_iters[0].second != 10000) return 9;
  std::printf("tasks %zu + sublist %zu (max %d iterations)\n", tl.tasks.size(), it.size(), tl.sublist_iters[0].second);
  std::printf("adapter ok\n");
  return 0;
}
