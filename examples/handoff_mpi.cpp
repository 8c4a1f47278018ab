// handoff_mpi.cpp -- the inter-rank particle hand-off of the history loop driven from C++ over
// MPI, with nothing but the C ABI (include/jaybenne_amd.h), the HIP runtime and MPI in the process:
// the iterate-sublist of reference jaybenne.cpp:113-131 (transport -> MeshSend / MeshReceive ->
// CheckCompletion with its global_sync) as a host would write it around
// jb_transport_photons / jb_pack_outgoing / jb_unpack_incoming.
//
//   mpiexec -n R ./handoff_mpi [cells_per_block] [blocks] [particles] [cycles] [halo_rings] [dump_prefix] [exchange]
//
// exchange = "tasks" (default): the three tasks as a host strings them together itself (jb_pack_outgoing,
//   its own MPI calls, jb_unpack_incoming);
// "mpi" / "rccl": ONE call per transport iteration, jb_exchange (count on the device -> all-gather of
//   the rank x rank count matrix, read back once -> pack -> all-to-all-v -> unpack, all on the
//   context's stream), over a jb_exchange_transport: "rccl" = jb_transport_rccl on a communicator this
//   program bootstraps over MPI (ncclGetUniqueId / ncclCommInitRank: one rank per GPU -- the production
//   path; the records never leave the device), "mpi" = the same two collectives written with MPI and
//   host staging below (runs with all ranks on one GPU, which RCCL refuses).
//
// Problem: inputs/stepdiff.in in 1-D (x in [-0.5, 0.5], sigma_s = 1e3, no absorption, T = 1e5 K
// for x < 0 and 1 K for x >= 0, reflecting walls), `blocks` meshblocks dealt to the R ranks in
// contiguous runs.  With halo_rings = 1 (the default) a rank also keeps read-only HALO COPIES of the
// other ranks' blocks that touch its own (jaybenne_amd::PlanHalo; their material state arrives
// through the refresh plan PlanHaloRefresh -> jb_gather_cells -> MPI_Alltoallv -> jb_fill_cells): a
// photon that wanders across the rank boundary is tracked on, and handed to the owner of the block
// it ends in once -- two transport iterations per cycle.  With halo_rings = 0 a rank keeps ONLY the
// blocks it owns and every photon that leaves them comes back from the transport kernel at once
// (one iteration per rank-boundary crossing of the most persistent photon: ~75 per cycle).  Either
// way an OUTGOING photon carries the global id of its destination block, is packed into 104-byte
// records per destination rank, travels through MPI_Alltoallv (staged through host memory: the MPI
// of this image is not GPU-aware; with a GPU-aware MPI or RCCL the device buffers go in directly)
// and is appended to the receiver's swarm.  The loop ends when no rank moved a particle.
// dump_prefix: every rank writes its photons (id, x, vx, t, w, stream state; global block, cell) to
// <dump_prefix>.<rank>.bin at the end -- what tests/test_gpu_multirank.py compares with the oracle.
//
// Checks (exit code 0 iff all hold): the photon count and the total weight are conserved over the
// cycles (sigma_a = 0), every photon ends each cycle exactly at census, and the energy tally
// integrates to the radiation energy.  All ranks may share one GPU (rank % device_count).
#include <hip/hip_runtime.h>
#include <mpi.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jaybenne_amd.hpp"

#define HIP_OK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                 \
      MPI_Abort(MPI_COMM_WORLD, 3);                                                          \
    }                                                                                        \
  } while (0)
#define JB_OK(call)                                                                          \
  do {                                                                                       \
    jb_status s_ = (call);                                                                   \
    if (s_ < 0) {                                                                            \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, (int)s_, jb_last_error());         \
      MPI_Abort(MPI_COMM_WORLD, 4);                                                          \
    }                                                                                        \
  } while (0)

// ---- a jb_exchange_transport over MPI with host staging (what "mpi" runs; "rccl" uses the library's)
struct MpiTransport { int nranks; };
static int mpi_all_gather_u64(void *h, const uint64_t *in_dev, uint64_t *out_dev, int count, void *stream) {
  const int nranks = ((MpiTransport *)h)->nranks;
  std::vector<uint64_t> in((size_t)count), out((size_t)count * nranks);
  if (hipMemcpyAsync(in.data(), in_dev, sizeof(uint64_t) * count, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
  MPI_Allgather(in.data(), count, MPI_UINT64_T, out.data(), count, MPI_UINT64_T, MPI_COMM_WORLD);
  return hipMemcpy(out_dev, out.data(), sizeof(uint64_t) * out.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
static int mpi_all_to_all_v(void *h, const int64_t *send_dev, const int64_t *sc, const int64_t *so, int64_t *recv_dev,
                            const int64_t *rc, const int64_t *ro, int words, void *stream) {
  const int nranks = ((MpiTransport *)h)->nranks;
  std::vector<int> c1(nranks), d1(nranks), c2(nranks), d2(nranks);
  long long ns = 0, nr = 0;
  for (int r = 0; r < nranks; ++r) {
    c1[r] = (int)(sc[r] * words); d1[r] = (int)(so[r] * words); ns += sc[r];
    c2[r] = (int)(rc[r] * words); d2[r] = (int)(ro[r] * words); nr += rc[r];
  }
  std::vector<int64_t> sbuf((size_t)ns * words + 1), rbuf((size_t)nr * words + 1);
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;   // the pack kernel has written send_dev
  if (ns && hipMemcpy(sbuf.data(), send_dev, (size_t)ns * words * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  MPI_Alltoallv(sbuf.data(), c1.data(), d1.data(), MPI_INT64_T, rbuf.data(), c2.data(), d2.data(), MPI_INT64_T,
                MPI_COMM_WORLD);
  if (nr && hipMemcpy(recv_dev, rbuf.data(), (size_t)nr * words * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

template <class T>
static T *dev_alloc(size_t n) {
  T *p = nullptr;
  HIP_OK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
  HIP_OK(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
  return p;
}

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int rank = 0, nranks = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &nranks);
  const int nx = argc > 1 ? std::atoi(argv[1]) : 16;
  const int nblocks_total = argc > 2 ? std::atoi(argv[2]) : 8;
  const long long nparticles = argc > 3 ? std::atoll(argv[3]) : 200000;
  const int cycles = argc > 4 ? std::atoi(argv[4]) : 3;
  const int halo_rings = argc > 5 ? std::atoi(argv[5]) : 1;
  const char *dump_prefix = argc > 6 && std::strcmp(argv[6], "-") != 0 ? argv[6] : nullptr;
  const char *exchange = argc > 7 ? argv[7] : "tasks";
  if (std::strcmp(exchange, "tasks") != 0 && std::strcmp(exchange, "mpi") != 0 && std::strcmp(exchange, "rccl") != 0) {
    if (rank == 0) std::fprintf(stderr, "exchange must be tasks, mpi or rccl\n");
    MPI_Finalize();
    return 2;
  }
  if (nblocks_total < nranks) {
    if (rank == 0) std::fprintf(stderr, "need at least one block per rank\n");
    MPI_Finalize();
    return 2;
  }
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  const int device = rank % ndev;
  HIP_OK(hipSetDevice(device));

  // ---- package (jaybenne::Initialize; deck values of inputs/stepdiff.in)
  const double c = 2.99792458e10, sb = 5.670373e-5;  // (CGS, CODATA 2010: jaybenne_amd/constants.py, examples/mcblock_amd.cpp)
  jb_params p{};
  p.num_particles = nparticles;
  p.dt = 3.335641e-11;
  p.min_swarm_occupancy = 0.0;
  p.numin = 0.0; p.numax = 1.0e300;
  p.tau_ddmc = 5.0;
  p.unique_rank_seeds = 1;
  p.seed = 349857;
  p.max_transport_iterations = 10000;
  p.use_ddmc = 0;
  p.source_strategy = JB_STRATEGY_UNIFORM;
  p.do_emission = 0;
  p.do_feedback = 0;
  p.rank = rank;
  jb_eos eos{JB_EOS_IDEAL_GAS, 0, 1.66666666667 - 1.0, 1.0 / (1.66666666667 - 1.0)};
  jb_opacity opac{JB_OPAC_GRAY, 0, 0.0, c, sb};
  jb_scattering scat{JB_SCAT_GRAY, 0, 1.0e3, 1.0};
  jb_context *ctx = nullptr;
  JB_OK(jb_initialize(&p, &eos, &opac, &scat, device, &ctx));

  // ---- mesh: blocks [b0, b1) belong to this rank; resident = those + the halo copies
  const int ng = 2;
  const int ni = nx + 2 * ng;
  const double blen = 1.0 / nblocks_total, dx = blen / nx;
  std::vector<int32_t> leaf_map(nblocks_total), owner(nblocks_total);
  std::vector<double> gxmin(3 * nblocks_total), gxmax(3 * nblocks_total);
  for (int g = 0; g < nblocks_total; ++g) {
    leaf_map[g] = g;
    int r = 0;
    while ((int)((long long)nblocks_total * (r + 1) / nranks) <= g) ++r;
    owner[g] = r;
    gxmin[3 * g] = -0.5 + g * blen; gxmax[3 * g] = -0.5 + (g + 1) * blen;
    for (int d = 1; d < 3; ++d) { gxmin[3 * g + d] = -0.5; gxmax[3 * g + d] = 0.5; }
  }
  jaybenne_amd::MeshTopology topo;
  topo.ndim = 1;
  for (int d = 0; d < 3; ++d) { topo.gmin[d] = -0.5; topo.gmax[d] = 0.5; }
  topo.nleaf[0] = nblocks_total;
  topo.leaf_map = leaf_map.data(); topo.nblocks_total = nblocks_total;
  topo.blk_xmin = gxmin.data(); topo.blk_xmax = gxmax.data(); topo.owner = owner.data();
  const jaybenne_amd::HaloPlan halo = jaybenne_amd::PlanHalo(topo, rank, halo_rings);
  const int nb = (int)halo.resident_gids.size(), nowned = halo.nowned;
  std::vector<int32_t> local_index = halo.local_index, gid = halo.resident_gids, level(nb, 0), nbr_lev(6 * nb, 0);
  std::vector<double> xmin(3 * nb), xmax(3 * nb), dxs(3 * nb);
  for (int l = 0; l < nb; ++l) {
    const int g = gid[l];
    for (int d = 0; d < 3; ++d) { xmin[3 * l + d] = gxmin[3 * g + d]; xmax[3 * l + d] = gxmax[3 * g + d]; }
    dxs[3 * l] = dx; dxs[3 * l + 1] = dxs[3 * l + 2] = 1.0;
  }
  // fields: one [ni] array per block and field, filled with the initial condition of mcblock.cpp:187-199
  const char *fields[] = {"rho", "sie", "u", "fleck", "tally", "edelta", "src_ew", "src_num"};
  std::vector<std::vector<double *>> fptr(8, std::vector<double *>(nb));
  const double cv = eos.cv, t_hot = 1.0e5, t_cold = 1.0;
  for (int l = 0; l < nb; ++l) {
    std::vector<double> rho(ni, 1.0), sie(ni), u(ni);
    for (int i = 0; i < ni; ++i) {
      double xcen = xmin[3 * l] + (i - ng + 0.5) * dx;
      xcen = std::fmin(std::fmax(xcen, -0.5 + 0.5 * dx), 0.5 - 0.5 * dx);  // outflow ghost = edge cell
      sie[i] = cv * (xcen < 0.0 ? t_hot : t_cold);
      u[i] = rho[i] * sie[i];
    }
    for (int f = 0; f < 8; ++f) fptr[f][l] = dev_alloc<double>(ni);
    if (l >= nowned)  // a halo copy: its interior arrives from the owner below (ghost cells: as computed)
      for (int i = ng; i < ng + nx; ++i) rho[i] = sie[i] = u[i] = 0.0;
    HIP_OK(hipMemcpy(fptr[0][l], rho.data(), ni * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fptr[1][l], sie.data(), ni * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(fptr[2][l], u.data(), ni * sizeof(double), hipMemcpyHostToDevice));
  }
  (void)fields;
  jb_mesh_view v{};
  v.ndim = 1; v.ng = ng; v.nblocks = nb; v.nblocks_total = nblocks_total;
  v.nx[0] = nx; v.nx[1] = 1; v.nx[2] = 1;
  v.nleaf[0] = nblocks_total; v.nleaf[1] = 1; v.nleaf[2] = 1;
  v.bc[0] = v.bc[1] = JB_BC_REFLECT;
  v.bc[2] = v.bc[3] = v.bc[4] = v.bc[5] = JB_BC_PERIODIC;
  v.rank = rank;
  for (int d = 0; d < 3; ++d) { v.gmin[d] = -0.5; v.gmax[d] = 0.5; }
  v.leaf_map = leaf_map.data(); v.owner = owner.data(); v.local_index = local_index.data();
  v.gid = gid.data(); v.owned = halo.owned.data();
  v.blk_xmin = xmin.data(); v.blk_xmax = xmax.data(); v.blk_dx = dxs.data();
  v.blk_level = level.data(); v.blk_nbr_lev = nbr_lev.data();
  v.rho = fptr[0].data(); v.sie = fptr[1].data(); v.u = fptr[2].data(); v.fleck = fptr[3].data();
  v.tally = fptr[4].data(); v.edelta = fptr[5].data(); v.src_ew = fptr[6].data(); v.src_num = fptr[7].data();
  jb_mesh *mesh = nullptr;
  JB_OK(jb_mesh_create(ctx, &v, &mesh));

  // ---- the halo copies' material state: the owners' interior cells (the refresh a host repeats
  //      after every UpdateFluid when do_feedback is on; here once)
  {
    const int nxs[3] = {nx, 1, 1};
    const jaybenne_amd::HaloRefreshPlan rp = jaybenne_amd::PlanHaloRefresh(topo, rank, nranks, nxs, ng, halo_rings);
    auto upload = [&](const std::vector<int32_t> &h) {
      int32_t *d = dev_alloc<int32_t>(h.size());
      if (!h.empty()) HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      return d;
    };
    int32_t *serve_blk = upload(rp.serve_blk), *serve_cell = upload(rp.serve_cell);
    int32_t *dst_blk = upload(rp.dst_blk), *dst_cell = upload(rp.dst_cell);
    int32_t *src_blk = upload(rp.src_blk), *src_cell = upload(rp.src_cell);
    const size_t nsend = rp.serve_blk.size(), nrecv = rp.dst_blk.size();
    double *send_dev = dev_alloc<double>(nsend), *recv_dev = dev_alloc<double>(nrecv);
    std::vector<double> sbuf(nsend + 1), rbuf(nrecv + 1);
    std::vector<int> hc(nranks), hd(nranks), gc(nranks), gd(nranks);
    int so = 0, ro = 0;
    for (int r = 0; r < nranks; ++r) {
      hc[r] = (int)rp.send_counts[r]; hd[r] = so; so += hc[r];
      gc[r] = (int)rp.recv_counts[r]; gd[r] = ro; ro += gc[r];
    }
    for (int field : {JB_FIELD_RHO, JB_FIELD_SIE, JB_FIELD_U}) {
      JB_OK(jb_gather_cells(ctx, mesh, field, (int64_t)nsend, serve_blk, serve_cell, send_dev));
      JB_OK(jb_synchronize(ctx));
      if (nsend) HIP_OK(hipMemcpy(sbuf.data(), send_dev, nsend * sizeof(double), hipMemcpyDeviceToHost));
      MPI_Alltoallv(sbuf.data(), hc.data(), hd.data(), MPI_DOUBLE, rbuf.data(), gc.data(), gd.data(), MPI_DOUBLE,
                    MPI_COMM_WORLD);
      if (nrecv) HIP_OK(hipMemcpy(recv_dev, rbuf.data(), nrecv * sizeof(double), hipMemcpyHostToDevice));
      JB_OK(jb_fill_cells(ctx, mesh, field, (int64_t)nrecv, 1, dst_blk, dst_cell, src_blk, src_cell, recv_dev));
    }
    JB_OK(jb_synchronize(ctx));
  }

  // ---- swarm: one pool per rank, room for every photon of the problem (they may all come here)
  jb_swarm_view sw{};
  sw.capacity = nparticles + nparticles / 4 + 4096;
  // (test knobs: a swarm with little room makes the runs close their holes on the way -- jb_exchange's
  // capacity protocol --, a tiny record buffer makes every rank stop together instead of one hanging the rest)
  if (const char *e = std::getenv("JB_HANDOFF_CAPACITY")) sw.capacity = std::atoll(e);
  sw.x = dev_alloc<double>(sw.capacity); sw.y = dev_alloc<double>(sw.capacity); sw.z = dev_alloc<double>(sw.capacity);
  sw.vx = dev_alloc<double>(sw.capacity); sw.vy = dev_alloc<double>(sw.capacity); sw.vz = dev_alloc<double>(sw.capacity);
  sw.t = dev_alloc<double>(sw.capacity); sw.w = dev_alloc<double>(sw.capacity); sw.e = dev_alloc<double>(sw.capacity);
  sw.ip = dev_alloc<int32_t>(sw.capacity); sw.jp = dev_alloc<int32_t>(sw.capacity); sw.kp = dev_alloc<int32_t>(sw.capacity);
  sw.blk = dev_alloc<int32_t>(sw.capacity); sw.status = dev_alloc<int32_t>(sw.capacity);
  sw.id = dev_alloc<uint64_t>(sw.capacity); sw.rng = dev_alloc<uint64_t>(sw.capacity);
  int32_t *prefix = dev_alloc<int32_t>((size_t)nb * nx);

  // ---- InitializeRadiation (jaybenne.cpp:570-578): thermal source block by block; stream ids
  //      are global creation indices, so every rank must know every block's count
  std::vector<int32_t> nper(nb);
  JB_OK(jb_update_derived_transport_fields(ctx, mesh, p.dt));
  JB_OK(jb_source_photons_count(ctx, mesh, JB_SOURCE_THERMAL, 0.0, 1, 0u, nper.data(), prefix));
  std::vector<long long> counts(nblocks_total, 0), all_counts(nblocks_total, 0);
  for (int l = 0; l < nb; ++l) counts[gid[l]] = nper[l];   // (halo copies source nothing: zeros)
  MPI_Allreduce(counts.data(), all_counts.data(), nblocks_total, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
  // (the plan shared with the single-rank tasks and the Parthenon adapter: include/jaybenne_amd.hpp)
  const jaybenne_amd::SourcePlan plan = jaybenne_amd::PlanSource(nper, gid, all_counts, 0ull, 0);
  const long long n_global0 = (long long)plan.next_id;
  JB_OK(jb_source_photons_fill(ctx, mesh, &sw, JB_SOURCE_THERMAL, 0.0, 0.0, nper.data(), prefix,
                               plan.slot_base.data(), plan.id_base.data()));
  sw.n = plan.total_local;

  // total weight (host sum of this rank's photons), reduced over ranks
  auto total_weight = [&]() {
    std::vector<double> w((size_t)sw.n);
    std::vector<int32_t> st((size_t)sw.n);
    if (sw.n) {
      HIP_OK(hipMemcpy(w.data(), sw.w, sw.n * sizeof(double), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(st.data(), sw.status, sw.n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    double s = 0.0;
    for (long long i = 0; i < sw.n; ++i) if (st[i] == JB_ST_ACTIVE) s += w[i];
    double g = 0.0;
    MPI_Allreduce(&s, &g, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    return g;
  };
  const double e0 = total_weight();

  // ---- cycles
  int64_t *rec_dev = dev_alloc<int64_t>((size_t)sw.capacity / 4 * JB_RECORD_WORDS + JB_RECORD_WORDS);   // (allocated at full size)
  const int64_t rec_cap = std::getenv("JB_HANDOFF_REC_CAP") ? std::atoll(std::getenv("JB_HANDOFF_REC_CAP")) : sw.capacity / 4;
  // (jb_exchange: a receive buffer of its own -- both directions are in flight at once)
  const bool one_call = std::strcmp(exchange, "tasks") != 0;
  int64_t *recv_dev = one_call ? dev_alloc<int64_t>((size_t)rec_cap * JB_RECORD_WORDS + JB_RECORD_WORDS) : nullptr;
  jb_exchange_transport tr{};
  MpiTransport mpi_tr{nranks};
  ncclComm_t nccl = nullptr;
  if (std::strcmp(exchange, "mpi") == 0) {
    tr.handle = &mpi_tr; tr.all_gather_u64 = mpi_all_gather_u64; tr.all_to_all_v = mpi_all_to_all_v;
  } else if (std::strcmp(exchange, "rccl") == 0) {
    ncclUniqueId uid;
    if (rank == 0 && ncclGetUniqueId(&uid) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); MPI_Abort(MPI_COMM_WORLD, 6); }
    MPI_Bcast(&uid, (int)sizeof uid, MPI_BYTE, 0, MPI_COMM_WORLD);
    const ncclResult_t nr_ = ncclCommInitRank(&nccl, nranks, uid, rank);
    if (nr_ != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed: %s\n", ncclGetErrorString(nr_)); MPI_Abort(MPI_COMM_WORLD, 6); }
    JB_OK(jb_transport_rccl(nccl, rank, nranks, &tr));
    // self-test of the transport's payload path (a rank may send to itself in a grouped send / recv):
    // 5 records from one buffer into the other through ncclSend / ncclRecv, bit for bit
    {
      std::vector<int64_t> pat(5 * JB_RECORD_WORDS), back(5 * JB_RECORD_WORDS, 0);
      for (size_t q = 0; q < pat.size(); ++q) pat[q] = (int64_t)(0x0123456789abcdefll * (long long)(q + 1) + rank);
      HIP_OK(hipMemcpy(rec_dev, pat.data(), pat.size() * 8, hipMemcpyHostToDevice));
      std::vector<int64_t> c(nranks, 0), o(nranks, 0);
      c[rank] = 5;
      if (tr.all_to_all_v(tr.handle, rec_dev, c.data(), o.data(), recv_dev, c.data(), o.data(), JB_RECORD_WORDS, nullptr) != 0) {
        std::fprintf(stderr, "RCCL self send / recv failed\n"); MPI_Abort(MPI_COMM_WORLD, 6);
      }
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipMemcpy(back.data(), recv_dev, back.size() * 8, hipMemcpyDeviceToHost));
      if (back != pat) { std::fprintf(stderr, "RCCL self send / recv changed the records\n"); MPI_Abort(MPI_COMM_WORLD, 6); }
      if (rank == 0) std::printf("RCCL transport: grouped self send / recv of 5 records ok\n");
    }
  }
  std::vector<int64_t> send_counts(nranks), recv_counts(nranks);
  std::vector<int> sc(nranks), sd(nranks), rc(nranks), rd(nranks);
  long long handed_total = 0, iterations_total = 0, compactions_total = 0;
  bool ok = true;
  double time = 0.0;
  for (int cyc = 0; cyc < cycles; ++cyc) {
    JB_OK(jb_update_derived_transport_fields(ctx, mesh, p.dt));
    JB_OK(jb_zero_energy_tally(ctx, mesh));
    int64_t first = 0;
    for (int it = 0; it < p.max_transport_iterations; ++it) {
      const int64_t last = sw.n;
      JB_OK(jb_transport_photons(ctx, mesh, &sw, time, p.dt, first, last, /*fuse_census_tally=*/1));
      if (one_call) {
        // MeshResetCommunication -> MeshSend -> MeshReceive in one call (include/jaybenne_amd.h: jb_exchange)
        int64_t nsent = 0, nrecv1 = 0, moved1 = 0;
        const int64_t n_before = sw.n;
        jb_status xs = jb_exchange(ctx, mesh, &sw, first, last, rank, nranks, &tr, rec_dev, rec_cap, recv_dev, rec_cap,
                                   &nsent, &nrecv1, &moved1);
        if (xs == JB_ERR_CAPACITY) {
          // Some rank has no room (every rank gets this answer in the same call, so all of them are here):
          // close the holes earlier departures left and go again -- nothing was packed yet, and what is
          // still to go is found by its status, wherever the compaction has moved it.  (The buffers of this
          // program are fixed: if one of THEM is too small the second call fails and the program stops.)
          JB_OK(jb_remove_marked_particles(ctx, &sw));
          ++compactions_total;
          xs = jb_exchange(ctx, mesh, &sw, 0, sw.n, rank, nranks, &tr, rec_dev, rec_cap, recv_dev, rec_cap, &nsent,
                           &nrecv1, &moved1);
        }
        JB_OK(xs);
        ++iterations_total;
        if (moved1 == 0) break;
        handed_total += nsent;
        first = sw.n - nrecv1;      // the arrivals, appended at the end: the next transport pass covers only them
        (void)n_before;
        continue;
      }
      // MeshSend: records grouped by destination rank, counts on the host
      JB_OK(jb_pack_outgoing(ctx, mesh, &sw, first, last, nranks, rec_dev, rec_cap, send_counts.data()));
      MPI_Alltoall(send_counts.data(), 1, MPI_INT64_T, recv_counts.data(), 1, MPI_INT64_T, MPI_COMM_WORLD);
      long long nsend = 0, nrecv = 0;
      for (int r = 0; r < nranks; ++r) {
        sd[r] = (int)(nsend * JB_RECORD_WORDS); sc[r] = (int)(send_counts[r] * JB_RECORD_WORDS);
        rd[r] = (int)(nrecv * JB_RECORD_WORDS); rc[r] = (int)(recv_counts[r] * JB_RECORD_WORDS);
        nsend += send_counts[r]; nrecv += recv_counts[r];
      }
      // completion test of the sublist (jaybenne.cpp:130-131): did anything move anywhere?
      long long moved = 0;
      MPI_Allreduce(&nsend, &moved, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
      ++iterations_total;
      if (moved == 0) break;
      handed_total += nsend;
      std::vector<int64_t> sbuf((size_t)nsend * JB_RECORD_WORDS + 1), rbuf((size_t)nrecv * JB_RECORD_WORDS + 1);
      if (nsend) HIP_OK(hipMemcpy(sbuf.data(), rec_dev, nsend * JB_RECORD_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost));
      MPI_Alltoallv(sbuf.data(), sc.data(), sd.data(), MPI_INT64_T, rbuf.data(), rc.data(), rd.data(),
                    MPI_INT64_T, MPI_COMM_WORLD);
      // MeshReceive: append the arrivals; the next transport pass covers only them
      if (sw.n + nrecv > sw.capacity) JB_OK(jb_remove_marked_particles(ctx, &sw));  // close the holes
      first = sw.n;
      if (nrecv) {
        HIP_OK(hipMemcpy(rec_dev, rbuf.data(), nrecv * JB_RECORD_WORDS * sizeof(int64_t), hipMemcpyHostToDevice));
        JB_OK(jb_unpack_incoming(ctx, mesh, &sw, rec_dev, nrecv));
      }
    }
    JB_OK(jb_remove_marked_particles(ctx, &sw));
    time += p.dt;
    // checks
    int64_t unfinished = 0;
    JB_OK(jb_check_completion(ctx, &sw, time, &unfinished));
    long long n_local = sw.n, n_global = 0, unf = unfinished, unf_g = 0;
    MPI_Allreduce(&n_local, &n_global, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
    MPI_Allreduce(&unf, &unf_g, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
    double tally_local = 0.0, tally_global = 0.0;
    std::vector<double> tl(ni);
    for (int l = 0; l < nowned; ++l) {
      HIP_OK(hipMemcpy(tl.data(), fptr[4][l], ni * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = ng; i < ng + nx; ++i) tally_local += tl[i] * dx;
    }
    MPI_Allreduce(&tally_local, &tally_global, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    const double e = total_weight();
    const bool cyc_ok = n_global == n_global0 && unf_g == 0 && std::fabs(e - e0) <= 1e-12 * e0 &&
                        std::fabs(tally_global - e0) <= 1e-10 * e0;
    ok = ok && cyc_ok;
    if (rank == 0)
      std::printf("cycle %d: photons %lld (start %lld), unfinished %lld, weight %.15e (start %.15e), "
                  "tally integral %.15e  %s\n", cyc + 1, n_global, n_global0, unf_g, e, e0, tally_global,
                  cyc_ok ? "ok" : "MISMATCH");
  }
  jb_transport_stats st{};
  JB_OK(jb_get_transport_stats(ctx, &st, 0));
  long long ev = st.n_events, ev_g = 0, handed_g = 0;
  MPI_Allreduce(&ev, &ev_g, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
  MPI_Allreduce(&handed_total, &handed_g, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
  if (rank == 0 && compactions_total)
    std::printf("jb_exchange reported JB_ERR_CAPACITY %lld time(s): holes closed, exchange repeated\n", compactions_total);
  if (rank == 0)
    std::printf("%d rank(s), %d blocks, %d halo ring(s): %lld events, %lld photons handed between ranks in %lld "
                "transport iterations (%.2f per cycle)  -> %s\n", nranks, nblocks_total, halo_rings, ev_g, handed_g,
                iterations_total, (double)iterations_total / cycles,
                ok && (nranks == 1 || handed_g > 0) ? "HANDOFF OK" : "HANDOFF FAILED");
  if (dump_prefix) {  // this rank's photons: n, then per photon id, x, vx, t, w, rng (8 bytes each), global block, ip (4 + 4)
    const size_t n = (size_t)sw.n;
    std::vector<uint64_t> id(n), rng(n);
    std::vector<double> x(n), vx(n), t(n), w(n);
    std::vector<int32_t> blk(n), ip(n);
    if (n) {
      HIP_OK(hipMemcpy(id.data(), sw.id, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(rng.data(), sw.rng, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(x.data(), sw.x, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(vx.data(), sw.vx, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(t.data(), sw.t, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(w.data(), sw.w, n * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(blk.data(), sw.blk, n * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(ip.data(), sw.ip, n * 4, hipMemcpyDeviceToHost));
    }
    char name[512];
    std::snprintf(name, sizeof name, "%s.%d.bin", dump_prefix, rank);
    FILE *fh = std::fopen(name, "wb");
    if (!fh) { std::fprintf(stderr, "cannot write %s\n", name); MPI_Abort(MPI_COMM_WORLD, 5); }
    const uint64_t n64 = n;
    std::fwrite(&n64, 8, 1, fh);
    for (size_t q = 0; q < n; ++q) {
      const int32_t g = gid[(size_t)blk[q]];
      std::fwrite(&id[q], 8, 1, fh); std::fwrite(&x[q], 8, 1, fh); std::fwrite(&vx[q], 8, 1, fh);
      std::fwrite(&t[q], 8, 1, fh); std::fwrite(&w[q], 8, 1, fh); std::fwrite(&rng[q], 8, 1, fh);
      std::fwrite(&g, 4, 1, fh); std::fwrite(&ip[q], 4, 1, fh);
    }
    std::fclose(fh);
  }
  const int rcode = ok && (nranks == 1 || handed_g > 0) ? 0 : 1;
  if (nccl) { jb_transport_release(&tr); ncclCommDestroy(nccl); }
  jb_mesh_destroy(mesh);
  jb_finalize(ctx);
  MPI_Finalize();
  return rcode;
}
