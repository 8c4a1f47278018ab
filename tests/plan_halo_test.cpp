// Host-side arithmetic of the halo copies shared by every C++ host (include/jaybenne_amd.hpp:
// PlanHalo, FaceNeighbourLevels, PlanHaloRefresh) -- what examples/handoff_mpi.cpp and the Parthenon
// adapter call.  Reads a mesh topology from stdin (written by tests/test_cabi.py from
// jaybenne_amd.mesh.Mesh, the Python host's tested statement of the same rules), prints, per rank:
// the halo copies, the neighbour levels of every resident block and the refresh counts.
// Compiled with the host compiler only (no HIP, no library needed: header-only).
#include <cstdio>
#include <vector>

#include "jaybenne_amd.hpp"

int main() {
  jaybenne_amd::MeshTopology T;
  int nranks = 1, nx[3], ng = 2, per[6];
  if (std::scanf("%d %d %d", &T.ndim, &T.nblocks_total, &nranks) != 3) return 2;
  for (int d = 0; d < 3; ++d)
    if (std::scanf("%lf %lf %d %d", &T.gmin[d], &T.gmax[d], &T.nleaf[d], &nx[d]) != 4) return 2;
  for (int f = 0; f < 6; ++f) { if (std::scanf("%d", &per[f]) != 1) return 2; T.periodic[f] = per[f] != 0; }
  const size_t nleaf = (size_t)T.nleaf[0] * T.nleaf[1] * T.nleaf[2];
  std::vector<int32_t> leaf(nleaf), owner(T.nblocks_total), level(T.nblocks_total);
  std::vector<double> xmin(3 * (size_t)T.nblocks_total), xmax(3 * (size_t)T.nblocks_total);
  for (auto &v : leaf) if (std::scanf("%d", &v) != 1) return 2;
  for (int g = 0; g < T.nblocks_total; ++g) {
    if (std::scanf("%d %d", &owner[g], &level[g]) != 2) return 2;
    for (int d = 0; d < 3; ++d) if (std::scanf("%lf %lf", &xmin[3 * g + d], &xmax[3 * g + d]) != 2) return 2;
  }
  T.leaf_map = leaf.data(); T.owner = owner.data(); T.level = level.data();
  T.blk_xmin = xmin.data(); T.blk_xmax = xmax.data();
  long long sent = 0, received = 0;
  for (int r = 0; r < nranks; ++r) {
    const jaybenne_amd::HaloPlan pl = jaybenne_amd::PlanHalo(T, r);
    std::printf("rank %d halo", r);
    for (size_t q = (size_t)pl.nowned; q < pl.resident_gids.size(); ++q) std::printf(" %d", pl.resident_gids[q]);
    std::printf("\n");
    for (size_t q = 0; q < pl.resident_gids.size(); ++q) {
      if (pl.local_index[(size_t)pl.resident_gids[q]] != (int32_t)q || pl.owned[q] != (q < (size_t)pl.nowned)) return 3;
      int32_t lev[6];
      jaybenne_amd::FaceNeighbourLevels(T, pl.resident_gids[q], lev);
      std::printf("nbr %d %d %d %d %d %d %d\n", pl.resident_gids[q], lev[0], lev[1], lev[2], lev[3], lev[4], lev[5]);
    }
    const jaybenne_amd::HaloRefreshPlan rp = jaybenne_amd::PlanHaloRefresh(T, r, nranks, nx, ng);
    const long long ncell = (long long)nx[0] * nx[1] * nx[2];
    long long s = 0, v = 0;
    for (int q = 0; q < nranks; ++q) { s += rp.send_counts[q]; v += rp.recv_counts[q]; }
    if (v != ncell * (long long)(pl.resident_gids.size() - pl.nowned)) return 4;   // every copy filled once
    if ((long long)rp.dst_blk.size() != v || (long long)rp.serve_blk.size() != s) return 5;
    sent += s; received += v;
    std::printf("refresh %d send %lld recv %lld\n", r, s, v);
  }
  if (sent != received) return 6;   // what the ranks send is what the ranks expect
  std::printf("plan_halo ok\n");
  return 0;
}
