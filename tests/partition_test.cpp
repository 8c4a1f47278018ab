// Block -> rank partition shared by the C++ hosts (include/jaybenne_amd.hpp: PartitionBlocks, SiblingGroups,
// BlockCost, RankShare).  Reads from stdin: ndim nblocks nranks, then per block (Z-order): level, logical
// location (3), cost.  Prints the run boundaries and every rank's share of a block of 1000003 new photons.
// tests/test_cabi.py feeds it the reference's SMR meshes and holds it to jaybenne_amd.mesh.Mesh.partition.
// Compiled with the host compiler only (header-only: no HIP, no library).
#include <cstdio>
#include <vector>

#include "jaybenne_amd.hpp"

int main() {
  int ndim = 0, nb = 0, nranks = 0;
  if (std::scanf("%d %d %d", &ndim, &nb, &nranks) != 3) return 2;
  std::vector<int32_t> level((size_t)nb), lloc(3 * (size_t)nb);
  std::vector<double> cost((size_t)nb);
  for (int b = 0; b < nb; ++b)
    if (std::scanf("%d %d %d %d %lf", &level[b], &lloc[3 * b], &lloc[3 * b + 1], &lloc[3 * b + 2], &cost[b]) != 5) return 2;
  const std::vector<int64_t> group = jaybenne_amd::SiblingGroups(ndim, level, lloc);
  std::printf("group");
  for (int b = 0; b < nb; ++b) std::printf(" %lld", (long long)group[b]);
  std::printf("\n");
  const std::vector<int32_t> bounds = jaybenne_amd::PartitionBlocks(cost, nranks, group);
  std::printf("bounds");
  for (int r = 0; r <= nranks; ++r) std::printf(" %d", bounds[r]);
  std::printf("\n");
  // a rank's share of a block's photons: the shares tile 0 .. n exactly
  const std::vector<int32_t> nper = {1000003, 0, 7};
  for (size_t q = 0; q < nper.size(); ++q) {
    long long next = 0;
    for (int r = 0; r < nranks; ++r) {
      std::vector<int32_t> first, count;
      jaybenne_amd::RankShare(nper, r, nranks, &first, &count);
      if (first[q] != next) return 3;
      next += count[q];
    }
    if (next != nper[q]) return 4;
  }
  // BlockCost on BASELINE configs[1]'s cell: c dt = 1 cm, sigma_s = 1e3, dx = 1/256 in 3-D -> 1384 events
  const double dx[3] = {1.0 / 256, 1.0 / 256, 1.0 / 256};
  std::printf("cost_c2 %.17g\n", jaybenne_amd::BlockCost(3, dx, 1.0, 0.0, 1.0e3, false, 5.0));
  const double dxc[3] = {1.0 / 128, 1.0 / 128, 1.0};
  std::printf("cost_ddmc %.17g\n", jaybenne_amd::BlockCost(2, dxc, 1.0, 0.0, 1.0e3, true, 5.0));
  std::printf("partition ok\n");
  return 0;
}
