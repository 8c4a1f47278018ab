// Host-side arithmetic of SourcePhotons shared by every host (include/jaybenne_amd.hpp: PlanSource,
// SourceEpoch) -- what examples/mcblock_amd.cpp, examples/handoff_mpi.cpp and the Parthenon adapter
// (adapters/parthenon/jaybenne_amd_tasks.cpp) all call.  Stream ids must not depend on the
// block -> rank partition: 20 blocks dealt unevenly to 1, 3 and 8 ranks give every block the same
// first id, each rank packs its own photons without gaps, and every rank agrees on the next
// unused id.  Compiled with the host compiler only (no HIP, no library needed: header-only).
#include <cassert>
#include <cstdio>
#include <vector>

#include "jaybenne_amd.hpp"

using jaybenne_amd::PlanSource;
using jaybenne_amd::SourceEpoch;
using jaybenne_amd::SourcePlan;
using jaybenne_amd::SourceType;

int main() {
  const int nblocks = 20;
  std::vector<long long> all(nblocks);
  for (int g = 0; g < nblocks; ++g) all[g] = (g * 37 + 11) % 29;   // some blocks source nothing
  const uint64_t next0 = 1000;
  std::vector<uint64_t> want(nblocks);
  uint64_t run = next0;
  for (int g = 0; g < nblocks; ++g) { want[g] = run; run += (uint64_t)all[g]; }
  for (int nranks : {1, 3, 8}) {
    for (int r = 0; r < nranks; ++r) {
      std::vector<int32_t> gid, nper;
      for (int g = 0; g < nblocks; ++g)
        if ((g * 7 + 3) % nranks == r) { gid.push_back(g); nper.push_back((int32_t)all[g]); }   // uneven deal
      const int64_t n_now = 5 * r;
      const SourcePlan pl = PlanSource(nper, gid, all, next0, n_now);
      assert(pl.next_id == run);
      int64_t slot = n_now;
      for (size_t b = 0; b < gid.size(); ++b) {
        assert(pl.id_base[b] == want[gid[b]]);
        assert(pl.slot_base[b] == slot);
        slot += nper[b];
      }
      assert(pl.total_local == slot - n_now);
    }
  }
  // epochs: the same on every rank by construction, distinct per (cycle, type), 0 at initialisation,
  // k for the emission source of cycle k (the order a run makes its source calls in)
  assert(SourceEpoch(0, SourceType::thermal) == 0u);
  assert(SourceEpoch(1, SourceType::emission) == 1u && SourceEpoch(2, SourceType::emission) == 2u);
  assert(SourceEpoch(1, SourceType::thermal) != SourceEpoch(1, SourceType::emission));
  assert(SourceEpoch(7, SourceType::emission) != SourceEpoch(8, SourceType::thermal));
  assert(SourceEpoch(500000, SourceType::emission) < (1u << 20) && SourceEpoch(500000, SourceType::thermal) < (1u << 20));
  std::puts("plan_source ok");
  return 0;
}
