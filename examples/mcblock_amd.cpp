// mcblock_amd.cpp -- a native host application on the C++ mirror (include/jaybenne_amd.hpp):
// what the reference's src/mcblock does around the jaybenne package, on one GPU, with no Python
// and no PyTorch in the process.
//
//   mcblock_amd -i <deck> [block/key=value ...] [--tolerance X] [--dump file]
//
// Input deck syntax and the trailing overrides are Parthenon's (reference inputs/*.in; the
// regression harness tst/regression_test.py:85-145 rewrites decks the same way).  It sets up the
// material state of mcblock.cpp:155-203 (problem ids stepdiff, inf, inf_stiff), calls
// InitializeRadiation and then RadiationStep per cycle (mcblock_driver.cpp:38-74), and finally
// prints the five error numbers of tst/regression_test.py:408-412 against the analytic solution
// of tst/stepdiff.py:33-46 (stepdiff) or a T0^4 (inf decks).  Exit code 0 iff the solution-
// weighted mean fractional error is within --tolerance (when given).
//
// Static mesh refinement (<parthenon/static_refinementN> regions, 2:1 balanced block tree in
// Z-order) is built the way jaybenne_amd/mesh.py builds it, so all seven decks of the reference's
// inputs/ run.  With do_feedback the host update tasks of mcblock_driver.cpp:58-74 follow every
// step: ghost zones of internal_energy refreshed through jb_fill_cells (the source map is the one
// of jaybenne_amd/mesh.py: copy / injection / 2^ndim-cell average), then UpdateDerived
// (sie = u / rho, mcblock.cpp:208-232) as this application's own kernel.
// Not covered here (the Python driver jaybenne_amd/mcblock.py does it): several ranks.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <tuple>
#include <string>
#include <vector>

#include "jaybenne_amd.hpp"

namespace jb = jaybenne_amd;

#define HIP_OK(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                \
      std::exit(3);                                                                         \
    }                                                                                       \
  } while (0)

// ---- Parthenon-style input deck ---------------------------------------------------------------
struct ParameterInput {
  std::map<std::string, std::map<std::string, std::string>> blocks;

  static std::string trim(const std::string &s) {
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(" \t\r\n") - a + 1);
  }
  void Load(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open deck " + path);
    std::string line, block, pending;
    while (std::getline(in, line)) {
      line = trim(line.substr(0, line.find('#')));
      if (line.empty()) continue;
      if (line.front() == '<' && line.back() == '>') {
        block = trim(line.substr(1, line.size() - 2));
        blocks[block];
        pending.clear();
        continue;
      }
      const bool cont = line.back() == '&';
      if (cont) line = trim(line.substr(0, line.size() - 1));
      if (!pending.empty()) {
        blocks[block][pending] += line;
      } else {
        const size_t eq = line.find('=');
        if (eq == std::string::npos) throw std::runtime_error("deck line without '=': " + line);
        pending = trim(line.substr(0, eq));
        blocks[block][pending] = trim(line.substr(eq + 1));
      }
      if (!cont) pending.clear();
    }
  }
  void Override(const std::string &item) {  // block/key=value
    const size_t eq = item.find('='), sl = item.rfind('/', eq);
    if (eq == std::string::npos || sl == std::string::npos)
      throw std::runtime_error("override '" + item + "' is not of the form block/key=value");
    blocks[item.substr(0, sl)][item.substr(sl + 1, eq - sl - 1)] = item.substr(eq + 1);
  }
  bool Has(const std::string &b, const std::string &k) const {
    auto it = blocks.find(b);
    return it != blocks.end() && it->second.count(k);
  }
  std::string GetString(const std::string &b, const std::string &k) const {
    if (!Has(b, k)) throw std::runtime_error("Parameter name '" + k + "' not found in block '" + b + "'");
    return blocks.at(b).at(k);
  }
  std::string GetOrAddString(const std::string &b, const std::string &k, const std::string &d) {
    if (!Has(b, k)) blocks[b][k] = d;
    return blocks[b][k];
  }
  double GetReal(const std::string &b, const std::string &k) const { return std::stod(GetString(b, k)); }
  double GetOrAddReal(const std::string &b, const std::string &k, double d) {
    return Has(b, k) ? GetReal(b, k) : d;
  }
  long GetInteger(const std::string &b, const std::string &k) const { return (long)std::stod(GetString(b, k)); }
  long GetOrAddInteger(const std::string &b, const std::string &k, long d) {
    return Has(b, k) ? GetInteger(b, k) : d;
  }
  bool GetOrAddBoolean(const std::string &b, const std::string &k, bool d) {
    if (!Has(b, k)) return d;
    const std::string v = GetString(b, k);
    return v == "true" || v == "1" || v == "True";
  }
};

// ---- device buffers ---------------------------------------------------------------------------
template <class T>
struct DeviceArray {
  T *d = nullptr;
  size_t n = 0;
  void Alloc(size_t count) {
    Free();
    n = count;
    HIP_OK(hipMalloc((void **)&d, std::max<size_t>(count, 1) * sizeof(T)));
    HIP_OK(hipMemset(d, 0, std::max<size_t>(count, 1) * sizeof(T)));
  }
  void Free() {
    if (d) HIP_OK(hipFree(d));
    d = nullptr;
    n = 0;
  }
  void Upload(const std::vector<T> &h) { HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> Download(size_t count) const {
    std::vector<T> h(count);
    if (count) HIP_OK(hipMemcpy(h.data(), d, count * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
  ~DeviceArray() { Free(); }
};

// the photons swarm as the host owns it; Grow() is the pool growth of Swarm::AddEmptyParticles
struct Swarm {
  DeviceArray<double> f64[9];
  DeviceArray<int32_t> i32[5];
  DeviceArray<uint64_t> u64[2];
  void Bind(jb_swarm_view &v, int64_t cap) {
    double **pf[9] = {&v.x, &v.y, &v.z, &v.vx, &v.vy, &v.vz, &v.t, &v.w, &v.e};
    int32_t **pi[5] = {&v.ip, &v.jp, &v.kp, &v.blk, &v.status};
    uint64_t **pu[2] = {&v.id, &v.rng};
    for (int q = 0; q < 9; ++q) *pf[q] = f64[q].d;
    for (int q = 0; q < 5; ++q) *pi[q] = i32[q].d;
    for (int q = 0; q < 2; ++q) *pu[q] = u64[q].d;
    v.capacity = cap;
  }
  template <class T>
  static void GrowOne(DeviceArray<T> &a, size_t keep, size_t cap) {
    DeviceArray<T> b;
    b.Alloc(cap);
    if (keep) HIP_OK(hipMemcpy(b.d, a.d, keep * sizeof(T), hipMemcpyDeviceToDevice));
    std::swap(a.d, b.d);
    std::swap(a.n, b.n);
  }
  void Grow(jb_swarm_view &v, int64_t need) {
    HIP_OK(hipDeviceSynchronize());
    const int64_t cap = 2 * need;
    for (auto &a : f64) GrowOne(a, (size_t)v.n, (size_t)cap);
    for (auto &a : i32) GrowOne(a, (size_t)v.n, (size_t)cap);
    for (auto &a : u64) GrowOne(a, (size_t)v.n, (size_t)cap);
    Bind(v, cap);
  }
};

// mcblock::UpdateDerived (mcblock.cpp:208-232): sie = u / rho over entire blocks
__global__ void UpdateDerivedKernel(const double *rho, const double *u, double *sie, size_t n) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x)
    sie[q] = u[q] / rho[q];
}

static uint64_t Morton(const int l[3], int bits) {
  uint64_t key = 0;
  for (int b = 0; b < bits; ++b)
    for (int d = 0; d < 3; ++d) key |= (uint64_t)((l[d] >> b) & 1) << (3 * b + d);
  return key;
}

int main(int argc, char **argv) {
  try {
    std::string deck, dump;
    double tolerance = -1.0;
    std::vector<std::string> overrides;
    for (int a = 1; a < argc; ++a) {
      const std::string s = argv[a];
      if (s == "-i" && a + 1 < argc) deck = argv[++a];
      else if (s == "--tolerance" && a + 1 < argc) tolerance = std::stod(argv[++a]);
      else if (s == "--dump" && a + 1 < argc) dump = argv[++a];
      else overrides.push_back(s);
    }
    if (deck.empty()) {
      std::fprintf(stderr, "usage: mcblock_amd -i deck [block/key=value ...] [--tolerance X] [--dump file]\n");
      return 2;
    }
    ParameterInput pin;
    pin.Load(deck);
    for (const auto &o : overrides) pin.Override(o);

    // ---- mcblock::Initialize (mcblock.cpp:37-150) ----------------------------------------------
    if (pin.GetString("parthenon/time", "integrator") != "rk1")
      throw std::runtime_error("McBlock driver only supports first order time integration");
    const std::string problem_id = pin.GetString("parthenon/job", "problem_id");
    const double tt0 = pin.GetReal("mcblock", "initial_temperature");
    const double rho0 = pin.GetReal("mcblock", "initial_density");
    const std::string initial_radiation = pin.GetString("mcblock", "initial_radiation");
    const double gamma = pin.GetOrAddReal("mcblock", "gamma", 1.66666666667);
    const double cv = pin.GetOrAddReal("mcblock", "cv", 1. / (gamma - 1.));
    jb_eos eos = {JB_EOS_IDEAL_GAS, 0, gamma - 1., cv};
    // units: conversion factors from code to CGS (mcblock.cpp:84-91), the NonCGSUnits arguments
    const double ts = pin.GetOrAddReal("mcblock", "time_scale", 1.), ms = pin.GetOrAddReal("mcblock", "mass_scale", 1.),
                 ls = pin.GetOrAddReal("mcblock", "length_scale", 1.), tks = pin.GetOrAddReal("mcblock", "temperature_scale", 1.);
    const std::string om = pin.GetString("mcblock", "opacity_model");
    if (om != "none" && om != "constant" && om != "ep_bremss") throw std::runtime_error("Only none or constant opacity models supported!");
    // (the library takes kappa and the physical constants in code units)
    jb_opacity opacity = {om == "ep_bremss" ? JB_OPAC_EPBREMSS : JB_OPAC_GRAY, 0,
                          (om == "constant" ? pin.GetReal("mcblock", "opacity_constant_value") : 0.0) * (ms / (ls * ls)),
                          2.99792458e10 * (ts / ls), 5.670373e-5 * (ts * ts * ts * (tks * tks) * (tks * tks) / ms),
                          ts, ms, ls, tks};
    const std::string sm = pin.GetOrAddString("mcblock", "scattering_model", "none");
    if (sm != "none" && sm != "constant") throw std::runtime_error("Only none or constant scattering models supported!");
    jb_scattering scattering = {JB_SCAT_GRAY, 0,
                                (sm == "constant" ? pin.GetReal("mcblock", "scattering_constant_value") : 0.0) * (ms / (ls * ls)),
                                pin.GetOrAddReal("mcblock", "apm", 1.), ts, ms, ls, tks};

    // ---- jaybenne::Initialize (jaybenne.cpp:158-266) -------------------------------------------
    jb_params p;
    std::memset(&p, 0, sizeof p);
    p.num_particles = pin.GetInteger("jaybenne", "num_particles");
    p.dt = pin.GetOrAddReal("jaybenne", "dt", 1.7976931348623157e308);
    p.min_swarm_occupancy = pin.GetOrAddReal("jaybenne", "min_swarm_occupancy", 0.);
    p.numin = pin.GetOrAddReal("jaybenne", "numin", 2.2250738585072014e-308);
    p.numax = pin.GetOrAddReal("jaybenne", "numax", 1.7976931348623157e308);
    p.tau_ddmc = pin.GetOrAddReal("jaybenne", "tau_ddmc", 5.0);
    p.unique_rank_seeds = pin.GetOrAddBoolean("jaybenne", "unique_rank_seeds", true);
    p.seed = (int32_t)pin.GetOrAddInteger("jaybenne", "seed", 123);
    p.max_transport_iterations = (int32_t)pin.GetOrAddInteger("jaybenne", "max_transport_iterations", 10000);
    p.use_ddmc = pin.GetOrAddBoolean("jaybenne", "use_ddmc", false);
    const std::string strategy = pin.GetOrAddString("jaybenne", "source_strategy", "uniform");
    if (strategy != "uniform" && strategy != "energy")
      throw std::runtime_error("Only uniform or energy source strategies supported!");
    p.source_strategy = strategy == "uniform" ? JB_STRATEGY_UNIFORM : JB_STRATEGY_ENERGY;
    p.do_emission = pin.GetOrAddBoolean("jaybenne", "do_emission", true);
    p.do_feedback = pin.GetOrAddBoolean("jaybenne", "do_feedback", true);
    auto pkg = jb::Initialize(p, opacity, scattering, eos, 0);

    // ---- block tree: leaves in Z-order, geometry as jaybenne_amd/mesh.py forms it ------------
    const std::string mb = "parthenon/mesh";
    int mesh_nx[3], nx[3], nroot[3], ndim = 1;
    double gmin[3], gmax[3];
    for (int d = 0; d < 3; ++d) {
      const std::string a = std::to_string(d + 1);
      mesh_nx[d] = (int)pin.GetOrAddInteger(mb, "nx" + a, 1);
      nx[d] = (int)pin.GetOrAddInteger("parthenon/meshblock", "nx" + a, mesh_nx[d]);
      gmin[d] = pin.GetReal(mb, "x" + a + "min");
      gmax[d] = pin.GetReal(mb, "x" + a + "max");
      if (mesh_nx[d] % nx[d]) throw std::runtime_error("meshblock size does not divide the mesh");
      nroot[d] = mesh_nx[d] / nx[d];
      if (mesh_nx[d] > 1) ndim = d + 1;
    }
    const int ng = (int)pin.GetOrAddInteger(mb, "nghost", 2);
    auto bc_code = [&](const std::string &block, const std::string &key) {
      const std::string v = pin.GetOrAddString(block, key, "periodic");
      if (v == "periodic") return (int)JB_BC_PERIODIC;
      if (v == "jaybenne_reflecting") return (int)JB_BC_REFLECT;
      if (v == "outflow") return (int)JB_BC_OUTFLOW;
      throw std::runtime_error("unsupported boundary condition '" + v + "'");
    };
    bool mesh_periodic[6];
    for (int d = 0; d < 3; ++d) {
      mesh_periodic[2 * d] = bc_code(mb, "ix" + std::to_string(d + 1) + "_bc") == JB_BC_PERIODIC;
      mesh_periodic[2 * d + 1] = bc_code(mb, "ox" + std::to_string(d + 1) + "_bc") == JB_BC_PERIODIC;
    }
    struct Region { int level; double lo[3], hi[3]; };
    std::vector<Region> regions;
    if (pin.GetOrAddString(mb, "refinement", "none") == "static") {
      for (int n = 1; pin.blocks.count("parthenon/static_refinement" + std::to_string(n)); ++n) {
        const std::string rb = "parthenon/static_refinement" + std::to_string(n);
        Region r;
        r.level = (int)pin.GetInteger(rb, "level");
        for (int d = 0; d < 3; ++d) {
          r.lo[d] = pin.GetOrAddReal(rb, "x" + std::to_string(d + 1) + "min", gmin[d]);
          r.hi[d] = pin.GetOrAddReal(rb, "x" + std::to_string(d + 1) + "max", gmax[d]);
        }
        regions.push_back(r);
      }
    }
    struct Leaf {
      int level, l[3];
      bool operator<(const Leaf &o) const {
        return std::tie(level, l[0], l[1], l[2]) < std::tie(o.level, o.l[0], o.l[1], o.l[2]);
      }
    };
    auto bounds = [&](const Leaf &L, double lo[3], double hi[3]) {
      for (int d = 0; d < 3; ++d) {
        const int nbd = nroot[d] * (d < ndim ? (1 << L.level) : 1);
        const double ext = gmax[d] - gmin[d];
        lo[d] = gmin[d] + ext * ((double)L.l[d] / nbd);
        hi[d] = gmin[d] + ext * ((double)(L.l[d] + 1) / nbd);
      }
    };
    auto children = [&](const Leaf &L) {
      std::vector<Leaf> out;
      for (int c = 0; c < (1 << ndim); ++c) {
        Leaf ch{L.level + 1, {L.l[0], L.l[1], L.l[2]}};
        for (int d = 0; d < ndim; ++d) ch.l[d] = 2 * L.l[d] + ((c >> d) & 1);
        out.push_back(ch);
      }
      return out;
    };
    std::set<Leaf> leaves;
    for (int k = 0; k < nroot[2]; ++k)
      for (int j = 0; j < nroot[1]; ++j)
        for (int i = 0; i < nroot[0]; ++i) leaves.insert(Leaf{0, {i, j, k}});
    int max_level = 0;
    for (const auto &r : regions) max_level = std::max(max_level, r.level);
    for (int target = 1; target <= max_level; ++target) {  // refine what overlaps a finer region
      const std::vector<Leaf> snapshot(leaves.begin(), leaves.end());
      for (const Leaf &L : snapshot) {
        if (L.level != target - 1) continue;
        double lo[3], hi[3];
        bounds(L, lo, hi);
        for (const auto &r : regions) {
          if (r.level < target) continue;
          bool overlap = true;
          for (int d = 0; d < ndim; ++d) overlap = overlap && lo[d] < r.hi[d] && hi[d] > r.lo[d];
          if (overlap) {
            leaves.erase(L);
            for (const Leaf &c : children(L)) leaves.insert(c);
            break;
          }
        }
      }
    }
    // which leaf covers a block position of the finest level
    auto finest_extent = [&](int lev, int d) { return nroot[d] * (d < ndim ? (1 << lev) : 1); };
    auto build_leaf_grid = [&](int lev, std::vector<Leaf> &grid) {
      const int e0 = finest_extent(lev, 0), e1 = finest_extent(lev, 1), e2 = finest_extent(lev, 2);
      grid.assign((size_t)e0 * e1 * e2, Leaf{-1, {0, 0, 0}});
      for (const Leaf &L : leaves) {
        const int sft = lev - L.level;
        int r0[3], r1[3];
        for (int d = 0; d < 3; ++d) {
          r0[d] = d < ndim ? L.l[d] << sft : 0;
          r1[d] = d < ndim ? (L.l[d] + 1) << sft : 1;
        }
        for (int k = r0[2]; k < r1[2]; ++k)
          for (int j = r0[1]; j < r1[1]; ++j)
            for (int i = r0[0]; i < r1[0]; ++i) grid[((size_t)k * e1 + j) * e0 + i] = L;
      }
    };
    for (bool changed = true; changed;) {  // 2:1 balance across faces, edges and corners
      changed = false;
      int ml = 0;
      for (const Leaf &L : leaves) ml = std::max(ml, L.level);
      std::vector<Leaf> grid;
      build_leaf_grid(ml, grid);
      const int e0 = finest_extent(ml, 0), e1 = finest_extent(ml, 1), e2 = finest_extent(ml, 2);
      const int ext[3] = {e0, e1, e2};
      const std::vector<Leaf> snapshot(leaves.begin(), leaves.end());
      for (const Leaf &L : snapshot) {
        if (L.level < 2) continue;
        const int scale = 1 << (ml - L.level);
        for (int oz = (ndim > 2 ? -1 : 0); oz <= (ndim > 2 ? 1 : 0) && !changed; ++oz)
          for (int oy = (ndim > 1 ? -1 : 0); oy <= (ndim > 1 ? 1 : 0) && !changed; ++oy)
            for (int ox = -1; ox <= 1 && !changed; ++ox) {
              const int o[3] = {ox, oy, oz};
              int q[3];
              bool ok = true;
              for (int d = 0; d < 3; ++d) {
                int c = d < ndim ? (L.l[d] + o[d]) * scale : 0;
                if (c < 0 || c >= ext[d]) {
                  if (mesh_periodic[2 * d]) c = ((c % ext[d]) + ext[d]) % ext[d];
                  else ok = false;
                }
                q[d] = c;
              }
              if (!ok) continue;
              const Leaf nbl = grid[((size_t)q[2] * e1 + q[1]) * e0 + q[0]];
              if (nbl.level < L.level - 1 && leaves.count(nbl)) {
                leaves.erase(nbl);
                for (const Leaf &c : children(nbl)) leaves.insert(c);
                changed = true;
              }
            }
        if (changed) break;
      }
    }
    max_level = 0;
    for (const Leaf &L : leaves) max_level = std::max(max_level, L.level);
    const int nb = (int)leaves.size();
    int bits = 1;
    while ((1 << bits) < std::max({nroot[0], nroot[1], nroot[2]}) * (1 << max_level) + 1) ++bits;
    struct Loc { Leaf leaf; uint64_t key; };
    std::vector<Loc> locs;
    for (const Leaf &L : leaves) {
      int fl[3];
      for (int d = 0; d < 3; ++d) fl[d] = d < ndim ? L.l[d] << (max_level - L.level) : 0;
      locs.push_back(Loc{L, Morton(fl, bits)});
    }
    std::sort(locs.begin(), locs.end(), [](const Loc &a, const Loc &b) { return a.key < b.key; });
    int nleaf[3];
    for (int d = 0; d < 3; ++d) nleaf[d] = finest_extent(max_level, d);
    std::vector<double> xmin(3 * nb), xmax(3 * nb), dx(3 * nb);
    std::vector<int32_t> leaf_map((size_t)nleaf[0] * nleaf[1] * nleaf[2]), owner(nb, 0), ident(nb),
        level(nb, 0), nbr_lev(6 * nb, 0);
    for (int b = 0; b < nb; ++b) {
      ident[b] = b;
      level[b] = locs[b].leaf.level;
      double lo[3], hi[3];
      bounds(locs[b].leaf, lo, hi);
      for (int d = 0; d < 3; ++d) {
        xmin[3 * b + d] = lo[d];
        xmax[3 * b + d] = hi[d];
        dx[3 * b + d] = (hi[d] - lo[d]) / (double)nx[d];
      }
      const int sft = max_level - level[b];
      for (int k = (ndim > 2 ? locs[b].leaf.l[2] << sft : 0); k < (ndim > 2 ? (locs[b].leaf.l[2] + 1) << sft : 1); ++k)
        for (int j = (ndim > 1 ? locs[b].leaf.l[1] << sft : 0); j < (ndim > 1 ? (locs[b].leaf.l[1] + 1) << sft : 1); ++j)
          for (int i = locs[b].leaf.l[0] << sft; i < (locs[b].leaf.l[0] + 1) << sft; ++i)
            leaf_map[((size_t)k * nleaf[1] + j) * nleaf[0] + i] = b;
    }
    auto find_block = [&](const double pt[3]) {
      int q[3];
      for (int d = 0; d < 3; ++d) {
        const double ln = (gmax[d] - gmin[d]) / nleaf[d];
        q[d] = (int)std::floor((pt[d] - gmin[d]) / ln);
        q[d] = std::min(std::max(q[d], 0), nleaf[d] - 1);
      }
      return leaf_map[((size_t)q[2] * nleaf[1] + q[1]) * nleaf[0] + q[0]];
    };
    for (int b = 0; b < nb; ++b)  // neighbour level per face; own level at a physical boundary
      for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
          int &out = nbr_lev[6 * b + 2 * d + side];
          out = level[b];
          if (d >= ndim) continue;
          double pt[3];
          for (int dd = 0; dd < 3; ++dd) {
            const double fine = (gmax[dd] - gmin[dd]) / nleaf[dd];
            pt[dd] = 0.5 * (xmin[3 * b + dd] + xmax[3 * b + dd]) + 0.25 * fine * (dd < ndim ? 1.0 : 0.0);
          }
          const double fine_d = (gmax[d] - gmin[d]) / nleaf[d];
          pt[d] = side == 0 ? xmin[3 * b + d] - 0.25 * fine_d : xmax[3 * b + d] + 0.25 * fine_d;
          if (pt[d] < gmin[d] || pt[d] > gmax[d]) {
            if (!mesh_periodic[2 * d + side]) continue;
            pt[d] += (gmax[d] - gmin[d]) * (side == 0 ? 1 : -1);
          }
          out = level[find_block(pt)];
        }
    int is[3], ntot_dim[3];
    for (int d = 0; d < 3; ++d) {
      is[d] = d < ndim ? ng : 0;
      ntot_dim[d] = nx[d] + 2 * is[d];
    }
    const size_t ntot = (size_t)ntot_dim[0] * ntot_dim[1] * ntot_dim[2];
    const size_t ncell = (size_t)nx[0] * nx[1] * nx[2];

    // ---- fields: one device array per field, block b at offset b * ntot -----------------------
    const char *names[11] = {"rho", "sie", "u", "fleck", "tally", "edelta", "src_ew", "src_num", "P1", "P2", "P3"};
    DeviceArray<double> fields[11];
    std::vector<std::vector<double *>> ptrs(11, std::vector<double *>(nb, nullptr));
    for (int f = 0; f < 11; ++f) {
      if (f >= 8 && !p.use_ddmc) continue;
      fields[f].Alloc(nb * ntot);
      for (int b = 0; b < nb; ++b) ptrs[f][b] = fields[f].d + (size_t)b * ntot;
    }
    (void)names;

    // ---- ProblemGenerator (mcblock.cpp:155-203) + PostInitialization + ghost state -------------
    std::vector<double> h_rho(nb * ntot, rho0), h_sie(nb * ntot, cv * tt0), h_u(nb * ntot);
    if (problem_id == "stepdiff") {
      const double ttlow = 1.0e-5 * tt0;
      for (int b = 0; b < nb; ++b) {
        const double dxb = dx[3 * b], x0 = xmin[3 * b] - is[0] * dxb, half = 0.5 * dxb;
        for (int i = 0; i < ntot_dim[0]; ++i) {
          double x1v = x0 + (i + 0.5) * dxb;  // ghost zones: same function, clamped into the domain
          x1v = std::min(std::max(x1v, gmin[0] + half), gmax[0] - half);
          if (x1v >= 0.0)
            for (int k = 0; k < ntot_dim[2]; ++k)
              for (int j = 0; j < ntot_dim[1]; ++j)
                h_sie[(size_t)b * ntot + ((size_t)k * ntot_dim[1] + j) * ntot_dim[0] + i] = cv * ttlow;
        }
      }
    }
    for (size_t q = 0; q < h_u.size(); ++q) h_u[q] = h_rho[q] * h_sie[q];
    for (size_t q = 0; q < h_u.size(); ++q) h_sie[q] = h_u[q] / h_rho[q];  // UpdateDerived
    fields[0].Upload(h_rho);
    fields[1].Upload(h_sie);
    fields[2].Upload(h_u);

    // ---- the rank's MeshData ---------------------------------------------------------------
    jb_mesh_view view;
    std::memset(&view, 0, sizeof view);
    view.ndim = ndim; view.ng = ng; view.nblocks = nb; view.nblocks_total = nb; view.rank = 0;
    for (int d = 0; d < 3; ++d) {
      view.nx[d] = nx[d]; view.nleaf[d] = nleaf[d]; view.gmin[d] = gmin[d]; view.gmax[d] = gmax[d];
      view.bc[2 * d] = bc_code("parthenon/swarm", "ix" + std::to_string(d + 1) + "_bc");
      view.bc[2 * d + 1] = bc_code("parthenon/swarm", "ox" + std::to_string(d + 1) + "_bc");
    }
    view.leaf_map = leaf_map.data(); view.owner = owner.data(); view.local_index = ident.data();
    view.gid = ident.data(); view.owned = nullptr;
    view.blk_xmin = xmin.data(); view.blk_xmax = xmax.data(); view.blk_dx = dx.data();
    view.blk_level = level.data(); view.blk_nbr_lev = nbr_lev.data();
    view.rho = ptrs[0].data(); view.sie = ptrs[1].data(); view.u = ptrs[2].data();
    view.fleck = ptrs[3].data(); view.tally = ptrs[4].data(); view.edelta = ptrs[5].data();
    view.src_ew = ptrs[6].data(); view.src_num = ptrs[7].data();
    view.P1 = p.use_ddmc ? ptrs[8].data() : nullptr;
    view.P2 = p.use_ddmc ? ptrs[9].data() : nullptr;
    view.P3 = p.use_ddmc ? ptrs[10].data() : nullptr;
    DeviceArray<int32_t> prefix;
    prefix.Alloc(nb * ncell);
    Swarm pool;
    jb::MeshData md(pkg, view, prefix.d, [&pool](jb_swarm_view &v, int64_t need) { pool.Grow(v, need); });
    md.swarm.n = 0;
    pool.Grow(md.swarm, (int64_t)(0.65 * (double)p.num_particles) + 2048);

    // ---- ghost-zone source map of internal_energy (only needed with feedback) -----------------
    DeviceArray<int32_t> g_dst_blk, g_dst_cell, g_src_blk, g_src_cell;
    int64_t g_n = 0;
    const int nsamples = 1 << ndim;
    if (p.do_feedback) {
      std::vector<int32_t> dblk, dcell, sblk, scell;
      for (int b = 0; b < nb; ++b)
        for (int k = 0; k < ntot_dim[2]; ++k)
          for (int j = 0; j < ntot_dim[1]; ++j)
            for (int i = 0; i < ntot_dim[0]; ++i) {
              const int idx[3] = {i, j, k};
              bool ghost = false;
              for (int d = 0; d < ndim; ++d) ghost = ghost || idx[d] < is[d] || idx[d] >= is[d] + nx[d];
              if (!ghost) continue;
              dblk.push_back(b);
              dcell.push_back((k * ntot_dim[1] + j) * ntot_dim[0] + i);
              double base[3];
              for (int d = 0; d < 3; ++d)
                base[d] = (xmin[3 * b + d] - is[d] * dx[3 * b + d]) + (idx[d] + 0.5) * dx[3 * b + d];
              for (int q = 0; q < nsamples; ++q) {  // dimension 0 varies slowest (itertools.product)
                double pt[3];
                int bit = ndim - 1;
                for (int d = 0; d < 3; ++d) {
                  double off = 0.0;
                  if (d < ndim) off = ((q >> (bit--)) & 1) ? 0.25 : -0.25;
                  pt[d] = base[d] + off * dx[3 * b + d];
                }
                for (int d = 0; d < ndim; ++d) {
                  const double ext = gmax[d] - gmin[d];
                  if (mesh_periodic[2 * d]) { if (pt[d] < gmin[d]) pt[d] += ext; }
                  else pt[d] = std::max(pt[d], gmin[d] + 0.25 * dx[3 * b + d]);
                  if (mesh_periodic[2 * d + 1]) { if (pt[d] > gmax[d]) pt[d] -= ext; }
                  else pt[d] = std::min(pt[d], gmax[d] - 0.25 * dx[3 * b + d]);
                }
                const int nbk = find_block(pt);
                int c[3] = {0, 0, 0};
                for (int d = 0; d < ndim; ++d) {
                  int cd = (int)std::floor((pt[d] - xmin[3 * nbk + d]) / dx[3 * nbk + d]);
                  c[d] = std::min(std::max(cd, 0), nx[d] - 1) + is[d];
                }
                sblk.push_back(nbk);
                scell.push_back((c[2] * ntot_dim[1] + c[1]) * ntot_dim[0] + c[0]);
              }
            }
      g_n = (int64_t)dblk.size();
      g_dst_blk.Alloc(dblk.size()); g_dst_blk.Upload(dblk);
      g_dst_cell.Alloc(dcell.size()); g_dst_cell.Upload(dcell);
      g_src_blk.Alloc(sblk.size()); g_src_blk.Upload(sblk);
      g_src_cell.Alloc(scell.size()); g_src_cell.Upload(scell);
    }

    // (as the Python driver: the swarm sorted by block and cell after every k-th cycle; 0 = never)
    md.defrag_interval = (int)pin.GetOrAddInteger("jaybenne", "defrag_interval", -1);
    jb::InitializeRadiation(&md, initial_radiation == "thermal");
    std::printf("problem %s: %d-D, %d meshblocks, %d level(s), %lld photons\n", problem_id.c_str(), ndim,
                nb, max_level + 1, (long long)md.swarm.n);

    // ---- McblockDriver::Execute (mcblock_driver.cpp:38-74) --------------------------------------
    const double tlim = pin.GetReal("parthenon/time", "tlim");
    const long nlim = pin.GetOrAddInteger("parthenon/time", "nlim", -1);
    double time = 0.0;
    long ncycle = 0;
    while (time < tlim && (nlim < 0 || ncycle < nlim)) {
      const double dt = jb::EstimateTimestepMesh(&md);
      if (jb::RadiationStep(&md, time, dt) != jb::TaskStatus::complete)
        throw std::runtime_error("radiation step did not complete");
      if (p.do_feedback) {  // HostUpdateTasks: boundary exchange of u, FillDerived -> UpdateDerived
        jb::Check(jb_fill_cells(md.ctx(), md.mesh(), JB_FIELD_U, g_n, nsamples, g_dst_blk.d, g_dst_cell.d,
                                g_src_blk.d, g_src_cell.d, nullptr));
        jb::Check(jb_synchronize(md.ctx()));
        hipLaunchKernelGGL(UpdateDerivedKernel, dim3(1024), dim3(256), 0, 0, fields[0].d, fields[2].d,
                           fields[1].d, (size_t)nb * ntot);
        HIP_OK(hipDeviceSynchronize());
      }
      time += dt;
      ++ncycle;
      std::printf("cycle=%ld time=%.6e dt=%.6e photons=%lld events=%lld\n", ncycle, time, dt,
                  (long long)md.swarm.n, (long long)md.events);
    }
    HIP_OK(hipDeviceSynchronize());

    // ---- analytic comparison (tst/regression_test.py:361-412, tst/stepdiff.py:33-46) ----------
    const std::vector<double> tally = fields[4].Download(nb * ntot);
    const double tau = 1.000692e-7, ur0 = 7.5646e5, shift = 0.5;
    const double ur_eq = 4.0 * opacity.sb / opacity.c * tt0 * tt0 * tt0 * tt0;
    double sum_err = 0, max_err = 0, sum_frac = 0, max_frac = 0, sum_wfrac = 0, sum_sol = 0;
    size_t count = 0;
    std::vector<double> interior;
    for (int b = 0; b < nb; ++b)
      for (int k = is[2]; k < is[2] + nx[2]; ++k)
        for (int j = is[1]; j < is[1] + nx[1]; ++j)
          for (int i = is[0]; i < is[0] + nx[0]; ++i) {
            const double val = tally[(size_t)b * ntot + ((size_t)k * ntot_dim[1] + j) * ntot_dim[0] + i];
            interior.push_back(val);
            const double x = (xmin[3 * b] - is[0] * dx[3 * b]) + (i + 0.5) * dx[3 * b];
            double sol = ur_eq;
            if (problem_id == "stepdiff") {
              const double s = 2.0 * std::sqrt(time / tau);
              sol = ur0 / 2.0 * (std::erf(((x + shift) + 0.5) / s) - std::erf(((x + shift) - 0.5) / s));
            }
            const double err = std::fabs(sol - val), frac = err / std::fabs((sol + val) / 2.0);
            sum_err += err; max_err = std::max(max_err, err);
            sum_frac += frac; max_frac = std::max(max_frac, frac);
            sum_wfrac += frac * sol; sum_sol += sol;
            ++count;
          }
    std::printf("Mean error:                     %.2e\n", sum_err / count);
    std::printf("Mean fractional error:          %.2e\n", sum_frac / count);
    std::printf("Mean weighted fractional error: %.2e\n", sum_wfrac / sum_sol);
    std::printf("Max error:                      %.2e\n", max_err);
    std::printf("Max fractional error:           %.2e\n", max_frac);

    if (!dump.empty()) {  // interior tally (block-major, k, j, i), then particle ids and x
      const int64_t n = md.swarm.n;
      const std::vector<uint64_t> ids = pool.u64[0].Download((size_t)n);
      const std::vector<double> xs = pool.f64[0].Download((size_t)n);
      std::ofstream out(dump, std::ios::binary);
      const int64_t hdr[3] = {(int64_t)interior.size(), n, md.events};
      out.write((const char *)hdr, sizeof hdr);
      out.write((const char *)interior.data(), interior.size() * sizeof(double));
      out.write((const char *)ids.data(), ids.size() * sizeof(uint64_t));
      out.write((const char *)xs.data(), xs.size() * sizeof(double));
    }
    if (tolerance < 0) return 0;
    const bool ok = sum_wfrac / sum_sol <= tolerance;
    std::printf(ok ? "TEST PASSED\n" : "TEST FAILED\n");
    return ok ? 0 : 1;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "mcblock_amd: %s\n", e.what());
    return 2;
  }
}
