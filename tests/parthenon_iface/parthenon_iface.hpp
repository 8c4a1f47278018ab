// parthenon_iface.hpp -- DECLARED INTERFACE ONLY: the part of Parthenon's / Kokkos' public API that
// adapters/parthenon/jaybenne_amd_tasks.{hpp,cpp} touches, written from the call sites in the reference
// (src/jaybenne/*.cpp, src/mcblock/*.cpp) with small HOST-ONLY bodies, so that
//   (1) the adapter goes through a compiler (`make -C adapters/parthenon check`: syntax + types), and
//   (2) its host-side logic -- the MeshData -> jb_mesh_view builder, the source plan, the halo-refresh
//       index lists, the task graph -- can be driven on a CPU against a recording stand-in for the C ABI
//       (tests/parthenon_adapter_test.cpp) and compared with what the Python host builds for the same mesh.
// This is NOT Parthenon and proves nothing about Parthenon's behaviour: a syntax / host-logic check of OUR
// adapter, never parity evidence.  Where Parthenon itself is available the adapter is compiled against
// the real headers instead (adapters/parthenon/README.md).
#ifndef PARTHENON_IFACE_HPP_
#define PARTHENON_IFACE_HPP_

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

extern "C" int hipGetDevice(int *device);   // (Kokkos' HIP backend brings the runtime in; the check has a stub)

#define KOKKOS_LAMBDA [=]
#define KOKKOS_INLINE_FUNCTION inline
#define PARTHENON_REQUIRE(cond, msg) \
  do { if (!(cond)) throw std::runtime_error(std::string("PARTHENON_REQUIRE: ") + (msg)); } while (0);
#define PARTHENON_FAIL(msg) throw std::runtime_error(std::string("PARTHENON_FAIL: ") + (msg))
#define PARTHENON_MPI_CHECK(call) do { if ((call) != 0) throw std::runtime_error("MPI call failed"); } while (0)
#define MPI_PARTHENON_REAL MPI_DOUBLE
#define DEFAULT_LOOP_PATTERN 0

namespace Kokkos {
struct HostSpace {};
// a one-dimensional view: shared host storage (the check never runs on a device)
template <class T>
class View1 {
 public:
  View1() = default;
  View1(const std::string &, size_t n) : d_(std::make_shared<std::vector<T>>(n)), n_(n) {}
  T *data() const { return d_ ? d_->data() + off_ : nullptr; }
  size_t size() const { return n_; }
  T &operator()(size_t i) const { return (*d_)[off_ + i]; }
  View1 sub(size_t a, size_t b) const { View1 v = *this; v.off_ = off_ + a; v.n_ = b - a; return v; }
 private:
  std::shared_ptr<std::vector<T>> d_;
  size_t off_ = 0, n_ = 0;
};
template <class T> View1<T> create_mirror_view(const View1<T> &v) { return v; }
template <class S, class T> View1<T> create_mirror_view_and_copy(const S &, const View1<T> &v) { return v; }
template <class T> void deep_copy(const View1<T> &dst, const View1<T> &src) {
  for (size_t i = 0; i < src.size() && i < dst.size(); ++i) dst(i) = src(i);
}
template <class T, class I> View1<T> subview(const View1<T> &v, std::pair<I, I> r) { return v.sub((size_t)r.first, (size_t)r.second); }
}  // namespace Kokkos

namespace parthenon {
using Real = double;
template <class T> using ParArray1D = Kokkos::View1<T>;
struct DevExecSpace {};
template <class F> void par_for(int, const char *, DevExecSpace, int lo, int hi, const F &f) { for (int q = lo; q <= hi; ++q) f(q); }

enum class TaskStatus { complete, incomplete, iterate };
enum class IndexDomain { interior, entire };
enum CoordinateDirection { NODIR = 0, X1DIR = 1, X2DIR = 2, X3DIR = 3 };
enum class BoundaryFlag { periodic, outflow, reflect, user };
enum class TopologicalElement { CC, F1, F2, F3 };
struct IndexRange { int s = 0, e = 0; };

namespace Globals { extern int my_rank, nranks, nghost; }

class ParameterInput {
 public:
  std::map<std::string, std::string> kv;
  int GetInteger(const std::string &b, const std::string &k) { return std::atoi(need(b, k).c_str()); }
  Real GetOrAddReal(const std::string &b, const std::string &k, Real d) { return has(b, k) ? std::atof(kv[b + "/" + k].c_str()) : d; }
  int GetOrAddInteger(const std::string &b, const std::string &k, int d) { return has(b, k) ? std::atoi(kv[b + "/" + k].c_str()) : d; }
  bool GetOrAddBoolean(const std::string &b, const std::string &k, bool d) { return has(b, k) ? kv[b + "/" + k] == "true" : d; }
  std::string GetOrAddString(const std::string &b, const std::string &k, const std::string &d) { return has(b, k) ? kv[b + "/" + k] : d; }
 private:
  bool has(const std::string &b, const std::string &k) const { return kv.count(b + "/" + k) != 0; }
  const std::string &need(const std::string &b, const std::string &k) {
    if (!has(b, k)) throw std::runtime_error("missing parameter " + b + "/" + k);
    return kv[b + "/" + k];
  }
};

struct MetadataFlag { int v; };
class Metadata {
 public:
  static constexpr MetadataFlag Provides{0}, None{1}, Real{2}, Integer{3}, Cell{4}, Face{5}, Independent{6}, OneCopy{7},
      Derived{8}, FillGhost{9};
  Metadata() = default;
  Metadata(std::initializer_list<MetadataFlag>) {}
  Metadata(const std::vector<MetadataFlag> &, const std::vector<int> &) {}
};

class MeshData_;
class StateDescriptor {
 public:
  explicit StateDescriptor(const std::string &n) : name(n) {}
  std::string name;
  template <class T> void AddParam(const std::string &k, T v) { params_[k] = std::make_shared<Holder<T>>(std::move(v)); }
  template <class T> const T &Param(const std::string &k) const {
    auto it = params_.find(k);
    if (it == params_.end()) throw std::runtime_error("no parameter " + k);
    auto *h = dynamic_cast<Holder<T> *>(it->second.get());
    if (!h) throw std::runtime_error("parameter " + k + " has another type");
    return h->v;
  }
  void AddSwarm(const std::string &n, const Metadata &) { swarms.push_back(n); }
  void AddSwarmValue(const std::string &n, const std::string &, const Metadata &) { swarm_values.push_back(n); }
  void AddField(const std::string &n, const Metadata &) { fields.push_back(n); }
  std::vector<std::string> swarms, swarm_values, fields;
  std::function<Real(MeshData_ *)> EstimateTimestepMeshAny;
  // (assigned a function pointer Real(*)(MeshData<Real>*) by the package)
  struct TimestepSlot {
    void *fn = nullptr;
    template <class F> TimestepSlot &operator=(F f) { fn = (void *)f; return *this; }
  } EstimateTimestepMesh;
 private:
  struct Base { virtual ~Base() = default; };
  template <class T> struct Holder : Base { explicit Holder(T x) : v(std::move(x)) {} T v; };
  std::map<std::string, std::shared_ptr<Base>> params_;
};
class Packages_t {
 public:
  void Add(const std::shared_ptr<StateDescriptor> &p) { p_[p->name] = p; }
  std::shared_ptr<StateDescriptor> &Get(const std::string &n) {
    auto it = p_.find(n);
    if (it == p_.end()) throw std::runtime_error("no package " + n);
    return it->second;
  }
 private:
  std::map<std::string, std::shared_ptr<StateDescriptor>> p_;
};

struct LogicalLocation {
  int lev = 0;
  std::int64_t l[3] = {0, 0, 0};
  int level() const { return lev; }
  std::int64_t lx1() const { return l[0]; }
  std::int64_t lx2() const { return l[1]; }
  std::int64_t lx3() const { return l[2]; }
};
struct RegionSize {
  Real lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  Real xmin(CoordinateDirection d) const { return lo[d - 1]; }
  Real xmax(CoordinateDirection d) const { return hi[d - 1]; }
};
struct Coordinates_t {
  Real dx[3] = {1, 1, 1};
  Real Dxc(CoordinateDirection d) const { return dx[d - 1]; }
};

class Mesh;
template <class T> class MeshBlockData;
class SwarmContainer;
class MeshBlock {
 public:
  int gid = 0, lid = 0;
  LogicalLocation loc;
  Coordinates_t coords;
  RegionSize block_size;
  Mesh *pmy_mesh = nullptr;
  // every cell field of the block, [nk][nj][ni] with ghosts, by variable name (face fields: name + "/F1" ...)
  std::map<std::string, std::vector<Real>> vars;
  int nbr_level[6] = {0, 0, 0, 0, 0, 0};   // level behind each face, in the adapter's (ok, oj, oi) probe order
  bool phys_bdry[6] = {false, false, false, false, false, false};
};

// ---- variable packs: MakePackDescriptor<vars...>(resolved packages).GetPack(md)(b, var, k, j, i) ------
template <class T> class MeshData;
class SparsePack {
 public:
  MeshData<Real> *md = nullptr;
  int ni = 1, nj = 1;
  template <class V> Real &operator()(int b, const V &, int k, int j, int i) const { return at(b, V::name(), k, j, i); }
  template <class V> Real &operator()(int b, TopologicalElement te, const V &, int k, int j, int i) const {
    return at(b, V::name() + (te == TopologicalElement::F1 ? "/F1" : te == TopologicalElement::F2 ? "/F2" : "/F3"), k, j, i);
  }
  bool IsPhysicalBoundary(int b, int ok, int oj, int oi) const;
  int GetLevel(int b, int ok, int oj, int oi) const;
 private:
  Real &at(int b, const std::string &name, int k, int j, int i) const;
};
template <class... Vs>
struct PackDescriptor {
  SparsePack GetPack(MeshData<Real> *md) const;
};
template <class... Vs, class R> PackDescriptor<Vs...> MakePackDescriptor(R *) { return {}; }

template <class T>
class MeshBlockData {
 public:
  std::shared_ptr<MeshBlock> pmb;
  MeshBlock *GetBlockPointer() const { return pmb.get(); }
  Mesh *GetParentPointer() const { return pmb->pmy_mesh; }
};

// ---- swarms (only what ExportToParthenonSwarm touches) -----------------------------------------------
struct NewParticlesContext { int GetNewParticleIndex(int q) const { return q; } };
template <class T> struct ParticleVariable {
  struct Arr {
    std::shared_ptr<std::vector<T>> d = std::make_shared<std::vector<T>>();
    int n1 = 1;
    T &operator()(int n) const { if ((size_t)n >= d->size()) d->resize(n + 1); return (*d)[n]; }
    T &operator()(int c, int n) const { const size_t q = (size_t)c + (size_t)3 * n; if (q >= d->size()) d->resize(q + 1); return (*d)[q]; }
  } a;
  Arr &Get() { return a; }
};
class Swarm {
 public:
  void RemoveMarkedParticles() {}
  NewParticlesContext AddEmptyParticles(size_t) { return {}; }
  template <class T> ParticleVariable<T> &Get(const std::string &n) { return vars_[n]; }
 private:
  std::map<std::string, ParticleVariable<Real>> vars_;
};
class SwarmContainer {
 public:
  std::shared_ptr<Swarm> Get(const std::string &) { if (!s_) s_ = std::make_shared<Swarm>(); return s_; }
 private:
  std::shared_ptr<Swarm> s_;
};
namespace swarm_position {
struct x { static std::string name() { return "x"; } };
struct y { static std::string name() { return "y"; } };
struct z { static std::string name() { return "z"; } };
}  // namespace swarm_position

template <class T>
class MeshData {
 public:
  Mesh *pmesh = nullptr;
  std::vector<std::shared_ptr<MeshBlockData<T>>> blocks;
  IndexRange ib, jb, kb;
  Mesh *GetParentPointer() const { return pmesh; }
  int NumBlocks() const { return (int)blocks.size(); }
  IndexRange GetBoundsI(IndexDomain) const { return ib; }
  IndexRange GetBoundsJ(IndexDomain) const { return jb; }
  IndexRange GetBoundsK(IndexDomain) const { return kb; }
  const std::shared_ptr<MeshBlockData<T>> &GetBlockData(int b) const { return blocks[(size_t)b]; }
  std::shared_ptr<SwarmContainer> GetSwarmData(int b) {
    if (swarms_.size() < blocks.size()) swarms_.resize(blocks.size());
    if (!swarms_[(size_t)b]) swarms_[(size_t)b] = std::make_shared<SwarmContainer>();
    return swarms_[(size_t)b];
  }
 private:
  std::vector<std::shared_ptr<SwarmContainer>> swarms_;
};

template <class T>
class DataCollection {
 public:
  std::shared_ptr<T> &Get() { return GetOrAdd("base", 0); }
  std::shared_ptr<T> &GetOrAdd(const std::string &n, int part) { return d_[n + "#" + std::to_string(part)]; }
 private:
  std::map<std::string, std::shared_ptr<T>> d_;
};

class Mesh {
 public:
  Packages_t packages;
  std::shared_ptr<Packages_t> resolved_packages = std::make_shared<Packages_t>();
  int nbtotal = 0, ndim = 1;
  RegionSize mesh_size;
  std::array<int, 3> nrbx{{1, 1, 1}};
  BoundaryFlag mesh_bcs[6] = {BoundaryFlag::outflow, BoundaryFlag::outflow, BoundaryFlag::periodic,
                              BoundaryFlag::periodic, BoundaryFlag::periodic, BoundaryFlag::periodic};
  std::string mesh_swarm_bc_names[6];
  DataCollection<MeshData<Real>> mesh_data;
  int GetCurrentLevel() const { return root_level + max_level; }
  int GetRootLevel() const { return root_level; }
  const std::vector<LogicalLocation> &GetLocList() const { return locs; }
  const std::vector<int> &GetRankList() const { return ranks; }
  int DefaultNumPartitions() const { return 1; }
  std::vector<LogicalLocation> locs;
  std::vector<int> ranks;
  int root_level = 0, max_level = 0;
};

inline SparsePack PackDescriptorGet(MeshData<Real> *md) {
  SparsePack p;
  p.md = md;
  const int ng = Globals::nghost;
  p.ni = md->ib.e - md->ib.s + 1 + 2 * ng;
  p.nj = md->pmesh->ndim > 1 ? md->jb.e - md->jb.s + 1 + 2 * ng : 1;
  return p;
}
template <class... Vs> SparsePack PackDescriptor<Vs...>::GetPack(MeshData<Real> *md) const { return PackDescriptorGet(md); }
inline Real &SparsePack::at(int b, const std::string &name, int k, int j, int i) const {
  auto &v = md->blocks[(size_t)b]->pmb->vars[name];
  if (v.empty()) throw std::runtime_error("block has no variable " + name);
  return v[((size_t)k * nj + j) * ni + i];
}
inline int face_of_probe(int ok, int oj, int oi) {   // (ok, oj, oi) with one of them +-1 -> x-, x+, y-, y+, z-, z+
  return oi < 0 ? 0 : oi > 0 ? 1 : oj < 0 ? 2 : oj > 0 ? 3 : ok < 0 ? 4 : 5;
}
inline bool SparsePack::IsPhysicalBoundary(int b, int ok, int oj, int oi) const {
  return md->blocks[(size_t)b]->pmb->phys_bdry[face_of_probe(ok, oj, oi)];
}
inline int SparsePack::GetLevel(int b, int ok, int oj, int oi) const {
  return md->blocks[(size_t)b]->pmb->nbr_level[face_of_probe(ok, oj, oi)];
}

// ---- tasks (jaybenne.cpp:68-151 as the call sites use them) ---------------------------------------------
struct TaskID {
  explicit TaskID(int v = 0) : id(v) {}
  int id;
};
enum class TaskQualifier : unsigned { none = 0, once_per_region = 1, global_sync = 2, completion = 4 };
inline TaskQualifier operator|(TaskQualifier a, TaskQualifier b) { return (TaskQualifier)((unsigned)a | (unsigned)b); }
using TQ = TaskQualifier;
class TaskList {
 public:
  struct Entry { int id, dep; unsigned qual; std::function<TaskStatus()> run; };
  std::vector<Entry> tasks;
  std::vector<std::unique_ptr<TaskList>> sublists;
  std::vector<std::pair<int, int>> sublist_iters;   // (min, max) iterations
  std::vector<int> sublist_dep, sublist_id;
  template <class F, class... A> TaskID AddTask(TaskID dep, F f, A... a) { return AddTask(TQ::none, dep, f, a...); }
  template <class F, class... A> TaskID AddTask(TaskQualifier q, TaskID dep, F f, A... a) {
    const int id = next_id();
    tasks.push_back({id, dep.id, (unsigned)q, [=]() { return f(a...); }});
    return TaskID(id);
  }
  std::pair<TaskList &, TaskID> AddSublist(TaskID dep, std::pair<int, int> iters) {
    sublists.push_back(std::make_unique<TaskList>());
    sublists.back()->counter_ = counter_;
    sublist_iters.push_back(iters);
    sublist_dep.push_back(dep.id);
    const int id = next_id();
    sublist_id.push_back(id);
    return {*sublists.back(), TaskID(id)};
  }
  std::shared_ptr<int> counter_ = std::make_shared<int>(0);
 private:
  int next_id() { return ++*counter_; }
};
class TaskRegion {
 public:
  explicit TaskRegion(int n) : lists_((size_t)n) {}
  TaskList &operator[](int i) { return lists_[(size_t)i]; }
  size_t size() const { return lists_.size(); }
 private:
  std::vector<TaskList> lists_;
};
class TaskCollection {
 public:
  TaskRegion &AddRegion(int n) { regions.emplace_back(n); return regions.back(); }
  std::vector<TaskRegion> regions;
};

namespace driver { namespace prelude {} }
namespace package { namespace prelude {} }
}  // namespace parthenon

#endif  // PARTHENON_IFACE_HPP_
