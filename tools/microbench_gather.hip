// tools/microbench_gather.hip -- what the memory path gives the all-DDMC event loop (k_ddmc_all<3>, DESIGN
// section 4.2) when the arithmetic is taken away: persistent waves, one lane = one walker, every pass gathers the
// 64-byte record of the walker's cell (the product's quad-cooperative LDS-direct form, or one lane = four
// 16-byte loads), the walker then moves to one of its six neighbour records (+-1, +-ni, +-ni nj) chosen by an
// LCG draw, FILL dependent FP64 fma per pass stand in for the step's arithmetic, a walker that has made its
// STEPS moves is replaced from its XCD's queue (cell-ordered walkers, PER_CELL per cell, as sourced).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_gather.hip -o gpurun_out/microbench_gather
//   gpurun_out/microbench_gather [walkers=100000000] [steps=34] [per_cell=40]
//
// Prints ms per launch for waves per SIMD x FILL x gather form: the ceiling the real kernel's 25 ms stand against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kQueues = 8;
constexpr long long kChunk = 128;
constexpr int kPerCell = 40;
constexpr unsigned long long kMul = 6364136223846793005ull, kInc = 1442695040888963407ull;

struct Args {
  const double *rec;        // 8 doubles per record
  const unsigned *code;     // FORM 2 / 3: one 32-bit code per cell (class id; bit 31: ghost)
  const unsigned char *code8;  // FORM 4: one byte per cell (class id; bit 7: ghost)
  unsigned long long *queue;  // kQueues cursors
  double *sink;
  long long nwalk;
  int steps, per_cell;
  int ni, nj, nk, ng;       // block shape incl. ghosts, ghost width
  int nblocks;
  int mode;                 // 0: the walk; 1: every gather within 256 records (16 KB: vector-L1 hits); 2: gathers
                            // scattered over the whole table (L2 misses)
  unsigned nrec;
};

typedef double v4d __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ unsigned quad_bcast_add(unsigned v, unsigned add) {
  constexpr int ctrl = K * 0x55;
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xf, 0xf, true) + add;
}

// FORM 0: quad-cooperative LDS-direct gather; 1: four 16-byte loads per lane (two dwordx4 pairs);
// 2: a 4-byte CELL CODE gathered per step, the (few) distinct records in LDS, addressed by the code;
// 3: the code gathered, ONE record held in scalar registers; 4: a 1-byte code, records in LDS; 5: no gather at all
template <int FORM, int FILL, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_walk(Args A) {
  __shared__ __attribute__((aligned(16))) char lds_rec[FORM == 0 ? 4 : 1][4][FORM == 0 ? 1024 : 16];
  __shared__ __attribute__((aligned(16))) double lds_cls[(FORM == 2 || FORM == 4) ? 8 * 16 : 8];
  if constexpr (FORM == 2 || FORM == 4) {
    if (threadIdx.x < 8 * 16) lds_cls[threadIdx.x] = A.rec[(size_t)8 * (size_t)(2 * A.ni * A.nj + 2 * A.ni + 2) + (threadIdx.x & 7)];
    __syncthreads();
  }
  double sr[8];
  if constexpr (FORM == 3 || FORM == 5) {
#pragma unroll
    for (int q = 0; q < 8; ++q) sr[q] = A.rec[(size_t)8 * (size_t)(2 * A.ni * A.nj + 2 * A.ni + 2) + q];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  typedef __attribute__((address_space(3))) char *lchar;
  lchar const wave_buf = (lchar)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lchar)&lds_rec[FORM == 0 ? wv : 0][0][0]);
  const v4d *my_rec = (const v4d *)&lds_rec[FORM == 0 ? wv : 0][lane & 3][FORM == 0 ? 64 * (lane >> 2) : 0];
  const unsigned sub16 = 16u * (unsigned)(lane & 3);
  const int ni = A.ni, nij = A.ni * A.nj, ntot = nij * A.nk;
  const int nx = A.ni - 2 * A.ng, ny = A.nj - 2 * A.ng, nz = A.nk - 2 * A.ng;
  const long long ncell_int = (long long)nx * ny * nz;
  const long long per_q = (A.nwalk + kQueues - 1) / kQueues;
  int cur = blockIdx.x % kQueues, tried = 0;
  unsigned rec = 0u;
  unsigned long long s = 0ull;
  int left = 0, last = 0;
  double acc = 0.0;
  bool more = true;
  long long chunk_pos = 0, chunk_end = 0;
  for (;;) {
    // refill the lanes whose walker is done: from the wave's chunk of its XCD's queue (kChunk slots per atomic)
    const bool need = left == 0;
    const unsigned long long nm = __ballot(need);
    if (nm != 0ull && (more || chunk_pos < chunk_end)) {
      if (chunk_pos >= chunk_end) {
        for (;;) {
          unsigned long long base = 0ull;
          if (lane == 0) base = atomicAdd(&A.queue[cur], (unsigned long long)kChunk);
          base = __shfl(base, 0, 64);
          if ((long long)base < per_q) {
            chunk_pos = (long long)cur * per_q + (long long)base;
            const long long qe = (long long)(cur + 1) * per_q;
            chunk_end = chunk_pos + kChunk < qe ? chunk_pos + kChunk : qe;
            if (chunk_end > A.nwalk) chunk_end = A.nwalk;
            break;
          }
          if (++tried >= kQueues) { more = false; break; }
          cur = (cur + 1) % kQueues;
        }
      }
      const long long w = chunk_pos + __popcll(nm & ((1ull << lane) - 1ull));
      if (need && w < chunk_end) {
        // (cheap on purpose: 32 x 32 x 32 interior cells, 64 blocks, kPerCell walkers per cell, ~34 moves)
        const unsigned c = (unsigned)w / (unsigned)kPerCell;
        const unsigned b = (c >> 15) & 63u;
        const unsigned i = c & 31u, j = (c >> 5) & 31u, k = (c >> 10) & 31u;
        rec = b * (unsigned)ntot + ((k + (unsigned)A.ng) * (unsigned)A.nj + (j + (unsigned)A.ng)) * (unsigned)ni + (i + (unsigned)A.ng);
        s = (unsigned long long)w * 0x9E3779B97F4A7C15ull + 12345ull;
        s = s * kMul + kInc;
        left = 1 + (int)((unsigned)(s >> 40) % 67u);
      }
      const long long took = chunk_end - chunk_pos < (long long)__popcll(nm) ? chunk_end - chunk_pos : (long long)__popcll(nm);
      chunk_pos += took > 0 ? took : 0;
    }
    if (__ballot(left > 0) == 0ull) break;
    const bool run = left > 0;
    unsigned rq = run ? rec : 0u;
    if (A.mode == 1) rq &= 255u;
    if (A.mode == 2) rq = (unsigned)(((unsigned long long)(rq * 2654435761u) * A.nrec) >> 32);
    v4d r0, r1;
    if constexpr (FORM == 0) {
      typedef const __attribute__((address_space(1))) void *gvoid;
      const unsigned off = rq << 6;
      const char *base = (const char *)A.rec;
      __builtin_amdgcn_global_load_lds((gvoid)(base + quad_bcast_add<0>(off, sub16)), wave_buf, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid)(base + quad_bcast_add<1>(off, sub16)), wave_buf + 1024, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid)(base + quad_bcast_add<2>(off, sub16)), wave_buf + 2048, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gvoid)(base + quad_bcast_add<3>(off, sub16)), wave_buf + 3072, 16, 0, 0);
      s = s * kMul + kInc;
      __builtin_amdgcn_s_waitcnt(0x0f70);
      r0 = my_rec[0];
      r1 = my_rec[1];
    } else if constexpr (FORM == 1) {
      const v4d *rp = (const v4d *)(A.rec + 8ull * rq);
      r0 = rp[0];
      r1 = rp[1];
      s = s * kMul + kInc;
    } else if constexpr (FORM == 2 || FORM == 4) {
      unsigned code;
      if constexpr (FORM == 2) code = A.code[rq];
      else { const unsigned c8 = A.code8[rq]; code = (c8 & 0x7fu) | ((c8 & 0x80u) << 24); }
      s = s * kMul + kInc;
      const v4d *rp = (const v4d *)(lds_cls + 8u * (code & 15u));
      r0 = rp[0];
      r1 = rp[1];
      if ((int)code < 0) r1.w = -1.0;
    } else {
      unsigned code = 0u;
      if constexpr (FORM == 3) code = A.code[rq];
      else code = (rq * 2654435761u) < 40000000u ? 0x80000000u : 0u;   // (about 1 % "ghosts", from arithmetic)
      s = s * kMul + kInc;
      r0.x = sr[0]; r0.y = sr[1]; r0.z = sr[2]; r0.w = sr[3];
      r1.x = sr[4]; r1.y = sr[5]; r1.z = sr[6]; r1.w = (int)code < 0 ? -1.0 : sr[7];
    }
    double u = (double)(s >> 11) * 0x1.0p-53;
    double f = r0.x + r1.w;
#pragma unroll
    for (int q = 0; q < FILL; ++q) f = __builtin_fma(f, 0.999999, u);
    // the channel walk: six thresholds (here 1/6 each, read from the record so that the record is used)
    const double xim = u * r1.z;
    int delta = nij;
    delta = (xim < r1.y) ? -nij : delta;
    delta = (xim < r1.x) ? ni : delta;
    delta = (xim < r0.w) ? -ni : delta;
    delta = (xim < r0.z) ? 1 : delta;
    delta = (xim < r0.y) ? -1 : delta;
    if (run) {
      // stay on interior cells: a ghost record carries a negative last word (as the product's ghost codes do);
      // a walker that finds itself on one goes back where it came from, the pass is not a move
      const bool ghost = A.mode == 0 && r1.w < 0.0;
      rec = ghost ? rec - (unsigned)last : rec + (unsigned)delta;
      if (A.mode != 0) rec = rec % A.nrec;   // (no ghost reflection in these modes: stay inside the table)
      last = delta;
      acc += f;
      left -= ghost ? 0 : 1;
    }
  }
  if (acc == 1.2345) A.sink[0] = acc;
}

template <int FORM, int FILL, int WAVES>
static float run(const Args &A0, int reps) {
  Args A = A0;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    HIP_OK(hipMemset(A.queue, 0, kQueues * 8));
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_walk<FORM, FILL, WAVES>), dim3(256 * WAVES), dim3(256), 0, 0, A);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  Args A{};
  A.nwalk = argc > 1 ? std::atoll(argv[1]) : 100000000ll;
  A.steps = argc > 2 ? std::atoi(argv[2]) : 34;
  A.per_cell = argc > 3 ? std::atoi(argv[3]) : 40;
  A.ng = 2; A.ni = A.nj = A.nk = 32 + 2 * A.ng;
  A.nblocks = 64;
  const size_t nrec = (size_t)A.nblocks * A.ni * A.nj * A.nk;
  std::vector<double> h(8 * nrec);
  for (size_t q = 0; q < nrec; ++q) {
    double *r = &h[8 * q];
    r[0] = 0.5; r[1] = 1.0 / 6; r[2] = 2.0 / 6; r[3] = 3.0 / 6; r[4] = 4.0 / 6; r[5] = 5.0 / 6; r[6] = 1.0; r[7] = 0.25;
    const size_t c = q % ((size_t)A.ni * A.nj * A.nk);
    const int i = (int)(c % A.ni), j = (int)((c / A.ni) % A.nj), k = (int)(c / ((size_t)A.ni * A.nj));
    if (i < A.ng || i >= A.ni - A.ng || j < A.ng || j >= A.nj - A.ng || k < A.ng || k >= A.nk - A.ng) r[7] = -1.0;
  }
  std::vector<unsigned> hc(nrec);
  std::vector<unsigned char> hc8(nrec);
  for (size_t q = 0; q < nrec; ++q) { hc[q] = h[8 * q + 7] < 0.0 ? 0x80000000u : 0u; hc8[q] = h[8 * q + 7] < 0.0 ? 0x80u : 0u; }
  unsigned *code_d; unsigned char *code8_d;
  HIP_OK(hipMalloc(&code_d, nrec * 4)); HIP_OK(hipMemcpy(code_d, hc.data(), nrec * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&code8_d, nrec)); HIP_OK(hipMemcpy(code8_d, hc8.data(), nrec, hipMemcpyHostToDevice));
  double *rec_d, *sink;
  unsigned long long *queue;
  HIP_OK(hipMalloc(&rec_d, h.size() * 8));
  HIP_OK(hipMemcpy(rec_d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&sink, 8));
  HIP_OK(hipMalloc(&queue, kQueues * 8));
  A.rec = rec_d; A.sink = sink; A.queue = queue; A.code = code_d; A.code8 = code8_d;
  std::printf("%lld walkers, ~%d moves each, %d per cell, %zu records of 64 B (%.0f MB)\n", A.nwalk, A.steps, A.per_cell,
              nrec, nrec * 64 / 1e6);
  std::printf("form  fill  waves/SIMD   ms\n");
#define RUN(F, L, W) std::printf("%4d  %4d  %10d  %6.2f\n", F, L, W, run<F, L, W>(A, 3)); std::fflush(stdout)
  A.nrec = (unsigned)nrec;
  for (int mode = 0; mode < 3; ++mode) {
    A.mode = mode;
    std::printf("mode %d (%s)\n", mode, mode == 0 ? "the walk" : mode == 1 ? "gathers within 16 KB" : "gathers scattered over the table");
    RUN(0, 0, 4); RUN(0, 0, 8); RUN(0, 100, 4); RUN(1, 0, 4);
    RUN(2, 0, 4); RUN(2, 0, 8); RUN(2, 60, 4); RUN(2, 100, 4); RUN(3, 0, 4); RUN(3, 100, 4); RUN(4, 0, 4); RUN(4, 100, 4);
    if (mode == 0) { RUN(5, 0, 4); RUN(5, 100, 4); RUN(2, 100, 5); RUN(2, 100, 6); RUN(2, 100, 8); RUN(3, 100, 8); }
  }
  A.mode = 0;
  RUN(0, 0, 2); RUN(0, 0, 3); RUN(0, 0, 6); RUN(0, 60, 4); RUN(0, 100, 2); RUN(0, 100, 3); RUN(0, 100, 6); RUN(0, 100, 8);
  RUN(1, 0, 8); RUN(1, 100, 4);
  return 0;
}
