// tools/microbench.hip -- per-primitive issue cost of the history loop's building blocks on
// gfx950, measured with the product's own device functions (jb_rng.hpp, jb_math.hpp).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I jaybenne_amd/csrc \
//         tools/microbench.hip -o gpurun_out/microbench && gpurun_out/microbench
//
// Every kernel runs ITER dependent evaluations per lane on 256 CUs x 8 workgroups x 256 lanes and
// reports SIMD cycles per wave-level call (assuming 2.4 GHz): the number to multiply by the
// per-event call counts when budgeting the tracking kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "jb_math.hpp"
#include "jb_rng.hpp"

using namespace jb;

constexpr int ITER = 2048;

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__device__ __forceinline__ double seed_val(int i) {
  return 0.1 + 0.8 * (double)((threadIdx.x * 7 + blockIdx.x * 13 + i) % 1000) * 1e-3;
}

__global__ void k_empty(double *out) {
  double a = seed_val(0);
  for (int i = 0; i < ITER; ++i) a = a * 0.999 + 1e-3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_philox(double *out) {
  uint32_t acc = threadIdx.x;
  for (int i = 0; i < ITER; ++i) {
    const PhiloxBlock b = philox4x32_10(i, acc, threadIdx.x, blockIdx.x, 349857u, 0u);
    acc ^= b.w0 ^ b.w1 ^ b.w2 ^ b.w3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)acc;
}
__global__ void k_drand2(double *out) {  // two uniforms from the product generator
  LcgRng rng(rng_seed_state(349857u, 0u, blockIdx.x * 256ull + threadIdx.x));
  double a = 0.0;
  for (int i = 0; i < ITER; ++i) a += rng.drand() * rng.drand();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_xorshift64s(double *out) {  // Kokkos-style xorshift64* uniform
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256ull + threadIdx.x + 1);
  double a = 0.0;
  for (int i = 0; i < ITER; ++i) {
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    const uint64_t r = s * 2685821657736338717ull;
    a += ((double)(r >> 12) + 0.5) * 2.220446049250313080847263336181640625e-16;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_xorwow(double *out) {  // rocRAND default generator's recurrence, 2 words / uniform
  uint32_t x = 123456789u ^ threadIdx.x, y = 362436069u ^ blockIdx.x, z = 521288629u, w = 88675123u,
           v = 5783321u, d = 6615241u;
  double a = 0.0;
  for (int i = 0; i < ITER; ++i) {
    uint32_t r[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t t = x ^ (x >> 2);
      x = y; y = z; z = w; w = v;
      v = (v ^ (v << 4)) ^ (t ^ (t << 1));
      d += 362437u;
      r[q] = d + v;
    }
    a += u52_to_double((((uint64_t)r[1] << 32) | r[0]) >> 12);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_log(double *out) {
  load_math_tables();
  double a = seed_val(1);
  for (int i = 0; i < ITER; ++i) a = 0.5 + 0.4 * (m_log(a) * -0.3);  // stays in (0,1)
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_log_ocml(double *out) {
  double a = seed_val(1);
  for (int i = 0; i < ITER; ++i) a = 0.5 + 0.4 * (log(a) * -0.3);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_div(double *out) {
  double a = seed_val(2), b = 1.0 + seed_val(3);
  for (int i = 0; i < ITER; ++i) a = 0.7 + a / b;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_rcp_mul(double *out) {  // what a reciprocal-multiply costs instead
  double a = seed_val(2), b = 1.0 / (1.0 + seed_val(3));
  for (int i = 0; i < ITER; ++i) a = 0.7 + a * b;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_sqrt(double *out) {
  double a = seed_val(4);
  for (int i = 0; i < ITER; ++i) a = 0.3 + sqrt(a);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_sincos(double *out) {
  load_math_tables();
  double a = seed_val(5);
  for (int i = 0; i < ITER; ++i) {
    double s, c;
    m_sincos(6.2 * a, s, c);
    a = 0.5 + 0.25 * (s * c + s);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_sincos_ocml(double *out) {
  double a = seed_val(5);
  for (int i = 0; i < ITER; ++i) {
    double s, c;
    sincos(6.2 * a, &s, &c);
    a = 0.5 + 0.25 * (s * c + s);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_fma(double *out) {
  double a = seed_val(6);
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) a = fma(a, 0.999, 1e-3);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_floor_idx(double *out) {  // Xtoijk: floor((x - xmin) / dx) as written
  double a = seed_val(7);
  const double dx = 1.0 / 256.0;
  int acc = 0;
  for (int i = 0; i < ITER; ++i) {
    acc += (int)floor((a - (-0.5)) / dx);
    a = 0.1 + 0.5 * a;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc;
}

template <class K>
int run(const char *name, K kernel, double *out, double calls_per_iter, double base_ms = 0.0) {
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double wave_calls = (double)blocks * threads / 64.0 * ITER * calls_per_iter;
  const double simd_cycles = (ms - base_ms) * 1e-3 * 2.4e9 * 1024.0 / wave_calls;
  printf("%-16s %8.3f ms  %8.1f SIMD-cycles per wave call   %10.3e lane-calls/s\n", name, ms,
         simd_cycles, wave_calls * 64.0 / (ms * 1e-3));
  return 0;
}

int main() {
  double *out;
  CHECK(hipMalloc(&out, sizeof(double) * 2 * 256 * 8 * 256));
  run("loop+fma", k_empty, out, 1);
  run("fma x16", k_fma, out, 16);
  run("philox block", k_philox, out, 1);
  run("drand pair", k_drand2, out, 1);
  run("xorshift64* u", k_xorshift64s, out, 1);
  run("xorwow u", k_xorwow, out, 1);
  run("m_log", k_log, out, 1);
  run("ocml log", k_log_ocml, out, 1);
  run("f64 divide", k_div, out, 1);
  run("f64 mul (rcp)", k_rcp_mul, out, 1);
  run("f64 sqrt", k_sqrt, out, 1);
  run("m_sincos", k_sincos, out, 1);
  run("ocml sincos", k_sincos_ocml, out, 1);
  run("floor idx", k_floor_idx, out, 1);
  CHECK(hipFree(out));
  return 0;
}
