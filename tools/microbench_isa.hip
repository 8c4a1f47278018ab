// tools/microbench_isa.hip -- issue cost of single gfx950 vector instructions, as the history loop
// uses them: SIMD cycles per wave-level instruction with 1, 2 and 3 waves per SIMD.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_isa.hip -o gpurun_out/microbench_isa
//   gpurun_out/microbench_isa > gpurun_out/microbench_isa.txt
//
// Each kernel runs ITER iterations of 16 INDEPENDENT copies of one instruction (inline asm, so the
// compiler neither removes nor fuses them) in every wave of a workgroup of W waves per SIMD, on every
// CU, and reads s_memtime around the loop.  Reported: (cycles of the slowest wave) / (ITER * 16 * W)
// = cycles the SIMD spends per wave-instruction when W waves compete for it; with W = 1 the figure
// also contains the instruction's dependent-issue spacing, with W = 3 it is the throughput cost the
// tracking kernels pay.  The shader clock behind s_memtime is 100 MHz on this part (constant), so
// the time base is wall-clock microseconds of the slowest wave x the measured core clock.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e = (x);                                                                   \
    if (e != hipSuccess) {                                                                \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);        \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

constexpr int ITER = 4096;

// 16 independent instances per iteration: destination registers cycle through 16 pairs
#define REP16(S)                                                                                   \
  S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define KERNEL_D(name, ASM)                                                                        \
  __global__ void __launch_bounds__(1024) name(double *out, double a, double b) {                  \
    double r[16];                                                                                  \
    for (int i = 0; i < 16; ++i) r[i] = a + (double)(threadIdx.x + i) * 1e-3;                      \
    double x = b + (double)threadIdx.x * 1e-4, y = a * 0.5 + 1.0;                                  \
    for (int it = 0; it < ITER; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                              \
    double s = 0.0;                                                                                \
    for (int i = 0; i < 16; ++i) s += r[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
  }
#define KERNEL_I(name, ASM)                                                                        \
  __global__ void __launch_bounds__(1024) name(double *out, double a, double b) {                  \
    unsigned r[16];                                                                                \
    for (int i = 0; i < 16; ++i) r[i] = (unsigned)(threadIdx.x * 7 + i * 13 + (int)a);             \
    unsigned x = (unsigned)(b * 1e6) + threadIdx.x, y = 0x5851f42du;                               \
    for (int it = 0; it < ITER; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                              \
    unsigned s = 0;                                                                                \
    for (int i = 0; i < 16; ++i) s += r[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;                                        \
  }
// 64-bit integer destination
#define KERNEL_L(name, ASM)                                                                        \
  __global__ void __launch_bounds__(1024) name(double *out, double a, double b) {                  \
    unsigned long long r[16];                                                                      \
    for (int i = 0; i < 16; ++i) r[i] = (unsigned long long)(threadIdx.x * 7 + i * 13 + (int)a);   \
    unsigned x = (unsigned)(b * 1e6) + threadIdx.x, y = 0x5851f42du;                               \
    for (int it = 0; it < ITER; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                              \
    unsigned long long s = 0;                                                                      \
    for (int i = 0; i < 16; ++i) s += r[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;                                        \
  }
// f64 destination from an integer source / integer destination from an f64 source
#define KERNEL_DI(name, ASM)                                                                       \
  __global__ void __launch_bounds__(1024) name(double *out, double a, double b) {                  \
    double r[16];                                                                                  \
    for (int i = 0; i < 16; ++i) r[i] = a;                                                         \
    int x = (int)(b * 1e6) + threadIdx.x;                                                          \
    double y = a;                                                                                  \
    for (int it = 0; it < ITER; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                              \
    double s = 0.0;                                                                                \
    for (int i = 0; i < 16; ++i) s += r[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
  }
#define KERNEL_ID(name, ASM)                                                                       \
  __global__ void __launch_bounds__(1024) name(double *out, double a, double b) {                  \
    int r[16];                                                                                     \
    for (int i = 0; i < 16; ++i) r[i] = (int)a;                                                    \
    double x = b * 1e3 + (double)threadIdx.x, y = a;                                               \
    for (int it = 0; it < ITER; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x), "v"(y)); \
    }                                                                                              \
    int s = 0;                                                                                     \
    for (int i = 0; i < 16; ++i) s += r[i];                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;                                        \
  }

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %1, %0")
KERNEL_D(k_add_f64, "v_add_f64 %0, %1, %0")
KERNEL_D(k_min_f64, "v_min_f64 %0, %1, %0")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %1, %0")
KERNEL_D(k_frexp_mant, "v_frexp_mant_f64 %0, %0")
KERNEL_D(k_fract_f64, "v_fract_f64 %0, %0")
KERNEL_D(k_trunc_f64, "v_trunc_f64 %0, %0")
KERNEL_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL_D(k_lshr_b64, "v_lshrrev_b64 %0, 12, %0")
KERNEL_I(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
KERNEL_I(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
KERNEL_I(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL_I(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %0")
KERNEL_I(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0")
KERNEL_I(k_add_u32, "v_add_u32 %0, %1, %0")
KERNEL_I(k_add3_u32, "v_add3_u32 %0, %1, %2, %0")
KERNEL_I(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
KERNEL_I(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL_I(k_cndmask_e64, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
KERNEL_I(k_cndmask_neg, "v_cndmask_b32_e64 %0, %1, -%0, s[10:11]")
KERNEL_I(k_cndmask_const, "v_cndmask_b32_e64 %0, 0, %0, s[10:11]")
KERNEL_I(k_cndmask_2v, "v_cndmask_b32_e64 %0, %1, %2, s[10:11]")
KERNEL_I(k_addc, "v_addc_co_u32 %0, vcc, 0, %0, vcc")
KERNEL_I(k_ashr, "v_ashrrev_i32 %0, 31, %0")
KERNEL_I(k_and, "v_and_b32 %0, %1, %0")
KERNEL_I(k_or, "v_or_b32 %0, %1, %0")
KERNEL_I(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_I(k_sub_u32, "v_sub_u32 %0, %1, %0")
KERNEL_I(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL_I(k_cmp_i32_sgpr, "v_cmp_lt_i32 s[10:11], %1, %0")
KERNEL_D(k_cmp_f64_sgpr, "v_cmp_lt_f64 s[10:11], %1, %0")
KERNEL_D(k_cmp_f64_abs, "v_cmp_gt_f64 s[10:11], |%0|, %1")
KERNEL_D(k_fma_f64_3op, "v_fma_f64 %0, %1, %2, %1")
KERNEL_D(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_D(k_fma_f64_neg, "v_fma_f64 %0, -%1, %2, %0")
KERNEL_I(k_xor, "v_xor_b32 %0, %1, %0")
KERNEL_I(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_I(k_alignbit, "v_alignbit_b32 %0, %1, %0, 12")
KERNEL_L(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL_L(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL_DI(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNEL_DI(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
KERNEL_ID(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1")
KERNEL_I(k_nop, "s_nop 0")
// packed / dual forms that could carry two 32-bit operations per instruction
KERNEL_D(k_pk_add_f32, "v_pk_add_f32 %0, %1, %0")
KERNEL_D(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_D(k_pk_mul_f32, "v_pk_mul_f32 %0, %1, %0")

__global__ void k_clock(long long *out) {
  const long long t0 = __builtin_readcyclecounter();
  const long long w0 = wall_clock64();
  double a = (double)threadIdx.x;
  for (int it = 0; it < 200000; ++it) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a));
  const long long t1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (long long)a; }
}

typedef void (*kern_t)(double *, double, double);

static int run(const char *name, kern_t k, int per_iter, double *out_d) {
  // time the whole launch with events at W = 1, 2, 3 waves per SIMD on all 256 CUs: the launch runs
  // ITER * 16 * per_iter instructions in each of 4 W waves per CU
  printf("%-18s", name);
  for (int W = 1; W <= 3; ++W) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int threads = 256 * W;  // 4 W waves: W per SIMD
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out_d, 1.25, 0.75);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out_d, 1.25, 0.75);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    // cycles (at 2.4 GHz nominal; the clock line below gives the measured rate) per wave-instruction per SIMD
    const double cyc = (double)best * 1e-3 * 2.4e9 / ((double)ITER * 16.0 * per_iter * W);
    printf("  W=%d %7.2f", W, cyc);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
  }
  printf("   (cycles at 2.4 GHz per wave-instruction per SIMD)\n");
  return 0;
}

int main() {
  double *out_d;
  CHECK(hipMalloc(&out_d, sizeof(double) * 256 * 1024));
  long long *clk_d, clk_h[3];
  CHECK(hipMalloc(&clk_d, 3 * sizeof(long long)));
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, clk_d);
  CHECK(hipMemcpy(clk_h, clk_d, sizeof clk_h, hipMemcpyDeviceToHost));
  printf("clock check: 200000 dependent v_fma_f64: s_memtime ticks %lld, wall_clock64 ticks %lld (100 MHz)\n",
         clk_h[0], clk_h[1]);
  printf("  -> %.2f ns per dependent fma = %.2f cycles at 2.4 GHz\n", clk_h[1] * 10.0 / 200000.0,
         clk_h[1] * 10.0 / 200000.0 * 2.4);
#define RUN(k, n) if (run(#k, k, n, out_d)) return 1;
  RUN(k_nop, 1)
  RUN(k_fma_f64, 1) RUN(k_mul_f64, 1) RUN(k_add_f64, 1) RUN(k_min_f64, 1) RUN(k_cmp_f64, 1)
  RUN(k_mov_b64, 1)
  RUN(k_rcp_f64, 1) RUN(k_rsq_f64, 1) RUN(k_sqrt_f64, 1)
  RUN(k_frexp_mant, 1) RUN(k_fract_f64, 1) RUN(k_trunc_f64, 1) RUN(k_ldexp_f64, 1)
  RUN(k_cvt_f64_i32, 1) RUN(k_cvt_f64_u32, 1) RUN(k_cvt_i32_f64, 1)
  RUN(k_lshr_b64, 1) RUN(k_lshl_add_u64, 1)
  RUN(k_mul_lo_u32, 1) RUN(k_mul_hi_u32, 1) RUN(k_mad_u64_u32, 1)
  RUN(k_mad_i32_i24, 1) RUN(k_mul_u32_u24, 1) RUN(k_mul_hi_u32_u24, 1)
  RUN(k_add_u32, 1) RUN(k_add3_u32, 1) RUN(k_bfi_b32, 1) RUN(k_cndmask, 1) RUN(k_xor, 1)
  RUN(k_lshl_add, 1) RUN(k_alignbit, 1)
  RUN(k_cndmask_e64, 1) RUN(k_cndmask_neg, 1) RUN(k_cndmask_const, 1) RUN(k_cndmask_2v, 1) RUN(k_addc, 1)
  RUN(k_ashr, 1) RUN(k_and, 1) RUN(k_or, 1) RUN(k_mov_b32, 1) RUN(k_sub_u32, 1) RUN(k_mad_u32_u24, 1)
  RUN(k_cmp_i32_sgpr, 1) RUN(k_cmp_f64_sgpr, 1) RUN(k_cmp_f64_abs, 1)
  RUN(k_fma_f64_3op, 1) RUN(k_fmac_f64, 1) RUN(k_fma_f64_neg, 1)
  RUN(k_pk_add_f32, 1) RUN(k_pk_fma_f32, 1) RUN(k_pk_mul_f32, 1)
  return 0;
}
