// Checks the quad-cooperative record gather used by k_ddmc_all: four lanes fetch the four
// 16-byte pieces of ONE 64-byte record with one global_load_lds_dwordx4 (one L1 access instead of
// four), instruction k serves the record of quad-lane k; LDS destination = M0 base + lane * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long quad_bcast(unsigned long long v, int k) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  switch (k) {
  case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
  case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
  case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, true); break;
  default: lo = __builtin_amdgcn_mov_dpp(lo, 0xff, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xff, 0xf, 0xf, true); break;
  }
  return ((unsigned long long)hi << 32) | lo;
}
__global__ void k(const double *base, const unsigned *idx, const int *want, double *out) {
  __shared__ __attribute__((aligned(16))) char buf[4][4][64 * 16];  // [wave][k][lane * 16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool need = want[g] != 0;
  const unsigned long long a = need ? (unsigned long long)(base + 8ull * idx[g]) : 0ull;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned long long ak = quad_bcast(a, kk);
    if (ak != 0ull)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ak + 16u * (lane & 3)),
                                       (__attribute__((address_space(3))) void *)buf[wave][kk], 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0)
  if (need) {
    const v4d *r = (const v4d *)(buf[wave][lane & 3] + 64 * (lane >> 2));
    const v4d r0 = r[0], r1 = r[1];
    double *o = out + 8ull * g;
    o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; o[3] = r0.w; o[4] = r1.x; o[5] = r1.y; o[6] = r1.z; o[7] = r1.w;
  }
}
int main() {
  const int nrec = 1 << 20, n = 256 * 64;
  std::vector<double> h(8ull * nrec);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i * 0.5 + 1.0;
  std::vector<unsigned> idx(n);
  std::vector<int> want(n);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; idx[i] = (s >> 8) % nrec; want[i] = ((s >> 3) & 7) != 0; }
  double *db, *dout; unsigned *di; int *dw;
  hipMalloc(&db, h.size() * 8); hipMalloc(&dout, 8ull * n * 8); hipMalloc(&di, n * 4); hipMalloc(&dw, n * 4);
  hipMemcpy(db, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(di, idx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, want.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0, 8ull * n * 8);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, db, di, dw, dout);
  std::vector<double> o(8ull * n);
  hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 8; ++j) {
      const double exp = want[i] ? h[8ull * idx[i] + j] : 0.0;
      if (o[8ull * i + j] != exp) { if (bad < 5) printf("lane %d piece %d got %g want %g\n", i, j, o[8ull * i + j], exp); ++bad; }
    }
  printf("%s: %ld mismatches of %d values\n", bad ? "FAIL" : "OK", bad, 8 * n);
  return bad != 0;
}
