// jb_rng.hpp -- per-particle counter-based random streams for the history loop (device side).
//
// The reference checks a Kokkos::Random_XorShift64 generator out of a pool once per particle per
// launch (reference src/jaybenne/transport.cpp:73,172; jaybenne.hpp:24-27), so which uniforms a
// particle sees depends on launch geometry.  Here every particle owns an independent stream that
// travels with it (id + draw count, 12 bytes): Philox4x32-10 keyed by the deck seed, laid out
// exactly like rocRAND's rocrand_init(seed, subsequence = id, offset = 0):
//     key = {key0, key1},  counter = {draw/2, 0, id_lo, id_hi}
// One Philox block gives two doubles in the open interval (0,1):
//     k52 = (w_hi << 20) | (w_lo >> 12);  xi = (k52 + 0.5) * 2^-52
// Results are therefore independent of wave scheduling, block -> GPU partition and hand-off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jb {

struct PhiloxBlock {
  uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ PhiloxBlock philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                     uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return PhiloxBlock{c0, c1, c2, c3};
}

__device__ __forceinline__ double u52_to_double(uint32_t w_lo, uint32_t w_hi) {
  const uint64_t k = ((uint64_t)w_hi << 20) | (uint64_t)(w_lo >> 12);
  return ((double)k + 0.5) * 2.220446049250313080847263336181640625e-16;  // 2^-52
}

// One particle's stream.  Invariant: when ctr is odd, (c2, c3) hold words 2,3 of block ctr/2.
struct PhiloxRng {
  uint32_t key0, key1, id_lo, id_hi, ctr;
  uint32_t c2, c3;

  __device__ __forceinline__ PhiloxRng(uint32_t k0, uint32_t k1, uint64_t id, uint32_t ctr_)
      : key0(k0), key1(k1), id_lo((uint32_t)id), id_hi((uint32_t)(id >> 32)), ctr(ctr_), c2(0),
        c3(0) {
    if (ctr & 1u) {
      const PhiloxBlock b = philox4x32_10(ctr >> 1, 0u, id_lo, id_hi, key0, key1);
      c2 = b.w2;
      c3 = b.w3;
    }
  }

  __device__ __forceinline__ double drand() {
    uint32_t lo, hi;
    if (ctr & 1u) {
      lo = c2;
      hi = c3;
    } else {
      const PhiloxBlock b = philox4x32_10(ctr >> 1, 0u, id_lo, id_hi, key0, key1);
      lo = b.w0;
      hi = b.w1;
      c2 = b.w2;
      c3 = b.w3;
    }
    ++ctr;
    return u52_to_double(lo, hi);
  }
};

// Replays a caller-supplied list of uniforms (debug entry points / golden-vector tests only).
struct TapeRng {
  const double *tape;
  int ntape;
  uint32_t ctr;
  __device__ __forceinline__ TapeRng(const double *t, int n) : tape(t), ntape(n), ctr(0) {}
  __device__ __forceinline__ double drand() {
    const double v = tape[ctr % (uint32_t)ntape];
    ++ctr;
    return v;
  }
};

}  // namespace jb
