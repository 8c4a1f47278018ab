// jb_rng.hpp -- one independent random stream per particle (device side).
//
// The reference checks a Kokkos::Random_XorShift64 generator out of a pool once per particle per
// launch (reference src/jaybenne/transport.cpp:73,172; jaybenne.hpp:24-27), so which uniforms a
// particle sees depends on launch geometry.  Here every particle owns its generator, whose
// 64-bit state travels with the particle (8 bytes in the swarm) -- one linear congruential
// stream per history, as in MCNP and OpenMC:
//
//   seeding  the streams are disjoint segments of the generator's single cycle, a fixed stride
//            S = 2^34 - 3 apart (rng_stream_start below); the base point comes from
//            Philox4x32-10 laid out like rocRAND's rocrand_init(seed, subsequence, offset = 0).
//            Done once, when the particle is sourced.
//   draw     s = s * 6364136223846793005 + 1442695040888963407 (mod 2^64, Knuth's MMIX LCG);
//            xi = ((s >> 12) + 0.5) * 2^-52   in the open interval (0,1)
//
// Results are therefore independent of wave scheduling, block -> GPU partition and hand-off.
// Cost on MI355X (tools/microbench.hip), SIMD-cycles per wave-level uniform: this generator 43,
// xorshift64* (the recurrence of the reference's pool; built and measured first) 77,
// Philox4x32-10 per pair of uniforms 163 each.  An IMC event draws four.
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

constexpr uint32_t kRngDomainParticle = 0u;  // Philox key word 1
constexpr uint32_t kRngDomainCell = 1u;      // per-cell streams of the source's stochastic rounding

__device__ __forceinline__ uint64_t rng_seed_state(uint32_t seed, uint32_t domain, uint64_t id) {
  const PhiloxBlock b = philox4x32_10(0u, 0u, (uint32_t)id, (uint32_t)(id >> 32), seed, domain);
  return ((uint64_t)b.w1 << 32) | b.w0;
}

// Start of the stream of the particle with creation index `id`.  All particle streams are
// DISJOINT segments of the one 2^64-cycle of the generator, a fixed stride apart (the
// arrangement of MCNP / OpenMC; Brown, "Random number generation with arbitrary strides", 1994):
//   start(id) = T_S^(id mod 2^30) ( base(id >> 30) ),   T_S = S steps of the recurrence,
//   S = 2^34 - 3 (odd, so a skip changes every bit of the state; 2^30 S < 2^64: no wrap),
//   base(g) = words (0,1) of Philox4x32-10(counter = {0, 0, g_lo, g_hi}, key = {seed, 0}).
// Two particles whose ids lie in the same block of 2^30 consecutive ids cannot see a common
// draw unless one of them draws more than S = 1.7e10 uniforms in its life (an IMC history of
// the stepdiff decks draws ~5e3 per cycle); blocks of ids start at unrelated (Philox) points.
constexpr uint64_t kLcgMul = 6364136223846793005ull, kLcgInc = 1442695040888963407ull;
constexpr uint64_t kStrideMul = 0xb0c24fccf6a8435dull;  // a^S mod 2^64
constexpr uint64_t kStrideInc = 0x574abb626a358eebull;  // c (a^S - 1) / (a - 1) mod 2^64
__device__ __forceinline__ uint64_t rng_stream_start(uint32_t seed, uint64_t id) {
  uint64_t s = rng_seed_state(seed, kRngDomainParticle, id >> 30);
  uint64_t A = kStrideMul, C = kStrideInc;
  for (uint32_t n = (uint32_t)id & 0x3fffffffu; n != 0u; n >>= 1) {  // T_S^n by squaring
    if (n & 1u) s = A * s + C;
    C = (A + 1ull) * C;
    A = A * A;
  }
  return s;
}

// (k52 + 0.5) * 2^-52, formed without an integer -> double conversion from 1.k52 (exponent bits
// of 1.0 over the 52-bit mantissa k52).  Same value as ((double)k52 + 0.5) * 2^-52.
__device__ __forceinline__ double u52_to_double(uint64_t k52) {
  const double one_to_two = __longlong_as_double((long long)(0x3ff0000000000000ull | k52));
  // 1.k52 - (1 - 2^-53) = (2 k52 + 1) 2^-53: an odd 53-bit integer times 2^-53, so the single
  // subtraction is exact
  return one_to_two - 0.99999999999999988897769753748434595763683319091796875;  // 1 - 2^-53
}

struct LcgRng {
  uint64_t s;
  __device__ __forceinline__ explicit LcgRng(uint64_t state) : s(state) {}
  __device__ __forceinline__ double drand() {
    s = s * kLcgMul + kLcgInc;
    return u52_to_double(s >> 12);
  }
  // the next two draws at once: both states are formed from the current one (two independent
  // multiply-adds instead of a chain of two); same states, same uniforms as drand(); drand()
  __device__ __forceinline__ void drand2(double &u1, double &u2) {
    constexpr uint64_t kMul2 = kLcgMul * kLcgMul, kInc2 = (kLcgMul + 1ull) * kLcgInc;
    const uint64_t s1 = s * kLcgMul + kLcgInc;
    s = s * kMul2 + kInc2;
    u1 = u52_to_double(s1 >> 12);
    u2 = u52_to_double(s >> 12);
  }
  // consume one draw whose value cannot influence the result
  __device__ __forceinline__ void skip() { s = s * kLcgMul + kLcgInc; }
  // ... and two (one multiply-add by the constants of two steps: the same state)
  __device__ __forceinline__ void skip2() {
    constexpr uint64_t kMul2 = kLcgMul * kLcgMul, kInc2 = (kLcgMul + 1ull) * kLcgInc;
    s = s * kMul2 + kInc2;
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
  __device__ __forceinline__ void skip() { ++ctr; }
  __device__ __forceinline__ void drand2(double &u1, double &u2) {
    u1 = drand();
    u2 = drand();
  }
};

}  // namespace jb
