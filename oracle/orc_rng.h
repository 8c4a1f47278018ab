/* oracle/orc_rng.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Random-number source of the CPU oracle.
 *
 * The reference draws every uniform through Kokkos::Random_XorShift64_Pool
 * (reference src/jaybenne/jaybenne.hpp:24-27; checkout/return at transport.cpp:73,172): a pool of
 * generators shared by whatever threads run, so which uniforms a particle sees depends on the
 * launch.  Kokkos is an un-vendored dependency (absent from /root/reference, pin unknown) and the
 * reference's tests pin nothing about the generator, so parity at the level of individual uniforms
 * is UNPINNED.  This build gives every particle its own generator, whose 64-bit state travels
 * with the particle -- the arrangement of the production Monte Carlo transport codes (MCNP,
 * OpenMC: one linear congruential stream per history):
 *
 *     seeding   the particle streams are disjoint segments of the generator's single cycle,
 *               a fixed stride apart (orc_rng_stream_start below); base points come from
 *               Philox4x32-10, the published Random123 algorithm (Salmon et al., SC'11), laid
 *               out like rocRAND's rocrand_init(seed, subsequence, offset = 0).
 *     draw      s = s * 6364136223846793005 + 1442695040888963407  (mod 2^64; Knuth's MMIX LCG)
 *               xi = ((s >> 12) + 0.5) * 2^-52      in the OPEN interval (0,1)
 *
 * Pinned by the Random123 Philox known-answer vectors and by a big-integer restatement of the
 * recurrence in tests/test_oracle_rng.py.
 *
 * A "tape" mode replays a caller-supplied list of uniforms, so that every branch of the step
 * functions can be driven deterministically (golden vectors, tests/golden/).
 */
#ifndef ORC_RNG_H_
#define ORC_RNG_H_

#include <stdint.h>

#define ORC_PHILOX_M0 0xD2511F53u
#define ORC_PHILOX_M1 0xCD9E8D57u
#define ORC_PHILOX_W0 0x9E3779B9u
#define ORC_PHILOX_W1 0xBB67AE85u

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                     uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)ORC_PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)ORC_PHILOX_M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += ORC_PHILOX_W0;
    k1 += ORC_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct orc_rng {
  uint64_t s;          /* LCG state */
  uint32_t ctr;        /* number of uniforms drawn through this handle (diagnostic) */
  const double *tape;  /* if non-NULL: replay tape[pos++ % ntape] instead */
  int ntape;
} orc_rng;

#define ORC_RNG_DOMAIN_PARTICLE 0u /* Philox key word 1: particle streams */
#define ORC_RNG_DOMAIN_CELL 1u     /* per-cell streams of the source's stochastic rounding */

static inline uint64_t orc_rng_seed_state(uint32_t seed, uint32_t domain, uint64_t id) {
  const uint32_t c[4] = {0u, 0u, (uint32_t)id, (uint32_t)(id >> 32)};
  const uint32_t k[2] = {seed, domain};
  uint32_t o[4];
  orc_philox4x32_10(c, k, o);
  return ((uint64_t)o[1] << 32) | o[0];
}

/* Start of the stream of the particle with creation index `id`: disjoint segments of the
 * generator's single cycle, a fixed odd stride S = 2^34 - 3 apart within each block of 2^30
 * consecutive ids (MCNP / OpenMC arrangement; Brown 1994, "Random number generation with
 * arbitrary strides"); the block's base point is Philox-derived.  T_S^n by squaring of the
 * affine map x -> A x + C. */
#define ORC_LCG_MUL 6364136223846793005ull
#define ORC_LCG_INC 1442695040888963407ull
#define ORC_STRIDE ((1ull << 34) - 3ull)
#define ORC_STRIDE_MUL 0xb0c24fccf6a8435dull /* a^S mod 2^64 */
#define ORC_STRIDE_INC 0x574abb626a358eebull /* c (a^S - 1) / (a - 1) mod 2^64 */
static inline uint64_t orc_rng_stream_start(uint32_t seed, uint64_t id) {
  uint64_t s = orc_rng_seed_state(seed, ORC_RNG_DOMAIN_PARTICLE, id >> 30);
  uint64_t A = ORC_STRIDE_MUL, C = ORC_STRIDE_INC;
  for (uint32_t n = (uint32_t)id & 0x3fffffffu; n != 0u; n >>= 1) {
    if (n & 1u) s = A * s + C;
    C = (A + 1ull) * C;
    A = A * A;
  }
  return s;
}

static inline orc_rng orc_rng_from_state(uint64_t state) {
  orc_rng r = {state, 0u, NULL, 0};
  return r;
}

static inline double orc_u52_to_double(uint64_t k52) {
  return ((double)k52 + 0.5) * 2.220446049250313080847263336181640625e-16; /* 2^-52 */
}

static inline double orc_drand(orc_rng *r) {
  if (r->tape) {
    const double v = r->tape[r->ctr % (uint32_t)r->ntape];
    r->ctr++;
    return v;
  }
  const uint64_t s = r->s * ORC_LCG_MUL + ORC_LCG_INC;
  r->s = s;
  r->ctr++;
  return orc_u52_to_double(s >> 12);
}

#endif /* ORC_RNG_H_ */
