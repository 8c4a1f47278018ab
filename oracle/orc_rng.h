/* oracle/orc_rng.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Random-number source of the CPU oracle.
 *
 * The reference draws every uniform through Kokkos::Random_XorShift64_Pool
 * (reference src/jaybenne/jaybenne.hpp:24-27; checkout/return at transport.cpp:73,172).
 * Kokkos is an un-vendored dependency (absent from /root/reference, pin unknown) and the
 * reference's tests pin nothing about the generator, so parity at the level of individual
 * uniforms is UNPINNED.  This build replaces the pool by one independent counter-based stream
 * per particle: Philox4x32-10 (Salmon et al., SC'11; the published Random123 algorithm, also
 * rocRAND's rocrand_state_philox4x32_10).  Stream layout (identical to
 * rocrand_init(seed, subsequence = id, offset = 0)):
 *
 *     key     = { key0, key1 }                       (seed lo / hi)
 *     counter = { blk_lo, blk_hi, id_lo, id_hi }     (blk = draw_index / 2)
 *
 * One Philox block yields two doubles:  draw 2*blk uses words (0,1), draw 2*blk+1 words (2,3).
 *     k52 = (w_hi << 20) | (w_lo >> 12);   xi = (k52 + 0.5) * 2^-52   in the OPEN interval (0,1)
 *
 * The generator is pinned by the Random123 known-answer vectors (tests/test_oracle_rng.py).
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
  uint32_t key0, key1; /* Philox key: (seed, stream domain) */
  uint64_t id;         /* stream id (particle id / cell id)  */
  uint32_t ctr;        /* number of uniforms drawn so far    */
  const double *tape;  /* if non-NULL: replay tape[pos++ % ntape] instead */
  int ntape;
} orc_rng;

static inline double orc_u52_to_double(uint32_t w_lo, uint32_t w_hi) {
  const uint64_t k = ((uint64_t)w_hi << 20) | (uint64_t)(w_lo >> 12);
  return ((double)k + 0.5) * 2.220446049250313080847263336181640625e-16; /* 2^-52 */
}

static inline double orc_drand(orc_rng *r) {
  if (r->tape) {
    const double v = r->tape[r->ctr % (uint32_t)r->ntape];
    r->ctr++;
    return v;
  }
  const uint32_t blk = r->ctr >> 1;
  const uint32_t c[4] = {blk, 0u, (uint32_t)r->id, (uint32_t)(r->id >> 32)};
  const uint32_t k[2] = {r->key0, r->key1};
  uint32_t o[4];
  orc_philox4x32_10(c, k, o);
  const int h = (int)(r->ctr & 1u);
  r->ctr++;
  return orc_u52_to_double(o[2 * h], o[2 * h + 1]);
}

#endif /* ORC_RNG_H_ */
