"""Pins the oracle's random-number source: Philox4x32-10 against the Random123 known-answer
vectors, the 64-bit LCG against an independent big-integer restatement of the recurrence."""
import numpy as np

from oracle import orc

# Random123 kat_vectors, philox4x32 10 rounds: counter, key -> output
KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        assert tuple(orc.philox(ctr, key)) == want


def test_lcg_stream_matches_recurrence():
    """s <- s * 6364136223846793005 + 1442695040888963407 mod 2^64 (Knuth, MMIX);
    xi = ((s >> 12) + 0.5) 2^-52."""
    for state in (0, 1, 0x9E3779B97F4A7C15, (1 << 64) - 1):
        got, final = orc.draw_stream(state, 1000)
        s = state
        for i in range(1000):
            s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
            assert got[i] == ((s >> 12) + 0.5) * 2.0 ** -52
        assert final == s
        assert got.min() > 0.0 and got.max() < 1.0


def test_stream_seeding_layout():
    """state = words (0,1) of Philox(counter = {0, 0, id_lo, id_hi}, key = {seed, domain})."""
    for seed, domain, sid in ((349857, 0, 0), (123, 1, (5 << 44) | (7 << 24) | 99), (1, 0, 2 ** 40 + 17)):
        w = orc.philox((0, 0, sid & 0xffffffff, sid >> 32), (seed, domain))
        assert orc.seed_state(seed, domain, sid) == (w[1] << 32) | w[0]


A, C_, MASK = 6364136223846793005, 1442695040888963407, (1 << 64) - 1
STRIDE = (1 << 34) - 3


def _jump(s, n):
    """s after n steps of the recurrence (big-integer restatement of Brown's algorithm)."""
    a, c, h, f = 1, 0, A, C_
    while n:
        if n & 1:
            a, c = (a * h) & MASK, (c * h + f) & MASK
        f, h = (f * (h + 1)) & MASK, (h * h) & MASK
        n >>= 1
    return (a * s + c) & MASK


def test_jump_matches_stepping():
    s0 = 0x0123456789abcdef
    s = s0
    for n in range(1, 300):
        s = (s * A + C_) & MASK
        assert _jump(s0, n) == s
    assert _jump(_jump(s0, 12345), STRIDE) == _jump(s0, 12345 + STRIDE)


def test_particle_streams_are_disjoint_strided_segments():
    """start(id) = base(id >> 30) advanced by (id mod 2^30) * S steps, S = 2^34 - 3: the streams
    of one block of 2^30 ids are segments of the one cycle that cannot meet before S draws."""
    seed = 349857
    for g in (0, 1, 5):
        base = orc.seed_state(seed, 0, g)
        assert orc.stream_start(seed, g << 30) == base
        for lo in (1, 2, 3, 1000, 99999, 2 ** 29 + 12345, 2 ** 30 - 1):
            assert orc.stream_start(seed, (g << 30) | lo) == _jump(base, lo * STRIDE), (g, lo)
    # neighbours are exactly S steps apart: stream id + 1 starts where stream id would be after
    # S draws
    s5 = orc.stream_start(seed, 5)
    assert _jump(s5, STRIDE) == orc.stream_start(seed, 6)
    # the last stream of a block ends before the cycle closes: no wrap onto the first
    assert (2 ** 30 - 1) * STRIDE + STRIDE <= 1 << 64
    assert STRIDE % 2 == 1          # an odd skip changes every bit of the state


def test_streams_are_distinct_and_uniform():
    firsts = np.array([orc.draw_stream(orc.stream_start(349857, i), 4)[0] for i in range(4000)])
    assert len(np.unique(firsts[:, 0])) == 4000
    # crude uniformity of the first draw over neighbouring ids, and no lag-1 correlation
    assert abs(firsts[:, 0].mean() - 0.5) < 0.02
    assert abs(np.corrcoef(firsts[:-1, 0], firsts[1:, 0])[0, 1]) < 0.05
    long_run, _ = orc.draw_stream(orc.stream_start(349857, 42), 200000)
    assert abs(long_run.mean() - 0.5) < 0.003 and abs(long_run.var() - 1 / 12) < 0.002
