"""Pin of the noise generator: the oracle's Philox4x32-10 against the known-answer vectors that ship with Random123
(kat_vectors, lines `philox4x32 10 ...`; Salmon et al., SC'11).  The device generator is compared with this oracle in
tests/test_gpu_ops.py::test_philox_matches_oracle, so the chain KAT -> oracle -> HIP kernel is closed."""
import numpy as np

from oracle import philox as OP

# (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        got = OP.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want, (ctr, key)


def test_philox_known_answers_vectorised():
    """The same vectors through one vectorised call (the oracle is used on arrays)."""
    c = [np.array([k[0][i] for k in KAT[:1] * 3], dtype=np.uint32) for i in range(4)]
    got = OP.philox4x32_10(*c, KAT[0][1][0], KAT[0][1][1])
    for i in range(4):
        assert np.all(got[i] == np.uint32(KAT[0][2][i]))


def test_uniform_words_are_the_counter_words():
    """uniform(): element e uses word e & 3 of the block with counter (e >> 2 [lo, hi], stream, step), key = seed."""
    u = OP.uniform(4, 0, step=0, stream_id=0, first_index=0)
    want = [((np.uint32(w) >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
            for w in KAT[0][2]]
    assert np.array_equal(u, np.array(want, dtype=np.float32))
    # counter word 0 = e >> 2, words 2, 3 = (stream, step), key = (seed lo, seed hi): the third vector's block
    seed, step, stream = 0x299f31d0a4093822, 0x03707344, 0x13198a2e
    c0, c1 = 0x243f6a88, 0x85a308d3 & 0x3fffffff        # e = ctr << 2 must fit 64 bits
    got = OP.philox4x32_10(np.array([c0], np.uint32), np.array([c1], np.uint32), np.array([stream], np.uint32),
                           np.array([step], np.uint32), seed & 0xffffffff, seed >> 32)
    u = OP.uniform(4, seed, step=step, stream_id=stream, first_index=((c1 << 32) | c0) << 2)
    want = [((g >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for g in got]
    assert np.array_equal(u, np.concatenate(want).astype(np.float32))
