"""CPU tests of the numpy restatement of the engine's device RNG (oracle/philox_oracle.py): the block function against
Random123's published known-answer vectors (kat_vectors, philox4x32 with 10 rounds), the Box-Muller mapping's basic
statistics, and the keying properties the multi-GPU sharding relies on."""
import numpy as np

from oracle import philox_oracle as po

# (counter, key, expected) -- Random123 kat_vectors, "philox4x32 10"
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_block_function_matches_random123_known_answers():
    for ctr, key, want in KAT:
        got = po.philox4x32_10(np.array(ctr, np.uint32), key)
        assert tuple(int(v) for v in got) == want
    batch = po.philox4x32_10(np.array([k[0] for k in KAT], np.uint32).reshape(3, 1, 4)[[0, 0, 0]], KAT[0][1])
    assert batch.shape == (3, 1, 4) and tuple(int(v) for v in batch[2, 0]) == KAT[0][2]      # vectorised over leading axes


def test_streams_are_standard_normal_and_distinct():
    a = po.normals(7, np.arange(256), 3, 1, 512).astype(np.float64)
    n = a.size
    assert abs(a.mean()) < 4 / np.sqrt(n) and abs(a.var() - 1) < 0.02
    assert abs((a ** 3).mean()) < 0.03 and abs((a ** 4).mean() - 3) < 0.1 and np.abs(a).max() < 6.7
    for other in (po.normals(8, np.arange(256), 3, 1, 512), po.normals(7, np.arange(256), 4, 1, 512),
                  po.normals(7, np.arange(256), 3, 2, 512), po.normals(7, np.arange(256) + 256, 3, 1, 512)):
        assert abs(np.mean(a * other)) < 0.01                 # different seed / step / stream / samples: uncorrelated


def test_streams_follow_the_global_sample_index():
    whole = po.normals(99, np.arange(10), 5, 3, 918)
    assert np.array_equal(po.normals(99, np.arange(4, 10), 5, 3, 918), whole[4:])
    assert np.array_equal(po.normals(99, [2 ** 33 + 5], 5, 3, 10), po.normals(99, np.array([2 ** 33 + 5], np.uint64), 5, 3, 10))
    assert not np.array_equal(po.normals(99, [2 ** 33 + 5], 5, 3, 10), po.normals(99, [5], 5, 3, 10))   # high counter word used
    assert np.array_equal(po.normals(99, [3], 5, 3, 918)[0, :101], po.normals(99, [3], 5, 3, 101)[0])   # prefix-stable


def test_tapes_have_the_sampling_loop_layout():
    eps, noise = po.step_tapes(5, [0, 7], 3, (9, 3, 34))
    assert eps.shape == (3, 2, 2, 512) and noise.shape == (3, 2, 9, 3, 34)
    assert np.array_equal(eps[2, 1, 1], po.normals(5, [7], 2, 2, 512)[0])
    assert np.array_equal(noise[1, 0].ravel(), po.normals(5, [0], 1, 3, 918)[0])
    assert np.array_equal(po.x_init(5, [7], 27, 34, (9, 3))[0].ravel(), po.normals(5, [7], po.X_T_STEP, 0, 918)[0])
