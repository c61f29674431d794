"""The jump-ahead the Nikon reconstruction kernel relies on (rsx_ljpeg_recon.hip:
nk_predict_kernel): RawImageDataU16::setWithLookUp (common/RawImage.h:335-353)
advances its dither state once per pixel with the multiply-with-carry step
    r' = 15700 * (r & 65535) + (r >> 16),
and with m = 15700 * 2^16 - 1 one has 2^16 * r' = r (mod m), so the state after n
pixels is r_0 * 15700^n mod m for seeds r_0 < m (the seed is 24 bits).  A lane
jumps to its first pixel with that closed form and then steps like the reference."""
import numpy as np

A, M = 15700, 15700 * 65536 - 1


def step(r):
    return (A * (r & 65535) + (r >> 16)) & 0xFFFFFFFF


def test_mwc_jump_ahead_matches_stepping():
    rng = np.random.default_rng(99)
    for seed in [0, 1, 2, 0xFFFFFF, 0x800000] + [int(x) for x in rng.integers(0, 1 << 24, 200)]:
        r = seed
        for n in range(1, 3000):
            r = step(r)
            if n in (1, 2, 7, 8, 511, 512, 513, 2999) or n % 977 == 0:
                assert r == seed * pow(A, n, M) % M or (seed == 0 and r == 0), (seed, n)
    # the per-row / per-lane powers the host and the kernel precompute compose
    w, y, lane = 6016, 1234, 37
    seed = 0xABCDE
    at_row = seed * pow(A, y * w, M) % M
    at_lane = at_row * pow(A, 8 * lane, M) % M
    r = seed
    for _ in range(y * w + 8 * lane):
        r = step(r)
    assert r == at_lane


def test_mwc_states_stay_below_the_modulus():
    """The closed form needs r < m; the step keeps every state below m once the
    seed is (24-bit seeds are)."""
    rng = np.random.default_rng(5)
    for seed in rng.integers(0, 1 << 24, 50):
        r = int(seed)
        for _ in range(2000):
            r = step(r)
            assert r < M
