"""SamsungV2Decompressor: the oracle's restatement (oracle/rsx_oracle.c) against the
reference build, on streams of the writer in samsung_v2_cases.py (every block mode, scale
changes, all three optimisation flags) and on damaged ones.  The codec is not served by
the GPU library yet; this is the checker a later kernel will be held to."""
import numpy as np
import pytest

import samsung_v2_cases as V2
from oracle_lib import HostImage, Ref


@pytest.fixture(scope="module")
def ref():
    if not Ref.available():
        pytest.skip("oracle/_ref absent")
    r = Ref()
    if not hasattr(r.lib, "ref_samsung_v2_decompress"):
        pytest.skip("oracle/_ref predates the SamsungV2 entry point")
    return r


def _target(rng, h, w, bits):
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    hi = (1 << bits) - 1
    t = 0.3 * hi + 0.3 * hi * x / w + 0.2 * hi * y / h + rng.normal(0, 0.004 * hi, (h, w))
    return np.clip(t, 0, hi).astype(np.int64)


@pytest.mark.parametrize("optflags", range(8))
@pytest.mark.parametrize("bits", [12, 14])
def test_writer_reference_and_oracle_agree(ref, oracle, optflags, bits):
    rng = np.random.default_rng([90, optflags, bits])
    h, w = int(rng.integers(2, 40)), 16 * int(rng.integers(1, 12))
    data, want = V2.encode(rng, _target(rng, h, w, bits), bits, optflags)
    img = ref.image(w, h)
    st = ref.samsung_v2(bits, data, img)
    assert st == 0, ref.last_error()
    assert np.array_equal(img.pixels(), want)
    host = HostImage(w, h)
    assert oracle.samsung_v2(bits, data, host) == 0
    assert np.array_equal(host.pixels(), want)


@pytest.mark.parametrize("seed", range(40))
def test_damaged_streams_same_verdict(ref, oracle, seed):
    rng = np.random.default_rng([91, seed])
    bits = int(rng.choice([12, 14]))
    h, w = int(rng.integers(2, 24)), 16 * int(rng.integers(1, 8))
    data, _ = V2.encode(rng, _target(rng, h, w, bits), bits, int(rng.integers(0, 8)))
    data = data.copy()
    kind = seed % 4
    if kind == 0:                      # flipped bits somewhere in the rows
        for _ in range(int(rng.integers(1, 6))):
            data[int(rng.integers(16, data.size))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:                    # truncated
        data = data[:int(rng.integers(16, data.size))]
    elif kind == 2:                    # header fields
        data[int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
    else:                              # random bits after the header
        data[16:] = rng.integers(0, 256, size=data.size - 16, dtype=np.uint8)
    img = ref.image(w, h)
    s_ref = ref.samsung_v2(bits, data, img)
    host = HostImage(w, h)
    s_or = oracle.samsung_v2(bits, data, host)
    assert (s_ref == 0) == (s_or == 0), (s_ref, s_or, ref.last_error())
    if s_ref == 0:
        assert np.array_equal(img.pixels(), host.pixels())
    else:
        assert s_ref == s_or, (s_ref, s_or, ref.last_error())
