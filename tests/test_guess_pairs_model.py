"""CPU model of lj_guess_parse_pairs (rawspeed_amd/csrc/rsx_ljpeg.hip, round 4): the start
guesses parsed two symbols per window read, with a 256-byte (bank-conflict-free) length table
for codes of at most 8 bits next to the 10-bit one.  No GPU: the claim is that the pair loop
visits exactly the positions of the one-symbol loop -- same exit, same symbol count -- for
every table the single-pass kernel accepts, special entries (codes longer than 10 bits,
SSSS = 16, holes in the code space) included."""
import numpy as np
import pytest

import cases as C
from test_lut10_model import canonical


def fast_table(counts, values, fix16=False):
    """ljpeg_build_fast_table (rsx_ljpeg_fast.hip): per 10-bit index (x, y) with x = shift |
    total << 5 | special << 31 and y = 2^SSSS - 1; special entries carry an advance."""
    codes = canonical(counts, values)
    x = np.zeros(1024, np.uint32)
    y = np.zeros(1024, np.uint32)
    lut11 = {}
    for code, l, ssss in codes:
        if l <= 11:
            lo = code << (11 - l)
            for i in range(lo, lo + (1 << (11 - l))):
                total = l + (ssss if ssss != 16 else (16 if fix16 else 0))
                lut11[i] = (l, ssss, total)
    for i in range(1024):
        ea, eb = lut11.get(2 * i), lut11.get(2 * i + 1)
        plain = (ea is not None and ea[0] <= 10 and ea[1] < 16 and ea[2] == ea[0] + ea[1]
                 and 1 <= ea[2] <= 26)
        if plain:
            x[i] = (32 - ea[2]) | (ea[2] << 5)
            y[i] = (1 << ea[1]) - 1
        else:
            adv = 16
            if ea is not None and ea[2] >= 1:
                adv = ea[2]
            elif eb is not None and eb[2] >= 1:
                adv = eb[2]
            x[i] = 0x80000000 | (min(adv, 63) << 5)
    return x, y


def k0_tables(x, y):
    """what lj_unstuff_kernel parks in LDS: lut10[i] = total (or the advance of a special
    entry), lut8[j] = total of entry 4j if its code has at most 8 bits, else 0x80"""
    lut10 = ((x >> 5) & 63).astype(np.int64)
    lut8 = np.full(256, 0x80, np.int64)
    for j in range(256):
        e = 4 * j
        total = int((x[e] >> 5) & 63)
        code = total - bin(int(y[e])).count("1")
        if not (x[e] & 0x80000000) and code <= 8 and total >= 1:
            lut8[j] = total
    return lut10, lut8


def window(bits, pos):
    w = 0
    for b in bits[pos:pos + 32]:
        w = (w << 1) | int(b)
    return w


def parse_single(bits, lut10, start, end):
    pos, n = start, 0
    while pos < end:
        pos += int(lut10[window(bits, pos) >> 22])
        n += 1
    return pos - end, n


def parse_pairs(bits, lut10, lut8, start, end):
    pos, n = start, 0
    pair_end = end - 26 if end > 26 else 0
    while pos < pair_end:
        w = window(bits, pos)
        l1 = int(lut8[w >> 24])
        w2 = (w << (l1 & 31)) & 0xFFFFFFFF
        l2 = int(lut8[w2 >> 24])
        if l1 + l2 < 128:
            assert l1 <= 24          # what makes the second index the stream's bits
            pos += l1 + l2
            n += 2
        else:
            pos += int(lut10[w >> 22])
            n += 1
    while pos < end:
        pos += int(lut10[window(bits, pos) >> 22])
        n += 1
    return pos - end, n


@pytest.mark.parametrize("seed", range(12))
def test_pair_parse_equals_single_symbol_parse(seed):
    rng = np.random.default_rng(900 + seed)
    # random tables: skewed ones have codes of 11-16 bits (special entries), n_cat 17 has SSSS 16
    counts, values = C.random_huffman_table(rng, n_cat=int(rng.integers(9, 18)),
                                            skew=None if seed % 3 else 2.5)
    x, y = fast_table(counts, values, fix16=bool(seed & 1))
    lut10, lut8 = k0_tables(x, y)
    # a hit of the small table is a code of <= 8 bits + <= 16 difference bits
    assert all(v <= 24 for v in lut8 if v != 0x80)
    # ... and agrees with all four 10-bit entries it stands for
    for j in range(256):
        if lut8[j] != 0x80:
            assert all(lut10[4 * j + k] == lut8[j] for k in range(4))
    for trial in range(40):
        nbits = int(rng.choice([512, 504, 496, 40, 27, 26, 8, 0]))
        kind = trial % 3
        if kind == 0:
            bits = rng.integers(0, 2, size=nbits + 96).astype(np.uint8)
        elif kind == 1:                     # mostly zeros / ones: long runs of one code
            bits = (rng.random(nbits + 96) < (0.05 if trial & 1 else 0.95)).astype(np.uint8)
        else:
            bits = np.tile(rng.integers(0, 2, size=int(rng.integers(2, 9))), 400)[:nbits + 96] \
                .astype(np.uint8)
        for start in (0, 1, 7, 31):
            if start >= max(nbits, 1):
                continue
            assert parse_pairs(bits, lut10, lut8, start, nbits) == \
                parse_single(bits, lut10, start, nbits)
