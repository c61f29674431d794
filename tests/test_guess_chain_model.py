"""CPU model of the start guesses lj_unstuff_kernel leaves for the single-pass LJPEG kernel
(rawspeed_amd/csrc/rsx_ljpeg.hip).  No GPU: the algebra of the scheme.

1. The chain over the slots -- A[c] = parse(c, 0), B[c] = parse(c, A[c-1]), C[c] = parse(c,
   B[c-1]), with C[c] = B[c] wherever B[c-1] = A[c-1] -- gives exactly the guesses of "parse
   the three slots before slot t from bit 0" (what the kernel did before, three parses a
   slot), with 2.25 parses a slot on sensor-like data.
2. How good the guesses are: against the true symbol grid, one / two / three slots.
3. Constant regions: a slot that is the zero-difference code over and over has the period of
   that code and the code at exactly one phase; the first symbol boundary behind the slot
   follows from the phase, while a parse from bit 0 can stay off the grid for ever.
"""
import numpy as np
import pytest

from rawspeed_amd import synth

import cases as C

SLOT = 512


def canonical_lengths(table):
    """code -> (code length, SSSS) of a canonical JPEG table (counts[16], values)."""
    counts, values = table
    out, code, k = {}, 0, 0
    for l in range(1, 17):
        for _ in range(counts[l - 1]):
            out[(l, code)] = values[k]
            k += 1
            code += 1
        code <<= 1
    return out


def symbol_lengths(bits, table):
    """total length (code + difference bits) of the symbol that would start at every bit"""
    codes = canonical_lengths(table)
    maxl = max(l for l, _ in codes)
    lut = np.zeros(1 << maxl, np.int64)
    for w in range(1 << maxl):
        for l in range(1, maxl + 1):
            c = w >> (maxl - l)
            if (l, c) in codes:
                lut[w] = l + codes[(l, c)]
                break
    pad = np.concatenate([bits, np.zeros(64, np.uint8)])
    win = np.zeros(len(bits), np.int64)
    for i in range(maxl):
        win = (win << 1) | pad[i:i + len(bits)]
    return lut[win]


def make_stream(rng, table, px):
    h, w = px.shape
    rows = C.ljpeg_stream_rows(px, 2, 1, w // 2, h, rng)
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [table] * 2, 0, False)
    b = np.asarray(scan, np.uint8)
    keep = np.ones(len(b), bool)
    ff = np.where(b[:-1] == 0xFF)[0]
    keep[ff + 1] &= ~(b[ff + 1] == 0)          # un-stuff FF 00
    return np.unpackbits(b[keep])


def parse(slen, pos, end):
    end = min(end, len(slen))
    while pos < end:
        pos += int(slen[pos]) or 16
    return pos - end


def true_entries(slen, n_symbols, n_slots):
    starts = np.zeros(n_symbols, np.int64)
    pos = 0
    for i in range(n_symbols):
        starts[i] = pos
        pos += int(slen[pos])
    ent = np.full(n_slots + 1, -1, np.int64)
    for k in range(n_slots + 1):
        j = np.searchsorted(starts, SLOT * k)
        if j < n_symbols:
            ent[k] = starts[j] - SLOT * k
    return ent


def test_chain_over_the_slots_is_the_three_slot_parse():
    rng = np.random.default_rng(31)
    table = C.NIKON
    px = C.smooth_image(rng, 120, 1024, sigma=25.0)
    bits = make_stream(rng, table, px)
    slen = symbol_lengths(bits, table)
    n = len(bits) // SLOT
    # direct: the guess for slot t from bit 0 of slot t - 3
    direct = np.zeros(n + 1, np.int64)
    for t in range(1, n + 1):
        e = 0
        for k in (3, 2, 1):
            c = t - k
            if c >= 0:
                e = parse(slen, SLOT * c + e, SLOT * c + SLOT)
        direct[t] = e
    # chain, with the shortcut
    A = np.array([parse(slen, SLOT * c, SLOT * c + SLOT) for c in range(n)])
    B = A.copy()
    parses = n
    for c in range(1, n):
        if A[c - 1] != 0:
            B[c] = parse(slen, SLOT * c + A[c - 1], SLOT * c + SLOT)
            parses += 1
    Cc = B.copy()
    third = 0
    for c in range(1, n):
        if B[c - 1] != A[c - 1]:
            Cc[c] = parse(slen, SLOT * c + B[c - 1], SLOT * c + SLOT)
            third += 1
    assert np.array_equal(Cc, direct[1:n + 1])
    assert third < 0.05 * n                      # 98 % of the slots need no third parse
    assert (parses + third) / n < 2.1


def test_guesses_against_the_true_symbol_grid():
    rng = np.random.default_rng(32)
    table = C.NIKON
    px = C.smooth_image(rng, 200, 1024, sigma=25.0)
    bits = make_stream(rng, table, px)
    slen = symbol_lengths(bits, table)
    n = len(bits) // SLOT - 1
    truth = true_entries(slen, px.size, n)
    wrong = []
    for slots in (1, 2, 3):
        bad = 0
        for t in range(slots, n):
            e = 0
            for k in range(slots, 0, -1):
                e = parse(slen, SLOT * (t - k) + e, SLOT * (t - k) + SLOT)
            bad += int(truth[t] >= 0 and e != truth[t])
        wrong.append(bad / (n - slots))
    # (every slot parsed buys certainty: 7 % / 2 % / 0.5 % of the guesses wrong for this
    # table and noise; the BASELINE frames, with shorter symbols, 1.7 % / 0.03 % / next to none)
    assert wrong[0] < 0.15 and wrong[1] < wrong[0] / 2 and wrong[2] < wrong[1] / 2, wrong


def test_constant_slots_give_their_symbol_grid_away():
    rng = np.random.default_rng(33)
    table = C.NIKON
    codes = canonical_lengths(table)
    (zl, zc), = [(l, c) for (l, c), ssss in codes.items() if ssss == 0]
    px = C.smooth_image(rng, 64, 2048, sigma=20.0)
    px[8:40, 256:1800] = 16383                   # a blown region
    bits = make_stream(rng, table, px)
    slen = symbol_lengths(bits, table)
    n = len(bits) // SLOT - 1
    truth = true_entries(slen, px.size, n)
    pad = np.concatenate([bits, np.zeros(64, np.uint8)])
    zbits = np.array([(zc >> (zl - 1 - i)) & 1 for i in range(zl)], np.uint8)
    found = off_grid = 0
    for c in range(n - 1):
        seg = pad[SLOT * c:SLOT * c + SLOT + zl]
        if not np.array_equal(seg[:SLOT], seg[zl:]):
            continue                             # not periodic with the code's length
        hits = [p for p in range(zl) if np.array_equal(seg[p:p + zl], zbits)]
        if len(hits) != 1:
            continue
        r = (SLOT - hits[0]) % zl
        guess = zl - r if r else 0
        found += 1
        assert guess == truth[c + 1], c          # the rule is right where it applies
        off_grid += int(parse(slen, SLOT * c, SLOT * c + SLOT) != truth[c + 1])
    assert found > 20
    assert off_grid > 0                          # ... and a parse from bit 0 is not
