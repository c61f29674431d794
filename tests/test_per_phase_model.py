"""CPU models of what round 6 added for streams with a Huffman table PER COMPONENT
(LjStreamDev::fast == 3; rawspeed_amd/csrc/rsx_ljpeg.hip: lj_guess_parse_pt, lj_guess_constant_pt;
rsx_ljpeg_fast.hip: lf_step_pt and its 2-byte LUT entries).  No GPU: the algebra.

1. A slot that is the N zero-difference codes in turn: entry state, symbol count and the state behind
   the slot from the ONE place at which the concatenated codes sit in the period -- for every phase
   and every slot length stuffing bytes can leave.
2. K0's chain with the phase (symbol index mod N) in the state ends on the true symbol grid of
   sensor-like 3- and 4-component streams whose tables differ -- as little as two swapped values --
   within the kernel's six rounds in (nearly) every workgroup.
3. The 2-byte LUT entry (shift | total << 5 | SSSS << 11) decodes every symbol of a stream like the
   canonical decoder does: 2^SSSS - 1 from the entry by shift + bit-field mask, the JPEG EXTEND
   without a shift left, the position advanced by 32 * total.
"""
import numpy as np
import pytest

from rawspeed_amd import synth

import cases as C
import test_guess_chain_model as M

SLOT = 512


def zero_code(table):
    codes = M.canonical_lengths(table)
    (l, code), = [(k[0], k[1]) for k, v in codes.items() if v == 0]
    return code, l


def constant_guess_pt(bits_arr, bits, zl, zc, cum, n):
    """lj_guess_constant_pt: (state behind the slot, symbols, entry state) or None"""
    w64 = int("".join(map(str, bits_arr[:64])), 2)
    hits = [p for p in range(zl) if ((w64 << p) & ((1 << 64) - 1)) >> (64 - zl) == zc]
    if len(hits) != 1:
        return None
    p0 = hits[0]
    best_in = best_out = None
    cnt = 0
    for k in range(n):
        f = (p0 + cum[k]) % zl
        m = (bits - f + zl - 1) // zl
        cnt += m
        out = f + m * zl - bits
        best_in = min(best_in, (f, k)) if best_in else (f, k)
        best_out = min(best_out, (out, k)) if best_out else (out, k)
    return best_out[0] | (best_out[1] << 6), cnt, best_in[0] | (best_in[1] << 6)


@pytest.mark.parametrize("tables", [(C.NIKON, C.ALT, C.NIKON), (C.ALT, C.NIKON, C.NIKON, C.ALT)])
def test_constant_slot_of_n_zero_codes(tables):
    n = len(tables)
    zs = [zero_code(t) for t in tables]
    zl = sum(l for _, l in zs)
    cum = [sum(l for _, l in zs[:k]) for k in range(n)]
    zc = 0
    for code, l in zs:
        zc = (zc << l) | code
    pat = [int(ch) for code, l in zs for ch in format(code, "0%db" % l)]
    if sum(1 for p in range(zl) if pat[p:] + pat[:p] == pat) != 1:
        pytest.skip("the codes one behind the other have a shorter period")
    for bits in range(64, SLOT + 1, 8):
        for off in range(zl):
            lo = -2 * zl
            stream = [pat[(i - off) % zl] for i in range(lo, SLOT + 128)]
            starts = sorted((i, k) for k in range(n) for i in range(lo, SLOT + 128)
                            if (i - off - cum[k]) % zl == 0)
            inside = [s for s in starts if 0 <= s[0] < bits]
            nxt = next(s for s in starts if s[0] >= bits)
            want = ((nxt[0] - bits) | (nxt[1] << 6), len(inside), inside[0][0] | (inside[0][1] << 6))
            assert constant_guess_pt(stream[-lo:], bits, zl, zc, cum, n) == want, (bits, off)


def _stream_bits(rng, n, cpp, tables, index, th=120):
    tw = 1536 if cpp == 1 else 768
    px = (6000 + 30 * np.arange(tw * cpp)[None, :] % 4000 + rng.normal(0, 25, (th, tw * cpp))).astype(np.int64)
    px = np.clip(px, 1, 16000).astype(np.uint16)                  # (no clipped stretches: see 1.)
    rows = C.ljpeg_stream_rows(px, n, 1, tw * cpp // n, th, rng, 14)
    scan, _ = synth.ljpeg_encode_scan(rows, n, [1 << 13] * n, [tables[i] for i in index], 0, False)
    b = np.asarray(scan, np.uint8)
    keep = np.ones(len(b), bool)
    ff = np.where(b[:-1] == 0xFF)[0]
    keep[ff + 1] &= ~(b[ff + 1] == 0)
    return np.unpackbits(b[keep])


def _chain(bits, tables, index, n, rounds=12):
    """lj_unstuff_kernel<2, .>'s chain of one stream: A parse from (0, phase 0), B parse from the
    predecessor's A, rounds on whole states; a workgroup that has not settled then takes the PHASE
    PASS -- every slot's entry phase from a prefix sum of the slots' symbol counts (from the phase
    the workgroup starts in: the look-back's), one more parse of the slots whose last parse
    started in another phase, settled if no exit offset and no count moved.  Returns workgroups,
    those still unsettled, wrong states in the settled ones, workgroups the phase pass settled."""
    sl = [M.symbol_lengths(bits, t) for t in tables]
    n_slots = len(bits) // SLOT

    def parse(c, st):
        pos, ph, end, cnt = c * SLOT + (st & 63), (st >> 6) & 3, (c + 1) * SLOT, 0
        while pos < end:
            pos += int(sl[index[ph]][pos]) or 16
            ph = (ph + 1) % n
            cnt += 1
        return (pos - end) | (ph << 6), cnt
    true = [0]
    for c in range(n_slots):
        true.append(parse(c, true[-1])[0])
    own, unsettled, wrong, by_phase_pass = 255, 0, 0, 0
    a = {c: parse(c, 0) for c in range(n_slots)}
    b = {c: parse(c, a[c - 1][0]) if c > 0 else parse(c, 0) for c in range(n_slots)}
    n_wg = (n_slots + own - 1) // own
    for wg in range(n_wg):
        idx = list(range(wg * own, min(n_slots, (wg + 1) * own)))
        eb = {c: b[c][0] for c in idx}
        cnt = {c: b[c][1] for c in idx}
        eu = {c: (a[c - 1][0] if c > 0 else 0) for c in idx}
        entry = true[idx[0]]                                      # (hand-over + look-back)

        def pred(c):
            return eb[c - 1] if c > idx[0] else entry

        def run_rounds():
            for _ in range(rounds):
                lst = [c for c in idx if pred(c) != eu[c]]
                if not lst:
                    return True
                new = {c: (parse(c, pred(c)), pred(c)) for c in lst}
                moved = any(new[c][0][0] != eb[c] for c in lst)
                for c, ((e, k), f) in new.items():
                    eb[c], cnt[c], eu[c] = e, k, f
                if not moved:
                    return True
            return False
        ok = run_rounds()
        if not ok:
            ph, moved = (entry >> 6) & 3, False
            for c in idx:                                          # (all slots at a time on the device)
                want_in = (pred(c) & 63) | (ph << 6)
                ph = (ph + cnt[c]) % n                             # (the counts of BEFORE the pass)
                if want_in != eu[c]:
                    e, k = parse(c, want_in)
                    moved |= (e & 63) != (eb[c] & 63) or k != cnt[c]
                    eb[c], cnt[c], eu[c] = e, k, want_in
            ok = not moved or run_rounds()
            by_phase_pass += ok
        unsettled += not ok
        wrong += sum(eb[c] != true[c + 1] for c in idx) if ok else 0
    return n_wg, unsettled, wrong, by_phase_pass


def test_chain_with_the_phase_in_the_state():
    counts, values = C.NIKON
    v2 = list(values)
    v2[-1], v2[-2] = v2[-2], v2[-1]
    v3 = list(values)
    v3[3], v3[4] = v3[4], v3[3]
    rt = np.random.default_rng(5)
    rnd = [C.random_huffman_table(rt, 15, skew=1.5) for _ in range(4)]
    cases_ = [("A B C", 3, 3, [C.NIKON, C.ALT, rnd[0]], [0, 1, 2]),
              ("A B B", 3, 3, [C.NIKON, C.ALT], [0, 1, 1]),
              ("A B C D", 4, 1, rnd, [0, 1, 2, 3]),
              ("A A B B", 4, 1, [C.NIKON, C.ALT], [0, 0, 1, 1])]
    helped = 0
    for name, n, cpp, tabs, index in cases_:
        bits = _stream_bits(np.random.default_rng(1), n, cpp, tabs, index)
        n_wg, unsettled, wrong, by_pass = _chain(bits, tabs, index, n)
        assert wrong == 0, name                       # a chain that settles settles on the truth
        assert unsettled == 0, (name, unsettled, n_wg)   # (twelve rounds: LJ_GUESS_ROUNDS_PT)
        # ... and with the six rounds of the other streams the phase pass settles workgroups the
        # rounds left, but not all of them (four tables: more than half stay "uncertain")
        _, unsettled6, wrong6, by_pass = _chain(bits, tabs, index, n, rounds=6)
        assert wrong6 == 0
        helped += by_pass
    assert helped > 0
    # ... and what the plan keeps OFF this route (rsx_ljpeg.hip, "NEARLY the same"): tables that share
    # most of their code space -- here NIKON with two values swapped, twice -- differ too rarely for a
    # parse to find its phase and too often for the counts to be right whatever the phase
    near = [C.NIKON, (counts, v2), (counts, v3)]
    bits = _stream_bits(np.random.default_rng(1), 3, 3, near, [0, 1, 2])
    n_wg, unsettled, wrong, _ = _chain(bits, near, [0, 1, 2], 3, rounds=12)
    assert wrong == 0 and unsettled * 5 >= n_wg      # (a workgroup in three or four stays "uncertain")
    assert max(_agreement(near[x], near[y]) for x in range(3) for y in range(x + 1, 3)) > 0.5
    assert max(_agreement(x, y) for x, y in ((C.NIKON, C.ALT), (C.NIKON, rnd[0]), (rnd[0], rnd[1]),
                                              (rnd[2], rnd[3]))) < 0.25


def _agreement(ta, tb, bits=11):
    """the share of the 11-bit patterns that are the same symbol (code length and SSSS) under both
    tables: the plan's test for "nearly the same" """
    def lut(t):
        out = [None] * (1 << bits)
        for (l, code), ssss in M.canonical_lengths(t).items():
            if l <= bits:
                for i in range(code << (bits - l), (code + 1) << (bits - l)):
                    out[i] = (l, ssss)
        return out
    la, lb = lut(ta), lut(tb)
    return sum(x is not None and x == y for x, y in zip(la, lb)) / len(la)


def _entry16(counts, values):
    """the 10-bit LUT in its 2-byte form, from the canonical code (ljpeg_build_fast_table +
    ljpeg_build_fast_table16)"""
    codes = M.canonical_lengths((counts, values))        # (length, code) -> SSSS
    lut = [0x8000 | (63 << 5)] * 1024
    for (l, code), ssss in codes.items():
        total = l + ssss
        if l <= 10 and ssss < 16 and 1 <= total <= 26:
            for i in range(code << (10 - l), (code + 1) << (10 - l)):
                lut[i] = ((32 - total) & 31) | (total << 5) | (ssss << 11)
    return lut


def test_two_byte_lut_entries_decode_like_the_canonical_decoder():
    rng = np.random.default_rng(9)
    for table in (C.NIKON, C.ALT, C.random_huffman_table(rng, 15, skew=1.2)):
        lut = _entry16(*table)
        codes = M.canonical_lengths(table)
        diffs = rng.integers(-6000, 6000, 3000)
        # a bit stream of the differences under this table
        out = []
        for d in diffs:
            d = int(d)
            ssss = 0 if d == 0 else int(abs(d)).bit_length()
            (l, code), = [k for k, v in codes.items() if v == ssss]
            out += [int(ch) for ch in format(code, "0%db" % l)]
            if ssss:
                v = d if d > 0 else d + (1 << ssss) - 1
                out += [int(ch) for ch in format(v, "0%db" % ssss)]
        out += [0] * 64
        pos, got = 0, []
        for _ in diffs:
            w = int("".join(map(str, out[pos:pos + 32])), 2)
            e = lut[w >> 22]
            if e & 0x8000:                               # (a code of more than 10 bits: the general way)
                l = next(l for l in range(11, 17) if (l, w >> (32 - l)) in codes)
                ssss = codes[(l, w >> (32 - l))]
                v = (w >> (32 - l - ssss)) & ((1 << ssss) - 1) if ssss else 0
                total = l + ssss
            else:
                all_ = (1 << (e >> 11)) - 1              # v_bfm_b32 of SSSS
                v = (w >> (e & 31)) & all_
                total = (e >> 5) & 63
                ssss = e >> 11
            all_ = (1 << ssss) - 1
            u = all_ - v
            m = -1 if (u - v) < 0 else 0                 # int32(u - v) >> 31
            got.append(((all_ & m) - u))
            pos += total
        assert got == [int(d) for d in diffs]
