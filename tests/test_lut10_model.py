"""CPU model of the two code-table forms the synchronisation kernels use
(rawspeed_amd/csrc/rsx_host.cpp build_device_table, rsx_ljpeg_bits.h lj_stage_tables10 /
lj_slow_entry / TabLds10): the 11-bit LUT with the Annex-F search for longer codes, and
the 10-bit LUT derived from it on the device -- every other entry, 11-bit codes turned
into "not in the LUT" -- with the search starting at 11 bits.  Both must find the same
symbol for every 16-bit prefix, and that symbol must be the canonical code's.  Also the
packed 32-bit slot record (rec_make / rec_st / rec_su / rec_cn)."""
import numpy as np
import pytest

import cases as C


def canonical(counts, values):
    """[(code, length, value)] of a JPEG table (Annex C)."""
    out, code, k = [], 0, 0
    for l in range(1, 17):
        for _ in range(counts[l - 1]):
            out.append((code, l, values[k]))
            code += 1
            k += 1
        code <<= 1
    return out


def device_table(counts, values, lut_bits=11):
    """build_device_table: LUT entry = len | ssss << 5 | (len + ssss) << 10, 0 = longer
    than the LUT; max_code / val_offset per length."""
    lut = np.zeros(1 << lut_bits, np.uint16)
    max_code = [None] * 18
    val_offset = [0] * 18
    code, k = 0, 0
    for l in range(1, 17):
        n = counts[l - 1]
        if n:
            val_offset[l] = (code - k) & 0xFFFF
            max_code[l] = code + n - 1
            for _ in range(n):
                ssss = values[k]
                total = l + (0 if ssss == 16 else ssss)
                if l <= lut_bits:
                    lo = code << (lut_bits - l)
                    lut[lo:lo + (1 << (lut_bits - l))] = l | (ssss << 5) | (total << 10)
                code += 1
                k += 1
        code <<= 1
    max_len = max(l for l in range(1, 17) if counts[l - 1])
    return lut, max_code, val_offset, max_len


def slow_entry(w16, first_len, max_code, val_offset, values, max_len):
    for l in range(first_len, max_len + 1):
        c = w16 >> (16 - l)
        if max_code[l] is not None and c <= max_code[l]:
            ssss = values[(c - val_offset[l]) & 0xFFFF]
            return l | (ssss << 5) | ((l + (0 if ssss == 16 else ssss)) << 10)
    return 0


def stage_lut10(lut11):
    """lj_stage_tables10"""
    e = lut11[0::2].copy()
    e[(e & 31) > 10] = 0
    return e


@pytest.mark.parametrize("seed", range(24))
def test_lut10_finds_what_lut11_finds(seed):
    rng = np.random.default_rng([77, seed])
    n_cat = int(rng.choice([9, 13, 15, 17]))
    counts, values = C.random_huffman_table(rng, n_cat, skew=float(rng.choice([0.3, 1.0, 3.0])))
    lut11, max_code, val_offset, max_len = device_table(counts, values)
    lut10 = stage_lut10(lut11)
    truth = {}
    for code, l, v in canonical(counts, values):
        truth[(code, l)] = v
    prefixes = range(0, 1 << 16, 1) if seed < 4 else rng.integers(0, 1 << 16, size=4096)
    for w in prefixes:
        w = int(w)
        e11 = int(lut11[w >> 5])
        if (e11 & 31) == 0:
            e11 = slow_entry(w, 12, max_code, val_offset, values, max_len)
        e10 = int(lut10[w >> 6])
        if (e10 & 31) == 0:
            e10 = slow_entry(w, 11, max_code, val_offset, values, max_len)
        assert e10 == e11
        if e11:
            l = e11 & 31
            assert truth[(w >> (16 - l), l)] == (e11 >> 5) & 31


def test_slot_record_round_trip():
    ST_ERR = 1 << 9
    rng = np.random.default_rng(5)
    for _ in range(2000):
        su = int(rng.integers(0, 64)) | (int(rng.integers(0, 8)) << 6) | (ST_ERR if rng.random() < 0.1 else 0)
        st = int(rng.integers(0, 64)) | (int(rng.integers(0, 8)) << 6) | (ST_ERR if rng.random() < 0.1 else 0)
        cn = int(rng.integers(0, 513))
        r = (st & 0x3FF) | ((su & 0x3FF) << 10) | (cn << 20)
        assert r < (1 << 32)
        assert (r & 0x3FF, (r >> 10) & 0x3FF, r >> 20) == (st, su, cn)
