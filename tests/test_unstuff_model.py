"""The branch-free un-stuffing of K0 (rsx_ljpeg.hip: lj_fix_regs) restated on
Python integers and checked against a byte-by-byte walk of BitStreamerJPEG's
rules (bitstreams/BitStreamerJPEG.h:106-183) on random slots: FF 00 -> FF; FF xx
(xx != 0) ends the data; bytes past `valid` do not exist.  The device code is
checked end to end by the GPU parity tests; this pins the mask ALGORITHM."""
import numpy as np

SLOT, OWN = 80, 64      # 64 own bytes + 16 bytes of look-ahead


def walk(b, prev, valid):
    """Reference walk: (kept bytes, own_bits, own_drops, marker_off)."""
    kept, own_kept, drops, marker = [], 0, 0, -1
    i = 0
    drop_next = prev == 0xFF and valid > 0 and b[0] == 0
    while i < valid:
        if drop_next:
            drop_next = False
            if i < OWN:
                drops += 1
            i += 1
            continue
        c = b[i]
        if c == 0xFF:
            nxt = b[i + 1] if i + 1 < min(valid, SLOT) else 0
            if nxt != 0:
                if i < OWN:
                    marker = i
                break
            drop_next = True
        kept.append(c)
        if i < OWN:
            own_kept += 1
        i += 1
    return kept, 8 * own_kept, drops, marker


def masks(b, prev, valid):
    """lj_fix_regs: 80-bit masks, bit i = byte i."""
    full = (1 << SLOT) - 1
    ff = sum(1 << i for i in range(SLOT) if b[i] == 0xFF)
    z = sum(1 << i for i in range(SLOT) if b[i] == 0)
    v = (1 << valid) - 1
    nz = ((v & ~z) >> 1) & full                   # byte i + 1 exists and is not zero
    m = ff & v & nz
    e = (m & -m).bit_length() - 1 if m else valid
    below = (1 << e) - 1
    pf = ((ff << 1) | (1 if prev == 0xFF else 0)) & full
    d = z & pf & below
    k = below & ~d
    own = (1 << OWN) - 1
    kept = [b[i] for i in range(SLOT) if (k >> i) & 1]
    return kept, 8 * bin(k & own).count("1"), bin(d & own).count("1"), (e if m and e < OWN else -1)


def test_unstuff_masks_match_the_byte_walk():
    rng = np.random.default_rng(2718)
    n_marker = n_drop = 0
    for trial in range(20000):
        mode = trial % 5
        b = rng.integers(0, 256, size=SLOT, dtype=np.uint8)
        if mode >= 1:       # FF-heavy, with stuffing bytes after most of them
            pos = rng.integers(0, SLOT, size=rng.integers(1, 12))
            b[pos] = 0xFF
            for p in pos:
                if p + 1 < SLOT and rng.random() < (0.95 if mode < 4 else 0.5):
                    b[p + 1] = 0
        if mode == 3:       # runs: FF 00 FF 00, FF FF, 00 00
            s = int(rng.integers(0, SLOT - 8))
            b[s:s + 8] = rng.choice([[0xFF, 0, 0xFF, 0, 0xFF, 0, 0, 0], [0xFF, 0xFF, 0, 0, 0xFF, 0, 1, 2],
                                     [0, 0, 0xFF, 0, 0, 0xFF, 0, 0xFF]])
        prev = int(rng.choice([0, 0xFF, 0x12]))
        valid = int(rng.choice([SLOT, SLOT, SLOT, 64, 63, 65, 1, 0, int(rng.integers(0, SLOT + 1))]))
        bl = [int(x) for x in b]
        want, got = walk(bl, prev, valid), masks(bl, prev, valid)
        assert want == got, (trial, bl, prev, valid, want, got)
        n_marker += want[3] >= 0
        n_drop += want[2] > 0
    assert n_marker > 500 and n_drop > 2000
