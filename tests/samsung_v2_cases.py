"""TEST INFRASTRUCTURE: a writer of SamsungV2Decompressor streams
(decompressors/SamsungV2Decompressor.cpp:87-338) for the oracle / reference pin of that
codec (not served by the GPU library yet).  The writer makes every choice the format
allows -- reference-pixel mode ("motion") per block, scale changes, explicit or relative
difference lengths, skipped blocks -- at random, tracks the image the DECODER will
reconstruct (clamping included) and codes each block against it, so the decoded image is
known exactly."""
import numpy as np


class Msb32Writer:
    """BitStreamerMSB32: 32-bit little-endian words, most significant bit first."""

    def __init__(self):
        self.words, self.acc, self.n = [], 0, 0

    def put(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0
        for k in range(bits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> k) & 1)
            self.n += 1
            if self.n == 32:
                self.words.append(self.acc)
                self.acc, self.n = 0, 0

    def bytes_used(self):
        return 4 * len(self.words) + (self.n + 7) // 8

    def finish(self, align=16):
        """the row's bytes: whole 32-bit words, padded to `align`"""
        if self.n:
            self.words.append(self.acc << (32 - self.n))
        b = np.array(self.words, dtype="<u4").view(np.uint8)
        pad = (-b.size) % align
        return np.concatenate([b, np.zeros(pad, np.uint8)])


MOTION_OFFSET = [-4, -2, -2, 0, 0, 2, 4]
MOTION_AVG = [0, 0, 1, 0, 1, 0, 0]


def _baseline(img, row, col, motion, init_val, width):
    if motion == 7:
        if col == 0:
            return [init_val] * 16
        return [int(img[row, col + (i & 1) - 2]) for i in range(16)]
    base = []
    for i in range(16):
        rr, rc = row, col + i + MOTION_OFFSET[motion]
        if (row + i) & 1:
            rr -= 2
        else:
            rr -= 1
            rc += -1 if (i & 1) else 1
        if rc < 0 or rc >= width or (MOTION_AVG[motion] and rc + 2 >= width):
            return None
        if MOTION_AVG[motion]:
            base.append((int(img[rr, rc]) + int(img[rr, rc + 2]) + 1) >> 1)
        else:
            base.append(int(img[rr, rc]))
    return base


def encode(rng, target, bits, optflags=0, init_val=None, allow_scale=True, rows_out=None):
    """target: (h, w) wanted values (w % 16 == 0).  Returns (stream incl. the 16-byte
    header, the image a decoder reconstructs -- equal to `target` wherever the quantisation
    scale is 0)."""
    h, w = target.shape
    assert w % 16 == 0
    hi = (1 << bits) - 1
    if init_val is None:
        init_val = int(rng.integers(0, 1 << 14))
    hdr = Msb32Writer()
    for v, n in ((0x100, 16), (0, 4), (bits - 1, 4), (0, 4), (0, 4), (w, 16), (h, 16), (0, 16),
                 (0, 4), (optflags, 4), (0, 8), (0, 8), (0, 8), (0, 2), (init_val, 14)):
        hdr.put(v, n)
    out = [hdr.finish(align=16)]
    img = np.zeros((h, w), np.int64)
    for row in range(h):
        wr = Msb32Writer()
        motion, scale = 7, 0
        mode = [[7, 7] if row < 2 else [4, 4] for _ in range(3)]
        for col in range(0, w, 16):
            if not (optflags & 4) and col % 64 == 0:
                pick = int(rng.integers(0, 4)) if allow_scale else 0
                if pick == 3:
                    scale = int(rng.integers(0, 6))
                    wr.put(3, 2)
                    wr.put(scale, 12)
                else:
                    new = scale + (0, -2, 2)[pick]
                    if new < 0:
                        pick, new = 0, scale
                    wr.put(pick, 2)
                    scale = new
            # reference pixels
            cands = [7] if row < 2 else [m for m in range(8)]
            while True:
                m = int(rng.choice(cands))
                if optflags & 2 and m not in (3, 7):
                    continue
                base = _baseline(img, row, col, m, init_val, w)
                if base is not None:
                    break
            if optflags & 2:
                wr.put(1 if m == 3 else 0, 1)
            elif m == motion:
                wr.put(1, 1)
            else:
                wr.put(0, 1)
                wr.put(m, 3)
            motion = m
            # differences (in the shuffled order of the stream), quantised by the scale
            q = 2 * scale + 1
            want = [int(target[row, col + i]) - base[i] for i in range(16)]
            coded = [int(np.floor((d - scale) / q + 0.5)) for d in want]
            lmax = min(15, bits + 1)  # the longest difference field the decoder accepts
            coded = [max(-(1 << (lmax - 1)), min((1 << (lmax - 1)) - 1, c)) for c in coded]
            stream_order = [0] * 16
            for i in range(16):
                p = ((i % 8) << 1) - (i >> 3) + 1 if row % 2 else ((i % 8) << 1) + (i >> 3)
                stream_order[i] = coded[p]
            skip = all(c == 0 for c in coded) and not (optflags & 1) and rng.random() < 0.7
            if not (optflags & 1):
                wr.put(1 if skip else 0, 1)
            lens = [0] * 4
            if not skip:
                need = []
                for g in range(4):
                    vals = stream_order[4 * g:4 * g + 4]
                    n = 0
                    while any(not (-(1 << (n - 1)) <= v < (1 << (n - 1))) if n else v != 0
                              for v in vals):
                        n += 1
                    need.append(n)
                flags = []
                sim = [list(m_) for m_ in mode]
                for g in range(4):
                    colornum = (g >> 1) if row % 2 else ((g >> 1) + 2) % 3
                    cur = sim[colornum][0]
                    opts = [3]
                    if cur >= need[g]:
                        opts.append(0)
                    if cur + 1 >= need[g] and cur + 1 <= bits + 1:
                        opts.append(1)
                    if cur >= 1 and cur - 1 >= need[g]:
                        opts.append(2)
                    f = int(rng.choice(opts))
                    n = {0: cur, 1: cur + 1, 2: cur - 1}.get(f)
                    if f == 3:
                        n = int(rng.integers(need[g], lmax + 1))
                    flags.append((f, n))
                    sim[colornum][0] = sim[colornum][1]
                    sim[colornum][1] = n
                for f, _ in flags:
                    wr.put(f, 2)
                for g, (f, n) in enumerate(flags):
                    if f == 3:
                        wr.put(n, 4)
                    lens[g] = n
                mode = sim
                for i in range(16):
                    n = lens[i >> 2]
                    if n:
                        wr.put(stream_order[i] & ((1 << n) - 1), n)
            # what the decoder stores
            for i in range(16):
                c = 0 if skip else coded[i]
                v = base[i] + c * q + scale
                img[row, col + i] = min(max(v, 0), hi)
        out.append(wr.finish(align=16))
        if rows_out is not None:
            rows_out.append(out[-1])  # the row's bytes, padded to the next 16-byte boundary
    return np.concatenate(out + [np.zeros(16, np.uint8)]), img.astype(np.uint16)
