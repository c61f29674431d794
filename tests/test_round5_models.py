"""CPU models of three pieces of device logic that changed in round 5 (no GPU):

* look-back 1 of the single-pass kernel (rsx_ljpeg_fast.hip, lb1_walk): a pass asks for the 16
  nearest predecessors' records first and 64 a pass behind them, one wavefront, instead of 256
  by four -- the claim is that the walk arrives at the state of the SERIAL fold of the
  workgroups' transfers whatever mix of LOCAL / inclusive records it meets, and reports
  "blocked" exactly when a record it needs is missing;
* lj_scan_kernel's first-pass scan (rsx_ljpeg.hip, lj_scan_first_pass): a thread owns K
  consecutive workgroups (K = 4 or 16), one block scan per 256 * K of them -- against the
  plain exclusive scan, ragged sizes included;
* the stuffing bytes in front of the last symbol's slot and between it and the end of data,
  counted 16 bytes a lane with a byte mask at the end of data (the pre-loaded tail of
  lj_scan_kernel) -- against a byte-by-byte count, and the consumed-bytes rule built on it
  against BitStreamerJPEG's refill arithmetic."""
import numpy as np
import pytest

P = 65521  # (the transfers of the model: x -> a * x + b mod P, composition is not commutative)


def compose(outer, inner):
    """outer after inner"""
    return (outer[0] * inner[0] % P, (outer[0] * inner[1] + outer[1]) % P)


def apply(t, x):
    return (t[0] * x + t[1]) % P


def lb1_walk_model(b, first_block, local, incl, init, win0=16):
    """One call of lb1_walk<N, 1> without its retry loop: 'blocked' where the kernel sleeps and
    starts over.  local[i]: workgroup i's transfer or None (not published yet); incl[i]: the
    state behind workgroup i or None.  Lane k of a pass looks at workgroup pos - k."""
    g = (1, 0)          # the workgroups nearer than the window, composed
    pos, win = b - 1, win0
    for _ in range(10000):
        lanes = []
        for k in range(64):
            idx = pos - k
            inwin = k < win
            real = idx >= first_block
            loc = real and inwin and local[idx] is not None
            pre = (real and inwin and incl[idx] is not None) or (not real)
            if not real:
                loc = False
            lanes.append((idx, real, inwin, loc and not pre, pre and inwin))
        m_pre = [k for k in range(64) if lanes[k][4]]
        f = m_pre[0] if m_pre else win
        if not all(lanes[k][3] for k in range(f)):
            return "blocked"
        h = (1, 0)      # lanes 0 .. f-1: the farther one first, the nearest last
        for k in range(f - 1, -1, -1):
            h = compose(local[lanes[k][0]], h)
        g = compose(g, h)
        if f < win:
            idx, real = lanes[f][0], lanes[f][1]
            return apply(g, incl[idx] if real else init)
        pos -= win
        win = 64
    raise AssertionError("no end")


@pytest.mark.parametrize("seed", range(40))
def test_look_back_window_of_16_then_64_equals_the_serial_fold(seed):
    rng = np.random.default_rng([5001, seed])
    first_block = int(rng.integers(0, 50))
    n = int(rng.integers(1, 400))
    init = int(rng.integers(0, P))
    T = {first_block + i: (int(rng.integers(1, P)), int(rng.integers(0, P))) for i in range(n)}
    state, s = {}, init
    for i in range(first_block, first_block + n):
        s = apply(T[i], s)
        state[i] = s
    for _ in range(20):
        b = first_block + int(rng.integers(0, n))
        # a frontier: everything at or in front of `done` has its inclusive state, the LOCAL
        # transfers reach further; some records in between are late
        done = b - 1 - int(rng.integers(0, 140))
        late = set(int(x) for x in rng.integers(first_block, b + 1, int(rng.integers(0, 3)))) \
            if rng.integers(0, 3) == 0 else set()
        local = {i: (T[i] if i not in late else None) for i in T}
        incl = {i: (state[i] if i <= done and i not in late else None) for i in T}
        want = init if b == first_block else state[b - 1]
        for win0 in (16, 64, 4):
            got = lb1_walk_model(b, first_block, local, incl, init, win0)
            # the serial answer: the nearest inclusive state, every LOCAL transfer behind it present
            k = b - 1
            while k >= first_block and incl[k] is None:
                k -= 1
            need = range(k + 1, b)
            if all(local[i] is not None for i in need):
                assert got == want, (b, first_block, done, late, win0)
            else:
                assert got == "blocked", (b, first_block, done, late, win0)


def scan_first_pass_model(v, K):
    """lj_scan_first_pass<K> over one stream: exclusive bases, 256 threads, K workgroups each"""
    nb, base, carry = len(v), 0, 0
    out = np.zeros(nb, np.int64)
    while base < nb:
        n_here = min(nb - base, 256 * K)
        kk = (n_here + 255) // 256
        run = np.zeros(256, np.int64)
        for t in range(256):
            for q in range(K):
                i = base + t * kk + q
                if q < kk and i < nb:
                    run[t] += v[i]
        excl_t = carry + np.concatenate([[0], np.cumsum(run)[:-1]])
        for t in range(256):
            e = excl_t[t]
            for q in range(K):
                i = base + t * kk + q
                if q < kk and i < nb:
                    out[i] = e
                    e += v[i]
        carry += run.sum()
        base += 256 * K
    return out, carry


@pytest.mark.parametrize("nb", [1, 2, 255, 256, 257, 680, 1023, 1024, 1025, 1930, 3200, 4096, 4097, 9000])
def test_scan_by_threads_of_k_consecutive_workgroups(nb):
    rng = np.random.default_rng([5002, nb])
    v = rng.integers(0, 40000, nb)
    want = np.concatenate([[0], np.cumsum(v)[:-1]])
    # (the kernel: K = 4 while at most 1024 workgroups are left, 16 otherwise)
    got, base, carry = np.zeros(nb, np.int64), 0, 0
    while base < nb:
        K = 4 if nb - base <= 1024 else 16
        part, c = scan_first_pass_model(v[base:base + 256 * K], K)
        got[base:base + 256 * K] = part + carry
        carry += c
        base += 256 * K
    assert np.array_equal(got, want)
    assert carry == v.sum()


def zero_bytes(d):
    """lj_zero_bytes: 0x80 per zero byte of a dword, exact"""
    d &= 0xFFFFFFFF
    return (~(((d & 0x7F7F7F7F) + 0x7F7F7F7F) | d | 0x7F7F7F7F)) & 0xFFFFFFFF


def drops_by_pieces(data, r0, slot_phys, M):
    """the pre-loaded tail of lj_scan_kernel: stuffing bytes (00 behind FF) of [r0, slot_phys)
    and [slot_phys, M), 16-byte pieces, the bytes at and behind M masked out"""
    a = b = 0
    in_bytes = len(data)
    for u in range(4):
        for tid in range(256):
            p0 = r0 + (u * 256 + tid) * 16
            if p0 >= M:
                continue
            n = 0
            if p0 + 16 <= in_bytes:
                prev = data[p0 - 1] if p0 > 0 else 0
                for k in range(4):
                    q = p0 + 4 * k
                    d = int.from_bytes(bytes(data[q:q + 4]), "little")
                    live = 0x80808080 if q + 4 <= M else (0 if q >= M else 0x80808080 >> (8 * (q + 4 - M)))
                    z, f = zero_bytes(d), zero_bytes(~d)
                    n += bin(z & live & (((f << 8) & 0xFFFFFFFF) | (0x80 if prev == 0xFF else 0))).count("1")
                    prev = d >> 24
            else:
                prev = data[p0 - 1] if p0 > 0 else 0
                for q in range(p0, min(M, in_bytes)):
                    n += 1 if data[q] == 0 and prev == 0xFF else 0
                    prev = data[q]
            if p0 < slot_phys:
                a += n
            else:
                b += n
    return a, b


@pytest.mark.parametrize("seed", range(30))
def test_stuffing_bytes_by_16_byte_pieces_with_a_mask_at_the_end_of_data(seed):
    rng = np.random.default_rng([5003, seed])
    R, Pslot = 255 * 64, 64
    n = int(rng.integers(200, 3 * R))
    data = rng.integers(0, 256, n, dtype=np.uint8)
    # plenty of FF 00 pairs, FF FF 00, 00 at piece boundaries
    for pos in rng.integers(1, n - 2, n // 12):
        data[pos] = 0xFF
        data[pos + 1] = 0x00
    for pos in range(15, n - 1, 16):
        if rng.integers(0, 4) == 0:
            data[pos], data[pos + 1] = 0xFF, 0x00
    M = int(rng.integers(1, n + 1))
    lbs = M // R
    r0 = lbs * R
    slot_phys = r0 + Pslot * int(rng.integers(0, (M - r0) // Pslot + 1))
    assert r0 <= slot_phys <= M <= r0 + R

    def count(lo, hi):
        return sum(1 for q in range(lo, hi) if data[q] == 0 and q > 0 and data[q - 1] == 0xFF)

    assert drops_by_pieces(data, r0, slot_phys, M) == (count(r0, slot_phys), count(slot_phys, M))


def test_consumed_rule_when_the_last_refill_touches_the_marker():
    """K7's rule (lj_consumed_body) as the kernel's tail applies it: with c the
    un-stuffed bit offset of the last symbol's start, K = ceil(c / 32) + 1 refills of 4 data
    bytes have happened; D data bytes lie in front of the marker at M.  4 K > D: the stream
    position is the marker's offset (or runs past the buffer's end by 4 K - D without one)."""
    for c, D, M, has_marker, in_bytes, want in [
        (0, 3, 3, True, 10, 3),            # a tiny scan: the first refill already meets the marker
        (31, 8, 9, True, 20, None),        # K = 2, 4 K = 8 == D: NOT this case (position inside the data)
        (33, 8, 9, True, 20, 9),           # K = 3: 12 > 8
        (64, 10, 10, False, 10, 10 + 2),   # no marker: past the end by 4 K - D = 12 - 10
    ]:
        K = (c + 31) // 32 + 1
        if 4 * K > D:
            got = M if has_marker else in_bytes + (4 * K - D)
            assert got == want, (c, D, M)
        else:
            assert want is None
