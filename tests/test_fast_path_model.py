"""CPU models of the arithmetic the single-pass LJPEG kernel rests on
(rawspeed_amd/csrc/rsx_ljpeg_fast.hip).  No GPU: these pin the algebra, the GPU tests
pin the kernel against the oracle and the reference build.

1. Predictor state across workgroups.  The kernel cuts the stream of differences into
   workgroup chunks at ARBITRARY symbol indices (also in the middle of a row's first MCU)
   and carries (T, Vc) = (left-neighbour values, first-MCU values of the last started
   row) from chunk to chunk as a transfer  T' = f ? Vc + a : T + a,  Vc' = Vc + v  that
   composes associatively (decoupled look-back).  Pixels = Ploc + C(row, component).
   Compared with the reference's recurrence
   (LJpegDecompressor.cpp:184-251, 326-332; Cr2DecompressorImpl.h:437-465).
2. JPEG EXTEND without a shift left, as the fast loop computes it.
3. The bit window: position kept as Pn = -32 * pos - 32, image delayed by one bit,
   v_alignbit with the low 5 bits of Pn >> 5.
"""
import numpy as np
import pytest

M = 0xFFFF


def reference_pixels(D, RS, N, init):
    """X[r][s] = (s >= N ? X[r][s-N] : (r ? X[r-1][s] : init[s])) + D[r][s]  (mod 2^16)"""
    rows = len(D) // RS
    X = np.zeros(len(D), np.int64)
    for r in range(rows):
        for s in range(RS):
            i = r * RS + s
            if s >= N:
                p = X[i - N]
            elif r:
                p = X[i - RS]
            else:
                p = init[s]
            X[i] = (p + D[i]) & M
    return X


def chunk_transfer(D, base, lim, RS, N):
    """What one workgroup publishes for the symbols [base, lim): per component the flag
    f, a, v -- and what it keeps for itself: Ploc and the C table without the incoming
    state.  Follows the kernel's steps 5-8."""
    n = lim - base
    # Ploc(i): running sum of i's component over the chunk's symbols up to i
    Ploc = np.zeros(n, np.int64)
    run = {}
    for e in range(n):
        c = (base + e) % N
        run[c] = (run.get(c, 0) + D[base + e]) & M
        Ploc[e] = run[c]
    S = [0] * N  # sums by component
    for e in range(n):
        S[(base + e) % N] = (S[(base + e) % N] + D[base + e]) & M
    r0 = base // RS
    r_end = (lim - 1) // RS if n else r0
    nr = r_end - r0 + 1
    E = np.zeros((nr, N), np.int64)
    Dm = np.zeros((nr, N), np.int64)
    present = np.zeros((nr, N), bool)
    for t in range(nr):
        for c in range(N):
            i = (r0 + t) * RS + c
            if base <= i < lim:
                e = i - base
                pv = Ploc[e - N] if e >= N else 0
                E[t, c] = pv
                Dm[t, c] = (Ploc[e] - pv) & M
                present[t, c] = True
    Vex = np.zeros((nr, N), np.int64)
    acc = np.zeros(N, np.int64)
    for t in range(nr):
        Vex[t] = acc
        acc = (acc + Dm[t]) & M
    Vsum = acc
    Cloc = (Vex - E) & M
    f = [False] * N
    a = [0] * N
    for c in range(N):
        tl = -1
        il = r_end * RS + c
        if base <= il < lim:
            tl = nr - 1
        elif nr >= 2 and (r_end - 1) * RS + c >= base:
            tl = nr - 2
        if tl >= 0 and n:
            f[c] = True
            a[c] = (Cloc[tl, c] + S[c]) & M
        else:
            a[c] = S[c]
    return dict(f=f, a=a, v=[int(x) for x in Vsum], Ploc=Ploc, Cloc=Cloc, present=present,
                r0=r0, nr=nr)


def compose(h, g):
    """first h, then g"""
    N = len(h["f"])
    return dict(f=[h["f"][c] or g["f"][c] for c in range(N)],
                a=[((h["v"][c] if g["f"][c] else h["a"][c]) + g["a"][c]) & M for c in range(N)],
                v=[(h["v"][c] + g["v"][c]) & M for c in range(N)])


def apply(x, T, V):
    N = len(T)
    return ([((V[c] if x["f"][c] else T[c]) + x["a"][c]) & M for c in range(N)],
            [(V[c] + x["v"][c]) & M for c in range(N)])


@pytest.mark.parametrize("N,RS,rows", [(2, 16, 9), (1, 7, 11), (4, 24, 7), (2, 6, 40), (4, 8, 33)])
def test_predictor_state_transfer(N, RS, rows):
    rng = np.random.default_rng([N, RS, rows])
    D = rng.integers(0, 1 << 16, size=rows * RS)
    init = [int(x) for x in rng.integers(0, 1 << 16, size=N)]
    want = reference_pixels(D, RS, N, init)
    for trial in range(30):
        # chunk boundaries anywhere, also inside a first MCU and several rows per chunk
        k = int(rng.integers(1, 9))
        cuts = sorted(set(int(x) for x in rng.integers(1, len(D), size=k)))
        bounds = [0] + cuts + [len(D)]
        chunks = [chunk_transfer(D, bounds[q], bounds[q + 1], RS, N) for q in range(len(bounds) - 1)]
        got = np.zeros(len(D), np.int64)
        T, V = list(init), list(init)
        for q, ch in enumerate(chunks):
            base, lim = bounds[q], bounds[q + 1]
            # the state before the chunk two ways: chained, and by a look-back that
            # composes the transfers of ALL predecessors first (associativity)
            if q:
                g = chunks[q - 1]
                for p in range(q - 2, -1, -1):
                    g = compose(chunks[p], g)
                T2, V2 = apply(g, list(init), list(init))
                assert (T2, V2) == (T, V)
            C = np.zeros((ch["nr"], N), np.int64)
            for t in range(ch["nr"]):
                for c in range(N):
                    C[t, c] = (V[c] + ch["Cloc"][t, c]) & M if ch["present"][t, c] else T[c]
            for e in range(lim - base):
                i = base + e
                got[i] = (ch["Ploc"][e] + C[i // RS - ch["r0"], i % N]) & M
            T, V = apply(ch, T, V)
        assert np.array_equal(got, want), (trial, bounds)


def extend_reference(bits, ssss):
    """JPEG F.2.2.1 EXTEND of the ssss difference bits (as an integer), 16-bit result"""
    if ssss == 0:
        return 0
    if bits < (1 << (ssss - 1)):
        bits += (-1 << ssss) + 1
    return bits & M


def test_extend_without_shift_left():
    """u = all - v;  m = (u - v) >> 31 (arithmetic);  diff = (all & m) - u"""
    for ssss in range(0, 16):
        allm = (1 << ssss) - 1
        for v in range(0, 1 << ssss, max(1, (1 << ssss) // 257)):
            u = (allm - v) & 0xFFFFFFFF
            t = (u - v) & 0xFFFFFFFF
            m = 0xFFFFFFFF if t & 0x80000000 else 0
            diff = ((allm & m) - u) & M
            assert diff == extend_reference(v, ssss), (ssss, v)
        for v in {0, allm, allm >> 1, min(allm, (allm >> 1) + 1)}:
            u = (allm - v) & 0xFFFFFFFF
            m = 0xFFFFFFFF if (u - v) & 0x80000000 else 0
            assert ((allm & m) - u) & M == extend_reference(v, ssss)


def test_window_of_the_delayed_image():
    """32 stream bits at bit position pos = v_alignbit(d0', d1', (Pn >> 5) & 31) on the image
    delayed by one bit, with Pn = -32 * pos - 32 and the window's dwords wi, wi + 1 of the
    DELAYED image found at byte offset (Pn & ~1023) of a reversed [dword][256 lanes] layout."""
    rng = np.random.default_rng(5)
    words = rng.integers(0, 1 << 32, size=18, dtype=np.uint64)
    bits = "".join(format(int(w), "032b") for w in words)
    delayed_bits = "0" + bits[:-1]
    delayed = [int(delayed_bits[32 * k:32 * k + 32], 2) for k in range(18)]
    for pos in range(0, 512):
        want = int(bits[pos:pos + 32], 2)
        Pn = (-32 * pos - 32) & 0xFFFFFFFF
        row_off = Pn & ~1023 & 0xFFFFFFFF              # = -1024 * (wi + 1)
        wi = (-(row_off - (1 << 32)) // 1024) - 1
        assert wi == pos // 32
        d0, d1 = delayed[wi], delayed[wi + 1]
        s = (Pn >> 5) & 31
        got = (((d0 << 32) | d1) >> s) & 0xFFFFFFFF    # v_alignbit_b32(d0, d1, s)
        assert got == want, pos
        # the advance: Pn -= 32 * total, live while Pn > Pend (unsigned)
        tot = 9
        assert ((Pn - 32 * tot) & 0xFFFFFFFF) == ((-32 * (pos + tot) - 32) & 0xFFFFFFFF)


def test_fast_table_entry_fields():
    """entry.x = (32 - total) | total << 5 (| 1 << 31 for symbols the 10-bit LUT does not
    cover), entry.y = 2^SSSS - 1:  v = (w >> (x & 31)) & y are the difference bits."""
    rng = np.random.default_rng(6)
    for _ in range(2000):
        cl = int(rng.integers(1, 11))
        ssss = int(rng.integers(0, 16))
        total = cl + ssss
        w = int(rng.integers(0, 1 << 32))
        x = (32 - total) | (total << 5)
        y = (1 << ssss) - 1
        v = (w >> (x & 31)) & y
        want = (w >> (32 - cl - ssss)) & y if ssss else 0
        assert v == want
        assert (x >> 5) & 63 == total and not (x & 0x80000000)
