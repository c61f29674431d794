"""CPU model of the reconstruction arithmetic the fused LJPEG decode kernel uses
(rsx_ljpeg_direct.hip) against the sequential definition of the predictor
(SURVEY A.4 / LJpegDecompressor.cpp:184-251, 326-332).

The kernel never sees a row as a unit: a lane owns the symbols that START in one
64-byte subsequence of the entropy stream (any number, any alignment to rows or
components).  What it is given:

  * per subsequence: the sums of its differences by RELATIVE phase (k mod N for
    the k-th symbol of the subsequence; the synchronisation kernel does not know
    the absolute symbol index when it decodes), turned into absolute-component
    sums by a rotation once the symbol-count prefix is known,
  * P(i) = running sum over the WHOLE stream of the differences of i's
    component (no reset at row starts) from an exclusive scan of those sums,
  * per row r the offsets O(r, c) = init[c] + V(r, c) - E(r, c) with
        E(r, c) = P just before the row's first symbol (component c),
        F(r, c) = P at the row's c-th symbol,
        V(r, c) = sum_{r' < r} (F(r', c) - E(r', c))   (the vertical chain),
and then X(i) = P(i) + O(row(i), comp(i))  (mod 2^16).
"""
import numpy as np
import pytest


def sequential(D, rows, row_samples, n, init):
    """X[r][s] = (s >= n ? X[r][s-n] : (r ? X[r-1][s] : init[s])) + D[r][s]  mod 2^16"""
    D = D.reshape(rows, row_samples).astype(np.int64)
    X = np.zeros_like(D)
    for r in range(rows):
        for s in range(row_samples):
            if s >= n:
                p = X[r, s - n]
            elif r:
                p = X[r - 1, s]
            else:
                p = init[s]
            X[r, s] = (p + D[r, s]) & 0xFFFF
    return X.reshape(-1)


def rot_fields(rel, f, n):
    """abs[c] = rel[(c - f) mod n]"""
    return [rel[(c - f) % n] for c in range(n)]


def slot_model(D, rows, row_samples, n, init, counts):
    total = rows * row_samples
    assert sum(counts) == total
    # ---- synchronisation kernel: relative-phase sums per subsequence ----------
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rel = []
    for f0, cnt in zip(first, counts):
        s = [0] * n
        for k in range(cnt):
            s[k % n] = (s[k % n] + int(D[f0 + k])) & 0xFFFF
        rel.append(s)
    # ---- scan: absolute-component exclusive prefix per subsequence ------------
    pbase = []
    run = [0] * n
    for f0, s in zip(first, rel):
        pbase.append(list(run))
        a = rot_fields(s, int(f0) % n, n)
        run = [(run[c] + a[c]) & 0xFFFF for c in range(n)]

    def decode_from(slot, upto):
        """P (per component) just before global symbol `upto`, walking from `slot`"""
        p = list(pbase[slot])
        i = int(first[slot])
        while i < upto:
            p[i % n] = (p[i % n] + int(D[i])) & 0xFFFF
            i += 1
        return p

    # ---- row-edge kernel: E, F per row; then the row offsets ------------------
    ends = np.cumsum(counts)
    O = np.zeros((rows, n), dtype=np.int64)
    V = [0] * n
    for r in range(rows):
        t = r * row_samples
        slot = int(np.searchsorted(ends, t, side="right"))
        E = decode_from(slot, t)
        F = list(E)
        for c in range(n):
            F[c] = (E[c] + int(D[t + c])) & 0xFFFF
        for c in range(n):
            O[r, c] = (init[c] + V[c] - E[c]) & 0xFFFF
            V[c] = (V[c] + F[c] - E[c]) & 0xFFFF
    # ---- decode kernel: every subsequence on its own --------------------------
    X = np.zeros(total, dtype=np.int64)
    for slot, (f0, cnt) in enumerate(zip(first, counts)):
        # the lane works in relative phases: run_rel[q] = P of component (f0 + q) % n
        run_rel = [pbase[slot][(int(f0) + q) % n] for q in range(n)]
        for k in range(cnt):
            i = int(f0) + k
            q = k % n
            run_rel[q] = (run_rel[q] + int(D[i])) & 0xFFFF
            r = i // row_samples
            X[i] = (run_rel[q] + O[r, i % n]) & 0xFFFF
    return X


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(5, 8), (7, 24), (3, 120), (16, 12)])
def test_slot_formulation_matches_sequential(n, shape):
    rows, mcus = shape
    row_samples = mcus * n
    rng = np.random.default_rng(100 * n + rows)
    total = rows * row_samples
    D = rng.integers(-32768, 32768, size=total)
    init = [int(v) for v in rng.integers(0, 65536, size=n)]
    # ragged subsequences, including empty ones and ones spanning several rows
    counts = []
    left = total
    while left:
        c = int(rng.choice([0, 1, 2, 3, 5, 17, 61, 200]))
        c = min(c, left)
        counts.append(c)
        left -= c
    want = sequential(D, rows, row_samples, n, init)
    got = slot_model(D, rows, row_samples, n, init, counts)
    assert np.array_equal(got, want)


def merge_fixup(D, n, old_start, new_start, lens):
    """Two-pointer re-synchronisation of one subsequence: the parse from `new_start`
    and the recorded parse from `old_start` are advanced alternately (always the one
    that is behind) until they stand on the same bit; from there on they are the same
    parse, so count and relative-phase sums of the recorded tail are reused -- the
    tail's phases shift by (new symbols - old symbols) before the merge point.
    (The kernel variant of this was measured slower than a plain re-decode and is not in
    librsx; the model stays because the phase rotation it checks is the identity
    lj_rot_fields<N> applies when slot-relative sums become workgroup-relative ones.)
    lens[p] = length of the symbol that starts at bit p; D[p] = its difference."""
    end = len(lens)

    def full(start):
        p, k, s = start, 0, [0] * n
        while p < end:
            s[k % n] = (s[k % n] + D[p]) & 0xFFFF
            k += 1
            p += lens[p]
        return k, s, p - end

    old_cnt, old_sums, old_exit = full(old_start)
    pa, pb = new_start, old_start
    na = nb = 0
    pre_new, pre_old = [0] * n, [0] * n
    merged = False
    while pa < end:
        if pa == pb:
            merged = True
            break
        if pb < pa and pb < end:
            pre_old[nb % n] = (pre_old[nb % n] + D[pb]) & 0xFFFF
            nb += 1
            pb += lens[pb]
        else:
            pre_new[na % n] = (pre_new[na % n] + D[pa]) & 0xFFFF
            na += 1
            pa += lens[pa]
    if merged:
        tail_old = [(old_sums[q] - pre_old[q]) & 0xFFFF for q in range(n)]
        sums = [(pre_new[q] + tail_old[(q + nb - na) % n]) & 0xFFFF for q in range(n)]
        return na + old_cnt - nb, sums, old_exit
    return na, pre_new, pa - end


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_two_pointer_merge_fixup(n):
    rng = np.random.default_rng(7 + n)
    for _ in range(200):
        nbits = 512
        lens = rng.integers(2, 33, size=nbits)
        D = [int(v) for v in rng.integers(0, 65536, size=nbits)]
        old_start, new_start = (int(v) for v in rng.integers(0, 32, size=2))
        got = merge_fixup(D, n, old_start, new_start, lens)
        # truth: a full parse from the new start
        p, k, s = new_start, 0, [0] * n
        while p < nbits:
            s[k % n] = (s[k % n] + D[p]) & 0xFFFF
            k += 1
            p += int(lens[p])
        assert got == (k, s, p - nbits)
