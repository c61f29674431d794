"""CPU model of the single-pass decode planned for the next round: what a
workgroup has to publish, and how a workgroup turns its predecessors' records into pixels,
against the sequential definition of the predictor (LJpegDecompressor.cpp:184-251,
:326-332).  Nothing here runs on the GPU yet; the model pins the arithmetic so that the
kernel can be written against it.

A workgroup g holds the symbols that start in its slots -- cnt[g] of them, any alignment to
rows or components -- and, after ONE decode, knows only things relative to its own first
symbol:
  S[g][q]   sums of its differences by relative phase q = k mod N (k-th symbol of g)
Look-back 1 walks g-1, g-2, ... and accumulates, RELATIVE TO g's first symbol,
  base[g]   = sum of cnt over the predecessors
  Pb[g][q]  = P before g's first symbol for the component that has phase q in g
A predecessor's sums are by ITS phases; a predecessor that ends c symbols before g's first
symbol has phase p <-> g's phase (p - (c + its cnt)) ... i.e. a rotation by the symbols in
between, which the walk knows.  Only when the walk ends is base[g] absolute.
With the base g knows which of its symbols open a row (i mod RS < N) and publishes
  R[g][c]     sum of its row-opening differences of component c
  Rtail[g][c] the part of R that belongs to the LAST row start inside g
  Elast[g]    P (all components) just before that last row start, if there is one
Look-back 2 finds the nearest predecessor h with a row start -- the row that is open at g's
first symbol began there: E = Elast[h], V = sum of R before h + R[h] - Rtail[h].
Then   X(i) = P(i) + init[c] + V(row(i), c) - E(row(i), c)        (mod 2^16),
because F(r, c) - E(r, c) is the row's first difference of component c.
"""
import numpy as np
import pytest

from test_direct_recon_model import sequential


def single_pass(D, rows, RS, N, init, counts, slots_per_group):
    total = rows * RS
    assert sum(counts) == total
    M = 0xFFFF
    # ---- the groups and what each knows after its one decode -----------------------------
    groups = [counts[i:i + slots_per_group] for i in range(0, len(counts), slots_per_group)]
    cnt = [sum(g) for g in groups]
    first_true = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(int)  # (not known to a group)
    S = []
    for g, c in enumerate(cnt):
        s = [0] * N
        for k in range(c):
            s[k % N] = (s[k % N] + int(D[first_true[g] + k])) & M
        S.append(s)
    # ---- look-back 1: base and P before the group, by the group's own phases --------------
    base, Pb = [], []
    for g in range(len(cnt)):
        b, acc = 0, [0] * N
        for h in range(g - 1, -1, -1):  # nearest predecessor first
            # h's k-th symbol lies b + cnt[h] - k symbols before g's first one:
            # its phase p is g's phase (p - (b + cnt[h])) mod N
            shift = (b + cnt[h]) % N
            for p in range(N):
                q = (p - shift) % N
                acc[q] = (acc[q] + S[h][p]) & M
            b += cnt[h]
        base.append(b)
        Pb.append(acc)
    assert base == list(first_true)
    # ---- with the base: row-opening sums and the P before the group's last row start ------
    R, Rtail, Elast = [], [], []
    for g in range(len(cnt)):
        r_sum, r_tail = [0] * N, [0] * N
        p = [Pb[g][(c - base[g]) % N] for c in range(N)]  # by absolute component now
        e_last = None
        for k in range(cnt[g]):
            i = base[g] + k
            if i % RS == 0:
                e_last = list(p)
                r_tail = [0] * N  # opening differences of the group's LAST row start
            if i % RS < N:
                r_sum[i % RS] = (r_sum[i % RS] + int(D[i])) & M
                if e_last is not None:
                    r_tail[i % RS] = (r_tail[i % RS] + int(D[i])) & M
            p[i % N] = (p[i % N] + int(D[i])) & M
        R.append(r_sum)
        Rtail.append(r_tail)
        Elast.append(e_last)
    # ---- look-back 2 and the pixels ----------------------------------------------------
    X = np.zeros(total, dtype=np.int64)
    for g in range(len(cnt)):
        if cnt[g] == 0:
            continue
        # the row that is open at the group's first symbol started in the nearest
        # predecessor h that holds a row start: V for it = the opening differences of the
        # rows ABOVE it = everything before h plus h's own minus that last row's share
        Vrow, E, Vrun = None, None, [0] * N
        for h in range(g - 1, -1, -1):
            if Vrow is None and Elast[h] is not None:
                E = list(Elast[h])
                Vrow = [(R[h][c] - Rtail[h][c]) & M for c in range(N)]
                for hh in range(h):
                    Vrow = [(Vrow[c] + R[hh][c]) & M for c in range(N)]
            Vrun = [(Vrun[c] + R[h][c]) & M for c in range(N)]
        p = [Pb[g][(c - base[g]) % N] for c in range(N)]
        for k in range(cnt[g]):
            i = base[g] + k
            c = i % N
            if i % RS == 0:
                E = list(p)
                Vrow = list(Vrun)  # every opening difference so far belongs to a row above
            p[c] = (p[c] + int(D[i])) & M
            if i % RS < N:
                Vrun[c] = (Vrun[c] + int(D[i])) & M
            X[i] = (p[c] + init[c] + Vrow[c] - E[c]) & M
    return X


@pytest.mark.parametrize("n", [1, 2, 4])
@pytest.mark.parametrize("shape", [(5, 8), (7, 24), (3, 120), (16, 12)])
@pytest.mark.parametrize("spg", [1, 3, 7])
def test_single_pass_matches_sequential(n, shape, spg):
    rows, mcus = shape
    RS = mcus * n
    rng = np.random.default_rng([300, n, rows, spg])
    total = rows * RS
    D = rng.integers(-32768, 32768, size=total)
    init = [int(v) for v in rng.integers(0, 65536, size=n)]
    counts, left = [], total
    while left:
        c = min(int(rng.choice([0, 1, 2, 3, 5, 17, 61, 200])), left)
        counts.append(c)
        left -= c
    want = sequential(D, rows, RS, n, init)
    got = single_pass(D, rows, RS, n, init, counts, spg)
    assert np.array_equal(got, want)
