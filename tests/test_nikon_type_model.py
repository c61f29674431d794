"""CPU models of what the single-pass kernel's Nikon-type instantiation (round 6) rests on.
No GPU: numpy restatements of the device logic against the serial loop of the reference
(NikonDecompressor.cpp:518-560: pUp1 / pUp2 by row parity, pLeft1 / pLeft2 by column parity,
clampBits(., 15), setWithLookUp's dither -- common/RawImage.h:335-353).

1. The look-back's transfers with FOUR fields -- the two column parities of the even stream rows and
   of the odd ones -- composed over any cut of the stream into workgroups give the serial predictor
   state (T' = f ? Vc + a : T + a, Vc' = Vc + v; rsx_ljpeg_fast.hip: XferT, lj_fast_kernel's NK branch).
2. Bit 15 is a sound tell-tale: with sums mod 2^16, "some value so far has bit 15 set" is exactly
   "some int value so far is outside 0..32767", whatever the differences (|d| <= 2^15).
3. The dither generator by position: seed * 15700^(y W) * 15700^x mod (15700 * 2^16 - 1), then steps,
   is the serial multiply-with-carry sequence.
"""
import numpy as np
import pytest

M16 = 0xFFFF


def serial_nikon(diffs, W, p_up):
    """ints, as the reference sums them: diffs (H, W) -> values (H, W) before the clamp"""
    H = diffs.shape[0]
    up = [[int(p_up[0]), int(p_up[1])], [int(p_up[2]), int(p_up[3])]]
    out = np.zeros((H, W), np.int64)
    for y in range(H):
        left = [0, 0]
        for x in range(W):
            c = x & 1
            if x < 2:
                up[y & 1][c] += int(diffs[y, x])
                left[c] = up[y & 1][c]
            else:
                left[c] += int(diffs[y, x])
            out[y, x] = left[c]
    return out


def workgroup_model(diffs, W, p_up, cuts):
    """The kernel's way.  Every workgroup [cuts[k], cuts[k + 1]) of the symbol stream works out from
    ITS symbols only:
      Ploc(i)    running sum of the differences of i's column parity, from 0 at the workgroup's start;
      for a row r whose first symbol of parity c lies in it: D(r, c) that difference, E(r, c) = Ploc
                 in front of it, vex(r, c) = sum of D(r', c) over the rows r' < r of r's PARITY that
                 start here (the four-field exclusive scan), Cloc(r, c) = vex - E;
      its transfer in four fields k = c + 2 * (row parity):  v[k] = sum of D over the rows of that
                 parity that start here;  S[c] = the workgroup's total of parity c;  for the LAST row
                 tl(c) that starts here f[k] = 1, a[k] = Cloc(tl, c) + S[c] at k = c + 2 * (tl & 1)
                 and a = S[c] in the other half (a component that starts no row here: both halves);
    and from the state in front of it (Vc_in, T_in, four fields each):
      C(r, c)  = Vc_in[c + 2 (r & 1)] + Cloc(r, c)    for a row that starts here,
               = T_in[c + 2 (r0 & 1)]                  for the row r0 that is open when it starts;
      pixel(i) = Ploc(i) + C(row, column parity);
      T_out = a + (f ? Vc_in : T_in),  Vc_out = Vc_in + v.           All of it mod 2^16."""
    H = diffs.shape[0]
    flat = diffs.reshape(-1).astype(np.int64)
    out = np.zeros(flat.size, np.int64)
    Vc = [int(p) & M16 for p in p_up]     # (init_pred: pUp by stream-row parity, [c + 2 * parity])
    T = list(Vc)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi <= lo:
            continue
        run = [0, 0]
        ploc = np.zeros(hi - lo, np.int64)
        D, E = {}, {}
        for i in range(lo, hi):
            r, x = divmod(i, W)
            c = x & 1
            if x < 2:
                D[(r, c)], E[(r, c)] = int(flat[i]), run[c]
            run[c] = (run[c] + int(flat[i])) & M16
            ploc[i - lo] = run[c]
        r0 = lo // W
        vex, v, cloc = [0, 0, 0, 0], [0, 0, 0, 0], {}
        for (r, c) in sorted(D):
            k = c + 2 * (r & 1)
            cloc[(r, c)] = (vex[k] - E[(r, c)]) & M16
            vex[k] = (vex[k] + D[(r, c)]) & M16
            v[k] = vex[k]
        f, a = [0, 0, 0, 0], [run[0], run[1], run[0], run[1]]
        for c in (0, 1):
            rows = [r for (r, cc) in D if cc == c]
            if rows:
                tl = max(rows)
                k = c + 2 * (tl & 1)
                f[k] = 1
                a[k] = (cloc[(tl, c)] + run[c]) & M16
        for i in range(lo, hi):
            r, x = divmod(i, W)
            c = x & 1
            const = (Vc[c + 2 * (r & 1)] + cloc[(r, c)]) & M16 if (r, c) in D else T[c + 2 * (r0 & 1)]
            out[i] = (ploc[i - lo] + const) & M16
        T = [(a[k] + (Vc[k] if f[k] else T[k])) & M16 for k in range(4)]
        Vc = [(Vc[k] + v[k]) & M16 for k in range(4)]
    return out.reshape(H, W)


@pytest.mark.parametrize("seed", range(12))
def test_four_field_transfers_give_the_serial_predictor(seed):
    rng = np.random.default_rng([71, seed])
    W = 2 * int(rng.integers(1, 40))
    H = int(rng.integers(1, 30))
    diffs = rng.integers(-300, 300, size=(H, W))
    p_up = rng.integers(0, 16384, size=4)
    want = serial_nikon(diffs, W, p_up) & M16
    n = W * H
    for trial in range(6):
        k = int(rng.integers(1, 12))
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=k)]))
        got = workgroup_model(diffs, W, p_up, cuts)
        assert np.array_equal(got, want), (seed, trial, W, H, cuts)


def test_cuts_inside_the_first_pair_and_single_symbol_workgroups():
    rng = np.random.default_rng(72)
    W, H = 6, 7
    diffs = rng.integers(-50, 50, size=(H, W))
    p_up = [100, 200, 300, 400]
    want = serial_nikon(diffs, W, p_up) & M16
    assert np.array_equal(workgroup_model(diffs, W, p_up, list(range(W * H + 1))), want)
    for cut in range(1, W * H):
        assert np.array_equal(workgroup_model(diffs, W, p_up, [0, cut, W * H]), want), cut


@pytest.mark.parametrize("seed", range(8))
def test_bit_15_is_a_sound_tell_tale(seed):
    """sums mod 2^16 against the reference's ints: the FIRST value outside 0..32767 as an int is also
    the first with bit 15 set mod 2^16 -- so "no bit 15 anywhere" means "every int value inside", and a
    set bit hands the stream to the int route before anything depends on a wrong value"""
    rng = np.random.default_rng([73, seed])
    W, H = 2 * int(rng.integers(2, 30)), int(rng.integers(2, 20))
    scale = int(rng.choice([40, 2000, 32768]))
    diffs = rng.integers(-scale, scale, size=(H, W))  # |d| <= 2^15: SSSS <= 15 (16: -32768)
    p_up = rng.integers(0, 32768, size=4)
    ints = serial_nikon(diffs, W, p_up).reshape(-1)
    mod = (ints & M16)
    outside = (ints < 0) | (ints > 32767)
    bit15 = (mod & 0x8000) != 0
    # decode order is raster order; a value depends on earlier values of its chain only, and the chains
    # (left neighbours of the same parity, first pairs of the rows of the same parity) run forward
    if outside.any():
        first = int(np.argmax(outside))
        assert bit15[first]
        assert not bit15[:first].any()
    else:
        assert not bit15.any()
        assert np.array_equal(mod, ints)


def test_bit_15_with_the_largest_differences():
    """+-32767 and -32768 steps from the edges of the range: the cases the induction's bound is about"""
    for start, d in ((0, -32768), (0, 32767), (32767, 1), (32767, 32767), (32767, -32768), (1, -2)):
        v = start + d
        inside = 0 <= v <= 32767
        assert (((v & M16) & 0x8000) == 0) == inside, (start, d)


def test_dither_state_by_position():
    A, Mm = 15700, 15700 * 65536 - 1
    rng = np.random.default_rng(74)
    for _ in range(20):
        seed = int(rng.integers(0, 1 << 24))
        W = int(rng.integers(2, 300))
        n = W * int(rng.integers(1, 12))
        st, serial = seed, []
        for _ in range(n):
            serial.append(st)
            st = 15700 * (st & 65535) + (st >> 16)
        for _ in range(30):
            i = int(rng.integers(0, n))
            y, x = divmod(i, W)
            by_pos = seed * pow(A, y * W, Mm) % Mm * pow(A, x, Mm) % Mm
            assert by_pos == serial[i], (seed, W, i)
        # (the states stay below m after the first step: the jump's canonical residue IS the state)
        assert max(serial[1:], default=0) < Mm
