"""Seeded synthetic case builders shared by the oracle and GPU parity tests.

Every builder returns the descriptor (include/rsx.h layout), the input bytes and
the expected output, so a decoder can be checked both against the source image
(round trip) and against the oracle / reference.
"""
import numpy as np

from rawspeed_amd import abi, synth

NIKON = (synth.NIKON14_COUNTS, synth.NIKON14_VALUES)
FULL17 = (synth.FULL17_COUNTS, synth.FULL17_VALUES)
ALT = (synth.ALT_COUNTS, synth.ALT_VALUES)


def smooth_image(rng, h, w, prec=14, sigma=20.0, full_range=False):
    """Random-walk-ish image whose neighbour differences are small."""
    maxv = (1 << prec) - 1
    if full_range:
        return rng.integers(0, maxv + 1, size=(h, w), dtype=np.uint16)
    margin = min(1000, maxv // 4)
    base = rng.integers(margin, maxv - margin)
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    # bounded ramps: wide images must not run into the clip value (a saturated,
    # constant region is a separate, deliberately slow, case for the decoder)
    sx, sy = min(3.0, 0.4 * maxv / w), min(2.0, 0.2 * maxv / h)
    img = base + sx * x + sy * y + rng.normal(0, sigma, size=(h, w))
    return np.clip(img, 0, maxv).astype(np.uint16)


def ljpeg_stream_rows(tile, mcu_w, mcu_h, frame_w, frame_h, rng, prec=14):
    """tile: (tile_h, req_w) uint16 samples to be produced.  Returns the
    (frame_h, frame_w*mcu_w*mcu_h) stream-order sample array; samples the decoder
    discards (beyond req_w / below tile_h) are filled with plausible data."""
    tile_h, req_w = tile.shape
    n = mcu_w * mcu_h
    full_h, full_w = frame_h * mcu_h, frame_w * mcu_w
    canvas = smooth_image(rng, full_h, full_w, prec)
    canvas[:tile_h, :req_w] = tile
    # MCU (k, m) covers canvas rows mcu_h*k.., cols mcu_w*m..; order row-major in MCU
    c = canvas.reshape(frame_h, mcu_h, frame_w, mcu_w).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(c.reshape(frame_h, frame_w * n))


def make_ljpeg_case(rng, img_w, img_h, cpp, tile, mcu, frame=None, tables=(NIKON,),
                    table_index=None, rows_per_ri=0, fix16=False, prec=14,
                    full_range=False, init_pred=None, sigma=20.0):
    """tile = (x, y, w, h) in pixels; mcu = (mcu_w, mcu_h); frame = (w, h) in MCUs."""
    tx, ty, tw, th = tile
    mw, mh = mcu
    n = mw * mh
    req_w = cpp * tw
    if frame is None:
        frame = ((req_w + mw - 1) // mw, th // mh)
    fw, fh = frame
    tile_px = smooth_image(rng, th, req_w, prec, sigma=sigma, full_range=full_range)
    rows = ljpeg_stream_rows(tile_px, mw, mh, fw, fh, rng, prec)
    if table_index is None:
        table_index = [0] * n
    if init_pred is None:
        init_pred = [1 << (prec - 1)] * n
    comp_tables = [tables[i] for i in table_index]
    scan, bits = synth.ljpeg_encode_scan(rows, n, init_pred, comp_tables,
                                         rows_per_ri, fix16)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, ty, tw, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = mw, mh, fw, fh
    d.n_comp = n
    d.rows_per_restart_interval = rows_per_ri if rows_per_ri else fh
    abi.fill_recipe(d, synth.huff_tables(*tables, fix16=fix16), table_index, init_pred)
    # what LJpegDecoder hands over: scan data up to the end of the tile buffer
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8),
                           np.zeros(16, np.uint8)])
    return d, data, tile_px, len(scan)


def random_huffman_table(rng, n_cat=17, skew=None):
    """A random canonical JPEG table (counts[16], values) that has a code for every
    category 0..n_cat-1: Huffman code lengths of random weights (so the Kraft sum
    is exactly 1 before JPEG's reserved all-ones code is accounted for by adding
    one dummy symbol), redrawn until no code is longer than 16 bits."""
    while True:
        if skew is None:
            skew_ = rng.uniform(0.3, 3.0)
        else:
            skew_ = skew
        wts = rng.random(n_cat + 1) ** skew_ + 1e-6
        nodes = [(wt, [i]) for i, wt in enumerate(wts)]   # symbol n_cat = dummy
        length = [0] * (n_cat + 1)
        while len(nodes) > 1:
            nodes.sort(key=lambda t: t[0])
            (w0, s0), (w1, s1) = nodes[0], nodes[1]
            for i in s0 + s1:
                length[i] += 1
            nodes = nodes[2:] + [(w0 + w1, s0 + s1)]
        if max(length) > 16:
            continue
        # the dummy must own the all-ones code: make it (one of) the longest
        lmax = max(length)
        if length[n_cat] != lmax:
            j = length.index(lmax)
            length[j], length[n_cat] = length[n_cat], length[j]
        order = sorted(range(n_cat), key=lambda i: (length[i], rng.random()))
        counts = [0] * 16
        for i in range(n_cat):
            counts[length[i] - 1] += 1
        return counts, order


def cr2_slices(num_slices, slice_w, last_w):
    return [slice_w] * (num_slices - 1) + [last_w]


def cr2_stream_from_image(img, n_comp, frame_w, frame_h, slices):
    """Inverse of the Cr2Decompressor output walk for <N,1,1> when
    frame_h == image height: the stream is strip 0's rows
    top->bottom, then strip 1's ...; returned as (frame_h, frame_w*n_comp) rows."""
    h, w = img.shape
    assert sum(slices) == w
    parts, x = [], 0
    for sw in slices:
        parts.append(img[:, x:x + sw].reshape(-1))
        x += sw
    flat = np.concatenate(parts)
    assert flat.size == frame_w * n_comp * frame_h
    return np.ascontiguousarray(flat.reshape(frame_h, frame_w * n_comp))


def make_cr2_case(rng, img_w, img_h, n_comp, slices, tables=(NIKON,),
                  table_index=None, prec=14, full_range=False, sigma=20.0):
    """<N,1,1> CR2 stream whose LJPEG frame is img_w/n_comp x img_h."""
    frame_w, frame_h = img_w // n_comp, img_h
    img = smooth_image(rng, img_h, img_w, prec, sigma=sigma, full_range=full_range)
    sl = cr2_slices(*slices)
    rows = cr2_stream_from_image(img, n_comp, frame_w, frame_h, sl)
    if table_index is None:
        table_index = [0] * n_comp
    init_pred = [1 << (prec - 1)] * n_comp
    scan, bits = synth.ljpeg_encode_scan(rows, n_comp, init_pred,
                                         [tables[i] for i in table_index])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = n_comp, 1, 1
    d.frame_w, d.frame_h = frame_w, frame_h
    d.num_slices, d.slice_width, d.last_slice_width = slices
    abi.fill_recipe(d, synth.huff_tables(*tables), table_index, init_pred)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8),
                           np.zeros(16, np.uint8)])
    return d, data, img, len(scan)


def cr2_output_tiles(widths, dim_x, dim_y, frame_y):
    """The Cr2OutputTileIterator walk (Cr2DecompressorImpl.h:104-154) in groups:
    every slice is frame_y rows tall and wraps to the next output column when it
    reaches the bottom of the image."""
    tiles, ox, oy = [], 0, 0
    for w in widths:
        left = frame_y
        while left and ox < dim_x:
            h = min(left, dim_y - oy)
            tiles.append((ox, oy, w, h))
            left -= h
            oy += h
            if oy == dim_y:
                oy, ox = 0, ox + w
    return tiles


def make_cr2_sraw_case(rng, ysf, slices, dim_y, dim_x=None, frame_y=None,
                       tables=(NIKON,), table_index=(0, 0, 0), prec=14,
                       full_range=False, with_rows=False, img=None):
    """Canon sRaw <3,2,ysf> stream (Cr2DecompressorImpl.h:250-275): groups of
    gs = 2 + 2*ysf samples (Y.. Cb Cr).  `slices` = (num, width, last_width) in
    GROUPS; the image is dim_x groups (default: sum of the widths) = dim_x*gs
    samples wide, dim_y rows, cpp 1, not CFA.  frame_y = LJPEG frame rows after
    the /Y_S_F (default dim_y; smaller values make slices wrap into further
    output columns); the last frame row may be partial (frame area > image area)."""
    gs = 2 + 2 * ysf
    widths = cr2_slices(*slices)
    if dim_x is None:
        dim_x = sum(widths)
    if frame_y is None:
        frame_y = dim_y
    if img is None:
        img = smooth_image(rng, dim_y, dim_x * gs, prec, full_range=full_range)
    assert img.shape == (dim_y, dim_x * gs)
    parts = [img[y:y + h, x * gs:(x + w) * gs].reshape(-1)
             for (x, y, w, h) in cr2_output_tiles(widths, dim_x, dim_y, frame_y)]
    flat = np.concatenate(parts)
    assert flat.size == dim_x * gs * dim_y, "tiles do not cover the image"
    frame_x = -(-dim_x * dim_y // frame_y)
    row = frame_x * gs
    pad = frame_y * row - flat.size
    if pad:  # never decoded (only dim.area() groups are walked)
        flat = np.concatenate([flat, np.resize(flat, pad)])
    rows = np.ascontiguousarray(flat.reshape(frame_y, row))
    init_pred = [1 << (prec - 1)] * 3
    scan, bits = synth.ljpeg_encode_scan(rows, 3, init_pred,
                                         [tables[i] for i in table_index],
                                         pattern=synth.SRAW_PATTERN[gs])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 3, 2, ysf
    d.frame_w, d.frame_h = frame_x * 2, frame_y * ysf
    d.num_slices = slices[0]
    d.slice_width, d.last_slice_width = slices[1] * 6, slices[2] * 6
    abi.fill_recipe(d, synth.huff_tables(*tables), list(table_index), init_pred)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8),
                           np.zeros(16, np.uint8)])
    if with_rows:
        return d, data, img, len(scan), rows
    return d, data, img, len(scan)
