"""Synthetic input generators (numpy front-end of librsx_synth.so).

Mirrors what the reference's tests/benchmarks do with BitVacuumer* and
PrefixCodeVectorEncoder (bitstreams/BitVacuumer.h, codes/PrefixCodeVectorEncoder.h):
packed-integer strips for UncompressedDecompressor and lossless-JPEG streams /
containers for LJpegDecompressor, Cr2Decompressor and DNG tiles.
"""
import ctypes as C
import os

import numpy as np

from . import abi, build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build.LIB_SYNTH
        if not os.path.exists(path):
            build.build_synth()
        _lib = C.CDLL(path)
        _lib.rsx_synth_pack_rows.restype = C.c_size_t
        _lib.rsx_synth_pack_rows.argtypes = [
            C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
            C.c_void_p]
        _lib.rsx_synth_uniform.argtypes = [C.c_void_p, C.c_size_t, C.c_int,
                                           C.c_uint64]
        _lib.rsx_synth_sensor_image.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_uint64]
        _lib.rsx_synth_ljpeg_encode_scan.restype = C.c_size_t
        _lib.rsx_synth_ljpeg_encode_scan.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
            C.c_size_t, C.c_void_p]
        _lib.rsx_synth_ljpeg_encode_pattern.restype = C.c_size_t
        _lib.rsx_synth_ljpeg_encode_pattern.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
            C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.rsx_synth_nikon_encode.restype = C.c_size_t
        _lib.rsx_synth_nikon_encode.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.rsx_synth_prefix_encode.restype = C.c_size_t
        _lib.rsx_synth_prefix_encode.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.rsx_synth_hasselblad_encode.restype = C.c_size_t
        _lib.rsx_synth_hasselblad_encode.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint, C.c_void_p,
            C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.rsx_synth_sony_arw1_encode.restype = C.c_size_t
        _lib.rsx_synth_sony_arw1_encode.argtypes = [
            C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.rsx_synth_ljpeg_header.restype = C.c_size_t
        _lib.rsx_synth_ljpeg_header.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


# The 14-bit lossless Nikon tree (decompressors/NikonDecompressor.cpp:64-66) is
# the table SURVEY.md 8(d) uses for the synthetic LJPEG streams: 15 categories
# (0..14), so 14-bit data never needs SSSS 15/16.
NIKON14_COUNTS = [0, 1, 4, 2, 2, 3, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0]
NIKON14_VALUES = [7, 6, 8, 5, 9, 4, 10, 3, 11, 12, 2, 0, 1, 13, 14]

# A table with all 17 categories 0..16 (JPEG Annex K luminance-DC shaped, extended
# to 17 codes the way Hasselblad files do, AbstractLJpegDecoder.cpp:249-252).
FULL17_COUNTS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0]
FULL17_VALUES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]

# Short-code-heavy table (different shape from the two above).
ALT_COUNTS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]
ALT_VALUES = [5, 6, 4, 7, 3, 8, 2, 9, 1, 10, 0, 11, 12, 13, 14, 15, 16]


def uniform(n, bits, seed):
    out = np.empty(n, dtype=np.uint16)
    lib().rsx_synth_uniform(out.ctypes.data, n, bits, seed)
    return out


def sensor_image(w, h, prec=14, seed=1):
    out = np.empty((h, w), dtype=np.uint16)
    lib().rsx_synth_sensor_image(out.ctypes.data, w, h, w, prec, seed)
    return out


def pack_rows(samples, bps, order, pitch_bytes=None):
    """samples: (rows, cols) uint16 -> packed strip bytes (rows*pitch)."""
    samples = np.ascontiguousarray(samples, dtype=np.uint16)
    rows, cols = samples.shape
    if pitch_bytes is None:
        pitch_bytes = cols * bps // 8
    out = np.empty(rows * pitch_bytes, dtype=np.uint8)
    n = lib().rsx_synth_pack_rows(order, bps, samples.ctypes.data, cols, cols,
                                  rows, pitch_bytes, out.ctypes.data)
    if n == 0:
        raise ValueError("cannot pack: bad bps/pitch/order combination")
    return out


def _table_ptrs(tables):
    counts = [np.asarray(t[0], dtype=np.uint8) for t in tables]
    values = [np.asarray(t[1], dtype=np.uint8) for t in tables]
    n = len(tables)
    cp = (C.c_void_p * n)(*[c.ctypes.data for c in counts])
    vp = (C.c_void_p * n)(*[v.ctypes.data for v in values])
    nv = (C.c_int * n)(*[len(v) for v in values])
    return counts, values, cp, vp, nv


SRAW_PATTERN = {4: (0, 0, 1, 2), 6: (0, 0, 0, 0, 1, 2)}  # <3,2,1>, <3,2,2>


def ljpeg_encode_scan(stream_rows, n_comp, init_pred, comp_tables,
                      rows_per_ri=0, fix16=False, pattern=None):
    """stream_rows: (rows, frame_w*n_comp) uint16 in stream order.
    comp_tables: one (counts, values) per component.  pattern: component of
    sample s is pattern[s % len(pattern)] (default: s % n_comp; SRAW_PATTERN
    for Canon sRaw groups).  Returns (entropy bytes as np.uint8, symbol bits)."""
    stream_rows = np.ascontiguousarray(stream_rows, dtype=np.uint16)
    rows, row_samples = stream_rows.shape
    if pattern is None:
        pattern = tuple(range(n_comp))
    assert row_samples % len(pattern) == 0 and len(comp_tables) == n_comp
    keep = _table_ptrs(comp_tables)
    _, _, cp, vp, nv = keep
    ip = np.asarray(init_pred, dtype=np.uint16)
    pat = np.asarray(pattern, dtype=np.uint8)
    cap = rows * row_samples * 5 + 4096
    out = np.empty(cap, dtype=np.uint8)
    bits = C.c_uint64(0)
    n = lib().rsx_synth_ljpeg_encode_pattern(
        stream_rows.ctypes.data, row_samples, row_samples, rows, n_comp,
        len(pattern), pat.ctypes.data,
        ip.ctypes.data, cp, vp, nv, rows_per_ri, 1 if fix16 else 0,
        out.ctypes.data, cap, C.byref(bits))
    if n == 0:
        raise ValueError("LJPEG encode failed (category missing from table?)")
    return out[:n].copy(), bits.value


def ljpeg_container(stream_rows, n_comp, prec, comp_slot, slot_tables,
                    rows_per_ri=0, fix16=False, frame_wh=None, samp=None,
                    tail=16, pattern=None):
    """Full SOI..EOI blob.  comp_slot[c] = DHT slot of component c,
    slot_tables[i] = (counts, values) of slot i.  frame_wh overrides the SOF
    (w, h) (e.g. Canon's half-height quirk); default (row_samples/n_comp, rows).
    samp = per-component (H, V) sampling factors, pattern = the component of
    each sample of a group (SRAW_PATTERN) for Canon sRaw scans.
    Returns (blob, scan_offset, scan_bytes, symbol_bits)."""
    stream_rows = np.ascontiguousarray(stream_rows, dtype=np.uint16)
    rows, row_samples = stream_rows.shape
    fw, fh = frame_wh if frame_wh else (row_samples // n_comp, rows)
    init_pred = [1 << (prec - 1)] * n_comp
    scan, bits = ljpeg_encode_scan(
        stream_rows, n_comp, init_pred, [slot_tables[s] for s in comp_slot],
        rows_per_ri, fix16, pattern)
    keep = _table_ptrs(slot_tables)
    _, _, cp, vp, nv = keep
    hdr = np.empty(1024, dtype=np.uint8)
    cs = (C.c_int * n_comp)(*comp_slot)
    hs = vs = None
    if samp is not None:
        hs = (C.c_int * n_comp)(*[s[0] for s in samp])
        vs = (C.c_int * n_comp)(*[s[1] for s in samp])
    ri_mcus = rows_per_ri * fw if rows_per_ri else 0
    nh = lib().rsx_synth_ljpeg_header(hdr.ctypes.data, prec, fw, fh, n_comp, cs,
                                      len(slot_tables), cp, vp, nv, ri_mcus,
                                      hs, vs)
    blob = np.concatenate([hdr[:nh], scan,
                           np.array([0xFF, 0xD9], dtype=np.uint8),
                           np.zeros(tail, dtype=np.uint8)])
    return blob, nh, len(scan), bits


def ljpeg_header(prec, frame_w, frame_h, n_comp, comp_slot, slot_tables):
    """SOI, SOF3, DHT(s), SOS of a lossless-JPEG container (no scan data)."""
    keep = _table_ptrs(slot_tables)
    _, _, cp, vp, nv = keep
    hdr = np.empty(1024, dtype=np.uint8)
    cs = (C.c_int * n_comp)(*comp_slot)
    nh = lib().rsx_synth_ljpeg_header(hdr.ctypes.data, prec, frame_w, frame_h, n_comp, cs,
                                      len(slot_tables), cp, vp, nv, 0, None, None)
    return hdr[:nh].copy()


# NikonDecompressor::nikon_tree (decompressors/NikonDecompressor.cpp:47-66): the
# six Huffman trees of the NEF format, (16 counts, values).  Trees 1 and 4
# ("after split") carry len | shl << 4 values decoded by NikonLASDecompressor.
NIKON_TREE = [
    ([0, 1, 5, 1, 1, 1, 1, 1, 1, 2, 0, 0, 0, 0, 0, 0],
     [5, 4, 3, 6, 2, 7, 1, 0, 8, 9, 11, 10, 12, 0]),  # 12-bit lossy (14 codes: the
                                                       # std::array's zero fill is the 14th value)
    ([0, 1, 5, 1, 1, 1, 1, 1, 1, 2, 0, 0, 0, 0, 0, 0],
     [0x39, 0x5a, 0x38, 0x27, 0x16, 5, 4, 3, 2, 1, 0, 11, 12, 12]),      # ... after split
    ([0, 1, 4, 2, 3, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0],
     [5, 4, 6, 3, 7, 2, 8, 1, 9, 0, 10, 11, 12]),                        # 12-bit lossless
    ([0, 1, 4, 3, 1, 1, 1, 1, 1, 2, 0, 0, 0, 0, 0, 0],
     [5, 6, 4, 7, 8, 3, 9, 2, 1, 0, 10, 11, 12, 13, 14]),                # 14-bit lossy
    ([0, 1, 5, 1, 1, 1, 1, 1, 1, 1, 2, 0, 0, 0, 0, 0],
     [8, 0x5c, 0x4b, 0x3a, 0x29, 7, 6, 5, 4, 3, 2, 1, 0, 13, 14]),       # ... after split
    (NIKON14_COUNTS, NIKON14_VALUES),                                    # 14-bit lossless
]


# PentaxDecompressor::pentax_tree (decompressors/PentaxDecompressor.cpp:46-53)
PENTAX_TREE = ([0, 2, 3, 1, 1, 1, 1, 1, 1, 2, 0, 0, 0, 0, 0, 0],
               [3, 4, 2, 5, 1, 6, 0, 7, 8, 9, 10, 11, 12])


def nikon_encode(img, p_up, table):
    """img: (h, w) uint16 of 15-bit values -> NikonDecompressor MSB stream
    (np.uint8) that decodes to it with uncorrectedRawValues; p_up = the four
    metadata predictors [row0col0, row0col1, row1col0, row1col1]."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    counts = np.asarray(table[0], dtype=np.uint8)
    values = np.asarray(table[1], dtype=np.uint8)
    pu = np.asarray(p_up, dtype=np.int32)
    cap = h * w * 4 + 64
    out = np.empty(cap, dtype=np.uint8)
    bits = C.c_uint64(0)
    n = lib().rsx_synth_nikon_encode(img.ctypes.data, w, w, h, pu.ctypes.data,
                                     counts.ctypes.data, values.ctypes.data,
                                     len(values), out.ctypes.data, cap, C.byref(bits))
    if n == 0:
        raise ValueError("Nikon encode failed (difference too large for the table?)")
    return out[:n].copy(), bits.value


# The encoding table of SamsungV1Decompressor::decompress
# (decompressors/SamsungV1Decompressor.cpp:88-101): (encLen, diffLen) pairs.
SAMSUNG_V1_TAB = [(3, 4), (3, 7), (2, 6), (2, 5), (4, 3), (6, 0), (7, 9), (8, 10), (9, 11),
                  (10, 12), (10, 13), (5, 1), (4, 8), (4, 2)]


def prefix_encode(img, p_up, tab):
    """Like nikon_encode, with the prefix code given as (encLen, diffLen) pairs
    in table-fill order (SamsungV1)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    enc = np.asarray([t[0] for t in tab], dtype=np.uint8)
    dif = np.asarray([t[1] for t in tab], dtype=np.uint8)
    pu = np.asarray(p_up, dtype=np.int32)
    cap = h * w * 4 + 64
    out = np.empty(cap, dtype=np.uint8)
    bits = C.c_uint64(0)
    n = lib().rsx_synth_prefix_encode(img.ctypes.data, w, w, h, pu.ctypes.data,
                                      enc.ctypes.data, dif.ctypes.data, len(tab),
                                      out.ctypes.data, cap, C.byref(bits))
    if n == 0:
        raise ValueError("prefix encode failed")
    return out[:n].copy(), bits.value


def hasselblad_encode(img, init_pred, table):
    """img: (h, w) uint16 -> HasselbladDecompressor stream (MSB32 words, pairs of
    [len code][len code][bits][bits]); table = (counts, values) with the lengths
    0..16 it needs.  Returns (np.uint8 bytes, symbol bits)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    counts = np.asarray(table[0], dtype=np.uint8)
    values = np.asarray(table[1], dtype=np.uint8)
    cap = h * w * 5 + 64
    out = np.empty(cap, dtype=np.uint8)
    bits = C.c_uint64(0)
    n = lib().rsx_synth_hasselblad_encode(img.ctypes.data, w, w, h, init_pred,
                                          counts.ctypes.data, values.ctypes.data,
                                          len(values), out.ctypes.data, cap, C.byref(bits))
    if n == 0:
        raise ValueError("Hasselblad encode failed")
    return out[:n].copy(), bits.value


def sony_arw1_encode(img):
    """img: (h, w) values in the decoder's range 0..4095 (or, to provoke its range
    error, any int16 stored as uint16); SonyArw1Decompressor.cpp:59-93 stream:
    columns right to left, even rows then odd rows, one running predictor."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    cap = h * w * 5 + 64
    out = np.empty(cap, dtype=np.uint8)
    bits = C.c_uint64(0)
    n = lib().rsx_synth_sony_arw1_encode(img.ctypes.data, w, w, h, out.ctypes.data, cap,
                                         C.byref(bits))
    if n == 0:
        raise ValueError("Sony ARW1 encode failed")
    return out[:n].copy(), bits.value


def huff_tables(*pairs, fix16=False):
    return [abi.HuffTable.make(c, v, fix16) for c, v in pairs]
