"""TEST INFRASTRUCTURE: synthetic raw FILES (TIFF containers) for the decoder-level
drop-in tests -- whole files that go through the reference's own front door
(RawParser::getDecoder -> DngDecoder / ArwDecoder -> decodeRaw) in both builds of the
reference (unmodified and GPU-backed).  Only the tags those decoders read are written:
  DngDecoder::decodeRawInternal / decodeData / getTilingDescription / parseCFA
      (decoders/DngDecoder.cpp:448-534, :361-446, :303-359, :230-278)
  ArwDecoder::decodeRawInternal / DecodeLJpeg / DecodeUncompressed
      (decoders/ArwDecoder.cpp:166-260, :296-411)
  Cr2Decoder::decodeNewFormat (Cr2Decoder.cpp:125-209), NefDecoder::decodeRawInternal
      (NefDecoder.cpp:73-138), PefDecoder::decodeRawInternal (PefDecoder.cpp:58-117),
      ThreefrDecoder::decodeRawInternal (ThreefrDecoder.cpp:56-84), SrwDecoder (SrwDecoder.cpp:56-120)
  TiffParser::parse, TiffIFD (parsers/TiffParser.cpp:52-78, tiff/TiffIFD.cpp:46-120)
"""
import struct

import numpy as np

# TIFF field types
BYTE, ASCII, SHORT, LONG, RATIONAL, UNDEFINED = 1, 2, 3, 4, 5, 7
_SIZE = {BYTE: 1, ASCII: 1, SHORT: 2, LONG: 4, RATIONAL: 8, UNDEFINED: 1}

# tags (tiff/TiffTag.h)
NEWSUBFILETYPE, IMAGEWIDTH, IMAGELENGTH, BITSPERSAMPLE, COMPRESSION = 254, 256, 257, 258, 259
PHOTOMETRIC, MAKE, MODEL, STRIPOFFSETS, SAMPLESPERPIXEL = 262, 271, 272, 273, 277
ROWSPERSTRIP, STRIPBYTECOUNTS, TILEWIDTH, TILELENGTH = 278, 279, 322, 323
TILEOFFSETS, TILEBYTECOUNTS, SUBIFDS = 324, 325, 330
CFAREPEATPATTERNDIM, CFAPATTERN = 33421, 33422
DNGVERSION, DNGBACKWARDVERSION, UNIQUECAMERAMODEL = 50706, 50707, 50708
ACTIVEAREA = 50829
SONYRAWIMAGESIZE, SONYCURVE = 0x7038, 0x7010
CANON_CAMERA_SETTINGS, CANON_SENSOR_INFO, CANONCR2SLICE = 0x0001, 0x00E0, 0xC640
CANON_SRAWTYPE, CANONCOLORDATA = 0xC6C5, 0x4001


def _payload(typ, values, en="<"):
    if typ == ASCII:
        b = values.encode() + b"\0"
        return b, len(b)
    if typ in (BYTE, UNDEFINED):
        return bytes(values), len(values)
    if typ == SHORT:
        return struct.pack(en + "%dH" % len(values), *values), len(values)
    if typ == LONG:
        return struct.pack(en + "%dI" % len(values), *values), len(values)
    if typ == RATIONAL:
        flat = [x for pair in values for x in pair]
        return struct.pack(en + "%dI" % len(flat), *flat), len(values)
    raise ValueError(typ)


class Ifd:
    """One IFD: add(tag, type, values); blobs (image data) are placed after all IFDs and
    their offsets patched into the tag given to add_blobs()."""

    def __init__(self):
        self.entries = {}
        self.blobs = None  # (offsets_tag, counts_tag, [bytes])
        self.subs = []
        self.next = None  # the IFD this one chains to ("next IFD" pointer)

    def add(self, tag, typ, values):
        if typ != ASCII and not isinstance(values, (list, tuple, bytes, bytearray)):
            values = [values]
        self.entries[tag] = (typ, values)
        return self

    def add_blobs(self, offsets_tag, counts_tag, blobs):
        self.blobs = (offsets_tag, counts_tag, [bytes(np.asarray(b, np.uint8).tobytes())
                                                for b in blobs])
        return self

    def add_sub(self, ifd):
        self.subs.append(ifd)
        return self


def tiff_file(root, gap=0, big=False):
    """Serialise (little-endian, or big-endian like the files of Nikon and Pentax): header,
    IFDs (depth first) with their out-of-line values, then the blobs (each followed by
    `gap` bytes that belong to nobody)."""
    en = ">" if big else "<"
    ifds = []

    def walk(i):
        ifds.append(i)
        for s in i.subs:
            walk(s)
        if i.next is not None:
            walk(i.next)

    walk(root)
    # sizes: entries incl. the placeholders for blob offsets / counts / sub-IFD pointers
    def table(i):
        e = dict(i.entries)
        if i.blobs:
            n = len(i.blobs[2])
            e[i.blobs[0]] = (LONG, [0] * n)
            e[i.blobs[1]] = (LONG, [len(b) for b in i.blobs[2]])
        if i.subs:
            e[SUBIFDS] = (LONG, [0] * len(i.subs))
        return e

    tables = [table(i) for i in ifds]

    def ifd_bytes(e):
        n = 2 + 12 * len(e) + 4
        for typ, values in e.values():
            p, _ = _payload(typ, values, en)
            if len(p) > 4:
                n += len(p) + (len(p) & 1)
        return n

    pos = 8
    ifd_pos = []
    for e in tables:
        ifd_pos.append(pos)
        pos += ifd_bytes(e)
    blob_pos = {}
    for k, i in enumerate(ifds):
        if i.blobs:
            offs = []
            for b in i.blobs[2]:
                offs.append(pos)
                pos += len(b) + gap
            blob_pos[k] = offs
    total = pos

    out = bytearray(total)
    out[0:8] = (b"MM" if big else b"II") + struct.pack(en + "HI", 42, 8)
    index = {id(i): k for k, i in enumerate(ifds)}
    for k, (i, e) in enumerate(zip(ifds, tables)):
        if i.blobs:
            e[i.blobs[0]] = (LONG, blob_pos[k])
        if i.subs:
            e[SUBIFDS] = (LONG, [ifd_pos[index[id(s)]] for s in i.subs])
        p = ifd_pos[k]
        struct.pack_into(en + "H", out, p, len(e))
        q = p + 2
        extra = p + 2 + 12 * len(e) + 4
        for tag in sorted(e):
            typ, values = e[tag]
            data, count = _payload(typ, values, en)
            struct.pack_into(en + "HHI", out, q, tag, typ, count)
            if len(data) <= 4:
                out[q + 8:q + 8 + len(data)] = data
            else:
                struct.pack_into(en + "I", out, q + 8, extra)
                out[extra:extra + len(data)] = data
                extra += len(data) + (len(data) & 1)
            q += 12
        struct.pack_into(en + "I", out, q, ifd_pos[index[id(i.next)]] if i.next is not None else 0)
        if i.blobs:
            for off, b in zip(blob_pos[k], i.blobs[2]):
                out[off:off + len(b)] = b
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def dng_file(width, height, tile_w, tile_h, blobs, compression=7, bps=16, cpp=1,
             version=(1, 4, 0, 0), strips=False, active_area=None, gap=0):
    """A DNG whose raw IFD is the root IFD.  blobs: one per tile (row-major) -- LJPEG
    SOI..EOI containers (compression 7) or packed strips / tiles (compression 1).
    strips=True: ROWSPERSTRIP = tile_h, STRIPOFFSETS instead of tiles."""
    i = Ifd()
    i.add(NEWSUBFILETYPE, LONG, 0)
    i.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    i.add(BITSPERSAMPLE, SHORT, [bps] * cpp)
    i.add(COMPRESSION, SHORT, compression)
    i.add(PHOTOMETRIC, SHORT, 32803 if cpp == 1 else 34892)
    i.add(MAKE, ASCII, "RSX").add(MODEL, ASCII, "Synthetic")
    i.add(SAMPLESPERPIXEL, SHORT, cpp)
    if cpp == 1:
        i.add(CFAREPEATPATTERNDIM, SHORT, [2, 2])
        i.add(CFAPATTERN, BYTE, [0, 1, 1, 2])
    i.add(DNGVERSION, BYTE, list(version))
    i.add(DNGBACKWARDVERSION, BYTE, [1, 1, 0, 0])
    i.add(UNIQUECAMERAMODEL, ASCII, "RSX Synthetic")
    if active_area:
        i.add(ACTIVEAREA, LONG, list(active_area))  # top, left, bottom, right
    if strips:
        i.add(ROWSPERSTRIP, LONG, tile_h)
        i.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, blobs)
    else:
        i.add(TILEWIDTH, LONG, tile_w).add(TILELENGTH, LONG, tile_h)
        i.add_blobs(TILEOFFSETS, TILEBYTECOUNTS, blobs)
    return tiff_file(i, gap)


def arw_file(width, height, tile_w, tile_h, blobs, bps=14, gap=0):
    """A Sony ARW of the LJPEG-tile kind (ArwDecoder.cpp:296-411: compression 7): root
    IFD with MAKE "SONY", one sub-IFD with the tiles.  The decoder looks the raw IFD up
    by STRIPOFFSETS (:167), so a one-entry strip table pointing at the first tile is
    present as well, as in the camera's files."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(BITSPERSAMPLE, SHORT, bps)
    raw.add(COMPRESSION, SHORT, 7)
    raw.add(PHOTOMETRIC, SHORT, 32803)
    raw.add(SAMPLESPERPIXEL, SHORT, 1)
    raw.add(STRIPOFFSETS, LONG, 8).add(STRIPBYTECOUNTS, LONG, 1)
    raw.add(TILEWIDTH, LONG, tile_w).add(TILELENGTH, LONG, tile_h)
    raw.add(SONYRAWIMAGESIZE, LONG, [width, height])
    raw.add_blobs(TILEOFFSETS, TILEBYTECOUNTS, blobs)
    root = Ifd()
    root.add(MAKE, ASCII, "SONY").add(MODEL, ASCII, "ILCE-RSX")
    root.add_sub(raw)
    return tiff_file(root, gap)


def arw_uncompressed_file(width, height, packed, bps=14):
    """Compression 1 (ArwDecoder::DecodeUncompressed, :262-294): one strip of
    little-endian 16-bit containers (bps 14/12) or packed 8-bit."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(BITSPERSAMPLE, SHORT, bps)
    raw.add(COMPRESSION, SHORT, 1)
    raw.add(PHOTOMETRIC, SHORT, 32803)
    raw.add(SAMPLESPERPIXEL, SHORT, 1)
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [packed])
    root = Ifd()
    root.add(MAKE, ASCII, "SONY").add(MODEL, ASCII, "ILCE-RSX")
    root.add_sub(raw)
    return tiff_file(root)


def arw1_file(width, height, packed, curve_points=(0, 0, 0, 0)):
    """The compressed format of ArwDecoder::decodeRawInternal (:183-260): compression
    32767 whose strip size is NOT width*height*bpp/8 marks SonyArw1Decompressor; the
    decoder then adds 8 rows to the tag's height, so the tag says height - 8."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height - 8)
    raw.add(BITSPERSAMPLE, SHORT, 12)
    raw.add(COMPRESSION, SHORT, 32767)
    raw.add(PHOTOMETRIC, SHORT, 32803)
    raw.add(SAMPLESPERPIXEL, SHORT, 1)
    raw.add(SONYCURVE, SHORT, list(curve_points))
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [packed])
    root = Ifd()
    root.add(MAKE, ASCII, "SONY").add(MODEL, ASCII, "DSLR-RSX")
    root.add_sub(raw)
    return tiff_file(root)


def cr2_sraw_file(sensor_w, sensor_h, blob, slices, ysf, coeffs=(1900, 1024, 1024, 1500)):
    """A Canon sRaw CR2: SRAWType 4 in the fourth IFD, SRAWQuality (CameraSettings[46]) 1 =
    2x2 or 2 = 2x1 subsampling (Cr2Decoder.cpp:511-541), and the white-balance block the
    sRaw interpolation takes its three coefficients from (ColorData[78..81], :563-576).
    After the decode Cr2Decoder runs Cr2sRawInterpolator (version 1, hue 0 without a
    camera database / model id)."""
    i0 = Ifd()
    i0.add(MAKE, ASCII, "Canon").add(MODEL, ASCII, "Canon EOS RSX")
    cs = [0] * 48
    cs[46] = 1 if ysf == 2 else 2
    i0.add(CANON_CAMERA_SETTINGS, SHORT, cs)
    i0.add(CANON_SENSOR_INFO, SHORT, [0, sensor_w, sensor_h, 0, 0, 0, 0, 0])
    cd = [0] * 90
    cd[78:82] = list(coeffs)
    i0.add(CANONCOLORDATA, SHORT, cd)
    i1 = Ifd().add(IMAGEWIDTH, LONG, 160)
    i2 = Ifd().add(IMAGEWIDTH, LONG, 320)
    i3 = Ifd()
    i3.add(COMPRESSION, SHORT, 6)
    i3.add(CANONCR2SLICE, SHORT, list(slices))
    i3.add(CANON_SRAWTYPE, LONG, 4)
    i3.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [blob])
    i0.next, i1.next, i2.next = i1, i2, i3
    return tiff_file(i0)


def cr2_file(width, height, blob, slices):
    """A Canon CR2 of the "new format" (Cr2Decoder::decodeNewFormat, Cr2Decoder.cpp:125-209):
    four chained IFDs, the fourth holds the LJPEG strip and the slice table
    (slices = (n_slices - 1, slice_width, last_slice_width)); SensorInfo / CameraSettings
    -- in a camera file inside the MakerNote, looked up recursively (:126, :520) -- sit in
    IFD0.  CameraSettings has fewer than 47 entries: not subsampled (:528)."""
    i0 = Ifd()
    i0.add(MAKE, ASCII, "Canon").add(MODEL, ASCII, "Canon EOS RSX")
    i0.add(CANON_CAMERA_SETTINGS, SHORT, [0] * 8)
    i0.add(CANON_SENSOR_INFO, SHORT, [0, width, height, 0, 0, 0, 0, 0])
    i1 = Ifd().add(IMAGEWIDTH, LONG, 160)
    i2 = Ifd().add(IMAGEWIDTH, LONG, 320)
    i3 = Ifd()
    i3.add(COMPRESSION, SHORT, 6)
    i3.add(CANONCR2SLICE, SHORT, list(slices))
    i3.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [blob])
    i0.next, i1.next, i2.next = i1, i2, i3
    return tiff_file(i0)


def pef_file(width, height, data, meta):
    """Pentax PEF, compression 65535 (PefDecoder.cpp:58-117): one strip, and the Huffman
    table of the makernote (tag 0x220, type UNDEFINED), found recursively."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(COMPRESSION, LONG, 65535)
    raw.add(PHOTOMETRIC, SHORT, 32803)
    raw.add(0x220, UNDEFINED, bytes(np.asarray(meta, np.uint8).tobytes()))
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [data])
    root = Ifd()
    root.add(MAKE, ASCII, "PENTAX").add(MODEL, ASCII, "PENTAX RSX")
    root.add_sub(raw)
    return tiff_file(root, big=True)  # (the table's fields are read in the file's byte order)


def nef_file(width, height, bps, data, meta):
    """Nikon NEF, compression 34713 (NefDecoder.cpp:73-138): the raw IFD is the one with
    CFAPATTERN, one strip whose size is not that of an uncompressed image, the
    linearisation blob is tag 0x96 (makernote; found recursively)."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(BITSPERSAMPLE, SHORT, bps)
    raw.add(COMPRESSION, LONG, 34713)
    raw.add(PHOTOMETRIC, SHORT, 32803)
    raw.add(CFAREPEATPATTERNDIM, SHORT, [2, 2])
    raw.add(CFAPATTERN, BYTE, [0, 1, 1, 2])
    raw.add(0x96, UNDEFINED, bytes(np.asarray(meta, np.uint8).tobytes()))
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [data])
    root = Ifd()
    root.add(MAKE, ASCII, "NIKON CORPORATION").add(MODEL, ASCII, "NIKON RSX")
    root.add_sub(raw)
    return tiff_file(root, big=True)  # (the blob's fields are read in the file's byte order)


def threefr_file(width, height, blob):
    """Hasselblad 3FR, compression 7 (ThreefrDecoder.cpp:56-84): the raw IFD is the SECOND
    one that has STRIPOFFSETS; its strip is an LJPEG container whose scan holds the
    pair-coded Hasselblad stream."""
    thumb = Ifd()
    thumb.add(IMAGEWIDTH, LONG, 8).add(IMAGELENGTH, LONG, 8)
    thumb.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [np.zeros(64, np.uint8)])
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(COMPRESSION, SHORT, 7)
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [blob])
    root = Ifd()
    root.add(MAKE, ASCII, "Hasselblad").add(MODEL, ASCII, "RSX")
    root.add_sub(thumb).add_sub(raw)
    return tiff_file(root)


def srw_v1_file(width, height, data, bits=12, compression=32772):
    """Samsung SRW, compression 32772 (SrwDecoder.cpp:56-120): SamsungV1Decompressor;
    32773: SamsungV2Decompressor (:121-135)."""
    raw = Ifd()
    raw.add(IMAGEWIDTH, LONG, width).add(IMAGELENGTH, LONG, height)
    raw.add(BITSPERSAMPLE, SHORT, bits)
    raw.add(COMPRESSION, LONG, compression)
    raw.add_blobs(STRIPOFFSETS, STRIPBYTECOUNTS, [data])
    root = Ifd()
    root.add(MAKE, ASCII, "SAMSUNG").add(MODEL, ASCII, "RSX")
    root.add_sub(raw)
    return tiff_file(root)
