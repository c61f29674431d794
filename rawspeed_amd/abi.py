"""ctypes mirror of include/rsx.h (the C-ABI of the decompression core).

Only plain C types cross the boundary.  These structures are shared by the
product binding (rawspeed_amd/capi.py) and by the test-only bindings of the
oracle (tests/oracle_lib.py), which use the same descriptors.
"""
import ctypes as C

import numpy as np

RSX_ABI_VERSION = 4

# rsx_status
RSX_OK = 0
RSX_ERR_INVALID_ARG = 1
RSX_ERR_IO = 2
RSX_ERR_BAD_HUFFMAN_CODE = 3
RSX_ERR_RESTART_MARKER = 4
RSX_ERR_INPUT_OVERFLOW = 5
RSX_ERR_DEVICE = 6
RSX_ERR_UNSUPPORTED = 7
RSX_ERR_NOMEM = 8
RSX_ERR_TILE_ERRORS = 9
RSX_ERR_VALUE_RANGE = 10
RSX_ERR_VALUE_RANGE = 10

STATUS_NAMES = {
    0: "RSX_OK", 1: "RSX_ERR_INVALID_ARG", 2: "RSX_ERR_IO",
    3: "RSX_ERR_BAD_HUFFMAN_CODE", 4: "RSX_ERR_RESTART_MARKER",
    5: "RSX_ERR_INPUT_OVERFLOW", 6: "RSX_ERR_DEVICE", 7: "RSX_ERR_UNSUPPORTED",
    8: "RSX_ERR_NOMEM", 9: "RSX_ERR_TILE_ERRORS",
    10: "RSX_ERR_VALUE_RANGE",
}

# rsx_bit_order == rawspeed::BitOrder (bitstreams/BitStreams.h:27-35)
ORDER_LSB, ORDER_MSB, ORDER_MSB16, ORDER_MSB32, ORDER_JPEG = range(5)

RSX_MAX_CODE_VALUES = 162
RSX_MAX_COMPONENTS = 4


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("pitch_bytes", C.c_uint32),
                ("dim_x", C.c_int32), ("dim_y", C.c_int32),
                ("cpp", C.c_int32), ("is_cfa", C.c_int32)]


class UnpackDesc(C.Structure):
    _fields_ = [("crop_x", C.c_int32), ("crop_y", C.c_int32),
                ("crop_w", C.c_int32), ("crop_h", C.c_int32),
                ("input_pitch_bytes", C.c_int32),
                ("bits_per_pixel", C.c_int32), ("bit_order", C.c_int32)]


(UNPACK_8BIT_RAW, UNPACK_12BIT_WITH_CONTROL, UNPACK_12BIT_UNPACKED_LEFT_ALIGNED,
 UNPACK_8BIT_LOOKUP) = range(4)


class UnpackVariantDesc(C.Structure):
    _fields_ = [("variant", C.c_int32), ("big_endian", C.c_int32),
                ("w", C.c_int32), ("h", C.c_int32), ("lut", C.c_uint16 * 256)]

    def set_lut(self, lut):
        for i, v in enumerate(lut):
            self.lut[i] = int(v)
        return self


class HuffTable(C.Structure):
    _fields_ = [("n_codes_per_length", C.c_uint8 * 16),
                ("code_values", C.c_uint8 * RSX_MAX_CODE_VALUES),
                ("n_code_values", C.c_uint8), ("fix_dng_bug16", C.c_uint8)]

    @classmethod
    def make(cls, counts, values, fix_dng_bug16=False):
        t = cls()
        assert len(counts) == 16 and len(values) <= RSX_MAX_CODE_VALUES
        for i, c in enumerate(counts):
            t.n_codes_per_length[i] = c
        for i, v in enumerate(values):
            t.code_values[i] = v
        t.n_code_values = len(values)
        t.fix_dng_bug16 = 1 if fix_dng_bug16 else 0
        return t


class LJpegDesc(C.Structure):
    _fields_ = [("tile_x", C.c_int32), ("tile_y", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("mcu_w", C.c_int32), ("mcu_h", C.c_int32),
                ("frame_w", C.c_int32), ("frame_h", C.c_int32),
                ("n_comp", C.c_int32),
                ("rows_per_restart_interval", C.c_int32),
                ("init_pred", C.c_uint16 * RSX_MAX_COMPONENTS),
                ("table_index", C.c_uint8 * RSX_MAX_COMPONENTS),
                ("n_tables", C.c_int32),
                ("tables", HuffTable * RSX_MAX_COMPONENTS)]


class Cr2Desc(C.Structure):
    _fields_ = [("n_comp", C.c_int32), ("x_s_f", C.c_int32),
                ("y_s_f", C.c_int32),
                ("frame_w", C.c_int32), ("frame_h", C.c_int32),
                ("num_slices", C.c_int32), ("slice_width", C.c_int32),
                ("last_slice_width", C.c_int32),
                ("init_pred", C.c_uint16 * RSX_MAX_COMPONENTS),
                ("table_index", C.c_uint8 * RSX_MAX_COMPONENTS),
                ("n_tables", C.c_int32),
                ("tables", HuffTable * RSX_MAX_COMPONENTS)]


class NikonDesc(C.Structure):
    _fields_ = [("bits_ps", C.c_int32), ("split", C.c_int32),
                ("p_up", (C.c_int32 * 2) * 2),
                ("uncorrected_raw_values", C.c_int32), ("curve_size", C.c_int32),
                ("curve", C.c_void_p), ("tables", HuffTable * 2)]

    def set_curve(self, curve):
        """Keeps the numpy array alive on the descriptor."""
        self._curve = np.ascontiguousarray(curve, dtype=np.uint16)
        self.curve = self._curve.ctypes.data
        self.curve_size = self._curve.size


class PentaxDesc(C.Structure):
    _fields_ = [("table", HuffTable)]


class PentaxJob(C.Structure):
    _fields_ = [("desc", PentaxDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class SamsungV1Desc(C.Structure):
    _fields_ = [("bits", C.c_int32), ("n_entries", C.c_int32),
                ("enc_len", C.c_uint8 * 32), ("diff_len", C.c_uint8 * 32)]

    @classmethod
    def make(cls, tab, bits=12):
        d = cls()
        d.bits, d.n_entries = bits, len(tab)
        for i, (e, l) in enumerate(tab):
            d.enc_len[i], d.diff_len[i] = e, l
        return d


class SamsungV1Job(C.Structure):
    _fields_ = [("desc", SamsungV1Desc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class SamsungV2Desc(C.Structure):
    _fields_ = [("bit_depth", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("optflags", C.c_uint32), ("init_val", C.c_uint32)]

    @classmethod
    def from_header(cls, hdr):
        """The fields SamsungV2Decompressor's constructor reads from the 16-byte header
        (SamsungV2Decompressor.cpp:106-132; BitStreamerMSB32: little-endian 32-bit words,
        most significant bit first).  Returns (desc, raw optflags)."""
        w = [int.from_bytes(bytes(hdr[4 * k:4 * k + 4]), "little") for k in range(4)]
        bits = "".join(format(x, "032b") for x in w)
        pos = [0]

        def get(n):
            v = int(bits[pos[0]:pos[0] + n], 2)
            pos[0] += n
            return v
        d = cls()
        get(16), get(4)
        d.bit_depth = get(4) + 1
        get(4), get(4)
        d.width, d.height = get(16), get(16)
        get(16), get(4)
        flags = get(4)
        get(8), get(8), get(8), get(2)
        d.init_val = get(14)
        d.optflags = flags
        return d, flags


class SamsungV2Job(C.Structure):
    _fields_ = [("desc", SamsungV2Desc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class SonyArw1Job(C.Structure):
    _fields_ = [("in_offset", C.c_uint64), ("in_bytes", C.c_uint64),
                ("img_offset", C.c_uint64), ("img", Image)]


class SrawDesc(C.Structure):
    _fields_ = [("version", C.c_int32), ("subsampling_y", C.c_int32),
                ("sraw_coeffs", C.c_int32 * 3), ("hue", C.c_int32)]

    @classmethod
    def make(cls, version, subsampling_y, coeffs, hue):
        d = cls()
        d.version, d.subsampling_y, d.hue = version, subsampling_y, hue
        for i, c in enumerate(coeffs):
            d.sraw_coeffs[i] = c
        return d


class SrawJob(C.Structure):
    _fields_ = [("desc", SrawDesc), ("in_offset", C.c_uint64), ("img_offset", C.c_uint64),
                ("in_", Image), ("img", Image)]


class HasselbladDesc(C.Structure):
    _fields_ = [("table", HuffTable), ("init_pred", C.c_uint16)]

    @classmethod
    def make(cls, table, init_pred):
        d = cls()
        d.table = HuffTable.make(*table)
        d.init_pred = init_pred
        return d


class HasselbladJob(C.Structure):
    _fields_ = [("desc", HasselbladDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class DngLJpegTile(C.Structure):
    _fields_ = [("desc", LJpegDesc), ("in_", C.c_void_p),
                ("in_bytes", C.c_size_t)]


class DngUnpackTile(C.Structure):
    _fields_ = [("desc", UnpackDesc), ("in_", C.c_void_p),
                ("in_bytes", C.c_size_t)]


class UnpackJob(C.Structure):
    _fields_ = [("desc", UnpackDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class UnpackVariantJob(C.Structure):
    _fields_ = [("desc", UnpackVariantDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class LJpegJob(C.Structure):
    _fields_ = [("desc", LJpegDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class NikonJob(C.Structure):
    _fields_ = [("desc", NikonDesc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


class Cr2Job(C.Structure):
    _fields_ = [("desc", Cr2Desc), ("in_offset", C.c_uint64),
                ("in_bytes", C.c_uint64), ("img_offset", C.c_uint64),
                ("img", Image)]


def fill_recipe(desc, tables, table_index, init_pred):
    """Fill the PerComponentRecipe part of an LJpegDesc / Cr2Desc."""
    assert 1 <= len(tables) <= RSX_MAX_COMPONENTS
    desc.n_tables = len(tables)
    for i, t in enumerate(tables):
        desc.tables[i] = t
    for c, ti in enumerate(table_index):
        desc.table_index[c] = ti
    for c, p in enumerate(init_pred):
        desc.init_pred[c] = p
