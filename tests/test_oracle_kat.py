"""Pins the oracle's bit readers and Huffman decoder on the reference's own
known-answer vectors:
  test/librawspeed/bitstreams/BitSteramerMSBTest.cpp:35-50, BitStreamerLSBTest.cpp:35-50,
  BitStreamerMSB16Test.cpp:35-50, BitStreamerMSB32Test.cpp:35-50,
  BitStreamerJPEGTest.cpp:45-101 (+ access patterns BitStreamerTest.h:41-255),
  test/librawspeed/codes/HuffmanTableTest.cpp:69-132.
The byte patterns below are the vectors of those files; the expected values are
the patterns' definitions ("1, 01, 001, ..." etc.).
"""
import numpy as np
import pytest

from rawspeed_amd import abi

LSB, MSB, MSB16, MSB32, JPEG = range(5)

# 8-byte arrays (the tests declare std::array<uint8_t, 8>, rest zero)
ONES = {
    MSB: [0b10100100, 0b01000010, 0b00001000, 0b00011111],
    JPEG: [0b10100100, 0b01000010, 0b00001000, 0b00011111],
    LSB: [0b01001011, 0b10000100, 0b00100000, 0b11110000],
    MSB16: [0b01000010, 0b10100100, 0b00011111, 0b00001000],
    MSB32: [0b00011111, 0b00001000, 0b01000010, 0b10100100],
}
INV_ONES = {
    MSB: [0b11010010, 0b00100001, 0b00000100, 0b00001111],
    JPEG: [0b11010010, 0b00100001, 0b00000100, 0b00001111],
    LSB: [0b00100101, 0b01000010, 0b00010000, 0b11111000],
    MSB16: [0b00100001, 0b11010010, 0b00001111, 0b00000100],
    MSB32: [0b00001111, 0b00000100, 0b00100001, 0b11010010],
}


def pad8(b):
    return np.array(list(b) + [0] * (8 - len(b)), dtype=np.uint8)


def saturated(order):
    if order == JPEG:  # FF must be stuffed (BitStreamerJPEGTest.cpp:62-64)
        return np.array([0xFF, 0, 0xFF, 0, 0xFF, 0, 0xFF, 0], dtype=np.uint8)
    return pad8([0xFF] * 4)


def gen_ones_be(zeros_to_output, zeros_outputted):
    v, bits = [], 0
    for _ in range(29):
        if zeros_to_output == zeros_outputted:
            bits |= 1
            zeros_to_output += 1
            zeros_outputted = 0
        v.append(bits)
        zeros_outputted += 1
        bits = (bits << 1) & 0xFFFFFFFF
    return v


def gen_ones_le(zeros_to_output, zeros_outputted):
    v, bits, curr = [], 0, -1
    for _ in range(29):
        if zeros_to_output == zeros_outputted:
            bits |= 1 << curr
            zeros_to_output += 1
            zeros_outputted = 0
        v.append(bits)
        zeros_outputted += 1
        curr += 1
    return v


LENS = list(range(1, 8))


@pytest.mark.parametrize("order", [LSB, MSB, MSB16, MSB32, JPEG])
def test_get_bits_patterns(oracle, order):
    # GetTest: for len in 1..7: getBits(len) == element(len)
    st, v = oracle.bitreader_get(order, pad8([0] * 4), LENS)
    assert st == 0 and list(v) == [0] * 7
    st, v = oracle.bitreader_get(order, pad8(ONES[order]), LENS)
    assert st == 0 and list(v) == [1] * 7
    st, v = oracle.bitreader_get(order, pad8(INV_ONES[order]), LENS)
    assert st == 0 and list(v) == [1 << (n - 1) for n in LENS]
    st, v = oracle.bitreader_get(order, saturated(order), LENS)
    assert st == 0 and list(v) == [(1 << n) - 1 for n in LENS]


@pytest.mark.parametrize("order", [LSB, MSB, MSB16, MSB32, JPEG])
def test_increasing_peek_length(oracle, order):
    # IncreasingPeekLengthTest: peekBits(len) == data(len), len = 1..28
    if order == LSB:
        ones, inv = gen_ones_le(0, -1), gen_ones_le(1, 0)
    else:
        ones, inv = gen_ones_be(1, 0), gen_ones_be(0, -1)
    st, v = oracle.peek_increasing(order, pad8(ONES[order]), 28)
    assert st == 0 and list(v) == [ones[n] for n in range(1, 29)]
    st, v = oracle.peek_increasing(order, pad8(INV_ONES[order]), 28)
    assert st == 0 and list(v) == [inv[n] for n in range(1, 29)]
    st, v = oracle.peek_increasing(order, saturated(order), 28)
    assert st == 0 and list(v) == [(1 << n) - 1 for n in range(1, 29)]


def test_jpeg_ff00_is_ff(oracle):
    # BitStreamerJPEGTest.cpp:71-85
    data = pad8([0xFF, 0x00, 0b10100100, 0b01000010, 0b00001000, 0b00011111])
    data = np.concatenate([data, np.zeros(2, np.uint8)])
    st, v = oracle.bitreader_get(JPEG, data, [8] + LENS)
    assert st == 0 and list(v) == [0xFF] + [1] * 7


def test_jpeg_ffxx_is_the_end(oracle):
    # BitStreamerJPEGTest.cpp:87-101: >= 96 zero bits after FF xx
    for end in range(1, 0xFF):
        data = np.array([0xFF, end, 0xFF, 0xFF, 0xFF, 0xFF, 0, 0, 0, 0], dtype=np.uint8)
        st, v = oracle.bitreader_get(JPEG, data, [1] * 96)
        assert st == 0 and not v.any(), end


def test_short_input_is_ioe(oracle):
    # BitStreamer.h:58-59
    assert oracle.bitreader_get(MSB, np.zeros(3, np.uint8), [1])[0] == abi.RSX_ERR_IO
    assert oracle.bitreader_get(JPEG, np.zeros(7, np.uint8), [1])[0] == abi.RSX_ERR_IO


def test_huffman_difference_identity(oracle):
    # HuffmanTableTest.cpp:87-103: codes {2 of length 1} -> values {7, 15}
    t = abi.HuffTable.make([2] + [0] * 15, [7, 15])
    data = np.array([0b00000000, 0b11010101, 0b01010101, 0b01111111], dtype=np.uint8)
    st, v = oracle.huff_decode(t, data, 3)
    assert st == 0 and list(v) == [-127, 21845, 127]


def test_huffman_bad_code(oracle):
    # HuffmanTableTest.cpp:120-132: one 1-bit code "0" -> value 1; "1" is invalid
    t = abi.HuffTable.make([1] + [0] * 15, [1])
    data = np.array([0b00100000, 0, 0, 0], dtype=np.uint8)
    st, v = oracle.huff_decode(t, data, 1)
    assert st == 0 and v[0] == -1
    st, v = oracle.huff_decode(t, data, 2)
    assert st == abi.RSX_ERR_BAD_HUFFMAN_CODE


def test_extend_truth_table(oracle):
    # HuffmanCodeTest.cpp:507-580: extend(diff, len); spot rows of the table via
    # a table whose single code "0" has SSSS = len.
    for ssss, bits, want in [(1, 0b0, -1), (1, 0b1, 1), (2, 0b00, -3), (2, 0b01, -2),
                             (2, 0b10, 2), (2, 0b11, 3), (3, 0b000, -7), (3, 0b011, -4),
                             (3, 0b100, 4), (3, 0b111, 7), (15, 0, -32767),
                             (15, 0x7FFF, 32767), (15, 0x4000, 16384), (15, 0x3FFF, -16384)]:
        t = abi.HuffTable.make([1] + [0] * 15, [ssss])
        word = (bits << (31 - ssss)) & 0xFFFFFFFF  # code bit 0, then the diff bits
        data = np.array([(word >> s) & 0xFF for s in (24, 16, 8, 0)], dtype=np.uint8)
        st, v = oracle.huff_decode(t, data, 1)
        assert st == 0 and v[0] == want, (ssss, bits)
