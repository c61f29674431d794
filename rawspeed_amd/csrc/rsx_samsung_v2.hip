// SamsungV2Decompressor on the device (include/rsx.h section 3g).
//
// What the reference does (decompressors/SamsungV2Decompressor.cpp): an image row is a
// bit stream of its own (BitStreamerMSB32) that starts at the 16-byte boundary behind
// the previous row's last byte (:312-338).  A row is W/16 blocks; a block is a few bits
// that choose the reference pixels ("motion": the two pixels to the left, or 16 pixels
// of the two rows above, optionally averaged, :152-230), the lengths of its 16
// differences in four groups -- coded against a short history per colour (:232-277) --,
// and the differences (:279-311).  Pixel = clampBits(reference + difference * (2 * scale
// + 1) + scale).
//
// Two things are serial in that: where a row STARTS (only known when the row before it
// has been parsed), and the pixel values (a block needs the block to its left or the two
// rows above it).  Neither is serial in the way the reference walks it:
//
//   sv2_spec_kernel     one lane per 16-byte boundary of the input: "if a row started
//                       here, at which boundary would the next one start?"  The parse of a
//                       row depends on nothing but its bits -- the history is reset at
//                       every row start (:322-329), the pixel values never enter it --
//                       and not on the row's parity either: the parity only renames the
//                       two colour histories a row uses, and both start equal.  (Rows 0
//                       and 1 start from another history: parsed in the chain kernel.)
//   sv2_double_kernel   pointer doubling over that table, five times: 32 rows per hop
//   sv2_chain_kernel    rows 0 and 1, then every 32nd row start by hops
//   sv2_fill_kernel     the 31 row starts behind each of those
//   sv2_parse_kernel    one lane per row: the parse proper -- motion, scale, where the
//                       differences of every block lie -- and every error the reference
//                       throws, in the reference's order (none of them depends on pixel
//                       values); the first failing row's status is the job's
//   sv2_diffs_kernel    one lane per block: its sixteen differences, in pixel order
//   sv2_recon_kernel    one workgroup per image: the blocks with the same c + 2 r do not
//                       depend on each other (a block reads block c - 1 of its row and
//                       blocks c - 1 .. c + 1 of the two rows above), so the image is
//                       swept in W/16 + 2 H anti-diagonals; the pixels a diagonal reads
//                       are the ones the last few diagonals wrote and live in an LDS ring
//                       (66 KB: 256 rows x the last 8 blocks, padded against bank conflicts), differences and block
//                       headers are loaded four diagonals ahead.
//
// All of it is bit-exact against oracle_samsung_v2_decompress (tests/test_gpu_samsung_v2.py),
// which is pinned against the reference build including damaged streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "rsx_internal.h"
#include "rsx_ljpeg_dev.h"
#include "rsx_samsung_v2.h"

namespace rsx {

namespace {

constexpr uint32_t SV2_NONE = 0xFFFFFFFFu;
constexpr int SV2_HOP = 32; // rows per hop of the doubled table

struct Sv2JobDev {
  uint64_t in_offset;  // the rows' data (behind the 16-byte header), 16-byte aligned
  uint64_t in_bytes;
  uint64_t img_offset;
  uint64_t px_base;    // first difference (int16) of the job
  uint32_t pitch;
  uint32_t width, height, bits, optflags, init_val;
  uint32_t nb;         // blocks per row
  uint32_t n_bounds;   // 16-byte boundaries of the data: in_bytes / 16 + 1
  uint32_t bound_base; // first entry of the job in the boundary tables
  uint32_t row_base;   // first entry in row_start[] / row_status[]
  uint32_t blk_base;   // first entry in hdr[]
  uint32_t valid;      // 0: the host rejected the job
};

struct Sv2Args {
  const uint8_t* in_base;
  uint8_t* out_base;
  const Sv2JobDev* jobs;
  uint32_t* next;      // [boundary]: boundary at which the following row starts
  uint32_t* jump_a;    // doubling buffers
  uint32_t* jump_b;
  uint32_t* row_start; // [row]: boundary index, SV2_NONE = not reached
  uint32_t* row_status; // [row]: rsx_status of the row's parse
  uint32_t* hdr;       // [block]: motion | length of difference group 3 << 4 | scale << 16
  uint32_t* dq;        // [block]: bit offset of its differences | lengths of groups 0..2 << 20
  int16_t* diffs;      // [pixel]: the differences in pixel order, unscaled
  uint32_t* job_status; // [job]: first failing row << 8 | status, SV2_NONE = fine
  uint32_t n_jobs;
};

// ---------------------------------------------------------------------------
// BitStreamerMSB32 over one row (bitstreams/BitStreamerMSB32.h: little-endian 32-bit
// words, most significant bit first; io/BitStreamer.h:100-132: loads past the end of the
// input are zero-padded, a load that starts more than 8 bytes past it throws), read by
// POSITION: the bits at an offset are a funnel shift of two words, and the reference's
// overflow rule becomes a limit on the offsets a getBits may reach.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sv2_word(const uint8_t* base, uint32_t size, uint32_t k) {
  const uint32_t off = 4u * k;
  if (off + 4u <= size)
    return *reinterpret_cast<const uint32_t*>(base + off);
  uint32_t v = 0;
  for (uint32_t b = 0; b < 4u; ++b)
    if (off + b < size)
      v |= uint32_t(base[off + b]) << (8u * b);
  return v;
}

// The 32 bits at bit offset q of a row (MSB32 words; zeros behind the end of the data)
__device__ __forceinline__ uint32_t sv2_peek(const uint8_t* base, uint32_t size, uint32_t q) {
  const uint32_t k = q >> 5;
  uint32_t w0, w1;
  if (4u * k + 8u <= size) {
    const uint2 w = *reinterpret_cast<const uint2*>(base + 4u * k); // (4-byte aligned)
    w0 = w.x;
    w1 = w.y;
  } else {
    w0 = sv2_word(base, size, k);
    w1 = sv2_word(base, size, k + 1u);
  }
  return uint32_t((((uint64_t(w0) << 32) | w1) << (q & 31u)) >> 32);
}

// How many bytes a row takes, by position: everything in front of a block's differences
// fits one 32-bit peek (2 + 12 scale, 1 + 3 motion, 1 coded, 8 flags = 27 bits), the up to
// four explicit lengths a second one, the differences themselves are skipped.  Less than
// half the instructions of a bit-by-bit reader, which the speculative parse is bound by.  Returns 0, or 1 if the reference would throw in such a row.
__device__ __forceinline__ uint32_t sv2_row_bytes(const Sv2JobDev& J, const uint8_t* base,
                                                  uint32_t size, int first_mode, uint32_t* used) {
  if (size < 4u)
    return 1u;
  const uint32_t limit = 32u * ((size + 8u) / 4u + 1u);
  const uint32_t optflags = J.optflags, nb = J.nb, max_len = J.bits + 1u;
  const bool qp = (optflags & 4u) != 0, mv = (optflags & 2u) != 0, skip = (optflags & 1u) != 0;
  uint32_t m00 = uint32_t(first_mode), m01 = m00, m10 = m00, m11 = m00;
  uint32_t q = 0;
  for (uint32_t blk = 0; blk < nb; ++blk) {
    const uint32_t w = sv2_peek(base, size, q);
    uint32_t n = 0;
    if (!qp && (blk & 3u) == 0u) {
      n = (w >> 30) == 3u ? 14u : 2u;
    }
    if (mv) {
      n += 1u;
    } else {
      n += ((w << n) >> 31) ? 1u : 4u;
    }
    bool coded = true;
    if (!skip) {
      coded = ((w << n) >> 31) == 0u;
      n += 1u;
    }
    uint32_t total = 0;
    if (coded) {
      const uint32_t flags = (w << n) >> 24;
      n += 8u;
      q += n;
      // (all four explicit lengths in one peek, whether or not they are there)
      const uint32_t w2 = sv2_peek(base, size, q);
      uint32_t n2 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t f = (flags >> (6 - 2 * i)) & 3u;
        uint32_t& a0 = i < 2 ? m00 : m10;
        uint32_t& a1 = i < 2 ? m01 : m11;
        uint32_t v;
        if (f == 0u) {
          v = a0;
        } else if (f == 1u) {
          v = a0 + 1u;
        } else if (f == 2u) {
          if (a0 == 0u)
            return 1u;
          v = a0 - 1u;
        } else {
          v = (w2 << n2) >> 28;
          n2 += 4u;
        }
        a0 = a1;
        a1 = v;
        if (v > max_len)
          return 1u;
        total += v;
      }
      q += n2 + 4u * total;
    } else {
      q += n;
    }
    if (q > limit)
      return 1u; // some getBits has run past the end of the data
  }
  *used = (q + 7u) >> 3;
  return 0u;
}

// One row, the parse proper: for every block motion, scale and where its differences lie
// (bit offset and the four lengths), with every check of the reference in its order.  The
// bit pump's sticky overflow (a getBits past `limit`) is looked at where the reference's
// control flow would meet the exception: behind the bits of prepareBaselineValues
// (:159-170), behind the difference lengths (:234-276) and behind the differences
// (:286-290) -- the offsets only grow, so "some getBits has run past the limit" is "the
// offset behind the last one has".  Returns an rsx_status; *used = getStreamPosition().
// (A first version read bit by bit like the reference and wrote the 16 differences of a
// block itself: 800 instructions a block on a lane that is alone on its SIMD, 4.4 ms for
// the rows of four frames; the differences are sv2_diffs_kernel's now, a lane per block.)
__device__ __forceinline__ uint32_t sv2_row_parse(const Sv2JobDev& J, const uint8_t* base,
                                                  uint32_t size, int row, int first_mode,
                                                  uint32_t* hdr, uint32_t* dq, uint32_t* used) {
  if (size < 4u)
    return uint32_t(RSX_ERR_IO); // (BitStreamer.h:56-60)
  const uint32_t limit = 32u * ((size + 8u) / 4u + 1u);
  const uint32_t optflags = J.optflags, nb = J.nb, max_len = J.bits + 1u;
  const bool qp = (optflags & 4u) != 0, mv = (optflags & 2u) != 0, skip = (optflags & 1u) != 0;
  const int width = int(J.width);
  // diffBitsMode: a row uses two of the three colour histories (groups 0-1 one, groups
  // 2-3 the other, :246-247) and all of them start equal: two histories, by group pair
  uint32_t m00 = uint32_t(first_mode), m01 = m00, m10 = m00, m11 = m00;
  int motion = 7, scale = 0;
  uint32_t q = 0;
  for (uint32_t blk = 0; blk < nb; ++blk) {
    const int col = int(blk) * 16;
    const uint32_t w = sv2_peek(base, size, q);
    uint32_t n = 0;
    // prepareBaselineValues :152-230 (the bits it reads and its checks)
    if (!qp && (blk & 3u) == 0u) {
      const uint32_t i = w >> 30;
      n = 2u;
      if (i < 3u) {
        scale += i == 1u ? -2 : (i == 2u ? 2 : 0);
      } else {
        scale = int((w << 2) >> 20);
        n = 14u;
      }
    }
    if (mv) {
      motion = ((w << n) >> 31) ? 3 : 7;
      n += 1u;
    } else {
      const uint32_t keep = (w << n) >> 31;
      n += 1u;
      if (!keep) {
        motion = int((w << n) >> 29);
        n += 3u;
      }
    }
    if (q + n > limit)
      return uint32_t(RSX_ERR_INPUT_OVERFLOW);
    if (row < 2 && motion != 7)
      return uint32_t(RSX_ERR_INVALID_ARG); // :172-173
    if (motion != 7) {
      const int slide = motion == 0 ? -4 : (motion <= 2 ? -2 : (motion <= 4 ? 0 : (motion == 5 ? 2 : 4)));
      const bool avg = motion == 2 || motion == 4;
      for (int i = 0; i < 16; ++i) { // :202-227
        int ref_col = col + i + slide;
        if (!((row + i) & 1))
          ref_col += (i & 1) ? -1 : 1;
        if (ref_col < 0 || ref_col >= width || (avg && ref_col + 2 >= width))
          return uint32_t(RSX_ERR_INVALID_ARG);
      }
    }
    // decodeDiffLengths :232-277
    uint32_t len[4] = {0, 0, 0, 0};
    bool coded = true;
    if (!skip) {
      coded = ((w << n) >> 31) == 0u;
      n += 1u;
    }
    if (coded) {
      const uint32_t flags = (w << n) >> 24; // (n <= 19: all of it is in the first peek)
      n += 8u;
      const uint32_t w2 = sv2_peek(base, size, q + n);
      uint32_t n2 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t f = (flags >> (6 - 2 * i)) & 3u;
        uint32_t& a0 = i < 2 ? m00 : m10;
        uint32_t& a1 = i < 2 ? m01 : m11;
        uint32_t v;
        if (f == 0u) {
          v = a0;
        } else if (f == 1u) {
          v = a0 + 1u;
        } else if (f == 2u) {
          if (a0 == 0u)
            return uint32_t(RSX_ERR_INVALID_ARG); // :258-259
          v = a0 - 1u;
        } else {
          v = (w2 << n2) >> 28;
          n2 += 4u;
        }
        a0 = a1;
        a1 = v;
        if (v > max_len)
          return uint32_t(RSX_ERR_INVALID_ARG); // :271-272
        len[i] = v;
      }
      n += n2;
    }
    q += n;
    if (q > limit)
      return uint32_t(RSX_ERR_INPUT_OVERFLOW);
    // decodeDifferences :279-311: sv2_diffs_kernel's; here only where they lie
    dq[blk] = q | (len[0] << 20) | (len[1] << 24) | (len[2] << 28);
    hdr[blk] = uint32_t(motion) | (len[3] << 4) | (uint32_t(uint16_t(int16_t(scale))) << 16);
    q += 4u * (len[0] + len[1] + len[2] + len[3]);
    if (q > limit)
      return uint32_t(RSX_ERR_INPUT_OVERFLOW);
  }
  *used = (q + 7u) >> 3; // getStreamPosition(): whole bytes the pump has taken
  return 0u;
}

// "a row that starts at boundary i ends where?" -> the boundary the next row starts at,
// SV2_NONE if such a row would throw (then the reference stops there as well)
__device__ __forceinline__ uint32_t sv2_next_boundary(const Sv2JobDev& J, const uint8_t* data,
                                                      uint32_t i, int row, int first_mode) {
  const uint64_t a = uint64_t(i) * 16u;
  if (a > J.in_bytes)
    return SV2_NONE;
  uint32_t used = 0;
  (void)row;
  const uint32_t st = sv2_row_bytes(J, data + a, uint32_t(J.in_bytes - a), first_mode, &used);
  if (st != 0u || a + used > J.in_bytes)
    return SV2_NONE;
  return uint32_t((a + used + 15u) >> 4);
}

// One step along a successor table.  Boundary n_bounds -- 16 * n_bounds > in_bytes, reached
// when in_bytes % 16 != 0 and a row ends in the last partial 16 bytes -- is a legal
// successor: "the next row starts past the data".  It stays what it is (the row that starts
// there fails with RSX_ERR_IO in sv2_parse_kernel, data.skipBytes() :314-316, like the rows
// the fill kernel reaches the same way); only NONE and values beyond it mean "not reached".
__device__ __forceinline__ uint32_t sv2_follow(const uint32_t* tab, uint32_t x, uint32_t n_bounds) {
  if (x == SV2_NONE || x > n_bounds)
    return SV2_NONE;
  return x == n_bounds ? n_bounds : tab[x];
}

__global__ __launch_bounds__(256) void sv2_spec_kernel(Sv2Args A) {
  const Sv2JobDev& J = A.jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (!J.valid || i >= J.n_bounds)
    return;
  // (row 2: any row but the first two -- the parity does not matter for the lengths)
  A.next[J.bound_base + i] = sv2_next_boundary(J, A.in_base + J.in_offset, i, 2, 4);
}

__global__ __launch_bounds__(256) void sv2_double_kernel(Sv2Args A, const uint32_t* from,
                                                         uint32_t* to) {
  const Sv2JobDev& J = A.jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (!J.valid || i >= J.n_bounds)
    return;
  const uint32_t x = from[J.bound_base + i];
  to[J.bound_base + i] = sv2_follow(from + J.bound_base, x, J.n_bounds);
}

// rows 0 and 1 (their histories start at 7, :325-326), then every SV2_HOP-th row
__global__ void sv2_chain_kernel(Sv2Args A, const uint32_t* hop) {
  const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
  if (job >= A.n_jobs)
    return;
  const Sv2JobDev& J = A.jobs[job];
  if (!J.valid)
    return;
  uint32_t* rs = A.row_start + J.row_base;
  const uint8_t* data = A.in_base + J.in_offset;
  const uint32_t height = J.height, n_bounds = J.n_bounds;
  const uint32_t* hop_j = hop + J.bound_base;
  uint32_t x = 0;
  for (uint32_t r = 0; r < 2u && r < height; ++r) {
    rs[r] = x;
    if (x != SV2_NONE)
      x = sv2_next_boundary(J, data, x, int(r), 7);
  }
  for (uint32_t r = 2; r < height; r += uint32_t(SV2_HOP)) {
    rs[r] = x;
    x = sv2_follow(hop_j, x, n_bounds);
  }
}

__global__ __launch_bounds__(256) void sv2_fill_kernel(Sv2Args A) {
  const Sv2JobDev& J = A.jobs[blockIdx.y];
  const uint32_t seg = blockIdx.x * 256u + threadIdx.x;
  const uint32_t r0 = 2u + seg * uint32_t(SV2_HOP);
  if (!J.valid || r0 >= J.height)
    return;
  uint32_t* rs = A.row_start + J.row_base;
  const uint32_t height = J.height, n_bounds = J.n_bounds;
  const uint32_t* next_j = A.next + J.bound_base;
  uint32_t x = rs[r0];
  for (uint32_t r = r0 + 1; r < r0 + uint32_t(SV2_HOP) && r < height; ++r) {
    x = sv2_follow(next_j, x, n_bounds);
    rs[r] = x;
  }
}

__global__ __launch_bounds__(64) void sv2_parse_kernel(Sv2Args A) {
  const Sv2JobDev& J = A.jobs[blockIdx.y];
  const uint32_t row = blockIdx.x * 64u + threadIdx.x;
  if (!J.valid || row >= J.height)
    return;
  uint32_t st = 0;
  const uint32_t x = A.row_start[J.row_base + row];
  if (x == SV2_NONE) {
    st = SV2_NONE; // not reached: an earlier row has failed
  } else {
    const uint64_t a = uint64_t(x) * 16u;
    if (a > J.in_bytes) {
      st = uint32_t(RSX_ERR_IO); // data.skipBytes() to the boundary, :314-316
    } else {
      uint32_t used = 0;
      st = sv2_row_parse(J, A.in_base + J.in_offset + a, uint32_t(J.in_bytes - a), int(row),
                         row < 2u ? 7 : 4, A.hdr + J.blk_base + size_t(row) * J.nb,
                         A.dq + J.blk_base + size_t(row) * J.nb, &used);
      if (st == 0u && a + used > J.in_bytes)
        st = uint32_t(RSX_ERR_IO); // data.skipBytes(pump.getStreamPosition()), :337
      // (the table of row starts comes from the same parse: a row that is fine here has a
      // successor there)
      if (st == 0u && row + 1u < J.height && A.row_start[J.row_base + row + 1u] == SV2_NONE)
        st = uint32_t(RSX_ERR_DEVICE);
    }
  }
  A.row_status[J.row_base + row] = st;
  if (st != 0u && st != SV2_NONE)
    atomicMin(&A.job_status[blockIdx.y], (row << 8) | (st & 0xFFu));
}

// The differences, one lane per block: sixteen fields of the block's four lengths from the
// offset the row's parse has left (getDiff :79-85: sign extension of `len` bits), into
// pixel order (the shuffle of :293-304: pixels 2k, 2k + 1 are differences k and 8 + k of
// the stream, in this order on even rows and the other way round on odd ones).
__global__ __launch_bounds__(256) void sv2_diffs_kernel(Sv2Args A) {
  const Sv2JobDev& J = A.jobs[blockIdx.y];
  if (!J.valid)
    return;
  if (__hip_atomic_load(&A.job_status[blockIdx.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
      SV2_NONE)
    return; // (some row has failed: what lies behind the failure was never parsed)
  const uint32_t nb = J.nb;
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= J.height * nb)
    return;
  const uint32_t row = g / nb;
  const uint64_t a = uint64_t(A.row_start[J.row_base + row]) * 16u;
  const uint8_t* base = A.in_base + J.in_offset + a;
  const uint32_t size = uint32_t(J.in_bytes - a);
  const uint32_t d = A.dq[J.blk_base + g], h = A.hdr[J.blk_base + g];
  const uint32_t len[4] = {(d >> 20) & 15u, (d >> 24) & 15u, d >> 28, (h >> 4) & 15u};
  uint32_t q = d & 0xFFFFFu;
  uint32_t dv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t l = len[i >> 2];
    int v = 0;
    if (l) {
      const uint32_t u = sv2_peek(base, size, q);
      v = int(u) >> (32u - l);
      q += l;
    }
    dv[i] = uint32_t(v) & 0xFFFFu;
  }
  uint32_t pk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    pk[k] = (row & 1u) ? (dv[8 + k] | (dv[k] << 16)) : (dv[k] | (dv[8 + k] << 16));
  uint4* o = reinterpret_cast<uint4*>(A.diffs + J.px_base + size_t(g) * 16);
  o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
}

// ---------------------------------------------------------------------------
// Reconstruction: anti-diagonals t = c + 2 r
// ---------------------------------------------------------------------------
constexpr int SV2_RT = 1024;                 // lanes: 256 blocks x 4 lanes of 4 pixels
constexpr int SV2_RING_ROWS = 256, SV2_RING_BLKS = 8;
// A ring row is 8 blocks = 256 bytes, padded to 264: the 16 blocks a wavefront works on lie
// on a diagonal (row + 1, block - 2), 264 - 64 = 200 bytes apart = 18 banks, and 18 k mod 32
// is a different even bank for each of them (unpadded: 16 banks apart, two bank groups for
// sixteen blocks).  Worth 4 % of the kernel's cycles (SQ_LDS_BANK_CONFLICT): the kernel is
// bound by the vector instructions of its 16 wavefronts on one CU (SQ_ACTIVE_INST_VALU x 4
// cycles = 94 % of its wave cycles, profiles/r03/samsung_v2_pmc/).
constexpr int SV2_RING_STRIDE = SV2_RING_BLKS * 16 + 4; // in pixels
constexpr int SV2_RING_PX = SV2_RING_ROWS * SV2_RING_STRIDE;
constexpr size_t SV2_RING_BYTES = size_t(SV2_RING_PX + 16) * 2;
constexpr int SV2_AHEAD = 8;                 // diagonals the loads run ahead (and the unroll: the
                                             // loop edge costs one full wait)

__device__ __forceinline__ uint32_t sv2_ring_addr(int row, int col) {
  return uint32_t((row & (SV2_RING_ROWS - 1)) * SV2_RING_STRIDE +
                  ((col >> 4) & (SV2_RING_BLKS - 1)) * 16 + (col & 15));
}

typedef uint32_t sv2_u32x2 __attribute__((ext_vector_type(2)));
struct Sv2Fetch {
  uint32_t hdr;
  uint2 diff; // four differences
};

// the block lane group `slot` has on diagonal t
__device__ __forceinline__ bool sv2_diag_block(int nb, int H, int t, int slot, int* r, int* c) {
  int r_lo = (t - (nb - 1) + 1) >> 1; // ceil((t - nb + 1) / 2)
  if (r_lo < 0)
    r_lo = 0;
  const int rr = r_lo + slot;
  const int cc = t - 2 * rr;
  *r = rr;
  *c = cc;
  return rr < H && cc >= 0 && cc < nb;
}

// Every lane issues the same global loads and stores on every path of a step -- offsets
// out of range for the lanes that have no block (buffer instructions drop those): with a load or a
// store under an `if` the compiler cannot count what is in flight behind the loads it waits
// for and waits for everything (s_waitcnt vmcnt(0)), which puts the latency of the loads
// issued four diagonals ahead AND of the pixel stores into every step (1.35 us a step,
// 12.2 ms a frame, with the conditional version).
template <bool ALIGNED8>
__global__ __launch_bounds__(SV2_RT) void sv2_recon_kernel(Sv2Args A) {
  extern __shared__ __attribute__((aligned(16))) uint16_t ring[]; // SV2_RING_BYTES
  const Sv2JobDev& J = A.jobs[blockIdx.x];
  if (!J.valid)
    return;
  // (a vector load: the word was written by the kernel before this one)
  if (__hip_atomic_load(&A.job_status[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
      SV2_NONE)
    return; // the reference throws somewhere: nothing of the image is promised
  // a block = 4 lanes of 4 pixels; at most (nb + 1) / 2 <= 203 blocks on a diagonal
  const int tid = threadIdx.x, px0 = (tid & 3) * 4, slot = tid >> 2;
  const int nb = int(J.nb), H = int(J.height);
  const int T = nb + 2 * (H - 1);
  const uint32_t* hdr = A.hdr + J.blk_base;
  const int16_t* diffs = A.diffs + J.px_base;
  uint8_t* out = A.out_base + J.img_offset;
  const int hi = (1 << J.bits) - 1;
  const int init = int(J.init_val);
  const int W = int(J.width);
  const size_t pitch = J.pitch;
  Sv2Fetch f[SV2_AHEAD];
  // Buffer loads and stores (a resource in scalar registers + a 32-bit offset; out of range =
  // zero / dropped): with 64-bit address arithmetic in vector registers the compiler's
  // temporaries landed on registers that loads were still in flight to, and every step
  // waited for the loads of the step before.
  const __amdgpu_buffer_rsrc_t rs_hdr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(hdr), 0, int(uint32_t(H) * uint32_t(nb) * 4u), 0x00027000);
  const __amdgpu_buffer_rsrc_t rs_diff = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int16_t*>(diffs), 0, int(uint32_t(H) * uint32_t(W) * 2u), 0x00027000);
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, int(uint32_t(H) * uint32_t(pitch)), 0x00027000);
  auto fetch = [&](int t, Sv2Fetch& dst) {
    int r, c;
    const bool ok = sv2_diag_block(nb, H, t, slot, &r, &c) && t < T;
    const uint32_t okm = ok ? 0xFFFFFFFFu : 0u;
    const uint32_t oh = (uint32_t(r) * uint32_t(nb) + uint32_t(c)) * 4u;
    const uint32_t od = (uint32_t(r) * uint32_t(W) + uint32_t(c) * 16u + uint32_t(px0)) * 2u;
    dst.hdr = __builtin_amdgcn_raw_buffer_load_b32(rs_hdr, oh | ~okm, 0, 0);
    const sv2_u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs_diff, od | ~okm, 0, 0);
    dst.diff = make_uint2(d.x, d.y);
  };
#pragma unroll
  for (int k = 0; k < SV2_AHEAD; ++k)
    fetch(k, f[k]);
  for (int t0 = 0; t0 < T; t0 += SV2_AHEAD) {
#pragma unroll
    for (int k = 0; k < SV2_AHEAD; ++k) {
      const int t = t0 + k;
      int r, c;
      const bool ok = sv2_diag_block(nb, H, t, slot, &r, &c) && t < T;
      const uint32_t h = f[k].hdr;
      const int motion = int(h & 7u), scale = int(int16_t(h >> 16));
      const int col = c * 16;
      const int d4[4] = {int(int16_t(f[k].diff.x)), int(int16_t(f[k].diff.x >> 16)),
                         int(int16_t(f[k].diff.y)), int(int16_t(f[k].diff.y >> 16))};
      // Without branches, and with the reference's per-pixel case distinctions folded into
      // two offsets: every lane reads two ring pixels per pixel and selects.  (As branches
      // -- motion 7 or not, averaged or not, the pixel's parity -- the lanes of a wavefront
      // take all of them, one after the other: 266 instructions a step, 13 of them exec-mask
      // saves; the kernel is bound by the instructions of its 16 wavefronts on one CU:
      // 157 vector instructions a step and wavefront now.)
      //   motion 7 (:175-188): the two pixels to the left of the block, init in block 0
      //   else (:194-227): where row + pixel is odd, row - 2, same column + slide; where it
      //   is even, row - 1, one column to the side (+1 for even pixels, -1 for odd ones);
      //   averaged with the pixel two further (motion 2, 4).
      // A lane's pixels are px0 + i with px0 a multiple of 4: pixels 0, 2 have the parity of
      // the row, pixels 1, 3 the other one.
      const bool left = motion == 7;
      const int slide =
          motion == 0 ? -4 : (motion <= 2 ? -2 : (motion <= 4 ? 0 : (motion == 5 ? 2 : 4)));
      const uint32_t av = (motion == 2 || motion == 4) ? 2u : 0u;
      const bool rodd = (r & 1) != 0;
      const uint32_t rb0 = uint32_t(r & (SV2_RING_ROWS - 1)) * uint32_t(SV2_RING_STRIDE);
      const uint32_t rb1 = uint32_t((r - 1) & (SV2_RING_ROWS - 1)) * uint32_t(SV2_RING_STRIDE);
      const uint32_t rb2 = uint32_t((r - 2) & (SV2_RING_ROWS - 1)) * uint32_t(SV2_RING_STRIDE);
      // pixels 0, 2 (A) and 1, 3 (B): ring row and column of the first of the two
      const uint32_t rowA = left ? rb0 : (rodd ? rb2 : rb1);
      const uint32_t rowB = left ? rb0 : (rodd ? rb1 : rb2);
      const int colA = col + (left ? -2 : px0 + slide + (rodd ? 0 : 1));
      const int colB = col + (left ? -1 : px0 + 1 + slide + (rodd ? -1 : 0));
      const uint32_t step2 = left ? 0u : 2u; // pixels 2, 3 lie two columns further
      constexpr uint32_t CM = uint32_t(SV2_RING_BLKS * 16 - 1);
      const bool first = left && c == 0;
      const int mul = scale * 2 + 1;
      int v4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t row = (i & 1) ? rowB : rowA;
        const uint32_t cc0 = uint32_t((i & 1) ? colB : colA) + ((i & 2) ? step2 : 0u);
        const int p0 = int(ring[row + (cc0 & CM)]);
        const int p2 = int(ring[row + ((cc0 + av) & CM)]);
        int base = (p0 + p2 + 1) >> 1;
        base = first ? init : base;
        const int v = base + d4[i] * mul + scale;
        v4[i] = v < 0 ? 0 : (v > hi ? hi : v);
      }
      const uint2 pk = make_uint2(uint32_t(v4[0]) | (uint32_t(v4[1]) << 16),
                                  uint32_t(v4[2]) | (uint32_t(v4[3]) << 16));
      // (lanes without a block: the dump words behind the ring, an offset out of range)
      const uint32_t okm = ok ? 0xFFFFFFFFu : 0u;
      const uint32_t wi = (sv2_ring_addr(r, col + px0) & okm) | (uint32_t(SV2_RING_PX) & ~okm);
      *reinterpret_cast<uint2*>(&ring[wi]) = pk;
      const uint32_t oo = (uint32_t(r) * uint32_t(pitch) + uint32_t(col + px0) * 2u) | ~okm;
      if (ALIGNED8) {
        const sv2_u32x2 pv = {pk.x, pk.y};
        __builtin_amdgcn_raw_buffer_store_b64(pv, rs_out, oo, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b16(uint16_t(v4[0]), rs_out, oo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16(uint16_t(v4[1]), rs_out, oo | ~okm, 2, 0);
        __builtin_amdgcn_raw_buffer_store_b16(uint16_t(v4[2]), rs_out, oo | ~okm, 4, 0);
        __builtin_amdgcn_raw_buffer_store_b16(uint16_t(v4[3]), rs_out, oo | ~okm, 6, 0);
      }
      fetch(t + SV2_AHEAD, f[k]);
      __syncthreads();
    }
  }
}

} // namespace

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct Sv2Plan {
  rsx_ctx* ctx = nullptr;
  std::vector<Sv2JobDev> jobs;
  std::vector<int32_t> host_status; // validation result per job
  DeviceBuffer d_jobs, d_next, d_ja, d_jb, d_row_start, d_row_status, d_hdr, d_dq, d_diffs, d_status;
  std::vector<uint32_t> h_status;
  uint32_t max_bounds = 0, max_rows = 0, max_blocks = 0;
  bool aligned8 = true; // every job's image rows start at multiples of 8 bytes
};

int samsung_v2_validate(const rsx_samsung_v2_desc& d, const rsx_image& img) {
  // the constructor, SamsungV2Decompressor.cpp:87-141
  if (img.cpp != 1)
    return RSX_ERR_INVALID_ARG;
  if (d.bit_depth != 12 && d.bit_depth != 14)
    return RSX_ERR_INVALID_ARG;
  if (d.optflags > 7u)
    return RSX_ERR_INVALID_ARG;
  if (d.width <= 0 || d.height <= 0 || d.width % 16 != 0 || d.width > 6496 || d.height > 4336)
    return RSX_ERR_INVALID_ARG;
  if (d.width != img.dim_x || d.height != img.dim_y)
    return RSX_ERR_INVALID_ARG;
  if (img.pitch_bytes < uint32_t(d.width) * 2u)
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

int samsung_v2_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_samsung_v2_job* jobs,
                           Sv2Plan** out) {
  auto p = std::make_unique<Sv2Plan>();
  p->ctx = ctx;
  p->host_status.assign(n_jobs, RSX_OK);
  p->jobs.resize(n_jobs);
  uint64_t bounds = 0, rows = 0, blocks = 0, px = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_samsung_v2_job& j = jobs[i];
    Sv2JobDev& J = p->jobs[i];
    std::memset(&J, 0, sizeof J);
    int st = samsung_v2_validate(j.desc, j.img);
    // (row starts are 16-byte boundaries of the data and the kernels load whole words)
    if (st == RSX_OK && (j.in_offset % 16 != 0 || j.img_offset % 2 != 0 || j.img.pitch_bytes % 2 != 0))
      st = RSX_ERR_INVALID_ARG;
    if (st == RSX_OK && j.in_bytes >= (1ull << 32) - 64)
      st = RSX_ERR_UNSUPPORTED;
    p->host_status[i] = st;
    if (st != RSX_OK)
      continue;
    J.valid = 1;
    if (j.img_offset % 8 != 0 || j.img.pitch_bytes % 8 != 0)
      p->aligned8 = false;
    J.in_offset = j.in_offset;
    J.in_bytes = j.in_bytes;
    J.img_offset = j.img_offset;
    J.pitch = j.img.pitch_bytes;
    J.width = uint32_t(j.desc.width);
    J.height = uint32_t(j.desc.height);
    J.bits = uint32_t(j.desc.bit_depth);
    J.optflags = j.desc.optflags;
    J.init_val = j.desc.init_val & 0x3FFFu;
    J.nb = J.width / 16u;
    J.n_bounds = uint32_t(j.in_bytes / 16u) + 1u;
    J.bound_base = uint32_t(bounds);
    J.row_base = uint32_t(rows);
    J.blk_base = uint32_t(blocks);
    J.px_base = px;
    bounds += J.n_bounds;
    rows += J.height;
    blocks += uint64_t(J.height) * J.nb;
    px += uint64_t(J.height) * J.width;
    p->max_bounds = std::max(p->max_bounds, J.n_bounds);
    p->max_rows = std::max(p->max_rows, J.height);
    p->max_blocks = std::max(p->max_blocks, J.height * J.nb);
    if (bounds >= (1ull << 32) || blocks >= (1ull << 32))
      return RSX_ERR_UNSUPPORTED;
  }
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st;
  if ((st = p->d_jobs.ensure(p->jobs.size() * sizeof(Sv2JobDev))) ||
      (st = p->d_next.ensure(bounds * 4 + 16)) || (st = p->d_ja.ensure(bounds * 4 + 16)) ||
      (st = p->d_jb.ensure(bounds * 4 + 16)) || (st = p->d_row_start.ensure(rows * 4 + 16)) ||
      (st = p->d_row_status.ensure(rows * 4 + 16)) || (st = p->d_hdr.ensure(blocks * 4 + 16)) || (st = p->d_dq.ensure(blocks * 4 + 16)) ||
      (st = p->d_diffs.ensure(px * 2 + 16)) || (st = p->d_status.ensure(size_t(n_jobs) * 4 + 16)))
    return st;
  RSX_HIP_CHECK(ctx, hipMemcpy(p->d_jobs.ptr, p->jobs.data(), p->jobs.size() * sizeof(Sv2JobDev),
                               hipMemcpyHostToDevice));
  p->h_status.assign(n_jobs, SV2_NONE);
  *out = p.release();
  return RSX_OK;
}

void samsung_v2_plan_destroy(Sv2Plan* p) {
  if (!p)
    return;
  for (DeviceBuffer* b : {&p->d_jobs, &p->d_next, &p->d_ja, &p->d_jb, &p->d_row_start,
                          &p->d_row_status, &p->d_hdr, &p->d_dq, &p->d_diffs, &p->d_status})
    b->release();
  delete p;
}

int samsung_v2_plan_run(Sv2Plan* p, const void* in_dev, void* out_dev, hipStream_t s,
                        KernelTimer* timer) {
  rsx_ctx* ctx = p->ctx;
  const uint32_t n = uint32_t(p->jobs.size());
  if (p->max_rows == 0)
    return RSX_OK; // (every job was rejected by the host)
  // The reconstruction's LDS ring is 66 KB: more than a kernel gets without asking, and
  // the attribute is per DEVICE (a function-local static set it for whichever device was
  // current on the process's first call).  Asked for on the context's device, before the
  // timer is begun.
  if (hipSetDevice(ctx->device) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&sv2_recon_kernel<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(SV2_RING_BYTES)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&sv2_recon_kernel<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(SV2_RING_BYTES)) != hipSuccess)
    return RSX_ERR_DEVICE;
  if (timer)
    timer->begin(s);
  Sv2Args A{};
  A.in_base = static_cast<const uint8_t*>(in_dev);
  A.out_base = static_cast<uint8_t*>(out_dev);
  A.jobs = static_cast<const Sv2JobDev*>(p->d_jobs.ptr);
  A.next = static_cast<uint32_t*>(p->d_next.ptr);
  A.jump_a = static_cast<uint32_t*>(p->d_ja.ptr);
  A.jump_b = static_cast<uint32_t*>(p->d_jb.ptr);
  A.row_start = static_cast<uint32_t*>(p->d_row_start.ptr);
  A.row_status = static_cast<uint32_t*>(p->d_row_status.ptr);
  A.hdr = static_cast<uint32_t*>(p->d_hdr.ptr);
  A.dq = static_cast<uint32_t*>(p->d_dq.ptr);
  A.diffs = static_cast<int16_t*>(p->d_diffs.ptr);
  A.job_status = static_cast<uint32_t*>(p->d_status.ptr);
  A.n_jobs = n;
  auto mark = [&](const char* name) {
    if (timer)
      timer->mark(name);
  };
  RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_status.ptr, 0xFF, size_t(n) * 4, s));
  const dim3 gb((p->max_bounds + 255) / 256, n);
  hipLaunchKernelGGL(sv2_spec_kernel, gb, dim3(256), 0, s, A);
  mark("sv2_spec_kernel");
  // 2, 4, 8, 16, 32 rows per hop
  const uint32_t* from = A.next;
  uint32_t* to = A.jump_a;
  for (int k = 0; k < 5; ++k) {
    hipLaunchKernelGGL(sv2_double_kernel, gb, dim3(256), 0, s, A, from, to);
    from = to;
    to = to == A.jump_a ? A.jump_b : A.jump_a;
  }
  mark("sv2_double_kernel");
  hipLaunchKernelGGL(sv2_chain_kernel, dim3((n + 63) / 64), dim3(64), 0, s, A, from);
  mark("sv2_chain_kernel");
  const uint32_t segs = (p->max_rows + SV2_HOP - 1) / SV2_HOP;
  hipLaunchKernelGGL(sv2_fill_kernel, dim3((segs + 255) / 256, n), dim3(256), 0, s, A);
  mark("sv2_fill_kernel");
  hipLaunchKernelGGL(sv2_parse_kernel, dim3((p->max_rows + 63) / 64, n), dim3(64), 0, s, A);
  mark("sv2_parse_kernel");
  hipLaunchKernelGGL(sv2_diffs_kernel, dim3((p->max_blocks + 255) / 256, n), dim3(256), 0, s, A);
  mark("sv2_diffs_kernel");
  if (p->aligned8)
    hipLaunchKernelGGL(sv2_recon_kernel<true>, dim3(n), dim3(SV2_RT), SV2_RING_BYTES, s, A);
  else
    hipLaunchKernelGGL(sv2_recon_kernel<false>, dim3(n), dim3(SV2_RT), SV2_RING_BYTES, s, A);
  mark("sv2_recon_kernel");
  RSX_HIP_CHECK(ctx, hipGetLastError());
  return RSX_OK;
}

int samsung_v2_plan_results(Sv2Plan* p, hipStream_t s, bool ran, int32_t* job_status) {
  rsx_ctx* ctx = p->ctx;
  if (ran && p->max_rows != 0) {
    RSX_HIP_CHECK(ctx, hipMemcpyAsync(p->h_status.data(), p->d_status.ptr, p->h_status.size() * 4,
                                      hipMemcpyDeviceToHost, s));
    RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  int rc = RSX_OK;
  for (size_t i = 0; i < p->jobs.size(); ++i) {
    int st = p->host_status[i];
    if (st == RSX_OK && ran && p->h_status[i] != SV2_NONE)
      st = int(int8_t(p->h_status[i] & 0xFFu));
    if (job_status)
      job_status[i] = st;
    if (st != RSX_OK)
      rc = st;
  }
  return rc;
}

} // namespace rsx
