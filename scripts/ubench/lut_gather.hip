// lut_gather.hip -- what the 64-lane gather into the single-pass kernel's look-up table costs
// (round 6; the review's item: K1f's SQ_LDS_BANK_CONFLICT is 1.45 x its active LDS cycles).
//
// A symbol of lf_step is two dependent LDS reads: the window (two dwords of the lane's own
// column: lane-strided, conflict-free) and the LUT entry at an index the DATA chooses -- 64
// lanes, 64 unrelated places in an 8 KB table.  This benchmark times that second read alone,
// as a dependent chain (the next index comes out of the entry just read, like the next window
// position does), for
//   e8   8-byte entries, 1024 of them (the one-table kernel: ds_read_b64)
//   e4   4-byte entries, 1024 (the two-table instantiation)
//   e2   2-byte entries, 1024 (the table-per-phase instantiation: ds_read_u16)
//   e4b  4-byte entries, 64 of them REPLICATED per bank (entry i of lane l's copy at dword
//        64 i + (l & 63): every lane its own bank -- what a conflict-free first-level table would be)
//   e8l  8-byte entries indexed by the lane itself (the conflict-free reference for e8)
// with index sequences that are uniform-random (worst case) or drawn with the skew of a real
// stream's 10-bit windows (most windows start with one of a few short codes, but the bits
// behind the code are noise: 10-bit indices are nearly uniform inside a code's block),
// at 1, 2 and 4 wavefronts a SIMD (4, 8, 16 waves a CU = 1, 2, 4 workgroups of 256).
//
//   hipcc --offload-arch=gfx950 -O3 -o lut_gather lut_gather.hip && ./lut_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
typedef const __attribute__((address_space(3))) uint16_t* lds_u16p;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_u2p;

// MODE 0 e8, 1 e4, 2 e2, 3 e4b (replicated per bank), 4 e8l (lane-indexed)
template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ seeds, uint32_t* out, int steps,
                                              uint32_t skew_mask) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int j = threadIdx.x;
  // the table: entry i holds a pseudo-random next index in its low bits (and junk above)
  for (int i = j; i < 4096; i += 256)
    reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t(i) * 2654435761u) ^ (uint32_t(i) >> 3) * 40503u;
  __syncthreads();
  uint32_t x = seeds[blockIdx.x * 256 + j];
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    // (what lf_step does with the window: shift + mask to the entry's address; skew_mask keeps
    // the index inside a smaller part of the table to mimic a skewed code distribution)
    uint32_t e;
    if (MODE == 0) {
      const u32x2 v = *(lds_u2p)(((x >> 7) & skew_mask & 0x3FFu) << 3);
      e = v.x ^ v.y;
    } else if (MODE == 1) {
      e = *(lds_u32p)(((x >> 7) & skew_mask & 0x3FFu) << 2);
    } else if (MODE == 2) {
      e = *(lds_u16p)(((x >> 7) & skew_mask & 0x3FFu) << 1);
      e |= e << 16;
    } else if (MODE == 3) {
      e = *(lds_u32p)(((((x >> 7) & 63u) << 6) | (uint32_t(j) & 63u)) << 2);
    } else {
      const u32x2 v = *(lds_u2p)((uint32_t(j) & 0x3FFu) << 3);
      e = v.x ^ v.y ^ x;
    }
    acc += e;
    x = x * 1664525u + e + 1013904223u; // the next index depends on the entry (a dependent chain)
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((j & 63) == 0) {
    out[2 * (blockIdx.x * 4 + (j >> 6))] = uint32_t(t1 - t0);
    out[2 * (blockIdx.x * 4 + (j >> 6)) + 1] = acc;
  }
}

static hipEvent_t g_e0, g_e1;
static float g_ms = 0;

template <int MODE>
double run(int wg_per_cu, int steps, uint32_t skew_mask, const uint32_t* d_seeds, uint32_t* d_out) {
  // 256 CUs x wg_per_cu workgroups, each with 160 KB / wg_per_cu of LDS (capped at 64 KB): that
  // many are resident on a CU, all at once
  size_t lds = size_t(160 * 1024) / size_t(wg_per_cu);
  if (lds > 64 * 1024)
    lds = 64 * 1024;
  lds = lds / 1280 * 1280;
  const int blocks = 256 * wg_per_cu;
  CK(hipEventRecord(g_e0, 0));
  hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), lds, 0, d_seeds, d_out, steps, skew_mask);
  CK(hipEventRecord(g_e1, 0));
  CK(hipEventSynchronize(g_e1));
  CK(hipEventElapsedTime(&g_ms, g_e0, g_e1));
  std::vector<uint32_t> h(size_t(blocks) * 8);
  CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < blocks * 4; ++i)
    sum += h[2 * size_t(i)];
  return sum / (blocks * 4) / steps; // s_memtime ticks a step, per wavefront
}

int main() {
  const int steps = 8192;
  uint32_t *d_seeds, *d_out;
  std::vector<uint32_t> seeds(256 * 4 * 256);
  for (size_t i = 0; i < seeds.size(); ++i)
    seeds[i] = uint32_t(rand()) * 2654435761u + uint32_t(i);
  CK(hipMalloc(&d_seeds, seeds.size() * 4));
  CK(hipMalloc(&d_out, 256 * 4 * 8 * 4));
  CK(hipMemcpy(d_seeds, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice));
  CK(hipEventCreate(&g_e0));
  CK(hipEventCreate(&g_e1));
  run<0>(1, steps, 0x3FFu, d_seeds, d_out); // (warm-up)
  printf("# dependent LUT gathers, one per step: ns per step of a wavefront (kernel time / steps; all\n"
         "# wavefronts of a CU run their chains side by side), at 1 / 2 / 4 wavefronts a SIMD\n");
  printf("%-6s %-10s %12s %12s %12s\n", "mode", "indices", "1 wave/SIMD", "2", "4");
  const char* names[5] = {"e8", "e4", "e2", "e4b", "e8l"};
  for (int mode = 0; mode < 5; ++mode)
    for (int sk = 0; sk < 2; ++sk) {
      const uint32_t mask = sk ? 0x0FFu : 0x3FFu; // (a quarter of the table: skewed)
      double ns[3], tk[3];
      for (int w = 0; w < 3; ++w) {
        const int wg = 1 << w;
        switch (mode) {
        case 0: tk[w] = run<0>(wg, steps, mask, d_seeds, d_out); break;
        case 1: tk[w] = run<1>(wg, steps, mask, d_seeds, d_out); break;
        case 2: tk[w] = run<2>(wg, steps, mask, d_seeds, d_out); break;
        case 3: tk[w] = run<3>(wg, steps, mask, d_seeds, d_out); break;
        default: tk[w] = run<4>(wg, steps, mask, d_seeds, d_out); break;
        }
        ns[w] = double(g_ms) * 1e6 / steps;
      }
      printf("%-6s %-10s %12.1f %12.1f %12.1f   (s_memtime ticks a step: %.2f %.2f %.2f)\n", names[mode],
             sk ? "quarter" : "uniform", ns[0], ns[1], ns[2], tk[0], tk[1], tk[2]);
    }
  return 0;
}
