// Micro-benchmarks behind DESIGN.md 4.2 (the symbol step of the LJPEG kernels):
//   1. dependent LDS reads (pointer chase)            -> LDS round-trip latency
//   2. the window-form symbol step (two dependent LDS round trips + ALU), one
//      wavefront per CU, on random data with an 11-bit LUT of realistic entries
//   3. the same with 2..8 wavefronts per SIMD          -> where latency hiding saturates
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_latency.hip -o /tmp/lds_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int T = 256, BW = 17, LUT_BITS = 11;

__global__ void chase(uint32_t* out, int steps) {
  __shared__ uint32_t tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x)
    tab[i] = (i * 1237u + 17u) & 4095u;
  __syncthreads();
  uint32_t p = threadIdx.x;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i)
    p = tab[p];
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = uint32_t(t1 - t0);
    out[2 * blockIdx.x + 1] = p;
  }
}

__global__ void steps_kernel(const uint32_t* __restrict__ data, const uint16_t* __restrict__ lut_g,
                             uint32_t* out, int rounds) {
  extern __shared__ uint32_t smem[];
  uint32_t* B = smem;                                   // [BW][T]
  uint16_t* lut = reinterpret_cast<uint16_t*>(B + BW * T); // [2048]
  const int j = threadIdx.x;
  for (int k = 0; k < BW; ++k)
    B[k * T + j] = data[(blockIdx.x * BW + k) * T + j];
  for (int i = j; i < (1 << LUT_BITS); i += blockDim.x)
    lut[i] = lut_g[i];
  __syncthreads();
  uint32_t total = 0, n = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    uint32_t pos = (j + r) & 31u;
    while (__any(pos < 512u)) {
      const bool live = pos < 512u;
      const uint32_t wi = pos >> 5;
      const uint32_t d0 = B[wi * T + j], d1 = B[(wi + 1) * T + j];
      const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << (pos & 31u)) >> 32);
      const uint32_t e = lut[w >> (32 - LUT_BITS)];
      pos += live ? (e >> 10) : 0u;
      n += live ? 1u : 0u;
    }
    total += pos;
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((j & 63) == 0) {
    out[4 * (blockIdx.x * (T / 64) + (j >> 6))] = uint32_t(t1 - t0);
    out[4 * (blockIdx.x * (T / 64) + (j >> 6)) + 1] = n;
    out[4 * (blockIdx.x * (T / 64) + (j >> 6)) + 2] = total;
  }
}

int main() {
  uint32_t* d_out;
  hipMalloc(&d_out, 1 << 20);
  std::vector<uint32_t> h(1 << 18);
  // 1. LDS latency
  hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d_out, 4096);
  hipMemcpy(h.data(), d_out, 8, hipMemcpyDeviceToHost);
  printf("LDS dependent read: %.1f cycles (shader clock counter)\n", h[0] / 4096.0);
  // LUT: entries with code 3 + diff 5..6 bits (8-9 bit symbols), like the cfg-3 data
  std::vector<uint16_t> lut(1 << LUT_BITS);
  for (size_t i = 0; i < lut.size(); ++i) {
    const uint32_t ssss = 5 + (i >> 9) % 2, cl = 3;
    lut[i] = uint16_t(cl | (ssss << 5) | ((cl + ssss) << 10));
  }
  uint16_t* d_lut;
  hipMalloc(&d_lut, lut.size() * 2);
  hipMemcpy(d_lut, lut.data(), lut.size() * 2, hipMemcpyHostToDevice);
  const int blocks = 256 * 8;
  std::vector<uint32_t> data(size_t(blocks) * BW * T);
  uint32_t x = 12345;
  for (auto& v : data) {
    x = x * 1664525u + 1013904223u;
    v = x;
  }
  uint32_t* d_data;
  hipMalloc(&d_data, data.size() * 4);
  hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_base = BW * T * 4 + 4096;
  // occupancy sweep: pad LDS so that k workgroups (of 4 wavefronts) fit per CU
  for (int wg_per_cu : {1, 2, 3, 4, 5, 6, 7}) {
    const size_t lds = 163840 / wg_per_cu / 1280 * 1280;
    if (lds < lds_base || lds > 65536)
      continue;
    hipFuncSetAttribute(reinterpret_cast<const void*>(steps_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(steps_kernel, dim3(blocks), dim3(T), lds, 0, d_data, d_lut, d_out, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(steps_kernel, dim3(blocks), dim3(T), lds, 0, d_data, d_lut, d_out, 4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, 16 * 4, hipMemcpyDeviceToHost);
    const double steps_per_wave = h[1] / 64.0;
    printf("%d workgroups/CU (LDS %zu): kernel %.1f us; wave 0: %u cycles for %.0f steps = %.0f "
           "cycles/step\n",
           wg_per_cu, lds, ms * 1e3, h[0], steps_per_wave, h[0] / steps_per_wave);
  }
  return 0;
}
