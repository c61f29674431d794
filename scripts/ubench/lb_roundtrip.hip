// lb_roundtrip.hip -- what does one hop of a decoupled look-back cost on MI355X, and does it
// matter whether producer and consumer sit on the same XCD?
//
// The single-pass LJPEG kernel (rawspeed_amd/csrc/rsx_ljpeg_fast.hip) hands symbol counts
// and predictor state from workgroup to workgroup through 8-byte granules written and polled
// with RELAXED, AGENT-scope atomics (lb_store / lb_load).  11 of a workgroup's 31 us are such
// dependent round trips (ticket, look-back 0, look-back 1).  This measures, with the very same
// instructions:
//   1. ping-pong between two workgroups over one granule: the round trip, by (XCC of A, XCC of B)
//   2. a dependent chain of agent-scope atomic loads / plain loads through a small table
//      (what a poll that finds its record costs), and of atomicAdd on one word (the ticket)
// Build: hipcc --offload-arch=gfx950 -O3 -o lb_roundtrip lb_roundtrip.hip ; run: ./lb_roundtrip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ void lb_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v;
}
constexpr uint32_t SPIN_LIMIT = 1u << 20;

// workgroup b plays with workgroup b ^ x over granule words[16 * min(b, b ^ x)]
__global__ void pingpong(u64* words, int x, int K, uint32_t* info, u64* ticks) {
  const uint32_t b = blockIdx.x, q = b ^ uint32_t(x);
  const uint32_t pair = b < q ? b : q;
  u64* w = words + size_t(pair) * 16;
  if (threadIdx.x != 0)
    return;
  info[2 * b] = xcc_id();
  uint32_t fail = 0;
  // rendezvous: both sides are resident before the clock starts
  u64* ready = w + 8;
  atomicAdd(reinterpret_cast<unsigned long long*>(ready), 1ull);
  for (uint32_t s = 0; s < SPIN_LIMIT && lb_load(ready) < 2ull; ++s)
    __builtin_amdgcn_s_sleep(1);
  const u64 t0 = wall_clock64();
  if (b < q) {
    for (int i = 1; i <= K && !fail; ++i) {
      lb_store(w, u64(2 * i - 1));
      uint32_t s = 0;
      while (lb_load(w) != u64(2 * i))
        if (++s > SPIN_LIMIT) { fail = 1; break; }
    }
  } else {
    for (int i = 1; i <= K && !fail; ++i) {
      uint32_t s = 0;
      while (lb_load(w) != u64(2 * i - 1))
        if (++s > SPIN_LIMIT) { fail = 1; break; }
      lb_store(w, u64(2 * i));
    }
  }
  ticks[b] = wall_clock64() - t0;
  info[2 * b + 1] = fail;
}

// dependent chains, one lane per workgroup: mode 0 plain loads, 1 agent-scope atomic loads,
// 2 atomicAdd on the workgroup's own word, 3 atomicAdd on ONE word (the ticket counter)
__global__ void chain(u64* table, int n, int K, int mode, u64* ticks, u64* sink) {
  if (threadIdx.x != 0)
    return;
  const uint32_t b = blockIdx.x;
  u64 idx = (b * 97u) % uint32_t(n), acc = 0;
  const u64 t0 = wall_clock64();
  for (int i = 0; i < K; ++i) {
    if (mode == 0)
      idx = *(volatile u64*)(table + idx);
    else if (mode == 1)
      idx = lb_load(table + idx);
    else if (mode == 2)
      acc += atomicAdd(reinterpret_cast<unsigned long long*>(table + size_t(n) + 16 * b), 1ull + (acc & 1ull));
    else
      acc += atomicAdd(reinterpret_cast<unsigned long long*>(table + size_t(n) + 16 * 4096), 1ull + (acc & 1ull));
  }
  ticks[b] = wall_clock64() - t0;
  sink[b] = idx + acc;
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  int wall_khz = 0;
  CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
  const double ns_per_tick = 1e6 / double(wall_khz);
  printf("wall clock %d kHz (%.1f ns per tick)\n", wall_khz, ns_per_tick);
  const int NB = 256, K = 400;
  u64 *words, *ticks, *table, *sink;
  uint32_t* info;
  CHECK(hipMalloc(&words, size_t(NB) * 16 * 8));
  CHECK(hipMalloc(&ticks, NB * 8));
  CHECK(hipMalloc(&info, NB * 8));
  std::vector<u64> ht(NB);
  std::vector<uint32_t> hi(2 * NB);
  for (int x : {1, 2, 4, 8, 16, 32, 64}) {
    CHECK(hipMemset(words, 0, size_t(NB) * 16 * 8));
    hipLaunchKernelGGL(pingpong, dim3(NB), dim3(64), 0, 0, words, x, K, info, ticks);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ht.data(), ticks, NB * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hi.data(), info, NB * 8, hipMemcpyDeviceToHost));
    double same = 0, cross = 0;
    int ns = 0, nc = 0, nf = 0;
    std::map<int, int> hist;
    for (int b = 0; b < NB; ++b) {
      const int q = b ^ x;
      if (b > q)
        continue;
      if (hi[2 * b + 1] || hi[2 * q + 1]) { ++nf; continue; }
      const double rt = double(ht[b]) * ns_per_tick / K;
      if ((hi[2 * b] & 15u) == (hi[2 * q] & 15u)) { same += rt; ++ns; } else { cross += rt; ++nc; }
    }
    for (int b = 0; b < NB; ++b)
      hist[hi[2 * b] & 15u]++;
    printf("pingpong partner b^%-3d: same-XCC pairs %3d round trip %7.1f ns | cross-XCC pairs %3d round trip %7.1f ns | failed %d | blocks per XCC:",
           x, ns, ns ? same / ns : 0.0, nc, nc ? cross / nc : 0.0, nf);
    for (auto& kv : hist)
      printf(" %d:%d", kv.first, kv.second);
    printf("\n");
  }
  // block -> XCC of the first 32 blocks (is it blockIdx %% 8?)
  printf("XCC of blocks 0..31:");
  for (int b = 0; b < 32; ++b)
    printf(" %u", hi[2 * b] & 15u);
  printf("\n");
  // chains
  const int N = 32768; // 256 KB table: L2-resident
  CHECK(hipMalloc(&table, (size_t(N) + 16 * 4097) * 8));
  CHECK(hipMalloc(&sink, NB * 8));
  std::vector<u64> h(N);
  for (int i = 0; i < N; ++i)
    h[i] = (u64(i) * 7919u + 13u) % N;
  CHECK(hipMemcpy(table, h.data(), N * 8, hipMemcpyHostToDevice));
  CHECK(hipMemset(table + N, 0, 16 * 4097 * 8));
  const char* names[4] = {"plain dependent loads (256 KB table)", "agent-scope atomic loads (256 KB table)",
                          "atomicAdd, a word per workgroup", "atomicAdd, ONE word (tickets)"};
  for (int blocks : {1, 256, 1024})
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<u64> t(blocks);
      u64 *tk, *sk;
      CHECK(hipMalloc(&tk, blocks * 8));
      CHECK(hipMalloc(&sk, blocks * 8));
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, table, N, 256, mode, tk, sk);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(t.data(), tk, blocks * 8, hipMemcpyDeviceToHost));
      double s = 0;
      for (u64 v : t)
        s += double(v);
      printf("chain %-42s %4d workgroups: %7.1f ns per step\n", names[mode], blocks,
             s / blocks * ns_per_tick / 256);
      CHECK(hipFree(tk));
      CHECK(hipFree(sk));
    }
  return 0;
}
