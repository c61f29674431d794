// Is hipMemset (null stream) asynchronous with respect to the host on this runtime?
// (round 6: plan creation zeroes a few device arrays with hipMemset; the plan's kernels run on a
// hipStreamNonBlocking stream, which the null stream does not order)
//   hipcc --offload-arch=gfx950 -O2 -o memset_async memset_async.hip && ./memset_async
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) {}
  if (out) *out = 1;
}
int main() {
  uint8_t *big, *small;
  const size_t BIG = size_t(8) << 30;
  CK(hipMalloc(&big, BIG));
  CK(hipMalloc(&small, 4096));
  CK(hipMemset(big, 1, 1 << 20));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    CK(hipMemset(big, rep, BIG));
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    printf("hipMemset of 8 GB: the call returned after %.3f ms, the device was done after %.3f ms -> %s\n", t1 - t0,
           t2 - t0, (t1 - t0) < 0.5 * (t2 - t0) ? "ASYNCHRONOUS with respect to the host" : "synchronous");
  }
  // a small memset behind a long kernel on the null stream, and a kernel on a non-blocking stream that
  // reads the small buffer right after the call returns
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* flag;
  CK(hipMalloc(&flag, 4));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(small, 0xFF, 4096));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, 0, 100000000ull, (unsigned*)nullptr); // ~40 ms on the null stream
    const double t0 = now();
    CK(hipMemset(small, 0, 4096));
    const double t1 = now();
    uint8_t h = 0xEE;
    CK(hipMemcpyAsync(&h, small, 1, hipMemcpyDeviceToHost, s)); // (another stream: not ordered behind the null stream)
    CK(hipStreamSynchronize(s));
    const double t2 = now();
    CK(hipDeviceSynchronize());
    printf("small hipMemset behind a 40 ms kernel on the null stream: returned after %.3f ms; a non-blocking stream then read "
           "0x%02x (0x00 = the memset had run, 0xff = it had not) after %.3f ms\n", t1 - t0, h, t2 - t0);
  }
  // ... and hipMemcpy host -> device from pageable memory (what the plan's tables and stream records
  // are uploaded with): does the call return before the bytes are on the device?
  std::vector<uint8_t> hostbuf(1 << 16, 0x5A);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(small, 0xFF, 4096));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, 0, 100000000ull, (unsigned*)nullptr);
    const double t0 = now();
    CK(hipMemcpy(small, hostbuf.data(), 4096, hipMemcpyHostToDevice));
    const double t1 = now();
    uint8_t h = 0xEE;
    CK(hipMemcpyAsync(&h, small, 1, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const double t2 = now();
    CK(hipDeviceSynchronize());
    printf("hipMemcpy H2D (pageable, 4 KB) behind a 40 ms kernel on the null stream: returned after %.3f ms; a non-blocking "
           "stream then read 0x%02x (0x5a = delivered, 0xff = not yet) after %.3f ms\n", t1 - t0, h, t2 - t0);
  }
  for (int rep = 0; rep < 2; ++rep) {
    uint8_t* mid;
    CK(hipMalloc(&mid, 1 << 16));
    CK(hipMemset(mid, 0xFF, 1 << 16));
    CK(hipDeviceSynchronize());
    const double t0 = now();
    CK(hipMemcpy(mid, hostbuf.data(), 1 << 16, hipMemcpyHostToDevice));
    const double t1 = now();
    uint8_t h = 0xEE;
    CK(hipMemcpyAsync(&h, mid + 65535, 1, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    printf("hipMemcpy H2D (pageable, 64 KB), idle device: returned after %.3f ms; a non-blocking stream then read 0x%02x\n",
           t1 - t0, h);
    CK(hipFree(mid));
  }
  return 0;
}
