// Micro-benchmark: issue rate / latency of the instructions the LJPEG decode loop is made of
// (gfx950).  For every op: a chain of DEPENDENT instructions (latency, 1 wave per SIMD) and
// ILP-4 independent chains at 1 / 2 / 4 / 8 waves per SIMD (throughput).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 32; // instructions per chain per iteration

// OP(d, s): one instruction writing d reading d (dependency) and s
#define DEFINE_KERNEL(NAME, ASM)                                                              \
  __global__ void k_dep_##NAME(uint32_t* out, uint32_t seed, long long* cyc) {                \
    uint32_t a = threadIdx.x + seed, s = seed | 1u, t = seed ^ 0x55u;                         \
    (void)t;                                                                                  \
    long long t0 = clock64();                                                                 \
    for (int i = 0; i < ITERS; ++i) {                                                         \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) asm volatile(ASM : "+v"(a) : "v"(s), "v"(t)); \
    }                                                                                         \
    long long t1 = clock64();                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                \
  }                                                                                           \
  __global__ void k_ilp_##NAME(uint32_t* out, uint32_t seed, long long* cyc) {                \
    uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; \
    (void)t;                                                                                  \
    long long t0 = clock64();                                                                 \
    for (int i = 0; i < ITERS; ++i) {                                                         \
      _Pragma("unroll") for (int u = 0; u < UNROLL / 4; ++u) {                                \
        asm volatile(ASM : "+v"(a) : "v"(s), "v"(t));                                         \
        asm volatile(ASM : "+v"(b) : "v"(s), "v"(t));                                         \
        asm volatile(ASM : "+v"(c) : "v"(s), "v"(t));                                         \
        asm volatile(ASM : "+v"(d) : "v"(s), "v"(t));                                         \
      }                                                                                       \
    }                                                                                         \
    long long t1 = clock64();                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                \
  }

DEFINE_KERNEL(add, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEFINE_KERNEL(lshl, "v_lshlrev_b32 %0, %1, %0")
DEFINE_KERNEL(bfe, "v_bfe_u32 %0, %0, %1, %2")
DEFINE_KERNEL(bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(mad24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFINE_KERNEL(pk_add, "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEFINE_KERNEL(lshr_sdwa, "v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1")
DEFINE_KERNEL(ashr, "v_ashrrev_i32 %0, 31, %0")
DEFINE_KERNEL(xor_, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(fma, "v_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEFINE_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
DEFINE_KERNEL(mov, "v_mov_b32 %0, %1")

// 64-bit shift
__global__ void k_dep_lshl64(uint32_t* out, uint32_t seed, long long* cyc) {
  uint64_t a = threadIdx.x + seed;
  uint32_t s = seed & 3u;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(s));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = uint32_t(a) ^ uint32_t(a >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ilp_lshl64(uint32_t* out, uint32_t seed, long long* cyc) {
  uint64_t a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7;
  uint32_t s = seed & 3u;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(c) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d) : "v"(s));
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = uint32_t(a ^ b ^ c ^ d);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS: dependent pointer chase (latency) and independent reads (throughput)
template <int MODE> // 0: ds_read_b32 chase, 1: ds_read_u16 chase, 2: ds_read2st64 chase
__global__ void k_lds_chase(uint32_t* out, uint32_t seed, long long* cyc) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = ((i * 4 + 64 * 4) & (4096 * 4 - 1) & ~3u);
  __syncthreads();
  uint32_t a = (threadIdx.x * 4) & (4096 * 4 - 1);
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE == 0) asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));
      if (MODE == 1) asm volatile("ds_read_u16 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));
      if (MODE == 2) { uint64_t r; asm volatile("ds_read2st64_b32 %0, %1 offset1:4\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a)); a = uint32_t(r) & 0x3FFCu; }
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> // independent reads, 8 in flight: 0 b32, 1 u16, 2 read2st64, 3 write_b32, 4 write_b16
__global__ void k_lds_thru(uint32_t* out, uint32_t seed, long long* cyc) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  uint32_t a = (threadIdx.x * 4) & (4096 * 4 - 1);
  uint32_t acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
      uint32_t r[8]; uint64_t r2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (MODE == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a), "n"(k * 256));
        if (MODE == 1) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a), "n"(k * 256));
        if (MODE == 2) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r2[k]) : "v"(a), "n"(k), "n"(k + 4));
        if (MODE == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(a), "v"(acc), "n"(k * 256));
        if (MODE == 4) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(a), "v"(acc), "n"(k * 256));
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (MODE <= 1) asm volatile("" : "+v"(r[k]));
        if (MODE == 2) asm volatile("" : "+v"(r2[k]));
      }
      if (MODE <= 1) acc ^= r[0];
      if (MODE == 2) acc ^= uint32_t(r2[0]);
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, uint32_t, long long*);

struct Result { double ms; long long cyc; };
static int run(kern_t k, int blocks, int threads, size_t lds, uint32_t* out, long long* cyc, Result* r) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 12345u, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, 12345u, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  r->ms = ms; r->cyc = c;
  return 0;
}

int main() {
  uint32_t* out; long long* cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4)); CHECK(hipMalloc(&cyc, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int CUS = prop.multiProcessorCount;
  struct Op { const char* name; kern_t dep, ilp; int per_asm; };
#define OPENTRY(N, P) {#N, k_dep_##N, k_ilp_##N, P}
  Op ops[] = { OPENTRY(add,1), OPENTRY(and_or,1), OPENTRY(alignbit,1), OPENTRY(lshl,1), OPENTRY(bfe,1), OPENTRY(bfi,1),
               OPENTRY(mad24,1), OPENTRY(pk_add,1), OPENTRY(add_sdwa,1), OPENTRY(lshr_sdwa,1), OPENTRY(ashr,1), OPENTRY(xor_,1),
               OPENTRY(perm,1), OPENTRY(mul_lo,1), OPENTRY(fma,1), OPENTRY(cmp_cnd,2), OPENTRY(add3,1), OPENTRY(lshl_add,1), OPENTRY(mov,1),
               OPENTRY(lshl64,1) };
  const double n_inst = double(ITERS) * UNROLL;
  printf("%-10s %10s | ILP4 cycles/wave-instr per SIMD at 1,2,4,8 waves/SIMD (wall-clock based, 2.4GHz assumed) | s_memtime cyc/instr 1 wave dep, ilp\n", "op", "dep-lat");
  for (const Op& o : ops) {
    Result r;
    if (run(o.dep, CUS, 256, 0, out, cyc, &r)) return 1; // 1 wave per SIMD
    const double dep_lat = double(r.cyc) / (n_inst * o.per_asm);
    const double dep_wall = r.ms * 1e-3 * 2.4e9 / (n_inst * o.per_asm);
    printf("%-10s %6.2f(%5.2f) |", o.name, dep_lat, dep_wall);
    double ilp1 = 0;
    for (int w = 1; w <= 8; w *= 2) {
      if (run(o.ilp, CUS * w, 256, 0, out, cyc, &r)) return 1;
      // per SIMD: w waves each n_inst instrs
      const double cpi = r.ms * 1e-3 * 2.4e9 / (n_inst * o.per_asm * w);
      if (w == 1) ilp1 = double(r.cyc) / (n_inst * o.per_asm);
      printf(" %6.2f", cpi);
    }
    printf(" | %6.2f\n", ilp1);
  }
  // LDS
  const char* cn[3] = {"ds_read_b32", "ds_read_u16", "ds_read2st64_b32"};
  kern_t chase[3] = {k_lds_chase<0>, k_lds_chase<1>, k_lds_chase<2>};
  for (int m = 0; m < 3; ++m) {
    printf("%-18s chase latency (cycles) at 1,2,4,8 waves/SIMD:", cn[m]);
    for (int w = 1; w <= 8; w *= 2) {
      Result r;
      if (run(chase[m], CUS * w, 256, 16384, out, cyc, &r)) return 1;
      printf(" %7.1f (wall/SIMD-instr %6.2f)", double(r.cyc) / n_inst, r.ms * 1e-3 * 2.4e9 / (n_inst * w));
    }
    printf("\n");
  }
  const char* tn[5] = {"ds_read_b32", "ds_read_u16", "ds_read2st64_b32", "ds_write_b32", "ds_write_b16"};
  kern_t thru[5] = {k_lds_thru<0>, k_lds_thru<1>, k_lds_thru<2>, k_lds_thru<3>, k_lds_thru<4>};
  for (int m = 0; m < 5; ++m) {
    printf("%-18s throughput: cycles per wave-instr per CU at 1,2,4,8 waves/SIMD:", tn[m]);
    for (int w = 1; w <= 8; w *= 2) {
      Result r;
      if (run(thru[m], CUS * w, 256, 16384, out, cyc, &r)) return 1;
      printf(" %6.2f", r.ms * 1e-3 * 2.4e9 / (n_inst * w * 4));
    }
    printf("\n");
  }
  return 0;
}
