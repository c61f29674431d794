
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 1024;
constexpr int UNROLL = 32;
typedef void (*kern_t)(uint32_t*, uint32_t, long long*);

__global__ void k_add_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_sub_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_sub_u32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_subrev_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_co(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(b) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(c) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(d) : "v"(s), "v"(t) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_addc_co(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(b) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(d) : "v"(s), "v"(t) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_and_b32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_and_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_and_b32 %0, 31, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 31, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 31, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 31, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_and_lit(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_and_b32 %0, 0x7fe, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 0x7fe, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 0x7fe, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_and_b32 %0, 0x7fe, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_or_b32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_or_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_or_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_or_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_xor_b32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_xor_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_xor_b32 %0, 31, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, 31, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, 31, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_xor_b32 %0, 31, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_not_b32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_not_b32 %0, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_not_b32 %0, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_not_b32 %0, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_not_b32 %0, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mov_b32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mov_b32 %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mov_b32 %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mov_b32 %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_var(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshr_var(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshr_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_ashr_var(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_ashr_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_min_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_min_u32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_min_u32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_min_u32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_max_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_max_u32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_max_u32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_max_u32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_cndmask(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d) : "v"(s), "v"(t) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_cmp_lt(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(b) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(c) : "v"(s), "v"(t) : "vcc");
      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(d) : "v"(s), "v"(t) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_cmp_lt_sgpr(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_cmp_lt_u32 s[10:11], %0, %1" : "+v"(a) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_cmp_lt_u32 s[10:11], %0, %1" : "+v"(b) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_cmp_lt_u32 s[10:11], %0, %1" : "+v"(c) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_cmp_lt_u32 s[10:11], %0, %1" : "+v"(d) : "v"(s), "v"(t) : "s10", "s11");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_cmpx_lt(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_cmpx_lt_u32 exec, %0, %1\n s_mov_b64 exec, -1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_cmpx_lt_u32 exec, %0, %1\n s_mov_b64 exec, -1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_cmpx_lt_u32 exec, %0, %1\n s_mov_b64 exec, -1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_cmpx_lt_u32 exec, %0, %1\n s_mov_b64 exec, -1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_bfm(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_bfe_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_bfe_const(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_bfe_u32 %0, %0, 5, 5" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, 5, 5" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, 5, 5" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_bfe_u32 %0, %0, 5, 5" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_bfi(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_alignbit(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_alignbyte(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_and_or(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_or3(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add3(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_add(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_lshl(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_or(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_xad(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mad_i32_i24(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mad_u32_u24(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mul_u32_u24(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mul_lo_u32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_pk_add_u16(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_pk_sub_u16(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_u16(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_u16 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_u16 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_u16 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_u16 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_pack_b32_f16(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_perm(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_sdwa_b0(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_sdwa(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshr_sdwa(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mov_sdwa(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mov_dpp_shr1(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_dpp_shr1(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_ffbh(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_bcnt(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_mbcnt(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_fma_f32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_add_f32(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_movreld(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  asm volatile("s_mov_b32 s12, 0" ::: "s12");
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(DST)\n v_mov_b32 %0, %1\n s_set_gpr_idx_off" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(DST)\n v_mov_b32 %0, %1\n s_set_gpr_idx_off" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(DST)\n v_mov_b32 %0, %1\n s_set_gpr_idx_off" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(DST)\n v_mov_b32 %0, %1\n s_set_gpr_idx_off" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_movrels(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  asm volatile("s_mov_b32 s12, 0" ::: "s12");
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off" : "+v"(a) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off" : "+v"(b) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off" : "+v"(c) : "v"(s), "v"(t));
      asm volatile("s_set_gpr_idx_on s12, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off" : "+v"(d) : "v"(s), "v"(t));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_readlane(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_readlane_b32 s10, %0, 3" : "+v"(a) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readlane_b32 s10, %0, 3" : "+v"(b) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readlane_b32 s10, %0, 3" : "+v"(c) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readlane_b32 s10, %0, 3" : "+v"(d) : "v"(s), "v"(t) : "s10", "s11");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_readfirst(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_readfirstlane_b32 s10, %0" : "+v"(a) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readfirstlane_b32 s10, %0" : "+v"(b) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readfirstlane_b32 s10, %0" : "+v"(c) : "v"(s), "v"(t) : "s10", "s11");
      asm volatile("v_readfirstlane_b32 s10, %0" : "+v"(d) : "v"(s), "v"(t) : "s10", "s11");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_swap(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, s = seed | 1u, t = seed ^ 0x55u; (void)t;
  
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(s));
      asm volatile("v_swap_b32 %0, %1" : "+v"(b), "+v"(s));
      asm volatile("v_swap_b32 %0, %1" : "+v"(c), "+v"(s));
      asm volatile("v_swap_b32 %0, %1" : "+v"(d), "+v"(s));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s;
}

__global__ void k_lshl_b64(uint32_t* out, uint32_t seed, long long* cyc) {
  uint64_t a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7; uint32_t s = seed & 3u;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(c) : "v"(s));
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d) : "v"(s));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = uint32_t(a ^ b ^ c ^ d);
}

__global__ void k_lshr_b64(uint32_t* out, uint32_t seed, long long* cyc) {
  uint64_t a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7; uint32_t s = seed & 3u;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a) : "v"(s));
      asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b) : "v"(s));
      asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(c) : "v"(s));
      asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(d) : "v"(s));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = uint32_t(a ^ b ^ c ^ d);
}

static int run(kern_t k, int blocks, double* ms_out) {
  static uint32_t* out = nullptr; static long long* cyc = nullptr;
  if (!out) { CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4)); CHECK(hipMalloc(&cyc, 8)); }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
  CHECK(hipDeviceSynchronize());
  double best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  *ms_out = best; return 0;
}
int main() {
  struct Op { const char* name; kern_t k; int n; };
  Op ops[] = {
    {"add_u32", k_add_u32, 1},
    {"sub_u32", k_sub_u32, 1},
    {"subrev_u32", k_subrev_u32, 1},
    {"add_co", k_add_co, 1},
    {"addc_co", k_addc_co, 1},
    {"and_b32", k_and_b32, 1},
    {"and_const", k_and_const, 1},
    {"and_lit", k_and_lit, 1},
    {"or_b32", k_or_b32, 1},
    {"xor_b32", k_xor_b32, 1},
    {"xor_const", k_xor_const, 1},
    {"not_b32", k_not_b32, 1},
    {"mov_b32", k_mov_b32, 1},
    {"lshl_var", k_lshl_var, 1},
    {"lshl_const", k_lshl_const, 1},
    {"lshr_var", k_lshr_var, 1},
    {"lshr_const", k_lshr_const, 1},
    {"ashr_var", k_ashr_var, 1},
    {"ashr_const", k_ashr_const, 1},
    {"min_u32", k_min_u32, 1},
    {"max_u32", k_max_u32, 1},
    {"cndmask", k_cndmask, 1},
    {"cmp_lt", k_cmp_lt, 1},
    {"cmp_lt_sgpr", k_cmp_lt_sgpr, 1},
    {"cmpx_lt", k_cmpx_lt, 1},
    {"bfm", k_bfm, 1},
    {"bfe_u32", k_bfe_u32, 1},
    {"bfe_const", k_bfe_const, 1},
    {"bfi", k_bfi, 1},
    {"alignbit", k_alignbit, 1},
    {"alignbyte", k_alignbyte, 1},
    {"and_or", k_and_or, 1},
    {"or3", k_or3, 1},
    {"add3", k_add3, 1},
    {"lshl_add", k_lshl_add, 1},
    {"add_lshl", k_add_lshl, 1},
    {"lshl_or", k_lshl_or, 1},
    {"xad", k_xad, 1},
    {"mad_i32_i24", k_mad_i32_i24, 1},
    {"mad_u32_u24", k_mad_u32_u24, 1},
    {"mul_u32_u24", k_mul_u32_u24, 1},
    {"mul_lo_u32", k_mul_lo_u32, 1},
    {"pk_add_u16", k_pk_add_u16, 1},
    {"pk_sub_u16", k_pk_sub_u16, 1},
    {"add_u16", k_add_u16, 1},
    {"pack_b32_f16", k_pack_b32_f16, 1},
    {"perm", k_perm, 1},
    {"add_sdwa_b0", k_add_sdwa_b0, 1},
    {"lshl_sdwa", k_lshl_sdwa, 1},
    {"lshr_sdwa", k_lshr_sdwa, 1},
    {"mov_sdwa", k_mov_sdwa, 1},
    {"mov_dpp_shr1", k_mov_dpp_shr1, 1},
    {"add_dpp_shr1", k_add_dpp_shr1, 1},
    {"ffbh", k_ffbh, 1},
    {"bcnt", k_bcnt, 1},
    {"mbcnt", k_mbcnt, 1},
    {"fma_f32", k_fma_f32, 1},
    {"add_f32", k_add_f32, 1},
    {"movreld", k_movreld, 1},
    {"movrels", k_movrels, 1},
    {"readlane", k_readlane, 1},
    {"readfirst", k_readfirst, 1},
    {"swap", k_swap, 1},
    {"lshl_b64", k_lshl_b64, 1},
    {"lshr_b64", k_lshr_b64, 1},
  };
  printf("%-14s %8s %8s %8s   (cycles per wave-instruction per SIMD, 2.4 GHz assumed, ILP 4)\n", "op", "2w/SIMD", "4w/SIMD", "8w/SIMD");
  for (const Op& o : ops) {
    printf("%-14s", o.name);
    for (int w = 2; w <= 8; w *= 2) {
      double ms; if (run(o.k, 256 * w, &ms)) return 1;
      printf(" %8.2f", ms * 1e-3 * 2.4e9 / (double(ITERS) * UNROLL * o.n * w));
    }
    printf("\n");
  }
  return 0;
}
