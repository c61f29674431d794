// rsx_pin.h -- the reference-side binding of rsx.h's OPTIONAL page-locked host memory
// (rsx_host_alloc / rsx_host_free, ABI 4), and the process-wide context the hunks share.
//
// No rawspeed type in here, so that adt/AlignedAllocator.h itself can include it
// (INTEGRATION.md 6): with the pool switched on, every allocation of 1 MiB and more that
// goes through rawspeed's AlignedAllocator -- i.e. RawImageData::createData()'s
// `data.resize(pitch * dim.y)`, common/RawImage.cpp:68-100, :199 -- comes from recycled
// hipHostMalloc'ed blocks: the image a decompressor writes is page-locked, the device
// DMA-copies straight into it at the link's rate, and the download runs asynchronously
// under the next upload.  Off (the default) nothing changes.  Blocks are recycled because
// page-locking costs milliseconds per hundred megabytes: a pool pays it once per size.
//   RSX_PINNED_POOL=1 in the environment, or rsx_shim::set_pinned_pool(true)
#pragma once

#include "rsx.h"

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace rawspeed::rsx_shim {

// One lazily created context per process; rsx calls are re-entrant per context
// (the DNG tile threads may all enter: AbstractDngDecompressor.cpp:112-131).
// nullptr when there is no usable device: every hunk then falls through to the
// method's original body, so a patched rawspeed still works without a GPU.
inline rsx_ctx* context() {
  static rsx_ctx* ctx = [] {
    rsx_ctx* c = nullptr;
    // (a librsx.so with another ABI than this header's: every hunk falls through)
    if (rsx_abi_version() != RSX_ABI_VERSION || rsx_ctx_create(/*device=*/0, &c) != RSX_OK)
      c = nullptr;
    return c;
  }();
  return ctx;
}

inline std::atomic<int>& pinned_pool_flag() {
  static std::atomic<int> on{[] {
    const char* e = std::getenv("RSX_PINNED_POOL");
    return (e && e[0] == '1') ? 1 : 0;
  }()};
  return on;
}
inline void set_pinned_pool(bool on) { pinned_pool_flag().store(on ? 1 : 0); }

struct PinnedPool {
  // blocks ever handed out: while it is zero pool_free() answers without the mutex or the
  // look-up -- deallocate() is on every decoder thread's path, the pool is off by default
  std::atomic<size_t> ever_used{0};
  std::mutex m;
  std::multimap<size_t, void*> free_blocks; // by capacity
  std::unordered_map<void*, size_t> live;   // handed out: block -> capacity
  size_t cached_bytes = 0;
  static constexpr size_t MIN_BYTES = size_t(1) << 20;   // smaller allocations stay on the heap
  static constexpr size_t MAX_CACHED = size_t(2) << 30;  // free blocks kept, at most
};
inline PinnedPool& pinned_pool() {
  static PinnedPool* p = new PinnedPool; // (never destroyed: blocks may outlive static teardown)
  return *p;
}

// nullptr: not taken (pool off, small request, no device, pages cannot be locked) --
// the caller allocates as it always did
inline void* pool_alloc(size_t bytes) {
  if (bytes < PinnedPool::MIN_BYTES || !pinned_pool_flag().load(std::memory_order_relaxed))
    return nullptr;
  rsx_ctx* c = context();
  if (!c)
    return nullptr;
  PinnedPool& P = pinned_pool();
  const size_t cap = (bytes + PinnedPool::MIN_BYTES - 1) / PinnedPool::MIN_BYTES * PinnedPool::MIN_BYTES;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_blocks.lower_bound(cap);
    if (it != P.free_blocks.end() && it->first <= cap + cap / 4) {
      void* p = it->second;
      P.live.emplace(p, it->first);
      P.ever_used.fetch_add(1, std::memory_order_release);
      P.cached_bytes -= it->first;
      P.free_blocks.erase(it);
      return p;
    }
  }
  void* p = nullptr;
  if (rsx_host_alloc(c, cap, &p) != RSX_OK || !p)
    return nullptr;
  std::lock_guard<std::mutex> g(P.m);
  P.live.emplace(p, cap);
  P.ever_used.fetch_add(1, std::memory_order_release);
  return p;
}

// false: not one of the pool's blocks -- the caller frees it as it always did
inline bool pool_free(void* p) {
  PinnedPool& P = pinned_pool();
  if (P.ever_used.load(std::memory_order_acquire) == 0)
    return false; // (the pool never handed a block out: nothing of it can come back)
  void* evict = nullptr;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.live.find(p);
    if (it == P.live.end())
      return false;
    const size_t cap = it->second;
    P.live.erase(it);
    if (P.cached_bytes + cap <= PinnedPool::MAX_CACHED) {
      P.free_blocks.emplace(cap, p);
      P.cached_bytes += cap;
    } else {
      evict = p;
    }
  }
  if (evict)
    (void)rsx_host_free(context(), evict);
  return true;
}

} // namespace rawspeed::rsx_shim
