// memcpy2d_pageable.hip -- librsx-free reproducer for the round-5 defect "16 bytes in the middle of
// a row of the host image keep the caller's fill after a device-to-host 2-D copy into pageable
// memory" (profiles/r05/fuzz_big_and_ragged_download.txt).  Plain HIP runtime calls only.
//
//   hipcc --offload-arch=gfx950 -O2 -o memcpy2d_pageable memcpy2d_pageable.hip -lpthread
//   ./memcpy2d_pageable <copy> <stress> <alloc> <threads> <seconds> [seed]
//
//   copy    rect     one hipMemcpy2DAsync per tile, device -> the pageable image, rectangle on
//                    2-byte boundaries (what the library did until round 5)
//           rows1d   ONE contiguous hipMemcpyAsync of the rows into a pageable buffer, the host
//                    moves the row segments (round 5's mitigation)
//           pinned   hipMemcpy2DAsync per tile into a hipHostMalloc'ed packed buffer, the host
//                    moves the row segments (round 6)
//           rect16   as rect, every rectangle ON the 16-byte grid (start and width)
//   stress  none | collapse | move | fork
//           collapse a second thread calls madvise(MADV_COLLAPSE) on the image while copies run
//                    (THP collapse = the pages of the destination MOVE under the copy)
//           move     a second thread bounces the image's pages between NUMA nodes (move_pages)
//           fork     a second thread fork()s + waits (copy-on-write protection of the image)
//   alloc   mmap     a fresh anonymous mapping per image, unmapped after it
//           malloc   malloc/free (glibc: mmap for large blocks until its dynamic threshold has grown,
//                    the heap afterwards)
//           numpy    malloc/free + madvise(MADV_HUGEPAGE) on images of 4 MB and more, as numpy's
//                    allocator does for the arrays of the Python tests
//
// Device rows are written by a kernel with aligned 16-byte stores; byte (row, col) of run `salt` is
// pat(row, col, salt) < 0x80; the host image is pre-filled with 0xA5.  After the copies every byte
// of every rectangle must be the pattern and every byte outside still 0xA5.  A miss is printed
// with its rectangle-relative position and the process goes on counting.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <malloc.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <unistd.h>

#ifndef MADV_COLLAPSE
#define MADV_COLLAPSE 25
#endif

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

__host__ __device__ inline uint8_t pat(uint32_t row, uint32_t col, uint32_t salt) {
  return uint8_t((col * 37u + row * 101u + salt * 29u + (col >> 8) * 11u) & 0x7Fu);
}

__global__ void fill_kernel(uint8_t* d, size_t pitch, uint32_t rows, uint32_t salt) {
  const size_t chunks_per_row = pitch / 16;
  const size_t n = chunks_per_row * rows;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
       i += size_t(gridDim.x) * blockDim.x) {
    const uint32_t row = uint32_t(i / chunks_per_row), c0 = uint32_t(i % chunks_per_row) * 16;
    uint32_t w[4];
    for (int k = 0; k < 4; ++k) {
      uint32_t v = 0;
      for (int b = 0; b < 4; ++b)
        v |= uint32_t(pat(row, c0 + 4 * k + b, salt)) << (8 * b);
      w[k] = v;
    }
    *reinterpret_cast<uint4*>(d + size_t(row) * pitch + c0) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

struct Rect {
  size_t row0, rows, byte0, bytes;
};

struct Config {
  std::string copy, stress, alloc;
  int threads = 1;
  double seconds = 20;
  uint64_t seed = 1;
};

static std::atomic<uint64_t> g_images{0}, g_rects{0}, g_bad_images{0}, g_bad_bytes{0}, g_outside{0};
static std::atomic<bool> g_stop{false};
static bool g_trim = false;

struct Target { // what the stress thread works on
  std::atomic<uint8_t*> base{nullptr};
  std::atomic<size_t> len{0};
};

static void stress_thread(const Config& cfg, Target* t, std::atomic<uint64_t>* events) {
  const long page = sysconf(_SC_PAGESIZE);
  int node = 0;
  while (!g_stop.load()) {
    uint8_t* b = t->base.load();
    const size_t len = t->len.load();
    if (!b || !len) {
      std::this_thread::yield();
      continue;
    }
    uint8_t* lo = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(b) + page - 1) & ~uintptr_t(page - 1));
    uint8_t* hi = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(b) + len) & ~uintptr_t(page - 1));
    if (hi <= lo)
      continue;
    if (cfg.stress == "collapse") {
      // 2 MB aligned sub-ranges only collapse; ask for everything, errors are fine
      if (madvise(lo, size_t(hi - lo), MADV_COLLAPSE) == 0)
        ++*events;
      usleep(200);
    } else if (cfg.stress == "move") {
      const size_t n = std::min<size_t>(size_t(hi - lo) / page, 4096);
      std::vector<void*> pages(n);
      std::vector<int> nodes(n, node), status(n, 0);
      // a window that walks through the image
      static size_t at = 0;
      const size_t total = size_t(hi - lo) / page;
      for (size_t i = 0; i < n; ++i)
        pages[i] = lo + ((at + i) % total) * page;
      at = (at + n) % total;
      if (syscall(SYS_move_pages, 0, n, pages.data(), nodes.data(), status.data(), 2 /*MPOL_MF_MOVE*/) == 0)
        ++*events;
      node ^= 1;
      usleep(100);
    } else if (cfg.stress == "fork") {
      pid_t p = fork();
      if (p == 0)
        _exit(0);
      if (p > 0) {
        int st;
        waitpid(p, &st, 0);
        ++*events;
      }
      usleep(500);
    } else {
      usleep(1000);
    }
  }
}

static uint8_t* image_alloc(const Config& cfg, size_t bytes, void** handle, size_t* map_len,
                            unsigned skew) {
  if (cfg.alloc == "mmap") {
    // numpy / glibc: an mmapped chunk starts 16 bytes into its mapping
    *map_len = bytes + 8192;
    void* p = mmap(nullptr, *map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) {
      perror("mmap");
      exit(2);
    }
    *handle = p;
    return static_cast<uint8_t*>(p) + 16 + (skew & 0xFE0u);
  }
  *map_len = 0;
  void* p = malloc(bytes + 64);
  if (!p) {
    perror("malloc");
    exit(2);
  }
  *handle = p;
  if (cfg.alloc == "numpy" && bytes >= (size_t(1) << 22)) {
    // numpy's allocator (numpy/core/src/multiarray/alloc.c): arrays of 4 MB and more are
    // advised for transparent huge pages from their first page boundary on
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const size_t off = 4096u - a % 4096u;
    madvise(reinterpret_cast<void*>(a + off), bytes - off, MADV_HUGEPAGE);
  }
  return static_cast<uint8_t*>(p);
}
static void image_free(const Config& cfg, void* handle, size_t map_len) {
  if (cfg.alloc == "mmap")
    munmap(handle, map_len);
  else {
    free(handle);
    if (g_trim)
      malloc_trim(128 << 10);
  }
}

static void worker(const Config& cfg, int tid, Target* target) {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::mt19937_64 rng(cfg.seed * 1000003u + tid);
  auto uni = [&](int lo, int hi) { return int(lo + rng() % uint64_t(hi - lo)); }; // [lo, hi)
  const size_t dev_cap = size_t(64) << 20, in_cap = size_t(32) << 20;
  uint8_t *d_out = nullptr, *d_in = nullptr, *h_pin = nullptr;
  CK(hipMalloc(&d_out, dev_cap));
  CK(hipMalloc(&d_in, in_cap));
  if (cfg.copy == "pinned")
    CK(hipHostMalloc(&h_pin, dev_cap, hipHostMallocDefault));
  std::vector<uint8_t> h_in(in_cap, 0x11), h_rows;
  uint32_t salt = uint32_t(tid) * 7919u;
  while (!g_stop.load()) {
    // --- geometry of scripts/fuzz_more.py big3: 3-sample pixels, 1-3 tiles side by side, heights
    // that differ by up to 2 rows
    // GEOM=seed10 in the environment: the ONE image of that generator on which the library's round-5
    // copies lost bytes (profiles/r06/host_path_defect: pitch 15408, three rectangles, the losses in row 8
    // of the third), over and over, at every 16-byte offset of the image in its page
    static const bool fixed_geom = getenv("GEOM") && std::string(getenv("GEOM")) == "seed10";
    const int cpp = 3, k = fixed_geom ? 0 : uni(1, 4);
    int H = fixed_geom ? 1034 : uni(900, 2400);
    int x = 0;
    std::vector<Rect> rects;
    if (fixed_geom) {
      rects.push_back({0, 1033, 0, 2388});
      rects.push_back({0, 1032, 2388, 8046});
      rects.push_back({0, 1034, 10434, 4950});
      x = 2568 - 4; // (W = 2568 below: pitch 15408)
    }
    for (int i = 0; i < k; ++i) {
      int tw = uni(300, 4200 / cpp); // pixels
      int th = H - uni(0, 3);
      size_t b0 = size_t(x) * cpp * 2, bw = size_t(tw) * cpp * 2;
      if (cfg.copy == "rect16") {
        b0 = (b0 + 15) & ~size_t(15);
        bw &= ~size_t(15);
      }
      rects.push_back({0, size_t(th), b0, bw});
      x = int((b0 + bw + cpp * 2 - 1) / (cpp * 2));
    }
    const int W = fixed_geom ? 2568 : x + uni(0, 5);
    const size_t pitch = (size_t(W) * cpp * 2 + 15) / 16 * 16;
    const size_t img_bytes = pitch * H;
    if (img_bytes + 64 > dev_cap)
      continue;
    ++salt;
    size_t map_len = 0;
    void* handle = nullptr;
    uint8_t* img = image_alloc(cfg, img_bytes, &handle, &map_len, unsigned(rng() & 0xFF0));
    memset(img, 0xA5, img_bytes);
    if (tid == 0) {
      target->len.store(img_bytes);
      target->base.store(img);
    }
    // --- what a host-pointer call does in front of its download: zero + upload the input from
    // pageable memory, a kernel that writes the rows
    const size_t in_bytes = size_t(uni(1, 24)) << 20;
    CK(hipMemsetAsync(d_in, 0, in_bytes, s));
    CK(hipMemcpyAsync(d_in, h_in.data(), in_bytes, hipMemcpyHostToDevice, s));
    fill_kernel<<<1024, 256, 0, s>>>(d_out, pitch, uint32_t(H), salt);
    // --- the download
    if (cfg.copy == "rect" || cfg.copy == "rect16") {
      for (const Rect& r : rects) {
        const size_t off = r.row0 * pitch + r.byte0;
        CK(hipMemcpy2DAsync(img + off, pitch, d_out + off, pitch, r.bytes, r.rows,
                            hipMemcpyDeviceToHost, s));
      }
      CK(hipStreamSynchronize(s));
    } else if (cfg.copy == "rows1d") {
      if (h_rows.size() < img_bytes)
        h_rows.resize(img_bytes);
      CK(hipMemcpyAsync(h_rows.data(), d_out, img_bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      for (const Rect& r : rects)
        for (size_t y = 0; y < r.rows; ++y)
          memcpy(img + (r.row0 + y) * pitch + r.byte0, h_rows.data() + (r.row0 + y) * pitch + r.byte0,
                 r.bytes);
    } else { // pinned
      size_t at = 0;
      std::vector<size_t> offs;
      for (const Rect& r : rects) {
        offs.push_back(at);
        CK(hipMemcpy2DAsync(h_pin + at, r.bytes, d_out + r.row0 * pitch + r.byte0, pitch, r.bytes,
                            r.rows, hipMemcpyDeviceToHost, s));
        at += (r.bytes * r.rows + 255) & ~size_t(255);
      }
      CK(hipStreamSynchronize(s));
      for (size_t i = 0; i < rects.size(); ++i) {
        const Rect& r = rects[i];
        for (size_t y = 0; y < r.rows; ++y)
          memcpy(img + (r.row0 + y) * pitch + r.byte0, h_pin + offs[i] + y * r.bytes, r.bytes);
      }
    }
    if (tid == 0)
      target->base.store(nullptr);
    // --- verify
    uint64_t bad = 0, outside = 0;
    std::vector<uint8_t> owner(pitch, 0xFF); // which rectangle owns a column
    for (size_t i = 0; i < rects.size(); ++i)
      for (size_t b = 0; b < rects[i].bytes; ++b)
        owner[rects[i].byte0 + b] = uint8_t(i);
    for (int row = 0; row < H; ++row) {
      const uint8_t* p = img + size_t(row) * pitch;
      for (size_t c = 0; c < pitch; ++c) {
        const uint8_t o = owner[c];
        const bool in = o != 0xFF && size_t(row) < rects[o].rows;
        const uint8_t want = in ? pat(uint32_t(row), uint32_t(c), salt) : uint8_t(0xA5);
        if (p[c] != want) {
          if (in) {
            if (bad < 4 || (bad & 1023) == 0)
              printf("MISS t%d image %dx%d pitch %zu host %p: rect %u (byte0 %zu bytes %zu rows %zu) "
                     "row %d rect-byte %zu (unit %zu + %zu) got 0x%02X want 0x%02X  host addr &4095 = %zu\n",
                     tid, W, H, pitch, static_cast<void*>(img), unsigned(o), rects[o].byte0,
                     rects[o].bytes, rects[o].rows, row, c - rects[o].byte0,
                     (c - rects[o].byte0) / 16, (c - rects[o].byte0) % 16, p[c], want,
                     size_t(reinterpret_cast<uintptr_t>(p + c) & 4095));
            ++bad;
          } else {
            if (outside < 4)
              printf("OUTSIDE t%d row %d byte %zu got 0x%02X\n", tid, row, c, p[c]);
            ++outside;
          }
        }
      }
    }
    ++g_images;
    g_rects += rects.size();
    if (bad || outside)
      ++g_bad_images;
    g_bad_bytes += bad;
    g_outside += outside;
    image_free(cfg, handle, map_len);
  }
  CK(hipStreamDestroy(s));
  CK(hipFree(d_out));
  CK(hipFree(d_in));
  if (h_pin)
    CK(hipHostFree(h_pin));
}

int main(int argc, char** argv) {
  Config cfg;
  if (argc < 6) {
    fprintf(stderr, "usage: %s rect|rows1d|pinned|rect16 none|collapse|move|fork mmap|malloc|numpy|heap|heaptrim threads seconds [seed [prefork]]\n", argv[0]);
    return 2;
  }
  cfg.copy = argv[1];
  cfg.stress = argv[2];
  cfg.alloc = argv[3];
  cfg.threads = atoi(argv[4]);
  cfg.seconds = atof(argv[5]);
  if (argc > 6)
    cfg.seed = strtoull(argv[6], nullptr, 10);
  // "heap": numpy's pattern on the brk heap (glibc's mmap threshold at its maximum from the start,
  // as it is in a Python process that has freed a few large arrays)
  // "heaptrim": the same, and after every image the top of the heap goes back to the system down
  // to 128 KB above the last block in use -- what glibc does by itself (M_TRIM_THRESHOLD, M_TOP_PAD)
  // when a Python process frees the array that lay topmost: the next image's pages from there on are
  // NEW pages at OLD addresses
  if (cfg.alloc == "heap" || cfg.alloc == "heaptrim") {
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    g_trim = cfg.alloc == "heaptrim";
    cfg.alloc = "numpy";
  }
  // prefork: what a Python process has behind it when the tests start -- a heap whose pages were
  // shared copy-on-write with a child (subprocess: fork + exec) that is gone: 1 GB touched, a child
  // forked and reaped, the memory freed again for the images to be carved from
  if (argc > 7 && std::string(argv[7]) == "prefork") {
    std::vector<void*> blocks;
    for (int i = 0; i < 64; ++i) {
      void* b = malloc(size_t(16) << 20);
      memset(b, 0x5A, size_t(16) << 20);
      blocks.push_back(b);
    }
    pid_t c = fork();
    if (c == 0)
      _exit(0);
    int st = 0;
    waitpid(c, &st, 0);
    for (size_t i = 0; i < blocks.size(); i += 2) // (every other block: the heap stays in place)
      free(blocks[i]);
    printf("prefork: 1 GB of heap shared with a child that is gone, half of it free again\n");
  }
  Target target;
  std::atomic<uint64_t> events{0};
  std::vector<std::thread> ws;
  std::thread st;
  if (cfg.stress != "none")
    st = std::thread(stress_thread, std::cref(cfg), &target, &events);
  for (int t = 0; t < cfg.threads; ++t)
    ws.emplace_back(worker, std::cref(cfg), t, &target);
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < cfg.seconds)
    usleep(100000);
  g_stop.store(true);
  for (auto& w : ws)
    w.join();
  if (st.joinable())
    st.join();
  printf("RESULT copy=%s stress=%s alloc=%s threads=%d seconds=%.0f: images %llu rects %llu stress-events %llu "
         "BAD images %llu undelivered bytes %llu bytes written outside %llu\n",
         cfg.copy.c_str(), cfg.stress.c_str(), cfg.alloc.c_str(), cfg.threads, cfg.seconds,
         (unsigned long long)g_images.load(), (unsigned long long)g_rects.load(),
         (unsigned long long)events.load(), (unsigned long long)g_bad_images.load(),
         (unsigned long long)g_bad_bytes.load(), (unsigned long long)g_outside.load());
  return g_bad_images.load() ? 1 : 0;
}
