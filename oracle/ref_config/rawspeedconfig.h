// TEST INFRASTRUCTURE.  Hand-written build configuration used only when the
// unmodified reference sources under /root/reference are compiled into
// oracle/_ref/ (see oracle/Makefile).  It supplies the values CMake would
// otherwise generate from the reference's src/config.h.in: OpenMP on,
// C++ thread_local on, no zlib / libjpeg / pugixml (those branches of
// AbstractDngDecompressor drop out), generic x86-64 cache/page constants.
#pragma once

#if defined(__SSE2__)
#define WITH_SSE2
#endif

static constexpr unsigned long long RAWSPEED_CACHELINESIZE = 64;
static constexpr unsigned long long RAWSPEED_PAGESIZE = 4096;
static constexpr unsigned long long RAWSPEED_LARGEPAGESIZE = 4096;

#define HAVE_OPENMP
#define HAVE_CXX_THREAD_LOCAL

#ifndef __has_feature
#define __has_feature(x) 0
#endif
#ifndef __has_extension
#define __has_extension __has_feature
#endif

#define RAWSPEED_UNLIKELY_FUNCTION __attribute__((cold))
#define RAWSPEED_NOINLINE __attribute__((noinline))
#define RAWSPEED_READONLY __attribute__((pure))
#define RAWSPEED_READNONE __attribute__((const))
