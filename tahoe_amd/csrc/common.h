// Internal helpers shared by the translation units of libtahoe_amd.so (not part of the ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "tahoe_amd.h"

namespace tahoe {

// Stores a formatted message for tahoe_last_error() (thread-local) and returns `code`.
tahoe_status fail(tahoe_status code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// SplitMix64 evaluated at counter i: element i of the stream seeded with `seed`.
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// 24-bit uniform in [0,1), exact in float32.
static inline float u01(uint64_t x) { return (float)(x >> 40) * (1.0f / 16777216.0f); }

// Host loops over independent items (trees, features) at create time: fn(lo, hi) on up to 16 threads
// (TAHOE_BUILD_THREADS overrides; 1 = in the caller's thread).  fn must not throw.
template <class F>
inline void parallel_for(size_t n, size_t grain, F fn)
{
    unsigned want = std::thread::hardware_concurrency();
    if (const char *e = getenv("TAHOE_BUILD_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) want = (unsigned)v;
    }
    size_t nt = want < 1 ? 1 : (want > 16 ? 16 : want);
    if (grain < 1) grain = 1;
    if (nt > (n + grain - 1) / grain) nt = (n + grain - 1) / grain;
    if (nt <= 1) {
        if (n) fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const size_t part = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) {
        const size_t lo = t * part, hi = lo + part < n ? lo + part : n;
        if (lo < hi) th.emplace_back([=] { fn(lo, hi); });
    }
    fn((size_t)0, part < n ? part : n);
    for (std::thread &x : th) x.join();
}

}  // namespace tahoe

#define TAHOE_HIP_TRY(call)                                                                         \
    do {                                                                                            \
        hipError_t err__ = (call);                                                                  \
        if (err__ != hipSuccess)                                                                    \
            return ::tahoe::fail(TAHOE_ERR_HIP, "%s failed: %s (%s:%d)", #call,                     \
                                 hipGetErrorString(err__), __FILE__, __LINE__);                     \
    } while (0)
