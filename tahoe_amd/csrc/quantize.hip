// QRING, part 1: float32 rows -> u16 rank codes (see qring.hip for the idea and the walk).  For every feature f,
// code_f(x) = #{ e in tab_f : e <= x } over the sorted distinct thresholds the forest uses on f; 0xFFFF marks the
// missing sentinel.  Three kernel forms, chosen per tree group at create:
//   quantize_multi_kernel<Q>       16 / 8 features per workgroup, float4 row reads, Eytzinger search trees (small tables)
//   quantize_bucket_pair_kernel    2 features per workgroup, direct-index bucket table + fixed-step window search
//   quantize_pair_kernel / quantize_kernel<1>   Eytzinger search trees, 2 / 1 features per workgroup (fallbacks)
// Replaces nothing in the reference (its kernels compare float32 features, Struct.h:359-407); it is the pre-pass
// that lets the walk run on 16-bit integers without changing a single branch.
#include "qring_internal.h"

namespace tahoe {

// ------------------------------------------------------------------------------------------------
// (1) float32 rows -> u16 codes.  One workgroup = F adjacent features x 2^cshift rows (quantize_launch), the F
// search trees in LDS (stride `tab_stride` floats).  A thread reads the F values of a row with one
// F*4-byte load: the row-major input is fetched in 64-byte lines of 16 features, and a workgroup uses
// F*4 bytes of each line it pulls through L2 -> L1, so F = 2 halves and F = 4 quarters that traffic
// (the kernel was bound by it at F = 1: 16 GB of line traffic for a 1 GB batch).
template <int F>
struct QVec;
template <>
struct QVec<1> { using T = float; };
template <>
struct QVec<2> { using T = float2; };
template <>
struct QVec<4> { using T = float4; };
__device__ __forceinline__ float qv_get(float v, int) { return v; }
__device__ __forceinline__ float qv_get(float2 v, int j) { return j == 0 ? v.x : v.y; }
__device__ __forceinline__ float qv_get(float4 v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int F>
__global__ void __launch_bounds__(kQuantThreads) quantize_kernel(const float *__restrict__ data, const float *__restrict__ tables,
                                                                 const int *__restrict__ offsets, uint16_t *__restrict__ xq,
                                                                 uint32_t *__restrict__ chunk_flags, size_t rows, int cols,
                                                                 float missing, int tab_stride, int trs, int cshift, int perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    // Workgroups are dealt round-robin over the 8 XCDs (private L2s): give every XCD a contiguous range of
    // (chunk, feature group) pairs, so the workgroups that re-read a line sit behind the same L2 (placement
    // is a speed matter only; any mapping is correct).
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;
    const int groups = cols / F;
    const int f0 = (int)(vid % groups) * F;
    const size_t chunk = vid / groups;
    // Each feature's thresholds as a perfect binary search tree in level order (1-based, NaN padding),
    // size = 2^p entries.  A probe sequence touches one entry per level and a level is contiguous, so the 64
    // lanes of a probe spread over the LDS banks (a sorted array probed at power-of-two strides puts every
    // lane in the same bank: 32-way conflicts, measured).
    int size[F];
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const int base = offsets[f0 + j];
        size[j] = offsets[f0 + j + 1] - base;  // 2^p, p >= 0 (size 1 = no thresholds)
        for (int i = threadIdx.x; i < size[j]; i += blockDim.x) tab[j * tab_stride + i] = tables[base + i];
    }
    __syncthreads();
    const size_t r0 = chunk << cshift;
    const size_t r1 = min(rows, r0 + ((size_t)1 << cshift));
    constexpr int U = (F == 1) ? 4 : 2;  // rows per thread and iteration: U * F independent search chains
    using V = typename QVec<F>::T;
    bool saw_missing = false;
    for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)blockDim.x * U) {
        V xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)u * blockDim.x, r1 - 1);  // clamped: in bounds, result unused
            xv[u] = *reinterpret_cast<const V *>(data + r * (size_t)cols + f0);
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float *tj = tab + j * tab_stride;
            float x[U];
            int cnt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u] = qv_get(xv[u], j);
                cnt[u] = 1;
            }
            // descend: k <- 2k + (tab[k] <= x); after p levels k - 2^p = #{thresholds <= x} (NaN x -> 0)
            for (int lim = size[j]; lim > 1; lim >>= 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) cnt[u] = 2 * cnt[u] + (tj[cnt[u]] <= x[u] ? 1 : 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = rb + (size_t)u * blockDim.x;
                if (r < r1) {
                    const bool ms = fabsf(x[u] - missing) <= kMissingEps;
                    saw_missing |= ms;
                    const uint32_t code = ms ? kCodeMissing : (uint32_t)(cnt[u] - size[j]);
                    q_store_code(xq, r, f0 + j, cols, trs, perm, code);
                }
            }
        }
    }
    // one atomic per wave at most: tells the walk kernel whether this row chunk needs the missing rule
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// Pair form used when num_cols is even: one workgroup = features (f0, f0+1) x 2^cshift rows, float2
// loads.  When both search trees fit the LDS budget they are resident together (one pass); a pair with an
// oversized tree is done in two passes with one tree resident at a time, so a few large features do not
// force the whole launch down to one feature per workgroup.
__global__ void __launch_bounds__(kQuantPairThreads)
    quantize_pair_kernel(const float *__restrict__ data, const float *__restrict__ tables, const int *__restrict__ offsets,
                         uint16_t *__restrict__ xq, uint32_t *__restrict__ chunk_flags, size_t rows, int cols, float missing,
                         int lds_floats, int trs, int cshift, int perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;  // XCD-contiguous
    const int groups = cols / 2;
    const int f0 = (int)(vid % groups) * 2;
    const size_t chunk = vid / groups;
    const int base0 = offsets[f0], base1 = offsets[f0 + 1];
    const int size0 = base1 - base0, size1 = offsets[f0 + 2] - base1;
    const bool together = size0 + size1 <= lds_floats;
    const size_t r0 = chunk << cshift;
    const size_t r1 = min(rows, r0 + ((size_t)1 << cshift));
    constexpr int U = 4;  // rows per thread and iteration: 8 independent search chains hide the LDS latency
    bool saw_missing = false;
    for (int pass = 0; pass < (together ? 1 : 2); ++pass) {
        if (pass) __syncthreads();  // everyone is done reading the first tree
        const bool do0 = together || pass == 0, do1 = together || pass == 1;
        const float *t0 = tab;
        const float *t1 = together ? tab + size0 : tab;
        if (do0)
            for (int i = threadIdx.x; i < size0; i += blockDim.x) tab[i] = tables[base0 + i];
        if (do1)
            for (int i = threadIdx.x; i < size1; i += blockDim.x) tab[(together ? size0 : 0) + i] = tables[base1 + i];
        __syncthreads();
        float2 nx[U];  // next iteration's rows, loaded one iteration ahead (see quantize_bucket_pair_kernel)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(r0 + threadIdx.x + (size_t)u * blockDim.x, r1 - 1);
            nx[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
        }
        for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)blockDim.x * U) {
            float2 xv[U];
            int c0[U], c1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = nx[u];
                c0[u] = 1;
                c1[u] = 1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = min(rb + (size_t)(U + u) * blockDim.x, r1 - 1);  // clamped: in bounds, result unused
                nx[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
            }
            // descend both trees: k <- 2k + (tab[k] <= x); k - 2^p = #{thresholds <= x} (NaN x -> 0)
            if (do0)
                for (int lim = size0; lim > 1; lim >>= 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) c0[u] = 2 * c0[u] + (t0[c0[u]] <= xv[u].x ? 1 : 0);
                }
            if (do1)
                for (int lim = size1; lim > 1; lim >>= 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) c1[u] = 2 * c1[u] + (t1[c1[u]] <= xv[u].y ? 1 : 0);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = rb + (size_t)u * blockDim.x;
                if (r < r1) {
                    if (do0) {
                        const bool ms = fabsf(xv[u].x - missing) <= kMissingEps;
                        saw_missing |= ms;
                        q_store_code(xq, r, f0, cols, trs, perm, ms ? kCodeMissing : (uint32_t)(c0[u] - size0));
                    }
                    if (do1) {
                        const bool ms = fabsf(xv[u].y - missing) <= kMissingEps;
                        saw_missing |= ms;
                        q_store_code(xq, r, f0 + 1, cols, trs, perm, ms ? kCodeMissing : (uint32_t)(c1[u] - size1));
                    }
                }
            }
        }
    }
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// ------------------------------------------------------------------------------------------------
// Bucketed form of the same conversion.  code(x) = #{e <= x} needs ~log2(n) dependent LDS probes in a search
// tree (14 at K3, the kernel's whole cost: bank conflicts on 64 random probes).  A monotone map
// bucket(x) = clamp(trunc((x - lo) * scale), 0, B - 1) splits the sorted thresholds into B runs; monotone means
// every threshold in a lower bucket is < x and every one in a higher bucket is > x, so
//   code(x) = start[bucket(x)] + #{e in x's bucket : e <= x}
// exactly, whatever the distribution -- a skewed one only makes some runs longer.  The run is searched with a
// fixed number of branch-free steps (steps_f = ceil(log2(longest run + 1)), over a window that may reach into the
// following runs (> x, harmless) and, at the end of the array, into NaN padding.  The thresholds' own buckets are
// computed by bucket_index_kernel with the very same q_bucket() on the device, so host and device arithmetic
// never have to agree.  NaN x: fmaxf(NaN, 0) = 0 -> bucket 0, every compare false -> code 0, as before.
__device__ __forceinline__ int q_bucket(float x, float lo, float scale, float bm1)
{
    float t = (x - lo) * scale;
    t = fminf(fmaxf(t, 0.0f), bm1);  // NaN -> 0; +-inf clamp
    return (int)t;
}

__global__ void bucket_index_kernel(const float *__restrict__ vals, const int *__restrict__ feat, const float4 *__restrict__ params,
                                    float bm1, int n, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = params[feat[i]];
    out[i] = q_bucket(vals[i], p.x, p.y, bm1);
}

__global__ void __launch_bounds__(kQuantPairThreads)
    quantize_bucket_pair_kernel(const float *__restrict__ data, const float *__restrict__ bsorted, const int *__restrict__ boffsets,
                                const uint16_t *__restrict__ bstarts, const float4 *__restrict__ bparams,
                                uint16_t *__restrict__ xq, uint32_t *__restrict__ chunk_flags, size_t rows, int cols, float missing,
                                int B, int trs, int cshift, int perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;  // XCD-contiguous
    const int groups = cols / 2;
    const int f0 = (int)(vid % groups) * 2;
    const size_t chunk = vid / groups;
    const int base0 = boffsets[f0], base1 = boffsets[f0 + 1];
    const int len0 = base1 - base0, len1 = boffsets[f0 + 2] - base1;
    float *s0 = reinterpret_cast<float *>(smem);
    float *s1 = s0 + len0;
    uint16_t *st0 = reinterpret_cast<uint16_t *>(s1 + len1);
    uint16_t *st1 = st0 + (B + 2);
    for (int i = threadIdx.x; i < len0; i += blockDim.x) s0[i] = bsorted[base0 + i];
    for (int i = threadIdx.x; i < len1; i += blockDim.x) s1[i] = bsorted[base1 + i];
    {   // both start tables are adjacent in global memory too: (B + 2) u16 each, copied as dwords
        const uint32_t *src = reinterpret_cast<const uint32_t *>(bstarts + (size_t)f0 * (B + 2));
        uint32_t *dst = reinterpret_cast<uint32_t *>(st0);
        for (int i = threadIdx.x; i < B + 2; i += blockDim.x) dst[i] = src[i];
    }
    const float4 p0 = bparams[f0], p1 = bparams[f0 + 1];
    const int steps0 = __float_as_int(p0.z), steps1 = __float_as_int(p1.z);
    const float bm1 = (float)(B - 1);
    __syncthreads();
    const size_t r0 = chunk << cshift;
    const size_t r1 = min(rows, r0 + ((size_t)1 << cshift));
    constexpr int U = 4;
    bool saw_missing = false;
    // the rows of the next iterations are loaded before the current ones are converted: the loads (a new cache line
    // each, HBM latency) fly under the searching
    float2 nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r = min(r0 + threadIdx.x + (size_t)u * blockDim.x, r1 - 1);
        nx[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
    }
    for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)blockDim.x * U) {
        float2 xv[U];
        int c0[U], c1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = nx[u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)(U + u) * blockDim.x, r1 - 1);  // clamped: in bounds, result unused
            nx[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {  // positions are kept in bytes: one add forms the probe address
            c0[u] = 4 * (int)st0[q_bucket(xv[u].x, p0.x, p0.y, bm1)];
            c1[u] = 4 * (int)st1[q_bucket(xv[u].y, p1.x, p1.y, bm1)];
        }
        // branch-free upper bound inside the window [c, c + 2^steps - 1): c += half when s[c + half - 1] <= x
        const char *b0 = reinterpret_cast<const char *>(s0) - 4, *b1 = reinterpret_cast<const char *>(s1) - 4;
        int k = max(steps0, steps1) - 1;
        for (; k >= min(steps0, steps1); --k) {  // the longer window's extra steps
            const int half4 = 4 << k;
            if (steps0 > steps1) {
#pragma unroll
                for (int u = 0; u < U; ++u) c0[u] += (*reinterpret_cast<const float *>(b0 + c0[u] + half4) <= xv[u].x) ? half4 : 0;
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) c1[u] += (*reinterpret_cast<const float *>(b1 + c1[u] + half4) <= xv[u].y) ? half4 : 0;
            }
        }
        for (; k >= 0; --k) {  // both features: 2 * U independent probes in flight
            const int half4 = 4 << k;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float v0 = *reinterpret_cast<const float *>(b0 + c0[u] + half4);
                const float v1 = *reinterpret_cast<const float *>(b1 + c1[u] + half4);
                c0[u] += (v0 <= xv[u].x) ? half4 : 0;
                c1[u] += (v1 <= xv[u].y) ? half4 : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c0[u] >>= 2;
            c1[u] >>= 2;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = rb + (size_t)u * blockDim.x;
            if (r < r1) {
                const bool ms0 = fabsf(xv[u].x - missing) <= kMissingEps, ms1 = fabsf(xv[u].y - missing) <= kMissingEps;
                saw_missing |= ms0 | ms1;
                q_store_code(xq, r, f0, cols, trs, perm, ms0 ? kCodeMissing : (uint32_t)c0[u]);
                q_store_code(xq, r, f0 + 1, cols, trs, perm, ms1 ? kCodeMissing : (uint32_t)c1[u]);
            }
        }
    }
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// ------------------------------------------------------------------------------------------------
// Many features per workgroup, for forests whose per-feature tables are small (K2: 3072 features x ~40 thresholds).
// The pair kernels read 8 bytes of every 64-byte line they touch and are bound by the texture path (one cache
// line per lane); with F = 4 * Q features per workgroup a row is read by Q adjacent lanes as float4, F = 16 uses
// whole lines.  Tables: the Eytzinger search trees (a few hundred bytes each), descent as in quantize_kernel; the
// trip count is per lane (adjacent lanes serve different features), a wave runs to its longest.
template <int Q>
__global__ void __launch_bounds__(kQuantPairThreads)
    quantize_multi_kernel(const float *__restrict__ data, const float *__restrict__ tables, const int *__restrict__ offsets,
                          uint16_t *__restrict__ xq, uint32_t *__restrict__ chunk_flags, size_t rows, int cols, float missing,
                          int tab_stride, int trs, int cshift, int perm)
{
    constexpr int F = 4 * Q;
    constexpr int RPI = kQuantPairThreads / Q;  // rows per block iteration
    constexpr int U = 2;                        // rows per thread and iteration: 8 independent descents
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;  // XCD-contiguous
    const int groups = cols / F;
    const int f0 = (int)(vid % groups) * F;
    const size_t chunk = vid / groups;
    for (int j = 0; j < F; ++j) {
        const int base = offsets[f0 + j], size = offsets[f0 + j + 1] - base;
        for (int i = threadIdx.x; i < size; i += blockDim.x) tab[j * tab_stride + i] = tables[base + i];
    }
    const int quad = threadIdx.x % Q, rsub = threadIdx.x / Q;
    const int fq = f0 + 4 * quad;  // this thread's four features
    int size[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) size[j] = offsets[fq + j + 1] - offsets[fq + j];
    __syncthreads();
    const size_t r0 = chunk << cshift;
    const size_t r1 = min(rows, r0 + ((size_t)1 << cshift));
    bool saw_missing = false;
    float4 nx[U];  // next iteration's rows, loaded one iteration ahead (see quantize_bucket_pair_kernel)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r = min(r0 + rsub + (size_t)u * RPI, r1 - 1);
        nx[u] = *reinterpret_cast<const float4 *>(data + r * (size_t)cols + fq);
    }
    for (size_t rb = r0 + rsub; rb < r1; rb += (size_t)RPI * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = nx[u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)(U + u) * RPI, r1 - 1);  // clamped: in bounds, result unused
            nx[u] = *reinterpret_cast<const float4 *>(data + r * (size_t)cols + fq);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *tj = tab + (4 * quad + j) * tab_stride;
            float x[U];
            int cnt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u] = qv_get(xv[u], j);
                cnt[u] = 1;
            }
            for (int lim = size[j]; lim > 1; lim >>= 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) cnt[u] = 2 * cnt[u] + (tj[cnt[u]] <= x[u] ? 1 : 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = rb + (size_t)u * RPI;
                if (r < r1) {
                    const bool ms = fabsf(x[u] - missing) <= kMissingEps;
                    saw_missing |= ms;
                    q_store_code(xq, r, fq + j, cols, trs, perm, ms ? kCodeMissing : (uint32_t)(cnt[u] - size[j]));
                }
            }
        }
    }
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// Features per workgroup of the many-features form: 4 x kQuantMultiMax at most (make QMULTI=4 keeps 16; 8 = 32 features: a row's
// 32 values are one whole 128-byte line, read by eight adjacent lanes).
#ifndef TAHOE_QUANT_MULTI_MAX
#define TAHOE_QUANT_MULTI_MAX 4  // 32 features per workgroup measured slower (profiles/r04/tune_qmulti.txt: KR3 0.47 -> 0.61 ms, K2 0.94 -> 1.10)
#endif
constexpr int kQuantMultiMax = TAHOE_QUANT_MULTI_MAX;

// ------------------------------------------------------------------------------------------------
// host side
// Bucketed quantiser tables for one group (see quantize_bucket_pair_kernel).  Leaves g.buckets = 0 when the form is
// unavailable (odd num_cols, a feature pair that does not fit LDS at any bucket count, TAHOE_QUANT_BUCKETS=0).
static tahoe_status build_buckets(tahoe_forest *f, const std::vector<std::vector<float>> &tab, tahoe_qgroup &g)
{
    const int cols = f->p.num_cols;
    g.buckets = 0;
    if (cols % 2 != 0) return TAHOE_OK;
    if (const char *e = getenv("TAHOE_QUANT_BUCKETS"))
        if (atoi(e) == 0) return TAHOE_OK;
    size_t total = 0;
    for (int c = 0; c < cols; ++c) total += tab[c].size();
    std::vector<float> vals;
    std::vector<int> feat;
    vals.reserve(total);
    feat.reserve(total);
    for (int c = 0; c < cols; ++c)
        for (float v : tab[c]) {
            vals.push_back(v);
            feat.push_back(c);
        }
    hipError_t e = hipSuccess;
    float *d_vals = nullptr;
    int *d_feat = nullptr, *d_out = nullptr;
    float4 *d_params = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)d_vals, (void *)d_feat, (void *)d_out, (void *)d_params})
            if (p) (void)hipFree(p);
    };
    auto bad = [&](const char *what) {
        cleanup();
        return fail(TAHOE_ERR_HIP, "qring_build(buckets): %s failed: %s", what, hipGetErrorString(e));
    };
    const size_t n1 = std::max<size_t>(total, 1);
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_vals), n1 * sizeof(float))) != hipSuccess) return bad("hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_feat), n1 * sizeof(int))) != hipSuccess) return bad("hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_out), n1 * sizeof(int))) != hipSuccess) return bad("hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_params), (size_t)cols * sizeof(float4))) != hipSuccess) return bad("hipMalloc");
    if (total) {
        if ((e = hipMemcpy(d_vals, vals.data(), total * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return bad("hipMemcpy");
        if ((e = hipMemcpy(d_feat, feat.data(), total * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bad("hipMemcpy");
    }
    std::vector<int> bucket(total);
    for (int B = 4096; B >= 256; B >>= 1) {
        std::vector<float4> params((size_t)cols);
        for (int c = 0; c < cols; ++c) {
            float lo = 0.f, hi = 0.f;
            bool any = false;
            for (float v : tab[c])
                if (std::isfinite(v)) {
                    lo = any ? std::min(lo, v) : v;
                    hi = any ? std::max(hi, v) : v;
                    any = true;
                }
            float scale = (any && hi > lo) ? (float)B / (hi - lo) : 0.f;
            if (!std::isfinite(scale)) scale = 0.f;
            params[c] = make_float4(lo, scale, 0.f, 0.f);
        }
        if ((e = hipMemcpy(d_params, params.data(), (size_t)cols * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess)
            return bad("hipMemcpy");
        if (total) {
            hipLaunchKernelGGL(bucket_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_vals, d_feat, d_params,
                               (float)(B - 1), (int)total, d_out);
            if ((e = hipGetLastError()) != hipSuccess) return bad("bucket_index_kernel");
            if ((e = hipMemcpy(bucket.data(), d_out, total * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return bad("hipMemcpy");
        }
        // runs, window sizes, LDS need
        std::vector<int> boffsets((size_t)cols + 1, 0);
        std::vector<uint16_t> starts((size_t)cols * (B + 2), 0);
        std::vector<int> steps((size_t)cols, 0);
        bool monotone = true;
        size_t at = 0;
        int len_total = 0;
        for (int c = 0; c < cols; ++c) {
            const int n = (int)tab[c].size();
            uint16_t *st = &starts[(size_t)c * (B + 2)];
            int longest = 0, i = 0;
            for (int b = 0; b <= B; ++b) {
                // st[b] = first index whose bucket is >= b
                while (i < n && bucket[at + i] < b) ++i;
                st[b] = (uint16_t)i;
                if (b > 0) longest = std::max(longest, (int)st[b] - (int)st[b - 1]);
            }
            st[B + 1] = (uint16_t)n;
            for (int k = 0; k < n; ++k) {
                if (bucket[at + k] < 0 || bucket[at + k] >= B || (k > 0 && bucket[at + k] < bucket[at + k - 1])) monotone = false;
            }
            int sp = 0;
            while ((1 << sp) - 1 < longest) ++sp;
            steps[c] = sp;
            boffsets[c] = len_total;
            len_total += n + (1 << sp) - 1;
            memcpy(&params[c].z, &sp, 4);
            at += (size_t)n;
        }
        boffsets[cols] = len_total;
        if (!monotone) break;  // cannot happen with a monotone q_bucket; be safe and keep the search-tree form
        int lds = 0;
        for (int c = 0; c < cols; c += 2) lds = std::max(lds, (boffsets[c + 2] - boffsets[c]) * 4 + 2 * (B + 2) * 2);
        if (lds > f->lds_limit - 256) continue;  // try fewer buckets (smaller start tables)
        std::vector<float> bsorted((size_t)len_total, std::nanf(""));
        at = 0;
        for (int c = 0; c < cols; ++c) {
            std::copy(tab[c].begin(), tab[c].end(), bsorted.begin() + boffsets[c]);
            at += tab[c].size();
        }
        if ((e = q_upload(&g.bsorted, bsorted.data(), bsorted.size(), &f->device_bytes)) != hipSuccess) return bad("bsorted");
        if ((e = q_upload(&g.boffsets, boffsets.data(), boffsets.size(), &f->device_bytes)) != hipSuccess) return bad("boffsets");
        if ((e = q_upload(&g.bstarts, starts.data(), starts.size(), &f->device_bytes)) != hipSuccess) return bad("bstarts");
        if ((e = q_upload(&g.bparams, params.data(), params.size(), &f->device_bytes)) != hipSuccess) return bad("bparams");
        g.buckets = B;
        g.bucket_lds_bytes = lds;
        break;
    }
    cleanup();
    return TAHOE_OK;
}

tahoe_status quantize_build_tables(tahoe_forest *f, const std::vector<std::vector<float>> &tab, tahoe_qgroup &g)
{
    const int cols = f->p.num_cols;
    // device form of each table: perfect BST in level order, 1-based, 2^p entries, NaN padding (compares
    // false, i.e. "greater than every x"); entry 0 unused
    std::vector<int> offsets((size_t)cols + 1, 0);
    std::vector<float> tables;
    int max_size = 1;
    for (int c = 0; c < cols; ++c) {
        const auto &v = tab[c];
        int size = 1;
        while (size - 1 < (int)v.size()) size *= 2;  // 2^p with 2^p - 1 >= n
        max_size = std::max(max_size, size);
        offsets[c] = (int)tables.size();
        const size_t at = tables.size();
        tables.resize(at + (size_t)size, std::nanf(""));
        size_t next = 0;  // in-order walk of the implicit tree assigns the sorted values
        std::vector<int> stack;
        int k = 1;
        while ((k < size) || !stack.empty()) {
            while (k < size) {
                stack.push_back(k);
                k = 2 * k;
            }
            k = stack.back();
            stack.pop_back();
            if (next < v.size()) tables[at + (size_t)k] = v[next++];
            k = 2 * k + 1;
        }
    }
    offsets[cols] = (int)tables.size();

    g.max_table = max_size;
    // quantise kernel form: feature pairs with both trees resident when they fit, else one feature per WG
    g.pair_lds_floats = 0;
    if (cols % 2 == 0) {
        const int budget = 36 * 1024;  // floats: 144 KiB of LDS
        int need = 0;
        for (int c = 0; c < cols; c += 2) {
            const int s0 = offsets[c + 1] - offsets[c], s1 = offsets[c + 2] - offsets[c + 1];
            need = std::max(need, s0 + s1 <= budget ? s0 + s1 : std::max(s0, s1));
        }
        g.pair_lds_floats = std::max(need, 1);
    }
    g.multi_q = 0;
    {
        const char *e = getenv("TAHOE_QUANT_MULTI");  // experiments: 0 keeps the pair kernels
        if (!(e && atoi(e) == 0))
            for (int qd : {kQuantMultiMax, 4, 2})
                if (qd <= kQuantMultiMax && cols % (4 * qd) == 0 && (long long)4 * qd * max_size * 4 <= f->lds_limit - 256) {
                    g.multi_q = qd;
                    break;
                }
    }
    {
        const tahoe_status bs = build_buckets(f, tab, g);
        if (bs != TAHOE_OK) return bs;
    }
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "qring_build: %s failed: %s", what, hipGetErrorString(e)); };
    if ((e = q_upload(&g.tables, tables.data(), tables.size(), &f->device_bytes)) != hipSuccess) return bad("tables");
    if ((e = q_upload(&g.offsets, offsets.data(), offsets.size(), &f->device_bytes)) != hipSuccess) return bad("offsets");
    return TAHOE_OK;
}

void quantize_free_tables(tahoe_qgroup &g)
{
    for (void *p : {(void *)g.tables, (void *)g.offsets, (void *)g.bsorted, (void *)g.boffsets, (void *)g.bstarts, (void *)g.bparams})
        if (p) (void)hipFree(p);
    g.tables = nullptr;
    g.offsets = nullptr;
    g.bsorted = nullptr;
    g.boffsets = nullptr;
    g.bstarts = nullptr;
    g.bparams = nullptr;
}

hipError_t quantize_allow_lds(const tahoe_forest *f)
{
    for (const void *k : {(const void *)&quantize_multi_kernel<8>, (const void *)&quantize_multi_kernel<4>, (const void *)&quantize_multi_kernel<2>,
                          (const void *)&quantize_bucket_pair_kernel, (const void *)&quantize_pair_kernel,
                          (const void *)&quantize_kernel<1>}) {
        const hipError_t e = allow_max_lds(k, f->lds_limit);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

tahoe_status quantize_launch(tahoe_forest *f, const tahoe_qgroup &g, const float *data, size_t rows, int trs, int perm,
                             hipStream_t stream, int *cshift_out)
{
    tahoe_qstate *q = f->q;
    const bool pair_ok = g.pair_lds_floats > 0 && (reinterpret_cast<uintptr_t>(data) % 8) == 0;  // float2 loads
    const bool multi_ok = g.multi_q > 0 && (reinterpret_cast<uintptr_t>(data) % 16) == 0;  // float4 loads
    // Rows per quantise workgroup (2^cshift): as many as 65536 so that staging the tables is amortised, fewer when
    // that would leave the chip short of workgroups (few columns or rows), never so few that the tables outweigh
    // the rows a workgroup converts.
    const int feats = multi_ok ? 4 * g.multi_q : pair_ok ? 2 : 1;
    const size_t fgroups = (size_t)f->p.num_cols / feats;
    const size_t table_bytes = multi_ok ? (size_t)feats * std::max(g.max_table, 1) * 4
                               : (pair_ok && g.buckets > 0) ? (size_t)g.bucket_lds_bytes
                               : pair_ok ? (size_t)g.pair_lds_floats * 4 : (size_t)std::max(g.max_table, 1) * 4;
    int cshift = kQuantMaxShift;
    while ((1 << cshift) > kQuantMinRowsPerBlock && ((rows + ((size_t)1 << cshift) - 1) >> cshift) * fgroups < (size_t)8 * f->num_cus &&
           ((size_t)1 << (cshift - 1)) * feats * 4 >= 2 * table_bytes)
        --cshift;
    const size_t chunks = (rows + ((size_t)1 << cshift) - 1) >> cshift;
    const size_t qgrid = chunks * fgroups;
    if (qgrid > 0x7fffffffu) return fail(TAHOE_ERR_INVALID_ARG, "too many rows x cols for one launch");
    if (multi_ok && g.multi_q == 8)
        hipLaunchKernelGGL(quantize_multi_kernel<8>, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                           (size_t)32 * std::max(g.max_table, 1) * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags,
                           rows, f->p.num_cols, f->p.missing, std::max(g.max_table, 1), trs, cshift, perm);
    else if (multi_ok && g.multi_q == 4)
        hipLaunchKernelGGL(quantize_multi_kernel<4>, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                           (size_t)16 * std::max(g.max_table, 1) * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags,
                           rows, f->p.num_cols, f->p.missing, std::max(g.max_table, 1), trs, cshift, perm);
    else if (multi_ok)
        hipLaunchKernelGGL(quantize_multi_kernel<2>, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                           (size_t)8 * std::max(g.max_table, 1) * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags,
                           rows, f->p.num_cols, f->p.missing, std::max(g.max_table, 1), trs, cshift, perm);
    else if (pair_ok && g.buckets > 0)
        hipLaunchKernelGGL(quantize_bucket_pair_kernel, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                           (size_t)g.bucket_lds_bytes, stream, data, g.bsorted, g.boffsets, g.bstarts, g.bparams, q->xq,
                           q->chunk_flags, rows, f->p.num_cols, f->p.missing, g.buckets, trs, cshift, perm);
    else if (pair_ok)
        hipLaunchKernelGGL(quantize_pair_kernel, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                           (size_t)g.pair_lds_floats * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags, rows,
                           f->p.num_cols, f->p.missing, g.pair_lds_floats, trs, cshift, perm);
    else
        hipLaunchKernelGGL(quantize_kernel<1>, dim3((unsigned)qgrid), dim3(kQuantThreads), (size_t)std::max(g.max_table, 1) * 4,
                           stream, data, g.tables, g.offsets, q->xq, q->chunk_flags, rows, f->p.num_cols, f->p.missing,
                           std::max(g.max_table, 1), trs, cshift, perm);
    TAHOE_HIP_TRY(hipGetLastError());
    *cshift_out = cshift;
    return TAHOE_OK;
}

}  // namespace tahoe
