// QRING: the traversal on rank-quantised features -- exact, and half the LDS bytes.
//
// Idea.  A walk only ever compares a feature value x with thresholds of that same feature.  Let
// tab_f be the sorted distinct (finite or infinite, non-NaN) thresholds the forest uses on feature f,
// and code_f(x) = #{ e in tab_f : e <= x } (float32 compares; 0 for NaN).  A node with threshold
// tab_f[p] then satisfies   x >= thr  <=>  code_f(x) >= p + 1   for every float x, NaN included, so
// the branch rule of infer_one_tree (BaseTahoeTest.h:450-453) is reproduced bit for bit with 16-bit
// integers: the row tile and the node records shrink to half, twice as many rows and three times as
// many trees fit in LDS, and one ds_read_b64 fetches both children of a node.
//   x code    u16: 0..n_f, or 0xFFFF when |x - missing| <= 1e-6 (float32)   (missing takes !def_left)
//   node      u32: threshold code (1..n_f; 0xFFFF for a NaN threshold: never >=) | fid << 16 | def_left << 31
//   top       [t][2^L] u32, 1-based heap positions (children of i = the aligned pair 2i, 2i+1), L <= 10
//   block     [t][2^(De-2)] 32 B = {node0, node1, node2, 0} {leaf0..3 f32}: last two levels + leaves
// Requires num_cols <= 32767 and at most 32767 distinct thresholds per feature within a tree group (the
// per-feature search tree must fit LDS in the quantise kernel; larger forests are cut into groups of
// consecutive trees, see tahoe_qgroup); otherwise the strategy is unavailable and float32 strategies serve.
// Rows too wide for an LDS tile (num_cols > ~550) use the GX form: same kernel, feature codes read from the
// quantised tile in L2 instead of LDS.
//
// Per predict: (1) quantize_kernel turns the row-major float32 batch into 128-row tiles
// xq[tile][fid][128] u16 (rows permuted inside a column so that lane = row reads are bank-conflict
// free), one binary search per value in an LDS-resident table; (2) qring_kernel walks: per CU one
// workgroup = NWALK walker waves (private top slot each, two 64-row chains per lane, next top
// prefetched in registers, no barrier in the tree loop) + one consumer wave that adds the leaf values
// in tree order through an LDS ring -- the float32 sums stay bit-identical to predict_on_cpu.
//
// Replaces, like forest.hip: the adaptive-format walkers/kernels of Struct.h:953-1704 and the
// layout build of Struct.h:1756-1986 (whose char/short/int "adaptive" widths compress only the
// fid/flag word; here the threshold is compressed too, losslessly).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "forest_internal.h"

// One group of consecutive trees with its own quantisation (a forest whose features see more than 32767
// distinct thresholds is cut into groups that each stay below; the running float32 sums are chained from
// group to group, so the result is still the single sequential sum over all trees).
struct tahoe_qgroup {
    int tree_lo = 0, num_trees = 0;
    int max_table = 0;            // floats of the largest per-feature search tree (2^p)
    int pair_lds_floats = 0;      // LDS floats of quantize_pair_kernel; 0 = odd num_cols, single-feature form
    int multi_q = 0;              // quantize_multi_kernel<Q>: 4 or 2 (16 / 8 features per workgroup); 0 = tables too large
    float *tables = nullptr;      // concatenated search trees
    int *offsets = nullptr;       // [cols + 1]
    // bucketed form (quantize_bucket_pair_kernel): sorted thresholds + a direct-index table per feature
    int buckets = 0;              // B (power of two); 0 = form unavailable for this group
    int bucket_lds_bytes = 0;     // LDS of the largest feature pair
    float *bsorted = nullptr;     // per feature: n_f sorted thresholds + (2^steps_f - 1) NaN pads
    int *boffsets = nullptr;      // [cols + 1] into bsorted
    uint16_t *bstarts = nullptr;  // [cols][B + 2]: first sorted index of each bucket (entry B = n_f)
    float4 *bparams = nullptr;    // [cols]: lo, scale, steps (int bits), unused
    uint32_t *top = nullptr;      // [T_g][top_stride]
    uint4 *blocks = nullptr;      // [T_g][2^(De-2)][2]
    uint32_t *qinner = nullptr;   // [T_g][2^De - 1] (only when have_mid)
};

struct tahoe_qstate {
    int top_levels = 0;
    bool have_mid = false;        // De - 2 > top_levels: heap of quantised nodes for the middle levels
    int top_stride = 0;           // u32 entries per tree in `top` (>= 4)
    bool narrow = false;          // node words in the NARROW layout (num_cols <= 256, 15 walkers, LDS tile)
    int wide_rt = 0;              // rows per tile of the wide-row form (qwide_kernel), fixed at create; 0 = not used
    int wide_lw = 0;              // ... and the top levels its LDS slots hold
    std::vector<tahoe_qgroup> groups;
    uint16_t *xq = nullptr;       // workspace: quantised tiles (re-used by every group)
    size_t xq_rows = 0;           // rows the workspace holds
    uint32_t *chunk_flags = nullptr;  // workspace: per kQuantRowsPerBlock rows, "a missing value was seen"
    size_t n_chunk_flags = 0;
};

namespace tahoe {

constexpr int kQRows = 128;                 // rows per tile
constexpr int kQRing = 16;                  // ring entries (trees)
constexpr int kQBatch = 8;                  // trees the consumer takes per poll
constexpr int kQSpinLimit = 1 << 22;
constexpr int kQSlotBytes = 4096;            // LDS per walker: a 10-level top (2^10 u32)
constexpr int kQMaxTable = 32767;
constexpr int kQuantMaxShift = 15;          // a quantise workgroup converts 2^cshift rows of its features; at most 32768
constexpr int kQuantMinRowsPerBlock = 512;  // ... and the fewest
constexpr int kQuantThreads = 512;
constexpr uint32_t kCodeMissing = 0xFFFFu;

// position of tile row r (0..127) inside a feature column of 128 u16: within a 32-lane group the
// rows land in 32 different LDS banks (two rows per dword come from different groups)
__host__ __device__ __forceinline__ int qrow_pos(int r) { return ((r & 31) << 1) | ((r >> 5) & 1) | ((r >> 6) << 6); }
// index of (row r, feature f) in the quantised workspace: tiles of 2^trs rows, xq[tile][f][2^trs]; 128-row tiles
// permute the rows inside a column (qrow_pos), the smaller tiles of the wide-row form keep them in order
__device__ __forceinline__ size_t q_tile_index(size_t r, int f, int cols, int trs)
{
    const int rr = (int)(r & (((size_t)1 << trs) - 1));
    return (((r >> trs) * (size_t)cols + (size_t)f) << trs) + (size_t)(trs == 7 ? qrow_pos(rr) : rr);
}

// ------------------------------------------------------------------------------------------------
// (1) float32 rows -> u16 codes.  One workgroup = F adjacent features x kQuantRowsPerBlock rows, the F
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
                                                                 float missing, int tab_stride, int trs, int cshift)
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
                    xq[q_tile_index(r, f0 + j, cols, trs)] =
                        (uint16_t)code;
                }
            }
        }
    }
    // one atomic per wave at most: tells the walk kernel whether this row chunk needs the missing rule
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// Pair form used when num_cols is even: one workgroup = features (f0, f0+1) x kQuantRowsPerBlock rows, float2
// loads.  When both search trees fit the LDS budget they are resident together (one pass); a pair with an
// oversized tree is done in two passes with one tree resident at a time, so a few large features do not
// force the whole launch down to one feature per workgroup.
constexpr int kQuantPairThreads = 1024;
__global__ void __launch_bounds__(kQuantPairThreads)
    quantize_pair_kernel(const float *__restrict__ data, const float *__restrict__ tables, const int *__restrict__ offsets,
                         uint16_t *__restrict__ xq, uint32_t *__restrict__ chunk_flags, size_t rows, int cols, float missing,
                         int lds_floats, int trs, int cshift)
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
        for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)blockDim.x * U) {
            float2 xv[U];
            int c0[U], c1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t r = min(rb + (size_t)u * blockDim.x, r1 - 1);  // clamped: in bounds, result unused
                xv[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
                c0[u] = 1;
                c1[u] = 1;
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
                    uint16_t *dst = xq + q_tile_index(r, f0, cols, trs);
                    if (do0) {
                        const bool ms = fabsf(xv[u].x - missing) <= kMissingEps;
                        saw_missing |= ms;
                        dst[0] = (uint16_t)(ms ? kCodeMissing : (uint32_t)(c0[u] - size0));
                    }
                    if (do1) {
                        const bool ms = fabsf(xv[u].y - missing) <= kMissingEps;
                        saw_missing |= ms;
                        dst[(size_t)1 << trs] = (uint16_t)(ms ? kCodeMissing : (uint32_t)(c1[u] - size1));
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
                                int B, int trs, int cshift)
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
    for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)blockDim.x * U) {
        float2 xv[U];
        int c0[U], c1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)u * blockDim.x, r1 - 1);  // clamped: in bounds, result unused
            xv[u] = *reinterpret_cast<const float2 *>(data + r * (size_t)cols + f0);
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
                uint16_t *dst = xq + q_tile_index(r, f0, cols, trs);
                const bool ms0 = fabsf(xv[u].x - missing) <= kMissingEps, ms1 = fabsf(xv[u].y - missing) <= kMissingEps;
                saw_missing |= ms0 | ms1;
                dst[0] = (uint16_t)(ms0 ? kCodeMissing : (uint32_t)c0[u]);
                dst[(size_t)1 << trs] = (uint16_t)(ms1 ? kCodeMissing : (uint32_t)c1[u]);
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
                          int tab_stride, int trs, int cshift)
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
    for (size_t rb = r0 + rsub; rb < r1; rb += (size_t)RPI * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)u * RPI, r1 - 1);  // clamped: in bounds, result unused
            xv[u] = *reinterpret_cast<const float4 *>(data + r * (size_t)cols + fq);
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
                    xq[q_tile_index(r, fq + j, cols, trs)] = (uint16_t)(ms ? kCodeMissing : (uint32_t)(cnt[u] - size[j]));
                }
            }
        }
    }
    if (__ballot(saw_missing) != 0 && (threadIdx.x & 63) == 0) atomicOr(&chunk_flags[chunk], 1u);
}

// Ring flags: relaxed workgroup-scope accesses (plain ds_read/ds_write that the compiler neither caches
// in a register nor reorders across the asm memory barriers around them).
__device__ __forceinline__ uint32_t lds_flag_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The branch rule on codes: right <=> (missing ? !def_left : code(x) >= code(thr)).  Written on wave
// masks: three v_cmp into SGPR pairs, three SALU ops, and the result is used directly as the lane
// predicate of v_cndmask / v_addc (hipcc's ?: form materialises both booleans in VGPRs: 6 more VALU).
// MS = false is the fast path for row chunks in which the quantise pass met no missing value (it
// reports that per chunk): the rule is then the single compare.
// NARROW (num_cols <= 256): node = code << 16 | fid << 8 | def_left, so that the feature column's LDS offset
// (fid * 256) is a bit field of the node word and one v_bfi forms the read address (q_xread).
template <bool MS, bool NARROW>
__device__ __forceinline__ uint64_t q_right_mask(uint32_t xc, uint32_t node)
{
    const uint64_t ge = __builtin_amdgcn_uicmp(xc, NARROW ? node >> 16 : node & 0xFFFFu, 35 /* ICMP_UGE */);
    if (!MS) return ge;
    const uint64_t ms = __builtin_amdgcn_uicmp(xc, kCodeMissing, 32 /* ICMP_EQ */);
    const uint64_t ndl = NARROW ? __builtin_amdgcn_uicmp(node & 0xFFu, 0u, 32 /* ICMP_EQ: def_left clear */)
                                : __builtin_amdgcn_sicmp((int)node, -1, 38 /* ICMP_SGT: bit 31 (def_left) clear */);
    return (ge & ~ms) | (ms & ndl);
}
template <bool MS, bool NARROW>
__device__ __forceinline__ bool q_go_right(uint32_t xc, uint32_t node)
{
    return __builtin_amdgcn_inverse_ballot_w64(q_right_mask<MS, NARROW>(xc, node));
}
// i <- 2i + (lane's bit of mask): one v_addc with the mask as carry-in
__device__ __forceinline__ uint32_t q_descend(uint32_t i, uint64_t right_mask)
{
    uint32_t r;
    uint64_t carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=&s"(carry_out) : "v"(i), "s"(right_mask));
    return r;
}
// The u16 code of feature fid(node) for this lane's row.  `posb` = LDS byte address of the row's slot in
// feature column 0 (tile base + position); a column is 256 bytes.  The address is formed as an LDS
// (address-space 3) integer, fid * 256 + posb: v_bfe + v_lshl_add, two VALU.  (Going through the generic
// `tile` pointer costs a third: hipcc adds the LDS base, a link-time 0, with its own v_add.)
// LDSX = false ("GX" form, for rows too wide for an LDS tile): the same 256-byte columns are read straight from
// the quantised tile in global memory (L2-resident: one 128-row tile of 3072 columns is 768 KiB); `gx` = the
// tile's base, posb = byte position inside a column.
typedef const uint16_t __attribute__((address_space(3))) *lds_u16_ptr;
template <bool LDSX, bool NARROW, int CSHIFT = 8>
__device__ __forceinline__ uint32_t q_xread(const unsigned char *gx, uint32_t node, uint32_t posb)
{
    static_assert(!NARROW || CSHIFT == 8, "the NARROW layout is defined for 256-byte feature columns");
    if (LDSX && NARROW) {
        // the tile starts at LDS address 0 (checked at kernel entry) and posb < 256: address = node[15:8] : posb[7:0]
        uint32_t addr;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr) : "s"(0xFF00u), "v"(node), "v"(posb));
        return *reinterpret_cast<lds_u16_ptr>(addr);
    }
    if (LDSX) {
        uint32_t addr;  // asm: hipcc re-canonicalises the C form into shift + and + add
        asm("v_bfe_u32 %0, %1, 16, 15\n\tv_lshl_add_u32 %0, %0, %3, %2" : "=&v"(addr) : "v"(node), "v"(posb), "n"(CSHIFT));
        return *reinterpret_cast<lds_u16_ptr>(addr);
    }
    if (NARROW) return *reinterpret_cast<const uint16_t *>(gx + (node & 0xFF00u) + posb);
    return *reinterpret_cast<const uint16_t *>(gx + (((node >> 16) & 0x7fffu) << CSHIFT) + posb);
}

// ------------------------------------------------------------------------------------------------
// (2) the walk.  Dynamic LDS: tile [cols][128] u16 | NWALK slots of (4 << L) bytes | ring [16][128] f32 |
// ready[16] | consumed.
template <int NWALK, bool WRITE_LEAF, bool LDSX, bool NARROW = false>
__global__ void __launch_bounds__((NWALK + 1) * 64)
    qring_kernel(const uint16_t *__restrict__ xq, const uint32_t *__restrict__ top, const uint4 *__restrict__ blocks,
                 const uint32_t *__restrict__ qinner, const uint32_t *__restrict__ leaf_orig, float *__restrict__ sums,
                 uint32_t *__restrict__ leaf_out, size_t rows, int cols, int num_trees, int depth, int top_levels,
                 int top_stride, const uint32_t *__restrict__ chunk_flags, int *__restrict__ error_flag,
                 const float *__restrict__ sums_in, int tree_base, int total_trees, int cshift)
{
    constexpr int K = kQRows / 64;  // two 64-row chains per walker lane
    constexpr int NT = (NWALK + 1) * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int slot_bytes = kQSlotBytes;  // fixed 4 KiB slot (2^10 u32): no size-dependent branches in the loop

    uint16_t *tile = reinterpret_cast<uint16_t *>(smem);  // LDSX only
    unsigned char *slots = smem + (LDSX ? (size_t)cols * kQRows * sizeof(uint16_t) : 0);
    const unsigned char *gx = reinterpret_cast<const unsigned char *>(xq + (size_t)blockIdx.x * ((size_t)cols * kQRows));
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * slot_bytes);
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + kQRing * kQRows);
    uint32_t *consumed = ring_ready + kQRing;

    const size_t row0 = (size_t)blockIdx.x * kQRows;
    if (LDSX && NARROW && (uint32_t)reinterpret_cast<uintptr_t>(tile) != 0u) {
        // q_xread's v_bfi needs the tile at LDS address 0 (true while the kernel has no static LDS)
        if (tid == 0) atomicOr(error_flag, 2);
        return;
    }

    // ---- stage the quantised tile (already in LDS order): straight 16-byte copies ----
    if (LDSX) {
        const uint4 *src = reinterpret_cast<const uint4 *>(xq + (size_t)blockIdx.x * ((size_t)cols * kQRows));
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
        const int n16 = cols * kQRows * 2 / 16;
        for (int e = tid; e < n16; e += NT) dst[e] = src[e];
    }
    if (tid < kQRing) ring_ready[tid] = 0u;
    if (tid == kQRing) *consumed = 0u;

    if (wave == NWALK) {
        // ================= consumer: ordered accumulation =================
        __syncthreads();
        float sum[K];  // continues the running sums of the previous tree group (sums_in may alias sums)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t irow = row0 + k * 64 + lane;
            sum[k] = (sums_in && irow < rows) ? sums_in[irow] : 0.0f;
        }
        bool dead = false;
        for (int t0 = 0; t0 < num_trees && !dead; t0 += kQBatch) {
            const int nb = min(kQBatch, num_trees - t0);
            int spins = 0;
            for (;;) {
                const bool ok = lane >= nb || lds_flag_load(&ring_ready[(t0 + lane) % kQRing]) == (uint32_t)(t0 + lane + 1);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kQSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            asm volatile("" ::: "memory");  // the values are read after the flags
            for (int j = 0; j < nb; ++j) {
                const int e = (t0 + j) % kQRing;
#pragma unroll
                for (int k = 0; k < K; ++k) sum[k] += ring_vals[e * kQRows + k * 64 + lane];  // tree order
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_flag_store(consumed, (uint32_t)(t0 + nb));
        }
        if (dead && lane == 0) atomicOr(error_flag, 1);
        if (sums) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const size_t orow = row0 + k * 64 + lane;
                if (orow < rows) sums[orow] = sum[k];
            }
        }
        return;
    }

    // ================= walkers =================
    uint32_t *slot = reinterpret_cast<uint32_t *>(slots + (size_t)wave * slot_bytes);
    const int n_chunks = (top_stride * 4) >> 4;  // 16-byte chunks of a top in global memory (<= 256)
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};   // named registers (an indexed array would go to scratch)
    auto prefetch_top = [&](int t) {
        const uint4 *g = reinterpret_cast<const uint4 *>(top + (size_t)t * top_stride);
        const int last = n_chunks - 1;  // clamped: branch-free and in bounds
        pf0 = g[min(0 * 64 + lane, last)];
        pf1 = g[min(1 * 64 + lane, last)];
        pf2 = g[min(2 * 64 + lane, last)];
        pf3 = g[min(3 * 64 + lane, last)];
    };
    // Unconditional stores (a smaller top just rewrites its last chunk into unused slot space): with
    // per-lane conditions hipcc branches around each store AND its s_waitcnt, leaves the load "pending" on
    // the skipped path and then waits for the newest gathers at the loop latch.
    auto commit_top = [&]() {
        uint4 *s = reinterpret_cast<uint4 *>(slot);
        s[0 * 64 + lane] = pf0;
        s[1 * 64 + lane] = pf1;
        s[2 * 64 + lane] = pf2;
        s[3 * 64 + lane] = pf3;
    };
    if (wave < num_trees) {
        prefetch_top(wave);
        commit_top();
    }
    __syncthreads();  // the tile, the ring state and (own wave) the first top are in LDS

    bool dead = false;
    auto run = [&](auto ms_tag) {
        constexpr bool MS = decltype(ms_tag)::value;
        uint32_t pos[K];  // byte position of this lane's row inside a 256-byte feature column, per chain
    #pragma unroll
        for (int k = 0; k < K; ++k)  // low 32 bits of a generic LDS pointer = the LDS byte address
            pos[k] = (LDSX ? (uint32_t)reinterpret_cast<uintptr_t>(tile) : 0u) + 2u * (uint32_t)qrow_pos(k * 64 + lane);
        const size_t n_inner = ((size_t)1 << depth) - 1;
        const uint32_t n_blocks = 1u << (depth - 2);
        const uint32_t first_block_node = n_blocks - 1;
        // last two levels + leaf of tree t from its 32-byte blocks, then the hand-over to the consumer
        auto finish = [&](int t, const uint4 (&na)[K], const uint4 (&nb)[K], const uint32_t (&bs)[K]) {
            float v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // two dependent steps (the kernel is VALU-bound: this is ~half the instructions of evaluating
                // both second-level nodes)
                const bool c0 = q_go_right<MS, NARROW>(q_xread<LDSX, NARROW>(gx, na[k].x, pos[k]), na[k].x);
                const uint32_t n1 = c0 ? na[k].z : na[k].y;
                const bool c1 = q_go_right<MS, NARROW>(q_xread<LDSX, NARROW>(gx, n1, pos[k]), n1);
                const uint32_t lo = c0 ? nb[k].z : nb[k].x, hi = c0 ? nb[k].w : nb[k].y;
                v[k] = __uint_as_float(c1 ? hi : lo);
                if (WRITE_LEAF) {
                    const size_t row = row0 + k * 64 + lane;
                    if (row < rows)
                        leaf_out[row * (size_t)total_trees + tree_base + t] =
                            leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bs[k] * 4 + 2 * (c0 ? 1 : 0) + (c1 ? 1 : 0)];
                }
            }
            if (t >= kQRing) {  // ring entry still in use by tree t - kQRing?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t - kQRing + 1)) {
                    if (++spins > kQSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const int e = t % kQRing;
    #pragma unroll
            for (int k = 0; k < K; ++k) ring_vals[e * kQRows + k * 64 + lane] = v[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // values before the flag (in-order LDS)
            if (lane == 0) lds_flag_store(&ring_ready[e], (uint32_t)(t + 1));
        };
        int t_p = -1;  // tree whose bottom blocks are in flight
        uint4 na_p[K] = {}, nb_p[K] = {};
        uint32_t bsel_p[K] = {};
        for (int t = wave; t < num_trees && !dead; t += NWALK) {
#if defined(TAHOE_ABLATE) && (TAHOE_ABLATE & 2)
            const bool more = false;  // timing-only build: every tree walks the first top
#else
            const bool more = t + NWALK < num_trees;
#endif
            if (more) prefetch_top(t + NWALK);
            uint32_t i[K];
    #pragma unroll
            for (int k = 0; k < K; ++k) i[k] = 1;
            if (top_levels > 0) {
                uint32_t node[K];
    #pragma unroll
                for (int k = 0; k < K; ++k) node[k] = slot[1];
                // Both children come with one ds_read_b64 issued beside the feature read, ahead of the compare.  (Reading
                // only the chosen child afterwards -- half the LDS bytes, twice the round trips -- times the same to
                // 0.5 %: neither LDS bandwidth nor LDS latency bounds this loop, see DESIGN.md.)
                for (int l = 0; l < top_levels - 1; ++l) {
                    uint32_t xc[K];
                    uint2 pr[K];
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        xc[k] = q_xread<LDSX, NARROW>(gx, node[k], pos[k]);
                        pr[k] = *reinterpret_cast<const uint2 *>(&slot[2 * i[k]]);  // children 2i, 2i+1
                    }
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint64_t cm = q_right_mask<MS, NARROW>(xc[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr[k].y : pr[k].x;
                    }
                }
    #pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t xc = q_xread<LDSX, NARROW>(gx, node[k], pos[k]);
                    i[k] = q_descend(i[k], q_right_mask<MS, NARROW>(xc, node[k]));
                }
            }
            uint32_t bsel[K];
    #pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t idx = i[k] - 1;  // 0-based heap index on level top_levels
                if (top_levels < depth - 2) {  // deep trees only (De > 12): quantised heap nodes from global memory
                    const uint32_t *tree = qinner + (size_t)t * n_inner;
                    for (int l = top_levels; l < depth - 2; ++l) {
                        const uint32_t n = tree[idx];
                        const uint32_t xc = q_xread<LDSX, NARROW>(gx, n, pos[k]);
                        idx = 2u * idx + 1u + (q_go_right<MS, NARROW>(xc, n) ? 1u : 0u);
                    }
                }
                bsel[k] = idx - first_block_node;
            }
            // ---- software pipeline: finish the PREVIOUS tree (its bottom-block gathers were issued one
            // iteration ago and have been flying under this top walk), then issue this tree's gathers
            // into the same registers ----
            if (t_p >= 0) finish(t_p, na_p, nb_p, bsel_p);
            t_p = t;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint4 *bp = blocks + ((size_t)t * n_blocks + bsel[k]) * 2;
                na_p[k] = bp[0];  // node0, node1, node2, 0
#if defined(TAHOE_ABLATE) && (TAHOE_ABLATE & 1)
                nb_p[k] = na_p[k];  // timing-only build: no leaf gather
#else
                nb_p[k] = bp[1];  // four leaf values
#endif
                bsel_p[k] = bsel[k];
            }
            if (more) commit_top();  // this wave's reads of its slot are done (in-order LDS)
        }
        if (t_p >= 0 && !dead) finish(t_p, na_p, nb_p, bsel_p);
    };
    // chunk_flags[c] != 0 <=> the quantise pass met a missing value in rows [c, c+1) << cshift
    if (chunk_flags[row0 >> cshift] != 0)
        run(std::true_type{});
    else
        run(std::false_type{});
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// (3) the walk for wide rows ("QWIDE").  When num_cols is too large for a 128-row u16 tile in LDS (K2: 3072
// columns = 768 KiB) the tile shrinks to RT = 64 / 32 / 16 rows and a walker wave walks TPW = 64 / RT trees at
// once: lane = (tree slot j, row r).  Everything else is the scheme above -- private top slots (TPW small tops per
// wave), 32-byte bottom blocks gathered one group ahead, a consumer wave adding leaf values in tree order through
// an LDS ring -- so the sums stay bit-identical.  (The alternative, 128-row tiles read from L2 -- the GX form --
// thrashes: 32 workgroups per XCD x 768 KiB against a 4 MiB L2.)  Tiles are xq[tile][fid][RT] u16, rows in order.
// Tops: the first 2^LW heap entries of each tree's `top` array, LW = min(top_levels, log2(1024 / TPW)): a 4 KiB slot
// per wave; levels LW .. De-3 come from the quantised heap in global memory (qinner).
constexpr int kWideRingBytes = 8192;
template <int RT, bool WRITE_LEAF>
__global__ void __launch_bounds__(16 * 64)
    qwide_kernel(const uint16_t *__restrict__ xq, const uint32_t *__restrict__ top, const uint4 *__restrict__ blocks,
                 const uint32_t *__restrict__ qinner, const uint32_t *__restrict__ leaf_orig, float *__restrict__ sums,
                 uint32_t *__restrict__ leaf_out, size_t rows, int cols, int num_trees, int depth, int top_levels,
                 int top_stride, const uint32_t *__restrict__ chunk_flags, int *__restrict__ error_flag,
                 const float *__restrict__ sums_in, int tree_base, int total_trees, int slot_bytes, int lw, int cshift)
{
    constexpr int NWALK = 15;
    constexpr int NT = (NWALK + 1) * 64;
    constexpr int TPW = 64 / RT;                       // trees a walker wave walks at once
    constexpr int RE = kWideRingBytes / (RT * 4);      // ring entries (trees): >= two rounds of NWALK * TPW
    constexpr int NBATCH = RE / 2;                     // trees the consumer takes per poll (<= 64)
    constexpr int CSHIFT = RT == 64 ? 7 : RT == 32 ? 6 : 5;  // log2 of a feature column's bytes
    static_assert(NBATCH <= 64 && RE >= 2 * NWALK * TPW, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *tile = reinterpret_cast<uint16_t *>(smem);
    unsigned char *slots = smem + (size_t)cols * RT * sizeof(uint16_t);
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * slot_bytes);  // slot = TPW tops, <= 4 KiB
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + RE * RT);
    uint32_t *consumed = ring_ready + RE;
    const size_t row0 = (size_t)blockIdx.x * RT;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(xq + (size_t)blockIdx.x * ((size_t)cols * RT));
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
        const int n16 = cols * RT * 2 / 16;
        for (int e = tid; e < n16; e += NT) dst[e] = src[e];
    }
    for (int e = tid; e < RE; e += NT) ring_ready[e] = 0u;
    if (tid == 0) *consumed = 0u;

    if (wave == NWALK) {
        // ================= consumer: ordered accumulation, lane = row =================
        __syncthreads();
        const size_t irow = row0 + lane;
        float sum = (sums_in && lane < RT && irow < rows) ? sums_in[irow] : 0.0f;
        bool dead = false;
        for (int t0 = 0; t0 < num_trees && !dead; t0 += NBATCH) {
            const int nb = min(NBATCH, num_trees - t0);
            int spins = 0;
            for (;;) {
                const bool ok = lane >= nb || lds_flag_load(&ring_ready[(t0 + lane) % RE]) == (uint32_t)(t0 + lane + 1);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kQSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            asm volatile("" ::: "memory");  // the values are read after the flags
            if (lane < RT)
                for (int j = 0; j < nb; ++j) sum += ring_vals[((t0 + j) % RE) * RT + lane];  // tree order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_flag_store(consumed, (uint32_t)(t0 + nb));
        }
        if (dead && lane == 0) atomicOr(error_flag, 1);
        if (sums && lane < RT && irow < rows) sums[irow] = sum;
        return;
    }

    // ================= walkers =================
    const int j = lane / RT, r = lane % RT;           // tree slot, row of the tile
    // lw (<= min(top_levels, LWMAX), chosen at create so that tile + slots fit): levels served from the LDS slot
    const int cpt = max(1, (1 << lw) >> 2);           // 16-byte chunks per staged top (top_stride >= 4 entries)
    unsigned char *wslot = slots + (size_t)wave * slot_bytes;
    const uint32_t *slot = reinterpret_cast<const uint32_t *>(wslot + (size_t)j * cpt * 16);
    const int n_groups = (num_trees + TPW - 1) / TPW;
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};
    // chunk c of the wave's slot = chunk (c % cpt) of tree g * TPW + c / cpt (clamped: in bounds, branch-free)
    auto prefetch_tops = [&](int g) {
        auto ld = [&](int c) {
            c = min(c, TPW * cpt - 1);
            const int t = min(g * TPW + c / cpt, num_trees - 1);
            return reinterpret_cast<const uint4 *>(top + (size_t)t * top_stride)[c % cpt];
        };
        pf0 = ld(0 * 64 + lane);
        pf1 = ld(1 * 64 + lane);
        pf2 = ld(2 * 64 + lane);
        pf3 = ld(3 * 64 + lane);
    };
    auto commit_tops = [&]() {  // TPW * cpt <= 256 chunks; clamped lanes rewrite the last chunk with its own value
        uint4 *s = reinterpret_cast<uint4 *>(wslot);
        const int last = TPW * cpt - 1;
        s[min(0 * 64 + lane, last)] = pf0;
        s[min(1 * 64 + lane, last)] = pf1;
        s[min(2 * 64 + lane, last)] = pf2;
        s[min(3 * 64 + lane, last)] = pf3;
    };
    if (wave < n_groups) {
        prefetch_tops(wave);
        commit_tops();
    }
    __syncthreads();

    bool dead = false;
    auto run = [&](auto ms_tag) {
        constexpr bool MS = decltype(ms_tag)::value;
        const uint32_t pos = (uint32_t)reinterpret_cast<uintptr_t>(tile) + 2u * (uint32_t)r;
        const size_t n_inner = ((size_t)1 << depth) - 1;
        const uint32_t n_blocks = 1u << (depth - 2);
        const uint32_t first_block_node = n_blocks - 1;
        const size_t row = row0 + r;
        auto finish = [&](int g, const uint4 &na, const uint4 &nb, uint32_t bs) {
            const int t = g * TPW + j;
            const bool c0 = q_go_right<MS, false>(q_xread<true, false, CSHIFT>(nullptr, na.x, pos), na.x);
            const uint32_t n1 = c0 ? na.z : na.y;
            const bool c1 = q_go_right<MS, false>(q_xread<true, false, CSHIFT>(nullptr, n1, pos), n1);
            const uint32_t lo = c0 ? nb.z : nb.x, hi = c0 ? nb.w : nb.y;
            const float v = __uint_as_float(c1 ? hi : lo);
            if (WRITE_LEAF) {
                if (t < num_trees && row < rows)
                    leaf_out[row * (size_t)total_trees + tree_base + t] =
                        leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bs * 4 + 2 * (c0 ? 1 : 0) + (c1 ? 1 : 0)];
            }
            const int t_last = min(g * TPW + TPW - 1, num_trees - 1);
            if (t_last >= RE) {  // the group's ring entries still in use?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t_last - RE + 1)) {
                    if (++spins > kQSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (t < num_trees) ring_vals[(t % RE) * RT + r] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // values before the flags (in-order LDS)
            if (r == 0 && t < num_trees) lds_flag_store(&ring_ready[t % RE], (uint32_t)(t + 1));
        };
        int g_p = -1;  // group whose bottom blocks are in flight
        uint4 na_p = {}, nb_p = {};
        uint32_t bsel_p = 0;
        for (int g = wave; g < n_groups && !dead; g += NWALK) {
            const bool more = g + NWALK < n_groups;
            if (more) prefetch_tops(g + NWALK);
            const int t = min(g * TPW + j, num_trees - 1);  // lanes of a missing tree repeat the last one, unused
            uint32_t i = 1;
            if (lw > 0) {
                uint32_t node = slot[1];
                for (int l = 0; l < lw - 1; ++l) {
                    const uint32_t xc = q_xread<true, false, CSHIFT>(nullptr, node, pos);
                    const uint2 pr = *reinterpret_cast<const uint2 *>(&slot[2 * i]);  // children 2i, 2i+1
                    const uint64_t cm = q_right_mask<MS, false>(xc, node);
                    i = q_descend(i, cm);
                    node = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr.y : pr.x;
                }
                const uint32_t xc = q_xread<true, false, CSHIFT>(nullptr, node, pos);
                i = q_descend(i, q_right_mask<MS, false>(xc, node));
            }
            uint32_t idx = i - 1;  // 0-based heap index on level lw
            if (lw < depth - 2) {  // the levels between the slot and the bottom blocks: quantised heap in global memory
                const uint32_t *tree = qinner + (size_t)t * n_inner;
                for (int l = lw; l < depth - 2; ++l) {
                    const uint32_t n = tree[idx];
                    const uint32_t xc = q_xread<true, false, CSHIFT>(nullptr, n, pos);
                    idx = 2u * idx + 1u + (q_go_right<MS, false>(xc, n) ? 1u : 0u);
                }
            }
            const uint32_t bsel = idx - first_block_node;
            if (g_p >= 0) finish(g_p, na_p, nb_p, bsel_p);
            g_p = g;
            const uint4 *bp = blocks + ((size_t)t * n_blocks + bsel) * 2;
            na_p = bp[0];
            nb_p = bp[1];
            bsel_p = bsel;
            if (more) commit_tops();
        }
        if (g_p >= 0 && !dead) finish(g_p, na_p, nb_p, bsel_p);
    };
    if (chunk_flags[row0 >> cshift] != 0)
        run(std::true_type{});
    else
        run(std::false_type{});
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

// ------------------------------------------------------------------------------------------------
static long long qring_lds_for(const tahoe_forest *f, int nwalk, bool lds_tile = true)
{
    return (lds_tile ? (long long)f->p.num_cols * kQRows * 2 : 0) + (long long)nwalk * kQSlotBytes +
           (long long)kQRing * kQRows * 4 + (kQRing + 1) * 4LL;
}

constexpr int kGxWalkers = 15;  // walkers of the GX form (no LDS tile)
// true: the 128-row u16 tile fits LDS beside at least 4 walkers; false: GX form
bool qring_lds_tile(const tahoe_forest *f) { return qring_lds_for(f, 4) <= f->lds_limit; }

// Wide form, `rt` rows per tile: a wave's slot holds TPW = 64 / rt tops of 2^lw u32, at most 4 KiB
static int qwide_lw_max(int top_levels, int rt) { return std::min(top_levels, rt == 64 ? 10 : rt == 32 ? 9 : 8); }
static long long qwide_slot_bytes(int lw, int rt) { return (64 / rt) * (long long)std::max(1, (1 << lw) >> 2) * 16; }
static long long qwide_lds_for(const tahoe_forest *f, int rt, int lw)
{
    return (long long)f->p.num_cols * rt * 2 + 15LL * qwide_slot_bytes(lw, rt) + kWideRingBytes +
           (kWideRingBytes / (rt * 4) + 1) * 4LL;
}
// rows per tile of the wide-row form (qwide_kernel); 0 = not used (the 128-row tile fits, or not even 16 rows do)
int qwide_rows(const tahoe_forest *f) { return f->q ? f->q->wide_rt : 0; }
// Largest tile first; within a tile size up to three top levels may move from the LDS slots to the global heap to
// make room for the tile.
static void qwide_pick(const tahoe_forest *f, int *rt_out, int *lw_out)
{
    *rt_out = *lw_out = 0;
    if (!f->q || qring_lds_tile(f)) return;
    if (const char *e = getenv("TAHOE_QRING_WIDE"))  // experiments: 0 keeps the GX form
        if (atoi(e) == 0) return;
    for (int rt : {64, 32, 16}) {
        const int hi = qwide_lw_max(f->q->top_levels, rt);
        for (int lw = hi; lw >= std::max(0, hi - 3); --lw)
            if (qwide_lds_for(f, rt, lw) <= f->lds_limit) {
                *rt_out = rt;
                *lw_out = lw;
                return;
            }
    }
}

int qring_walkers(const tahoe_forest *f)
{
    if (!f->q) return 0;
    if (f->q->narrow) return 15;  // the node words were encoded for that form at create
    static const int options[] = {15, 12, 8, 4};
    if (const char *e = getenv("TAHOE_QRING_WALKERS")) {  // tuning knob for experiments
        const int want = atoi(e);
        for (int n : options)
            if (n == want && qring_lds_for(f, n) <= f->lds_limit) return n;
    }
    for (int n : options)
        if (qring_lds_for(f, n) <= f->lds_limit) return n;
    return kGxWalkers;  // rows too wide for an LDS tile: features are read from the quantised tile in L2
}

long long qring_lds_bytes(const tahoe_forest *f)
{
    const int n = qring_walkers(f);
    return n ? qring_lds_for(f, n, qring_lds_tile(f)) : 0;
}

template <typename T>
static hipError_t q_upload(T **dst, const T *src, size_t count, size_t *total)
{
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes);
    if (e != hipSuccess) return e;
    *total += bytes;
    if (count) e = hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

template <int NWALK>
static hipError_t q_allow(long long lds)
{
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<NWALK, false, true>), (int)lds);
    if (e != hipSuccess) return e;
    e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<NWALK, true, true>), (int)lds);
    if (e != hipSuccess || NWALK != 15) return e;
    e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<15, false, true, true>), (int)lds);
    if (e != hipSuccess) return e;
    return allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<15, true, true, true>), (int)lds);
}

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

// Builds one tree group [lo, hi).  Returns TAHOE_OK with *too_many = the largest per-feature count when that
// exceeds the limit (nothing is allocated then).
static tahoe_status build_group(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                                const std::vector<float> &h_leaf, size_t lo, size_t hi, tahoe_qgroup &g, int *too_many)
{
    tahoe_qstate *q = f->q;
    const int cols = f->p.num_cols;
    const size_t n_inner = f->n_inner, n_leaf = f->n_leaf;
    *too_many = 0;
    // ---- per-feature tables of distinct thresholds ----
    std::vector<std::vector<float>> tab((size_t)cols);
    for (size_t i = lo * n_inner; i < hi * n_inner; ++i) {
        if (!h_real[i] || std::isnan(h_inner[i].thr)) continue;
        tab[h_inner[i].meta & 0x7fffffffu].push_back(h_inner[i].thr);
    }
    int max_count = 0;
    for (int c = 0; c < cols; ++c) {
        auto &v = tab[c];
        std::sort(v.begin(), v.end());                       // float order; -0.0f and 0.0f compare equal
        v.erase(std::unique(v.begin(), v.end()), v.end());   // ... and collapse into one entry
        max_count = std::max(max_count, (int)v.size());
    }
    if (max_count > kQMaxTable) {
        *too_many = max_count;
        return TAHOE_OK;
    }
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

    // ---- node codes ----
    auto encode = [&](const InnerNode &n, bool real) -> uint32_t {
        if (!real) return 0u;  // padding below a leaf: both children carry the same value
        const uint32_t fid = n.meta & 0x7fffffffu, dl = n.meta >> 31;
        uint32_t code;
        if (std::isnan(n.thr))
            code = 0xFFFFu;  // x >= NaN is never true; a missing x still follows def_left
        else {
            const auto &v = tab[fid];
            code = (uint32_t)(std::lower_bound(v.begin(), v.end(), n.thr) - v.begin()) + 1u;
        }
        return q->narrow ? (code << 16) | (fid << 8) | dl : code | (fid << 16) | (dl << 31);
    };
    const size_t Tg = hi - lo;
    g.tree_lo = (int)lo;
    g.num_trees = (int)Tg;
    g.max_table = max_size;
    const size_t top_n = (size_t)1 << q->top_levels;  // entries per tree (entry 0 unused)
    const size_t top_stride = (size_t)q->top_stride;
    const size_t n_blocks = (size_t)1 << (f->depth - 2);
    const size_t first = n_blocks - 1;
    std::vector<uint32_t> h_top(Tg * top_stride, 0u);
    std::vector<uint4> h_blocks(Tg * n_blocks * 2);
    std::vector<uint32_t> h_qinner(q->have_mid ? Tg * n_inner : 0);
    for (size_t t = 0; t < Tg; ++t) {
        const InnerNode *in = &h_inner[(lo + t) * n_inner];
        const unsigned char *re = &h_real[(lo + t) * n_inner];
        for (size_t i = 0; i + 1 < top_n; ++i) h_top[t * top_stride + i + 1] = encode(in[i], re[i] != 0);
        if (q->have_mid)
            for (size_t i = 0; i < n_inner; ++i) h_qinner[t * n_inner + i] = encode(in[i], re[i] != 0);
        for (size_t b = 0; b < n_blocks; ++b) {
            const size_t r = first + b, l = 2 * r + 1, rr = 2 * r + 2;
            uint4 a, v;
            a.x = encode(in[r], re[r] != 0);
            a.y = encode(in[l], re[l] != 0);
            a.z = encode(in[rr], re[rr] != 0);
            a.w = 0u;
            const float *lv = &h_leaf[(lo + t) * n_leaf + 4 * b];
            memcpy(&v.x, &lv[0], 4);
            memcpy(&v.y, &lv[1], 4);
            memcpy(&v.z, &lv[2], 4);
            memcpy(&v.w, &lv[3], 4);
            h_blocks[(t * n_blocks + b) * 2 + 0] = a;
            h_blocks[(t * n_blocks + b) * 2 + 1] = v;
        }
    }
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
            for (int qd : {4, 2})
                if (cols % (4 * qd) == 0 && (long long)4 * qd * max_size * 4 <= f->lds_limit - 256) {
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
    if ((e = q_upload(&g.top, h_top.data(), h_top.size(), &f->device_bytes)) != hipSuccess) return bad("top");
    if ((e = q_upload(&g.blocks, h_blocks.data(), h_blocks.size(), &f->device_bytes)) != hipSuccess) return bad("blocks");
    if (q->have_mid && (e = q_upload(&g.qinner, h_qinner.data(), h_qinner.size(), &f->device_bytes)) != hipSuccess)
        return bad("qinner");
    return TAHOE_OK;
}

static void free_group(tahoe_qgroup &g)
{
    for (void *p : {(void *)g.tables, (void *)g.offsets, (void *)g.top, (void *)g.blocks, (void *)g.qinner, (void *)g.bsorted,
                    (void *)g.boffsets, (void *)g.bstarts, (void *)g.bparams})
        if (p) (void)hipFree(p);
    g = tahoe_qgroup();
}

tahoe_status qring_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                         const std::vector<float> &h_leaf)
{
    const int cols = f->p.num_cols;
    const size_t T = (size_t)f->p.num_trees;
    if (cols < 1 || cols > 32767 || T == 0) return TAHOE_OK;  // strategy simply unavailable
    tahoe_qstate *q = new (std::nothrow) tahoe_qstate();
    if (!q) return fail(TAHOE_ERR_NO_MEMORY, "qring_build");
    f->q = q;
    q->top_levels = f->top_levels;
    q->have_mid = f->depth - 2 > q->top_levels;
    qwide_pick(f, &q->wide_rt, &q->wide_lw);
    if (q->wide_rt) q->have_mid = q->have_mid || f->depth - 2 > q->wide_lw;  // smaller tops in LDS
    q->top_stride = (int)std::max<size_t>((size_t)1 << q->top_levels, 4);  // >= 16 bytes per tree
    {
        const char *e = getenv("TAHOE_QRING_NARROW");  // experiments: 0 keeps the general node layout
        q->narrow = cols <= 256 && qring_walkers(f) == 15 && qring_lds_tile(f) && !(e && atoi(e) == 0);
    }
    // ---- cut the forest into tree groups whose features each see <= kQMaxTable distinct thresholds ----
    size_t lo = 0, gsize = T;
    {
        // first guess from the node counts per feature (an upper bound of the distinct thresholds): saves building
        // and throwing away the tables of a forest that obviously needs several groups (K4: 8 s -> 4 s of create)
        std::vector<size_t> per_feature((size_t)cols, 0);
        for (size_t i = 0; i < T * f->n_inner; ++i)
            if (h_real[i] && !std::isnan(h_inner[i].thr)) ++per_feature[h_inner[i].meta & 0x7fffffffu];
        const size_t most = *std::max_element(per_feature.begin(), per_feature.end());
        if (most > (size_t)kQMaxTable + kQMaxTable / 2)
            gsize = std::max<size_t>(1, (size_t)((double)T * kQMaxTable / (double)most));
    }
    while (lo < T) {
        size_t hi = std::min(T, lo + gsize);
        tahoe_qgroup g;
        for (;;) {
            int too_many = 0;
            const tahoe_status s = build_group(f, h_inner, h_real, h_leaf, lo, hi, g, &too_many);
            if (s != TAHOE_OK) {
                free_group(g);
                return s;  // create() destroys the handle, which frees the finished groups
            }
            if (too_many == 0) break;
            if (hi - lo == 1) {  // a single tree exceeds the limit: the strategy is unavailable
                qring_destroy(f);
                return TAHOE_OK;
            }
            gsize = std::max<size_t>(1, (size_t)((double)(hi - lo) * kQMaxTable / too_many * 0.9));
            hi = lo + gsize;
        }
        q->groups.push_back(g);
        lo = hi;
    }
    // kernels that need more than 64 KiB of dynamic LDS
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "qring_build: %s failed: %s", what, hipGetErrorString(e)); };
    if (qring_lds_for(f, 15) <= f->lds_limit && (e = q_allow<15>(f->lds_limit)) != hipSuccess) return bad("attr15");
    if (qring_lds_for(f, 12) <= f->lds_limit && (e = q_allow<12>(f->lds_limit)) != hipSuccess) return bad("attr12");
    if (qring_lds_for(f, 8) <= f->lds_limit && (e = q_allow<8>(f->lds_limit)) != hipSuccess) return bad("attr8");
    if (qring_lds_for(f, 4) <= f->lds_limit && (e = q_allow<4>(f->lds_limit)) != hipSuccess) return bad("attr4");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<kGxWalkers, false, false>), f->lds_limit)) != hipSuccess)
        return bad("attr(gx)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<kGxWalkers, true, false>), f->lds_limit)) != hipSuccess)
        return bad("attr(gx)");
    for (const void *k : {(const void *)&qwide_kernel<64, false>, (const void *)&qwide_kernel<64, true>, (const void *)&qwide_kernel<32, false>,
                          (const void *)&qwide_kernel<32, true>, (const void *)&qwide_kernel<16, false>, (const void *)&qwide_kernel<16, true>})
        if ((e = allow_max_lds(k, f->lds_limit)) != hipSuccess) return bad("attr(qwide)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&quantize_pair_kernel), f->lds_limit)) != hipSuccess)
        return bad("attr(quantize_pair)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&quantize_multi_kernel<4>), f->lds_limit)) != hipSuccess)
        return bad("attr(quantize_multi)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&quantize_multi_kernel<2>), f->lds_limit)) != hipSuccess)
        return bad("attr(quantize_multi)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&quantize_bucket_pair_kernel), f->lds_limit)) != hipSuccess)
        return bad("attr(quantize_bucket_pair)");
    if ((e = allow_max_lds(reinterpret_cast<const void *>(&quantize_kernel<1>), f->lds_limit)) != hipSuccess)
        return bad("attr(quantize)");
    return TAHOE_OK;
}

void qring_destroy(tahoe_forest *f)
{
    tahoe_qstate *q = f->q;
    if (!q) return;
    for (tahoe_qgroup &g : q->groups) free_group(g);
    if (q->xq) (void)hipFree(q->xq);
    if (q->chunk_flags) (void)hipFree(q->chunk_flags);
    delete q;
    f->q = nullptr;
}

int qring_groups(const tahoe_forest *f) { return f->q ? (int)f->q->groups.size() : 0; }

// The quantised copy of the batch lives in a grow-only workspace owned by the handle.
tahoe_status qring_reserve(tahoe_forest *f, size_t rows)
{
    tahoe_qstate *q = f->q;
    if (!q || qring_walkers(f) == 0) return TAHOE_OK;
    const size_t tiles = (rows + kQRows - 1) / kQRows;
    if (tiles * kQRows <= q->xq_rows) return TAHOE_OK;
    if (q->xq) {
        TAHOE_HIP_TRY(hipDeviceSynchronize());  // a previous launch may still read the old buffer
        TAHOE_HIP_TRY(hipFree(q->xq));
        f->device_bytes -= q->xq_rows * (size_t)f->p.num_cols * 2;
        q->xq = nullptr;
        q->xq_rows = 0;
    }
    const size_t bytes = tiles * kQRows * (size_t)f->p.num_cols * sizeof(uint16_t);
    TAHOE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&q->xq), bytes));
    q->xq_rows = tiles * kQRows;
    f->device_bytes += bytes;
    if (q->chunk_flags) TAHOE_HIP_TRY(hipFree(q->chunk_flags));
    q->n_chunk_flags = (q->xq_rows + kQuantMinRowsPerBlock - 1) / kQuantMinRowsPerBlock + 1;
    TAHOE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&q->chunk_flags), q->n_chunk_flags * sizeof(uint32_t)));
    return TAHOE_OK;
}

template <int NWALK, bool LDSX = true, bool NARROW = false>
static void q_launch(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                     size_t rows, hipStream_t stream, int cshift)
{
    tahoe_qstate *q = f->q;
    const unsigned grid = (unsigned)((rows + kQRows - 1) / kQRows);
    const int lds = (int)qring_lds_for(f, NWALK, LDSX);
    const uint32_t *leaf_orig = f->leaf_orig + (size_t)g.tree_lo * f->n_leaf;
    if (leaf_out)
        hipLaunchKernelGGL((qring_kernel<NWALK, true, LDSX, NARROW>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, q->xq, g.top,
                           g.blocks, g.qinner, leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth,
                           q->top_levels, q->top_stride, q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, cshift);
    else
        hipLaunchKernelGGL((qring_kernel<NWALK, false, LDSX, NARROW>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, q->xq, g.top,
                           g.blocks, g.qinner, leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth,
                           q->top_levels, q->top_stride, q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, cshift);
}

template <int RT>
static void qwide_launch(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                         size_t rows, hipStream_t stream, int cshift)
{
    tahoe_qstate *q = f->q;
    const unsigned grid = (unsigned)((rows + RT - 1) / RT);
    const int lds = (int)qwide_lds_for(f, RT, q->wide_lw);
    const uint32_t *leaf_orig = f->leaf_orig + (size_t)g.tree_lo * f->n_leaf;
    if (leaf_out)
        hipLaunchKernelGGL((qwide_kernel<RT, true>), dim3(grid), dim3(16 * 64), lds, stream, q->xq, g.top, g.blocks, g.qinner,
                           leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth, q->top_levels, q->top_stride,
                           q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees,
                           (int)qwide_slot_bytes(q->wide_lw, RT), q->wide_lw, cshift);
    else
        hipLaunchKernelGGL((qwide_kernel<RT, false>), dim3(grid), dim3(16 * 64), lds, stream, q->xq, g.top, g.blocks, g.qinner,
                           leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth, q->top_levels, q->top_stride,
                           q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees,
                           (int)qwide_slot_bytes(q->wide_lw, RT), q->wide_lw, cshift);
}

tahoe_status qring_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                          hipStream_t stream, hipEvent_t mid_event)
{
    tahoe_qstate *q = f->q;
    const int nwalk = qring_walkers(f);
    if (!q || nwalk == 0)
        return fail(TAHOE_ERR_UNSUPPORTED,
                    "QRING needs <= %d distinct thresholds per feature within one tree, num_cols <= 32767 and a 128-row "
                    "u16 tile in LDS", kQMaxTable);
    tahoe_status s = qring_reserve(f, rows);  // no-op unless this batch is larger than any before
    if (s != TAHOE_OK) return s;
    const int wide = qwide_rows(f);  // 0: 128-row tiles; else rows per tile of the wide-row form
    const int trs = wide == 64 ? 6 : wide == 32 ? 5 : wide == 16 ? 4 : 7;
    if ((rows + (wide ? wide : kQRows) - 1) / (wide ? wide : kQRows) > 0x7fffffffu)
        return fail(TAHOE_ERR_INVALID_ARG, "too many rows for one launch");
    bool first = true;
    for (const tahoe_qgroup &g : q->groups) {  // stream order: quantise for the group, walk the group, next group
        TAHOE_HIP_TRY(hipMemsetAsync(q->chunk_flags, 0, q->n_chunk_flags * sizeof(uint32_t), stream));
        const bool pair_ok = g.pair_lds_floats > 0 && (reinterpret_cast<uintptr_t>(data) % 8) == 0;  // float2 loads
        const bool multi_ok = g.multi_q > 0 && (reinterpret_cast<uintptr_t>(data) % 16) == 0;  // float4 loads
        // Rows per quantise workgroup (2^cshift): as many as 32768 so that staging the tables is amortised, fewer when
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
        if (multi_ok && g.multi_q == 4)
            hipLaunchKernelGGL(quantize_multi_kernel<4>, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                               (size_t)16 * std::max(g.max_table, 1) * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags,
                               rows, f->p.num_cols, f->p.missing, std::max(g.max_table, 1), trs, cshift);
        else if (multi_ok)
            hipLaunchKernelGGL(quantize_multi_kernel<2>, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                               (size_t)8 * std::max(g.max_table, 1) * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags,
                               rows, f->p.num_cols, f->p.missing, std::max(g.max_table, 1), trs, cshift);
        else if (pair_ok && g.buckets > 0)
            hipLaunchKernelGGL(quantize_bucket_pair_kernel, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                               (size_t)g.bucket_lds_bytes, stream, data, g.bsorted, g.boffsets, g.bstarts, g.bparams, q->xq,
                               q->chunk_flags, rows, f->p.num_cols, f->p.missing, g.buckets, trs, cshift);
        else if (pair_ok)
            hipLaunchKernelGGL(quantize_pair_kernel, dim3((unsigned)qgrid), dim3(kQuantPairThreads),
                               (size_t)g.pair_lds_floats * 4, stream, data, g.tables, g.offsets, q->xq, q->chunk_flags, rows,
                               f->p.num_cols, f->p.missing, g.pair_lds_floats, trs, cshift);
        else
            hipLaunchKernelGGL(quantize_kernel<1>, dim3((unsigned)qgrid), dim3(kQuantThreads), (size_t)std::max(g.max_table, 1) * 4,
                               stream, data, g.tables, g.offsets, q->xq, q->chunk_flags, rows, f->p.num_cols, f->p.missing,
                               std::max(g.max_table, 1), trs, cshift);
        TAHOE_HIP_TRY(hipGetLastError());
        if (first && mid_event) TAHOE_HIP_TRY(hipEventRecord(mid_event, stream));  // splits pre-pass / walk for 1 group
        const float *sums_in = first ? nullptr : sums;  // later groups continue the running float32 sums
        if (wide == 64)
            qwide_launch<64>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
        else if (wide == 32)
            qwide_launch<32>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
        else if (wide == 16)
            qwide_launch<16>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
        else if (!qring_lds_tile(f))
            q_launch<kGxWalkers, false>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
        else
        switch (nwalk) {
            case 15:
                if (q->narrow)
                    q_launch<15, true, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
                else
                    q_launch<15>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
                break;
            case 12: q_launch<12>(f, g, sums, sums_in, leaf_out, rows, stream, cshift); break;
            case 8: q_launch<8>(f, g, sums, sums_in, leaf_out, rows, stream, cshift); break;
            default: q_launch<4>(f, g, sums, sums_in, leaf_out, rows, stream, cshift); break;
        }
        TAHOE_HIP_TRY(hipGetLastError());
        first = false;
    }
    return TAHOE_OK;
}

}  // namespace tahoe
