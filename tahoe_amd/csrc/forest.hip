// Forest handle, device layout and traversal kernels for gfx950 (MI355X).
//
// What this replaces in the reference (file:line into sampathrg/Tahoe):
//   dense_forest / dense_adaptive_forest  (init, infer, predict)     Struct.h:802-861, :1710-2210
//   the walkers infer_one_tree_* and kernels infer_k_* / infer_adaptive_reorg_*   Struct.h:359-1704
//   cub::BlockReduce / DeviceSegmentedReduce call sites (table 2b of SURVEY.md)  -> ordered
//     per-row accumulation through LDS (no tree-parallel float reduction, see "sum order")
//   transform_k                                                      Struct.h:196-209
//
// Device layout, built once at create ("perfect-tree" normalisation):
//   Every tree is normalised to a perfect binary tree of depth De = max(depth, 2): a leaf found
//   above the bottom level is pushed down -- all bottom-level descendants inherit its value and
//   remember its original heap index.  A walk is then exactly De compare-and-step iterations with
//   no leaf test, so the 64 lanes of a wave never diverge.  Three views of the same tree t:
//     (1) heap records          inner[t][2^De - 1] {float thr; uint32 meta}, meta = fid | exchange<<30 | def_left<<31
//                               leaf_val[t][2^De] float, leaf_orig[t][2^De] uint32 (original heap index)
//     (2) top, SoA (<=10 levels) top[t]: thr[2^L] float | meta[2^L] uint16 (fid | def_left<<15), indexed by
//                               1-based heap position: staged whole into LDS by the TILEBLOCK kernel
//     (3) bottom blocks          blk[t][2^(De-2)] 32 B: {thr0, thr1, thr2, 3 x (fid:9 | def_left:1)} {leaf0..3}
//                               = the subtree of the last two inner levels and its four leaves, fetched
//                               with two 16-byte gathers that share one cache line
//   Views (2) and (3) exist when num_cols <= 512 (9-bit feature ids).
//
// Sum order: the CPU predictor adds leaf values in tree order 0..T-1 in float32
// (BaseTahoeTest.h:462-466).  Every kernel here adds them in exactly that order per row, so the raw
// sums are bit-identical to the CPU's, not merely within tolerance.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <vector>

#include "forest_internal.h"

namespace tahoe {

// ------------------------------------------------------------------------------------------------
// (go_right, the branch rule, lives in forest_internal.h)
__device__ __forceinline__ uint32_t step(uint32_t idx, float thr, uint32_t meta, float x, float missing)
{
    return 2u * idx + 1u + go_right_meta(x, thr, meta, missing);
}

// ------------------------------------------------------------------------------------------------
// DIRECT: lane = row; nodes and features come straight from global memory.  Works for any shape.
// Analogue of infer_adaptive_reorg_* (Struct.h:1196-1240).
template <bool WRITE_LEAF>
__global__ void __launch_bounds__(kBlock) direct_kernel(const InnerNode *__restrict__ inner,
                                                        const float *__restrict__ leaf_val,
                                                        const uint32_t *__restrict__ leaf_orig,
                                                        const float *__restrict__ data, float *sums,
                                                        uint32_t *__restrict__ leaf_out, const float *sums_in, size_t rows, int cols,
                                                        int num_trees, int depth, float missing)
{
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= rows) return;
    const float *x = data + row * (size_t)cols;
    const size_t n_inner = ((size_t)1 << depth) - 1;
    const size_t n_leaf = (size_t)1 << depth;
    float sum = sums_in ? sums_in[row] : 0.0f;  // continues a running sum (tree shards chained in order)
    for (int t = 0; t < num_trees; ++t) {
        const InnerNode *tree = inner + (size_t)t * n_inner;
        uint32_t idx = 0;
        for (int l = 0; l < depth; ++l) {
            const InnerNode n = tree[idx];
            idx = step(idx, n.thr, n.meta, x[n.meta & kMetaFidMask], missing);
        }
        const size_t b = (size_t)t * n_leaf + (idx - (uint32_t)n_inner);
        sum += leaf_val[b];
        if (WRITE_LEAF) leaf_out[row * (size_t)num_trees + t] = leaf_orig[b];
    }
    if (sums) sums[row] = sum;
}

// ------------------------------------------------------------------------------------------------
// ROWTILE: one workgroup owns 64 rows, kept feature-major in LDS (tile[fid][row]) so that lane = row
// reads hit 32 distinct banks for any per-lane fid.  The four waves split the trees round-robin
// (wave w walks trees w, w+4, ...); each wave stages the top `lds_levels` levels of its current
// tree in a private LDS slot (prefetched into registers during the previous walk) and reads deeper
// levels from global memory.  Leaf values are exchanged through LDS once per round of four trees and
// added by the row's owner lane in tree order.  Any num_cols whose 64-row tile fits LDS.
//
// Dynamic LDS: [cols][64] float | kWaves slots of slot_nodes InnerNode | [2][kWaves][64] float.
template <bool WRITE_LEAF>
__global__ void __launch_bounds__(kBlock) rowtile_kernel(const InnerNode *__restrict__ inner,
                                                         const float *__restrict__ leaf_val,
                                                         const uint32_t *__restrict__ leaf_orig,
                                                         const float *__restrict__ data, float *sums,
                                                         uint32_t *__restrict__ leaf_out, const float *sums_in, size_t rows, int cols,
                                                         int num_trees, int depth, int lds_levels, float missing,
                                                         int vec4_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot_nodes = (1 << lds_levels) - 1;            // nodes of the staged top
    const int slot_stride = (slot_nodes + 1) & ~1;           // keep 16-byte alignment of each slot
    float *tile = reinterpret_cast<float *>(smem);
    InnerNode *slots = reinterpret_cast<InnerNode *>(smem + (size_t)cols * kTileRows * sizeof(float));
    float *vals = reinterpret_cast<float *>(slots + (size_t)kWaves * slot_stride);
    InnerNode *slot = slots + (size_t)wave * slot_stride;

    const size_t row0 = (size_t)blockIdx.x * kTileRows;
    const size_t row = row0 + lane;
    const bool row_ok = row < rows;

    // ---- stage the row tile, transposed to feature-major ----
    {
        const float *src = data + (row_ok ? row : row0) * (size_t)cols;
        if (vec4_ok) {
            const float4 *src4 = reinterpret_cast<const float4 *>(src);
            for (int f4 = wave; f4 < cols / 4; f4 += kWaves) {
                float4 v = row_ok ? src4[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
                tile[(4 * f4 + 0) * kTileRows + lane] = v.x;
                tile[(4 * f4 + 1) * kTileRows + lane] = v.y;
                tile[(4 * f4 + 2) * kTileRows + lane] = v.z;
                tile[(4 * f4 + 3) * kTileRows + lane] = v.w;
            }
        } else {
            for (int f = wave; f < cols; f += kWaves) tile[f * kTileRows + lane] = row_ok ? src[f] : 0.0f;
        }
    }

    const size_t n_inner = ((size_t)1 << depth) - 1;
    const size_t n_leaf = (size_t)1 << depth;
    // Each lane carries up to ceil(slot_nodes / 64) records of the next tree's top in registers.
    constexpr int kMaxPref = ((1 << kMaxLdsLevels) + 63) / 64;  // 4
    uint2 pref[kMaxPref];
    auto prefetch_top = [&](int t) {
        const uint2 *g = reinterpret_cast<const uint2 *>(inner + (size_t)t * n_inner);
#pragma unroll
        for (int k = 0; k < kMaxPref; ++k) {
            const int i = k * 64 + lane;
            pref[k] = (i < slot_nodes) ? g[i] : make_uint2(0u, 0u);
        }
    };
    auto commit_top = [&]() {
        uint2 *s = reinterpret_cast<uint2 *>(slot);
#pragma unroll
        for (int k = 0; k < kMaxPref; ++k) {
            const int i = k * 64 + lane;
            if (i < slot_nodes) s[i] = pref[k];
        }
    };
    if (wave < num_trees) {
        prefetch_top(wave);
        commit_top();
    }
    __syncthreads();  // tile and every wave's first slot are in LDS

    float sum = 0.0f;  // meaningful in lanes 0..15: row 16*wave + lane of the tile
    if (sums_in && lane < 16 && row0 + 16 * wave + lane < rows) sum = sums_in[row0 + 16 * wave + lane];
    const int rounds = (num_trees + kWaves - 1) / kWaves;
    for (int r = 0; r < rounds; ++r) {
        const int t = r * kWaves + wave;
        float v = 0.0f;
        if (t < num_trees) {
            const bool more = t + kWaves < num_trees;
            if (more) prefetch_top(t + kWaves);
            uint32_t idx = 0;
            for (int l = 0; l < lds_levels; ++l) {
                const InnerNode n = slot[idx];
                const float x = tile[(n.meta & kMetaFidMask) * kTileRows + lane];
                idx = step(idx, n.thr, n.meta, x, missing);
            }
            const InnerNode *tree = inner + (size_t)t * n_inner;
            for (int l = lds_levels; l < depth; ++l) {
                const InnerNode n = tree[idx];
                const float x = tile[(n.meta & kMetaFidMask) * kTileRows + lane];
                idx = step(idx, n.thr, n.meta, x, missing);
            }
            const size_t b = (size_t)t * n_leaf + (idx - (uint32_t)n_inner);
            v = leaf_val[b];
            if (WRITE_LEAF) {
                if (row_ok) leaf_out[row * (size_t)num_trees + t] = leaf_orig[b];
            }
            if (more) commit_top();  // this wave's reads of the slot are done (in-order LDS)
        }
        float *vb = vals + (size_t)(r & 1) * kWaves * kTileRows;
        vb[wave * kTileRows + lane] = v;
        __syncthreads();
        if (lane < 16) {
            const int rr = 16 * wave + lane;
            const int nt = min(kWaves, num_trees - r * kWaves);
            for (int j = 0; j < nt; ++j) sum += vb[j * kTileRows + rr];  // tree order
        }
    }
    if (sums && lane < 16) {
        const size_t orow = row0 + 16 * wave + lane;
        if (orow < rows) sums[orow] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// TILEBLOCK: the K3-shaped strategy (num_cols <= 512).
//   * ROWS (128 or 64) rows of the batch live feature-major in LDS for the whole kernel.
//   * kSlots = 4 trees are in flight; slot s is walked by ROWS/64 waves (one per 64-row group).
//     A slot holds the tree's top `top_levels` levels as SoA (thr float | meta uint16): one
//     ds_read_b32 + one ds_read_u16 per visit, feature read conflict-free by layout.
//   * Levels below the LDS top and above the last two come from the heap records in global memory
//     (only when De > 12); the last two levels and the leaves come from one 32-byte block.
//   * The next round's tops are prefetched into registers while the current trees are walked and
//     committed to LDS between the two barriers of a round; leaf values cross LDS once per round
//     and are added by the row's owner lane in tree order (bit-exact float32 sums).
// Dynamic LDS: [cols][ROWS] float | kSlots x top_stride bytes | [2][kSlots][ROWS] float.
template <int ROWS, bool WRITE_LEAF>
__global__ void __launch_bounds__(kSlots *ROWS)
    tileblock_kernel(const unsigned char *__restrict__ top, const uint4 *__restrict__ blocks,
                     const InnerNode *__restrict__ inner, const uint32_t *__restrict__ leaf_orig,
                     const float *__restrict__ data, float *sums, uint32_t *__restrict__ leaf_out,
                     const float *sums_in, size_t rows, int cols, int num_trees, int depth, int top_levels, int top_stride, float missing,
                     int vec4_ok)
{
    constexpr int RG = ROWS / 64;            // waves per slot (row groups)
    constexpr int NT = kSlots * ROWS;        // threads
    constexpr int NW = NT / 64;              // waves
    constexpr int OWN = ROWS / NW;           // rows each wave accumulates (16)
    constexpr int kChunksMax = (4096 + 2048) / 16;
    constexpr int PF = (kChunksMax + ROWS - 1) / ROWS;  // 16-byte chunks a lane prefetches (3 or 6)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: wave-uniform branches below
    const int slot_id = wave / RG;
    const int rgrp = wave % RG;
    const int rloc = rgrp * 64 + lane;       // row within the tile
    const int sl = rgrp * 64 + lane;         // lane index within the slot's group of RG waves

    float *tile = reinterpret_cast<float *>(smem);
    unsigned char *slots = smem + (size_t)cols * ROWS * sizeof(float);
    float *vals = reinterpret_cast<float *>(slots + (size_t)kSlots * top_stride);
    unsigned char *slot = slots + (size_t)slot_id * top_stride;
    // slot = thr[2^L] float | meta[2^L] uint16, both indexed by 1-based heap position (entry 0 unused)
    const float *s_thr = reinterpret_cast<const float *>(slot);
    const uint16_t *s_meta = reinterpret_cast<const uint16_t *>(slot + (((4 << top_levels) + 15) & ~15));
    const int n_chunks = top_stride >> 4;

    const size_t row0 = (size_t)blockIdx.x * ROWS;
    const size_t row = row0 + rloc;
    const bool row_ok = row < rows;

    // ---- stage the row tile, transposed to feature-major (conflict-free: bank = row % 32) ----
    {
        const int trow = tid % ROWS, tpart = tid / ROWS;  // kSlots parts stride over the features
        const size_t grow = row0 + trow;
        const bool ok = grow < rows;
        const float *src = data + (ok ? grow : row0) * (size_t)cols;
        if (vec4_ok) {
            const float4 *src4 = reinterpret_cast<const float4 *>(src);
            for (int f4 = tpart; f4 < cols / 4; f4 += kSlots) {
                const float4 v = ok ? src4[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
                tile[(4 * f4 + 0) * ROWS + trow] = v.x;
                tile[(4 * f4 + 1) * ROWS + trow] = v.y;
                tile[(4 * f4 + 2) * ROWS + trow] = v.z;
                tile[(4 * f4 + 3) * ROWS + trow] = v.w;
            }
        } else {
            for (int f = tpart; f < cols; f += kSlots) tile[f * ROWS + trow] = ok ? src[f] : 0.0f;
        }
    }

    // Next round's top: PF 16-byte chunks per lane, held in registers while the current tree is walked.
    // Named registers (not an array): an indexed array ends up in scratch memory.
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {}, pf4 = {}, pf5 = {};
    static_assert(PF == 3 || PF == 6, "prefetch registers are written out for 3 or 6 chunks per lane");
    auto prefetch_top = [&](int t) {
        const uint4 *g = reinterpret_cast<const uint4 *>(top + (size_t)t * top_stride);
        const int last = n_chunks - 1;  // clamped: branch-free and in bounds
        pf0 = g[min(0 * ROWS + sl, last)];
        pf1 = g[min(1 * ROWS + sl, last)];
        pf2 = g[min(2 * ROWS + sl, last)];
        if constexpr (PF == 6) {
            pf3 = g[min(3 * ROWS + sl, last)];
            pf4 = g[min(4 * ROWS + sl, last)];
            pf5 = g[min(5 * ROWS + sl, last)];
        }
    };
    auto commit_top = [&]() {
        uint4 *s = reinterpret_cast<uint4 *>(slot);
        if (0 * ROWS + sl < n_chunks) s[0 * ROWS + sl] = pf0;
        if (1 * ROWS + sl < n_chunks) s[1 * ROWS + sl] = pf1;
        if (2 * ROWS + sl < n_chunks) s[2 * ROWS + sl] = pf2;
        if constexpr (PF == 6) {
            if (3 * ROWS + sl < n_chunks) s[3 * ROWS + sl] = pf3;
            if (4 * ROWS + sl < n_chunks) s[4 * ROWS + sl] = pf4;
            if (5 * ROWS + sl < n_chunks) s[5 * ROWS + sl] = pf5;
        }
    };
    if (slot_id < num_trees) {
        prefetch_top(slot_id);
        commit_top();
    }
    __syncthreads();

    const size_t n_inner = ((size_t)1 << depth) - 1;
    const uint32_t n_blocks = 1u << (depth - 2);
    const uint32_t first_block_node = n_blocks - 1;  // heap index of the first node of level depth-2
    float sum = 0.0f;                                // lanes 0..OWN-1: row OWN*wave + lane of the tile
    if (sums_in && lane < OWN && row0 + OWN * wave + lane < rows) sum = sums_in[row0 + OWN * wave + lane];
    const int rounds = (num_trees + kSlots - 1) / kSlots;
    for (int r = 0; r < rounds; ++r) {
        const int t = r * kSlots + slot_id;
        const bool more = t + kSlots < num_trees;
        float v = 0.0f;
        if (t < num_trees) {
            if (more) prefetch_top(t + kSlots);
            // Top levels from LDS, 1-based heap positions: the children of position i are the aligned
            // pair (2i, 2i+1), read together with the current node's feature -> one LDS round trip
            // per level.
            uint32_t i = 1;
            if (top_levels > 0) {
                float thr = s_thr[1];
                uint32_t m = s_meta[1];
                for (int l = 0; l < top_levels - 1; ++l) {
                    const float x = tile[(m & 0x7fffu) * ROWS + rloc];
                    const float2 t2 = *reinterpret_cast<const float2 *>(&s_thr[2 * i]);
                    const uint32_t m2 = *reinterpret_cast<const uint32_t *>(&s_meta[2 * i]);
                    const uint32_t c = go_right(x, thr, (m >> 15) != 0, missing);
                    i = 2u * i + c;
                    thr = c ? t2.y : t2.x;
                    m = c ? (m2 >> 16) : (m2 & 0xffffu);
                }
                const float x = tile[(m & 0x7fffu) * ROWS + rloc];
                i = 2u * i + go_right(x, thr, (m >> 15) != 0, missing);
            }
            uint32_t idx = i - 1;  // 0-based heap index on level top_levels
            if (top_levels < depth - 2) {  // deep trees only (De > 12): heap records from global memory
                const InnerNode *tree = inner + (size_t)t * n_inner;
                for (int l = top_levels; l < depth - 2; ++l) {
                    const InnerNode n = tree[idx];
                    const float x = tile[(n.meta & kMetaFidMask) * ROWS + rloc];
                    idx = step(idx, n.thr, n.meta, x, missing);
                }
            }
            const uint32_t b = idx - first_block_node;
            const uint4 *bp = blocks + ((size_t)t * n_blocks + b) * 2;
            const uint4 na = bp[0];  // thr0, thr1, thr2, packed metas
            const uint4 nb = bp[1];  // four leaf values
            // Both 16-byte loads are issued here, whole: without the pins the compiler sinks single
            // dwords of them into the branches of go_right and pays a second global round trip.
            asm volatile("" ::"v"(na.x), "v"(na.y), "v"(na.z), "v"(na.w));
            asm volatile("" ::"v"(nb.x), "v"(nb.y), "v"(nb.z), "v"(nb.w));
            const uint32_t m = na.w;
            const float x0 = tile[(m & 0x1ffu) * ROWS + rloc];
            const float x1 = tile[((m >> 10) & 0x1ffu) * ROWS + rloc];
            const float x2 = tile[((m >> 20) & 0x1ffu) * ROWS + rloc];
            const uint32_t c0 = go_right(x0, __uint_as_float(na.x), ((m >> 9) & 1u) != 0, missing);
            const uint32_t c1l = go_right(x1, __uint_as_float(na.y), ((m >> 19) & 1u) != 0, missing);
            const uint32_t c1r = go_right(x2, __uint_as_float(na.z), ((m >> 29) & 1u) != 0, missing);
            const uint32_t c1 = c0 ? c1r : c1l;
            const uint32_t lo = c1 ? nb.y : nb.x, hi = c1 ? nb.w : nb.z;
            v = __uint_as_float(c0 ? hi : lo);
            if (WRITE_LEAF) {
                if (row_ok)
                    leaf_out[row * (size_t)num_trees + t] =
                        leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)b * 4 + 2 * c0 + c1];
            }
        }
        float *vb = vals + (size_t)(r & 1) * kSlots * ROWS;
        vb[slot_id * ROWS + rloc] = v;
        __syncthreads();  // (A) every walk of this round has finished reading its slot
        if (t < num_trees && more) commit_top();
        if (lane < OWN) {
            const int rr = OWN * wave + lane;
            const int nt = min(kSlots, num_trees - r * kSlots);
            for (int j = 0; j < nt; ++j) sum += vb[j * ROWS + rr];  // tree order
        }
        __syncthreads();  // (B) the next round's tops are in LDS
    }
    if (sums && lane < OWN) {
        const size_t orow = row0 + OWN * wave + lane;
        if (orow < rows) sums[orow] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// TILERING: TILEBLOCK's data path with the waves decoupled.  No workgroup barrier inside the tree loop.
//   * NWALK walker waves, each with a private LDS slot for the top of its current tree; walker w walks
//     trees w, w+NWALK, ... for all ROWS rows of the tile (K = ROWS/64 interleaved chains per lane), and
//     prefetches the next top into registers meanwhile.  Waves drift freely, so one wave's global
//     gather latency is covered by the others' LDS walks.
//   * One consumer wave adds the leaf values in tree order (bit-exact float32 sums).  Walkers hand
//     them over through a ring in LDS: vals[RING][ROWS] + ready[RING] (holds tree+1) + one `consumed`
//     counter; single-CU LDS traffic is processed in order, so a flag written after its data is seen
//     after its data.  Every spin is bounded: on timeout the kernel sets *error_flag and drains.
// Dynamic LDS: [cols][ROWS] float | NWALK x top_stride | [RING][ROWS] float | ready[RING] | consumed.
constexpr int kRingSpinLimit = 1 << 22;

template <int ROWS>
struct RingGeom {
    static constexpr int RING = (ROWS == 64) ? 32 : 8;   // ring entries (trees)
    static constexpr int BATCH = (ROWS == 64) ? 8 : 4;   // trees the consumer takes per poll
};

template <int ROWS, int NWALK, bool WRITE_LEAF>
__global__ void __launch_bounds__((NWALK + 1) * 64)
    tilering_kernel(const unsigned char *__restrict__ top, const uint4 *__restrict__ blocks,
                    const InnerNode *__restrict__ inner, const uint32_t *__restrict__ leaf_orig,
                    const float *__restrict__ data, float *sums, uint32_t *__restrict__ leaf_out,
                    const float *sums_in, size_t rows, int cols, int num_trees, int depth, int top_levels, int top_stride, float missing,
                    int vec4_ok, int *__restrict__ error_flag)
{
    constexpr int K = ROWS / 64;
    constexpr int NT = (NWALK + 1) * 64;
    constexpr int RING = RingGeom<ROWS>::RING;
    constexpr int BATCH = RingGeom<ROWS>::BATCH;
    static_assert(BATCH <= RING && BATCH <= 64, "consumer batch must fit the ring");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    float *tile = reinterpret_cast<float *>(smem);
    unsigned char *slots = smem + (size_t)cols * ROWS * sizeof(float);
    volatile float *ring_vals = reinterpret_cast<volatile float *>(slots + (size_t)NWALK * top_stride);
    volatile uint32_t *ring_ready = reinterpret_cast<volatile uint32_t *>(
        slots + (size_t)NWALK * top_stride + (size_t)RING * ROWS * sizeof(float));
    volatile uint32_t *consumed = ring_ready + RING;

    const size_t row0 = (size_t)blockIdx.x * ROWS;

    // ---- stage the row tile, transposed to feature-major; reset the ring ----
    if (vec4_ok) {
        const int n4 = cols / 4;
        for (int e = tid; e < ROWS * n4; e += NT) {
            const int trow = e % ROWS, f4 = e / ROWS;
            const size_t grow = row0 + trow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < rows) v = reinterpret_cast<const float4 *>(data + grow * (size_t)cols)[f4];
            tile[(4 * f4 + 0) * ROWS + trow] = v.x;
            tile[(4 * f4 + 1) * ROWS + trow] = v.y;
            tile[(4 * f4 + 2) * ROWS + trow] = v.z;
            tile[(4 * f4 + 3) * ROWS + trow] = v.w;
        }
    } else {
        for (int e = tid; e < ROWS * cols; e += NT) {
            const int trow = e % ROWS, f = e / ROWS;
            const size_t grow = row0 + trow;
            tile[f * ROWS + trow] = grow < rows ? data[grow * (size_t)cols + f] : 0.0f;
        }
    }
    if (tid < RING) ring_ready[tid] = 0u;
    if (tid == RING) *consumed = 0u;

    if (wave == NWALK) {
        // ================= consumer: ordered accumulation =================
        __syncthreads();
        float sum[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t irow = row0 + k * 64 + lane;
            sum[k] = (sums_in && irow < rows) ? sums_in[irow] : 0.0f;
        }
        bool dead = false;
        for (int t0 = 0; t0 < num_trees && !dead; t0 += BATCH) {
            const int nb = min(BATCH, num_trees - t0);
            int spins = 0;
            for (;;) {
                const bool ok = lane >= nb || ring_ready[(t0 + lane) % RING] == (uint32_t)(t0 + lane + 1);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kRingSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            for (int j = 0; j < nb; ++j) {
                const int e = (t0 + j) % RING;
#pragma unroll
                for (int k = 0; k < K; ++k) sum[k] += ring_vals[e * ROWS + k * 64 + lane];  // tree order
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *consumed = (uint32_t)(t0 + nb);
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
    unsigned char *slot = slots + (size_t)wave * top_stride;
    const float *s_thr = reinterpret_cast<const float *>(slot);
    const uint16_t *s_meta = reinterpret_cast<const uint16_t *>(slot + (((4 << top_levels) + 15) & ~15));
    const int n_chunks = top_stride >> 4;

    // Next tree's top: six 16-byte chunks per lane (6 KiB per slot), named registers (an indexed array
    // would live in scratch memory).
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {}, pf4 = {}, pf5 = {};
    auto prefetch_top = [&](int t) {
        const uint4 *g = reinterpret_cast<const uint4 *>(top + (size_t)t * top_stride);
        const int last = n_chunks - 1;  // clamped: branch-free and in bounds
        pf0 = g[min(0 * 64 + lane, last)];
        pf1 = g[min(1 * 64 + lane, last)];
        pf2 = g[min(2 * 64 + lane, last)];
        pf3 = g[min(3 * 64 + lane, last)];
        pf4 = g[min(4 * 64 + lane, last)];
        pf5 = g[min(5 * 64 + lane, last)];
    };
    auto commit_top = [&]() {
        uint4 *s = reinterpret_cast<uint4 *>(slot);
        if (0 * 64 + lane < n_chunks) s[0 * 64 + lane] = pf0;
        if (1 * 64 + lane < n_chunks) s[1 * 64 + lane] = pf1;
        if (2 * 64 + lane < n_chunks) s[2 * 64 + lane] = pf2;
        if (3 * 64 + lane < n_chunks) s[3 * 64 + lane] = pf3;
        if (4 * 64 + lane < n_chunks) s[4 * 64 + lane] = pf4;
        if (5 * 64 + lane < n_chunks) s[5 * 64 + lane] = pf5;
    };
    if (wave < num_trees) {
        prefetch_top(wave);
        commit_top();
    }
    __syncthreads();  // the tile, the ring state and (own wave) the first top are in LDS

    const size_t n_inner = ((size_t)1 << depth) - 1;
    const uint32_t n_blocks = 1u << (depth - 2);
    const uint32_t first_block_node = n_blocks - 1;
    bool dead = false;
    for (int t = wave; t < num_trees && !dead; t += NWALK) {
        const bool more = t + NWALK < num_trees;
        if (more) prefetch_top(t + NWALK);
        // K interleaved chains: chain k walks row k*64 + lane.  1-based heap positions; the children
        // of position i are the aligned pair (2i, 2i+1), read together with the node's feature.
        uint32_t i[K];
#pragma unroll
        for (int k = 0; k < K; ++k) i[k] = 1;
        if (top_levels > 0) {
            float thr[K];
            uint32_t m[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                thr[k] = s_thr[1];
                m[k] = s_meta[1];
            }
            for (int l = 0; l < top_levels - 1; ++l) {
                float x[K];
                float2 t2[K];
                uint32_t m2[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    x[k] = tile[(m[k] & 0x7fffu) * ROWS + k * 64 + lane];
                    t2[k] = *reinterpret_cast<const float2 *>(&s_thr[2 * i[k]]);
                    m2[k] = *reinterpret_cast<const uint32_t *>(&s_meta[2 * i[k]]);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t c = go_right(x[k], thr[k], (m[k] >> 15) != 0, missing);
                    i[k] = 2u * i[k] + c;
                    thr[k] = c ? t2[k].y : t2[k].x;
                    m[k] = c ? (m2[k] >> 16) : (m2[k] & 0xffffu);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = tile[(m[k] & 0x7fffu) * ROWS + k * 64 + lane];
                i[k] = 2u * i[k] + go_right(x, thr[k], (m[k] >> 15) != 0, missing);
            }
        }
        float v[K];
        uint32_t bsel[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t idx = i[k] - 1;  // 0-based heap index on level top_levels
            if (top_levels < depth - 2) {  // deep trees only (De > 12): heap records from global memory
                const InnerNode *tree = inner + (size_t)t * n_inner;
                for (int l = top_levels; l < depth - 2; ++l) {
                    const InnerNode n = tree[idx];
                    const float x = tile[(n.meta & kMetaFidMask) * ROWS + k * 64 + lane];
                    idx = step(idx, n.thr, n.meta, x, missing);
                }
            }
            bsel[k] = idx - first_block_node;
        }
        uint4 na[K], nb[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint4 *bp = blocks + ((size_t)t * n_blocks + bsel[k]) * 2;
            na[k] = bp[0];  // thr0, thr1, thr2, packed metas
            nb[k] = bp[1];  // four leaf values
            // whole 16-byte loads, issued here (the compiler otherwise sinks single dwords into branches)
            asm volatile("" ::"v"(na[k].x), "v"(na[k].y), "v"(na[k].z), "v"(na[k].w));
            asm volatile("" ::"v"(nb[k].x), "v"(nb[k].y), "v"(nb[k].z), "v"(nb[k].w));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t m = na[k].w;
            const float x0 = tile[(m & 0x1ffu) * ROWS + k * 64 + lane];
            const float x1 = tile[((m >> 10) & 0x1ffu) * ROWS + k * 64 + lane];
            const float x2 = tile[((m >> 20) & 0x1ffu) * ROWS + k * 64 + lane];
            const uint32_t c0 = go_right(x0, __uint_as_float(na[k].x), ((m >> 9) & 1u) != 0, missing);
            const uint32_t c1l = go_right(x1, __uint_as_float(na[k].y), ((m >> 19) & 1u) != 0, missing);
            const uint32_t c1r = go_right(x2, __uint_as_float(na[k].z), ((m >> 29) & 1u) != 0, missing);
            const uint32_t c1 = c0 ? c1r : c1l;
            const uint32_t lo = c1 ? nb[k].y : nb[k].x, hi = c1 ? nb[k].w : nb[k].z;
            v[k] = __uint_as_float(c0 ? hi : lo);
            if (WRITE_LEAF) {
                const size_t row = row0 + k * 64 + lane;
                if (row < rows)
                    leaf_out[row * (size_t)num_trees + t] =
                        leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bsel[k] * 4 + 2 * c0 + c1];
            }
        }
        // ---- hand the leaf values to the consumer ----
        if (t >= RING) {
            int spins = 0;
            while (*consumed < (uint32_t)(t - RING + 1)) {
                if (++spins > kRingSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const int e = t % RING;
#pragma unroll
        for (int k = 0; k < K; ++k) ring_vals[e * ROWS + k * 64 + lane] = v[k];
        TAHOE_LDS_RELEASE();  // values before the flag: a wave's LDS operations are performed in issue order
        if (lane == 0) ring_ready[e] = (uint32_t)(t + 1);
        if (more) commit_top();  // this wave's reads of its slot are done (in-order LDS)
    }
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

// transform_k (Struct.h:196-209) with the CPU predictor's arithmetic (BaseTahoeTest.h:467-472):
// AVG divides by num_trees (the reference's GPU epilogue multiplies by 1/T instead).
__global__ void transform_kernel(float *preds, size_t n, int output, int num_trees, float threshold,
                                 float global_bias)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = preds[i];
    if ((output & TAHOE_OUT_AVG) != 0) r = r / (float)num_trees;
    r += global_bias;
    if ((output & TAHOE_OUT_SIGMOID) != 0) r = 1.0f / (1.0f + expf(-r));
    if ((output & TAHOE_OUT_THRESHOLD) != 0) r = r > threshold ? 1.0f : 0.0f;
    preds[i] = r;
}

// ------------------------------------------------------------------------------------------------
static int rowtile_lds_bytes(int cols, int lds_levels)
{
    const int slot_nodes = (1 << lds_levels) - 1;
    const int slot_stride = (slot_nodes + 1) & ~1;
    return cols * kTileRows * (int)sizeof(float) + kWaves * slot_stride * (int)sizeof(InnerNode) +
           2 * kWaves * kTileRows * (int)sizeof(float);
}

static bool rowtile_fits(const tahoe_forest *f)
{
    return f->p.num_cols >= 1 && (long long)rowtile_lds_bytes(f->p.num_cols, f->lds_levels) <= f->lds_limit;
}

static long long tileblock_lds_bytes(const tahoe_forest *f, int tile_rows)
{
    return (long long)f->p.num_cols * tile_rows * 4 + (long long)kSlots * top_stride_bytes(f->top_levels) +
           2LL * kSlots * tile_rows * 4;
}

// Rows per TILEBLOCK tile: 128 when it fits LDS, else 64, else 0 (strategy unavailable).
static int tileblock_rows(const tahoe_forest *f)
{
    if (!f->has_blocks || f->p.num_cols < 1) return 0;
    if (const int r = f->knob_tile_rows)  // TAHOE_TILE_ROWS, read at create
        if ((r == 64 || r == 128) && tileblock_lds_bytes(f, r) <= f->lds_limit) return r;
    if (tileblock_lds_bytes(f, 128) <= f->lds_limit) return 128;
    if (tileblock_lds_bytes(f, 64) <= f->lds_limit) return 64;
    return 0;
}

static long long tilering_lds_bytes(const tahoe_forest *f, int tile_rows)
{
    const int nwalk = tile_rows == 64 ? 8 : 4;
    const int ring = tile_rows == 64 ? RingGeom<64>::RING : RingGeom<128>::RING;
    return (long long)f->p.num_cols * tile_rows * 4 + (long long)nwalk * top_stride_bytes(f->top_levels) +
           (long long)ring * tile_rows * 4 + (ring + 1) * 4LL;
}

// Rows per TILERING tile: 64 (8 walkers x 1 chain) or 128 (4 walkers x 2 chains); 0 = unavailable.
static int tilering_rows(const tahoe_forest *f)
{
    if (!f->has_blocks || f->p.num_cols < 1) return 0;
    if (const int r = f->knob_tile_rows)  // TAHOE_TILE_ROWS, read at create
        if ((r == 64 || r == 128) && tilering_lds_bytes(f, r) <= f->lds_limit) return r;
    if (tilering_lds_bytes(f, 64) <= f->lds_limit) return 64;
    return 0;
}

static int resolve_strategy(const tahoe_forest *f, size_t rows)
{
    if (f->sp) {  // sparse handle: QRING = quantised 192-row tile + tree tops in LDS, TILEBLOCK = 64-row float32 tile + tree
                  // tops, ROWTILE = tile only, DIRECT = neither
        if (f->strategy == TAHOE_STRATEGY_DIRECT) return TAHOE_STRATEGY_DIRECT;
        if (f->strategy == TAHOE_STRATEGY_ROWTILE) return TAHOE_STRATEGY_ROWTILE;
        // the quantise pass over rows x cols pays when there is walking to do per feature value (the dense rule, with ten
        // levels standing in for the unknown path length) and the batch gives every CU a 128-row tile or so -- below that
        // the float32 kernel's 64-row tiles fill more of the chip (tools/selector_sparse.py, profiles/r02/selector_sparse.json)
        if (sparse_q_available(f) &&
            (f->strategy == TAHOE_STRATEGY_QRING ||
             (f->strategy == TAHOE_STRATEGY_AUTO && 20LL * f->p.num_trees >= 13LL * f->p.num_cols && rows >= (size_t)64 * (size_t)std::max(f->num_cus, 1))))
            return TAHOE_STRATEGY_QRING;
        if (sparse_top_waves(f) > 0) return TAHOE_STRATEGY_TILEBLOCK;
        return sparse_tile_fits(f) ? TAHOE_STRATEGY_ROWTILE : TAHOE_STRATEGY_DIRECT;
    }
    if (f->strategy != TAHOE_STRATEGY_AUTO) return f->strategy;
    // QRING pays a quantise pass over rows x cols to make every (row, tree, level) step ~3x cheaper.  Fitted on the
    // enumeration of tools/selector_check.py (profiles/r01/selector_vs_enumeration.json): it loses to ROWTILE on
    // very shallow trees (the whole tree sits in ROWTILE's LDS top, no pre-pass) and to the float32 tile kernels
    // when there is little walking per feature value (trees x depth < 6.5 x cols).  On ten shapes it was not fitted
    // on (--holdout) it picks the fastest strategy on 8 and stays within 1.16x on the other two.
    // (round 4: only while there is little of it -- trees x depth <= 1024.  With 384-row tiles on narrow / few-threshold forests QRING
    // overtook ROWTILE on large shallow forests: 2000 trees x depth 3 x 32 features 0.85 against 0.99 ms, 800 x 4 x 100 0.38 against 0.58,
    // while 100 x 4 x 16 and 30 x 3 x 8 stay with ROWTILE, profiles/r04/selector_*.json)
    const bool shallow = f->depth <= 4 && rowtile_fits(f) && (long long)f->p.num_trees * f->depth <= 1024;
    // (wide rows whose quantised form walks three trees per lane: that form is ~1.25 x faster, the float32 form pays off later)
    const long long per_col = qwide_chains(f) == 3 ? 10 : 13;
    const bool little_work = 2LL * f->p.num_trees * f->depth < per_col * f->p.num_cols &&
                             (tilering_rows(f) > 0 || widef_rows(f) > 0 || tileblock_rows(f) > 0 || rowtile_fits(f));
    if (shallow) return TAHOE_STRATEGY_ROWTILE;
    if (qring_walkers(f) > 0 && !little_work) return TAHOE_STRATEGY_QRING;
    if (tilering_rows(f) > 0 || widef_rows(f) > 0) return TAHOE_STRATEGY_TILERING;
    if (tileblock_rows(f) > 0) return TAHOE_STRATEGY_TILEBLOCK;
    return rowtile_fits(f) ? TAHOE_STRATEGY_ROWTILE : TAHOE_STRATEGY_DIRECT;
}

template <int ROWS>
static void launch_tileblock(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *sums_in, const float *data, size_t rows,
                             hipStream_t stream, int vec4_ok)
{
    const unsigned grid = (unsigned)((rows + ROWS - 1) / ROWS);
    const int lds = (int)tileblock_lds_bytes(f, ROWS);
    const int stride = top_stride_bytes(f->top_levels);
    if (leaf_out)
        hipLaunchKernelGGL((tileblock_kernel<ROWS, true>), dim3(grid), dim3(kSlots * ROWS), lds, stream, f->top,
                           f->blocks, f->inner, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                           f->p.num_trees, f->depth, f->top_levels, stride, f->p.missing, vec4_ok);
    else
        hipLaunchKernelGGL((tileblock_kernel<ROWS, false>), dim3(grid), dim3(kSlots * ROWS), lds, stream, f->top,
                           f->blocks, f->inner, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                           f->p.num_trees, f->depth, f->top_levels, stride, f->p.missing, vec4_ok);
}

template <int ROWS, int NWALK>
static void launch_tilering(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *sums_in, const float *data, size_t rows,
                            hipStream_t stream, int vec4_ok)
{
    const unsigned grid = (unsigned)((rows + ROWS - 1) / ROWS);
    const int lds = (int)tilering_lds_bytes(f, ROWS);
    const int stride = top_stride_bytes(f->top_levels);
    if (leaf_out)
        hipLaunchKernelGGL((tilering_kernel<ROWS, NWALK, true>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream,
                           f->top, f->blocks, f->inner, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                           f->p.num_trees, f->depth, f->top_levels, stride, f->p.missing, vec4_ok, f->error_flag);
    else
        hipLaunchKernelGGL((tilering_kernel<ROWS, NWALK, false>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream,
                           f->top, f->blocks, f->inner, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                           f->p.num_trees, f->depth, f->top_levels, stride, f->p.missing, vec4_ok, f->error_flag);
}

// A process that drives several GPUs (one handle per device) calls predict with any device current: the launches
// must be issued with the handle's device current.  Restores the caller's device on scope exit.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want)
    {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// sums_in (optional, may be `sums` itself): running float32 sums of the trees BEFORE this forest -- every kernel then
// continues that sum in tree order instead of starting from 0.0f (tahoe_forest_predict_accumulate).
static tahoe_status launch_traversal(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data,
                                     size_t rows, hipStream_t stream, const float *sums_in = nullptr)
{
    if (rows == 0) return TAHOE_OK;
    DeviceGuard on_device(f->device);
    const int strategy = resolve_strategy(f, rows);
    if ((rows + 63) / 64 > 0x7fffffffu) return fail(TAHOE_ERR_INVALID_ARG, "too many rows for one launch: %zu", rows);
    const bool timed = f->profiling && f->prof_count < f->ev_start.size();
    if (timed) TAHOE_HIP_TRY(hipEventRecord(f->ev_start[f->prof_count], stream));
    bool mid_recorded = false;
    if (timed && !(strategy == TAHOE_STRATEGY_QRING && !f->sp && f->p.num_trees != 0)) {
        // strategies without a separately timed pre-pass: mid = start, so tahoe_forest_prepass_times reads 0 and
        // tahoe_forest_kernel_times the whole launch (a sparse handle's quantise pass counts as part of its walk)
        TAHOE_HIP_TRY(hipEventRecord(f->ev_mid[f->prof_count], stream));
        mid_recorded = true;
    }
    const int vec4_ok = (f->p.num_cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0);
    if (f->p.num_trees == 0) {
        // Nothing to walk: sums are zero (an empty j-loop in predict_on_cpu).
        if (sums && sums_in && sums != sums_in)
            TAHOE_HIP_TRY(hipMemcpyAsync(sums, sums_in, rows * sizeof(float), hipMemcpyDeviceToDevice, stream));
        else if (sums && !sums_in)
            TAHOE_HIP_TRY(hipMemsetAsync(sums, 0, rows * sizeof(float), stream));
    } else if (f->sp) {
        const tahoe_status ss = sparse_launch(f, sums, leaf_out, data, rows, stream, strategy, sums_in);
        if (ss != TAHOE_OK) return ss;
    } else if (strategy == TAHOE_STRATEGY_QRING) {
        const tahoe_status qs = qring_launch(f, sums, leaf_out, data, rows, stream, timed ? f->ev_mid[f->prof_count] : nullptr, sums_in);
        mid_recorded = timed;
        if (qs != TAHOE_OK) return qs;
    } else if (strategy == TAHOE_STRATEGY_TILERING && tilering_rows(f) == 0 && widef_rows(f) > 0) {
        const tahoe_status ws = widef_launch(f, sums, leaf_out, data, rows, stream, sums_in);  // rows too wide for a 64-row tile
        if (ws != TAHOE_OK) return ws;
    } else if (strategy == TAHOE_STRATEGY_TILERING) {
        const int tr = tilering_rows(f);
        if (tr == 0)
            return fail(TAHOE_ERR_UNSUPPORTED, "TILERING needs a 64-row tile (num_cols <= %d) or an 8-row tile that fits %d B of LDS",
                        kBlockMaxCols, f->lds_limit);
        if (tr == 128)
            launch_tilering<128, 4>(f, sums, leaf_out, sums_in, data, rows, stream, vec4_ok);
        else
            launch_tilering<64, 8>(f, sums, leaf_out, sums_in, data, rows, stream, vec4_ok);
        TAHOE_HIP_TRY(hipGetLastError());
    } else if (strategy == TAHOE_STRATEGY_TILEBLOCK) {
        const int tr = tileblock_rows(f);
        if (tr == 0)
            return fail(TAHOE_ERR_UNSUPPORTED, "TILEBLOCK needs num_cols <= %d and a 64-row tile that fits %d B of LDS",
                        kBlockMaxCols, f->lds_limit);
        if (tr == 128)
            launch_tileblock<128>(f, sums, leaf_out, sums_in, data, rows, stream, vec4_ok);
        else
            launch_tileblock<64>(f, sums, leaf_out, sums_in, data, rows, stream, vec4_ok);
        TAHOE_HIP_TRY(hipGetLastError());
    } else if (strategy == TAHOE_STRATEGY_ROWTILE) {
        if (!rowtile_fits(f))
            return fail(TAHOE_ERR_UNSUPPORTED, "ROWTILE needs %d B of LDS for %d columns; device offers %d",
                        rowtile_lds_bytes(f->p.num_cols, f->lds_levels), f->p.num_cols, f->lds_limit);
        const size_t grid = (rows + kTileRows - 1) / kTileRows;
        const int lds = rowtile_lds_bytes(f->p.num_cols, f->lds_levels);
        if (leaf_out)
            hipLaunchKernelGGL(rowtile_kernel<true>, dim3((unsigned)grid), dim3(kBlock), lds, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->lds_levels, f->p.missing, vec4_ok);
        else
            hipLaunchKernelGGL(rowtile_kernel<false>, dim3((unsigned)grid), dim3(kBlock), lds, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->lds_levels, f->p.missing, vec4_ok);
        TAHOE_HIP_TRY(hipGetLastError());
    } else if (strategy == TAHOE_STRATEGY_DIRECT) {
        const size_t grid = (rows + kBlock - 1) / kBlock;
        if (leaf_out)
            hipLaunchKernelGGL(direct_kernel<true>, dim3((unsigned)grid), dim3(kBlock), 0, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->p.missing);
        else
            hipLaunchKernelGGL(direct_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, sums_in, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->p.missing);
        TAHOE_HIP_TRY(hipGetLastError());
    } else {
        return fail(TAHOE_ERR_INVALID_ARG, "unknown strategy %d", strategy);
    }
    if (timed) {
        if (!mid_recorded) TAHOE_HIP_TRY(hipEventRecord(f->ev_mid[f->prof_count], stream));
        TAHOE_HIP_TRY(hipEventRecord(f->ev_stop[f->prof_count], stream));
        ++f->prof_count;
    }
    return TAHOE_OK;
}

static tahoe_status launch_transform(float *preds, size_t rows, int output, int num_trees, float threshold,
                                     float global_bias, hipStream_t stream)
{
    // forest::predict runs transform_k only when it changes something (Struct.h:263).
    if (rows == 0 || (output == TAHOE_OUT_RAW && global_bias == 0.0f)) return TAHOE_OK;
    const size_t grid = (rows + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(transform_kernel, dim3((unsigned)grid), dim3(kBlock), 0, stream, preds, rows, output,
                       num_trees, threshold, global_bias);
    TAHOE_HIP_TRY(hipGetLastError());
    return TAHOE_OK;
}

// The reference's adaptive-format width rule (Struct.h:1827-1852): bytes of the packed bits word.
static int reference_bits_bytes(int max_fid)
{
    if (max_fid <= 0) return 1;
    const int len = (int)((std::log2((double)max_fid) + 3.0) / 8.0);
    return len == 0 ? 1 : (len == 1 ? 2 : 4);
}

template <typename T>
static hipError_t upload(T **dst, const std::vector<T> &src, size_t *total)
{
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes);
    if (e != hipSuccess) return e;
    *total += bytes;
    if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

}  // namespace tahoe

using namespace tahoe;

extern "C" {

tahoe_status tahoe_forest_create(tahoe_forest **out, const tahoe_dense_node *nodes, const tahoe_forest_params *p)
{
    unsigned flags = 0;
    if (const char *e = getenv("TAHOE_RELAYOUT"))  // experiments: the re-layout without touching the caller
        if (atoi(e) != 0) flags |= TAHOE_CREATE_PROB_RELAYOUT;
    return tahoe_forest_create_ex(out, nodes, p, flags);
}

tahoe_status tahoe_forest_create_ex(tahoe_forest **out, const tahoe_dense_node *nodes, const tahoe_forest_params *p, unsigned flags)
{
    if (!out || !p) return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_create: null argument");
    if ((flags & ~(unsigned)TAHOE_CREATE_PROB_RELAYOUT) != 0) return fail(TAHOE_ERR_INVALID_ARG, "unknown create flags 0x%x", flags);
    *out = nullptr;
    // check_params, BaseTahoeTest.h:490-516
    if (p->depth < 0 || p->depth > 30) return fail(TAHOE_ERR_INVALID_ARG, "depth must be in [0,30], got %d", p->depth);
    if (p->num_trees < 0) return fail(TAHOE_ERR_INVALID_ARG, "num_trees must be non-negative");
    if (p->num_cols < 0) return fail(TAHOE_ERR_INVALID_ARG, "num_cols must be non-negative");
    if (p->algo < TAHOE_ALGO_NAIVE || p->algo > TAHOE_ALGO_BATCH_TREE_REORG)
        return fail(TAHOE_ERR_INVALID_ARG, "algo should be NAIVE, TREE_REORG or BATCH_TREE_REORG");
    if ((p->output & ~(TAHOE_OUT_AVG | TAHOE_OUT_SIGMOID | TAHOE_OUT_THRESHOLD)) != 0)
        return fail(TAHOE_ERR_INVALID_ARG, "output should be a combination of RAW, AVG, SIGMOID and THRESHOLD");
    if (p->num_trees > 0 && !nodes) return fail(TAHOE_ERR_INVALID_ARG, "nodes is null");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(TAHOE_ERR_NO_DEVICE, "no HIP device is visible; libtahoe_amd has no CPU path");
    int dev = 0;
    TAHOE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    TAHOE_HIP_TRY(hipGetDeviceProperties(&prop, dev));

    tahoe_forest *f = new (std::nothrow) tahoe_forest();
    if (!f) return fail(TAHOE_ERR_NO_MEMORY, "tahoe_forest_create");
    const int D = p->depth;             // depth of the trees as given
    const int De = std::max(D, 2);      // depth of the normalised trees
    f->p = *p;
    f->depth = De;
    f->n_inner = ((size_t)1 << De) - 1;
    f->n_leaf = (size_t)1 << De;
    f->device = dev;
    f->num_cus = prop.multiProcessorCount;
    f->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor > 0 ? (int)prop.maxSharedMemoryPerMultiProcessor
                                                                   : (int)prop.sharedMemPerBlock;
    f->lds_levels = std::min(De, kMaxLdsLevels);
    f->top_levels = std::min(De - 2, kTopLevelsMax);
    if (const char *e = getenv("TAHOE_LDS_LEVELS")) {  // tuning knob for experiments; never raises the caps
        f->lds_levels = std::max(0, std::min(f->lds_levels, atoi(e)));
        f->top_levels = std::max(0, std::min(f->top_levels, atoi(e)));
    }
    if (const char *e = getenv("TAHOE_TILE_ROWS")) f->knob_tile_rows = atoi(e);
    if (const char *e = getenv("TAHOE_QRING_WALKERS")) f->knob_qring_walkers = atoi(e);
    f->relayout = (flags & TAHOE_CREATE_PROB_RELAYOUT) != 0;
    // the float32 top / block views pack fid and def_left into 10 / 16 bits: no room for the exchange bit
    f->has_blocks = p->num_cols <= kBlockMaxCols && !f->relayout;

    // ---- normalise: heap records of the perfect depth-De tree ----
    const size_t T = (size_t)p->num_trees;
    const size_t src_nodes = (size_t)tahoe_tree_num_nodes(D);
    const size_t all_nodes = f->n_inner + f->n_leaf;
    std::vector<InnerNode> h_inner(T * f->n_inner);
    std::vector<float> h_leaf(T * f->n_leaf);
    std::vector<uint32_t> h_orig(T * f->n_leaf);
    std::vector<unsigned char> h_real(T * f->n_inner, 0);  // heap records that exist in the original tree
    // trees are independent: normalise them on several host threads; the first invalid node (in tree order) wins
    struct Bad {
        size_t tree = SIZE_MAX, node = 0;
        int kind = 0, fid = 0;  // 1: reachable bottom-level node is not a leaf, 2: fid >= num_cols
    };
    std::vector<Bad> bad_of;
    std::vector<int> max_fid_of;
    std::mutex collect;
    const int num_cols = p->num_cols;
    const size_t n_inner = f->n_inner, n_leaf = f->n_leaf;
    parallel_for(T, 8, [&, n_inner, n_leaf, num_cols](size_t t_lo, size_t t_hi) {
        std::vector<int64_t> inherit(all_nodes);
        Bad bad;
        int max_fid_local = 0;
        for (size_t t = t_lo; t < t_hi && bad.kind == 0; ++t) {
            const tahoe_dense_node *tree = nodes + t * src_nodes;
            for (size_t i = 0; i < all_nodes; ++i) {
                const int64_t up = i ? inherit[(i - 1) / 2] : -1;
                int fid = 0, def_left = 0, is_leaf = 0;
                float value = 0.0f;
                if (up >= 0) {
                    inherit[i] = up;  // below a leaf: unreachable in the original tree
                } else {
                    // i < src_nodes here: a node without a leaf above it is at most on the original bottom
                    // level, because the check below rejects non-leaves there.
                    tahoe_decode_node(&tree[i], &value, nullptr, &fid, &def_left, &is_leaf);
                    if (is_leaf) {
                        inherit[i] = (int64_t)i;
                    } else {
                        inherit[i] = -1;
                        if (2 * i + 2 >= src_nodes || fid >= num_cols) {
                            bad.tree = t;
                            bad.node = i;
                            bad.kind = 2 * i + 2 >= src_nodes ? 1 : 2;
                            bad.fid = fid;
                            break;
                        }
                        max_fid_local = std::max(max_fid_local, fid);
                    }
                }
                if (i < n_inner) {
                    InnerNode &n = h_inner[t * n_inner + i];
                    if (inherit[i] >= 0) {
                        n.thr = 0.0f;  // padding below a leaf: either child leads to the same value
                        n.meta = 0u;
                    } else {
                        n.thr = value;
                        n.meta = (uint32_t)fid | (def_left ? 0x80000000u : 0u);
                        h_real[t * n_inner + i] = 1;
                    }
                } else {
                    const size_t b = t * n_leaf + (i - n_inner);
                    h_leaf[b] = tree[inherit[i]].val;
                    h_orig[b] = (uint32_t)inherit[i];
                }
            }
        }
        std::lock_guard<std::mutex> lock(collect);
        if (bad.kind) bad_of.push_back(bad);
        max_fid_of.push_back(max_fid_local);
    });
    if (!bad_of.empty()) {
        const Bad bad = *std::min_element(bad_of.begin(), bad_of.end(), [](const Bad &x, const Bad &y) { return x.tree < y.tree; });
        delete f;
        if (bad.kind == 1)
            return fail(TAHOE_ERR_INVALID_FOREST,
                        "tree %zu: reachable bottom-level node %zu is not a leaf (the reference would walk out of the tree)",
                        bad.tree, bad.node);
        return fail(TAHOE_ERR_INVALID_FOREST, "tree %zu node %zu: fid %d >= num_cols %d", bad.tree, bad.node, bad.fid, num_cols);
    }
    // ---- probability-guided re-layout (dense_adaptive_forest::init step (ii), Struct.h:1775-1825 with
    // swap_child :1712-1750): bottom-up, at every internal node whose left child weighs less than its right child the
    // two subtrees change places and the node gets the exchange bit, so the likelier child is always the left one.
    // leaf_orig moves with the leaves: leaf indices stay in the original heap numbering.
    if (f->relayout) {
        std::vector<size_t> swaps_of;
        parallel_for(T, 8, [&, n_inner, n_leaf](size_t t_lo, size_t t_hi) {
            std::vector<float> w(all_nodes);
            size_t swaps = 0;
            for (size_t t = t_lo; t < t_hi; ++t) {
                const tahoe_dense_node *tree = nodes + t * src_nodes;
                for (size_t i = 0; i < all_nodes; ++i) w[i] = i < src_nodes ? tree[i].weight : 0.0f;
                InnerNode *in = &h_inner[t * n_inner];
                unsigned char *re = &h_real[t * n_inner];
                float *lv = &h_leaf[t * n_leaf];
                uint32_t *lo = &h_orig[t * n_leaf];
                for (int l = De - 1; l >= 0; --l) {
                    for (size_t i = ((size_t)1 << l) - 1; i < ((size_t)2 << l) - 1; ++i) {
                        if (!re[i]) continue;                    // padding below a leaf: nothing to order
                        const size_t a = 2 * i + 1, b = 2 * i + 2;
                        if (!(w[a] < w[b])) continue;            // Struct.h:1791: swap when the left child is the lighter one
                        for (int d = 0;; ++d) {                  // the two subtrees, level by level
                            const size_t sa = ((a + 1) << d) - 1, sb = ((b + 1) << d) - 1, len = (size_t)1 << d;
                            if (sa >= n_inner) {                 // leaf level of the normalised tree
                                std::swap_ranges(lv + (sa - n_inner), lv + (sa - n_inner) + len, lv + (sb - n_inner));
                                std::swap_ranges(lo + (sa - n_inner), lo + (sa - n_inner) + len, lo + (sb - n_inner));
                                break;
                            }
                            std::swap_ranges(in + sa, in + sa + len, in + sb);
                            std::swap_ranges(re + sa, re + sa + len, re + sb);
                            std::swap_ranges(w.begin() + sa, w.begin() + sa + len, w.begin() + sb);
                        }
                        in[i].meta |= kMetaExchange;
                        ++swaps;
                    }
                }
            }
            std::lock_guard<std::mutex> lock(collect);
            swaps_of.push_back(swaps);
        });
        for (size_t n : swaps_of) f->relayout_swaps += n;
    }
    int max_fid = 0;
    for (int m : max_fid_of) max_fid = std::max(max_fid, m);
    f->bits_bytes = reference_bits_bytes(max_fid);

    // ---- TILEBLOCK views: SoA tops and 32-byte bottom blocks ----
    std::vector<unsigned char> h_top;
    std::vector<uint4> h_blocks;
    if (f->has_blocks) {
        const int n_top = top_nodes(f->top_levels);
        const int stride = top_stride_bytes(f->top_levels);
        const int meta_off = top_thr_bytes(f->top_levels);
        h_top.assign(T * (size_t)stride, 0);
        const size_t n_blocks = (size_t)1 << (De - 2);
        const size_t first = n_blocks - 1;
        h_blocks.resize(T * n_blocks * 2);
        parallel_for(T, 8, [&, n_inner, n_leaf](size_t t_lo, size_t t_hi) {
        for (size_t t = t_lo; t < t_hi; ++t) {
            const InnerNode *in = &h_inner[t * n_inner];
            float *thr = reinterpret_cast<float *>(&h_top[t * stride]);
            uint16_t *meta = reinterpret_cast<uint16_t *>(&h_top[t * stride + meta_off]);
            for (int i = 0; i < n_top; ++i) {  // heap node i -> position i + 1
                thr[i + 1] = in[i].thr;
                meta[i + 1] = (uint16_t)((in[i].meta & 0x7fffu) | ((in[i].meta >> 31) << 15));
            }
            for (size_t b = 0; b < n_blocks; ++b) {
                const size_t r = first + b, l = 2 * r + 1, rr = 2 * r + 2;  // subtree root and its children
                auto pack = [](const InnerNode &n) { return (n.meta & 0x1ffu) | ((n.meta >> 31) << 9); };
                uint4 a, v;
                memcpy(&a.x, &in[r].thr, 4);
                memcpy(&a.y, &in[l].thr, 4);
                memcpy(&a.z, &in[rr].thr, 4);
                a.w = pack(in[r]) | (pack(in[l]) << 10) | (pack(in[rr]) << 20);
                const float *lv = &h_leaf[t * n_leaf + 4 * b];
                memcpy(&v.x, &lv[0], 4);
                memcpy(&v.y, &lv[1], 4);
                memcpy(&v.z, &lv[2], 4);
                memcpy(&v.w, &lv[3], 4);
                h_blocks[(t * n_blocks + b) * 2 + 0] = a;
                h_blocks[(t * n_blocks + b) * 2 + 1] = v;
            }
        }
        });
    }

    auto bail = [&](hipError_t e, const char *what) {
        tahoe_forest_destroy(f);
        return fail(TAHOE_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = upload(&f->inner, h_inner, &f->device_bytes)) != hipSuccess) return bail(e, "upload(inner)");
    if ((e = upload(&f->leaf_val, h_leaf, &f->device_bytes)) != hipSuccess) return bail(e, "upload(leaf_val)");
    if ((e = upload(&f->leaf_orig, h_orig, &f->device_bytes)) != hipSuccess) return bail(e, "upload(leaf_orig)");
    if (f->has_blocks) {
        if ((e = upload(&f->top, h_top, &f->device_bytes)) != hipSuccess) return bail(e, "upload(top)");
        if ((e = upload(&f->blocks, h_blocks, &f->device_bytes)) != hipSuccess) return bail(e, "upload(blocks)");
    }

    // Kernels that may need more than the default 64 KiB of dynamic LDS.
    if (rowtile_fits(f)) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&rowtile_kernel<false>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(rowtile)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&rowtile_kernel<true>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(rowtile)");
    }
    if (f->has_blocks && tileblock_lds_bytes(f, 128) <= f->lds_limit) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tileblock_kernel<128, false>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tileblock)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tileblock_kernel<128, true>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tileblock)");
    }
    if ((e = hipMalloc(reinterpret_cast<void **>(&f->error_flag), sizeof(int))) != hipSuccess)
        return bail(e, "hipMalloc(error_flag)");
    if ((e = hipMemset(f->error_flag, 0, sizeof(int))) != hipSuccess) return bail(e, "hipMemset(error_flag)");
    {
        const tahoe_status qs = qring_build(f, h_inner, h_real, h_leaf);
        if (qs != TAHOE_OK) {
            tahoe_forest_destroy(f);
            return qs;
        }
    }
    if (tilering_rows(f) == 0 && tileblock_rows(f) == 0) {  // no 64-row float32 tile kernel for this shape: the wide-row form
        const tahoe_status ws = widef_build(f, h_inner, h_real, h_leaf);
        if (ws != TAHOE_OK) {
            tahoe_forest_destroy(f);
            return ws;
        }
    }
    if (f->has_blocks && tilering_lds_bytes(f, 64) <= f->lds_limit) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tilering_kernel<64, 8, false>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tilering)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tilering_kernel<64, 8, true>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tilering)");
    }
    if (f->has_blocks && tilering_lds_bytes(f, 128) <= f->lds_limit) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tilering_kernel<128, 4, false>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tilering)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tilering_kernel<128, 4, true>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tilering)");
    }
    if (f->has_blocks && tileblock_lds_bytes(f, 64) <= f->lds_limit) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tileblock_kernel<64, false>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tileblock)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&tileblock_kernel<64, true>), f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(tileblock)");
    }
    *out = f;
    return TAHOE_OK;
}

void tahoe_forest_destroy(tahoe_forest *f)
{
    if (!f) return;
    if (f->inner) (void)hipFree(f->inner);
    if (f->leaf_val) (void)hipFree(f->leaf_val);
    if (f->leaf_orig) (void)hipFree(f->leaf_orig);
    if (f->top) (void)hipFree(f->top);
    if (f->blocks) (void)hipFree(f->blocks);
    if (f->error_flag) (void)hipFree(f->error_flag);
    pipeline_destroy(f);
    qring_destroy(f);
    sparse_destroy(f);
    widef_destroy(f);
    for (hipEvent_t e : f->ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->ev_mid) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->ev_stop) (void)hipEventDestroy(e);
    delete f;
}

tahoe_status tahoe_forest_predict_raw(tahoe_forest *f, float *sums_dev, const float *data_dev, size_t rows,
                                      void *stream)
{
    if (!f || (rows && (!sums_dev || !data_dev)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict_raw: null argument");
    return launch_traversal(f, sums_dev, nullptr, data_dev, rows, (hipStream_t)stream);
}

tahoe_status tahoe_forest_predict_accumulate(tahoe_forest *f, float *sums_dev, const float *data_dev, size_t rows,
                                             void *stream)
{
    if (!f || (rows && (!sums_dev || !data_dev)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict_accumulate: null argument");
    return launch_traversal(f, sums_dev, nullptr, data_dev, rows, (hipStream_t)stream, sums_dev);
}

tahoe_status tahoe_forest_predict(tahoe_forest *f, float *preds_dev, const float *data_dev, size_t rows,
                                  void *stream)
{
    if (!f || (rows && (!preds_dev || !data_dev)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict: null argument");
    DeviceGuard on_device(f->device);
    tahoe_status s = launch_traversal(f, preds_dev, nullptr, data_dev, rows, (hipStream_t)stream);
    if (s != TAHOE_OK) return s;
    return launch_transform(preds_dev, rows, f->p.output, f->p.num_trees, f->p.threshold, f->p.global_bias,
                            (hipStream_t)stream);
}

tahoe_status tahoe_forest_predict_leaf_idx(tahoe_forest *f, uint32_t *leaf_dev, float *sums_dev,
                                           const float *data_dev, size_t rows, void *stream)
{
    if (!f || (rows && (!leaf_dev || !data_dev)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict_leaf_idx: null argument");
    return launch_traversal(f, sums_dev, leaf_dev, data_dev, rows, (hipStream_t)stream);
}

tahoe_status tahoe_transform_preds(float *preds_dev, size_t rows, int output, int num_trees_total, float threshold,
                                   float global_bias, void *stream)
{
    if (rows && !preds_dev) return fail(TAHOE_ERR_INVALID_ARG, "tahoe_transform_preds: null argument");
    return launch_transform(preds_dev, rows, output, num_trees_total, threshold, global_bias, (hipStream_t)stream);
}

tahoe_status tahoe_forest_set_strategy(tahoe_forest *f, int strategy)
{
    if (!f) return fail(TAHOE_ERR_INVALID_ARG, "null forest");
    if (strategy < TAHOE_STRATEGY_AUTO || strategy > TAHOE_STRATEGY_QRING)
        return fail(TAHOE_ERR_INVALID_ARG, "unknown strategy %d", strategy);
    if (f->sp) {
        if ((strategy > TAHOE_STRATEGY_TILEBLOCK && strategy != TAHOE_STRATEGY_QRING) ||
            (strategy == TAHOE_STRATEGY_ROWTILE && !sparse_tile_fits(f)) ||
            (strategy == TAHOE_STRATEGY_TILEBLOCK && sparse_top_waves(f) == 0) ||
            (strategy == TAHOE_STRATEGY_QRING && !sparse_q_available(f)))
            return fail(TAHOE_ERR_UNSUPPORTED,
                        "a sparse forest runs AUTO, DIRECT, ROWTILE (a 64-row tile fits LDS), TILEBLOCK (tile + tree tops in LDS; "
                        "trees of <= 65535 nodes, num_cols <= 32767) or QRING (quantised tile + tree tops; also num_cols <= 256)");
        f->strategy = strategy;
        return TAHOE_OK;
    }
    if (strategy == TAHOE_STRATEGY_ROWTILE && !rowtile_fits(f))
        return fail(TAHOE_ERR_UNSUPPORTED, "ROWTILE needs %d B of LDS for %d columns; device offers %d",
                    rowtile_lds_bytes(f->p.num_cols, f->lds_levels), f->p.num_cols, f->lds_limit);
    if (strategy == TAHOE_STRATEGY_TILEBLOCK && tileblock_rows(f) == 0)
        return fail(TAHOE_ERR_UNSUPPORTED, "TILEBLOCK needs num_cols <= %d and a 64-row tile that fits %d B of LDS",
                    kBlockMaxCols, f->lds_limit);
    if (strategy == TAHOE_STRATEGY_TILERING && tilering_rows(f) == 0 && widef_rows(f) == 0)
        return fail(TAHOE_ERR_UNSUPPORTED, "TILERING needs a 64-row tile (num_cols <= %d) or an 8-row tile that fits %d B of LDS",
                    kBlockMaxCols, f->lds_limit);
    if (strategy == TAHOE_STRATEGY_QRING && qring_walkers(f) == 0)
        return fail(TAHOE_ERR_UNSUPPORTED,
                    "QRING needs <= 32767 distinct thresholds per feature, num_cols <= 32767 and a 128-row u16 tile in LDS");
    f->strategy = strategy;
    return TAHOE_OK;
}

tahoe_status tahoe_forest_reserve(tahoe_forest *f, size_t rows)
{
    if (!f) return fail(TAHOE_ERR_INVALID_ARG, "null forest");
    const tahoe_status qs = qring_reserve(f, rows);
    if (qs != TAHOE_OK) return qs;
    return widef_reserve(f, rows);  // (the wide-row float32 form's leaf-value workspace, when it streams rows)
}

tahoe_status tahoe_forest_check(tahoe_forest *f, void *stream)
{
    if (!f) return fail(TAHOE_ERR_INVALID_ARG, "null forest");
    TAHOE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (f->error_flag) {
        int flag = 0;
        TAHOE_HIP_TRY(hipMemcpy(&flag, f->error_flag, sizeof(int), hipMemcpyDeviceToHost));
        if (flag != 0) return fail(TAHOE_ERR_HIP, "a bounded LDS ring wait timed out (TILERING/QRING); results are invalid");
    }
    return TAHOE_OK;
}

int tahoe_forest_get_strategy(const tahoe_forest *f, size_t rows) { return f ? resolve_strategy(f, rows) : -1; }

int tahoe_forest_get_kernel_form(const tahoe_forest *f, size_t rows)
{
    if (!f) return -1;
    const int strategy = resolve_strategy(f, rows);
    if (f->sp) {
        switch (strategy) {
        case TAHOE_STRATEGY_QRING: return TAHOE_FORM_SPARSE_QRING;
        case TAHOE_STRATEGY_TILEBLOCK: return TAHOE_FORM_SPARSE_TOP;
        case TAHOE_STRATEGY_ROWTILE: return TAHOE_FORM_SPARSE_ROWTILE;
        default: return TAHOE_FORM_SPARSE_DIRECT;
        }
    }
    switch (strategy) {
    case TAHOE_STRATEGY_QRING: return qring_form(f, rows);
    case TAHOE_STRATEGY_TILERING:
        if (tilering_rows(f) > 0) return TAHOE_FORM_TILERING_TILE;
        if (widef_rows(f) > 0) return widef_stream_slots(f) > 0 ? TAHOE_FORM_TILERING_WIDE_STREAM : TAHOE_FORM_TILERING_WIDE_TILE;
        return TAHOE_FORM_NONE;
    case TAHOE_STRATEGY_TILEBLOCK: return tileblock_rows(f) > 0 ? TAHOE_FORM_TILEBLOCK : TAHOE_FORM_NONE;
    case TAHOE_STRATEGY_ROWTILE: return rowtile_fits(f) ? TAHOE_FORM_ROWTILE : TAHOE_FORM_NONE;
    case TAHOE_STRATEGY_DIRECT: return TAHOE_FORM_DIRECT;
    default: return TAHOE_FORM_NONE;
    }
}

const char *tahoe_kernel_form_name(int form)
{
    static const char *const names[] = {"none", "direct", "rowtile", "tileblock", "tilering_tile", "tilering_wide_tile",
                                        "tilering_wide_stream", "qring_region3", "qring_region2", "qring_region_mixed",
                                        "qring_split", "qring_columns", "qring_wide", "qring_gx", "sparse_direct",
                                        "sparse_rowtile", "sparse_top", "sparse_qring", "qring_region8", "qring_region6"};
    return form >= 0 && form < (int)(sizeof(names) / sizeof(names[0])) ? names[form] : "?";
}

tahoe_status tahoe_forest_get_info(const tahoe_forest *f, tahoe_forest_info *info)
{
    if (!f || !info) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    memset(info, 0, sizeof(*info));
    info->is_sparse = f->sp != nullptr;
    if (f->sp) {
        info->num_trees = f->p.num_trees;
        info->num_cols = f->p.num_cols;
        info->bits_bytes = f->bits_bytes;
        info->device_bytes = f->device_bytes;
        info->device_id = f->device;
        info->num_cus = f->num_cus;
        return TAHOE_OK;
    }
    info->num_trees = f->p.num_trees;
    info->depth = f->p.depth;
    info->num_cols = f->p.num_cols;
    info->bits_bytes = f->bits_bytes;
    info->lds_levels = f->lds_levels;
    info->device_bytes = f->device_bytes;
    info->lds_bytes_per_block = rowtile_fits(f) ? rowtile_lds_bytes(f->p.num_cols, f->lds_levels) : 0;
    info->device_id = f->device;
    info->num_cus = f->num_cus;
    info->top_levels = f->top_levels;
    info->tile_rows = tileblock_rows(f);
    info->tileblock_lds_bytes = info->tile_rows ? (int)tileblock_lds_bytes(f, info->tile_rows) : 0;
    info->qring_walkers = qring_walkers(f);
    info->qring_lds_bytes = (int)qring_lds_bytes(f);
    info->qring_groups = qring_groups(f);
    info->ring_rows = tilering_rows(f) ? tilering_rows(f) : widef_rows(f);
    // (the wide-row float32 form has its own LDS plan: tile + walker slots + ring, or tops image + row slots when streaming)
    info->tilering_lds_bytes = tilering_rows(f) ? (int)tilering_lds_bytes(f, info->ring_rows) : (int)widef_lds_bytes(f);
    info->stream_slots = widef_stream_slots(f);
    info->stream_levels = widef_stream_levels(f);
    info->stream_key_ties = widef_stream_tie_estimate(f);
    info->qring_tile_rows = info->qring_walkers == 0 ? 0 : qwide_rows(f) ? qwide_rows(f) : (qring_code8(f) || qring_six16(f)) ? 384 : qring_regions(f) ? 192 : qring_lds_tile(f) ? 128 : 0;
    info->relayout = f->relayout ? 1 : 0;
    info->relayout_swaps = f->relayout_swaps;
    return TAHOE_OK;
}

tahoe_status tahoe_forest_set_profiling(tahoe_forest *f, int max_launches)
{
    if (!f || max_launches < 0 || max_launches > 4096) return fail(TAHOE_ERR_INVALID_ARG, "bad argument");
    while ((int)f->ev_start.size() < max_launches) {
        hipEvent_t a, b;
        TAHOE_HIP_TRY(hipEventCreate(&a));
        hipError_t e = hipEventCreate(&b);
        if (e != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail(TAHOE_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
        hipEvent_t m;
        e = hipEventCreate(&m);
        if (e != hipSuccess) {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
            return fail(TAHOE_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
        f->ev_start.push_back(a);
        f->ev_mid.push_back(m);
        f->ev_stop.push_back(b);
    }
    f->profiling = max_launches > 0;
    f->prof_count = 0;
    return TAHOE_OK;
}

tahoe_status tahoe_forest_kernel_times(tahoe_forest *f, float *ms_out, int capacity, int *count)
{
    if (!f || !count || (capacity > 0 && !ms_out)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    const int n = (int)std::min<size_t>(f->prof_count, (size_t)std::max(capacity, 0));
    for (int i = 0; i < n; ++i) {
        TAHOE_HIP_TRY(hipEventSynchronize(f->ev_stop[i]));
        TAHOE_HIP_TRY(hipEventElapsedTime(&ms_out[i], f->ev_mid[i], f->ev_stop[i]));
    }
    *count = n;
    return TAHOE_OK;
}

tahoe_status tahoe_forest_prepass_times(tahoe_forest *f, float *ms_out, int capacity, int *count)
{
    if (!f || !count || (capacity > 0 && !ms_out)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    const int n = (int)std::min<size_t>(f->prof_count, (size_t)std::max(capacity, 0));
    for (int i = 0; i < n; ++i) {
        TAHOE_HIP_TRY(hipEventSynchronize(f->ev_mid[i]));
        TAHOE_HIP_TRY(hipEventElapsedTime(&ms_out[i], f->ev_start[i], f->ev_mid[i]));
    }
    *count = n;
    return TAHOE_OK;
}

}  // extern "C"
