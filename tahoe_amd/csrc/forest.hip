// Forest handle, device layout and traversal kernels for gfx950 (MI355X).
//
// What this replaces in the reference (file:line into sampathrg/Tahoe):
//   dense_forest / dense_adaptive_forest  (init, infer, predict)     Struct.h:802-861, :1710-2210
//   the walkers infer_one_tree_* and kernels infer_k_* / infer_adaptive_reorg_*   Struct.h:359-1704
//   cub::BlockReduce / DeviceSegmentedReduce call sites (table 2b of SURVEY.md)  -> ordered
//     per-row accumulation (no tree-parallel float reduction, see "sum order" below)
//   transform_k                                                      Struct.h:196-209
//
// Device layout ("perfect-tree" SoA-of-records, built once at create):
//   Every tree is normalised to a perfect binary tree of depth D: a leaf found above the bottom
//   level is pushed down -- all bottom-level descendants inherit its value and remember its
//   original heap index.  A walk is then exactly D compare-and-step iterations with no leaf test,
//   so all 64 lanes of a wave stay converged, and the per-(row,tree) result is one gather from
//   the bottom array.  Per tree t:
//     inner[t][2^D - 1]  {float thr; uint32 meta}   meta = fid | def_left << 31   (heap order)
//     leaf_val[t][2^D]   float
//     leaf_orig[t][2^D]  uint32  original heap index of the leaf (for predict_leaf_idx)
//
// Sum order: the CPU predictor adds leaf values in tree order 0..T-1 in float32
// (BaseTahoeTest.h:462-466).  Every kernel here adds them in exactly that order per row, so the raw
// sums are bit-identical to the CPU's, not merely within tolerance.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

namespace tahoe {

struct InnerNode {
    float thr;
    uint32_t meta;  // fid | def_left << 31
};
static_assert(sizeof(InnerNode) == 8, "InnerNode must be 8 bytes");

constexpr int kBlock = 256;        // threads per workgroup (4 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kTileRows = 64;      // rows per ROWTILE workgroup = one wave of lanes
constexpr int kMaxLdsLevels = 8;   // top levels of a tree staged per wave (255 nodes = 2040 B)
constexpr float kMissingEps = 1.0e-6f;  // BaseTahoeTest.h:451

}  // namespace tahoe

struct tahoe_forest {
    tahoe_forest_params p{};
    int depth = 0;
    size_t n_inner = 0;  // 2^D - 1
    size_t n_leaf = 0;   // 2^D
    int bits_bytes = 0;
    int strategy = TAHOE_STRATEGY_AUTO;
    int device = 0;
    int num_cus = 0;
    int lds_limit = 0;
    int lds_levels = 0;
    tahoe::InnerNode *inner = nullptr;
    float *leaf_val = nullptr;
    uint32_t *leaf_orig = nullptr;
    size_t device_bytes = 0;
    // Profiling: one hipEvent pair per traversal launch, read back after the stream has drained.
    bool profiling = false;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t prof_count = 0;  // launches recorded since profiling was (re-)enabled
};

namespace tahoe {

// ------------------------------------------------------------------------------------------------
// One compare-and-step: the branch rule of infer_one_tree, BaseTahoeTest.h:450-453.
__device__ __forceinline__ uint32_t step(uint32_t idx, float thr, uint32_t meta, float x, float missing)
{
    const bool def_left = (meta >> 31) != 0;
    const bool is_missing = fabsf(x - missing) <= kMissingEps;
    const bool cond = is_missing ? !def_left : (x >= thr);
    return 2u * idx + 1u + (cond ? 1u : 0u);
}

// ------------------------------------------------------------------------------------------------
// DIRECT: lane = row; nodes and features come straight from global memory.  Works for any shape.
// Analogue of infer_adaptive_reorg_* (Struct.h:1196-1240).
template <bool WRITE_LEAF>
__global__ void __launch_bounds__(kBlock) direct_kernel(const InnerNode *__restrict__ inner,
                                                        const float *__restrict__ leaf_val,
                                                        const uint32_t *__restrict__ leaf_orig,
                                                        const float *__restrict__ data, float *__restrict__ sums,
                                                        uint32_t *__restrict__ leaf_out, size_t rows, int cols,
                                                        int num_trees, int depth, float missing)
{
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= rows) return;
    const float *x = data + row * (size_t)cols;
    const size_t n_inner = ((size_t)1 << depth) - 1;
    const size_t n_leaf = (size_t)1 << depth;
    float sum = 0.0f;
    for (int t = 0; t < num_trees; ++t) {
        const InnerNode *tree = inner + (size_t)t * n_inner;
        uint32_t idx = 0;
        for (int l = 0; l < depth; ++l) {
            const InnerNode n = tree[idx];
            idx = step(idx, n.thr, n.meta, x[n.meta & 0x7fffffffu], missing);
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
// added by the row's owner lane in tree order.
//
// Dynamic LDS: [cols][64] float | kWaves slots of slot_nodes InnerNode | [2][kWaves][64] float.
template <bool WRITE_LEAF>
__global__ void __launch_bounds__(kBlock) rowtile_kernel(const InnerNode *__restrict__ inner,
                                                         const float *__restrict__ leaf_val,
                                                         const uint32_t *__restrict__ leaf_orig,
                                                         const float *__restrict__ data, float *__restrict__ sums,
                                                         uint32_t *__restrict__ leaf_out, size_t rows, int cols,
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
                const float x = tile[(n.meta & 0x7fffffffu) * kTileRows + lane];
                idx = step(idx, n.thr, n.meta, x, missing);
            }
            const InnerNode *tree = inner + (size_t)t * n_inner;
            for (int l = lds_levels; l < depth; ++l) {
                const InnerNode n = tree[idx];
                const float x = tile[(n.meta & 0x7fffffffu) * kTileRows + lane];
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

static int resolve_strategy(const tahoe_forest *f, size_t /*rows*/)
{
    if (f->strategy != TAHOE_STRATEGY_AUTO) return f->strategy;
    return rowtile_fits(f) ? TAHOE_STRATEGY_ROWTILE : TAHOE_STRATEGY_DIRECT;
}

static tahoe_status launch_traversal(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data,
                                     size_t rows, hipStream_t stream)
{
    if (rows == 0) return TAHOE_OK;
    const int strategy = resolve_strategy(f, rows);
    const bool timed = f->profiling && f->prof_count < f->ev_start.size();
    if (timed) TAHOE_HIP_TRY(hipEventRecord(f->ev_start[f->prof_count], stream));
    if (f->p.num_trees == 0) {
        // Nothing to walk: sums are zero (an empty j-loop in predict_on_cpu).
        if (sums) TAHOE_HIP_TRY(hipMemsetAsync(sums, 0, rows * sizeof(float), stream));
    } else if (strategy == TAHOE_STRATEGY_ROWTILE) {
        if (!rowtile_fits(f))
            return fail(TAHOE_ERR_UNSUPPORTED, "ROWTILE needs %d B of LDS for %d columns; device offers %d",
                        rowtile_lds_bytes(f->p.num_cols, f->lds_levels), f->p.num_cols, f->lds_limit);
        const size_t grid = (rows + kTileRows - 1) / kTileRows;
        if (grid > 0x7fffffffu) return fail(TAHOE_ERR_INVALID_ARG, "too many rows for one launch: %zu", rows);
        const int lds = rowtile_lds_bytes(f->p.num_cols, f->lds_levels);
        const int vec4_ok = (f->p.num_cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0);
        if (leaf_out)
            hipLaunchKernelGGL(rowtile_kernel<true>, dim3((unsigned)grid), dim3(kBlock), lds, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->lds_levels, f->p.missing, vec4_ok);
        else
            hipLaunchKernelGGL(rowtile_kernel<false>, dim3((unsigned)grid), dim3(kBlock), lds, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->lds_levels, f->p.missing, vec4_ok);
        TAHOE_HIP_TRY(hipGetLastError());
    } else if (strategy == TAHOE_STRATEGY_DIRECT) {
        const size_t grid = (rows + kBlock - 1) / kBlock;
        if (grid > 0x7fffffffu) return fail(TAHOE_ERR_INVALID_ARG, "too many rows for one launch: %zu", rows);
        if (leaf_out)
            hipLaunchKernelGGL(direct_kernel<true>, dim3((unsigned)grid), dim3(kBlock), 0, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->p.missing);
        else
            hipLaunchKernelGGL(direct_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, stream, f->inner,
                               f->leaf_val, f->leaf_orig, data, sums, leaf_out, rows, f->p.num_cols,
                               f->p.num_trees, f->depth, f->p.missing);
        TAHOE_HIP_TRY(hipGetLastError());
    } else {
        return fail(TAHOE_ERR_INVALID_ARG, "unknown strategy %d", strategy);
    }
    if (timed) {
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

}  // namespace tahoe

using namespace tahoe;

extern "C" {

tahoe_status tahoe_forest_create(tahoe_forest **out, const tahoe_dense_node *nodes, const tahoe_forest_params *p)
{
    if (!out || !p) return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_create: null argument");
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
    f->p = *p;
    f->depth = p->depth;
    f->n_inner = ((size_t)1 << p->depth) - 1;
    f->n_leaf = (size_t)1 << p->depth;
    f->device = dev;
    f->num_cus = prop.multiProcessorCount;
    f->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor > 0 ? (int)prop.maxSharedMemoryPerMultiProcessor
                                                                   : (int)prop.sharedMemPerBlock;
    f->lds_levels = std::min(p->depth, kMaxLdsLevels);

    const size_t T = (size_t)p->num_trees;
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(p->depth);
    std::vector<InnerNode> h_inner(std::max<size_t>(T * f->n_inner, 1));
    std::vector<float> h_leaf(std::max<size_t>(T * f->n_leaf, 1));
    std::vector<uint32_t> h_orig(std::max<size_t>(T * f->n_leaf, 1));
    std::vector<int64_t> inherit(per_tree);
    int max_fid = 0;
    for (size_t t = 0; t < T; ++t) {
        const tahoe_dense_node *tree = nodes + t * per_tree;
        for (size_t i = 0; i < per_tree; ++i) {
            int fid, def_left, is_leaf;
            float value;
            tahoe_decode_node(&tree[i], &value, nullptr, &fid, &def_left, &is_leaf);
            const int64_t up = i ? inherit[(i - 1) / 2] : -1;
            if (up >= 0)
                inherit[i] = up;  // below a leaf: unreachable in the original tree
            else if (is_leaf)
                inherit[i] = (int64_t)i;
            else {
                inherit[i] = -1;
                if (i >= f->n_inner) {
                    delete f;
                    return fail(TAHOE_ERR_INVALID_FOREST,
                                "tree %zu: reachable bottom-level node %zu is not a leaf (the reference would walk "
                                "out of the tree)", t, i);
                }
                if (fid >= p->num_cols) {
                    delete f;
                    return fail(TAHOE_ERR_INVALID_FOREST, "tree %zu node %zu: fid %d >= num_cols %d", t, i, fid,
                                p->num_cols);
                }
                max_fid = std::max(max_fid, fid);
            }
            if (i < f->n_inner) {
                InnerNode &n = h_inner[t * f->n_inner + i];
                if (inherit[i] >= 0) {
                    n.thr = 0.0f;
                    n.meta = 0u;
                } else {
                    n.thr = value;
                    n.meta = (uint32_t)fid | (def_left ? 0x80000000u : 0u);
                }
            } else {
                const size_t b = t * f->n_leaf + (i - f->n_inner);
                h_leaf[b] = tree[inherit[i]].val;
                h_orig[b] = (uint32_t)inherit[i];
            }
        }
    }
    f->bits_bytes = reference_bits_bytes(max_fid);

    auto bail = [&](hipError_t e, const char *what) {
        tahoe_forest_destroy(f);
        return fail(TAHOE_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    const size_t inner_bytes = h_inner.size() * sizeof(InnerNode), leaf_bytes = h_leaf.size() * sizeof(float),
                 orig_bytes = h_orig.size() * sizeof(uint32_t);
    if ((e = hipMalloc(&f->inner, inner_bytes)) != hipSuccess) return bail(e, "hipMalloc(inner)");
    if ((e = hipMalloc(&f->leaf_val, leaf_bytes)) != hipSuccess) return bail(e, "hipMalloc(leaf_val)");
    if ((e = hipMalloc(&f->leaf_orig, orig_bytes)) != hipSuccess) return bail(e, "hipMalloc(leaf_orig)");
    f->device_bytes = inner_bytes + leaf_bytes + orig_bytes;
    if ((e = hipMemcpy(f->inner, h_inner.data(), inner_bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(inner)");
    if ((e = hipMemcpy(f->leaf_val, h_leaf.data(), leaf_bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(leaf_val)");
    if ((e = hipMemcpy(f->leaf_orig, h_orig.data(), orig_bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(leaf_orig)");

    // Kernels that may need more than the default 64 KiB of dynamic LDS.
    if (rowtile_fits(f)) {
        const int lds = rowtile_lds_bytes(p->num_cols, f->lds_levels);
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowtile_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess)
            return bail(e, "hipFuncSetAttribute(rowtile<false>)");
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowtile_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess)
            return bail(e, "hipFuncSetAttribute(rowtile<true>)");
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
    for (hipEvent_t e : f->ev_start) (void)hipEventDestroy(e);
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

tahoe_status tahoe_forest_predict(tahoe_forest *f, float *preds_dev, const float *data_dev, size_t rows,
                                  void *stream)
{
    if (!f || (rows && (!preds_dev || !data_dev)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict: null argument");
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
    if (strategy < TAHOE_STRATEGY_AUTO || strategy > TAHOE_STRATEGY_ROWTILE)
        return fail(TAHOE_ERR_INVALID_ARG, "unknown strategy %d", strategy);
    if (strategy == TAHOE_STRATEGY_ROWTILE && !rowtile_fits(f))
        return fail(TAHOE_ERR_UNSUPPORTED, "ROWTILE needs %d B of LDS for %d columns; device offers %d",
                    rowtile_lds_bytes(f->p.num_cols, f->lds_levels), f->p.num_cols, f->lds_limit);
    f->strategy = strategy;
    return TAHOE_OK;
}

int tahoe_forest_get_strategy(const tahoe_forest *f, size_t rows) { return f ? resolve_strategy(f, rows) : -1; }

tahoe_status tahoe_forest_get_info(const tahoe_forest *f, tahoe_forest_info *info)
{
    if (!f || !info) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    memset(info, 0, sizeof(*info));
    info->num_trees = f->p.num_trees;
    info->depth = f->depth;
    info->num_cols = f->p.num_cols;
    info->bits_bytes = f->bits_bytes;
    info->lds_levels = f->lds_levels;
    info->device_bytes = f->device_bytes;
    info->lds_bytes_per_block = rowtile_fits(f) ? rowtile_lds_bytes(f->p.num_cols, f->lds_levels) : 0;
    info->device_id = f->device;
    info->num_cus = f->num_cus;
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
        f->ev_start.push_back(a);
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
        TAHOE_HIP_TRY(hipEventElapsedTime(&ms_out[i], f->ev_start[i], f->ev_stop[i]));
    }
    *count = n;
    return TAHOE_OK;
}

}  // extern "C"
