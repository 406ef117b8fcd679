// Sparse (irregular) forests: the reference's sparse_node_t / sparse_storage (Struct.h:50-54, 334-354), the
// walker infer_one_tree_sparse (Struct.h:2217-2250) and sparse_forest (Struct.h:2327-2353), plus the
// dense -> sparse converter of the (commented-out) harness (BaseTahoeTest.h:728-764).  Dead code in the
// reference, but the only reference-defined format that can hold trees of depth 4..24 (config K5): a dense
// depth-24 tree would need 2^25 nodes.
//
// Branch rule: the live one (BaseTahoeTest.h:452, |x - missing| <= 1e-6 ? !def_left : x >= thr), not the
// isnan() of the dead kernel (Struct.h:2240) -- a forest converted with dense2sparse must predict exactly
// what predict_on_cpu predicts for the dense original, which is the check the reference's own sparse
// harness makes (BaseTahoeTest.h:836).
//
// Kernels, fastest first (strategy on a sparse handle):
//   sparse_q_kernel    QRING      rank-quantised codes (quantize.hip), 192- / 128-row region tiles, 3 / 2 chains per lane, the
//                                 first 9 levels of each tree as a complete heap in LDS, 32-byte two-level blocks below, ring
//                                 consumer; num_cols <= 256 (K5: 2.5 ms)
//   sparse_top_kernel  TILEBLOCK  64-row float32 tile + the first 512 nodes of each tree (breadth-first, children paired) in
//                                 LDS, ring consumer (K5: 4.9 ms)
//   sparse_kernel      ROWTILE / DIRECT   the 12-byte nodes as given: lane = row, the four waves of a workgroup split the
//                                 trees round-robin, leaf values cross LDS once per round of four trees and are added by
//                                 the row's owner lane in tree order; with TILE the 64 rows sit feature-major in LDS, else
//                                 features come from global memory (K5: 17 / 23 ms)
// Every form adds the leaf values in tree order: float32 sums bit-equal to a sequential CPU sum.  Lanes that have reached
// their leaf idle until the longest path of the wave ends (wave divergence is inherent to irregular trees).
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "qring_internal.h"

struct tahoe_sstate {
    tahoe_sparse_node *nodes = nullptr;  // device copy, as given
    int32_t *trees = nullptr;            // device: root offset per tree
    size_t num_nodes = 0;
    int max_tree_nodes = 0;
    // compact breadth-first copy for sparse_top_kernel (null when a tree has > 65535 reachable nodes or
    // num_cols > 32767): 8-byte nodes, every tree starts on a 16-byte boundary
    uint2 *cnodes = nullptr;    // x = value bits; y = left_idx << 16 | def_left << 15 | fid, left_idx == 0 <=> leaf; root at 1, pairs even
    int32_t *ctrees = nullptr;  // [T + 1] offsets into cnodes (even)
    uint32_t *corig = nullptr;  // compact position -> index relative to the root in the caller's numbering
    // quantised form for sparse_q_kernel (null when num_cols > 256 or a single tree exceeds the code range); the threshold
    // tables, the code workspace and the tree groups live in f->q (qring_internal.h).  Node word = code << 16 | def_left << 15 | fid << 7
    // (bits 6..0 free for flags).
    // qtop [T][1024] u32: the first kSQLevels levels of every tree as a complete heap (early leaves padded down), the form the
    //   walker slots hold: entries 1..511 node words (1-based heap positions, children of i = 2i, 2i+1; bits 1 / 2 of a
    //   last-level word: the left / right child continues below the top), entries 512..1023 what lies below each last-level
    //   branch: a leaf value, or the index (within the tree) of the block where the walk goes on
    // qblocks: 32-byte blocks {n0, n1, n2, flags} {e0..e3}: a node, its two children, and what lies below each of the four
    //   branches: a leaf value or (flags bit j) the index of the next block; a child that is a leaf is a padding word with its
    //   value below both branches.  qblkoff [T + 1]: first block of each tree.
    // qbotpos [T][512], qblkpos [blocks][4]: compact position (corig index) of every entry, for the leaf-index output
    uint32_t *qtop = nullptr;
    uint4 *qblocks = nullptr;
    int32_t *qblkoff = nullptr;
    uint32_t *qbotpos = nullptr;
    uint32_t *qblkpos = nullptr;
};

namespace tahoe {

constexpr int32_t kSFidMask = (int32_t)((1u << 30) - 1u);
constexpr int32_t kSDefLeft = (int32_t)(1u << 30);
constexpr int32_t kSIsLeaf = (int32_t)(1u << 31);

template <bool TILE, bool WRITE_LEAF>
__global__ void __launch_bounds__(kBlock) sparse_kernel(const tahoe_sparse_node *__restrict__ nodes,
                                                        const int32_t *__restrict__ trees, const float *__restrict__ data,
                                                        float *sums, uint32_t *__restrict__ leaf_out,
                                                        const float *sums_in, size_t rows, int cols, int num_trees, float missing,
                                                        int vec4_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *tile = reinterpret_cast<float *>(smem);
    float *vals = reinterpret_cast<float *>(smem + (TILE ? (size_t)cols * kTileRows * sizeof(float) : 0));
    const size_t row0 = (size_t)blockIdx.x * kTileRows;
    const size_t row = row0 + lane;
    const bool row_ok = row < rows;
    const float *xrow = data + (row_ok ? row : row0) * (size_t)cols;
    if (TILE) {
        if (vec4_ok) {
            const float4 *src4 = reinterpret_cast<const float4 *>(xrow);
            for (int f4 = wave; f4 < cols / 4; f4 += kWaves) {
                const float4 v = row_ok ? src4[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
                tile[(4 * f4 + 0) * kTileRows + lane] = v.x;
                tile[(4 * f4 + 1) * kTileRows + lane] = v.y;
                tile[(4 * f4 + 2) * kTileRows + lane] = v.z;
                tile[(4 * f4 + 3) * kTileRows + lane] = v.w;
            }
        } else {
            for (int f = wave; f < cols; f += kWaves) tile[f * kTileRows + lane] = row_ok ? xrow[f] : 0.0f;
        }
        __syncthreads();
    }
    float sum = 0.0f;  // lanes 0..15: row 16*wave + lane of the tile
    if (sums_in && lane < 16 && row0 + 16 * wave + lane < rows) sum = sums_in[row0 + 16 * wave + lane];
    const int rounds = (num_trees + kWaves - 1) / kWaves;
    for (int r = 0; r < rounds; ++r) {
        const int t = r * kWaves + wave;
        float v = 0.0f;
        if (t < num_trees) {
            const tahoe_sparse_node *root = nodes + trees[t];
            uint32_t curr = 0;
            for (;;) {  // create() guarantees left_idx > curr and in range: the walk terminates
                const tahoe_sparse_node n = root[curr];
                if (n.bits & kSIsLeaf) {
                    v = n.val;
                    break;
                }
                const int fid = n.bits & kSFidMask;
                const float x = TILE ? tile[fid * kTileRows + lane] : xrow[fid];
                curr = (uint32_t)n.left_idx + go_right(x, n.val, (n.bits & kSDefLeft) != 0, missing);
            }
            if (WRITE_LEAF) {
                if (row_ok) leaf_out[row * (size_t)num_trees + t] = curr;
            }
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

// Tree tops in LDS.  NW - 1 walker waves share one 64-row float32 tile and split the trees round-robin; each stages
// the first kSTop nodes of its tree (breadth-first order: the top levels) into a private 4 KiB LDS slot with four
// coalesced 16-byte loads per lane, issued before the previous tree is walked and committed after it.  A walk
// reads nodes from the slot while its position is < kSTop and gathers from global memory below.  The gathers are
// what bounds the plain kernel (64 lanes = 64 cache lines per step through the texture path, 12-byte nodes);
// here only the steps below the top pay them, with 8-byte nodes.
// Sums: irregular trees take 4 to 24 steps, so a barrier per round of trees would make every wave wait for the
// deepest tree of the round.  Instead the walkers publish leaf values through an LDS ring (vals, then a ready flag;
// LDS operations of one CU complete in order) and the last wave adds them in tree order, exactly the scheme of
// qring_kernel: walkers wait only for `consumed` (ring capacity), the consumer only for trees whose walkers
// cannot be blocked; every spin is bounded and raises the error flag.
constexpr int kSTop = 512;    // nodes per slot (8 B each)
constexpr int kSRing = 32;    // ring entries (trees)
#ifndef TAHOE_SPARSE_BATCH
#define TAHOE_SPARSE_BATCH 4  // K5: 16 -> 5.34 ms, 8 -> 5.19, 4 -> 5.15
#endif
constexpr int kSBatch = TAHOE_SPARSE_BATCH;   // trees the consumer takes per poll
constexpr int kSSpinLimit = 1 << 22;
template <int NW, bool WRITE_LEAF>
__global__ void __launch_bounds__(NW * 64) sparse_top_kernel(const uint2 *__restrict__ cnodes, const int32_t *__restrict__ ctrees,
                                                             const uint32_t *__restrict__ corig, const float *__restrict__ data,
                                                             float *sums, uint32_t *__restrict__ leaf_out,
                                                             const float *sums_in, size_t rows, int cols, int num_trees, float missing,
                                                             int vec4_ok, int *__restrict__ error_flag)
{
    constexpr int NWALK = NW - 1;
    static_assert(kSRing >= 2 * kSBatch && kSRing > NWALK, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *tile = reinterpret_cast<float *>(smem);
    unsigned char *slots = smem + (size_t)cols * kTileRows * sizeof(float);
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * (kSTop * 8));
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + kSRing * kTileRows);
    uint32_t *consumed = ring_ready + kSRing;
    const size_t row0 = (size_t)blockIdx.x * kTileRows;
    const size_t row = row0 + lane;
    const bool row_ok = row < rows;
    const float *xrow = data + (row_ok ? row : row0) * (size_t)cols;
    if (vec4_ok) {
        const float4 *src4 = reinterpret_cast<const float4 *>(xrow);
        for (int f4 = wave; f4 < cols / 4; f4 += NW) {
            const float4 v = row_ok ? src4[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
            tile[(4 * f4 + 0) * kTileRows + lane] = v.x;
            tile[(4 * f4 + 1) * kTileRows + lane] = v.y;
            tile[(4 * f4 + 2) * kTileRows + lane] = v.z;
            tile[(4 * f4 + 3) * kTileRows + lane] = v.w;
        }
    } else {
        for (int f = wave; f < cols; f += NW) tile[f * kTileRows + lane] = row_ok ? xrow[f] : 0.0f;
    }
    if (threadIdx.x < kSRing) ring_ready[threadIdx.x] = 0u;
    if (threadIdx.x == kSRing) *consumed = 0u;

    if (wave == NWALK) {
        // ================= consumer: lane = row, trees in order =================
        __syncthreads();
        float sum = (sums_in && row_ok) ? sums_in[row] : 0.0f;
        bool dead = false;
        for (int t0 = 0; t0 < num_trees && !dead; t0 += kSBatch) {
            const int nb = min(kSBatch, num_trees - t0);
            int spins = 0;
            for (;;) {
                const bool ok = lane >= nb || lds_flag_load(&ring_ready[(t0 + lane) % kSRing]) == (uint32_t)(t0 + lane + 1);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kSSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();  // the values are read after the flags
            for (int j = 0; j < nb; ++j) sum += ring_vals[((t0 + j) % kSRing) * kTileRows + lane];  // tree order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_flag_store(consumed, (uint32_t)(t0 + nb));
        }
        if (dead && lane == 0) atomicOr(error_flag, 1);
        if (sums && row_ok) sums[row] = sum;
        return;
    }

    // ================= walkers =================
    uint2 *slot = reinterpret_cast<uint2 *>(slots + (size_t)wave * (kSTop * 8));
    // top of tree t: uint4 j * 64 + lane holds nodes 2 * (j * 64 + lane) and the next; clamped loads stay inside
    // the tree's (16-byte padded) node range, slots keep whatever the clamp fetched beyond it (never reached)
    uint4 pf0, pf1, pf2, pf3;  // named registers: an array, even under full unrolling, costs 60 % (measured)
    auto prefetch_top = [&](int t) {
        const int32_t lo = ctrees[t], n = ctrees[t + 1] - lo;  // n is even (padded)
        const uint4 *src = reinterpret_cast<const uint4 *>(cnodes + lo);
        const int last = n / 2 - 1;
        pf0 = src[min(0 * 64 + lane, last)];
        pf1 = src[min(1 * 64 + lane, last)];
        pf2 = src[min(2 * 64 + lane, last)];
        pf3 = src[min(3 * 64 + lane, last)];
    };
    auto commit_top = [&]() {
        uint4 *s = reinterpret_cast<uint4 *>(slot);
        s[0 * 64 + lane] = pf0;
        s[1 * 64 + lane] = pf1;
        s[2 * 64 + lane] = pf2;
        s[3 * 64 + lane] = pf3;
    };
    if (wave < num_trees) prefetch_top(wave);
    __syncthreads();  // the tile and the ring state
    bool dead = false;
    for (int t = wave; t < num_trees && !dead; t += NWALK) {
        commit_top();  // own slot, own wave: LDS operations of one wave complete in order
        if (t + NWALK < num_trees) prefetch_top(t + NWALK);
        const uint2 *root = cnodes + ctrees[t];
        uint32_t curr = 1u;  // the root; position 0 is padding
        uint2 n = slot[1];
        float v;
        for (;;) {  // create() guarantees children after their parent and inside the tree: the walk terminates
            const uint32_t left = n.y >> 16;
            if (left == 0u) {
                v = __uint_as_float(n.x);
                break;
            }
            // the feature value and BOTH children (an aligned pair) are fetched together: one LDS round trip per level
            const float x = tile[(n.y & 0x7fffu) * kTileRows + lane];
            uint4 pr;
            if (left < (uint32_t)kSTop)
                pr = *reinterpret_cast<const uint4 *>(slot + left);
            else
                pr = *reinterpret_cast<const uint4 *>(root + left);
            const uint32_t r = go_right(x, __uint_as_float(n.x), (n.y & 0x8000u) != 0u, missing);
            n = r ? make_uint2(pr.z, pr.w) : make_uint2(pr.x, pr.y);
            curr = left + r;
        }
        if (WRITE_LEAF) {
            if (row_ok) leaf_out[row * (size_t)num_trees + t] = corig[ctrees[t] + curr];
        }
        if (t >= kSRing) {  // ring entry still in use by tree t - kSRing?
            int spins = 0;
            while (lds_flag_load(consumed) < (uint32_t)(t - kSRing + 1)) {
                if (++spins > kSSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const int e = t % kSRing;
        ring_vals[e * kTileRows + lane] = v;
        TAHOE_LDS_RELEASE();  // values before the flag: a wave's LDS operations are performed in issue order
        if (lane == 0) lds_flag_store(&ring_ready[e], (uint32_t)(t + 1));
    }
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// The walk on rank-quantised features (strategy QRING on a sparse handle; num_cols <= 256).  The scheme of qring_kernel's
// region form (qring.hip): the quantise pass (quantize.hip) turns the batch into u16 codes, a tile is K regions of 64 rows x
// [fid][64] u16 (32 KiB each), K = 3 chains per lane (192 rows) walk a tree whose top is staged once into the walker's
// 4 KiB LDS slot -- three independent dependent-read chains per walker wave where the float32 tile kernel has one.
// Top of a tree = its first kSQLevels = 9 levels as a COMPLETE heap (an early leaf is padded down to the last level: both
// children of a padding node carry the value), 4-byte node words code << 16 | def_left << 15 | fid << 7 at 1-based positions, so
// the children of i are the aligned pair (2i, 2i+1): one ds_read_b64 beside the code read, one v_addc per level, no
// per-lane "am I at a leaf yet" state -- the irregular shape costs nothing inside the top.  Entries 512..1023 hold what
// lies below each last-level branch: a leaf value (most lanes end here) or, flagged in the last-level node word, the
// index of a 32-byte BLOCK in global memory: a node, its two children and what lies below their four branches (a value, or
// the next block).  One gather of a block advances a walk by two levels; only the lanes whose path goes on take part (K5:
// 14 of 192 on average, but 4 more levels for the longest), and the first block of tree t is gathered before the top of
// tree t + NWALK is walked and used after it -- a software pipeline, as the dense form does with its bottom blocks.
// Leaf values go through the LDS ring to the consumer wave, which adds them in tree order: the float32 sums are
// bit-identical to a sequential CPU sum.  Tree groups (a feature with more than 32767 distinct thresholds) chain their
// sums like the dense form.
constexpr int kSQLevels = 9;
#ifndef TAHOE_SQ_DEP
#define TAHOE_SQ_DEP 0  // 1: the top walk reads only the chosen child (make SQDEP=1; K5 measured in profiles/r04/tune_dep.txt)
#endif
constexpr bool kSparseQDep = TAHOE_SQ_DEP != 0;
template <int NWALK, bool WRITE_LEAF, int K, int RING>
__global__ void __launch_bounds__((NWALK + 1) * 64)
    sparse_q_kernel(const uint16_t *__restrict__ xq, const uint32_t *__restrict__ qtop, const uint4 *__restrict__ qblocks,
                    const int32_t *__restrict__ qblkoff, const uint32_t *__restrict__ qbotpos, const uint32_t *__restrict__ qblkpos,
                    const int32_t *__restrict__ ctrees, const uint32_t *__restrict__ corig,
                    float *sums, uint32_t *__restrict__ leaf_out, size_t rows, int cols, int tree_lo, int num_trees,
                    int total_trees, const uint32_t *__restrict__ chunk_flags, int *__restrict__ error_flag, const float *sums_in,
                    int cshift, size_t row_begin)
{
    // `rows` is the END of the rows this launch walks, row_begin (a multiple of 384) their start (see qreg_plan)
    constexpr int TR = 64 * K;
    constexpr int NT = (NWALK + 1) * 64;
    constexpr int BATCH = RING >= 2 * kQBatch ? kQBatch : RING / 2;
    constexpr uint32_t NBOT = 1u << kSQLevels;  // entries below the top; the slot holds 2 * NBOT words
    static_assert(BATCH >= 1 && RING >= 2 * BATCH, "ring too small");
    static_assert(2 * NBOT * 4 == kQSlotBytes, "a top fills a walker slot");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *slots = smem + (size_t)K * kRegBytes;
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * kQSlotBytes);
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + RING * TR);
    uint32_t *consumed = ring_ready + RING;
    const size_t row0 = row_begin + (size_t)blockIdx.x * TR;
    if ((uint32_t)reinterpret_cast<uintptr_t>(smem) != 0u) {  // q_xread's v_bfi needs the regions at LDS address 0
        if (tid == 0) atomicOr(error_flag, 2);
        return;
    }
    {   // K consecutive regions of the workspace, each to its 32-KiB-aligned place
        const int n16 = cols * kRegRows * 2 / 16;
        for (int k = 0; k < K; ++k) {
            const uint4 *src = reinterpret_cast<const uint4 *>(xq + ((row0 >> 6) + (size_t)k) * ((size_t)cols * kRegRows));
            uint4 *dst = reinterpret_cast<uint4 *>(smem + (size_t)k * kRegBytes);
            for (int e = tid; e < n16; e += NT) dst[e] = src[e];
        }
    }
    if (tid < RING) ring_ready[tid] = 0u;
    if (tid == RING) *consumed = 0u;

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
        for (int t0 = 0; t0 < num_trees && !dead; t0 += BATCH) {
            const int nb = min(BATCH, num_trees - t0);
            int spins = 0;
            for (;;) {
                const bool ok = lane >= nb || lds_flag_load(&ring_ready[(t0 + lane) % RING]) == (uint32_t)(t0 + lane + 1);
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kSSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();  // the values are read after the flags
            for (int j = 0; j < nb; ++j) {
                const int e = (t0 + j) % RING;
#pragma unroll
                for (int k = 0; k < K; ++k) sum[k] += ring_vals[e * TR + k * 64 + lane];  // tree order
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
    uint32_t *slot = reinterpret_cast<uint32_t *>(slots + (size_t)wave * kQSlotBytes);
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};  // named registers (an indexed array would go to scratch)
    auto prefetch_top = [&](int t) {  // 4 KiB per tree, 16 bytes per lane and load
        const uint4 *g = reinterpret_cast<const uint4 *>(qtop + (size_t)(tree_lo + t) * (2 * NBOT));
        pf0 = g[0 * 64 + lane];
        pf1 = g[1 * 64 + lane];
        pf2 = g[2 * 64 + lane];
        pf3 = g[3 * 64 + lane];
    };
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
    __syncthreads();  // the regions, the ring state and (own wave) the first top are in LDS

    bool dead = false;
    auto run = [&](auto ms_tag) {
        constexpr bool MS = decltype(ms_tag)::value;
        uint32_t pos[K];  // LDS byte address of this lane's row in feature column 0 of region k
#pragma unroll
        for (int k = 0; k < K; ++k) pos[k] = (uint32_t)(k * kRegBytes) + 2u * (uint32_t)qreg_pos(lane);
        // State of the PREVIOUS tree of this walker (t_p): per chain what lies below the top -- a leaf value, or (act) the
        // block where the walk goes on, gathered into a / b.  Its walk below the top is advanced, one block = two levels at
        // a time, BETWEEN the levels of the current tree's top walk: the gathers of the next blocks fly under three LDS
        // levels of the other tree (software pipeline; a block step costs its compute, not its round trip to L2).
        int t_p = -1;
        const uint4 *tb_p = qblocks;
        uint32_t entry_p[K] = {}, act_p[K] = {}, cpos_p[K] = {};
        uint4 a_p[K] = {}, b_p[K] = {};
        auto deep_step = [&]() -> bool {  // false: no lane of the wave was still under way
            uint32_t open = act_p[0];
#pragma unroll
            for (int k = 1; k < K; ++k) open |= act_p[k];
            if (__ballot(open != 0u) == 0ull) return false;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // per chain, and only a chain that still has a lane under way (the long tails are one or two lanes of one
                // chain); a finished lane of such a chain computes on stale registers and keeps its value
                if (__ballot(act_p[k] != 0u) == 0ull) continue;
                const bool c0 = q_go_right<MS, true, false, kCodeMissing, 15>(q_xread<true, true, 7>(nullptr, a_p[k].x, pos[k]), a_p[k].x);
                const uint32_t n1 = c0 ? a_p[k].z : a_p[k].y;
                const bool c1 = q_go_right<MS, true, false, kCodeMissing, 15>(q_xread<true, true, 7>(nullptr, n1, pos[k]), n1);
                const uint32_t lo = c0 ? b_p[k].z : b_p[k].x, hi = c0 ? b_p[k].w : b_p[k].y;
                const uint32_t j = (c0 ? 2u : 0u) + (c1 ? 1u : 0u);
                const bool on = act_p[k] != 0u;
                const bool cont = ((a_p[k].w >> j) & 1u) != 0u;
                if (WRITE_LEAF) {
                    if (on && !cont) cpos_p[k] = qblkpos[4 * ((size_t)qblkoff[tree_lo + t_p] + entry_p[k]) + j];
                }
                entry_p[k] = on ? (c1 ? hi : lo) : entry_p[k];  // the next block, or the leaf value
                act_p[k] = (on && cont) ? 1u : 0u;
                if (act_p[k] != 0u) {  // the tree's blocks from SGPRs, the block's byte offset (32 bits) from a VGPR
                    const uint4 *bp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(tb_p) + (entry_p[k] << 5));
                    a_p[k] = bp[0];
                    b_p[k] = bp[1];
                }
            }
            return true;
        };
        auto publish = [&]() {  // leaf values of tree t_p to the consumer (and its leaf indices out)
            const int t = t_p;
            if (WRITE_LEAF) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const size_t row = row0 + k * 64 + lane;
                    if (row < rows) leaf_out[row * (size_t)total_trees + tree_lo + t] = corig[ctrees[tree_lo + t] + cpos_p[k]];
                }
            }
            if (t >= RING) {  // ring entry still in use by tree t - RING?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t - RING + 1)) {
                    if (++spins > kSSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            const int e = t % RING;
#pragma unroll
            for (int k = 0; k < K; ++k) ring_vals[e * TR + k * 64 + lane] = __uint_as_float(entry_p[k]);
            TAHOE_LDS_RELEASE();  // values before the flag: a wave's LDS operations are performed in issue order
            if (lane == 0) lds_flag_store(&ring_ready[e], (uint32_t)(t + 1));
        };
        for (int t = wave; t < num_trees && !dead; t += NWALK) {
            const bool more = t + NWALK < num_trees;
            if (more) prefetch_top(t + NWALK);
            uint32_t i[K], node[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                i[k] = 1u;
                node[k] = slot[1];
            }
            if (kSparseQDep) {  // only the chosen child, read after the compare (4 VALU + 2 LDS per chain-level; qring.hip, DEP)
                const uint32_t slot_a = (uint32_t)reinterpret_cast<uintptr_t>(slot);
                for (int l = 0; l < kSQLevels - 1; ++l) {
                    uint32_t xc[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) xc[k] = q_xread<true, true, 7>(nullptr, node[k], pos[k]);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        i[k] = q_descend(i[k], q_right_mask<MS, true, false, kCodeMissing, 15>(xc[k], node[k]));
                        node[k] = *reinterpret_cast<const uint32_t __attribute__((address_space(3))) *>(slot_a + 4u * i[k]);
                    }
                    if (t_p >= 0 && (l == 3 || l == 6)) (void)deep_step();  // the previous tree, below its top
                }
            } else {
                {   // level 0: every chain is at the root -- one read of its child pair serves them all
                    const uint2 pr0 = *reinterpret_cast<const uint2 *>(&slot[2]);
                    uint32_t xc0[K];
    #pragma unroll
                    for (int k = 0; k < K; ++k) xc0[k] = q_xread<true, true, 7>(nullptr, node[k], pos[k]);
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint64_t cm = q_right_mask<MS, true, false, kCodeMissing, 15>(xc0[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr0.y : pr0.x;
                    }
                }
                for (int l = 1; l < kSQLevels - 1; ++l) {  // both children come with one ds_read_b64 beside the code read
                    uint32_t xc[K];
                    uint2 pr[K];
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        xc[k] = q_xread<true, true, 7>(nullptr, node[k], pos[k]);
                        pr[k] = *reinterpret_cast<const uint2 *>(&slot[2 * i[k]]);
                    }
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint64_t cm = q_right_mask<MS, true, false, kCodeMissing, 15>(xc[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr[k].y : pr[k].x;
                    }
                    if (t_p >= 0 && (l == 3 || l == 6)) (void)deep_step();  // the previous tree, below its top
                }
            }
            uint32_t entry[K];  // what lies below the last level's branch: a leaf value, or the block where the walk goes on
            uint32_t act[K];    // 1: this chain's walk goes on below the top
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t xc = q_xread<true, true, 7>(nullptr, node[k], pos[k]);
                const uint64_t cm = q_right_mask<MS, true, false, kCodeMissing, 15>(xc, node[k]);
                i[k] = q_descend(i[k], cm);  // NBOT .. 2 NBOT - 1
                act[k] = (__builtin_amdgcn_inverse_ballot_w64(cm) ? node[k] >> 2 : node[k] >> 1) & 1u;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) entry[k] = slot[i[k]];
            uint32_t cpos[K];  // compact position of the leaf (leaf-index output only)
            if (WRITE_LEAF) {
#pragma unroll
                for (int k = 0; k < K; ++k) cpos[k] = qbotpos[(size_t)(tree_lo + t) * NBOT + (i[k] - NBOT)];
            }
            if (t_p >= 0) {  // whatever is left of the previous tree's walk, then hand it over
                while (deep_step()) {
                }
                publish();
            }
            t_p = t;
            tb_p = qblocks + 2 * (size_t)qblkoff[tree_lo + t];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                entry_p[k] = entry[k];
                act_p[k] = act[k];
                if (WRITE_LEAF) cpos_p[k] = cpos[k];
                if (act[k] != 0u) {  // this tree's first blocks: in flight while the next top is walked
                    const uint4 *bp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(tb_p) + (entry[k] << 5));
                    a_p[k] = bp[0];
                    b_p[k] = bp[1];
                }
            }
            if (more) commit_top();  // this wave's reads of its slot are done (in-order LDS)
        }
        if (t_p >= 0 && !dead) {
            while (deep_step()) {
            }
            publish();
        }
    };
    // chunk_flags[c] != 0 <=> the quantise pass met a missing value in rows [c, c+1) << cshift; a tile can straddle two chunks
    if ((chunk_flags[row0 >> cshift] | chunk_flags[(min(rows, row0 + TR) - 1) >> cshift]) != 0)
        run(std::true_type{});
    else
        run(std::false_type{});
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

bool sparse_q_available(const tahoe_forest *f) { return f->sp && f->sp->qtop && f->q; }

template <int NWALK, int K, int RING>
static void sparse_q_launch_form(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                                 size_t row_begin, size_t rows_end, hipStream_t stream, int cshift)
{
    const tahoe_sstate *sp = f->sp;
    const unsigned grid = (unsigned)((rows_end - row_begin + 64 * K - 1) / (64 * K));
    if (grid == 0) return;
    const int lds = (int)qreg_lds_for(K, NWALK, RING);
    if (leaf_out)
        hipLaunchKernelGGL((sparse_q_kernel<NWALK, true, K, RING>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, f->q->xq, sp->qtop, sp->qblocks,
                           sp->qblkoff, sp->qbotpos, sp->qblkpos, sp->ctrees, sp->corig, sums, leaf_out, rows_end, f->p.num_cols, g.tree_lo, g.num_trees, f->p.num_trees,
                           f->q->chunk_flags, f->error_flag, sums_in, cshift, row_begin);
    else
        hipLaunchKernelGGL((sparse_q_kernel<NWALK, false, K, RING>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, f->q->xq, sp->qtop, sp->qblocks,
                           sp->qblkoff, sp->qbotpos, sp->qblkpos, sp->ctrees, sp->corig, sums, leaf_out, rows_end, f->p.num_cols, g.tree_lo, g.num_trees, f->p.num_trees,
                           f->q->chunk_flags, f->error_flag, sums_in, cshift, row_begin);
}

// quantise + walk per tree group, in stream order; the tile plan of the dense region form (qreg_plan)
static tahoe_status sparse_q_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                                    hipStream_t stream, const float *sums_in0)
{
    if (!sparse_q_available(f))
        return fail(TAHOE_ERR_UNSUPPORTED, "sparse QRING needs num_cols <= 256, trees of <= 65535 nodes and <= %d distinct thresholds "
                                           "per feature within one tree", kQMaxTable);
    tahoe_qstate *q = f->q;
    const tahoe_status rs = qring_reserve(f, rows);  // no-op unless this batch is larger than any before
    if (rs != TAHOE_OK) return rs;
    size_t rows3 = 0;
    int chains = 2;
    qreg_plan(rows, f->num_cus, f->knob_qring_chains, &rows3, &chains, 161);  // K5: 0.555 ms per wave of 192-row tiles, 0.344 of 128
    bool first = true;
    for (const tahoe_qgroup &g : q->groups) {
        TAHOE_HIP_TRY(hipMemsetAsync(q->chunk_flags, 0, q->n_chunk_flags * sizeof(uint32_t), stream));
        int cshift = 0;
        const tahoe_status qs = quantize_launch(f, g, data, rows, 6, 1, stream, &cshift);
        if (qs != TAHOE_OK) return qs;
        const float *sums_in = first ? sums_in0 : sums;  // later groups continue the running float32 sums
        if (rows3 > 0) sparse_q_launch_form<kReg3Walkers, 3, kReg3Ring>(f, g, sums, sums_in, leaf_out, 0, rows3, stream, cshift);
        if (chains == 3)
            sparse_q_launch_form<kReg3Walkers, 3, kReg3Ring>(f, g, sums, sums_in, leaf_out, rows3, rows, stream, cshift);
        else
            sparse_q_launch_form<15, 2, kQRing>(f, g, sums, sums_in, leaf_out, rows3, rows, stream, cshift);
        TAHOE_HIP_TRY(hipGetLastError());
        first = false;
    }
    return TAHOE_OK;
}

static long long sparse_top_lds(const tahoe_forest *f, int nw)
{
    return (long long)f->p.num_cols * kTileRows * 4 + (long long)(nw - 1) * kSTop * 8 + (long long)kSRing * kTileRows * 4 +
           (kSRing + 1) * 4LL;
}

// waves per workgroup of sparse_top_kernel for this handle; 0 = that form is unavailable
int sparse_top_waves(const tahoe_forest *f)
{
    if (!f->sp || !f->sp->cnodes || f->p.num_cols < 1) return 0;
    if (sparse_top_lds(f, 16) <= f->lds_limit) return 16;
    if (sparse_top_lds(f, 8) <= f->lds_limit) return 8;
    return 0;
}

static long long sparse_lds(const tahoe_forest *f, bool tile)
{
    return (tile ? (long long)f->p.num_cols * kTileRows * 4 : 0) + 2LL * kWaves * kTileRows * 4;
}

bool sparse_tile_fits(const tahoe_forest *f) { return f->p.num_cols >= 1 && sparse_lds(f, true) <= f->lds_limit; }

tahoe_status sparse_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                           hipStream_t stream, int strategy, const float *sums_in)
{
    const tahoe_sstate *sp = f->sp;
    const unsigned grid = (unsigned)((rows + kTileRows - 1) / kTileRows);
    const int vec4_ok = (f->p.num_cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0);
    if (strategy == TAHOE_STRATEGY_QRING) return sparse_q_launch(f, sums, leaf_out, data, rows, stream, sums_in);
    if (strategy == TAHOE_STRATEGY_TILEBLOCK) {
        const int nw = sparse_top_waves(f);
        if (nw == 0) return fail(TAHOE_ERR_UNSUPPORTED, "sparse TILEBLOCK: the compact form or its LDS tile is unavailable");
        const int lds = (int)sparse_top_lds(f, nw);
#define TAHOE_SPARSE_TOP(NW_, LEAF_)                                                                                     \
    hipLaunchKernelGGL((sparse_top_kernel<NW_, LEAF_>), dim3(grid), dim3(NW_ * 64), lds, stream, sp->cnodes, sp->ctrees, \
                       sp->corig, data, sums, leaf_out, sums_in, rows, f->p.num_cols, f->p.num_trees, f->p.missing, vec4_ok, f->error_flag)
        if (nw == 16) {
            if (leaf_out)
                TAHOE_SPARSE_TOP(16, true);
            else
                TAHOE_SPARSE_TOP(16, false);
        } else {
            if (leaf_out)
                TAHOE_SPARSE_TOP(8, true);
            else
                TAHOE_SPARSE_TOP(8, false);
        }
#undef TAHOE_SPARSE_TOP
        TAHOE_HIP_TRY(hipGetLastError());
        return TAHOE_OK;
    }
    const bool tile = strategy == TAHOE_STRATEGY_ROWTILE;
    const int lds = (int)sparse_lds(f, tile);
#define TAHOE_SPARSE_LAUNCH(TILE_, LEAF_)                                                                            \
    hipLaunchKernelGGL((sparse_kernel<TILE_, LEAF_>), dim3(grid), dim3(kBlock), lds, stream, sp->nodes, sp->trees, data, \
                       sums, leaf_out, sums_in, rows, f->p.num_cols, f->p.num_trees, f->p.missing, vec4_ok)
    if (tile) {
        if (leaf_out)
            TAHOE_SPARSE_LAUNCH(true, true);
        else
            TAHOE_SPARSE_LAUNCH(true, false);
    } else {
        if (leaf_out)
            TAHOE_SPARSE_LAUNCH(false, true);
        else
            TAHOE_SPARSE_LAUNCH(false, false);
    }
#undef TAHOE_SPARSE_LAUNCH
    TAHOE_HIP_TRY(hipGetLastError());
    return TAHOE_OK;
}

void sparse_destroy(tahoe_forest *f)
{
    if (!f->sp) return;
    if (f->sp->nodes) (void)hipFree(f->sp->nodes);
    if (f->sp->trees) (void)hipFree(f->sp->trees);
    if (f->sp->cnodes) (void)hipFree(f->sp->cnodes);
    if (f->sp->ctrees) (void)hipFree(f->sp->ctrees);
    if (f->sp->corig) (void)hipFree(f->sp->corig);
    for (void *p : {(void *)f->sp->qtop, (void *)f->sp->qblocks, (void *)f->sp->qblkoff, (void *)f->sp->qbotpos, (void *)f->sp->qblkpos})
        if (p) (void)hipFree(p);
    delete f->sp;
    f->sp = nullptr;
}

// Quantised copy of the compact form (sparse_q_kernel) + the quantiser's tables, per tree group.  `cn` / `ct` = the compact
// nodes and tree offsets just built.  Leaves the strategy unavailable (TAHOE_OK, sp->qnodes null) when num_cols > 256, the
// region tile does not fit LDS, or a single tree uses more than kQMaxTable distinct thresholds on one feature.
static tahoe_status sparse_q_build(tahoe_forest *f, const std::vector<uint2> &cn, const std::vector<int32_t> &ct)
{
    tahoe_sstate *sp = f->sp;
    const int cols = f->p.num_cols;
    const size_t T = (size_t)f->p.num_trees;
    if (const char *e = getenv("TAHOE_SPARSE_QRING"))  // experiments: 0 keeps the float32 kernels only
        if (atoi(e) == 0) return TAHOE_OK;
    if (cols < 1 || cols > 256 || T == 0 || qreg_lds_for(3, kReg3Walkers, kReg3Ring) > f->lds_limit ||
        qreg_lds_for(2, 15, kQRing) > f->lds_limit)
        return TAHOE_OK;
    tahoe_qstate *q = new (std::nothrow) tahoe_qstate();
    if (!q) return fail(TAHOE_ERR_NO_MEMORY, "sparse_q_build");
    f->q = q;
    q->narrow = q->reg = q->sparse = true;  // region workspace (qring_reserve), region node words
    if (const char *k = getenv("TAHOE_QRING_CHAINS")) f->knob_qring_chains = atoi(k);  // 2 / 3: force the tile form
    auto is_inner = [](const uint2 &c) { return (c.y >> 16) != 0u; };
    auto thr_of = [](const uint2 &c) {
        float v;
        memcpy(&v, &c.x, 4);
        return v;
    };
    // tree groups: G as small as the busiest feature allows (first guess from the DISTINCT thresholds per feature of the whole
    // forest -- node counts would cut a forest of histogram-trained trees into groups for nothing --, then grow until every group fits)
    size_t G = 1;
    {
        std::vector<std::vector<float>> all((size_t)cols);
        for (const uint2 &c : cn)
            if (is_inner(c) && !std::isnan(thr_of(c))) all[c.y & 0x7fffu].push_back(thr_of(c));
        std::vector<size_t> distinct((size_t)cols, 0);
        parallel_for((size_t)cols, 4, [&](size_t c_lo, size_t c_hi) {
            for (size_t c = c_lo; c < c_hi; ++c) {
                auto &v = all[c];
                std::sort(v.begin(), v.end());
                distinct[c] = (size_t)(std::unique(v.begin(), v.end()) - v.begin());
                std::vector<float>().swap(v);
            }
        });
        const size_t most = *std::max_element(distinct.begin(), distinct.end());
        if (most > (size_t)kQMaxTable) G = (most + kQMaxTable - 1) / kQMaxTable;
    }
    std::vector<uint2> qn(cn.size());
    const size_t bytes_before = f->device_bytes;
    auto drop_groups = [&]() {
        for (tahoe_qgroup &g : q->groups) quantize_free_tables(g);
        q->groups.clear();
        f->device_bytes = bytes_before;
    };
    for (;;) {
        if (G > T) G = T;
        bool fits = true;
        int worst = 0;
        for (size_t k = 0; k < G && fits; ++k) {
            const size_t lo = T * k / G, hi = T * (k + 1) / G;
            const size_t n_lo = (size_t)ct[lo], n_hi = (size_t)ct[hi];
            std::vector<std::vector<float>> tab((size_t)cols);
            for (size_t i = n_lo; i < n_hi; ++i)
                if (is_inner(cn[i]) && !std::isnan(thr_of(cn[i]))) tab[cn[i].y & 0x7fffu].push_back(thr_of(cn[i]));
            parallel_for((size_t)cols, 4, [&tab](size_t c_lo, size_t c_hi) {
                for (size_t c = c_lo; c < c_hi; ++c) {
                    auto &v = tab[c];
                    std::sort(v.begin(), v.end());                      // float order; -0.0f and 0.0f compare equal
                    v.erase(std::unique(v.begin(), v.end()), v.end());  // ... and collapse into one entry
                }
            });
            int max_count = 0;
            for (int c = 0; c < cols; ++c) max_count = std::max(max_count, (int)tab[c].size());
            if (max_count > kQMaxTable) {
                fits = false;
                worst = max_count;
                break;
            }
            parallel_for(n_hi - n_lo, 1 << 16, [&](size_t a, size_t b) {
                for (size_t i = n_lo + a; i < n_lo + b; ++i) {
                    const uint2 c = cn[i];
                    if (!is_inner(c)) {
                        qn[i] = make_uint2(c.x, 0u);  // leaf (and padding): the value, no children
                        continue;
                    }
                    const uint32_t fid = c.y & 0x7fffu, dl = (c.y >> 15) & 1u;
                    const float thr = thr_of(c);
                    uint32_t code = 0xFFFFu;  // x >= NaN is never true; a missing x still follows def_left
                    if (!std::isnan(thr)) {
                        const auto &v = tab[fid];
                        code = (uint32_t)(std::lower_bound(v.begin(), v.end(), thr) - v.begin()) + 1u;
                    }
                    qn[i] = make_uint2((code << 16) | (dl << 15) | (fid << 7), c.y >> 16);  // def_left in bit 15 (q_right_mask, DLB)
                }
            });
            tahoe_qgroup g;
            g.tree_lo = (int)lo;
            g.num_trees = (int)(hi - lo);
            const tahoe_status bs = quantize_build_tables(f, tab, g);
            q->groups.push_back(g);  // owned by the handle from here (freed by qring_destroy)
            if (bs != TAHOE_OK) return bs;
        }
        if (fits) break;
        drop_groups();
        if (G == T) {  // a single tree exceeds the code range: the strategy is unavailable
            qring_destroy(f);
            return TAHOE_OK;
        }
        G = std::max(G + 1, (size_t)((double)G * worst / kQMaxTable + 0.999));
    }
    // ---- the tops (first kSQLevels levels of every tree as a complete heap) and the blocks below them (see sparse_q_kernel) ----
    constexpr uint32_t NBOT = 1u << kSQLevels;
    std::vector<uint32_t> h_top(T * 2 * NBOT, 0u), h_botpos(T * NBOT, 0u);
    std::vector<std::vector<uint4>> tree_blocks(T);      // two uint4 per block
    std::vector<std::vector<uint32_t>> tree_blkpos(T);   // four compact positions per block
    parallel_for(T, 8, [&](size_t t_lo, size_t t_hi) {
        struct Item {
            uint32_t h, p;  // heap position (1-based) or block index; compact position inside the tree
        };
        std::vector<Item> stack, work;
        for (size_t t = t_lo; t < t_hi; ++t) {
            const uint2 *tn = &qn[(size_t)ct[t]];
            uint32_t *top = &h_top[t * 2 * NBOT], *bot = &h_botpos[t * NBOT];
            std::vector<uint4> &blk = tree_blocks[t];
            std::vector<uint32_t> &bpos = tree_blkpos[t];
            work.clear();
            auto new_block = [&](uint32_t p) {  // block whose first node is the inner node at compact position p
                const uint32_t idx = (uint32_t)(blk.size() / 2);
                blk.resize(blk.size() + 2);
                bpos.resize(bpos.size() + 4);
                work.push_back(Item{idx, p});
                return idx;
            };
            stack.assign(1, Item{1u, 1u});  // the root sits at compact position 1
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                const uint2 c = tn[it.p];
                if (it.h >= NBOT) {  // below the last level of the top
                    bot[it.h - NBOT] = it.p;
                    if (c.y == 0u)
                        top[it.h] = c.x;  // a leaf: its value
                    else {
                        top[it.h] = new_block(it.p);  // the walk goes on; flagged in the parent's word (bit 1: left, 2: right)
                        top[it.h >> 1] |= (it.h & 1u) ? 4u : 2u;
                    }
                } else if (c.y == 0u) {  // early leaf: padding nodes (word 0) down to the last level, the value below all of them
                    for (uint32_t lo = it.h, n = 1u; lo < 2 * NBOT; lo *= 2u, n *= 2u)
                        for (uint32_t h = lo; h < lo + n; ++h) {
                            top[h] = h >= NBOT ? c.x : 0u;
                            if (h >= NBOT) bot[h - NBOT] = it.p;
                        }
                } else {
                    top[it.h] = c.x;
                    stack.push_back(Item{2u * it.h + 1u, c.y + 1u});
                    stack.push_back(Item{2u * it.h, c.y});
                }
            }
            for (size_t w = 0; w < work.size(); ++w) {  // grows while it runs: a block's branches that go on get blocks too
                const Item it = work[w];
                const uint2 n0 = tn[it.p];
                uint32_t word[2] = {0u, 0u}, e[4], ep[4], flags = 0u;
                for (uint32_t sd = 0; sd < 2; ++sd) {
                    const uint32_t cp = n0.y + sd;
                    const uint2 c = tn[cp];
                    if (c.y == 0u) {  // the child is a leaf: padding word, its value below both branches
                        e[2 * sd] = e[2 * sd + 1] = c.x;
                        ep[2 * sd] = ep[2 * sd + 1] = cp;
                        continue;
                    }
                    word[sd] = c.x;
                    for (uint32_t g = 0; g < 2; ++g) {
                        const uint32_t gp = c.y + g;
                        const uint2 gc = tn[gp];
                        ep[2 * sd + g] = gp;
                        if (gc.y == 0u)
                            e[2 * sd + g] = gc.x;
                        else {
                            e[2 * sd + g] = new_block(gp);
                            flags |= 1u << (2 * sd + g);
                        }
                    }
                }
                blk[2 * (size_t)it.h] = make_uint4(n0.x, word[0], word[1], flags);
                blk[2 * (size_t)it.h + 1] = make_uint4(e[0], e[1], e[2], e[3]);
                for (int j = 0; j < 4; ++j) bpos[4 * (size_t)it.h + j] = ep[j];
            }
        }
    });
    std::vector<int32_t> h_blkoff(T + 1, 0);
    {
        size_t total = 0;
        for (size_t t = 0; t < T; ++t) {
            h_blkoff[t] = (int32_t)total;
            total += tree_blocks[t].size() / 2;
            if (total > 0x3fffffffu) {  // block indices stay 32-bit with room to spare
                qring_destroy(f);
                return TAHOE_OK;
            }
        }
        h_blkoff[T] = (int32_t)total;
    }
    std::vector<uint4> h_blocks((size_t)h_blkoff[T] * 2);
    std::vector<uint32_t> h_blkpos((size_t)h_blkoff[T] * 4);
    parallel_for(T, 64, [&](size_t t_lo, size_t t_hi) {
        for (size_t t = t_lo; t < t_hi; ++t) {
            std::copy(tree_blocks[t].begin(), tree_blocks[t].end(), h_blocks.begin() + (size_t)h_blkoff[t] * 2);
            std::copy(tree_blkpos[t].begin(), tree_blkpos[t].end(), h_blkpos.begin() + (size_t)h_blkoff[t] * 4);
        }
    });
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "sparse_q_build: %s failed: %s", what, hipGetErrorString(e)); };
    if ((e = q_upload(&sp->qtop, h_top.data(), h_top.size(), &f->device_bytes)) != hipSuccess) return bad("qtop");
    if ((e = q_upload(&sp->qbotpos, h_botpos.data(), h_botpos.size(), &f->device_bytes)) != hipSuccess) return bad("qbotpos");
    if ((e = q_upload(&sp->qblocks, h_blocks.data(), h_blocks.size(), &f->device_bytes)) != hipSuccess) return bad("qblocks");
    if ((e = q_upload(&sp->qblkpos, h_blkpos.data(), h_blkpos.size(), &f->device_bytes)) != hipSuccess) return bad("qblkpos");
    if ((e = q_upload(&sp->qblkoff, h_blkoff.data(), h_blkoff.size(), &f->device_bytes)) != hipSuccess) return bad("qblkoff");
    for (const void *k : {(const void *)&sparse_q_kernel<kReg3Walkers, false, 3, kReg3Ring>, (const void *)&sparse_q_kernel<kReg3Walkers, true, 3, kReg3Ring>,
                          (const void *)&sparse_q_kernel<15, false, 2, kQRing>, (const void *)&sparse_q_kernel<15, true, 2, kQRing>})
        if ((e = allow_max_lds(k, f->lds_limit)) != hipSuccess)
            return fail(TAHOE_ERR_HIP, "sparse_q_build: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    if ((e = quantize_allow_lds(f)) != hipSuccess)
        return fail(TAHOE_ERR_HIP, "sparse_q_build: hipFuncSetAttribute(quantise kernels) failed: %s", hipGetErrorString(e));
    return TAHOE_OK;
}

}  // namespace tahoe

using namespace tahoe;

extern "C" {

tahoe_status tahoe_sparse_forest_create(tahoe_forest **out, const int32_t *trees, const tahoe_sparse_node *nodes,
                                        const tahoe_forest_params *p)
{
    if (!out || !p) return fail(TAHOE_ERR_INVALID_ARG, "tahoe_sparse_forest_create: null argument");
    *out = nullptr;
    // check_params(params, dense = false), BaseTahoeTest.h:490-516
    if (p->num_nodes < 0) return fail(TAHOE_ERR_INVALID_ARG, "num_nodes must be non-negative for sparse forests");
    if (p->algo != TAHOE_ALGO_NAIVE) return fail(TAHOE_ERR_INVALID_ARG, "only NAIVE algorithm is supported for sparse forests");
    if (p->num_trees < 0) return fail(TAHOE_ERR_INVALID_ARG, "num_trees must be non-negative");
    if (p->num_cols < 0) return fail(TAHOE_ERR_INVALID_ARG, "num_cols must be non-negative");
    if ((p->output & ~(TAHOE_OUT_AVG | TAHOE_OUT_SIGMOID | TAHOE_OUT_THRESHOLD)) != 0)
        return fail(TAHOE_ERR_INVALID_ARG, "output should be a combination of RAW, AVG, SIGMOID and THRESHOLD");
    if (p->num_trees > 0 && (!trees || !nodes)) return fail(TAHOE_ERR_INVALID_ARG, "trees / nodes is null");
    // Structure check (the reference trusts its input): roots ascending, every child pair inside its tree
    // and after its parent (which also rules out cycles, so device walks terminate), fid < num_cols.
    int max_tree_nodes = 0;
    for (int t = 0; t < p->num_trees; ++t) {
        const long long lo = trees[t], hi = (t + 1 < p->num_trees) ? trees[t + 1] : p->num_nodes;
        if (lo < 0 || hi <= lo || hi > p->num_nodes)
            return fail(TAHOE_ERR_INVALID_FOREST, "tree %d: root offsets must be ascending and inside [0, num_nodes)", t);
        max_tree_nodes = std::max(max_tree_nodes, (int)(hi - lo));
        for (long long i = 0; i < hi - lo; ++i) {
            const tahoe_sparse_node &n = nodes[lo + i];
            if (n.bits & kSIsLeaf) continue;
            if (n.left_idx <= i || (long long)n.left_idx + 1 >= hi - lo)
                return fail(TAHOE_ERR_INVALID_FOREST, "tree %d node %lld: children %d, %d are not after the node and inside the tree",
                            t, i, n.left_idx, n.left_idx + 1);
            if ((n.bits & kSFidMask) >= p->num_cols)
                return fail(TAHOE_ERR_INVALID_FOREST, "tree %d node %lld: fid %d >= num_cols %d", t, i, n.bits & kSFidMask,
                            p->num_cols);
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(TAHOE_ERR_NO_DEVICE, "no HIP device is visible; libtahoe_amd has no CPU path");
    int dev = 0;
    TAHOE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    TAHOE_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    tahoe_forest *f = new (std::nothrow) tahoe_forest();
    tahoe_sstate *sp = new (std::nothrow) tahoe_sstate();
    if (!f || !sp) {
        delete f;
        delete sp;
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_sparse_forest_create");
    }
    f->sp = sp;
    f->p = *p;
    f->depth = 0;  // a placeholder value, as in sparse_forest::init (Struct.h:2332)
    f->device = dev;
    f->num_cus = prop.multiProcessorCount;
    f->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor > 0 ? (int)prop.maxSharedMemoryPerMultiProcessor
                                                                   : (int)prop.sharedMemPerBlock;
    f->bits_bytes = 4;
    sp->num_nodes = (size_t)p->num_nodes;
    sp->max_tree_nodes = max_tree_nodes;
    auto bail = [&](hipError_t e, const char *what) {
        tahoe_forest_destroy(f);
        return fail(TAHOE_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    const size_t nbytes = std::max<size_t>(sp->num_nodes, 1) * sizeof(tahoe_sparse_node);
    const size_t tbytes = std::max<size_t>((size_t)p->num_trees, 1) * sizeof(int32_t);
    if ((e = hipMalloc(reinterpret_cast<void **>(&sp->nodes), nbytes)) != hipSuccess) return bail(e, "hipMalloc(nodes)");
    if ((e = hipMalloc(reinterpret_cast<void **>(&sp->trees), tbytes)) != hipSuccess) return bail(e, "hipMalloc(trees)");
    f->device_bytes = nbytes + tbytes;
    if (sp->num_nodes && (e = hipMemcpy(sp->nodes, nodes, sp->num_nodes * sizeof(tahoe_sparse_node), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(nodes)");
    if (p->num_trees && (e = hipMemcpy(sp->trees, trees, (size_t)p->num_trees * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(trees)");
    if ((e = hipMalloc(reinterpret_cast<void **>(&f->error_flag), sizeof(int))) != hipSuccess) return bail(e, "hipMalloc(error_flag)");
    if ((e = hipMemset(f->error_flag, 0, sizeof(int))) != hipSuccess) return bail(e, "hipMemset(error_flag)");
    if (sparse_tile_fits(f)) {
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&sparse_kernel<true, false>), f->lds_limit)) != hipSuccess)
            return bail(e, "hipFuncSetAttribute(sparse)");
        if ((e = allow_max_lds(reinterpret_cast<const void *>(&sparse_kernel<true, true>), f->lds_limit)) != hipSuccess)
            return bail(e, "hipFuncSetAttribute(sparse)");
    }
    // ---- compact breadth-first copy (sparse_top_kernel) ----
    if (p->num_cols <= 32767 && p->num_trees > 0) {
        std::vector<uint2> cn;
        std::vector<uint32_t> orig;
        std::vector<int32_t> ct((size_t)p->num_trees + 1, 0);
        std::vector<uint32_t> order, newpos;
        bool ok = true;
        for (int t = 0; t < p->num_trees && ok; ++t) {
            const long long lo = trees[t], hi = (t + 1 < p->num_trees) ? trees[t + 1] : p->num_nodes;
            const tahoe_sparse_node *tn = nodes + lo;
            // breadth-first order of the reachable nodes; a child pair stays adjacent
            order.assign(1, 0u);
            for (size_t q = 0; q < order.size() && order.size() <= 65535; ++q) {
                const tahoe_sparse_node &n = tn[order[q]];
                if (n.bits & kSIsLeaf) continue;
                order.push_back((uint32_t)n.left_idx);
                order.push_back((uint32_t)n.left_idx + 1u);
            }
            if (order.size() > 65535) {
                ok = false;
                break;
            }
            // position 0 of a tree is padding and the root sits at 1, so that every child pair (appended two at a time)
            // starts at an even position: one aligned 16-byte read fetches both children
            newpos.assign((size_t)(hi - lo), 0u);
            for (size_t q = 0; q < order.size(); ++q) newpos[order[q]] = (uint32_t)q + 1u;
            ct[(size_t)t] = (int32_t)cn.size();
            cn.push_back(make_uint2(0u, 0u));
            orig.push_back(0u);
            for (size_t q = 0; q < order.size(); ++q) {
                const tahoe_sparse_node &n = tn[order[q]];
                uint2 c;
                memcpy(&c.x, &n.val, 4);
                if (n.bits & kSIsLeaf)
                    c.y = 0u;
                else  // the left child of any node sits at a position >= 2: 0 marks a leaf
                    c.y = (newpos[(size_t)n.left_idx] << 16) | ((n.bits & kSDefLeft) ? 0x8000u : 0u) | (uint32_t)(n.bits & kSFidMask);
                cn.push_back(c);
                orig.push_back(order[q]);
            }
            if (cn.size() & 1) {  // next tree starts on a 16-byte boundary
                cn.push_back(make_uint2(0u, 0u));
                orig.push_back(0u);
            }
            if (cn.size() > 0x7fffffffu) ok = false;
        }
        if (ok) {
            ct[(size_t)p->num_trees] = (int32_t)cn.size();
            if ((e = hipMalloc(reinterpret_cast<void **>(&sp->cnodes), cn.size() * sizeof(uint2))) != hipSuccess) return bail(e, "hipMalloc(cnodes)");
            if ((e = hipMalloc(reinterpret_cast<void **>(&sp->ctrees), ct.size() * sizeof(int32_t))) != hipSuccess) return bail(e, "hipMalloc(ctrees)");
            if ((e = hipMalloc(reinterpret_cast<void **>(&sp->corig), orig.size() * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(corig)");
            if ((e = hipMemcpy(sp->cnodes, cn.data(), cn.size() * sizeof(uint2), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(cnodes)");
            if ((e = hipMemcpy(sp->ctrees, ct.data(), ct.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(ctrees)");
            if ((e = hipMemcpy(sp->corig, orig.data(), orig.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(corig)");
            f->device_bytes += cn.size() * sizeof(uint2) + ct.size() * sizeof(int32_t) + orig.size() * sizeof(uint32_t);
            for (const void *k : {(const void *)&sparse_top_kernel<16, false>, (const void *)&sparse_top_kernel<16, true>,
                                  (const void *)&sparse_top_kernel<8, false>, (const void *)&sparse_top_kernel<8, true>})
                if ((e = allow_max_lds(k, f->lds_limit)) != hipSuccess) return bail(e, "hipFuncSetAttribute(sparse_top)");
            const tahoe_status qs = sparse_q_build(f, cn, ct);
            if (qs != TAHOE_OK) {
                tahoe_forest_destroy(f);
                return qs;
            }
        }
    }
    *out = f;
    return TAHOE_OK;
}

// dense2sparse, BaseTahoeTest.h:728-764: per tree a root, then for every inner node its two children are
// appended together (left_idx is relative to the tree's root) and converted depth-first, left first.
tahoe_status tahoe_dense_to_sparse(const tahoe_dense_node *dense, int num_trees, int depth, tahoe_sparse_node **nodes_out,
                                   int32_t **trees_out, size_t *num_nodes_out)
{
    if (!dense || !nodes_out || !trees_out || !num_nodes_out || num_trees < 0 || depth < 0 || depth > 30)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_dense_to_sparse: bad argument");
    std::vector<tahoe_sparse_node> out;
    std::vector<int32_t> roots((size_t)num_trees);
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(depth);
    struct Item {
        size_t i_dense, i_sparse;
    };
    std::vector<Item> stack;
    for (int t = 0; t < num_trees; ++t) {
        const tahoe_dense_node *root = dense + (size_t)t * per_tree;
        const size_t i_root = out.size();
        out.push_back(tahoe_sparse_node{0.f, 0, 0});
        roots[(size_t)t] = (int32_t)i_root;
        stack.assign(1, Item{0, i_root});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.i_dense >= per_tree) return fail(TAHOE_ERR_INVALID_FOREST, "tree %d: a bottom-level node is not a leaf", t);
            int fid, def_left, is_leaf;
            float value;
            tahoe_decode_node(&root[it.i_dense], &value, nullptr, &fid, &def_left, &is_leaf);
            tahoe_sparse_node n;
            n.val = value;
            n.bits = (fid & kSFidMask) | (def_left ? kSDefLeft : 0) | (is_leaf ? kSIsLeaf : 0);
            n.left_idx = 0;
            if (!is_leaf) {
                const size_t left = out.size();
                out.push_back(tahoe_sparse_node{0.f, 0, 0});
                out.push_back(tahoe_sparse_node{0.f, 0, 0});
                n.left_idx = (int32_t)(left - i_root);
                // depth-first, left subtree first: push right, then left
                stack.push_back(Item{2 * it.i_dense + 2, left + 1});
                stack.push_back(Item{2 * it.i_dense + 1, left});
            }
            out[it.i_sparse] = n;
        }
    }
    tahoe_sparse_node *nodes = (tahoe_sparse_node *)malloc(std::max<size_t>(out.size(), 1) * sizeof(tahoe_sparse_node));
    int32_t *trees = (int32_t *)malloc(std::max<size_t>(roots.size(), 1) * sizeof(int32_t));
    if (!nodes || !trees) {
        free(nodes);
        free(trees);
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_dense_to_sparse");
    }
    if (!out.empty()) memcpy(nodes, out.data(), out.size() * sizeof(tahoe_sparse_node));
    if (!roots.empty()) memcpy(trees, roots.data(), roots.size() * sizeof(int32_t));
    *nodes_out = nodes;
    *trees_out = trees;
    *num_nodes_out = out.size();
    return TAHOE_OK;
}

// Synthetic irregular forest: tree t may grow to depth d_t = min_depth + (x mod (max_depth -
// min_depth + 1)); levels < min_depth are always inner, below that a node turns into a leaf with probability
// leaf_prob, level d_t is all leaves; at most max_tree_nodes per tree.  Children are appended in pairs,
// breadth-first.  Call with nodes == NULL to get the node count for `seed` first.
tahoe_status tahoe_synth_sparse_forest(tahoe_sparse_node *nodes, int32_t *trees, size_t *num_nodes, int num_trees,
                                       int num_cols, int min_depth, int max_depth, float leaf_prob, int max_tree_nodes,
                                       uint64_t seed)
{
    if (!num_nodes || num_trees < 0 || num_cols < 1 || min_depth < 0 || max_depth < min_depth || max_depth > 40 ||
        max_tree_nodes < 3)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_synth_sparse_forest: bad argument");
    size_t total = 0;
    std::vector<int> level;  // level of each node of the current tree
    for (int t = 0; t < num_trees; ++t) {
        const uint64_t ts = splitmix64_at(seed, (uint64_t)t);
        const int d_t = min_depth + (int)(splitmix64_at(ts, 0) % (uint64_t)(max_depth - min_depth + 1));
        const size_t root = total;
        if (trees) trees[t] = (int32_t)root;
        level.assign(1, 0);
        size_t count = 1;  // nodes allocated so far in this tree
        for (size_t i = 0; i < count; ++i) {
            const uint64_t x0 = splitmix64_at(ts, 4 * i + 1), x1 = splitmix64_at(ts, 4 * i + 2), x2 = splitmix64_at(ts, 4 * i + 3);
            const int lv = level[i];
            const bool room = count + 2 <= (size_t)max_tree_nodes;
            const bool is_leaf = lv >= d_t || !room || (lv >= min_depth && u01(x2) < leaf_prob);
            tahoe_sparse_node n;
            n.val = 2.0f * u01(x1) - 1.0f;
            n.bits = is_leaf ? kSIsLeaf : ((int32_t)(x0 % (uint64_t)num_cols) | ((x0 >> 40) & 1 ? kSDefLeft : 0));
            n.left_idx = 0;
            if (!is_leaf) {
                n.left_idx = (int32_t)count;
                level.push_back(lv + 1);
                level.push_back(lv + 1);
                count += 2;
            }
            if (nodes) nodes[root + i] = n;
        }
        total += count;
    }
    *num_nodes = total;
    return TAHOE_OK;
}

}  // extern "C"
