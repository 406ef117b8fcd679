// TILERING for wide rows ("WIDEF"): the float32 walk when a 64-row float32 tile does not fit LDS (num_cols > 512) and there
// is too little walking per feature value for the quantise pass of QRING to pay (K2, the reference's SVHN shape: 3072
// features, 500 trees of depth 8 -- every feature value is used 1.3 times).  The scheme of qwide_kernel (qring.hip) on the
// float32 batch itself:
//   * a tile is RT = 32 / 16 / 8 consecutive rows exactly as they lie in memory (row-major, each LDS row padded by four
//     floats so that the rows of one tree, which read the same feature near the root, fall into different banks): staging is
//     a straight 16-byte copy, no transposition, no pre-pass;
//   * a walker wave walks TPW = 64 / RT trees at once, lane = (tree slot j, row r); its LDS slot holds the first lw levels of
//     those trees as 8-byte records {threshold, fid | def_left << 31} at 1-based heap positions, so both children of a node
//     come with one 16-byte read beside the feature read; the next iteration's tops are prefetched into registers;
//   * levels lw .. De-3 (if any) come from the heap records in global memory, the last two levels and the leaf from one
//     48-byte block per walk ({thr0, thr1, thr2, meta0} {meta1, meta2, -, -} {leaf0..3}; the float32 tile kernels' 32-byte
//     blocks have 9-bit feature ids), gathered one iteration ahead;
//   * leaf values go through an LDS ring to a consumer wave that adds them in tree order: float32 sums bit-identical to
//     predict_on_cpu (BaseTahoeTest.h:462-466).
// While staging, every thread checks the values it copies against `missing`; a tile without a missing value takes the branch
// rule's fast path (x >= thr; NaN goes left, as in BaseTahoeTest.h:450-453 where the missing test is false for NaN).
// Replaces, like forest.hip: the walkers / kernels of Struct.h:953-1704 for shapes where a row does not fit shared memory.
#include <algorithm>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "forest_internal.h"

struct tahoe_wstate {
    uint2 *ftop = nullptr;     // [T][tstride] {thr bits, meta}: heap node i at entry i + 1, the first 2^lw - 1 nodes
    uint4 *fblocks = nullptr;  // [T][2^(De-2)][3]
    int rt = 0, nwalk = 0, lw = 0, tstride = 0;
    // row-streaming form (wstream_kernel): the first s_lw levels of ALL trees as one LDS image, node-major
    unsigned char *simg = nullptr;  // thr[2^s_lw - 1][s_ts] f32 | meta[2^s_lw - 1][s_ts] u16, padded to 1 KiB pieces
    uint4 *sblocks = nullptr;       // [T][2^(De-3)][8]: the last three levels + eight leaves of a subtree in ONE 128-byte line
    int s_lw = 0, s_ts = 0, s_slots = 0, s_img_bytes = 0, s_q = 0;
    bool s_on = false;              // the launch prefers it (shape rule or TAHOE_WSTREAM=1)
};

namespace tahoe {

constexpr int kWfRingBytes = 8192;
constexpr int kWfSpinLimit = 1 << 22;

template <int RT, int NWALK, bool WRITE_LEAF>
__global__ void __launch_bounds__((NWALK + 1) * 64)
    widef_kernel(const float *__restrict__ data, const uint2 *__restrict__ ftop, const uint4 *__restrict__ fblocks,
                 const InnerNode *__restrict__ inner, const uint32_t *__restrict__ leaf_orig, float *sums,
                 uint32_t *__restrict__ leaf_out, const float *sums_in, size_t rows, int cols, int num_trees, int depth, int lw,
                 int tstride, float missing, int vec4_ok, int *__restrict__ error_flag)
{
    constexpr int NT = (NWALK + 1) * 64;
    constexpr int TPW = 64 / RT;                    // trees a walker wave walks at once
    constexpr int RE = kWfRingBytes / (RT * 4);     // ring entries (trees): two rounds of NWALK * TPW
    constexpr int NBATCH = RE / 2 < 64 ? RE / 2 : 64;  // trees the consumer takes per poll
    static_assert(RE >= 2 * NWALK * TPW, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = cols + 4;  // floats per tile row
    float *tile = reinterpret_cast<float *>(smem);
    const int slot_bytes = TPW * tstride * 8;
    unsigned char *slots = smem + (size_t)RT * stride * sizeof(float);
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * slot_bytes);
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + RE * RT);
    uint32_t *consumed = ring_ready + RE;
    uint32_t *ms_seen = consumed + 1;
    const size_t row0 = (size_t)blockIdx.x * RT;

    for (int e = tid; e < RE; e += NT) ring_ready[e] = 0u;
    if (tid == 0) {
        *consumed = 0u;
        *ms_seen = 0u;
    }
    __syncthreads();  // ms_seen is zero before anyone sets it
    {   // ---- stage the rows as they lie in memory; note whether any value is a missing value ----
        bool ms = false;
        if (vec4_ok) {
            const int n4 = cols / 4;
            for (int e = tid; e < RT * n4; e += NT) {
                const int r = e / n4, c4 = e - r * n4;
                const size_t grow = row0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grow < rows) v = reinterpret_cast<const float4 *>(data + grow * (size_t)cols)[c4];
                *reinterpret_cast<float4 *>(tile + (size_t)r * stride + 4 * c4) = v;
                ms |= fabsf(v.x - missing) <= kMissingEps || fabsf(v.y - missing) <= kMissingEps || fabsf(v.z - missing) <= kMissingEps ||
                      fabsf(v.w - missing) <= kMissingEps;
            }
        } else {
            for (int e = tid; e < RT * cols; e += NT) {
                const int r = e / cols, c = e - r * cols;
                const size_t grow = row0 + r;
                const float v = grow < rows ? data[grow * (size_t)cols + c] : 0.0f;
                tile[(size_t)r * stride + c] = v;
                ms |= fabsf(v - missing) <= kMissingEps;
            }
        }
        if (__ballot(ms) != 0ull && lane == 0) atomicOr(ms_seen, 1u);
    }

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
                if (++spins > kWfSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();  // the values are read after the flags
            if (lane < RT) {  // tree order; eight loads in flight, eight adds in order (one tree at a time, a load's
                               // latency per tree, made this wave the bottleneck of a 16-row tile: 500 trees x ~100 clk)
                int jj = 0;
                for (; jj + 8 <= nb; jj += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ring_vals[((t0 + jj + u) % RE) * RT + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sum += v[u];
                }
                for (; jj < nb; ++jj) sum += ring_vals[((t0 + jj) % RE) * RT + lane];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_flag_store(consumed, (uint32_t)(t0 + nb));
        }
        if (dead && lane == 0) atomicOr(error_flag, 1);
        if (sums && lane < RT && irow < rows) sums[irow] = sum;
        return;
    }

    // ================= walkers =================
    const int j = lane / RT, r = lane % RT;  // tree slot, row of the tile
    const float *xrow = tile + (size_t)r * stride;
    const int cpt = tstride / 2;             // 16-byte chunks per staged top (tstride >= 2 entries)
    unsigned char *wslot = slots + (size_t)wave * slot_bytes;
    const uint2 *slot = reinterpret_cast<const uint2 *>(wslot) + (size_t)j * tstride;
    const int n_groups = (num_trees + TPW - 1) / TPW;
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};  // named registers (an indexed array would go to scratch)
    // chunk c of the wave's slot = chunk (c % cpt) of tree g * TPW + c / cpt (clamped: in bounds, branch-free); TPW * cpt <= 256
    auto prefetch_tops = [&](int g) {
        auto ld = [&](int c) {
            c = min(c, TPW * cpt - 1);
            const int t = min(g * TPW + c / cpt, num_trees - 1);
            return reinterpret_cast<const uint4 *>(ftop + (size_t)t * tstride)[c % cpt];
        };
        pf0 = ld(0 * 64 + lane);
        pf1 = ld(1 * 64 + lane);
        pf2 = ld(2 * 64 + lane);
        pf3 = ld(3 * 64 + lane);
    };
    auto commit_tops = [&]() {  // clamped lanes rewrite the last chunk with its own value
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
    __syncthreads();  // the tile, the ring state, ms_seen and (own wave) the first tops are in LDS

    bool dead = false;
    auto run = [&](auto ms_tag) {
        constexpr bool MS = decltype(ms_tag)::value;
        // the branch rule of infer_one_tree (BaseTahoeTest.h:450-453); without a missing value in the tile it is x >= thr
        auto right = [&](float x, uint32_t thr_bits, uint32_t meta) -> uint32_t {
            if (MS) return go_right(x, __uint_as_float(thr_bits), (meta >> 31) != 0u, missing);
            return x >= __uint_as_float(thr_bits) ? 1u : 0u;
        };
        const size_t n_inner = ((size_t)1 << depth) - 1;
        const uint32_t n_blocks = 1u << (depth - 2);
        const uint32_t first_block_node = n_blocks - 1;
        const size_t row = row0 + r;
        auto finish = [&](int g, const uint4 &a, const uint4 &b, const uint4 &c, uint32_t bs) {
            const int t = g * TPW + j;
            const uint32_t c0 = right(xrow[a.w & kMetaFidMask], a.x, a.w);
            const uint32_t thr1 = c0 ? a.z : a.y, m1 = c0 ? b.y : b.x;
            const uint32_t c1 = right(xrow[m1 & kMetaFidMask], thr1, m1);
            const uint32_t lo = c0 ? c.z : c.x, hi = c0 ? c.w : c.y;
            const float v = __uint_as_float(c1 ? hi : lo);
            if (WRITE_LEAF) {
                if (t < num_trees && row < rows)
                    leaf_out[row * (size_t)num_trees + t] = leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bs * 4 + 2 * c0 + c1];
            }
            const int t_last = min(g * TPW + TPW - 1, num_trees - 1);
            if (t_last >= RE) {  // the group's ring entries still in use?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t_last - RE + 1)) {
                    if (++spins > kWfSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (t < num_trees) ring_vals[(t % RE) * RT + r] = v;
            TAHOE_LDS_RELEASE();  // values before the flags: a wave's LDS operations are performed in issue order
            if (r == 0 && t < num_trees) lds_flag_store(&ring_ready[t % RE], (uint32_t)(t + 1));
        };
        int g_p = -1;  // group whose bottom blocks are in flight
        uint4 a_p = {}, b_p = {}, c_p = {};
        uint32_t bsel_p = 0;
        for (int g = wave; g < n_groups && !dead; g += NWALK) {
            const bool more = g + NWALK < n_groups;
            if (more) prefetch_tops(g + NWALK);
            const int t = min(g * TPW + j, num_trees - 1);  // lanes of a missing tree repeat the last one, unused
            uint32_t i = 1;
            if (lw > 0) {
                uint2 n = slot[1];
                for (int l = 0; l < lw - 1; ++l) {
                    const float x = xrow[n.y & kMetaFidMask];
                    const uint4 pr = *reinterpret_cast<const uint4 *>(&slot[2 * i]);  // children 2i, 2i+1
                    const uint32_t c = right(x, n.x, n.y);
                    i = 2u * i + c;
                    n = c ? make_uint2(pr.z, pr.w) : make_uint2(pr.x, pr.y);
                }
                i = 2u * i + right(xrow[n.y & kMetaFidMask], n.x, n.y);
            }
            uint32_t idx = i - 1;  // 0-based heap index on level lw
            if (lw < depth - 2) {  // the levels between the slot and the bottom blocks: heap records in global memory
                const InnerNode *tree = inner + (size_t)t * n_inner;
                for (int l = lw; l < depth - 2; ++l) {
                    const InnerNode nd = tree[idx];
                    idx = 2u * idx + 1u + right(xrow[nd.meta & kMetaFidMask], __float_as_uint(nd.thr), nd.meta);
                }
            }
            const uint32_t bsel = idx - first_block_node;
            if (g_p >= 0) finish(g_p, a_p, b_p, c_p, bsel_p);
            g_p = g;
            const uint4 *bp = fblocks + ((size_t)t * n_blocks + bsel) * 3;
            a_p = bp[0];
            b_p = bp[1];
            c_p = bp[2];
            bsel_p = bsel;
            if (more) commit_tops();  // this wave's reads of its slot are done (in-order LDS)
        }
        if (g_p >= 0 && !dead) finish(g_p, a_p, b_p, c_p, bsel_p);
    };
    if (lds_flag_load(ms_seen) != 0u)
        run(std::true_type{});
    else
        run(std::false_type{});
    if (dead && lane == 0) atomicOr(error_flag, 1);
}


// ================================================================================================
// Row-streaming form ("WSTREAM", round 3).  The tile form above stages a tile, walks it, and starts over: one workgroup per
// CU with LDS full, so nothing overlaps the staging (K2: 1.2 GB streamed at 1.5 TB/s, profiles/r03/pmc_k2.json), and every
// 8-row tile re-stages the tops of all trees from L2 (2.6 x the bytes of the rows).  Here the roles are swapped, the way the
// reference lays its forest out for "lane = tree" (node-major `reorg` arrays, Struct.h:1911-1923; walker :1035-1071):
//   * ONE persistent workgroup per CU keeps the first lw levels of ALL trees in LDS for its whole life, node-major:
//     thr[i][t] f32 and meta[i][t] u16 (fid | def_left << 15).  Lane = tree: the 64 lanes of a wave hold 64 consecutive trees,
//     so a node read is conflict-free whatever node each lane stands on (bank = t mod 32), the reference's coalescing argument
//     moved into LDS;
//   * the rows of the workgroup's share of the batch stream through a ring of S row slots, pulled by ONE loader wave with
//     LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, non-temporal), Q pieces in flight, rows
//     published in order through a counter of landed pieces (tools/ubench_wstream.hip: 5.5 - 6.9 TB/s with 4 - 8 slots);
//   * a walker wave takes item (row k, chunk c) = 64 trees of one row: lw levels from LDS (the row's feature values are
//     plain LDS reads at per-lane addresses), the levels below from L2: heap records, then the last three levels and the
//     eight leaves below them as ONE 128-byte line per walk (the tile form's 8-byte record + 48-byte block were 2.3 lines:
//     the L2 -> L1 line traffic of these gathers, not HBM, is what bounds this shape); the leaf value goes into
//     vals[k % 8][t]; the last walker of a row frees its slot for the loader;
//   * a consumer wave adds the 8-row ring of leaf values in tree order, one lane per row, four rows at a time: float32 sums
//     bit-identical to predict_on_cpu (BaseTahoeTest.h:462-466).
// The loader's own flag accesses are inline asm: while an LDS-DMA is in flight hipcc puts s_waitcnt vmcnt(0) in front of
// every LDS access it can see (the DMA is a pending LDS write), which would drain the stream once per row; and its per-piece
// work is the load alone -- a wave issues one instruction per ~5 clk, per-piece bookkeeping cost 3 x the stream's time.
// Timing-only ablation builds (make ABLATE=n; results are wrong on purpose; never shipped): 1 = the consumer adds nothing,
// 2 = no global gathers below the LDS levels, 3 = walkers only pass the rows on, 4 = the loader loads nothing.
#ifndef TAHOE_WS_ABLATE
#define TAHOE_WS_ABLATE 0
#endif
constexpr int kWsVals = 8;   // rows of leaf values between walkers and consumer
constexpr int kWsGroup = 4;  // rows the consumer adds at once (one lane each)

__device__ __forceinline__ uint32_t ws_lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }  // low half of a generic LDS address
__device__ __forceinline__ uint32_t ws_flag_load_asm(const uint32_t *p)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ws_lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void ws_flag_store_asm(uint32_t *p, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(ws_lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void ws_dma16(const unsigned char *src, unsigned char *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0,
                                     2 /* nt: every row is read once */);
}

template <int NWALK, bool WRITE_LEAF, int Q>
__global__ void __launch_bounds__((NWALK + 2) * 64)
    wstream_kernel(const float *__restrict__ data, const unsigned char *__restrict__ simg, const uint4 *__restrict__ sblocks,
                   const InnerNode *__restrict__ inner, const uint32_t *__restrict__ leaf_orig, float *sums, uint32_t *__restrict__ leaf_out,
                   const float *sums_in, size_t rows, int cols, int num_trees, int depth, int lw, int ts, int img_bytes, int S, float missing,
                   int *__restrict__ error_flag)
{
    constexpr int NW = NWALK + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nlv = (1 << lw) - 1;
    const int tv = (num_trees + 3) & ~3;
    const int row_bytes = cols * 4;
    const float *sthr = reinterpret_cast<const float *>(smem);
    const uint16_t *smeta = reinterpret_cast<const uint16_t *>(smem + (size_t)nlv * ts * 4);
    float *srows = reinterpret_cast<float *>(smem + img_bytes);
    float *vals = srows + (size_t)S * cols;
    uint32_t *landed = reinterpret_cast<uint32_t *>(vals + (size_t)kWsVals * tv);  // DMA pieces of this workgroup's rows in LDS
    uint32_t *consumed = landed + 1;                                             // rows the consumer has added
    uint32_t *walked = landed + 2;                                               // [S] chunk-walks finished in the slot, monotone
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per;
    if (r0 >= rows) return;
    const int n = (int)(rows - r0 < per ? rows - r0 : per);
    const int nch = (num_trees + 63) >> 6;  // 64-tree chunks per row
    const int P = (row_bytes + 1023) >> 10;  // DMA pieces per row

    // ---- the resident tops: the image lies in global memory exactly as in LDS ----
    for (int pc = wave; pc < (img_bytes >> 10); pc += NW) ws_dma16(simg + (size_t)pc * 1024 + lane * 16, smem + (size_t)pc * 1024);
    if (tid == 0) {
        *landed = 0u;
        *consumed = 0u;
    }
    if (tid < S) walked[tid] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (wave == 0) {
        // ================= loader =================
        const int p_full = row_bytes >> 10, rem = row_bytes & 1023;
        int issued = 0, pub = 0;
        for (int k = 0; k < n; ++k) {
            const int slot = k % S;
            if (k >= S) {
                const uint32_t need = (uint32_t)nch * (uint32_t)(k / S);  // every earlier row of this slot walked
                if (ws_flag_load_asm(&walked[slot]) < need) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // never block with unpublished pieces in flight
                    pub = issued;
                    ws_flag_store_asm(landed, (uint32_t)pub);
                    int spins = 0;
                    while (ws_flag_load_asm(&walked[slot]) < need) {
                        if (++spins > kWfSpinLimit) {
                            if (lane == 0) atomicOr(error_flag, 1);
                            return;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + (r0 + k) * (size_t)cols) + lane * 16;
            unsigned char *dst = reinterpret_cast<unsigned char *>(srows) + (size_t)slot * row_bytes;
#if TAHOE_WS_ABLATE != 4
            for (int p = 0; p < p_full; ++p) ws_dma16(src + p * 1024, dst + p * 1024);
            if (rem && lane * 16 < rem) ws_dma16(src + p_full * 1024, dst + p_full * 1024);
#endif
            issued += P;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory");  // all but the Q youngest pieces have landed
            if (issued - Q > pub) {
                pub = issued - Q;
                ws_flag_store_asm(landed, (uint32_t)pub);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ws_flag_store_asm(landed, (uint32_t)issued);
        return;
    }

    if (wave == 1) {
        // ================= consumer: ordered accumulation, lane = row of a group of kWsGroup =================
        bool dead = false;
        for (int g0 = 0; g0 < n && !dead; g0 += kWsGroup) {
            const int nb = min(kWsGroup, n - g0);
            const int k = g0 + min(lane, nb - 1);
            const uint32_t need = (uint32_t)nch * (uint32_t)(k / S + 1);
            int spins = 0;
            for (;;) {
                const bool ok = lds_flag_load(&walked[k % S]) >= need;
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kWfSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();  // the values are read after the flags
            if (lane < nb && TAHOE_WS_ABLATE != 1) {
                const float *v = vals + (size_t)(k % kWsVals) * tv;
                float sum = sums_in ? sums_in[r0 + k] : 0.0f;
                int t = 0;
                for (; t + 8 <= num_trees; t += 8) {  // tree order: two 16-byte reads in flight, eight adds in order
                    const float4 a = *reinterpret_cast<const float4 *>(v + t);
                    const float4 b = *reinterpret_cast<const float4 *>(v + t + 4);
                    sum += a.x;
                    sum += a.y;
                    sum += a.z;
                    sum += a.w;
                    sum += b.x;
                    sum += b.y;
                    sum += b.z;
                    sum += b.w;
                }
                for (; t < num_trees; ++t) sum += v[t];
                if (sums) sums[r0 + k] = sum;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the values have been read
            if (lane == 0) lds_flag_store(consumed, (uint32_t)(g0 + nb));
        }
        if (dead && lane == 0) atomicOr(error_flag, 1);
        return;
    }

    // ================= walkers: item = (row k, chunk c); lane = tree c * 64 + lane =================
    const size_t n_inner = ((size_t)1 << depth) - 1;
    const uint32_t n_blocks = 1u << (depth - 3);  // three-level subtrees
    const uint32_t first_block_node = n_blocks - 1;
    bool dead = false;
    int k = 0, c = wave - 2;
    while (c >= nch) {
        c -= nch;
        ++k;
    }
    while (k < n && !dead) {
        {   // the row is in LDS, and the consumer is done with the ring entry
            const uint32_t want = (uint32_t)(k + 1) * (uint32_t)P;
            int spins = 0;
            while (lds_flag_load(landed) < want || (k >= kWsVals && lds_flag_load(consumed) + kWsVals <= (uint32_t)k)) {
                if (++spins > kWfSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();
        }
        const int slot = k % S;
        const float *xr = srows + (size_t)slot * cols;
        const int t = c * 64 + lane;
        const int tt = min(t, num_trees - 1);  // lanes past the last tree repeat it, unused
        uint32_t i = 0;                       // heap index
#if TAHOE_WS_ABLATE == 3
        if (false)
#endif
        for (int l = 0; l < lw; ++l) {
            const float thr = sthr[i * ts + tt];
            const uint32_t m = smeta[i * ts + tt];
            const float x = xr[m & 0x7fffu];
            i = 2u * i + 1u + go_right(x, thr, (m >> 15) != 0u, missing);
        }
        uint32_t idx = i;
#if TAHOE_WS_ABLATE == 2 || TAHOE_WS_ABLATE == 3
        const uint32_t bsel = 0, j = 0;
        const float leaf = __uint_as_float(i);
#else
        if (lw < depth - 3) {  // the levels between the LDS tops and the bottom blocks: heap records in global memory
            const InnerNode *tree = inner + (size_t)tt * n_inner;
            for (int l = lw; l < depth - 3; ++l) {
                const InnerNode nd = tree[idx];
                idx = 2u * idx + 1u + go_right(xr[nd.meta & kMetaFidMask], nd.thr, (nd.meta >> 31) != 0u, missing);
            }
        }
        // the last three levels and the leaf: one 128-byte line per walk (five 16-byte loads of the same line)
        //   q0 {thr0 | thr1 thr2 | thr3}  q1 {thr4 thr5 thr6, meta0 | meta1 << 16}  q2 {meta2 | meta3 << 16, meta4 | meta5 << 16, meta6, -}
        //   q3 {leaf0..3}  q4 {leaf4..7};  nodes in heap order of the subtree, meta = fid | def_left << 15
        const uint32_t bsel = idx - first_block_node;
        const uint4 *bp = sblocks + ((size_t)tt * n_blocks + bsel) * 8;
#if TAHOE_WS_ABLATE == 5  // three loads inside one 64-byte half-line
        const uint4 q0 = bp[0], q1 = bp[1], q2 = bp[2], q3 = q0, q4 = q1;
#elif TAHOE_WS_ABLATE == 6  // one load
        const uint4 q0 = bp[0], q1 = q0, q2 = q0, q3 = q0, q4 = q0;
#elif TAHOE_WS_ABLATE == 7  // two loads, one per 64-byte half
        const uint4 q0 = bp[0], q1 = q0, q2 = q0, q3 = q0, q4 = bp[4];
#else
        const uint4 q0 = bp[0], q1 = bp[1], q2 = bp[2], q3 = bp[3], q4 = bp[4];
#endif
        const uint32_t m0 = q1.w & 0xffffu;
        const uint32_t c0 = go_right(xr[m0 & 0x7fffu], __uint_as_float(q0.x), (m0 >> 15) != 0u, missing);
        const uint32_t thr_b = c0 ? q0.z : q0.y, m_b = c0 ? (q2.x & 0xffffu) : (q1.w >> 16);
        const uint32_t c1 = go_right(xr[m_b & 0x7fffu], __uint_as_float(thr_b), (m_b >> 15) != 0u, missing);
        const uint32_t thr_c = c0 ? (c1 ? q1.z : q1.y) : (c1 ? q1.x : q0.w);
        const uint32_t m_c = c0 ? (c1 ? q2.z : (q2.y >> 16)) : (c1 ? (q2.y & 0xffffu) : (q2.x >> 16));
        const uint32_t c2 = go_right(xr[m_c & 0x7fffu], __uint_as_float(thr_c), ((m_c >> 15) & 1u) != 0u, missing);
        const uint32_t la = c0 ? q4.x : q3.x, lb = c0 ? q4.y : q3.y, lc = c0 ? q4.z : q3.z, ld = c0 ? q4.w : q3.w;
        const uint32_t le = c1 ? lc : la, lf = c1 ? ld : lb;
        const float leaf = __uint_as_float(c2 ? lf : le);
        const uint32_t j = 4u * c0 + 2u * c1 + c2;
#endif
        if (t < num_trees) vals[(size_t)(k % kWsVals) * tv + t] = leaf;
        if (WRITE_LEAF) {
            if (t < num_trees) leaf_out[(r0 + k) * (size_t)num_trees + t] = leaf_orig[(size_t)t * ((size_t)n_blocks * 8) + (size_t)bsel * 8 + j];
        }
        TAHOE_LDS_RELEASE();  // values before the counter; the row's last read precedes it too (in-order LDS)
        if (lane == 0) atomicAdd(&walked[slot], 1u);
        c += NWALK;
        while (c >= nch) {
            c -= nch;
            ++k;
        }
    }
    if (dead && lane == 0) atomicOr(error_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// host side
static int wf_lw_max(int rt) { return rt == 32 ? 8 : rt == 16 ? 7 : 6; }  // (64 / rt) * 2^lw / 2 <= 256 chunks
static int wf_tstride(int lw) { return std::max(2, 1 << lw); }
static long long wf_lds(int cols, int rt, int nwalk, int lw)
{
    const long long re = kWfRingBytes / (rt * 4);
    return (long long)rt * (cols + 4) * 4 + (long long)nwalk * (64 / rt) * wf_tstride(lw) * 8 + kWfRingBytes + (re + 2) * 4;
}

int widef_rows(const tahoe_forest *f) { return f->wf ? f->wf->rt : 0; }

// ---- row-streaming form ----
constexpr int kWsWalkers = 14;
static long long ws_img_bytes(int lw, int ts) { return ((((1LL << lw) - 1) * ts * 6) + 1023) & ~1023LL; }
static long long ws_lds(int cols, int num_trees, int lw, int ts, int slots)
{
    const long long tv = (num_trees + 3) & ~3;
    return ws_img_bytes(lw, ts) + (long long)slots * cols * 4 + (long long)kWsVals * tv * 4 + (2 + slots) * 4 + 16;
}
int widef_stream_slots(const tahoe_forest *f) { return f->wf && f->wf->s_on ? f->wf->s_slots : 0; }
int widef_stream_levels(const tahoe_forest *f) { return f->wf && f->wf->s_on ? f->wf->s_lw : 0; }
long long widef_lds_bytes(const tahoe_forest *f)
{
    const tahoe_wstate *w = f->wf;
    if (!w) return 0;
    return w->s_on ? ws_lds(f->p.num_cols, f->p.num_trees, w->s_lw, w->s_ts, w->s_slots) : wf_lds(f->p.num_cols, w->rt, w->nwalk, w->lw);
}

template <int Q>
static hipError_t ws_allow(int limit)
{
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(&wstream_kernel<kWsWalkers, false, Q>), limit);
    if (e != hipSuccess) return e;
    return allow_max_lds(reinterpret_cast<const void *>(&wstream_kernel<kWsWalkers, true, Q>), limit);
}

void widef_destroy(tahoe_forest *f)
{
    if (!f->wf) return;
    if (f->wf->ftop) (void)hipFree(f->wf->ftop);
    if (f->wf->fblocks) (void)hipFree(f->wf->fblocks);
    if (f->wf->simg) (void)hipFree(f->wf->simg);
    if (f->wf->sblocks) (void)hipFree(f->wf->sblocks);
    delete f->wf;
    f->wf = nullptr;
}

template <int RT, int NWALK>
static hipError_t wf_allow(int limit)
{
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(&widef_kernel<RT, NWALK, false>), limit);
    if (e != hipSuccess) return e;
    return allow_max_lds(reinterpret_cast<const void *>(&widef_kernel<RT, NWALK, true>), limit);
}

// Builds the tops and blocks of the wide float32 form when it is the only float32 tile kernel this shape can have (the caller
// checks that).  Leaves f->wf null (TAHOE_OK) when no tile of >= 8 rows fits LDS beside the walkers' slots.
tahoe_status widef_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<float> &h_leaf)
{
    const int cols = f->p.num_cols, De = f->depth;
    const size_t T = (size_t)f->p.num_trees;
    if (T == 0 || cols < 1 || De < 2 || f->relayout) return TAHOE_OK;  // the exchange bit is not part of this walk
    if (const char *e = getenv("TAHOE_WIDEF"))  // experiments: 0 leaves the shape to QRING / DIRECT
        if (atoi(e) == 0) return TAHOE_OK;
    int rt = 0, nwalk = 0, lw = 0;
    for (int cand : {32, 16, 8}) {
        const int hi = std::min(De - 2, wf_lw_max(cand));
        for (int l = hi; l >= std::max(0, hi - 2) && !rt; --l)
            for (int nw : {15, 12})
                if (wf_lds(cols, cand, nw, l) <= f->lds_limit) {
                    rt = cand;
                    nwalk = nw;
                    lw = l;
                    break;
                }
        if (rt) break;
    }
    if (!rt) return TAHOE_OK;
    tahoe_wstate *w = new (std::nothrow) tahoe_wstate();
    if (!w) return fail(TAHOE_ERR_NO_MEMORY, "widef_build");
    f->wf = w;
    w->rt = rt;
    w->nwalk = nwalk;
    w->lw = lw;
    w->tstride = wf_tstride(lw);
    const size_t n_inner = f->n_inner, n_leaf = f->n_leaf, n_blocks = (size_t)1 << (De - 2), first = n_blocks - 1;
    const size_t n_top = ((size_t)1 << lw) - 1;  // heap nodes in the LDS tops
    std::vector<uint2> h_top(T * (size_t)w->tstride, make_uint2(0u, 0u));
    std::vector<uint4> h_blocks(T * n_blocks * 3);
    parallel_for(T, 8, [&](size_t t_lo, size_t t_hi) {
        for (size_t t = t_lo; t < t_hi; ++t) {
            const InnerNode *in = &h_inner[t * n_inner];
            for (size_t i = 0; i < n_top; ++i) {  // heap node i -> position i + 1
                uint2 rec;
                memcpy(&rec.x, &in[i].thr, 4);
                rec.y = in[i].meta;
                h_top[t * (size_t)w->tstride + i + 1] = rec;
            }
            for (size_t b = 0; b < n_blocks; ++b) {
                const size_t r = first + b, l = 2 * r + 1, rr = 2 * r + 2;  // subtree root and its children
                uint4 a, m, v;
                memcpy(&a.x, &in[r].thr, 4);
                memcpy(&a.y, &in[l].thr, 4);
                memcpy(&a.z, &in[rr].thr, 4);
                a.w = in[r].meta;
                m = make_uint4(in[l].meta, in[rr].meta, 0u, 0u);
                const float *lv = &h_leaf[t * n_leaf + 4 * b];
                memcpy(&v.x, &lv[0], 4);
                memcpy(&v.y, &lv[1], 4);
                memcpy(&v.z, &lv[2], 4);
                memcpy(&v.w, &lv[3], 4);
                h_blocks[(t * n_blocks + b) * 3 + 0] = a;
                h_blocks[(t * n_blocks + b) * 3 + 1] = m;
                h_blocks[(t * n_blocks + b) * 3 + 2] = v;
            }
        }
    });
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "widef_build: %s failed: %s", what, hipGetErrorString(e)); };
    auto up = [&](auto **dst, const auto &src) {
        const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(src[0]);
        hipError_t er = hipMalloc(reinterpret_cast<void **>(dst), bytes);
        if (er != hipSuccess) return er;
        f->device_bytes += bytes;
        return src.empty() ? hipSuccess : hipMemcpy(*dst, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice);
    };
    if ((e = up(&w->ftop, h_top)) != hipSuccess) return bad("ftop");
    if ((e = up(&w->fblocks, h_blocks)) != hipSuccess) return bad("fblocks");
    e = rt == 32 ? (nwalk == 15 ? wf_allow<32, 15>(f->lds_limit) : wf_allow<32, 12>(f->lds_limit))
        : rt == 16 ? (nwalk == 15 ? wf_allow<16, 15>(f->lds_limit) : wf_allow<16, 12>(f->lds_limit))
                   : (nwalk == 15 ? wf_allow<8, 15>(f->lds_limit) : wf_allow<8, 12>(f->lds_limit));
    if (e != hipSuccess) return bad("hipFuncSetAttribute");

    // ---- the row-streaming form: all trees' first levels resident, rows through a ring of slots ----
    // levels: as many as leave room for >= 4 row slots (3 if that buys a level); rows of 16-byte multiples only
    int knob = -1;
    if (const char *e2 = getenv("TAHOE_WSTREAM")) knob = atoi(e2);
    if (knob == 1 && cols % 4 == 0 && cols <= 32768 && T <= (size_t)1 << 20 && De >= 3) {
        const int ts = (int)((T + 3) & ~(size_t)3);
        int s_lw = -1, s_slots = 0;  // (depth 3: no resident level at all, the whole tree is its one bottom block)
        for (int min_slots : {4, 3}) {
            for (int l = std::min(De - 3, 10); l >= 0 && s_lw < 0; --l)
                if (ws_lds(cols, (int)T, l, ts, min_slots) <= f->lds_limit) s_lw = l;
            if (s_lw >= 0) {
                s_slots = min_slots;
                while (s_slots < 16 && ws_lds(cols, (int)T, s_lw, ts, s_slots + 1) <= f->lds_limit) ++s_slots;
                break;
            }
        }
        if (s_lw >= 0) {
            const size_t nlv = ((size_t)1 << s_lw) - 1;
            const size_t img = (size_t)ws_img_bytes(s_lw, ts);
            std::vector<unsigned char> h_img(img, 0);
            float *thr = reinterpret_cast<float *>(h_img.data());
            uint16_t *meta = reinterpret_cast<uint16_t *>(h_img.data() + nlv * ts * 4);
            for (size_t t = 0; t < T; ++t)
                for (size_t i = 0; i < nlv; ++i) {
                    const InnerNode &nd = h_inner[t * n_inner + i];
                    thr[i * ts + t] = nd.thr;
                    meta[i * ts + t] = (uint16_t)((nd.meta & 0x7fffu) | ((nd.meta >> 31) << 15));
                }
            if ((e = up(&w->simg, h_img)) != hipSuccess) return bad("simg");
            const size_t nb3 = (size_t)1 << (De - 3), first3 = nb3 - 1;
            std::vector<uint4> h_sb(T * nb3 * 8, make_uint4(0u, 0u, 0u, 0u));
            parallel_for(T, 8, [&](size_t t_lo, size_t t_hi) {
                for (size_t t = t_lo; t < t_hi; ++t) {
                    const InnerNode *in = &h_inner[t * n_inner];
                    for (size_t b = 0; b < nb3; ++b) {
                        size_t nd[7];  // the subtree's nodes in heap order
                        nd[0] = first3 + b;
                        for (int q = 0; q < 3; ++q) {
                            nd[2 * q + 1] = 2 * nd[q] + 1;
                            nd[2 * q + 2] = 2 * nd[q] + 2;
                        }
                        uint32_t thr[7], m16[7];
                        for (int q = 0; q < 7; ++q) {
                            memcpy(&thr[q], &in[nd[q]].thr, 4);
                            m16[q] = (in[nd[q]].meta & 0x7fffu) | ((in[nd[q]].meta >> 31) << 15);
                        }
                        uint4 *o = &h_sb[(t * nb3 + b) * 8];
                        o[0] = make_uint4(thr[0], thr[1], thr[2], thr[3]);
                        o[1] = make_uint4(thr[4], thr[5], thr[6], m16[0] | (m16[1] << 16));
                        o[2] = make_uint4(m16[2] | (m16[3] << 16), m16[4] | (m16[5] << 16), m16[6], 0u);
                        const float *lv = &h_leaf[t * n_leaf + 8 * b];
                        memcpy(&o[3], lv, 16);
                        memcpy(&o[4], lv + 4, 16);
                    }
                }
            });
            if ((e = up(&w->sblocks, h_sb)) != hipSuccess) return bad("sblocks");
            w->s_lw = s_lw;
            w->s_ts = ts;
            w->s_slots = s_slots;
            w->s_img_bytes = (int)img;
            const int pieces = (cols * 4 + 1023) / 1024;
            w->s_q = (long long)(s_slots - 2) * pieces >= 24 ? 24 : 8;
            // Built and bit-exact, but NOT the default: on K2 it takes 0.91 ms against the tile form's 0.78 (profiles/r03/
            // wstream_ablation.txt, pmc_k2_wstream.json): the texture path is ~85 % busy in both forms, a divergent 16-byte
            // gather costs it about a cycle per lane, and five of them per walk (one 128-byte line, re-fetched 1.8 x because 14
            // walkers' lines in flight overflow the 32-KiB L1) cost more than the tile form's three plus its coalesced top
            // staging; stream and gathers do not overlap because both sit in that one queue.  TAHOE_WSTREAM=1 selects it.
            w->s_on = knob == 1;
            e = ws_allow<24>(f->lds_limit);
            if (e == hipSuccess) e = ws_allow<8>(f->lds_limit);
            if (e != hipSuccess) return bad("hipFuncSetAttribute");
        }
    }
    return TAHOE_OK;
}

template <int RT, int NWALK>
static void wf_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows, hipStream_t stream,
                      const float *sums_in, int vec4_ok)
{
    const tahoe_wstate *w = f->wf;
    const unsigned grid = (unsigned)((rows + RT - 1) / RT);
    const int lds = (int)wf_lds(f->p.num_cols, RT, NWALK, w->lw);
    if (leaf_out)
        hipLaunchKernelGGL((widef_kernel<RT, NWALK, true>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, data, w->ftop, w->fblocks,
                           f->inner, f->leaf_orig, sums, leaf_out, sums_in, rows, f->p.num_cols, f->p.num_trees, f->depth, w->lw,
                           w->tstride, f->p.missing, vec4_ok, f->error_flag);
    else
        hipLaunchKernelGGL((widef_kernel<RT, NWALK, false>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, data, w->ftop, w->fblocks,
                           f->inner, f->leaf_orig, sums, leaf_out, sums_in, rows, f->p.num_cols, f->p.num_trees, f->depth, w->lw,
                           w->tstride, f->p.missing, vec4_ok, f->error_flag);
}

tahoe_status widef_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows, hipStream_t stream,
                          const float *sums_in)
{
    const tahoe_wstate *w = f->wf;
    if (!w) return fail(TAHOE_ERR_UNSUPPORTED, "the wide-row float32 form is unavailable for this forest");
    if ((rows + 7) / 8 > 0x7fffffffu) return fail(TAHOE_ERR_INVALID_ARG, "too many rows for one launch: %zu", rows);
    const int vec4_ok = (f->p.num_cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0);
    if (w->s_on && vec4_ok && rows > 0) {  // the row-streaming form: one persistent workgroup per CU
        const size_t nch = ((size_t)f->p.num_trees + 63) / 64;
        const size_t max_per = ((size_t)1 << 30) / nch;  // rows x chunks of one workgroup stay within int
        size_t grid = std::min<size_t>(rows, (size_t)std::max(f->num_cus, 1));
        grid = std::max(grid, (rows + max_per - 1) / max_per);
        const int lds = (int)ws_lds(f->p.num_cols, f->p.num_trees, w->s_lw, w->s_ts, w->s_slots);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((kWsWalkers + 2) * 64), lds, stream, data, w->simg, w->sblocks, f->inner,
                               f->leaf_orig, sums, leaf_out, sums_in, rows, f->p.num_cols, f->p.num_trees, f->depth, w->s_lw, w->s_ts,
                               w->s_img_bytes, w->s_slots, f->p.missing, f->error_flag);
        };
        if (w->s_q == 24) {
            if (leaf_out)
                go(wstream_kernel<kWsWalkers, true, 24>);
            else
                go(wstream_kernel<kWsWalkers, false, 24>);
        } else {
            if (leaf_out)
                go(wstream_kernel<kWsWalkers, true, 8>);
            else
                go(wstream_kernel<kWsWalkers, false, 8>);
        }
        TAHOE_HIP_TRY(hipGetLastError());
        return TAHOE_OK;
    }
    if (w->rt == 32) {
        if (w->nwalk == 15)
            wf_launch<32, 15>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
        else
            wf_launch<32, 12>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
    } else if (w->rt == 16) {
        if (w->nwalk == 15)
            wf_launch<16, 15>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
        else
            wf_launch<16, 12>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
    } else {
        if (w->nwalk == 15)
            wf_launch<8, 15>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
        else
            wf_launch<8, 12>(f, sums, leaf_out, data, rows, stream, sums_in, vec4_ok);
    }
    TAHOE_HIP_TRY(hipGetLastError());
    return TAHOE_OK;
}

}  // namespace tahoe
