// TILERING for wide rows, row-streaming form on 16-bit keys ("WKEY", round 3).  Replaces, like widef.hip: the walkers / kernels
// of Struct.h:953-1704 for shapes where a row does not fit shared memory; the layout is the reference's own idea for
// "lane = tree" -- node-major `reorg` arrays (Struct.h:1911-1923, walker :1035-1071) -- moved into LDS.
//
// What the tile form (widef.hip) costs on K2 (500 trees of depth 8, 3072 features; profiles/r03/pmc_k2_tileform.json): one workgroup per
// CU with LDS full, so nothing overlaps the staging of a tile; every 8-row tile re-stages the tops of all trees from L2 (2.6 x
// the bytes of the rows); and every walk gathers three 16-byte pieces of a bottom block.  The texture path is what binds
// (TD 73 % busy, HBM at 20 % of its peak): a wave-load costs it ~16 cycles + a price per distinct 128-byte line (0.5 from L1, 2
// from L2, 9 from the Infinity Cache) whatever its width (tools/ubench_td.hip, profiles/r03/ubench_td.txt).  A first
// row-streaming form on float32 rows (profiles/r03/wstream_ablation.txt) could keep only five levels of all trees resident
// beside four 12-KiB rows and needed five gathers per walk: 0.91 ms against the tile form's 0.78.
//
// Here every number that is COMPARED is a 16-bit key, so that a row takes 6 KiB and a node 4 bytes:
//   key(x) = trunc(clamp((x - lo) * scale, 0, 65534)) in float32, [lo, hi] = the range of the forest's finite thresholds,
//   scale = 65534 / (hi - lo); NaN -> 0; a missing value (|x - missing| <= 1e-6f, tested on the float32 value while it passes
//   through the loader's registers) -> 0xFFFF.  Every step is monotone in x, so key(x) > key(thr) => x > thr and
//   key(x) < key(thr) => x < thr: the branch rule of infer_one_tree (BaseTahoeTest.h:450-453) is decided by one integer
//   compare, bit for bit -- EXCEPT when the two keys are equal, which is rare (1 / 65534 of the thresholds' range apart;
//   values outside that range sit on its ends) and detected: those lanes fetch the float32 value from the batch and the
//   float32 threshold from the heap records in global memory and compare them exactly.  The thresholds' keys are computed
//   ON THE DEVICE at create, by the same instructions as the features' (host and device float arithmetic never have to
//   agree).  A NaN threshold has key 0xFFFF: never >=, never equal.  (A first version took the upper 16 bits of the float
//   -- 7 mantissa bits: 0.15 % of the compares tied, 8 % of the wave-levels ran the float32 path, and its two loads per event
//   cost the texture path as much as a full gather each: 41 % of the kernel's texture-path time, profiles/r03.)
//   * ONE persistent workgroup per CU keeps the first lw levels of ALL trees in LDS for its whole life, node-major u32 words
//     key << 16 | fid << 1 | def_left, tree stride a power of two: lane = tree, so a node read is conflict-free whatever node
//     each lane stands on, the two children of heap position p are one ds_read2st64_b32 at (p << log2(8 stride)) + lane
//     constant, and a level costs 12 vector instructions per 64 walks (the first, portable version of this loop took 45:
//     the CU issues one vector instruction per cycle, and that -- not memory -- was what bound it);
//   * kWkLoaders loader waves pull the rows of the workgroup's share of the batch through registers (16-byte loads, a whole
//     row in flight per wave), turn them into keys at once and keep the keys in registers until their slot of the ring of S
//     row slots is free;
//   * a walker wave takes item (row k, two chunks of 64 trees): lw levels from LDS, the last two levels and the leaf
//     from ONE 32-byte block per walk (two gathers; the tile form's 48-byte float32 block takes three), levels in between
//     (deep trees only) from the heap records with the float32 rule; leaf value into the workspace; the last walker of a row
//     frees its slot;
//   * the leaf values go to a workspace in global memory, leaf[row][tree] (coalesced 256-byte stores), and a summer wave of the
//     same workgroup adds them 64 rows at a time, one lane per row, in tree order: float32 sums bit-identical to predict_on_cpu
//     (BaseTahoeTest.h:462-466).  (Consumer waves fed through LDS -- the scheme of the tile kernels -- were built first: with
//     four rows in LDS only four 500-term dependent chains can be under way, one lane each, ~19 cycles per add beside 12 busy
//     waves: 0.17 of 0.65 ms whatever their number, profiles/r03/experiments.json.  The workspace costs 2 x 4 bytes per (row,
//     tree) of traffic, most of it absorbed by the 256-MiB Infinity Cache.)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "widef_internal.h"

// The 16-byte LDS-DMA (global_load_lds_dwordx4) of the image load exists on gfx950 only; nothing here has a fallback.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "wkey.hip is written for gfx950 (CDNA4): global_load_lds with 16-byte pieces, 160 KiB of LDS per workgroup"
#endif

namespace tahoe {

#ifndef TAHOE_WK_LOADERS
#define TAHOE_WK_LOADERS 4
#endif
#ifndef TAHOE_WK_WALKERS
#define TAHOE_WK_WALKERS 11
#endif
constexpr int kWkLoaders = TAHOE_WK_LOADERS;  // loader waves (= the fewest row slots)
constexpr int kWkWalkers = TAHOE_WK_WALKERS;  // walker waves; 4 loaders + 1 summer + 11 walkers = 16 waves (K2: 0.59 ms; 5 + 10: 0.60, 3 + 12: 0.62)
constexpr int kWkSumRows = 64;                // rows the summer wave adds at once, one lane each
#ifndef TAHOE_WK_CHAINS
#define TAHOE_WK_CHAINS 2
#endif
constexpr int kWkChains = TAHOE_WK_CHAINS;  // 64-tree chunks of one row a walker wave walks at once (chains per lane; 3: 0.63 ms, 4: 0.61)
constexpr int kWkSpinLimit = 1 << 22;
constexpr uint32_t kWkMissing = 0xFFFFu;
// Largest estimated share of compares with equal keys at which the create-time rule still takes this form (K2's uniform
// thresholds: 1.6e-5, i.e. 0.1 % of the wave-levels on the float32 path; 1e-4 is ~0.6 % of them).
constexpr float kWkTieLimit = 1.0e-4f;

// Timing-only ablation builds (make ABLATE=n; results are wrong on purpose; never shipped): 1 = no bottom-block gathers,
// 2 = walkers only pass the rows on, 3 = the loaders load nothing, 4 = the summer adds nothing; 5 = 2 + 4 (loaders alone),
// 6 = 3 + 4 (walkers alone).
#ifndef TAHOE_WS_ABLATE
#define TAHOE_WS_ABLATE 0
#endif
// Non-temporal hints on the two read-once / write-once streams that share L2 with the bottom blocks (make WKNT=bits; an
// experiment knob, profiles/r04/experiments.json): bit 0 = the loaders' row loads, bit 1 = the walkers' leaf-value stores and
// the summer wave's loads of them.  Results are unchanged (only the cache policy of the instructions differs).
#ifndef TAHOE_WK_NT
#define TAHOE_WK_NT 1  // measured on K2 (profiles/r04/k2_nt.txt): row loads nt 0.597 -> 0.574 ms; leaf stores + summer loads nt 0.684 (worse)
#endif
typedef float __attribute__((ext_vector_type(4))) wk_f4;  // (HIP's float4 is a struct: __builtin_nontemporal_load wants a native vector)
__device__ __forceinline__ float4 wk_load_row16(const unsigned char *p)
{
#if TAHOE_WK_NT & 1
    const wk_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wk_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}
__device__ __forceinline__ float4 wk_load_leaf16(const float *p)
{
#if TAHOE_WK_NT & 2
    const wk_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wk_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}
__device__ __forceinline__ void wk_store_leaf(float *p, float v)
{
#if TAHOE_WK_NT & 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#define WK_NO_GATHER (TAHOE_WS_ABLATE == 1)
#define WK_NO_WALK (TAHOE_WS_ABLATE == 2 || TAHOE_WS_ABLATE == 5)
#define WK_NO_LOAD (TAHOE_WS_ABLATE == 3 || TAHOE_WS_ABLATE == 6)
#define WK_NO_ADD (TAHOE_WS_ABLATE == 4 || TAHOE_WS_ABLATE == 5 || TAHOE_WS_ABLATE == 6)

// The key of a non-missing value: NaN -> 0 (v_max_f32 returns its other operand), below lo -> 0, above hi -> 65534.
__device__ __forceinline__ uint32_t wk_key(float x, float lo, float scale)
{
    return (uint32_t)fminf(fmaxf((x - lo) * scale, 0.0f), 65534.0f);
}
// feature values (in the loader); *ms collects "a missing value was seen"
__device__ __forceinline__ uint32_t wk_key_x(float x, float lo, float scale, float missing, uint64_t &ms)
{
    const bool m = fabsf(x - missing) <= kMissingEps;
    ms |= __ballot(m);
    return m ? kWkMissing : wk_key(x, lo, scale);
}
// thresholds, at create: the same instructions on the same device
__global__ void wkey_threshold_keys_kernel(const float *__restrict__ thr, uint16_t *__restrict__ keys, size_t n, float lo, float scale)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = thr[i] != thr[i] ? (uint16_t)kWkMissing : (uint16_t)wk_key(thr[i], lo, scale);
}
__device__ __forceinline__ void wk_dma16(const unsigned char *src, unsigned char *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

typedef const uint32_t __attribute__((address_space(3))) *wk_lds_u32;
typedef const uint16_t __attribute__((address_space(3))) *wk_lds_u16;
__device__ __forceinline__ uint32_t wk_lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }  // low half of a generic LDS address
// p <- 2p + (lane's bit of mask): one v_addc with the mask as carry-in
__device__ __forceinline__ uint32_t wk_descend(uint32_t p, uint64_t right_mask)
{
    uint32_t r;
    uint64_t carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=&s"(carry_out) : "v"(p), "s"(right_mask));
    return r;
}

template <int TSL, bool WRITE_LEAF>
__global__ void __launch_bounds__((kWkLoaders + 1 + kWkWalkers) * 64)
    wkey_kernel(const float *__restrict__ data, const unsigned char *__restrict__ kimg, const uint4 *__restrict__ kblocks,
                const InnerNode *__restrict__ inner, const uint32_t *__restrict__ leaf_orig, float *leafbuf,
                float *sums, const float *sums_in, uint32_t *__restrict__ leaf_out, size_t rows, int cols, int num_trees, int depth, int lw,
                int img_bytes, int S, float missing, float key_lo, float key_scale, int *__restrict__ error_flag)
{
    constexpr int NL = kWkLoaders, NW = kWkLoaders + 1 + kWkWalkers;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tv = (num_trees + 31) & ~31;  // floats per row of the workspace: whole 128-byte lines, no line shared by two rows
    const int row_bytes = cols * 4;        // of a float32 row in the batch (a multiple of 16)
    const int slot_bytes = (cols * 2 + 15) & ~15;  // of a row of keys in LDS
    unsigned char *srows = smem + img_bytes;
    uint32_t *row_ready = reinterpret_cast<uint32_t *>(srows + (size_t)S * slot_bytes);  // [S] row index + 1 whose keys are in the slot
    uint32_t *walked = row_ready + S;                                                    // [S] items walked in the slot, monotone
    uint32_t *row_ms = walked + S;                                                       // [S] != 0: the row has a missing value
    uint32_t *prog = row_ms + S;  // [kWkWalkers] items of walker w whose leaf values have reached the workspace (stores complete)
    uint32_t *abort_flag = prog + kWkWalkers;  // != 0: a wave of this workgroup gave up (error_flag is raised): everyone leaves at once
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per;
    if (r0 >= rows) return;
    const int n = (int)(rows - r0 < per ? rows - r0 : per);
    const int nch = (num_trees + 63) >> 6;  // 64-tree chunks per row
    const int nit = (nch + kWkChains - 1) / kWkChains;  // items (kWkChains chunks each) per row

    // ---- the resident tops: the image lies in global memory exactly as in LDS (LDS-DMA: 1 KiB per wave-instruction) ----
    for (int pc = wave; pc < (img_bytes >> 10); pc += NW) wk_dma16(kimg + (size_t)pc * 1024 + lane * 16, smem + (size_t)pc * 1024);
    if (tid < kWkWalkers) prog[tid] = 0u;
    if (tid == kWkWalkers) *abort_flag = 0u;
    if (tid < S) {
        row_ready[tid] = 0u;
        walked[tid] = 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (wave < NL) {
        // ================= loaders: wave j takes rows j, j + NL, ...; a row = up to three chunks of four 1-KiB pieces =================
        // Chunk c of a row always uses register set c: the raw floats are turned into keys (kept in registers: half the raw
        // bytes) as soon as they land, and the next row's chunk c is requested into the freed set at once -- up to a whole
        // row in flight per wave.  Only then does the wave wait for its slot: what is left on the critical path behind a freed
        // slot is a dozen ds_write_b64, not the conversion (that wait used to cost as much as the walk of the row).
        const int cpr = (row_bytes + 4095) >> 12;  // chunks per row, <= 3 (the host checks num_cols <= 3072: 128 VGPRs per lane)
        bool dead = false;
        const int k_last = wave + ((n - 1 - wave) / NL) * NL;  // this wave's last row (n > wave is checked below)
        auto load = [&](float4(&r)[4], int k, int c) {
            // (a row past this wave's last one: that row again, unused -- a branch around the loads would make hipcc drain
            // all of them at the next use; pieces past the end of the row re-read its last 16 bytes for the same reason)
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + (r0 + min(k, k_last)) * (size_t)cols);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = min((c * 4 + u) * 1024 + lane * 16, row_bytes - 16);
                r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#if !WK_NO_LOAD
                r[u] = wk_load_row16(src + off);
#endif
            }
        };
        uint64_t ms_seen = 0;  // a missing value among the keys of the row under way (any lane)
        auto convert = [&](uint2(&kk)[4], const float4(&r)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k0 = wk_key_x(r[u].x, key_lo, key_scale, missing, ms_seen);
                const uint32_t k1 = wk_key_x(r[u].y, key_lo, key_scale, missing, ms_seen);
                const uint32_t k2 = wk_key_x(r[u].z, key_lo, key_scale, missing, ms_seen);
                const uint32_t k3 = wk_key_x(r[u].w, key_lo, key_scale, missing, ms_seen);
                kk[u] = make_uint2(k0 | (k1 << 16), k2 | (k3 << 16));
            }
        };
        auto write = [&](const uint2(&kk)[4], unsigned char *dst, int c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = (c * 4 + u) * 1024 + lane * 16;
                if (off < row_bytes) *reinterpret_cast<uint2 *>(dst + (off >> 1)) = kk[u];
            }
        };
        if (wave >= n) return;
        float4 raw0[4], raw1[4], raw2[4];
        uint2 key0[4], key1[4], key2[4];
        load(raw0, wave, 0);
        if (cpr > 1) load(raw1, wave, 1);
        if (cpr > 2) load(raw2, wave, 2);
        for (int k = wave; k < n && !dead; k += NL) {
            convert(key0, raw0);
            load(raw0, k + NL, 0);
            if (cpr > 1) {
                convert(key1, raw1);
                load(raw1, k + NL, 1);
            }
            if (cpr > 2) {
                convert(key2, raw2);
                load(raw2, k + NL, 2);
            }
            const int slot = k % S;
            if (k >= S) {  // every earlier row of this slot walked?
                const uint32_t need = (uint32_t)nit * (uint32_t)(k / S);
                int spins = 0;
                while (lds_flag_load(&walked[slot]) < need) {
                    if (++spins > kWkSpinLimit || lds_flag_load(abort_flag) != 0u) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead) break;
            }
            unsigned char *dst = srows + (size_t)slot * slot_bytes;
            write(key0, dst, 0);
            if (cpr > 1) write(key1, dst, 1);
            if (cpr > 2) write(key2, dst, 2);
            if (lane == 0) lds_flag_store(&row_ms[slot], ms_seen != 0ull ? 1u : 0u);
            ms_seen = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the keys are in LDS
            if (lane == 0) lds_flag_store(&row_ready[slot], (uint32_t)(k + 1));
        }
        if (dead && lane == 0) {
            lds_flag_store(abort_flag, 1u);
            atomicOr(error_flag, 1);
        }
        return;
    }

    if (wave == NL) {
        // ================= summer: rows [64 j, 64 j + 64) of this workgroup, one lane each, once all their leaf values are in the
        // workspace: the row's values in tree order, float32 -- the sequential sum of predict_on_cpu, 64 rows per add instruction
        // (consumer waves fed through LDS can only have as many chains under way as there are rows in LDS: four).  A lane sweeps
        // whole 128-byte lines that no other row shares and that this CU has not read before: nothing stale can sit in L1.
        if (!leafbuf) return;
        bool dead = false;
        for (int g0 = 0; g0 < n && !dead; g0 += kWkSumRows) {
            const int g1 = min(n, g0 + kWkSumRows);
            const int need_items = g1 * nit;  // items 0 .. need_items - 1 stored; item i is walker (i % NWALK)'s (i / NWALK)-th
            const int w = min(lane, kWkWalkers - 1);
            const uint32_t need = need_items > w ? (uint32_t)((need_items - w + kWkWalkers - 1) / kWkWalkers) : 0u;
            int spins = 0;
            for (;;) {
                const bool ok = lds_flag_load(&prog[w]) >= need;
                if (__ballot(ok) == ~0ull) break;
                if (++spins > kWkSpinLimit || lds_flag_load(abort_flag) != 0u) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (dead) break;
            // Hand-over through global memory INSIDE one workgroup: a walker stores its leaf values, executes a workgroup-scope
            // release fence (s_waitcnt vmcnt(0): the stores are complete, i.e. written through this CU's L1 to L2), then raises
            // prog[]; this wave reads prog[], executes the workgroup-scope acquire fence below, then loads.  What makes that
            // sufficient is the memory model of the target, not the access history: the waves of a workgroup run on one CU and
            // share its vector L1 (the library is never built in threadgroup-split mode), so a line one of them wrote cannot be
            // stale for another -- LLVM's AMDGPU memory model needs no cache invalidation for workgroup scope for exactly that
            // reason, and the fence builtin emits whatever the compile target does need.  It also keeps the compiler from
            // moving or merging the loads below across the flag loads (the former code relied on LLVM not speculating them).
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            TAHOE_LDS_ACQUIRE();
            const int k = g0 + lane;
            if (k < g1) {
                const float *v = leafbuf + (r0 + k) * (size_t)tv;
                float sum = sums_in ? sums_in[r0 + k] : 0.0f;
                int t = 0;
#if !WK_NO_ADD
                for (; t + 8 <= num_trees; t += 8) {
                    const float4 p = wk_load_leaf16(v + t), q = wk_load_leaf16(v + t + 4);
                    sum += p.x;
                    sum += p.y;
                    sum += p.z;
                    sum += p.w;
                    sum += q.x;
                    sum += q.y;
                    sum += q.z;
                    sum += q.w;
                }
                for (; t < num_trees; ++t) sum += v[t];
#endif
                sums[r0 + k] = sum;
            }
        }
        if (dead && lane == 0) {
            lds_flag_store(abort_flag, 1u);
            atomicOr(error_flag, 1);
        }
        return;
    }

    // ================= walkers: item = (row k, kWkChains chunks of 64 trees); chain j of a lane = tree (c * KC + j) * 64 + lane =========
    constexpr int KC = kWkChains;
    const size_t n_inner = ((size_t)1 << depth) - 1;
    const uint32_t n_blocks = 1u << (depth - 2);
    const uint32_t first_block_node = n_blocks - 1;
    if (wk_lds_addr(smem) != 0u) {  // the integer LDS addresses below take the tops at LDS address 0 (true without static LDS)
        if (lane == 0) {
            lds_flag_store(abort_flag, 1u);  // loaders and summer leave at once instead of waiting out their spin limits
            atomicOr(error_flag, 2);
        }
        return;
    }
    const uint32_t xbase0 = (uint32_t)img_bytes;
    bool dead = false;
    const int wid = wave - NL - 1;  // walker index
    int k = 0, c = wid;
    uint32_t items_done = 0;  // items of this walker whose leaf values are stored (counted once their stores are complete)
    while (c >= nit) {
        c -= nit;
        ++k;
    }
    auto walk_item = [&](auto ms_tag, int k, int c, int slot) {
        constexpr bool MS = decltype(ms_tag)::value;
        const uint32_t xbase = xbase0 + (uint32_t)slot * (uint32_t)slot_bytes;  // LDS address of the row's keys
        const float *xrow = data + (r0 + k) * (size_t)cols;  // the float32 row: equal keys and the levels between tops and blocks
        int t[KC], tt[KC];
        const InnerNode *tree[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            t[j] = (c * KC + j) * 64 + lane;
            tt[j] = min(t[j], num_trees - 1);  // lanes past the last tree repeat it, unused
            tree[j] = inner + (size_t)tt[j] * n_inner;
        }
        // right <=> (missing ? !def_left : x >= thr) as a wave mask, on keys; equal keys are decided on the float32 values
        // (heap0 = the node's 0-based heap index, for its float32 threshold)
        auto right_mask = [&](uint32_t node, uint32_t key_x, const InnerNode *tr, uint32_t heap0) -> uint64_t {
            const uint64_t gt = __builtin_amdgcn_uicmp(key_x, node >> 16, 34 /* ICMP_UGT */);
            uint64_t tie = __builtin_amdgcn_uicmp(key_x, node >> 16, 32 /* ICMP_EQ */);
            uint64_t right = gt;
            if (MS) {
                const uint64_t ms = __builtin_amdgcn_uicmp(key_x, kWkMissing, 32 /* ICMP_EQ */);
                const uint64_t ndl = __builtin_amdgcn_uicmp(node & 1u, 0u, 32 /* ICMP_EQ: def_left clear */);
                right = (gt & ~ms) | (ms & ndl);
                tie &= ~ms;
            }
            if (tie != 0ull) {  // rare: the two values fall into the same 1 / 65534 of the thresholds' range
                bool ge = false;
                if (__builtin_amdgcn_inverse_ballot_w64(tie)) ge = xrow[(node >> 1) & 0x7fffu] >= tr[heap0].thr;
                right = (right & ~tie) | (__ballot(ge) & tie);
            }
            return right;
        };
        auto xkey = [&](uint32_t node) -> uint32_t { return *reinterpret_cast<wk_lds_u16>((node & 0xFFFEu) + xbase); };
        // Chains are walked in groups of GS = 2: the LDS levels of a group interleave its chains (one LDS round trip per level
        // and two walks under way); with four chains the bottom-block gathers of the first group fly under the LDS levels of
        // the second.
        constexpr int GS = KC < 2 ? KC : 2;
        static_assert(KC == 1 || KC == 2 || KC == 4, "chains per walker: 1, 2 or 4");
        uint32_t p[KC];  // 1-based heap position per chain
        float leaf[KC];
        uint32_t bsel[KC], c0[KC], c1[KC], idx[KC];
        uint4 qa[KC], qb[KC];
        auto top = [&](auto j0c) {  // the resident levels of chains j0 .. j0 + GS - 1
            constexpr int J0 = decltype(j0c)::value;
            uint32_t node[GS], cbase[GS];  // node word, lane constant of the child-pair address
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                p[J0 + g] = 1;
                cbase[g] = 4u * (uint32_t)tt[J0 + g] - (4u << TSL);  // children of p: position 2p at ((2p - 1) << TSL) + tt dwords
                node[g] = lw > 0 ? *reinterpret_cast<wk_lds_u32>(4u * (uint32_t)tt[J0 + g]) : 0u;
            }
#if WK_NO_WALK
            if (false)
#endif
            if (lw > 0) {
                for (int l = 0; l < lw - 1; ++l) {
                    // one LDS round trip per level: the feature key of the node and BOTH children (one tree stride apart)
                    uint32_t kx[GS], nl[GS], nr[GS];
#pragma unroll
                    for (int g = 0; g < GS; ++g) {
                        kx[g] = xkey(node[g]);
                        const wk_lds_u32 cp = reinterpret_cast<wk_lds_u32>((p[J0 + g] << (TSL + 3)) + cbase[g]);
                        nl[g] = cp[0];
                        nr[g] = cp[1 << TSL];
                    }
#pragma unroll
                    for (int g = 0; g < GS; ++g) {
                        const uint64_t r = right_mask(node[g], kx[g], tree[J0 + g], p[J0 + g] - 1u);
                        p[J0 + g] = wk_descend(p[J0 + g], r);
                        node[g] = __builtin_amdgcn_inverse_ballot_w64(r) ? nr[g] : nl[g];
                    }
                }
#pragma unroll
                for (int g = 0; g < GS; ++g)
                    p[J0 + g] = wk_descend(p[J0 + g], right_mask(node[g], xkey(node[g]), tree[J0 + g], p[J0 + g] - 1u));
            }
        };
        auto gather = [&](auto j0c) {  // levels between tops and blocks (deep trees only), then the bottom block's two pieces
            constexpr int J0 = decltype(j0c)::value;
#pragma unroll
            for (int j = J0; j < J0 + GS; ++j) {
#if WK_NO_GATHER || WK_NO_WALK
                bsel[j] = c0[j] = c1[j] = idx[j] = 0;
                leaf[j] = __uint_as_float(p[j]);
#else
                idx[j] = p[j] - 1u;  // 0-based heap index on level lw
                for (int l = lw; l < depth - 2; ++l) {  // the float32 rule on the heap records in global memory
                    const InnerNode nd = tree[j][idx[j]];
                    idx[j] = 2u * idx[j] + 1u + go_right(xrow[nd.meta & kMetaFidMask], nd.thr, (nd.meta >> 31) != 0u, missing);
                }
                // the last two levels and the leaf: {n0, n1, n2, leaf0} {leaf1, leaf2, leaf3, -}, one 32-byte block per walk
                bsel[j] = idx[j] - first_block_node;
                const uint4 *bp = kblocks + ((size_t)tt[j] * n_blocks + bsel[j]) * 2;
                qa[j] = bp[0];
                qb[j] = bp[1];
#endif
            }
        };
        auto bottom = [&](auto j0c) {
            constexpr int J0 = decltype(j0c)::value;
#if !(WK_NO_GATHER || WK_NO_WALK)
#pragma unroll
            for (int j = J0; j < J0 + GS; ++j) {
                const bool r0b = __builtin_amdgcn_inverse_ballot_w64(right_mask(qa[j].x, xkey(qa[j].x), tree[j], idx[j]));
                c0[j] = r0b ? 1u : 0u;
                const uint32_t n1 = r0b ? qa[j].z : qa[j].y;
                const bool r1b = __builtin_amdgcn_inverse_ballot_w64(right_mask(n1, xkey(n1), tree[j], 2u * idx[j] + 1u + c0[j]));
                c1[j] = r1b ? 1u : 0u;
                const uint32_t lo = r0b ? qb[j].y : qa[j].w, hi = r0b ? qb[j].z : qb[j].x;
                leaf[j] = __uint_as_float(r1b ? hi : lo);
            }
#endif
        };
        top(std::integral_constant<int, 0>{});
        gather(std::integral_constant<int, 0>{});
        if constexpr (KC > GS) {
            top(std::integral_constant<int, GS>{});
            gather(std::integral_constant<int, GS>{});
        }
        bottom(std::integral_constant<int, 0>{});
        if constexpr (KC > GS) bottom(std::integral_constant<int, GS>{});
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            if (leafbuf && t[j] < num_trees) wk_store_leaf(&leafbuf[(r0 + k) * (size_t)tv + t[j]], leaf[j]);
            if (WRITE_LEAF) {
                if (t[j] < num_trees)
                    leaf_out[(r0 + k) * (size_t)num_trees + t[j]] =
                        leaf_orig[(size_t)t[j] * ((size_t)n_blocks * 4) + (size_t)bsel[j] * 4 + 2 * c0[j] + c1[j]];
            }
        }
    };
    while (k < n && !dead) {
        const int slot = k % S;
        {   // the row's keys are in LDS
            int spins = 0;
            while (lds_flag_load(&row_ready[slot]) != (uint32_t)(k + 1)) {
                if (++spins > kWkSpinLimit || lds_flag_load(abort_flag) != 0u) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            TAHOE_LDS_ACQUIRE();
        }
        if (leafbuf && items_done != 0u) {  // the previous item's leaf values have reached the workspace: tell the summer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the stores are complete before the flag
            if (lane == 0) lds_flag_store(&prog[wid], items_done);
        }
        if (lds_flag_load(&row_ms[slot]) != 0u)
            walk_item(std::true_type{}, k, c, slot);
        else
            walk_item(std::false_type{}, k, c, slot);
        TAHOE_LDS_RELEASE();  // the row's last read precedes the counter (in-order LDS)
        if (lane == 0) atomicAdd(&walked[slot], 1u);
        ++items_done;
        c += kWkWalkers;
        while (c >= nit) {
            c -= nit;
            ++k;
        }
    }
    if (leafbuf) {  // the last item's leaf values; a walker that never had an item reports what the summer expects of it: nothing
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_flag_store(&prog[wid], items_done);
    }
    if (dead && lane == 0) {
        lds_flag_store(abort_flag, 1u);
        atomicOr(error_flag, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
static long long wk_img_bytes(int lw, int ts) { return ((((1LL << lw) - 1) * ts * 4) + 1023) & ~1023LL; }
static long long wk_lds(int cols, int num_trees, int lw, int ts, int slots)
{
    return wk_img_bytes(lw, ts) + (long long)slots * ((cols * 2 + 15) & ~15) + (3LL * slots + kWkWalkers) * 4 + 16;
}

long long wkey_lds_bytes(const tahoe_forest *f)
{
    const tahoe_wstate *w = f->wf;
    return w ? wk_lds(f->p.num_cols, f->p.num_trees, w->s_lw, w->s_ts, w->s_slots) : 0;
}

void wkey_free(tahoe_wstate *w)
{
    if (w->kimg) (void)hipFree(w->kimg);
    if (w->kblocks) (void)hipFree(w->kblocks);
    if (w->leafbuf) (void)hipFree(w->leafbuf);
    w->leafbuf = nullptr;
    w->leaf_rows = 0;
    w->kimg = nullptr;
    w->kblocks = nullptr;
}

static uint32_t wk_node_word(const InnerNode &nd, uint16_t key)
{
    return ((uint32_t)key << 16) | ((nd.meta & 0x7fffu) << 1) | (nd.meta >> 31);
}

// Builds the image of the tops and the 32-byte bottom blocks when the shape suits the form: rows of 16-byte multiples, feature
// ids of 15 bits, and LDS for the tops of all trees down to the bottom blocks (levels in between would be walked on float32
// values from global memory: correct, tested with TAHOE_WSTREAM=1, but slower than the tile form) beside >= kWkLoaders row slots.
tahoe_status wkey_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                        const std::vector<float> &h_leaf)
{
    tahoe_wstate *w = f->wf;
    if (!w) return TAHOE_OK;
    const int cols = f->p.num_cols, De = f->depth;
    const size_t T = (size_t)f->p.num_trees, n_inner = f->n_inner, n_leaf = f->n_leaf;
    int knob = -1;  // TAHOE_WSTREAM: 0 = never, 1 = whenever it can be built (experiments, tests), unset = the shape rule
    if (const char *e = getenv("TAHOE_WSTREAM")) knob = atoi(e);
    if (knob == 0 || cols % 4 != 0 || cols > 3072 || T > ((size_t)1 << 20) || De < 2) return TAHOE_OK;
    int s_lw = -1, s_ts = 64;  // tree stride of the image: a power of two, 64 .. 1024 (the walk's shifts are immediates)
    while (s_ts < (int)T) s_ts *= 2;
    if (s_ts > 1024) return TAHOE_OK;
    for (int l = std::min(De - 2, 10); l >= 0 && s_lw < 0; --l)
        if (wk_lds(cols, (int)T, l, s_ts, kWkLoaders) <= f->lds_limit) s_lw = l;
    if (s_lw < 0) return TAHOE_OK;
    // The shape rule (tools/selector_wide.py, profiles/r03/selector_wide.json: the faster form of TILERING on 7 of the 8 shapes
    // it can serve, 3 % behind on the eighth): every level above the bottom blocks resident, and at most a tree per three
    // features -- more trees per row than that and the tile form's staged tops, shared by the rows of a tile, cost less than
    // this form's two gathers per walk.
    if (knob != 1 && (s_lw != De - 2 || 3 * T > (size_t)cols)) return TAHOE_OK;
    int slots = kWkLoaders;
    while (slots < 16 && wk_lds(cols, (int)T, s_lw, s_ts, slots + 1) <= f->lds_limit) ++slots;
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "wkey_build: %s failed: %s", what, hipGetErrorString(e)); };
    // ---- the key map: [lo, hi] = range of the finite thresholds of the nodes that exist (the padding below an early leaf
    // carries thr = 0.0 and decides nothing: it neither stretches the range nor ever ties, see below); keys from the device ----
    float lo = 0.f, hi = 0.f;
    bool any = false;
    for (size_t i = 0; i < h_inner.size(); ++i) {
        const InnerNode &nd = h_inner[i];
        if (h_real[i] && std::isfinite(nd.thr)) {
            lo = any ? std::min(lo, nd.thr) : nd.thr;
            hi = any ? std::max(hi, nd.thr) : nd.thr;
            any = true;
        }
    }
    float scale = (any && hi > lo) ? 65534.0f / (hi - lo) : 0.f;
    if (!std::isfinite(scale)) scale = 0.f;  // (everything then ties and is decided on the float32 values: slow, correct)
    std::vector<uint16_t> h_keys(h_inner.size());
    {
        std::vector<float> h_thr(h_inner.size());
        for (size_t i = 0; i < h_inner.size(); ++i) h_thr[i] = h_inner[i].thr;
        float *d_thr = nullptr;
        uint16_t *d_keys = nullptr;
        const size_t nn = std::max<size_t>(h_thr.size(), 1);
        if ((e = hipMalloc(reinterpret_cast<void **>(&d_thr), nn * sizeof(float))) != hipSuccess) return bad("hipMalloc");
        if ((e = hipMalloc(reinterpret_cast<void **>(&d_keys), nn * sizeof(uint16_t))) != hipSuccess) {
            (void)hipFree(d_thr);
            return bad("hipMalloc");
        }
        e = hipMemcpy(d_thr, h_thr.data(), h_thr.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess && !h_thr.empty()) {
            hipLaunchKernelGGL(wkey_threshold_keys_kernel, dim3((unsigned)((h_thr.size() + 255) / 256)), dim3(256), 0, 0, d_thr, d_keys,
                               h_thr.size(), lo, scale);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(h_keys.data(), d_keys, h_keys.size() * sizeof(uint16_t), hipMemcpyDeviceToHost);
        (void)hipFree(d_thr);
        (void)hipFree(d_keys);
        if (e != hipSuccess) return bad("threshold keys");
    }
    // Padding below an early leaf: both subtrees are the same leaf, so the branch is irrelevant -- key 0xFFFF never compares
    // greater and never ties (a missing x, the only other 0xFFFF, is taken out of the tie mask by the missing rule).
    for (size_t i = 0; i < h_inner.size(); ++i)
        if (!h_real[i]) h_keys[i] = (uint16_t)kWkMissing;
    // ---- how fine is ONE affine map for this forest?  A compare costs two divergent global loads when the feature's key
    // equals the threshold's (8 % of the wave-levels on that path cost 41 % of the texture path's time, header).  With one
    // [lo, hi] for all features, a feature whose thresholds span few key cells -- a small-scale feature beside a large-scale
    // one, or any forest with one outlier threshold -- ties often: if its values spread like its thresholds, a compare ties
    // with probability ~ 1 / (cells its thresholds span).  Estimate: that quantity per node, weighted by the node's reach
    // probability 2^-level, over the features that have at least two distinct keys.  Above kWkTieLimit the tile form
    // (widef.hip) serves TILERING instead; TAHOE_WSTREAM=1 still forces this form (tests), results are exact either way.
    {
        std::vector<uint32_t> kmin((size_t)cols, 0xFFFFu), kmax((size_t)cols, 0u);
        std::vector<float> tmin((size_t)cols, INFINITY), tmax((size_t)cols, -INFINITY);
        for (size_t i = 0; i < h_inner.size(); ++i) {
            if (!h_real[i] || !std::isfinite(h_inner[i].thr)) continue;
            const size_t fid = h_inner[i].meta & kMetaFidMask;
            if (fid >= (size_t)cols) continue;
            kmin[fid] = std::min<uint32_t>(kmin[fid], h_keys[i]);
            kmax[fid] = std::max<uint32_t>(kmax[fid], h_keys[i]);
            tmin[fid] = std::min(tmin[fid], h_inner[i].thr);
            tmax[fid] = std::max(tmax[fid], h_inner[i].thr);
        }
        double num = 0.0, den = 0.0;
        for (size_t t = 0; t < T; ++t)
            for (size_t i = 0; i < n_inner; ++i) {
                const size_t at = t * n_inner + i;
                if (!h_real[at] || !std::isfinite(h_inner[at].thr)) continue;
                const size_t fid = h_inner[at].meta & kMetaFidMask;
                if (fid >= (size_t)cols || !(tmax[fid] > tmin[fid])) continue;  // one distinct threshold: nothing to judge the scale by
                int level = 0;
                for (size_t v = i + 1; v > 1; v >>= 1) ++level;
                const double wgt = std::ldexp(1.0, -level);
                // (distinct thresholds that collapse into one key cell: every compare of that feature ties)
                num += wgt / (double)std::max<uint32_t>(kmax[fid] - kmin[fid], 1u);
                den += wgt;
            }
        w->key_tie_estimate = den > 0.0 ? (float)(num / den) : (scale > 0.f ? 0.f : 1.f);
    }
    if (knob != 1 && w->key_tie_estimate > kWkTieLimit) return TAHOE_OK;  // s_on stays false: the tile form runs
    const size_t nlv = ((size_t)1 << s_lw) - 1;
    const size_t img = (size_t)wk_img_bytes(s_lw, s_ts);
    std::vector<uint32_t> h_img(img / 4, 0u);
    for (size_t t = 0; t < T; ++t)
        for (size_t i = 0; i < nlv; ++i) h_img[i * (size_t)s_ts + t] = wk_node_word(h_inner[t * n_inner + i], h_keys[t * n_inner + i]);
    const size_t n_blocks = (size_t)1 << (De - 2), first = n_blocks - 1;
    std::vector<uint4> h_kb(T * n_blocks * 2);
    parallel_for(T, 8, [&](size_t t_lo, size_t t_hi) {
        for (size_t t = t_lo; t < t_hi; ++t) {
            const InnerNode *in = &h_inner[t * n_inner];
            const uint16_t *ky = &h_keys[t * n_inner];
            for (size_t b = 0; b < n_blocks; ++b) {
                const size_t r = first + b;
                uint32_t lv[4];
                memcpy(lv, &h_leaf[t * n_leaf + 4 * b], 16);
                h_kb[(t * n_blocks + b) * 2 + 0] = make_uint4(wk_node_word(in[r], ky[r]), wk_node_word(in[2 * r + 1], ky[2 * r + 1]),
                                                              wk_node_word(in[2 * r + 2], ky[2 * r + 2]), lv[0]);
                h_kb[(t * n_blocks + b) * 2 + 1] = make_uint4(lv[1], lv[2], lv[3], 0u);
            }
        }
    });
    auto up = [&](auto **dst, const auto &src) {
        const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(src[0]);
        hipError_t er = hipMalloc(reinterpret_cast<void **>(dst), bytes);
        if (er != hipSuccess) return er;
        f->device_bytes += bytes;
        return src.empty() ? hipSuccess : hipMemcpy(*dst, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice);
    };
    uint32_t *d_img = nullptr;
    if ((e = up(&d_img, h_img)) != hipSuccess) return bad("kimg");
    w->kimg = reinterpret_cast<unsigned char *>(d_img);
    if ((e = up(&w->kblocks, h_kb)) != hipSuccess) return bad("kblocks");
    w->s_lw = s_lw;
    w->s_ts = s_ts;
    w->s_slots = slots;
    w->s_img_bytes = (int)img;
    w->slab_bytes = (size_t)1 << 30;
    if (const char *sm = getenv("TAHOE_WSTREAM_SLAB_MB")) w->slab_bytes = (size_t)std::max(atoi(sm), 1) << 20;
    w->key_lo = lo;
    w->key_scale = scale;
    w->s_on = true;
    for (const void *kern : {(const void *)&wkey_kernel<6, false>, (const void *)&wkey_kernel<6, true>, (const void *)&wkey_kernel<7, false>,
                             (const void *)&wkey_kernel<7, true>, (const void *)&wkey_kernel<8, false>, (const void *)&wkey_kernel<8, true>,
                             (const void *)&wkey_kernel<9, false>, (const void *)&wkey_kernel<9, true>, (const void *)&wkey_kernel<10, false>,
                             (const void *)&wkey_kernel<10, true>})
        if ((e = allow_max_lds(kern, f->lds_limit)) != hipSuccess) return bad("hipFuncSetAttribute");
    return TAHOE_OK;
}

// The leaf-value workspace, [rows][trees rounded up to 32] floats (whole 128-byte lines per row), grow-only like the quantised
// forms' (tahoe_forest_reserve).  At most 1 GiB (TAHOE_WSTREAM_SLAB_MB, read at create): a larger batch is walked in slabs of
// rows, one launch after the other on the stream, each re-using the workspace.
static size_t wk_slab_rows(const tahoe_forest *f)
{
    const size_t tv = ((size_t)f->p.num_trees + 31) & ~(size_t)31;
    return std::max<size_t>(f->wf->slab_bytes / (tv * sizeof(float)), (size_t)std::max(f->num_cus, 1) * 64);
}

tahoe_status wkey_reserve(tahoe_forest *f, size_t rows)
{
    tahoe_wstate *w = f->wf;
    if (!w || !w->s_on) return TAHOE_OK;
    rows = std::min(rows, wk_slab_rows(f));
    if (rows <= w->leaf_rows) return TAHOE_OK;
    const size_t tv = ((size_t)f->p.num_trees + 31) & ~(size_t)31;
    if (w->leafbuf) {
        TAHOE_HIP_TRY(hipDeviceSynchronize());  // a previous launch may still use the old buffer
        TAHOE_HIP_TRY(hipFree(w->leafbuf));
        f->device_bytes -= w->leaf_rows * tv * sizeof(float);
        w->leafbuf = nullptr;
        w->leaf_rows = 0;
    }
    TAHOE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w->leafbuf), rows * tv * sizeof(float)));
    w->leaf_rows = rows;
    f->device_bytes += rows * tv * sizeof(float);
    return TAHOE_OK;
}

tahoe_status wkey_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows, hipStream_t stream,
                         const float *sums_in)
{
    tahoe_wstate *w = f->wf;
    if (sums) {
        const tahoe_status rs = wkey_reserve(f, rows);
        if (rs != TAHOE_OK) return rs;
    }
    const size_t nch = ((size_t)f->p.num_trees + 63) / 64;
    const size_t max_per = ((size_t)1 << 30) / nch;  // rows x chunks of one workgroup stay within int
    const int lds = (int)wkey_lds_bytes(f);
    const dim3 block((kWkLoaders + 1 + kWkWalkers) * 64);
    float *leafbuf = sums ? w->leafbuf : nullptr;
    const size_t slab = sums ? wk_slab_rows(f) : rows;
    for (size_t lo = 0; lo < rows; lo += slab) {
        const size_t n = std::min(slab, rows - lo);
        size_t grid = std::min<size_t>(n, (size_t)std::max(f->num_cus, 1));  // one persistent workgroup per CU
        grid = std::max(grid, (n + max_per - 1) / max_per);
        const float *d = data + lo * (size_t)f->p.num_cols;
        float *so = sums ? sums + lo : nullptr;
        const float *si = sums_in ? sums_in + lo : nullptr;
        uint32_t *lf = leaf_out ? leaf_out + lo * (size_t)f->p.num_trees : nullptr;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), block, lds, stream, d, w->kimg, w->kblocks, f->inner, f->leaf_orig, leafbuf, so, si, lf, n,
                               f->p.num_cols, f->p.num_trees, f->depth, w->s_lw, w->s_img_bytes, w->s_slots, f->p.missing, w->key_lo,
                               w->key_scale, f->error_flag);
        };
        switch (w->s_ts) {
        case 64: leaf_out ? go(wkey_kernel<6, true>) : go(wkey_kernel<6, false>); break;
        case 128: leaf_out ? go(wkey_kernel<7, true>) : go(wkey_kernel<7, false>); break;
        case 256: leaf_out ? go(wkey_kernel<8, true>) : go(wkey_kernel<8, false>); break;
        case 512: leaf_out ? go(wkey_kernel<9, true>) : go(wkey_kernel<9, false>); break;
        default: leaf_out ? go(wkey_kernel<10, true>) : go(wkey_kernel<10, false>); break;
        }
        TAHOE_HIP_TRY(hipGetLastError());
    }
    return TAHOE_OK;
}

}  // namespace tahoe
