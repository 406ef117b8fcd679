// Shared between quantize.hip (float32 rows -> u16 rank codes) and qring.hip (the walk on those codes):
// the per-group tables and views, constants of the tile layout, and the entry points each file offers the other.
#ifndef TAHOE_AMD_QRING_INTERNAL_H
#define TAHOE_AMD_QRING_INTERNAL_H

#include <algorithm>
#include <cmath>
#include <cstdint>
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
    int max_count = 0;            // distinct thresholds of the busiest feature (codes run 0 .. max_count)
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
    bool reg = false;             // ... in its region form: fid << 7, tiles of two or three 64-row regions (qring.hip)
    bool sparse = false;          // the handle is a sparse forest: tables, workspace and region tiles only (sparse.hip walks)
    bool narrow128 = false;       // region form, num_cols <= 128: regions at a 16-KiB stride -- six chains of u16 codes (384-row tiles), or
                                  // six chains of u8 codes with 15 walkers and a ring of 24
    bool code8 = false;           // region form with every table <= 254 entries: large batches are quantised to u8 codes and walked
                                  // in 384-row tiles (six chains per lane)
    int wide_rt = 0;              // rows per tile of the wide-row form (qwide_kernel), fixed at create; 0 = not used
    int wide_lw = 0;              // ... and the top levels its LDS slots hold
    int wide_kg = 1;              // ... and the groups of trees a walker wave holds (chains per lane): 1 or 3
    std::vector<tahoe_qgroup> groups;
    uint16_t *xq = nullptr;       // workspace: quantised tiles (re-used by every group)
    size_t xq_rows = 0;           // rows the workspace holds
    float *leafbuf = nullptr;     // workspace of the SPLIT form: [tree][row] leaf values of one tree group
    size_t leaf_stride = 0, leaf_trees = 0;
    uint32_t *chunk_flags = nullptr;  // workspace: per 2^cshift rows (the rows of one quantise workgroup), "a missing value was seen"
    size_t n_chunk_flags = 0;
};
namespace tahoe {


constexpr int kQRows = 128;                 // rows per tile
// Ring depth and consumer batch, tuned on K3 (tools/ablate_walk.sh style builds): 16/8 4.20 ms, 16/4 4.11, 24/4 4.08,
// 32/4 4.14, 24/6 4.10, 24/3 4.28; a batch of 2 or 1 makes the consumer's polling the bottleneck (4.9 / 6.6 ms).
#ifndef TAHOE_QRING_RING
#define TAHOE_QRING_RING 24
#endif
#ifndef TAHOE_QRING_BATCH
#define TAHOE_QRING_BATCH 4
#endif
constexpr int kQRing = TAHOE_QRING_RING;    // ring entries (trees)
constexpr int kQBatch = TAHOE_QRING_BATCH;  // trees the consumer takes per poll
constexpr int kQSpinLimit = 1 << 22;
constexpr int kQSlotBytes = 4096;            // LDS per walker: a 10-level top (2^10 u32)
// Region form (NARROW, num_cols <= 256): a tile is K regions of 64 rows, each [fid][64] u16 at a multiple of 32 KiB in LDS.
// K = 3: 14 walkers x 3 chains and a ring of 10 (96 + 56 + 7.5 KiB = 163,372 B of the 163,840); K = 2: 15 walkers x 2
// chains, ring of 24.  Throughput follows the walker count (13 walkers / ring 15: 3.80 ms on K3, 14 / 10: 3.71, 15 / 5: 3.73):
// a ring shorter than the walker count only makes a walker that finishes early wait for the trees before its own.
constexpr int kRegRows = 64;
constexpr int kRegBytes = 32768;
#ifndef TAHOE_R3_WALKERS
#define TAHOE_R3_WALKERS 14
#endif
#ifndef TAHOE_R3_RING
#define TAHOE_R3_RING 10
#endif
constexpr int kReg3Walkers = TAHOE_R3_WALKERS;
constexpr int kReg3Ring = TAHOE_R3_RING;
constexpr int kQMaxTable = 32767;
// LDS of the region form: K regions of 32 KiB, walker slots, ring
inline long long qreg_lds_for(int k, int nwalk, int ring, bool code8 = false, int regb = kRegBytes)
{
    return (long long)(code8 ? k / 2 : k) * regb + (long long)nwalk * kQSlotBytes + (long long)ring * k * kRegRows * 4 + (ring + 1) * 4LL;
}
// u8 form: six chains (384-row tiles of three 128-row regions) for whole waves of workgroups, two chains (one region) for
// the remainder; walkers / ring / consumer batch of the 384-row tile: 14 / 5 / 2 (96 KiB + 14 x 4 KiB + 5 x 1.5 KiB = 163,352 B); KR3
// (profiles/r04/tune_q8.txt): 13 / 7 / 3 2.98 ms, 14 / 5 / 2 2.79, 12 / 10 / 4 2.96, 12 / 10 / 5 2.94
#ifndef TAHOE_Q8_WALKERS
#define TAHOE_Q8_WALKERS 14
#endif
#ifndef TAHOE_Q8_RING
#define TAHOE_Q8_RING 5
#endif
#ifndef TAHOE_Q8_BATCH
#define TAHOE_Q8_BATCH 2
#endif
constexpr int kReg8Walkers = TAHOE_Q8_WALKERS, kReg8Ring = TAHOE_Q8_RING, kReg8Batch = TAHOE_Q8_BATCH;
// top walk of the 384-row u8 tile / of the 192-row u16 tile: 1 = only the chosen child is read, after the compare (4 VALU + 2 LDS
// per chain-level, two dependent LDS round trips), 0 = both children beside the feature read (5 VALU + 2 LDS, one round trip).
// Measured (profiles/r04/tune_dep.txt): KR3's u8 walk 2.799 -> 2.731 ms, K3's u16 walk 3.469 -> 3.418 ms: on for both (round 2
// measured "no faster" on the 128-row tile; with three / six chains per lane the second round trip hides behind the other chains)
#ifndef TAHOE_Q8_DEP
#define TAHOE_Q8_DEP 1
#endif
#ifndef TAHOE_R3_DEP
#define TAHOE_R3_DEP 1
#endif
constexpr bool kReg8Dep = TAHOE_Q8_DEP != 0, kReg3Dep = TAHOE_R3_DEP != 0;
constexpr int kReg3Batch = kReg3Ring >= 2 * kQBatch ? kQBatch : kReg3Ring / 2;  // consumer batch of the 192-row tile (the kernel's default rule)
#ifndef TAHOE_Q8_COST
#define TAHOE_Q8_COST 218  // time of a 384-row u8 tile in percent of a 128-row u8 tile: KR3, 983,040 rows, 10 waves of 384-row tiles 2.665 ms
                           // against 30 waves of 128-row tiles 3.672 ms (profiles/r04/q8_cost.txt)
#endif
constexpr size_t kReg8Cost = TAHOE_Q8_COST;
// Region form: 192-row tiles (three chains, 14 walkers) take 1.33 x the time of 128-row tiles (two chains, 15 walkers),
// i.e. 0.89 per row -- but a last, partly filled wave of workgroups costs a whole tile time.  A batch is therefore
// walked as n whole waves of 192-row tiles followed by a remainder in whichever form is cheaper, n chosen to minimise
// 1.33 n + remainder (125 k rows: 2 waves of 192-row tiles + 209 tiles of 128 = 3.8 instead of 4 waves of 128-row tiles;
// 10 k rows: 79 tiles of 128).  Any cut is correct.  *rows3 = rows [0, rows3) in 192-row tiles (a multiple of 384),
// *chains = form of the remaining rows [rows3, rows).  `force` = 2 / 3: one form for the whole batch.  cost3 = time of a
// 192-row tile in percent of a 128-row tile (dense walk: 133; the sparse walk's three chains cost more: 161, sparse.hip).
// Tree slices of a remainder (dense region forms): when the 128-row tiles behind the whole waves are at most half the CUs, every
// tile is given to `s` workgroups, each a slice of the trees, and an ordered-sum kernel adds the leaf values (qring_launch); the
// remainder then costs ~100 / s + 20 (the second kernel and the leaf buffer; measured on K3, profiles/r04/rem_slices.txt: 35 at s = 7, 52 at s = 3) instead of 100.  `slice_trees` = trees of the largest
// group (0: no slices -- sparse handles); a slice must keep >= 60 trees (4 per walker).
inline int qreg_rem_slices(size_t rem_rows, int num_cus, int slice_trees)
{
    if (rem_rows == 0 || slice_trees < 120) return 1;
    const size_t tiles = (rem_rows + 127) / 128;
    const int fit = (int)std::min<size_t>((size_t)std::max(num_cus, 1) / tiles, 8);
    return std::max(1, std::min(fit, slice_trees / 60));
}
inline void qreg_plan(size_t rows, int num_cus, int force, size_t *rows3, int *chains, size_t cost3 = 133, size_t big = 192, int slice_trees = 0)
{
    // (`big` = rows of the large tile: 192 for u16 codes; the u8 form and narrow forests plan 384-row tiles against 128-row ones
    // with their own cost ratio and read *chains == 3 as "the large tile")
    *rows3 = 0;
    *chains = 2;
    const size_t cus = (size_t)std::max(num_cus, 1);
    auto waves = [cus](size_t r, size_t tile) { return ((r + tile - 1) / tile + cus - 1) / cus; };
    if (force == 2 || force == 3) {
        *chains = force;
        return;
    }
    size_t best = SIZE_MAX;
    for (size_t n = 0; n <= waves(rows, big); ++n) {
        const size_t r3 = std::min(rows, n * cus * big / 384 * 384), rem = rows - r3;
        size_t c2 = 100 * waves(rem, 128);
        const size_t c3 = cost3 * waves(rem, big);
        if (r3 > 0) {  // a small remainder in tree slices
            const int s = qreg_rem_slices(rem, num_cus, slice_trees);
            if (s > 1) c2 = std::min(c2, (size_t)(100 / s + 20));
        }
        const size_t cost = cost3 * waves(r3, big) + std::min(c2, c3);
        if (cost < best) {
            best = cost;
            *rows3 = rem ? r3 : 0;               // a pure large-tile plan is "no first part, remainder in form 3"
            *chains = rem ? (c2 <= c3 ? 2 : 3) : 3;
        }
    }
}
#ifndef TAHOE_QUANT_MAX_SHIFT
#define TAHOE_QUANT_MAX_SHIFT 16  // K3: 15 -> 0.907 ms, 16 -> 0.874 ms (half the table staging per row)
#endif
constexpr int kQuantMaxShift = TAHOE_QUANT_MAX_SHIFT;  // a quantise workgroup converts 2^cshift rows of its features; at most 65536
constexpr int kQuantMinRowsPerBlock = 512;  // ... and the fewest
constexpr int kQuantThreads = 512;
constexpr uint32_t kCodeMissing = 0xFFFFu;

// position of tile row r (0..127) inside a feature column of 128 u16: within a 32-lane group the
// rows land in 32 different LDS banks (two rows per dword come from different groups)
__host__ __device__ __forceinline__ int qrow_pos(int r) { return ((r & 31) << 1) | ((r >> 5) & 1) | ((r >> 6) << 6); }
// index of (row r, feature f) in the quantised workspace: tiles of 2^trs rows, xq[tile][f][2^trs]; 128-row tiles
// permute the rows inside a column (qrow_pos), the smaller tiles of the wide-row form keep them in order
// slot of row r (0..63) inside a 128-byte column of a 64-row REGION: lanes 0..31 of a wave land in 32 different banks,
// lanes 32..63 in the upper halves of the same dwords
__host__ __device__ __forceinline__ int qreg_pos(int r) { return ((r & 31) << 1) | (r >> 5); }
// `perm` (with trs = 6): the region layout of the NARROW walk -- xq[region = r / 64][f][64] with qreg_pos slots; a walk tile
// is K consecutive regions (K = 2 or 3, chosen per batch), so the quantise pass does not depend on K.
__device__ __forceinline__ size_t q_tile_index(size_t r, int f, int cols, int trs, int perm = 0)
{
    const int rr = (int)(r & (((size_t)1 << trs) - 1));
    return (((r >> trs) * (size_t)cols + (size_t)f) << trs) + (size_t)(trs == 7 ? qrow_pos(rr) : perm ? qreg_pos(rr) : rr);
}

// ---- 8-bit codes (forests whose features each see <= 254 distinct thresholds within a tree group: histogram-trained models) ----
// A region of the same 32 KiB then holds 128 rows: column = 128 bytes, one byte per row; row r (0..127) sits at byte
// 4 * (r & 31) + 2 * (r >> 6) + ((r >> 5) & 1), i.e. dword d of a column holds rows d, d + 32 (first 64-row half) and d + 64, d + 96
// (second half): the 32 lanes of a half-wave that walk one half read 32 different dwords = 32 banks for any per-lane fid, as with
// u16 codes.  A walk tile is K / 2 consecutive regions, chain k of a lane = half k & 1 of region k >> 1 (row 64 k + lane of the
// tile): six chains = 384 rows per staged top where u16 codes hold 192.  The missing code is 0xFF.
constexpr int kReg8Rows = 128;
constexpr uint32_t kCodeMissing8 = 0xFFu;
constexpr int kQMaxTable8 = 254;  // codes 0 .. 254, 255 = missing
__host__ __device__ __forceinline__ int qreg8_pos(int r) { return ((r & 31) << 2) | ((r >> 6) << 1) | ((r >> 5) & 1); }
__device__ __forceinline__ size_t q_tile_index8(size_t r, int f, int cols)
{
    return (((r >> 7) * (size_t)cols + (size_t)f) << 7) + (size_t)qreg8_pos((int)(r & 127));
}
// Stores the code of (row r, feature f) in the layout `perm` selects: 0 = tiles of 2^trs rows in order, 1 = 64-row regions of
// u16 codes (qreg_pos), 2 = 128-row regions of u8 codes (the workspace is then a byte array; `missing` codes become 0xFF).
__device__ __forceinline__ void q_store_code(uint16_t *xq, size_t r, int f, int cols, int trs, int perm, uint32_t code)
{
    if (perm == 2)
        reinterpret_cast<uint8_t *>(xq)[q_tile_index8(r, f, cols)] = (uint8_t)(code == kCodeMissing ? kCodeMissing8 : code);
    else
        xq[q_tile_index(r, f, cols, trs, perm)] = (uint16_t)code;
}

constexpr int kQuantPairThreads = 1024;

template <typename T>
inline hipError_t q_upload(T **dst, const T *src, size_t count, size_t *total)
{
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes);
    if (e != hipSuccess) return e;
    *total += bytes;
    if (count) e = hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// ---- device helpers of the walks on codes (qring.hip, sparse.hip) ----
// The branch rule on codes: right <=> (missing ? !def_left : code(x) >= code(thr)).  Written on wave
// masks: three v_cmp into SGPR pairs, three SALU ops, and the result is used directly as the lane
// predicate of v_cndmask / v_addc (hipcc's ?: form materialises both booleans in VGPRs: 6 more VALU).
// MS = false is the fast path for row chunks in which the quantise pass met no missing value (it
// reports that per chunk): the rule is then the single compare.
// NARROW (num_cols <= 256): node = code << 16 | fid << 8 | exchange << 7 | def_left, so that the feature column's LDS offset
// (fid * 256) is a bit field of the node word and one v_bfi forms the read address (q_xread).
// EX (NARROW only; probability-guided re-layout): bit 7 of the node word marks a node whose children are stored swapped;
// the condition is inverted there.  One more v_cmp (the sign of byte 0) and one s_xor per step.
// DLB = bit of the node word that holds def_left in the NARROW layouts: 0 (128-slot columns, u8 words) or 15 (region form on u16
// codes, dense and sparse: "def_left clear" is then one signed compare of the sign-extended low half -- v_cmp_ge_i32_sdwa
// sext(WORD_0) -- instead of v_and + v_cmp; the walk with the missing rule takes three compares per level, not four operations).
template <bool MS, bool NARROW, bool EX = false, uint32_t MISSC = kCodeMissing, int DLB = 0>
__device__ __forceinline__ uint64_t q_right_mask(uint32_t xc, uint32_t node)
{
    static_assert(DLB == 0 || (DLB == 15 && NARROW && !EX && MISSC == kCodeMissing), "bit 15: the u16 region word only");
    static_assert(!EX || NARROW, "the exchange bit lives in the NARROW node word");
    if (MISSC == kCodeMissing8) {
        // u8 codes (handles built for them: q->code8): node = M << 24 | code8 << 16 | fid << 7 | def_left with M = def_left ? 0xFF : 0.
        // A missing x has code 0xFF, >= every threshold code (a NaN threshold's 0xFF too), so `ge` alone sends it right; it must go
        // LEFT exactly where def_left is set, i.e. where xc == M: right = ge & ~(xc == M) -- two SDWA byte compares and one s_andn2
        // (the u16 rule takes three compares and three scalar operations).  xc == M can also hold for xc == 0 at a node without
        // def_left: ge is false there anyway (codes start at 1; a padding node's code 0 has identical subtrees).
        const uint64_t ge = __builtin_amdgcn_uicmp(xc, (node >> 16) & 0xFFu, 35 /* ICMP_UGE */);
        if (!MS) return ge;
        return ge & ~__builtin_amdgcn_uicmp(xc, node >> 24, 32 /* ICMP_EQ */);
    }
    const uint64_t ge = __builtin_amdgcn_uicmp(xc, NARROW ? node >> 16 : node & 0xFFFFu, 35 /* ICMP_UGE */);
    uint64_t right = ge;
    if (MS) {
        const uint64_t ms = __builtin_amdgcn_uicmp(xc, MISSC, 32 /* ICMP_EQ: the row's code says "missing" (0xFFFF) */);
        const uint64_t ndl = DLB == 15 ? __builtin_amdgcn_sicmp((int)(int16_t)(node & 0xFFFFu), 0, 39 /* ICMP_SGE: bit 15 (def_left) clear */)
                             : NARROW  ? __builtin_amdgcn_uicmp(node & 0x1u, 0u, 32 /* ICMP_EQ: def_left clear */)
                                       : __builtin_amdgcn_sicmp((int)node, -1, 38 /* ICMP_SGT: bit 31 (def_left) clear */);
        right = ge & ~(ms & ~ndl);  // = (ge & ~ms) | (ms & ndl): a missing code (the largest) is >= every threshold code, so ms implies ge
    }
    if (EX) right ^= __builtin_amdgcn_sicmp((int)(int8_t)(node & 0xFFu), 0, 40 /* ICMP_SLT: bit 7 (exchange) set */);
    return right;
}
template <bool MS, bool NARROW, bool EX = false, uint32_t MISSC = kCodeMissing, int DLB = 0>
__device__ __forceinline__ bool q_go_right(uint32_t xc, uint32_t node)
{
    return __builtin_amdgcn_inverse_ballot_w64(q_right_mask<MS, NARROW, EX, MISSC, DLB>(xc, node));
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
typedef const uint8_t __attribute__((address_space(3))) *lds_u8_ptr;
// FB = bits of the fid field the address takes from the node word: 8, or 7 when the regions stand at a 16-KiB stride (num_cols <= 128:
// bit 14 then belongs to the region base in posb, not to the fid).
template <bool LDSX, bool NARROW, int CSHIFT = 8, bool U8 = false, int FB = 8>
__device__ __forceinline__ uint32_t q_xread(const unsigned char *gx, uint32_t node, uint32_t posb)
{
    static_assert(!NARROW || CSHIFT == 8 || CSHIFT == 7, "the NARROW layouts have 256- or 128-byte feature columns");
    static_assert(!U8 || (LDSX && NARROW && CSHIFT == 7), "u8 codes are a form of the region layout");
    if (LDSX && NARROW) {
        // The tile starts at LDS address 0 (checked at kernel entry).  CSHIFT = 8: posb < 256, address = node[15:8] : posb[7:0].
        // CSHIFT = 7 (64-row regions at multiples of 32 KiB): posb = region base + slot (< 128), address takes node[14:7].
        uint32_t addr;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr) : "s"(((1u << FB) - 1u) << (CSHIFT == 7 ? 7 : 8)), "v"(node), "v"(posb));
        if (U8) return *reinterpret_cast<lds_u8_ptr>(addr);  // 128-row regions of u8 codes: ds_read_u8
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

// ---- quantize.hip ----
// Builds the group's threshold tables in every form the quantise kernels use from `tab` (per feature: sorted distinct
// non-NaN thresholds, at most kQMaxTable), picks the kernel forms, uploads.  Sets g.max_table.
tahoe_status quantize_build_tables(tahoe_forest *f, const std::vector<std::vector<float>> &tab, tahoe_qgroup &g);
void quantize_free_tables(tahoe_qgroup &g);
hipError_t quantize_allow_lds(const tahoe_forest *f);  // kernels that need more than 64 KiB of dynamic LDS
// Launches the quantise pass of group g for `rows` rows of `data` into f->q->xq (tiles of 2^trs rows) and the
// per-chunk "missing seen" flags; *cshift_out = log2 of the rows per flag.
tahoe_status quantize_launch(tahoe_forest *f, const tahoe_qgroup &g, const float *data, size_t rows, int trs, int perm,
                             hipStream_t stream, int *cshift_out);

}  // namespace tahoe

#endif  // TAHOE_AMD_QRING_INTERNAL_H
