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
// Rows too wide for a 128-row tile (num_cols > ~550) use 64 / 32 / 16-row tiles with several trees per wave
// (qwide_kernel); only beyond ~4500 columns are feature codes read from the quantised tile in L2 (GX form).
//
// Per predict: (1) quantize.hip turns the row-major float32 batch into u16 codes in the walk's LDS order (rows
// permuted inside a column so that lane = row reads are bank-conflict free); (2) qring_kernel walks: per CU one
// workgroup = NWALK walker waves (private top slot each, K 64-row chains per lane, next top
// prefetched in registers, no barrier in the tree loop) + one consumer wave that adds the leaf values
// in tree order through an LDS ring -- the float32 sums stay bit-identical to predict_on_cpu.
// Forms of the walk, picked at create (node layout) and per batch (qring_launch):
//   region form (num_cols <= 256; what K3 runs)  codes as 64-row regions xq[row / 64][fid][64]; a tile is three regions
//       (192 rows, 14 walkers, ring of 10) or two (128 rows, 15 walkers, ring of 24); a batch runs as whole waves of 192-row
//       tiles plus a remainder in the cheaper form; small batches give every tile to several workgroups, each a slice of
//       the trees (SPLIT) followed by ordered_sum_kernel
//   128-slot columns xq[tile][fid][128]          the general node layout (num_cols <= ~550), the exchange-bit layout
//       of the probability-guided re-layout, and the GX form (features read from L2)
//   wide rows (qwide_kernel)                      64 / 32 / 16-row tiles, several trees per wave
//
// Replaces, like forest.hip: the adaptive-format walkers/kernels of Struct.h:953-1704 and the
// layout build of Struct.h:1756-1986 (whose char/short/int "adaptive" widths compress only the
// fid/flag word; here the threshold is compressed too, losslessly).
#include "qring_internal.h"

// s_sleep arguments of the two spin loops (consumer polling ready flags, walker waiting for ring space).  With the 192-row
// tile's ring of 10 for 14 walkers a finished walker does wait for ring space: walker sleep 4 / 8 / 16 / 32 / 64 -> 3.66 / 3.62 /
// 3.60 / 3.70 / 5.10 ms on K3; consumer sleep 1 / 4 / 8 -> 3.645 / 3.657 / 3.669.  Raising the consumer wave's priority
// (s_setprio): no difference.
#ifndef TAHOE_CONS_SLEEP
#define TAHOE_CONS_SLEEP 1
#endif
#ifndef TAHOE_WALK_SLEEP
#define TAHOE_WALK_SLEEP 16
#endif
// ... of the 384-row u8 tile (ring of 5 for 14 walkers of six chains: a walker waits for ring space more often, a tree takes twice as
// long); KR3 walk (profiles/r04/tune_sleep8.txt): walker 8 / 16 / 32 -> 2.76 / 2.80 / 2.93 ms, consumer 1 / 2 -> 2.80 / 2.79
#ifndef TAHOE_WALK_SLEEP8
#define TAHOE_WALK_SLEEP8 8
#endif
#ifndef TAHOE_CONS_SLEEP8
#define TAHOE_CONS_SLEEP8 1
#endif

namespace tahoe {

// ------------------------------------------------------------------------------------------------
// (2) the walk.  Dynamic LDS: tile [cols][128] u16 | NWALK slots of (4 << L) bytes | ring [RING][64 K] f32 |
// ready[RING] | consumed.
// REG (the form K3 runs; NARROW node words with fid << 7, num_cols <= 256): the tile is K regions of 64 rows, region k =
// [fid][64] u16 at LDS address k * 32 KiB (so that the column offset fid * 128 and the region base never share a bit and
// one v_bfi forms the read address), chain k of a lane walks row 64 k + lane.  K = 3: a top staged once serves 192 rows
// (profiles/r02/tiles_experiment.json: almost half of a tile's time does not depend on its rows) with 14 walkers and a
// ring of 10; K = 2 (15 walkers, ring of 24) for batches that fill the chip better with 128-row tiles.
// SPLIT (small batches; the counterpart of the reference's split-forest strategy idx 4, Struct.h:1414-1606 + :2103-2164): when a
// batch has fewer tiles than the chip has CUs, every tile is given to `slices` workgroups that each walk a slice of the trees
// and write the leaf values to leafbuf[tree][row]; ordered_sum_kernel then adds them per row in tree order -- the same
// sequential float32 sum, where the reference's cub::DeviceSegmentedReduce adds per-block partial sums in another order.
// CODE8 (region form on u8 codes, qring_internal.h): the tile is K / 2 regions of 128 rows, chain k = half k & 1 of region k >> 1;
// the loop is the same instruction for instruction except that the feature code is a ds_read_u8.
// REGB = LDS stride of a region: 32 KiB (num_cols <= 256), or 16 KiB for forests of <= 128 features (a region is cols x 128 bytes;
// fid < 128 leaves bit 14 of the v_bfi field clear) -- six 64-row regions of u16 codes then fit where three did (384-row tiles for
// narrow forests with any number of thresholds), and six chains of u8 codes leave room for 15 walkers and a ring of 24.
template <int NWALK, bool WRITE_LEAF, bool LDSX, bool NARROW = false, bool EXCH = false, int K = 2, bool REG = false, int RING = kQRing,
          bool SPLIT = false, int BATCH = (RING >= 2 * kQBatch ? kQBatch : RING / 2), bool CODE8 = false, bool DEP = false,
          int REGB = kRegBytes>
__global__ void __launch_bounds__((NWALK + 1) * 64)
    qring_kernel(const uint16_t *__restrict__ xq, const uint32_t *__restrict__ top, const uint4 *__restrict__ blocks,
                 const uint32_t *__restrict__ qinner, const uint32_t *__restrict__ leaf_orig, float *sums,
                 uint32_t *__restrict__ leaf_out, size_t rows, int cols, int num_trees, int depth, int top_levels,
                 int top_stride, const uint32_t *__restrict__ chunk_flags, int *__restrict__ error_flag,
                 const float *sums_in, int tree_base, int total_trees, int cshift, float *__restrict__ leafbuf, size_t leaf_stride,
                 int slices, size_t row_begin)
{
    // `rows` is the END of the rows this launch walks and row_begin (region form only; a multiple of 384) their start: a batch
    // may be walked as whole waves of 192-row tiles followed by a remainder of 128-row tiles (qring_launch).
    static_assert(!SPLIT || REG, "tree slices are a form of the region kernel");
    static_assert(REG || K == kQRows / 64, "the 128-slot column layout holds exactly two chains");
    static_assert(!REG || (LDSX && NARROW && !EXCH), "regions are a form of the NARROW LDS tile");
    static_assert(BATCH >= 1 && RING >= 2 * BATCH, "ring too small");
    static_assert(!CODE8 || (REG && K % 2 == 0), "u8 codes: whole 128-row regions");
    static_assert(REGB == kRegBytes || (REG && REGB == kRegBytes / 2), "region stride: 32 KiB, or 16 KiB for <= 128 features");
    constexpr uint32_t MISSC = CODE8 ? kCodeMissing8 : kCodeMissing;
    constexpr int NREG = CODE8 ? K / 2 : K;  // regions of the tile
    constexpr int FB = REGB == kRegBytes ? 8 : 7;  // fid bits of the LDS address (q_xread)
    constexpr int DLB = (REG && !CODE8) ? 15 : 0;  // def_left bit of the node word (q_right_mask)
    constexpr int TR = 64 * K;             // rows per tile
    constexpr int CS = REG ? 7 : 8;        // log2 of a feature column's bytes
    constexpr int NT = (NWALK + 1) * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int slot_bytes = kQSlotBytes;  // fixed 4 KiB slot (2^10 u32): no size-dependent branches in the loop

    uint16_t *tile = reinterpret_cast<uint16_t *>(smem);  // LDSX only
    const unsigned tile_id = SPLIT ? blockIdx.x / (unsigned)slices : blockIdx.x;
    const int slice = SPLIT ? (int)(blockIdx.x % (unsigned)slices) : 0;
    const int t_begin = SPLIT ? (int)((long long)num_trees * slice / slices) : 0;        // this workgroup's trees
    const int t_end = SPLIT ? (int)((long long)num_trees * (slice + 1) / slices) : num_trees;
    unsigned char *slots = smem + (REG ? (size_t)NREG * REGB : LDSX ? (size_t)cols * TR * sizeof(uint16_t) : 0);
    const unsigned char *gx = reinterpret_cast<const unsigned char *>(xq + (size_t)tile_id * ((size_t)cols * TR));
    float *ring_vals = reinterpret_cast<float *>(slots + (size_t)NWALK * slot_bytes);
    uint32_t *ring_ready = reinterpret_cast<uint32_t *>(ring_vals + RING * TR);
    uint32_t *consumed = ring_ready + RING;

    const size_t row0 = row_begin + (size_t)tile_id * TR;
    if (LDSX && NARROW && (uint32_t)reinterpret_cast<uintptr_t>(tile) != 0u) {
        // q_xread's v_bfi needs the tile at LDS address 0 (true while the kernel has no static LDS)
        if (tid == 0) atomicOr(error_flag, 2);
        return;
    }

    // ---- stage the quantised tile (already in LDS order): straight 16-byte copies ----
    if (REG) {  // consecutive regions of the workspace (cols x 128 bytes each: 64 rows of u16 or 128 rows of u8 codes), each to
                // its 32-KiB-aligned place
        const int n16 = cols * kRegRows * 2 / 16;  // 16-byte pieces of a region
        const size_t first_region = CODE8 ? row0 >> 7 : row0 >> 6;
        for (int k = 0; k < NREG; ++k) {
            const uint4 *src = reinterpret_cast<const uint4 *>(xq + (first_region + (size_t)k) * ((size_t)cols * kRegRows));
            uint4 *dst = reinterpret_cast<uint4 *>(smem + (size_t)k * REGB);
            for (int e = tid; e < n16; e += NT) dst[e] = src[e];
        }
    } else if (LDSX) {
        const uint4 *src = reinterpret_cast<const uint4 *>(xq + (size_t)tile_id * ((size_t)cols * TR));
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
        const int n16 = cols * TR * 2 / 16;
        for (int e = tid; e < n16; e += NT) dst[e] = src[e];
    }
    if (tid < RING) ring_ready[tid] = 0u;
    if (tid == RING) *consumed = 0u;

    if (wave == NWALK) {
        // ================= consumer: ordered accumulation =================
        __syncthreads();
        if (SPLIT) return;  // the leaf values go to leafbuf; ordered_sum_kernel adds them
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
                if (++spins > kQSpinLimit) {
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(CODE8 && K > 2 ? TAHOE_CONS_SLEEP8 : TAHOE_CONS_SLEEP);
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
    uint32_t *slot = reinterpret_cast<uint32_t *>(slots + (size_t)wave * slot_bytes);
    const int n_chunks = (top_stride * 4) >> 4;  // 16-byte chunks of a top in global memory (<= 256)
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};   // named registers (an indexed array would go to scratch)
    // byte offsets of this lane's four 16-byte chunks inside a top (clamped: branch-free and in bounds), computed once:
    // the loads then take the tree's base from SGPRs and a 32-bit offset from a VGPR, no per-tree address arithmetic
    const uint32_t last_chunk = (uint32_t)(n_chunks - 1);
    const uint32_t po0 = 16u * min((uint32_t)(0 * 64 + lane), last_chunk), po1 = 16u * min((uint32_t)(1 * 64 + lane), last_chunk);
    const uint32_t po2 = 16u * min((uint32_t)(2 * 64 + lane), last_chunk), po3 = 16u * min((uint32_t)(3 * 64 + lane), last_chunk);
    auto prefetch_top = [&](int t) {
        const unsigned char *g = reinterpret_cast<const unsigned char *>(top + (size_t)t * top_stride);
        pf0 = *reinterpret_cast<const uint4 *>(g + po0);
        pf1 = *reinterpret_cast<const uint4 *>(g + po1);
        pf2 = *reinterpret_cast<const uint4 *>(g + po2);
        pf3 = *reinterpret_cast<const uint4 *>(g + po3);
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
    if (t_begin + wave < t_end) {
        prefetch_top(t_begin + wave);
        commit_top();
    }
    __syncthreads();  // the tile, the ring state and (own wave) the first top are in LDS

    bool dead = false;
    auto run = [&](auto ms_tag) {
        constexpr bool MS = decltype(ms_tag)::value;
        uint32_t pos[K];  // byte position of this lane's row inside a 256-byte feature column, per chain
    #pragma unroll
        for (int k = 0; k < K; ++k)  // low 32 bits of a generic LDS pointer = the LDS byte address
            pos[k] = CODE8 ? (uint32_t)((k >> 1) * REGB) + (uint32_t)qreg8_pos((k & 1) * 64 + lane)
                     : REG ? (uint32_t)(k * REGB) + 2u * (uint32_t)qreg_pos(lane)
                           : (LDSX ? (uint32_t)reinterpret_cast<uintptr_t>(tile) : 0u) + 2u * (uint32_t)qrow_pos(k * 64 + lane);
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
                const bool c0 = q_go_right<MS, NARROW, EXCH, MISSC, DLB>(q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, na[k].x, pos[k]), na[k].x);
                const uint32_t n1 = c0 ? na[k].z : na[k].y;
                const bool c1 = q_go_right<MS, NARROW, EXCH, MISSC, DLB>(q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, n1, pos[k]), n1);
                const uint32_t lo = c0 ? nb[k].z : nb[k].x, hi = c0 ? nb[k].w : nb[k].y;
                v[k] = __uint_as_float(c1 ? hi : lo);
                if (WRITE_LEAF) {
                    const size_t row = row0 + k * 64 + lane;
                    if (row < rows)
                        leaf_out[row * (size_t)total_trees + tree_base + t] =
                            leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bs[k] * 4 + 2 * (c0 ? 1 : 0) + (c1 ? 1 : 0)];
                }
            }
            if (SPLIT) {  // straight to leafbuf[tree][row]: 64 consecutive floats per chain
    #pragma unroll
                for (int k = 0; k < K; ++k) {
                    const size_t row = row0 + k * 64 + lane;
                    if (row < rows) leafbuf[(size_t)t * leaf_stride + (row - row_begin)] = v[k];  // (rows of this launch: [row_begin, rows))
                }
                return;
            }
            if (t >= RING) {  // ring entry still in use by tree t - RING?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t - RING + 1)) {
                    if (++spins > kQSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(CODE8 && K > 2 ? TAHOE_WALK_SLEEP8 : TAHOE_WALK_SLEEP);
                }
            }
            const int e = t % RING;
    #pragma unroll
            for (int k = 0; k < K; ++k) ring_vals[e * TR + k * 64 + lane] = v[k];
            TAHOE_LDS_RELEASE();  // values before the flag: a wave's LDS operations are performed in issue order
            if (lane == 0) lds_flag_store(&ring_ready[e], (uint32_t)(t + 1));
        };
        int t_p = -1;  // tree whose bottom blocks are in flight
        uint4 na_p[K] = {}, nb_p[K] = {};
        uint32_t bsel_p[K] = {};
        for (int t = t_begin + wave; t < t_end && !dead; t += NWALK) {
            const bool more = t + NWALK < t_end;
            if (more) prefetch_top(t + NWALK);
            uint32_t i[K];
    #pragma unroll
            for (int k = 0; k < K; ++k) i[k] = 1;
            if (top_levels > 0) {
                uint32_t node[K];
    #pragma unroll
                for (int k = 0; k < K; ++k) node[k] = slot[1];
                if (DEP) {
                    // Only the chosen child, read AFTER the compare (ds_read_b32 at slot + 4 i): 4 VALU + 2 LDS per chain and
                    // level instead of 5 + 2 -- no v_cndmask, half the LDS bytes -- at the price of two dependent LDS round trips
                    // per level.  With six chains per lane and 14 walkers (84 chain-waves per CU) a chain's turn comes round
                    // every ~500 cycles of an issue-bound CU, so the second round trip hides; with three chains it did not pay
                    // ("reading only the chosen child ... times the same").
                    const uint32_t slot_a = (uint32_t)reinterpret_cast<uintptr_t>(slot);
                    for (int l = 0; l < top_levels - 1; ++l) {
                        uint32_t xc[K];
    #pragma unroll
                        for (int k = 0; k < K; ++k) xc[k] = q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, node[k], pos[k]);
    #pragma unroll
                        for (int k = 0; k < K; ++k) {
                            i[k] = q_descend(i[k], q_right_mask<MS, NARROW, EXCH, MISSC, DLB>(xc[k], node[k]));
                            node[k] = *reinterpret_cast<const uint32_t __attribute__((address_space(3))) *>(slot_a + 4u * i[k]);
                        }
                    }
                } else {
                // Both children come with one ds_read_b64 issued beside the feature read, ahead of the compare.  (Reading
                // only the chosen child afterwards -- half the LDS bytes, twice the round trips -- times the same to
                // 0.5 %: neither LDS bandwidth nor LDS latency bounds this loop, see DESIGN.md.)
                int l0 = 0;
                if (top_levels > 1) {  // level 0: every chain is at the root -- one read of its child pair serves them all
                    const uint2 pr0 = *reinterpret_cast<const uint2 *>(&slot[2]);
                    uint32_t xc0[K];
    #pragma unroll
                    for (int k = 0; k < K; ++k) xc0[k] = q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, node[k], pos[k]);
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint64_t cm = q_right_mask<MS, NARROW, EXCH, MISSC, DLB>(xc0[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr0.y : pr0.x;
                    }
                    l0 = 1;
                }
                for (int l = l0; l < top_levels - 1; ++l) {
                    uint32_t xc[K];
                    uint2 pr[K];
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        xc[k] = q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, node[k], pos[k]);
                        pr[k] = *reinterpret_cast<const uint2 *>(&slot[2 * i[k]]);  // children 2i, 2i+1
                    }
    #pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint64_t cm = q_right_mask<MS, NARROW, EXCH, MISSC, DLB>(xc[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr[k].y : pr[k].x;
                    }
                }
                }
    #pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t xc = q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, node[k], pos[k]);
                    i[k] = q_descend(i[k], q_right_mask<MS, NARROW, EXCH, MISSC, DLB>(xc, node[k]));
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
                        const uint32_t xc = q_xread<LDSX, NARROW, CS, CODE8, FB>(gx, n, pos[k]);
                        idx = 2u * idx + 1u + (q_go_right<MS, NARROW, EXCH, MISSC, DLB>(xc, n) ? 1u : 0u);
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
                // the tree's blocks from SGPRs, the block's byte offset (32 bits) from a VGPR
                const unsigned char *tb = reinterpret_cast<const unsigned char *>(blocks + (size_t)t * n_blocks * 2);
                const uint4 *bp = reinterpret_cast<const uint4 *>(tb + 32u * bsel[k]);
                na_p[k] = bp[0];  // node0, node1, node2, 0
                nb_p[k] = bp[1];  // four leaf values
                bsel_p[k] = bsel[k];
            }
            if (more) commit_top();  // this wave's reads of its slot are done (in-order LDS)
        }
        if (t_p >= 0 && !dead) finish(t_p, na_p, nb_p, bsel_p);
    };
    // chunk_flags[c] != 0 <=> the quantise pass met a missing value in rows [c, c+1) << cshift; a 192-row tile can
    // straddle two chunks (a chunk is at least 512 rows)
    if ((chunk_flags[row0 >> cshift] | chunk_flags[(min(rows, row0 + TR) - 1) >> cshift]) != 0)
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
// KG = 3 (small tops, e.g. K2's depth-8 trees: 256 B each): a walker wave holds three such groups of trees in its slot and
// every lane walks three trees at once, chain k = tree j + k * TPW of the 3 * TPW consecutive trees of the iteration -- three
// independent dependent-read chains per wave instead of one, with a ring of 16 KiB for the larger number of trees in flight.
constexpr int kWideRingBytes = 8192;
constexpr int kWideRingBytesK = 16384;  // ring of the KG = 3 form
template <int RT, bool WRITE_LEAF, int KG = 1, int RB = kWideRingBytes>
__global__ void __launch_bounds__(16 * 64)
    qwide_kernel(const uint16_t *__restrict__ xq, const uint32_t *__restrict__ top, const uint4 *__restrict__ blocks,
                 const uint32_t *__restrict__ qinner, const uint32_t *__restrict__ leaf_orig, float *sums,
                 uint32_t *__restrict__ leaf_out, size_t rows, int cols, int num_trees, int depth, int top_levels,
                 int top_stride, const uint32_t *__restrict__ chunk_flags, int *__restrict__ error_flag,
                 const float *sums_in, int tree_base, int total_trees, int slot_bytes, int lw, int cshift)
{
    constexpr int NWALK = 15;
    constexpr int NT = (NWALK + 1) * 64;
    constexpr int TPW = 64 / RT;                       // trees a walker wave walks at once, per chain
    constexpr int TPG = TPW * KG;                      // ... in all: the consecutive trees of one walker iteration
    constexpr int RE = RB / (RT * 4);                  // ring entries (trees)
#ifndef TAHOE_WIDE_BATCH
#define TAHOE_WIDE_BATCH 64  // K2 (four trees per wave): 64 -> 1.07 ms, 16 -> 1.16, 8 -> 1.44
#endif
    constexpr int NBATCH = RE / 2 < TAHOE_WIDE_BATCH ? RE / 2 : TAHOE_WIDE_BATCH;  // trees the consumer takes per poll (<= 64)
    constexpr int CSHIFT = RT == 64 ? 7 : RT == 32 ? 6 : 5;  // log2 of a feature column's bytes
    static_assert(NBATCH <= 64 && RE >= (KG == 1 ? 2 : 1) * NWALK * TPG, "ring too small");
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
    const int j = lane / RT, r = lane % RT;           // tree slot, row of the tile
    // lw (<= min(top_levels, LWMAX), chosen at create so that tile + slots fit): levels served from the LDS slot
    const int cpt = max(1, (1 << lw) >> 2);           // 16-byte chunks per staged top (top_stride >= 4 entries)
    unsigned char *wslot = slots + (size_t)wave * slot_bytes;  // TPG tops: tree jj of the iteration at jj * cpt * 16
    const int n_groups = (num_trees + TPG - 1) / TPG;
    uint4 pf0 = {}, pf1 = {}, pf2 = {}, pf3 = {};
    // chunk c of the wave's slot = chunk (c % cpt) of tree g * TPG + c / cpt (clamped: in bounds, branch-free); TPG * cpt <= 256
    auto prefetch_tops = [&](int g) {
        auto ld = [&](int c) {
            c = min(c, TPG * cpt - 1);
            const int t = min(g * TPG + c / cpt, num_trees - 1);
            return reinterpret_cast<const uint4 *>(top + (size_t)t * top_stride)[c % cpt];
        };
        pf0 = ld(0 * 64 + lane);
        pf1 = ld(1 * 64 + lane);
        pf2 = ld(2 * 64 + lane);
        pf3 = ld(3 * 64 + lane);
    };
    auto commit_tops = [&]() {  // clamped lanes rewrite the last chunk with its own value
        uint4 *s = reinterpret_cast<uint4 *>(wslot);
        const int last = TPG * cpt - 1;
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
        const uint32_t *slot[KG];  // chain k walks tree k * TPW + j of the iteration
#pragma unroll
        for (int k = 0; k < KG; ++k) slot[k] = reinterpret_cast<const uint32_t *>(wslot + (size_t)(k * TPW + j) * cpt * 16);
        auto finish = [&](int g, const uint4 (&na)[KG], const uint4 (&nb)[KG], const uint32_t (&bs)[KG]) {
            float v[KG];
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int t = g * TPG + k * TPW + j;
                const bool c0 = q_go_right<MS, false>(q_xread<true, false, CSHIFT>(nullptr, na[k].x, pos), na[k].x);
                const uint32_t n1 = c0 ? na[k].z : na[k].y;
                const bool c1 = q_go_right<MS, false>(q_xread<true, false, CSHIFT>(nullptr, n1, pos), n1);
                const uint32_t lo = c0 ? nb[k].z : nb[k].x, hi = c0 ? nb[k].w : nb[k].y;
                v[k] = __uint_as_float(c1 ? hi : lo);
                if (WRITE_LEAF) {
                    if (t < num_trees && row < rows)
                        leaf_out[row * (size_t)total_trees + tree_base + t] =
                            leaf_orig[(size_t)t * ((size_t)n_blocks * 4) + (size_t)bs[k] * 4 + 2 * (c0 ? 1 : 0) + (c1 ? 1 : 0)];
                }
            }
            const int t_last = min(g * TPG + TPG - 1, num_trees - 1);
            if (t_last >= RE) {  // the iteration's ring entries still in use?
                int spins = 0;
                while (lds_flag_load(consumed) < (uint32_t)(t_last - RE + 1)) {
                    if (++spins > kQSpinLimit) {
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int t = g * TPG + k * TPW + j;
                if (t < num_trees) ring_vals[(t % RE) * RT + r] = v[k];
            }
            TAHOE_LDS_RELEASE();  // values before the flags: a wave's LDS operations are performed in issue order
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int t = g * TPG + k * TPW + j;
                if (r == 0 && t < num_trees) lds_flag_store(&ring_ready[t % RE], (uint32_t)(t + 1));
            }
        };
        int g_p = -1;  // iteration whose bottom blocks are in flight
        uint4 na_p[KG] = {}, nb_p[KG] = {};
        uint32_t bsel_p[KG] = {};
        for (int g = wave; g < n_groups && !dead; g += NWALK) {
            const bool more = g + NWALK < n_groups;
            if (more) prefetch_tops(g + NWALK);
            int t[KG];  // lanes of a missing tree repeat the last one, unused
            uint32_t i[KG];
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                t[k] = min(g * TPG + k * TPW + j, num_trees - 1);
                i[k] = 1;
            }
            if (lw > 0) {
                uint32_t node[KG];
#pragma unroll
                for (int k = 0; k < KG; ++k) node[k] = slot[k][1];
                for (int l = 0; l < lw - 1; ++l) {
                    uint32_t xc[KG];
                    uint2 pr[KG];
#pragma unroll
                    for (int k = 0; k < KG; ++k) {
                        xc[k] = q_xread<true, false, CSHIFT>(nullptr, node[k], pos);
                        pr[k] = *reinterpret_cast<const uint2 *>(&slot[k][2 * i[k]]);  // children 2i, 2i+1
                    }
#pragma unroll
                    for (int k = 0; k < KG; ++k) {
                        const uint64_t cm = q_right_mask<MS, false>(xc[k], node[k]);
                        i[k] = q_descend(i[k], cm);
                        node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr[k].y : pr[k].x;
                    }
                }
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                    const uint32_t xc = q_xread<true, false, CSHIFT>(nullptr, node[k], pos);
                    i[k] = q_descend(i[k], q_right_mask<MS, false>(xc, node[k]));
                }
            }
            uint32_t bsel[KG];
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                uint32_t idx = i[k] - 1;  // 0-based heap index on level lw
                if (lw < depth - 2) {  // the levels between the slot and the bottom blocks: quantised heap in global memory
                    const uint32_t *tree = qinner + (size_t)t[k] * n_inner;
                    for (int l = lw; l < depth - 2; ++l) {
                        const uint32_t n = tree[idx];
                        const uint32_t xc = q_xread<true, false, CSHIFT>(nullptr, n, pos);
                        idx = 2u * idx + 1u + (q_go_right<MS, false>(xc, n) ? 1u : 0u);
                    }
                }
                bsel[k] = idx - first_block_node;
            }
            if (g_p >= 0) finish(g_p, na_p, nb_p, bsel_p);
            g_p = g;
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const uint4 *bp = blocks + ((size_t)t[k] * n_blocks + bsel[k]) * 2;
                na_p[k] = bp[0];
                nb_p[k] = bp[1];
                bsel_p[k] = bsel[k];
            }
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
// Second step of the SPLIT form: per row, the leaf values of the group's trees in tree order, continuing sums_in.
// (Round 4: 64-thread workgroups -- a 10 k-row batch spreads over 157 CUs instead of 40 -- and 32 loads in flight per thread: the
// kernel is a chain of dependent global-memory round trips, K1: 34 -> ~10 us.)
constexpr int kOrderedSumThreads = 64;
__global__ void __launch_bounds__(kOrderedSumThreads) ordered_sum_kernel(const float *__restrict__ leafbuf, size_t leaf_stride, int num_trees,
                                                                         const float *sums_in, float *sums, size_t rows, size_t row_begin)
{
    // rows [row_begin, rows) of the batch; leafbuf[tree][row - row_begin]
    const size_t r = (size_t)blockIdx.x * kOrderedSumThreads + threadIdx.x;
    const size_t row = row_begin + r;
    if (row >= rows) return;
    float sum = sums_in ? sums_in[row] : 0.0f;
    int t = 0;
    for (; t + 32 <= num_trees; t += 32) {  // 32 loads in flight, 32 adds in tree order
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = leafbuf[(size_t)(t + j) * leaf_stride + r];
#pragma unroll
        for (int j = 0; j < 32; ++j) sum += v[j];
    }
    for (; t + 8 <= num_trees; t += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = leafbuf[(size_t)(t + j) * leaf_stride + r];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; t < num_trees; ++t) sum += leafbuf[(size_t)t * leaf_stride + r];
    sums[row] = sum;
}

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
static long long qwide_lds_for(const tahoe_forest *f, int rt, int lw, int kg = 1)
{
    const long long ring = kg > 1 ? kWideRingBytesK : kWideRingBytes;
    return (long long)f->p.num_cols * rt * 2 + 15LL * kg * qwide_slot_bytes(lw, rt) + ring + (ring / (rt * 4) + 1) * 4LL;
}
// tree groups per walker (chains per lane) of the wide form: 3 when three groups of tops fill at most 256 16-byte chunks and
// fit LDS beside the tile and the larger ring, else 1
static int qwide_kg(const tahoe_forest *f, int rt, int lw)
{
    if (const char *e = getenv("TAHOE_QRING_WIDE_CHAINS"))  // experiments: 1 keeps one group per walker
        if (atoi(e) == 1) return 1;
    const long long chunks = 3LL * qwide_slot_bytes(lw, rt) / 16;
    return (chunks <= 256 && qwide_lds_for(f, rt, lw, 3) <= f->lds_limit) ? 3 : 1;
}
// rows per tile of the wide-row form (qwide_kernel); 0 = not used (the 128-row tile fits, or not even 16 rows do)
int qwide_rows(const tahoe_forest *f) { return f->q ? f->q->wide_rt : 0; }
int qwide_chains(const tahoe_forest *f) { return f->q && f->q->wide_rt ? f->q->wide_kg : 0; }  // trees a lane walks at once; 0 = form not used
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
    if (const int want = f->knob_qring_walkers)  // TAHOE_QRING_WALKERS, read at create
        for (int n : options)
            if (n == want && qring_lds_for(f, n) <= f->lds_limit) return n;
    for (int n : options)
        if (qring_lds_for(f, n) <= f->lds_limit) return n;
    return kGxWalkers;  // rows too wide for an LDS tile: features are read from the quantised tile in L2
}

long long qring_lds_bytes(const tahoe_forest *f)
{
    if (f->q && f->q->reg) return qreg_lds_for(3, kReg3Walkers, kReg3Ring);
    const int n = qring_walkers(f);
    return n ? qring_lds_for(f, n, qring_lds_tile(f)) : 0;
}

template <int NWALK>
static hipError_t q_allow(long long lds)
{
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<NWALK, false, true>), (int)lds);
    if (e != hipSuccess) return e;
    e = allow_max_lds(reinterpret_cast<const void *>(&qring_kernel<NWALK, true, true>), (int)lds);
    if (e != hipSuccess || NWALK != 15) return e;
    for (const void *k : {(const void *)&qring_kernel<15, false, true, true>, (const void *)&qring_kernel<15, true, true, true>,
                          (const void *)&qring_kernel<15, false, true, true, true>, (const void *)&qring_kernel<15, true, true, true, true>,
                          (const void *)&qring_kernel<15, false, true, true, false, 2, true>, (const void *)&qring_kernel<15, true, true, true, false, 2, true>,
                          (const void *)&qring_kernel<kReg3Walkers, false, true, true, false, 3, true, kReg3Ring, false, kReg3Batch, false, kReg3Dep>,
                          (const void *)&qring_kernel<kReg3Walkers, true, true, true, false, 3, true, kReg3Ring, false, kReg3Batch, false, kReg3Dep>,
                          (const void *)&qring_kernel<15, false, true, true, false, 2, true, kQRing, true>,
                          (const void *)&qring_kernel<15, true, true, true, false, 2, true, kQRing, true>,
                          (const void *)&qring_kernel<kReg8Walkers, false, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, true, kReg8Dep>,
                          (const void *)&qring_kernel<kReg8Walkers, true, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, true, kReg8Dep>,
                          (const void *)&qring_kernel<15, false, true, true, false, 2, true, kQRing, false, kQBatch, true>,
                          (const void *)&qring_kernel<15, true, true, true, false, 2, true, kQRing, false, kQBatch, true>,
                          (const void *)&qring_kernel<15, false, true, true, false, 2, true, kQRing, true, kQBatch, true>,
                          (const void *)&qring_kernel<15, true, true, true, false, 2, true, kQRing, true, kQBatch, true>,
                          // <= 128 features: six 16-KiB regions of u16 codes; three 16-KiB regions of u8 codes, 15 walkers, ring of 24
                          (const void *)&qring_kernel<kReg8Walkers, false, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, false, kReg8Dep, kRegBytes / 2>,
                          (const void *)&qring_kernel<kReg8Walkers, true, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, false, kReg8Dep, kRegBytes / 2>,
                          (const void *)&qring_kernel<15, false, true, true, false, 6, true, kQRing, false, kQBatch, true, kReg8Dep, kRegBytes / 2>,
                          (const void *)&qring_kernel<15, true, true, true, false, 6, true, kQRing, false, kQBatch, true, kReg8Dep, kRegBytes / 2>})
        if ((e = allow_max_lds(k, (int)lds)) != hipSuccess) return e;
    return hipSuccess;
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
        tab[h_inner[i].meta & kMetaFidMask].push_back(h_inner[i].thr);
    }
    parallel_for((size_t)cols, 4, [&tab](size_t c_lo, size_t c_hi) {  // features are independent
        for (size_t c = c_lo; c < c_hi; ++c) {
            auto &v = tab[c];
            std::sort(v.begin(), v.end());                       // float order; -0.0f and 0.0f compare equal
            v.erase(std::unique(v.begin(), v.end()), v.end());   // ... and collapse into one entry
        }
    });
    int max_count = 0;
    for (int c = 0; c < cols; ++c) max_count = std::max(max_count, (int)tab[c].size());
    if (max_count > kQMaxTable) {
        *too_many = max_count;
        return TAHOE_OK;
    }
    g.max_count = max_count;
    // ---- node codes ----
    auto encode = [&](const InnerNode &n, bool real) -> uint32_t {
        if (!real) return 0u;  // padding below a leaf: both children carry the same value
        const uint32_t fid = n.meta & kMetaFidMask, dl = n.meta >> 31, ex = (n.meta & kMetaExchange) ? 1u : 0u;
        uint32_t code;
        if (std::isnan(n.thr))
            code = 0xFFFFu;  // x >= NaN is never true; a missing x still follows def_left
        else {
            const auto &v = tab[fid];
            code = (uint32_t)(std::lower_bound(v.begin(), v.end(), n.thr) - v.begin()) + 1u;
        }
        // u8 handles (every table <= 254 entries): M << 24 | code8 << 16 | fid << 7 | def_left, M = def_left ? 0xFF : 0 (q_right_mask)
        if (q->code8) return ((dl ? 0xFFu : 0u) << 24) | ((code == 0xFFFFu ? 0xFFu : code) << 16) | (fid << 7) | dl;
        // region form: the column offset fid * 128 is a bit field of the node; def_left in bit 15, the sign of the low half
        if (q->reg) return (code << 16) | (dl << 15) | (fid << 7);
        return q->narrow ? (code << 16) | (fid << 8) | (ex << 7) | dl : code | (fid << 16) | (dl << 31);
    };
    const size_t Tg = hi - lo;
    g.tree_lo = (int)lo;
    g.num_trees = (int)Tg;
    const size_t top_n = (size_t)1 << q->top_levels;  // entries per tree (entry 0 unused)
    const size_t top_stride = (size_t)q->top_stride;
    const size_t n_blocks = (size_t)1 << (f->depth - 2);
    const size_t first = n_blocks - 1;
    std::vector<uint32_t> h_top(Tg * top_stride, 0u);
    std::vector<uint4> h_blocks(Tg * n_blocks * 2);
    std::vector<uint32_t> h_qinner(q->have_mid ? Tg * n_inner : 0);
    parallel_for(Tg, 8, [&](size_t t_lo, size_t t_hi) {  // trees are independent; `encode` only reads
    for (size_t t = t_lo; t < t_hi; ++t) {
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
    });
    {
        const tahoe_status qs = quantize_build_tables(f, tab, g);
        if (qs != TAHOE_OK) return qs;
    }
    hipError_t e;
    auto bad = [&](const char *what) { return fail(TAHOE_ERR_HIP, "qring_build: %s failed: %s", what, hipGetErrorString(e)); };
    if ((e = q_upload(&g.top, h_top.data(), h_top.size(), &f->device_bytes)) != hipSuccess) return bad("top");
    if ((e = q_upload(&g.blocks, h_blocks.data(), h_blocks.size(), &f->device_bytes)) != hipSuccess) return bad("blocks");
    if (q->have_mid && (e = q_upload(&g.qinner, h_qinner.data(), h_qinner.size(), &f->device_bytes)) != hipSuccess)
        return bad("qinner");
    return TAHOE_OK;
}

static void free_group(tahoe_qgroup &g)
{
    quantize_free_tables(g);
    for (void *p : {(void *)g.top, (void *)g.blocks, (void *)g.qinner})
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
    q->wide_kg = q->wide_rt ? qwide_kg(f, q->wide_rt, q->wide_lw) : 1;
    if (q->wide_rt) q->have_mid = q->have_mid || f->depth - 2 > q->wide_lw;  // smaller tops in LDS
    q->top_stride = (int)std::max<size_t>((size_t)1 << q->top_levels, 4);  // >= 16 bytes per tree
    {
        const char *e = getenv("TAHOE_QRING_NARROW");  // experiments: 0 keeps the general node layout
        q->narrow = cols <= 256 && qring_walkers(f) == 15 && qring_lds_tile(f) && !(e && atoi(e) == 0);
    }
    {
        const char *e = getenv("TAHOE_QRING_REGIONS");  // experiments: 0 keeps the 128-slot column layout
        q->reg = q->narrow && !f->relayout && qreg_lds_for(3, kReg3Walkers, kReg3Ring) <= f->lds_limit && !(e && atoi(e) == 0);
        if (const char *k = getenv("TAHOE_QRING_CHAINS")) f->knob_qring_chains = atoi(k);  // 2 / 3: force the tile form
        if (const char *k = getenv("TAHOE_QRING_SLICES")) f->knob_qring_slices = atoi(k);  // >= 1: force the tree slices per tile
    }
    if (f->relayout && !q->narrow) {  // only the NARROW node word has room for the exchange bit: the strategy steps aside
        qring_destroy(f);
        return TAHOE_OK;
    }
    // ---- cut the forest into tree groups whose features each see <= kQMaxTable distinct thresholds ----
    // G groups of (nearly) equal size, G as small as the busiest feature allows.  First guess from the distinct thresholds
    // per feature of the whole forest (a group of T / G trees sees at most that many, usually ~1 / G of them), then G grows
    // until every group fits; trees that differ wildly in size make that loop longer, never wrong.
    size_t G = 1;
    {
        // distinct thresholds per feature over the whole forest (sort + unique per feature, features in parallel): a forest of
        // histogram-trained trees has hundreds of thousands of nodes on its busiest feature and a few hundred distinct
        // thresholds -- counting nodes (round 1-3) cut such forests into several groups for nothing, each with its own
        // quantise pass and walk launches (KR3: 4 groups, 5.9 ms instead of one group)
        std::vector<std::vector<float>> all((size_t)cols);
        for (size_t i = 0; i < T * f->n_inner; ++i)
            if (h_real[i] && !std::isnan(h_inner[i].thr)) all[h_inner[i].meta & kMetaFidMask].push_back(h_inner[i].thr);
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
        if (most > (size_t)kQMaxTable) G = (most + kQMaxTable - 1) / kQMaxTable;  // a group sees at most the forest's distinct count
        // u8 codes (histogram-trained forests: <= 254 thresholds per feature): decided here, before the node words are encoded --
        // such a handle has its own node encoding and only ever quantises to u8 (large batches, remainders and tree slices alike)
        const char *e8 = getenv("TAHOE_QRING_CODE8");  // experiments: 0 keeps u16 codes
        q->code8 = q->reg && most <= (size_t)kQMaxTable8 && qreg_lds_for(6, kReg8Walkers, kReg8Ring, true) <= f->lds_limit && !(e8 && atoi(e8) == 0);
        if (const char *k = getenv("TAHOE_QRING_GROUPS"))  // experiments: at least this many groups (K4: 8 groups of 1000 trees
            G = std::max(G, (size_t)std::max(atoi(k), 1));  // take the bucketed quantise kernel, 4 of 2000 the two-pass one)
    }
    const size_t bytes_before = f->device_bytes;
    for (;;) {
        if (G > T) G = T;
        bool fits = true;
        int worst = 0;
        for (size_t k = 0; k < G && fits; ++k) {
            const size_t lo = T * k / G, hi = T * (k + 1) / G;
            tahoe_qgroup g;
            int too_many = 0;
            const tahoe_status s = build_group(f, h_inner, h_real, h_leaf, lo, hi, g, &too_many);
            if (s != TAHOE_OK) {
                free_group(g);
                return s;  // create() destroys the handle, which frees the finished groups
            }
            if (too_many) {
                fits = false;
                worst = too_many;
            } else {
                q->groups.push_back(g);
            }
        }
        if (fits) break;
        for (tahoe_qgroup &g : q->groups) free_group(g);
        q->groups.clear();
        f->device_bytes = bytes_before;
        if (G == T) {  // a single tree exceeds the limit: the strategy is unavailable
            qring_destroy(f);
            return TAHOE_OK;
        }
        G = std::max(G + 1, (size_t)((double)G * worst / kQMaxTable + 0.999));
    }
    {   // <= 128 features: regions at a 16-KiB LDS stride
        const char *e6 = getenv("TAHOE_QRING_NARROW128");  // experiments: 0 keeps the 32-KiB region stride for forests of <= 128 features
        q->narrow128 = q->reg && cols <= 128 && qreg_lds_for(6, kReg8Walkers, kReg8Ring, false, kRegBytes / 2) <= f->lds_limit &&
                       qreg_lds_for(6, 15, kQRing, true, kRegBytes / 2) <= f->lds_limit && !(e6 && atoi(e6) == 0);
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
                          (const void *)&qwide_kernel<32, true>, (const void *)&qwide_kernel<16, false>, (const void *)&qwide_kernel<16, true>,
                          (const void *)&qwide_kernel<64, false, 3, kWideRingBytesK>, (const void *)&qwide_kernel<64, true, 3, kWideRingBytesK>,
                          (const void *)&qwide_kernel<32, false, 3, kWideRingBytesK>, (const void *)&qwide_kernel<32, true, 3, kWideRingBytesK>,
                          (const void *)&qwide_kernel<16, false, 3, kWideRingBytesK>, (const void *)&qwide_kernel<16, true, 3, kWideRingBytesK>})
        if ((e = allow_max_lds(k, f->lds_limit)) != hipSuccess) return bad("attr(qwide)");
    if ((e = quantize_allow_lds(f)) != hipSuccess) return bad("attr(quantise kernels)");
    return TAHOE_OK;
}

void qring_destroy(tahoe_forest *f)
{
    tahoe_qstate *q = f->q;
    if (!q) return;
    for (tahoe_qgroup &g : q->groups) free_group(g);
    if (q->xq) (void)hipFree(q->xq);
    if (q->leafbuf) (void)hipFree(q->leafbuf);
    if (q->chunk_flags) (void)hipFree(q->chunk_flags);
    delete q;
    f->q = nullptr;
}

int qring_groups(const tahoe_forest *f) { return f->q ? (int)f->q->groups.size() : 0; }
bool qring_regions(const tahoe_forest *f) { return f->q && f->q->reg; }
bool qring_code8(const tahoe_forest *f) { return f->q && f->q->code8; }
bool qring_six16(const tahoe_forest *f) { return f->q && f->q->reg && f->q->narrow128; }

// The quantised copy of the batch lives in a grow-only workspace owned by the handle.
static int q_slices(const tahoe_forest *f, size_t rows, int *most_out);
static tahoe_status qring_reserve_leafbuf(tahoe_forest *f, size_t rows, int trees);

tahoe_status qring_reserve(tahoe_forest *f, size_t rows)
{
    tahoe_qstate *q = f->q;
    if (!q) return TAHOE_OK;  // no quantised form on this handle
    size_t tiles = (rows + kQRows - 1) / kQRows;
    // region form: a walk tile reads two or three whole 64-row regions -> room for the last tile to read past the batch
    // (six-region tiles of narrow forests: up to five regions past the last row)
    if (q->reg) tiles = ((rows + kRegRows - 1) / kRegRows + (q->narrow128 ? 5 : 2) + 1) / 2;
    if (tiles * kQRows > q->xq_rows) {
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
        if (q->chunk_flags) (void)hipFree(q->chunk_flags);
        q->chunk_flags = nullptr;
        q->n_chunk_flags = (q->xq_rows + kQuantMinRowsPerBlock - 1) / kQuantMinRowsPerBlock + 1;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&q->chunk_flags), q->n_chunk_flags * sizeof(uint32_t));
        if (e != hipSuccess) {  // all or nothing: the next reserve starts over
            q->chunk_flags = nullptr;
            (void)hipFree(q->xq);
            f->device_bytes -= bytes;
            q->xq = nullptr;
            q->xq_rows = 0;
            return fail(TAHOE_ERR_HIP, "qring_reserve: hipMalloc failed: %s", hipGetErrorString(e));
        }
    }
    // Batches small enough to be walked in tree slices (SPLIT) also need the leaf-value buffer.  Every batch of up to `rows`
    // rows must find it in place -- also a small one after a large reserve -- so it is sized for the largest split batch
    // that fits in `rows` (slices fall to 1 once the 128-row tiles outnumber half the CUs).
    int most = 1;
    const size_t split_max = f->knob_qring_slices >= 1 ? rows : std::min<size_t>(rows, (size_t)std::max(f->num_cus, 2) / 2 * 128);
    if (q_slices(f, split_max, &most) > 1) return qring_reserve_leafbuf(f, split_max, most);
    return TAHOE_OK;
}

// Small batches of the region form: with fewer 128-row tiles than CUs every tile is walked by `slices` workgroups, each a
// slice of the trees (SPLIT) -- while a slice still gives every walker a few trees.  *most = trees of the largest group.
static int q_slices(const tahoe_forest *f, size_t rows, int *most_out)
{
    const tahoe_qstate *q = f->q;
    int most = 1;
    for (const tahoe_qgroup &g : q->groups) most = std::max(most, g.num_trees);
    *most_out = most;
    if (!q->reg || q->sparse || rows == 0) return 1;
    const size_t tiles = (rows + 127) / 128;
    const int fit = (int)std::min<size_t>((size_t)std::max(f->num_cus, 1) / tiles, 8);
    int slices = std::max(1, std::min(fit, most / (4 * 15)));
    if (f->knob_qring_slices >= 1) slices = std::min(f->knob_qring_slices, most);
    return slices;
}

// Leaf-value buffer of the SPLIT form, [trees of the largest group][rows rounded up to 64] floats, grow-only like the
// quantised workspace (tahoe_forest_reserve sizes it too when the batch is small enough to be split).
static tahoe_status qring_reserve_leafbuf(tahoe_forest *f, size_t rows, int trees)
{
    tahoe_qstate *q = f->q;
    const size_t stride = (rows + 63) / 64 * 64;
    if (q->leafbuf && stride <= q->leaf_stride && (size_t)trees <= q->leaf_trees) return TAHOE_OK;
    if (q->leafbuf) {
        TAHOE_HIP_TRY(hipDeviceSynchronize());
        TAHOE_HIP_TRY(hipFree(q->leafbuf));
        f->device_bytes -= q->leaf_stride * q->leaf_trees * sizeof(float);
        q->leafbuf = nullptr;
    }
    const size_t ns = std::max(stride, q->leaf_stride), nt = std::max<size_t>((size_t)trees, q->leaf_trees);
    TAHOE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&q->leafbuf), ns * nt * sizeof(float)));
    q->leaf_stride = ns;
    q->leaf_trees = nt;
    f->device_bytes += ns * nt * sizeof(float);
    return TAHOE_OK;
}

template <int NWALK, bool LDSX = true, bool NARROW = false, bool EXCH = false, int K = 2, bool REG = false, int RING = kQRing,
          bool SPLIT = false, int BATCH = (RING >= 2 * kQBatch ? kQBatch : RING / 2), bool CODE8 = false, bool DEP = false, int REGB = kRegBytes>
static void q_launch(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                     size_t rows, hipStream_t stream, int cshift, int slices = 1, size_t row_begin = 0)
{
    tahoe_qstate *q = f->q;
    const unsigned grid = (unsigned)((rows - row_begin + 64 * K - 1) / (64 * K)) * (unsigned)(SPLIT ? slices : 1);
    if (grid == 0) return;
    float *leafbuf = SPLIT ? q->leafbuf : nullptr;
    const size_t leaf_stride = SPLIT ? q->leaf_stride : 0;
    const int lds = REG ? (int)qreg_lds_for(K, NWALK, RING, CODE8, REGB) : (int)qring_lds_for(f, NWALK, LDSX);
    const uint32_t *leaf_orig = f->leaf_orig + (size_t)g.tree_lo * f->n_leaf;
    if (leaf_out)
        hipLaunchKernelGGL((qring_kernel<NWALK, true, LDSX, NARROW, EXCH, K, REG, RING, SPLIT, BATCH, CODE8, DEP, REGB>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, q->xq, g.top,
                           g.blocks, g.qinner, leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth,
                           q->top_levels, q->top_stride, q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, cshift, leafbuf, leaf_stride,
                           slices, row_begin);
    else
        hipLaunchKernelGGL((qring_kernel<NWALK, false, LDSX, NARROW, EXCH, K, REG, RING, SPLIT, BATCH, CODE8, DEP, REGB>), dim3(grid), dim3((NWALK + 1) * 64), lds, stream, q->xq, g.top,
                           g.blocks, g.qinner, leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth,
                           q->top_levels, q->top_stride, q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, cshift, leafbuf, leaf_stride,
                           slices, row_begin);
    if (SPLIT && sums)
        hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)((rows - row_begin + kOrderedSumThreads - 1) / kOrderedSumThreads)), dim3(kOrderedSumThreads), 0, stream,
                           leafbuf, leaf_stride, g.num_trees, sums_in, sums, rows, row_begin);
}

template <int RT, int KG, int RB>
static void qwide_launch_form(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                              size_t rows, hipStream_t stream, int cshift)
{
    tahoe_qstate *q = f->q;
    const unsigned grid = (unsigned)((rows + RT - 1) / RT);
    const int lds = (int)qwide_lds_for(f, RT, q->wide_lw, KG);
    const int slot_bytes = KG * (int)qwide_slot_bytes(q->wide_lw, RT);
    const uint32_t *leaf_orig = f->leaf_orig + (size_t)g.tree_lo * f->n_leaf;
    if (leaf_out)
        hipLaunchKernelGGL((qwide_kernel<RT, true, KG, RB>), dim3(grid), dim3(16 * 64), lds, stream, q->xq, g.top, g.blocks, g.qinner,
                           leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth, q->top_levels, q->top_stride,
                           q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, slot_bytes, q->wide_lw, cshift);
    else
        hipLaunchKernelGGL((qwide_kernel<RT, false, KG, RB>), dim3(grid), dim3(16 * 64), lds, stream, q->xq, g.top, g.blocks, g.qinner,
                           leaf_orig, sums, leaf_out, rows, f->p.num_cols, g.num_trees, f->depth, q->top_levels, q->top_stride,
                           q->chunk_flags, f->error_flag, sums_in, g.tree_lo, f->p.num_trees, slot_bytes, q->wide_lw, cshift);
}

template <int RT>
static void qwide_launch(tahoe_forest *f, const tahoe_qgroup &g, float *sums, const float *sums_in, uint32_t *leaf_out,
                         size_t rows, hipStream_t stream, int cshift)
{
    if (f->q->wide_kg == 3)
        qwide_launch_form<RT, 3, kWideRingBytesK>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
    else
        qwide_launch_form<RT, 1, kWideRingBytes>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
}

tahoe_status qring_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                          hipStream_t stream, hipEvent_t mid_event, const float *sums_in0)
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
    const int trs = q->reg ? 6 : wide == 64 ? 6 : wide == 32 ? 5 : wide == 16 ? 4 : 7;
    size_t rows3 = 0;   // rows [0, rows3) in 192-row tiles, a multiple of 384
    int chains = 2;     // form of the remaining rows [rows3, rows)
    int most = 1;
    const int slices = q_slices(f, rows, &most);  // small batches of the region form: tree slices per tile (SPLIT)
    const bool code8 = q->code8;                  // u8 codes: 384-row tiles for whole waves, 128-row tiles for the remainder and for tree slices
    const bool six16 = !code8 && q->reg && q->narrow128 && slices <= 1;  // u16 codes, <= 128 features: six 16-KiB regions, the same plan
    if ((code8 && slices <= 1) || six16)
        qreg_plan(rows, f->num_cus, f->knob_qring_chains, &rows3, &chains, kReg8Cost, 384, f->knob_qring_slices == 1 ? 0 : most);
    else if (q->reg)  // TAHOE_QRING_CHAINS = 2 / 3 forces one form, TAHOE_QRING_SLICES = 1 keeps every remainder in plain tiles
        qreg_plan(rows, f->num_cus, f->knob_qring_chains, &rows3, &chains, 133, 192, f->knob_qring_slices == 1 ? 0 : most);
    if (slices > 1) {
        const tahoe_status ls = qring_reserve_leafbuf(f, rows, most);  // no-op after tahoe_forest_reserve / a first predict
        if (ls != TAHOE_OK) return ls;
        chains = 2;
        rows3 = 0;
    }
    // A remainder of few 128-row tiles behind the whole waves of large tiles would leave most of the chip idle for a whole tile
    // time (250 k rows of K3's forest: 5 waves of 192-row tiles + 34 tiles of 128): it is walked in tree slices too -- every
    // remainder tile by rem_slices workgroups, then the ordered sum over those rows (bit-identical: the same sequential sum).
    int rem_slices = 1;
    if (q->reg && slices <= 1 && rows3 > 0 && chains == 2 && f->knob_qring_slices != 1) {
        rem_slices = qreg_rem_slices(rows - rows3, f->num_cus, most);  // the planner priced the remainder with the same rule
        if (rem_slices > 1) {
            const tahoe_status ls = qring_reserve_leafbuf(f, rows - rows3, most);  // no-op after tahoe_forest_reserve / a first predict
            if (ls != TAHOE_OK) return ls;
        }
    }
    bool first = true;
    for (const tahoe_qgroup &g : q->groups) {  // stream order: quantise for the group, walk the group, next group
        TAHOE_HIP_TRY(hipMemsetAsync(q->chunk_flags, 0, q->n_chunk_flags * sizeof(uint32_t), stream));
        int cshift = 0;  // rows per quantise workgroup = rows per "missing seen" flag, as a shift
        {
            const tahoe_status qs = quantize_launch(f, g, data, rows, trs, code8 ? 2 : q->reg ? 1 : 0, stream, &cshift);
            if (qs != TAHOE_OK) return qs;
        }
        TAHOE_HIP_TRY(hipGetLastError());
        if (first && mid_event) TAHOE_HIP_TRY(hipEventRecord(mid_event, stream));  // splits pre-pass / walk for 1 group
        const float *sums_in = first ? sums_in0 : sums;  // later groups continue the running float32 sums
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
                if (q->reg && slices > 1 && code8)
                    q_launch<15, true, true, false, 2, true, kQRing, true, kQBatch, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, slices);
                else if (q->reg && slices > 1)
                    q_launch<15, true, true, false, 2, true, kQRing, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, slices);
                else if (code8 && q->narrow128) {  // u8 codes, <= 128 features: three 16-KiB regions, 15 walkers, ring of 24
                    if (rows3 > 0)
                        q_launch<15, true, true, false, 6, true, kQRing, false, kQBatch, true, kReg8Dep, kRegBytes / 2>(f, g, sums, sums_in, leaf_out, rows3, stream, cshift);
                    if (chains == 3)
                        q_launch<15, true, true, false, 6, true, kQRing, false, kQBatch, true, kReg8Dep, kRegBytes / 2>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1, rows3);
                    else if (rem_slices > 1)
                        q_launch<15, true, true, false, 2, true, kQRing, true, kQBatch, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, rem_slices, rows3);
                    else
                        q_launch<15, true, true, false, 2, true, kQRing, false, kQBatch, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1, rows3);
                }
                else if (code8) {
                    if (rows3 > 0)
                        q_launch<kReg8Walkers, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, true, kReg8Dep>(f, g, sums, sums_in, leaf_out, rows3, stream, cshift);
                    if (chains == 3)
                        q_launch<kReg8Walkers, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, true, kReg8Dep>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1,
                                                                                                         rows3);
                    else if (rem_slices > 1)
                        q_launch<15, true, true, false, 2, true, kQRing, true, kQBatch, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, rem_slices, rows3);
                    else
                        q_launch<15, true, true, false, 2, true, kQRing, false, kQBatch, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1, rows3);
                }
                else if (six16) {  // u16 codes, <= 128 features: six 16-KiB regions = 384-row tiles, the u8 tile's walkers and ring
                    if (rows3 > 0)
                        q_launch<kReg8Walkers, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, false, kReg8Dep, kRegBytes / 2>(f, g, sums, sums_in, leaf_out, rows3, stream, cshift);
                    if (chains == 3)
                        q_launch<kReg8Walkers, true, true, false, 6, true, kReg8Ring, false, kReg8Batch, false, kReg8Dep, kRegBytes / 2>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1,
                                                                                                                        rows3);
                    else if (rem_slices > 1)
                        q_launch<15, true, true, false, 2, true, kQRing, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, rem_slices, rows3);
                    else
                        q_launch<15, true, true, false, 2, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1, rows3);
                }
                else if (q->reg) {
                    if (rows3 > 0)
                        q_launch<kReg3Walkers, true, true, false, 3, true, kReg3Ring, false, kReg3Batch, false, kReg3Dep>(f, g, sums, sums_in, leaf_out, rows3, stream, cshift);
                    if (chains == 3)
                        q_launch<kReg3Walkers, true, true, false, 3, true, kReg3Ring, false, kReg3Batch, false, kReg3Dep>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1,
                                                                                      rows3);
                    else if (rem_slices > 1)
                        q_launch<15, true, true, false, 2, true, kQRing, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, rem_slices, rows3);
                    else
                        q_launch<15, true, true, false, 2, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift, 1, rows3);
                }
                else if (q->narrow && f->relayout)
                    q_launch<15, true, true, true>(f, g, sums, sums_in, leaf_out, rows, stream, cshift);
                else if (q->narrow)
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

// The form qring_launch takes for a batch of `rows` rows (TAHOE_FORM_*): the same decisions, nothing launched.
int qring_form(const tahoe_forest *f, size_t rows)
{
    const tahoe_qstate *q = f->q;
    if (!q || qring_walkers(f) == 0) return TAHOE_FORM_NONE;
    if (qwide_rows(f)) return TAHOE_FORM_QRING_WIDE;
    if (!qring_lds_tile(f)) return TAHOE_FORM_QRING_GX;
    if (!q->reg) return TAHOE_FORM_QRING_COLUMNS;
    int most = 1;
    if (q_slices(f, rows, &most) > 1) return TAHOE_FORM_QRING_SPLIT;
    size_t rows3 = 0;
    int chains = 2;
    if (q->code8) return TAHOE_FORM_QRING_REGION8;
    if (q->narrow128) return TAHOE_FORM_QRING_REGION6;
    qreg_plan(rows, f->num_cus, f->knob_qring_chains, &rows3, &chains, 133, 192, f->knob_qring_slices == 1 ? 0 : most);
    if (rows3 > 0 && chains == 2) return TAHOE_FORM_QRING_REGION_MIXED;
    return chains == 3 ? TAHOE_FORM_QRING_REGION3 : TAHOE_FORM_QRING_REGION2;
}

}  // namespace tahoe
