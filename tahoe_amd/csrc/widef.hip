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

#include "widef_internal.h"

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
int widef_stream_slots(const tahoe_forest *f) { return f->wf && f->wf->s_on ? f->wf->s_slots : 0; }
int widef_stream_levels(const tahoe_forest *f) { return f->wf && f->wf->s_on ? f->wf->s_lw : 0; }
float widef_stream_tie_estimate(const tahoe_forest *f) { return f->wf ? f->wf->key_tie_estimate : 0.f; }
long long widef_lds_bytes(const tahoe_forest *f)
{
    const tahoe_wstate *w = f->wf;
    if (!w) return 0;
    return w->s_on ? wkey_lds_bytes(f) : wf_lds(f->p.num_cols, w->rt, w->nwalk, w->lw);
}

tahoe_status widef_reserve(tahoe_forest *f, size_t rows) { return f->wf ? wkey_reserve(f, rows) : TAHOE_OK; }

void widef_destroy(tahoe_forest *f)
{
    if (!f->wf) return;
    if (f->wf->ftop) (void)hipFree(f->wf->ftop);
    if (f->wf->fblocks) (void)hipFree(f->wf->fblocks);
    wkey_free(f->wf);
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
tahoe_status widef_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                         const std::vector<float> &h_leaf)
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
    return wkey_build(f, h_inner, h_real, h_leaf);  // the row-streaming form on 16-bit keys (wkey.hip), where it applies
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
    if (w->s_on && vec4_ok && rows > 0) return wkey_launch(f, sums, leaf_out, data, rows, stream, sums_in);
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
