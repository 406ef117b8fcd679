// Definitions shared by the translation units that implement the forest operator (forest.hip,
// qring.hip).  Internal: not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"

struct tahoe_qstate;  // quantised views + workspace, owned by qring.hip
struct tahoe_sstate;  // sparse (irregular) forest, owned by sparse.hip
struct tahoe_pstate;  // host-batch upload pipeline, owned by pipeline.hip
struct tahoe_wstate;  // float32 walk for wide rows, owned by widef.hip

namespace tahoe {

struct InnerNode {
    float thr;
    uint32_t meta;  // fid (30 bits, FID_MASK of Struct.h:57) | exchange << 30 | def_left << 31
};
constexpr uint32_t kMetaFidMask = 0x3fffffffu;
constexpr uint32_t kMetaExchange = 1u << 30;  // probability-guided re-layout: children stored swapped, condition inverted
static_assert(sizeof(InnerNode) == 8, "InnerNode must be 8 bytes");

constexpr int kBlock = 256;             // threads per workgroup of DIRECT / ROWTILE (4 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kTileRows = 64;           // rows per ROWTILE workgroup = one wave of lanes
constexpr int kMaxLdsLevels = 8;        // ROWTILE: top levels staged per wave (255 nodes = 2040 B)
constexpr float kMissingEps = 1.0e-6f;  // BaseTahoeTest.h:451

// TILEBLOCK geometry
constexpr int kTopLevelsMax = 10;       // top levels kept in LDS (1023 nodes: 4 KiB thr + 2 KiB meta)
constexpr int kSlots = 4;               // trees in flight per workgroup
constexpr int kBlockFidBits = 9;        // bottom blocks pack 3 x (fid:9 | def_left:1) in one dword
constexpr int kBlockMaxCols = 1 << kBlockFidBits;

static inline int align16(int x) { return (x + 15) & ~15; }
static inline int top_nodes(int top_levels) { return (1 << top_levels) - 1; }
// A staged top is indexed by 1-based heap position (entry 0 unused), so that the two children of
// position i form the aligned pair (2i, 2i+1).
static inline int top_thr_bytes(int top_levels) { return align16((top_nodes(top_levels) + 1) * 4); }
static inline int top_stride_bytes(int top_levels)
{
    return top_thr_bytes(top_levels) + align16((top_nodes(top_levels) + 1) * 2);
}

}  // namespace tahoe

struct tahoe_forest {
    tahoe_forest_params p{};
    int depth = 0;        // De: depth of the normalised trees, max(p.depth, 2)
    size_t n_inner = 0;   // 2^De - 1
    size_t n_leaf = 0;    // 2^De
    int bits_bytes = 0;
    int strategy = TAHOE_STRATEGY_AUTO;
    int device = 0;
    int num_cus = 0;
    int lds_limit = 0;
    int lds_levels = 0;   // ROWTILE
    int top_levels = 0;   // TILEBLOCK: levels in LDS, min(De - 2, 10)
    bool has_blocks = false;
    tahoe::InnerNode *inner = nullptr;
    float *leaf_val = nullptr;
    uint32_t *leaf_orig = nullptr;
    unsigned char *top = nullptr;  // [T][top_stride]
    uint4 *blocks = nullptr;       // [T][2^(De-2)][2]
    int *error_flag = nullptr;     // set by TILERING if a bounded spin ever times out
    tahoe_qstate *q = nullptr;     // QRING: rank-quantised forest + row workspace (qring.hip)
    tahoe_sstate *sp = nullptr;    // non-null: this handle is a sparse forest (sparse.hip); the dense views are unused
    tahoe_pstate *pipe = nullptr;  // tahoe_forest_predict_host: chunk buffers, streams, events (created on first use)
    tahoe_wstate *wf = nullptr;    // non-null: TILERING runs the wide-row float32 form (widef.hip)
    size_t device_bytes = 0;
    // Tuning knobs for experiments, read from the environment ONCE, in tahoe_forest_create (never on the predict path):
    // TAHOE_TILE_ROWS (64 / 128: rows per TILEBLOCK / TILERING tile), TAHOE_QRING_WALKERS (15 / 12 / 8 / 4).  0 = unset.
    // Probability-guided re-layout (TAHOE_CREATE_PROB_RELAYOUT; Struct.h:1775-1825): subtrees swapped so that the likelier
    // child is the left one, nodes carry an exchange bit.  Served by DIRECT, ROWTILE and the NARROW form of QRING.
    bool relayout = false;
    size_t relayout_swaps = 0;
    int knob_tile_rows = 0;
    int knob_qring_walkers = 0;
    int knob_qring_slices = 0;  // TAHOE_QRING_SLICES >= 1: force the tree slices per tile of QRING's SPLIT form
    int knob_qring_chains = 0;  // TAHOE_QRING_CHAINS = 2 / 3: force the tile form of QRING's region layout
    // Profiling: one hipEvent pair per traversal launch, read back after the stream has drained.
    bool profiling = false;
    std::vector<hipEvent_t> ev_start, ev_mid, ev_stop;  // mid: between a pre-pass kernel and the walk kernel
    size_t prof_count = 0;  // launches recorded since profiling was (re-)enabled
};

namespace tahoe {

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

// Hand-over of values through LDS inside one workgroup: the producer stores its values, then the flag (or bumps a counter);
// the consumer reads the flag, then the values.  Both sides rely on "the LDS performs one wave's operations in issue order"
// and need only keep the COMPILER from reordering (TAHOE_LDS_RELEASE / TAHOE_LDS_ACQUIRE = compiler barriers).  Building with
// -DTAHOE_RING_RELEASE_ACQUIRE (make RING_FENCES=1) drains the wave's LDS queue at both points instead, so that a suspected
// ring failure can be bisected in one run (costs ~2 % on K3).
#ifdef TAHOE_RING_RELEASE_ACQUIRE
#define TAHOE_LDS_RELEASE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define TAHOE_LDS_ACQUIRE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define TAHOE_LDS_RELEASE() asm volatile("" ::: "memory")
#define TAHOE_LDS_ACQUIRE() asm volatile("" ::: "memory")
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is per function and process-wide, not per handle: always raise it to
// the device limit (less the kernel's static LDS), so that handles of different shapes can coexist in one process.
inline hipError_t allow_max_lds(const void *fn, int limit)
{
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, limit - (int)a.sharedSizeBytes);
}

// The branch rule of infer_one_tree, BaseTahoeTest.h:450-453: 1 = right child.
__device__ __forceinline__ uint32_t go_right(float x, float thr, bool def_left, float missing)
{
    const bool is_missing = fabsf(x - missing) <= kMissingEps;
    const bool cond = is_missing ? !def_left : (x >= thr);
    return cond ? 1u : 0u;
}
// ... on a heap record: the stored children are swapped where the exchange bit is set (Struct.h:1060-1063: cond = !cond)
__device__ __forceinline__ uint32_t go_right_meta(float x, float thr, uint32_t meta, float missing)
{
    return go_right(x, thr, (meta >> 31) != 0, missing) ^ ((meta >> 30) & 1u);
}

// QRING entry points (qring.hip).  h_real[i] != 0 marks heap records that exist in the original tree
// (padding below an early leaf is not real and contributes no threshold).
tahoe_status qring_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                         const std::vector<float> &h_leaf);
void qring_destroy(tahoe_forest *f);
int qring_walkers(const tahoe_forest *f);  // walker waves the kernel would use; 0 = strategy unavailable
long long qring_lds_bytes(const tahoe_forest *f);
// mid_event (optional) is recorded between the quantise kernel and the walk kernel
tahoe_status qring_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                          hipStream_t stream, hipEvent_t mid_event, const float *sums_in = nullptr);
tahoe_status qring_reserve(tahoe_forest *f, size_t rows);
int qwide_rows(const tahoe_forest *f);   // rows per tile of the wide-row form; 0 = not used
int qwide_chains(const tahoe_forest *f); // ... and the trees a lane walks at once (1 or 3)
bool qring_lds_tile(const tahoe_forest *f);
bool qring_regions(const tahoe_forest *f);  // region form: tiles of 192 (or 128) rows as 64-row regions
bool qring_six16(const tahoe_forest *f);    // ... <= 128 features: 16-KiB region stride, 384-row tiles also on u16 codes
bool qring_code8(const tahoe_forest *f);    // ... on u8 codes (<= 254 thresholds per feature): tiles of 384 rows as 128-row regions
int qring_groups(const tahoe_forest *f);  // tree groups with separate quantisation (1 for most forests)
int qring_form(const tahoe_forest *f, size_t rows);  // TAHOE_FORM_* of the launch for a batch of `rows` rows

// sparse forests (sparse.hip)
bool sparse_tile_fits(const tahoe_forest *f);
tahoe_status sparse_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows,
                           hipStream_t stream, int strategy, const float *sums_in = nullptr);
int sparse_top_waves(const tahoe_forest *f);
bool sparse_q_available(const tahoe_forest *f);  // the walk on quantised codes (strategy QRING on a sparse handle)
void sparse_destroy(tahoe_forest *f);
void pipeline_destroy(tahoe_forest *f);
// TILERING for rows too wide for a 64-row float32 tile (widef.hip)
tahoe_status widef_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                         const std::vector<float> &h_leaf);
int widef_rows(const tahoe_forest *f);  // rows per tile; 0 = unavailable
long long widef_lds_bytes(const tahoe_forest *f);   // LDS per workgroup of the form the launch takes
int widef_stream_slots(const tahoe_forest *f);      // row slots of the row-streaming form; 0 = the tile form runs
int widef_stream_levels(const tahoe_forest *f);     // ... and the levels of all trees it keeps in LDS
float widef_stream_tie_estimate(const tahoe_forest *f);  // estimated share of key compares that tie (0 when the form was never sized)
tahoe_status widef_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows, hipStream_t stream,
                          const float *sums_in);
void widef_destroy(tahoe_forest *f);
tahoe_status widef_reserve(tahoe_forest *f, size_t rows);

}  // namespace tahoe
