// Shared between quantize.hip (float32 rows -> u16 rank codes) and qring.hip (the walk on those codes):
// the per-group tables and views, constants of the tile layout, and the entry points each file offers the other.
#ifndef TAHOE_AMD_QRING_INTERNAL_H
#define TAHOE_AMD_QRING_INTERNAL_H

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

// ---- quantize.hip ----
// Builds the group's threshold tables in every form the quantise kernels use from `tab` (per feature: sorted distinct
// non-NaN thresholds, at most kQMaxTable), picks the kernel forms, uploads.  Sets g.max_table.
tahoe_status quantize_build_tables(tahoe_forest *f, const std::vector<std::vector<float>> &tab, tahoe_qgroup &g);
void quantize_free_tables(tahoe_qgroup &g);
hipError_t quantize_allow_lds(const tahoe_forest *f);  // kernels that need more than 64 KiB of dynamic LDS
// Launches the quantise pass of group g for `rows` rows of `data` into f->q->xq (tiles of 2^trs rows) and the
// per-chunk "missing seen" flags; *cshift_out = log2 of the rows per flag.
tahoe_status quantize_launch(tahoe_forest *f, const tahoe_qgroup &g, const float *data, size_t rows, int trs, hipStream_t stream,
                             int *cshift_out);

}  // namespace tahoe

#endif  // TAHOE_AMD_QRING_INTERNAL_H
