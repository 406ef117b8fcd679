// Shared by widef.hip (the tile form of the float32 walk for wide rows) and wkey.hip (its row-streaming form on 16-bit keys).
// Internal: not part of the ABI.
#pragma once
#include <vector>

#include "forest_internal.h"

struct tahoe_wstate {
    // tile form (widef_kernel)
    uint2 *ftop = nullptr;     // [T][tstride] {thr bits, meta}: heap node i at entry i + 1, the first 2^lw - 1 nodes
    uint4 *fblocks = nullptr;  // [T][2^(De-2)][3]
    int rt = 0, nwalk = 0, lw = 0, tstride = 0;
    // row-streaming form on keys (wkey_kernel)
    unsigned char *kimg = nullptr;  // tops of ALL trees, node-major: [2^s_lw - 1][s_ts] u32 key << 16 | fid << 1 | def_left, padded to 1 KiB
    uint4 *kblocks = nullptr;       // [T][2^(De-2)][2]: {n0, n1, n2, leaf0} {leaf1, leaf2, leaf3, -} -- the last two levels in 32 bytes
    int s_lw = 0, s_ts = 0, s_slots = 0, s_img_bytes = 0;
    float *leafbuf = nullptr;       // workspace [leaf_rows][trees rounded up to 32]: leaf values of a batch, added in tree order by the kernel's summer wave
    size_t leaf_rows = 0;
    size_t slab_bytes = 0;          // cap of the workspace: larger batches are walked in slabs of rows
    float key_lo = 0.f, key_scale = 0.f;  // key(x) = trunc(clamp((x - key_lo) * key_scale, 0, 65534))
    float key_tie_estimate = 0.f;   // estimated share of compares whose keys tie (wkey_build); the form is taken only below kWkTieLimit
    bool s_on = false;              // the launch takes it
};

namespace tahoe {

tahoe_status wkey_build(tahoe_forest *f, const std::vector<InnerNode> &h_inner, const std::vector<unsigned char> &h_real,
                        const std::vector<float> &h_leaf);
void wkey_free(tahoe_wstate *w);
long long wkey_lds_bytes(const tahoe_forest *f);
tahoe_status wkey_reserve(tahoe_forest *f, size_t rows);  // the leaf-value workspace for batches of up to `rows` rows
tahoe_status wkey_launch(tahoe_forest *f, float *sums, uint32_t *leaf_out, const float *data, size_t rows, hipStream_t stream,
                         const float *sums_in);

}  // namespace tahoe
