/*
 * tahoe_amd.h -- C ABI of libtahoe_amd.so: the MI355X (gfx950) implementation of Tahoe's batched
 * tree-ensemble traversal.  Plain pointers and sizes only; no HIP, torch or C++ types.
 *
 * This is the drop-in boundary for the reference's forest operator API (its layer L3):
 *   reference                                        -> entry point here
 *   init_dense / init_dense_adaptive                 -> tahoe_forest_create
 *     (BaseTahoeTest.h:519-525, :605-611; dense_forest::init Struct.h:815-833;
 *      dense_adaptive_forest::init Struct.h:1756-1986)
 *   predict_dense / predict_dense_adaptive           -> tahoe_forest_predict
 *     (BaseTahoeTest.h:544-547, :599-602; forest::predict Struct.h:245-269)
 *   global `selected_algorithm` (Struct.h:11)        -> tahoe_forest_set_strategy (per handle)
 *   delete forest (BaseTahoeTest.h:594,709; leaks)   -> tahoe_forest_destroy (frees device memory)
 *   generate_forest_from_file / generate_data_from_file (BaseTahoeTest.h:267-402)
 *                                                    -> tahoe_load_model / tahoe_load_data
 *   allocate/updateDevice/updateHost (cuda_base.h:28-50), compare_GPU (cuda_base.h:98-111)
 *                                                    -> tahoe_device_* / tahoe_compare_device
 * Every device pointer is a HIP device pointer on the handle's device; `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  All predict calls are asynchronous on `stream`
 * and allocate nothing, with one exception: two strategies keep a grow-only workspace on the handle --
 * QRING a 2-byte-per-value quantised copy of the batch (plus one float per (tree, row) for the small batches it
 * walks in tree slices), and the row-streaming form of TILERING (kernel form TAHOE_FORM_TILERING_WIDE_STREAM,
 * which the create-time shape rule may pick with no environment variable set) one float per (row, tree rounded
 * up to 32), capped at 1 GiB (TAHOE_WSTREAM_SLAB_MB, read at create; larger batches are walked in slabs of rows).
 * A batch larger than any before grows the workspace inside predict: hipDeviceSynchronize + hipFree + hipMalloc,
 * which is illegal during stream capture.  tahoe_forest_reserve(rows) sizes everything beforehand; after it no
 * batch of up to `rows` rows allocates, and predict may be captured into a HIP graph.
 *
 * Handles share no global state and different handles may be used from different threads and streams
 * at the same time.  One handle serves one predict at a time: its workspace is reused by the next call,
 * so calls on the same handle must be ordered on one stream (or separated by a synchronisation).  A
 * process that drives several GPUs keeps one handle per device; predict switches to the handle's
 * device for its launches and restores the caller's.
 *
 * There is no CPU fallback: every compute entry point fails with TAHOE_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef TAHOE_AMD_H
#define TAHOE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAHOE_AMD_ABI_VERSION 2

/* The library is built with -fvisibility=hidden and a linker version script: the declarations below are its only
 * exported symbols (tests/test_formats_capi.py checks `nm -D`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- status codes (the reference prints and continues, cuda_base.h:19-25; we return codes) ---- */
typedef enum {
    TAHOE_OK = 0,
    TAHOE_ERR_INVALID_ARG = 1,   /* NULL pointer, negative size, bad enum (check_params, BaseTahoeTest.h:490-516) */
    TAHOE_ERR_IO = 2,            /* file cannot be opened (BaseTahoeTest.h:273-277, :360-364) */
    TAHOE_ERR_NO_MEMORY = 3,
    TAHOE_ERR_NO_DEVICE = 4,     /* no usable gfx950 device: there is no CPU path */
    TAHOE_ERR_HIP = 5,           /* a HIP runtime call failed; text in tahoe_last_error() */
    TAHOE_ERR_INVALID_FOREST = 6,/* a reachable bottom-level node is not a leaf / fid >= num_cols */
    TAHOE_ERR_UNSUPPORTED = 7    /* requested strategy cannot run this shape */
} tahoe_status;

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *tahoe_last_error(void);
int tahoe_abi_version(void);

/* ---- node encoding: dense_node_t, Struct.h:44-48, masks Struct.h:57-59 ---- */
typedef struct {
    float weight; /* branch probability; unused at inference */
    float val;    /* threshold (internal node) or leaf value */
    int32_t bits; /* fid[0:29] | def_left<<30 | is_leaf<<31 */
} tahoe_dense_node;

/* encode_node Struct.h:103-108 / dense_node_decode Struct.h:110-117 */
void tahoe_encode_node(tahoe_dense_node *n, int fid, float value, int def_left, float weight, int is_leaf);
void tahoe_decode_node(const tahoe_dense_node *n, float *value, float *weight, int *fid, int *def_left,
                       int *is_leaf);
/* tree_num_nodes Struct.h:15-17 */
int tahoe_tree_num_nodes(int depth);

/* ---- enums: algo_t Struct.h:23-27, strategy_t :29-34, output_t :37-42 ---- */
enum { TAHOE_ALGO_NAIVE = 0, TAHOE_ALGO_TREE_REORG = 1, TAHOE_ALGO_BATCH_TREE_REORG = 2 };
enum { TAHOE_FIL_SHARED_DATA = 0, TAHOE_FIL_SHARED_FOREST = 1, TAHOE_FIL_SPLIT_FOREST = 2,
       TAHOE_FIL_SPLIT_FOREST_SHARED_DATA = 3 };
enum { TAHOE_OUT_RAW = 0x0, TAHOE_OUT_AVG = 0x1, TAHOE_OUT_SIGMOID = 0x10, TAHOE_OUT_THRESHOLD = 0x100 };

/* forest_params_t, Struct.h:166-189 (same fields, same order). */
typedef struct {
    int num_nodes;     /* ignored for dense forests */
    int depth;         /* ps.depth: levels - 1; a tree has 2^(depth+1)-1 nodes */
    int num_trees;
    int num_cols;
    int algo;          /* algo_t; accepted and ignored (the layout is ours) */
    int output;        /* output_t bit set */
    float threshold;
    float global_bias;
    int strategy;      /* strategy_t of the FIL baseline; accepted and ignored */
    float missing;     /* "missing" sentinel: |x - missing| <= 1e-6 takes the default branch */
} tahoe_forest_params;

/* Traversal strategies of this library (the analogue of selected_algorithm 0..4, Struct.h:2168-2179). */
enum {
    TAHOE_STRATEGY_AUTO = 0,     /* selector picks from shape and LDS capacity */
    TAHOE_STRATEGY_DIRECT = 1,   /* lane = row, nodes and features straight from global memory
                                    (analogue of infer_adaptive_reorg_*, Struct.h:1196-1240) */
    TAHOE_STRATEGY_ROWTILE = 2,  /* 64-row feature-major tile in LDS, waves split the trees, top 8
                                    levels of each tree staged in LDS, deeper levels gathered from
                                    global memory, ordered leaf-sum exchange; any num_cols whose
                                    tile fits LDS */
    TAHOE_STRATEGY_TILEBLOCK = 3,/* num_cols <= 512: 128- (or 64-) row tile in LDS, four trees in
                                    flight, top 10 levels SoA in LDS, last two levels + leaves from
                                    one 32-byte block per walk; two barriers per round of 4 trees */
    TAHOE_STRATEGY_TILERING = 4, /* TILEBLOCK's data path with decoupled waves: walker waves with
                                    private tops, no barrier in the tree loop, one consumer wave adds
                                    leaf values in tree order through an LDS ring.  num_cols > 512, two
                                    forms picked at create: (a) tiles -- 32- / 16- / 8-row float32 tiles
                                    staged as they lie in memory, 2 / 4 / 8 trees per wave, 48-byte bottom
                                    blocks; (b) row streaming on 16-bit keys (num_cols <= 3072 and a multiple
                                    of 4, at most a tree per three features and 1024 trees, LDS for every
                                    level above the last two of ALL trees): one persistent workgroup per CU,
                                    rows turned into monotone 16-bit keys on their way into a ring of LDS
                                    slots, lane = tree, equal keys decided on the float32 values, leaf values
                                    through a workspace (tahoe_forest_reserve) and added in tree order by a
                                    summer wave.  AUTO's choice for wide rows when 2 x trees x depth <
                                    13 x num_cols (10 x where QRING walks three trees per lane): no quantise
                                    pass */
    TAHOE_STRATEGY_QRING = 5     /* TILERING on rank-quantised data: features and thresholds become
                                    exact 16-bit ranks (a per-predict quantise pass), 4-byte nodes.
                                    num_cols <= 256: tiles of three (or two) 64-row regions = 192 (128)
                                    rows, 14 walker waves x 3 chains (15 x 2), a batch walked as whole
                                    waves of 192-row tiles + a remainder of 128-row tiles; small batches
                                    give every tile to several workgroups, each a slice of the trees, and
                                    an ordered-sum kernel adds the leaf values in tree order (SPLIT).
                                    Forests whose features each see <= 254 thresholds (histogram-trained
                                    models) are quantised to 8-bit ranks for batches of whole tiles: 384-row
                                    tiles of three 128-row regions, six chains per lane; forests of <= 128
                                    features walk 384-row tiles on 16-bit ranks too (regions at a 16-KiB stride).
                                    Wider rows: 128-row tiles, or 64- / 32- / 16-row tiles with several
                                    trees per wave.  Forests with more than 32767 distinct thresholds on a
                                    feature are walked in groups of consecutive trees with chained float32
                                    sums (still the one sequential sum); unavailable only if a single tree
                                    exceeds that */
};
/* On a sparse handle (tahoe_sparse_forest_create): DIRECT = nodes and features from global memory,
 * ROWTILE = 64-row float32 tile in LDS, TILEBLOCK = tile + the first 512 nodes of each tree (breadth-first)
 * in LDS, walker waves + ordered ring consumer (trees of <= 65535 nodes), QRING = the same on rank-quantised
 * data: 192- / 128-row u16 tiles, three / two chains per lane, the first 9 levels of each tree as a complete
 * heap in LDS, 32-byte two-level blocks below (num_cols <= 256; tree groups as for dense forests).  AUTO
 * takes QRING when 20 x trees >= 13 x num_cols (enough walking per feature value to pay the quantise pass)
 * and the batch has >= 64 rows per CU, else TILEBLOCK, else what fits. */

typedef struct tahoe_forest tahoe_forest; /* opaque */

/* Builds the device layout from host nodes in the reference encoding: num_trees trees, each
 * 2^(depth+1)-1 nodes in heap order (children of i are 2i+1, 2i+2), tree-major (what
 * generate_forest_from_file produces, BaseTahoeTest.h:319-328).  The forest lives on the current
 * HIP device.  Validates that every reachable path ends in a leaf inside the tree and that every
 * reachable fid < num_cols (the reference reads out of bounds instead). */
tahoe_status tahoe_forest_create(tahoe_forest **out, const tahoe_dense_node *nodes,
                                 const tahoe_forest_params *params);
/* The same with options.  TAHOE_CREATE_PROB_RELAYOUT: the reference's probability-guided re-layout
 * (dense_adaptive_forest::init, Struct.h:1775-1825; swap_child :1712-1750): bottom-up, wherever dense_node_t.weight of
 * a node's left child is smaller than that of its right child, the two subtrees change places and the node is marked
 * "exchange" (the walk inverts its condition there, Struct.h:1060-1063), so that the likelier child is always the left
 * one and hot paths sit next to each other in memory.  Results are unchanged: leaf indices are still reported in the
 * original heap numbering.  Served by the strategies whose node words have a spare bit (DIRECT, ROWTILE, and QRING
 * when num_cols <= 256); without the flag `weight` is ignored, as in round 1.  tahoe_forest_create honours the
 * environment variable TAHOE_RELAYOUT=1 (read once, at create) for experiments. */
#define TAHOE_CREATE_PROB_RELAYOUT 0x1u
tahoe_status tahoe_forest_create_ex(tahoe_forest **out, const tahoe_dense_node *nodes,
                                    const tahoe_forest_params *params, unsigned flags);
void tahoe_forest_destroy(tahoe_forest *f);

/* ---- sparse (irregular) forests: sparse_node_t Struct.h:50-54, sparse_storage Struct.h:343-354,
 * init_sparse / sparse_forest::init (BaseTahoeTest.h:766-772, Struct.h:2329-2343) ---- */
typedef struct {
    float val;        /* threshold, or the output of a leaf */
    int32_t bits;     /* fid[0:29] | def_left<<30 | is_leaf<<31 (sparse_node_init, BaseTahoeTest.h:719-724) */
    int32_t left_idx; /* left child, relative to the tree's root; the right child is left_idx + 1 */
} tahoe_sparse_node;

/* trees[t] = offset of tree t's root in nodes[] (ascending); params->num_nodes = total nodes; params->depth is
 * ignored.  The handle is used with the same tahoe_forest_predict* / destroy entry points; leaf indices are
 * relative to the tree's root.  The walk is infer_one_tree_sparse (Struct.h:2217-2250) with the branch rule of
 * the live dense path (BaseTahoeTest.h:452), so a forest converted with tahoe_dense_to_sparse predicts exactly
 * what the dense forest predicts.  Rejects (TAHOE_ERR_INVALID_FOREST) child links that leave the tree or point
 * backwards, and fid >= num_cols. */
tahoe_status tahoe_sparse_forest_create(tahoe_forest **out, const int32_t *trees, const tahoe_sparse_node *nodes,
                                        const tahoe_forest_params *params);
/* dense2sparse (BaseTahoeTest.h:728-764).  *nodes_out / *trees_out: tahoe_free_host. */
tahoe_status tahoe_dense_to_sparse(const tahoe_dense_node *dense, int num_trees, int depth,
                                   tahoe_sparse_node **nodes_out, int32_t **trees_out, size_t *num_nodes_out);
/* Deterministic irregular forest (BASELINE config 5): per tree a depth limit in [min_depth, max_depth]; nodes
 * below min_depth become leaves with probability leaf_prob; at most max_tree_nodes per tree.  With nodes == NULL
 * only *num_nodes is computed (call twice). */
tahoe_status tahoe_synth_sparse_forest(tahoe_sparse_node *nodes, int32_t *trees, size_t *num_nodes, int num_trees,
                                       int num_cols, int min_depth, int max_depth, float leaf_prob,
                                       int max_tree_nodes, uint64_t seed);

/* preds_dev[rows] <- per-row float32 sum of leaf values in tree order 0..T-1 (the order of
 * predict_on_cpu, BaseTahoeTest.h:462-466), then AVG / bias / sigmoid / threshold as
 * forest::predict + transform_k do (Struct.h:196-209, :263-268).  data_dev is row-major
 * rows x num_cols float32 (data_d of generate_data_from_file, BaseTahoeTest.h:378-392). */
tahoe_status tahoe_forest_predict(tahoe_forest *f, float *preds_dev, const float *data_dev, size_t rows,
                                  void *stream);

/* Raw float32 per-row sums only (no output transform) -- the quantity tree shards exchange. */
tahoe_status tahoe_forest_predict_raw(tahoe_forest *f, float *sums_dev, const float *data_dev, size_t rows,
                                      void *stream);

/* Chained tree shards ("sum-order caveat"): sums_dev[rows] holds, on entry, the float32 sums of the trees
 * that come BEFORE this forest's trees in the whole ensemble; on return, those sums continued through this forest's
 * trees in tree order.  Shard k called on shard k-1's output therefore reproduces the single sequential float32 sum of
 * predict_on_cpu (BaseTahoeTest.h:462-466) bit for bit, which an all-reduce of per-shard totals cannot.  Every
 * strategy supports it (the kernels start their per-row accumulator from sums_dev[row] instead of 0.0f). */
tahoe_status tahoe_forest_predict_accumulate(tahoe_forest *f, float *sums_dev, const float *data_dev, size_t rows,
                                             void *stream);

/* leaf_dev[row * num_trees + tree] <- index of the leaf the row ends in, in the tree's original heap
 * numbering (final `curr` of infer_one_tree, BaseTahoeTest.h:441-455).  sums_dev may be NULL. */
tahoe_status tahoe_forest_predict_leaf_idx(tahoe_forest *f, uint32_t *leaf_dev, float *sums_dev,
                                           const float *data_dev, size_t rows, void *stream);

/* Finishes sums -> preds in place (AVG, bias, sigmoid, threshold); what the ranks of a tree-sharded
 * forest call after the all-reduce.  num_trees_total = trees of the whole forest. */
tahoe_status tahoe_transform_preds(float *preds_dev, size_t rows, int output, int num_trees_total,
                                   float threshold, float global_bias, void *stream);

tahoe_status tahoe_forest_set_strategy(tahoe_forest *f, int strategy);
/* Waits for `stream` and reports TAHOE_ERR_HIP if a kernel flagged an internal error (a bounded
 * LDS wait of TILERING timing out); TAHOE_OK otherwise. */
tahoe_status tahoe_forest_check(tahoe_forest *f, void *stream);
/* Sizes the handle's device workspace for batches of up to `rows` rows: QRING's 2-byte-per-value quantised copy of
 * the batch and, for the batches small enough to be walked in tree slices (up to 64 rows per CU), one float per
 * (tree of the largest group, row); the row-streaming form of TILERING's leaf-value workspace, 4 bytes x rows x
 * (trees rounded up to 32), at most 1 GiB (TAHOE_WSTREAM_SLAB_MB at create).  Optional: predict grows the
 * workspace on demand, which is the only case in which a predict call allocates (and synchronises the device,
 * so it cannot be captured); after reserve(rows) no batch of up to `rows` rows does. */
tahoe_status tahoe_forest_reserve(tahoe_forest *f, size_t rows);

/* Host-resident batch (the reference uploads the data file once, BaseTahoeTest.h:378-389, and
 * has no per-batch host path).  Rows are uploaded in chunks of `chunk_rows` rows (0 = about 32 MiB of rows)
 * through two device buffers; the upload of chunk i+1 overlaps the traversal of chunk i and the download of
 * chunk i-1's predictions.  A pinned source (hipHostMalloc / hipHostRegister) is copied from directly,
 * pageable memory is staged through pinned buffers by a few host threads.  Synchronous: preds_host is
 * complete on return.  Results are identical to tahoe_forest_predict on the whole batch (rows are
 * independent).  The buffers and three streams are created on first use and kept by the handle. */
tahoe_status tahoe_forest_predict_host(tahoe_forest *f, float *preds_host, const float *data_host, size_t rows,
                                       size_t chunk_rows);
/* Pinned host memory for the call above (hipHostMalloc / hipHostFree). */
tahoe_status tahoe_host_alloc(void **ptr, size_t bytes);
tahoe_status tahoe_host_free(void *ptr);
/* Strategy the next predict will run (after AUTO resolution for `rows`). */
int tahoe_forest_get_strategy(const tahoe_forest *f, size_t rows);

typedef struct {
    int num_trees, depth, num_cols;
    int bits_bytes;          /* b of the reference's adaptive format rule (Struct.h:1827-1852): 1, 2 or 4 */
    int lds_levels;          /* top levels staged in LDS by ROWTILE */
    size_t device_bytes;     /* device memory owned by the handle */
    int lds_bytes_per_block; /* dynamic LDS of the ROWTILE kernel (0 = does not fit) */
    int device_id;
    int num_cus;
    int top_levels;          /* top levels staged in LDS by TILEBLOCK */
    int tile_rows;           /* rows per TILEBLOCK tile: 128, 64, or 0 = strategy unavailable */
    int tileblock_lds_bytes; /* dynamic LDS of the TILEBLOCK kernel */
    int qring_walkers;       /* walker waves of the QRING kernel; 0 = strategy unavailable */
    int qring_lds_bytes;     /* dynamic LDS of the QRING kernel */
    int qring_groups;        /* tree groups quantised separately (forests with > 32767 thresholds per feature) */
    int is_sparse;           /* 1: handle made by tahoe_sparse_forest_create (only the generic fields are set) */
    int ring_rows;           /* rows per TILERING tile: 64 or 128; 32 / 16 / 8 for the wide-row float32 form (num_cols > 512);
                              * 0 = strategy unavailable */
    int tilering_lds_bytes;  /* dynamic LDS of the TILERING kernel the launch takes (of the wide-row form where that runs) */
    int qring_tile_rows;     /* rows per quantised tile in LDS: 384 (8-bit ranks, or <= 128 features), 192 (region form), 128, or
                              * 64 / 32 / 16 for wide rows (several trees per wave); 0 = features read from the quantised tile in
                              * L2, or QRING unavailable.  The large tile of the form: a batch may end in 128-row tiles. */
    int relayout;            /* 1: created with TAHOE_CREATE_PROB_RELAYOUT */
    size_t relayout_swaps;   /* internal nodes whose subtrees changed places */
    int stream_slots;        /* > 0: TILERING runs as the row-streaming kernel on 16-bit keys (picked at create by the shape
                              * rule -- num_cols <= 3072 and a multiple of 4, every level above the last two of all trees
                              * resident, at most a tree per three features, key map fine enough -- or forced with
                              * TAHOE_WSTREAM=1): LDS row slots of the ring the rows stream through */
    int stream_levels;       /* ... and the top levels of ALL trees it keeps resident in LDS */
    float stream_key_ties;   /* create-time estimate of the share of that form's 16-bit key compares that tie and fall back
                              * to the float32 values (one affine key map for all features: small-scale features beside
                              * large-scale ones tie often); the shape rule takes the form only below 1e-4 */
} tahoe_forest_info;
tahoe_status tahoe_forest_get_info(const tahoe_forest *f, tahoe_forest_info *info);

/* Which kernel a predict of `rows` rows launches (one strategy number can stand for several kernels: TILERING is
 * tilering_kernel, widef_kernel or wkey_kernel; QRING has several tile forms).  -1 for a NULL handle.
 * REGION8 / REGION6 name the code layout: whole waves of the batch go through 384-row tiles, a remainder (or a batch too small
 * for one wave of them) through 128-row tiles of the same codes in a second launch, as REGION_MIXED does for 192 + 128. */
enum {
    TAHOE_FORM_NONE = 0,                  /* strategy unavailable for this shape */
    TAHOE_FORM_DIRECT = 1,                /* direct_kernel */
    TAHOE_FORM_ROWTILE = 2,               /* rowtile_kernel */
    TAHOE_FORM_TILEBLOCK = 3,             /* tileblock_kernel */
    TAHOE_FORM_TILERING_TILE = 4,         /* tilering_kernel: 64- / 128-row float32 tile, num_cols <= 512 */
    TAHOE_FORM_TILERING_WIDE_TILE = 5,    /* widef_kernel: 32- / 16- / 8-row float32 tiles of wide rows */
    TAHOE_FORM_TILERING_WIDE_STREAM = 6,  /* wkey_kernel: rows streamed through LDS as 16-bit keys, lane = tree */
    TAHOE_FORM_QRING_REGION3 = 7,         /* qring_kernel, 192-row tiles of three 64-row regions (K3) */
    TAHOE_FORM_QRING_REGION2 = 8,         /* qring_kernel, 128-row tiles of two regions */
    TAHOE_FORM_QRING_REGION_MIXED = 9,    /* whole waves of 192-row tiles + a remainder of 128-row tiles (two launches) */
    TAHOE_FORM_QRING_SPLIT = 10,          /* small batches: tree slices per tile + ordered_sum_kernel */
    TAHOE_FORM_QRING_COLUMNS = 11,        /* qring_kernel on 128-slot columns (general node word, or the exchange-bit layout) */
    TAHOE_FORM_QRING_WIDE = 12,           /* qwide_kernel: 64- / 32- / 16-row tiles, several trees per wave */
    TAHOE_FORM_QRING_GX = 13,             /* qring_kernel reading codes from L2 (rows too wide for any LDS tile) */
    TAHOE_FORM_SPARSE_DIRECT = 14,        /* sparse handle: sparse_kernel without a tile */
    TAHOE_FORM_SPARSE_ROWTILE = 15,       /* sparse_kernel with the 64-row float32 tile */
    TAHOE_FORM_SPARSE_TOP = 16,           /* sparse_top_kernel */
    TAHOE_FORM_SPARSE_QRING = 17,         /* sparse_q_kernel */
    TAHOE_FORM_QRING_REGION8 = 18,        /* qring_kernel on 8-bit rank codes (<= 254 thresholds per feature): 384-row tiles of three
                                             128-row regions, six chains per lane (<= 128 features: 15 walkers, ring of 24) */
    TAHOE_FORM_QRING_REGION6 = 19         /* qring_kernel on u16 codes, num_cols <= 128: 384-row tiles of six 64-row regions at a
                                             16-KiB stride, six chains per lane */
};
int tahoe_forest_get_kernel_form(const tahoe_forest *f, size_t rows);
const char *tahoe_kernel_form_name(int form);

/* Kernel timing with hipEvents on the stream the kernel runs on.  set_profiling(f, n) arms up to n
 * launches (0 disarms): each following predict brackets its traversal kernel with an event pair (a pre-pass kernel, if
 * the strategy has one, is timed apart: tahoe_forest_prepass_times).
 * kernel_times waits for the recorded launches and returns their durations in milliseconds. */
tahoe_status tahoe_forest_set_profiling(tahoe_forest *f, int max_launches);
tahoe_status tahoe_forest_kernel_times(tahoe_forest *f, float *ms_out, int capacity, int *count);
/* Durations of the pre-pass kernel of the same launches (QRING's quantise kernel; 0 for the others). */
tahoe_status tahoe_forest_prepass_times(tahoe_forest *f, float *ms_out, int capacity, int *count);

/* ---- file formats (BaseTahoeTest.h:267-402): one value per line ---- */
/* model: num_trees, levels(=depth+1), then per tree, per node in heap order: fid, value,
 * default_left, weight, is_leaf.  *num_trees / *depth are in/out (kept when a header line is
 * missing, as the reference keeps its constructor defaults).  nodes_out: tahoe_free_host. */
tahoe_status tahoe_load_model(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out);
/* data: num_rows, num_cols, missing, then rows*cols values row-major. */
tahoe_status tahoe_load_data(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out);
tahoe_status tahoe_write_model(const char *path, int num_trees, int depth, const tahoe_dense_node *nodes);
tahoe_status tahoe_write_data(const char *path, int num_rows, int num_cols, float missing, const float *data);

/* ---- binary files (no counterpart in the reference, whose text formats cost ~2.5 s for the K3
 * model and ~25 s for the K3 data on one core).  64-byte header + the payload as it sits in memory
 * (dense_node_t AoS / row-major float32) + a checksum; TAHOE_ERR_IO on a foreign, truncated or corrupt file. */
tahoe_status tahoe_save_model_bin(const char *path, int num_trees, int depth, const tahoe_dense_node *nodes);
tahoe_status tahoe_load_model_bin(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out);
tahoe_status tahoe_save_data_bin(const char *path, int num_rows, int num_cols, float missing, const float *data);
tahoe_status tahoe_load_data_bin(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out);
/* Text file with a cache beside it ("<path>.tbin"): used when it records the text file's current size and
 * mtime, else the text is parsed as by tahoe_load_model / tahoe_load_data and the cache rewritten (best
 * effort).  *from_cache (may be NULL) tells which happened.  The BaseTahoeTest mirror uses these when the
 * environment variable TAHOE_BIN_CACHE is set. */
tahoe_status tahoe_load_model_cached(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out,
                                     int *from_cache);
tahoe_status tahoe_load_data_cached(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out,
                                    int *from_cache);
void tahoe_free_host(void *p);

/* ---- deterministic synthetic inputs: SplitMix64, counter-based ---- */
/* Complete trees: levels < depth internal (fid = x mod num_cols, threshold 2u-1, def_left = x&1),
 * bottom level leaves (val 2u-1); internal nodes above the bottom turn into leaves with
 * probability leaf_prob. */
void tahoe_synth_forest(tahoe_dense_node *nodes, int num_trees, int depth, int num_cols, uint64_t seed,
                        float leaf_prob);
/* rows x cols float32 in [-1,1); each value replaced by `missing` with probability missing_prob and
 * by NaN with probability nan_prob.  first_row lets ranks generate disjoint row ranges. */
void tahoe_synth_data(float *out, size_t first_row, size_t rows, int num_cols, uint64_t seed,
                      float missing_prob, float missing, float nan_prob);

/* A second generator, in the style of histogram-trained gradient-boosted models (XGBoost / LightGBM with max_bin <= 255:
 * the model families the reference is run on, run_all_15_examples.sh:51-65): at most max_bins distinct thresholds per
 * feature (bin edges = quantiles of a 4096-value sample of the feature), Zipf-skewed feature usage (exponent zipf_s), every
 * node splits what its ancestors left of the feature's range at a skewed position (unbalanced branches; `weight` = reach
 * probability), early leaves where hardly any row arrives (+ leaf_prob), features on different scales (scale_decades decades
 * of spread) and of four shapes (uniform, exponential, bell, small integer counts with half-integer thresholds).
 * tahoe_synth_data_hist draws rows from the same per-feature distributions (same feature_seed and scale_decades). */
tahoe_status tahoe_synth_forest_hist(tahoe_dense_node *nodes, int num_trees, int depth, int num_cols, uint64_t seed,
                                     uint64_t feature_seed, int max_bins, float zipf_s, float leaf_prob, float scale_decades);
tahoe_status tahoe_synth_data_hist(float *out, size_t first_row, size_t rows, int num_cols, uint64_t seed, uint64_t feature_seed,
                                   float scale_decades, float missing_prob, float missing);

/* ---- thin device helpers so that host code above this ABI needs no HIP headers ---- */
tahoe_status tahoe_device_count(int *count);
tahoe_status tahoe_device_set(int device);
tahoe_status tahoe_device_alloc(void **ptr, size_t bytes, int set_zero);   /* allocate(), cuda_base.h:28-32 */
tahoe_status tahoe_device_free(void *ptr);
tahoe_status tahoe_device_memset(void *ptr_dev, int value, size_t bytes, void *stream);
/* float32 <-> float64 on the device: tree shards that exchange their per-row partial sums by an all-reduce do it
 * on float64 copies (8 bytes per row), so that combining the partials adds no float32 rounding of its own. */
tahoe_status tahoe_widen_f32_to_f64(double *dst_dev, const float *src_dev, size_t n, void *stream);
tahoe_status tahoe_narrow_f64_to_f32(float *dst_dev, const double *src_dev, size_t n, void *stream);
tahoe_status tahoe_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, void *stream);
tahoe_status tahoe_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, void *stream);
tahoe_status tahoe_stream_create(void **stream);
tahoe_status tahoe_stream_destroy(void *stream);
tahoe_status tahoe_stream_synchronize(void *stream);
/* Events and device-to-device copies between GPUs of one process: what a host that chains tree shards over several
 * devices needs (device g waits for device g-1's chunk, copies its running sums over xGMI, continues them). */
tahoe_status tahoe_event_create(void **event);
tahoe_status tahoe_event_destroy(void *event);
tahoe_status tahoe_event_record(void *event, void *stream);
tahoe_status tahoe_stream_wait_event(void *stream, void *event);
tahoe_status tahoe_copy_peer(void *dst_dev, int dst_device, const void *src_dev, int src_device, size_t bytes, void *stream);
tahoe_status tahoe_device_synchronize(void);
tahoe_status tahoe_device_lds_bytes(int *bytes);  /* sharedMemPerBlock analogue, Struct.h:215-220 */
/* compare_GPU, cuda_base.h:98-111: counts i with |a[i]-b[i]| > tol (on the device). */
tahoe_status tahoe_compare_device(const float *a_dev, const float *b_dev, size_t n, float tol,
                                  size_t *num_bad, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TAHOE_AMD_H */
