/*
 * tahoe_oracle.h -- CPU restatement of the reference's tree-ensemble traversal path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded by or called from the
 * product library (tahoe_amd/csrc -> libtahoe_amd.so).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (sampathrg/Tahoe) holds no golden vectors, known-answer tests or
 * fixtures for this path, and its sources cannot be compiled in this image
 * without writing stand-in headers (main.cu needs nvcc, <cuda_runtime.h>, CUDA thrust and cuRAND
 * headers -- see DESIGN.md "Oracle").  Every function below therefore restates the reference
 * source line by line and cites it; hand-derived known-answer cases live in tests/.
 */
#ifndef TAHOE_ORACLE_H
#define TAHOE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dense_node_t, Struct.h:44-48 */
typedef struct {
    float weight;
    float val;
    int32_t bits;
} oracle_node;

/* output_t, Struct.h:37-42 */
#define ORACLE_OUT_RAW 0x0
#define ORACLE_OUT_AVG 0x1
#define ORACLE_OUT_SIGMOID 0x10
#define ORACLE_OUT_THRESHOLD 0x100

/* encode_node, Struct.h:103-108 */
void oracle_encode_node(oracle_node *n, int fid, float value, int def_left, float weight, int is_leaf);
/* dense_node_decode, Struct.h:110-117 */
void oracle_decode_node(const oracle_node *n, float *value, float *weight, int *fid, int *def_left,
                        int *is_leaf);

/* tree_num_nodes, Struct.h:15-17 */
int oracle_tree_num_nodes(int depth);

/* generate_forest_from_file, BaseTahoeTest.h:267-352.  *num_trees / *depth are in/out: like the
 * reference they keep the caller's value when a header line is missing.  Returns 0, or -1 when the
 * file cannot be opened (the reference perror()s and exit(1)s there).  *nodes_out is malloc'd. */
int oracle_load_model(const char *path, int *num_trees, int *depth, oracle_node **nodes_out);

/* generate_data_from_file, BaseTahoeTest.h:354-402 (host half).  Same conventions. */
int oracle_load_data(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out);

void oracle_free(void *p);

/* infer_one_tree, BaseTahoeTest.h:440-456.  Returns the leaf value; *leaf_idx (optional) gets the
 * final `curr`, i.e. the leaf's index in the tree's heap numbering. */
float oracle_infer_one_tree(const oracle_node *root, const float *row, float missing, uint32_t *leaf_idx);

/* predict_on_cpu, BaseTahoeTest.h:458-474, for rows [row_begin, row_end).
 * preds (optional) is indexed by absolute row; leaf_idx (optional) is [row][tree], absolute row. */
void oracle_predict(const oracle_node *nodes, int num_trees, int depth, const float *data,
                    size_t row_begin, size_t row_end, int num_cols, float missing, int output,
                    float threshold, float global_bias, float *preds, uint32_t *leaf_idx);

/* Same walk, but the per-row sum is kept in double (not in the reference; used only to report how
 * far the float32 sequential sum is from the exact one). */
void oracle_predict_f64(const oracle_node *nodes, int num_trees, int depth, const float *data,
                        size_t row_begin, size_t row_end, int num_cols, float missing, double *sums);
/* Checkers of the multi-GPU tree shards (see tahoe_oracle.c). */
void oracle_predict_continue(const oracle_node *nodes, int num_trees, int depth, const float *data,
                             size_t row_begin, size_t row_end, int num_cols, float missing, float *sums);
void oracle_abs_leaf_sum(const oracle_node *nodes, int num_trees, int depth, const float *data,
                         size_t row_begin, size_t row_end, int num_cols, float missing, double *sums);

/* ---- sparse forests: sparse_node_t (Struct.h:50-54), sparse_tree / sparse_storage (Struct.h:334-354) ---- */
typedef struct {
    float val;        /* threshold, or the output of a leaf */
    int32_t bits;     /* fid | def_left<<30 | is_leaf<<31 (sparse_node_init, BaseTahoeTest.h:719-724) */
    int32_t left_idx; /* index of the left child relative to the tree's root; right child = left_idx + 1 */
} oracle_sparse_node;

/* The walk of infer_one_tree_sparse (Struct.h:2217-2250): curr = 0; until a leaf: curr = left_idx + cond.
 * The branch rule is the one of the LIVE path (BaseTahoeTest.h:452: |x - missing| <= 1e-6 ? !def_left : x >= thr),
 * not the isnan() of the dead sparse kernel (Struct.h:2240): the reference's own (commented) sparse harness checks
 * the sparse forest against predict_on_cpu of the dense forest it was converted from (BaseTahoeTest.h:746-764, 836),
 * which only holds with the live rule.  *leaf_idx = final curr (relative to the tree's root). */
float oracle_sparse_infer_one_tree(const oracle_sparse_node *root, const float *row, float missing, uint32_t *leaf_idx);

/* Per-row float32 sum in tree order over trees[t] = root offset of tree t (sparse_storage::operator[], Struct.h:351-353),
 * rows [row_begin, row_end); preds / leaf_idx as in oracle_predict (RAW output). */
void oracle_sparse_predict(const oracle_sparse_node *nodes, const int32_t *trees, int num_trees, const float *data,
                           size_t row_begin, size_t row_end, int num_cols, float missing, float *preds,
                           uint32_t *leaf_idx);

/* dense2sparse (BaseTahoeTest.h:728-764): converts one dense forest; *nodes_out / *trees_out are malloc'd;
 * returns the number of sparse nodes. */
size_t oracle_dense_to_sparse(const oracle_node *dense, int num_trees, int depth, oracle_sparse_node **nodes_out,
                              int32_t **trees_out);

#ifdef __cplusplus
}
#endif
#endif
