/*
 * tahoe_oracle.c -- CPU restatement of the reference's traversal path (see tahoe_oracle.h).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (header comment of tahoe_oracle.h explains why).
 * All file:line citations are into the reference repository (sampathrg/Tahoe).
 */
#include "tahoe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Struct.h:57-59 */
#define FID_MASK ((int32_t)((1u << 30) - 1u))
#define DEF_LEFT_MASK ((int32_t)(1u << 30))
#define IS_LEAF_MASK ((int32_t)(1u << 31))

/* BaseTahoeTest.h:269,356 */
#define MAX_LINE 1024

int oracle_tree_num_nodes(int depth) { return (1 << (depth + 1)) - 1; } /* Struct.h:15-17 */

void oracle_encode_node(oracle_node *n, int fid, float value, int def_left, float weight, int is_leaf)
{
    /* Struct.h:103-108 */
    n->weight = weight;
    n->val = value;
    n->bits = (fid & FID_MASK) | (def_left ? DEF_LEFT_MASK : 0) | (is_leaf ? IS_LEAF_MASK : 0);
}

void oracle_decode_node(const oracle_node *n, float *value, float *weight, int *fid, int *def_left,
                        int *is_leaf)
{
    /* Struct.h:110-117 (the bool conversions become != 0) */
    *value = n->val;
    *weight = n->weight;
    *fid = n->bits & FID_MASK;
    *def_left = (n->bits & DEF_LEFT_MASK) != 0;
    *is_leaf = (n->bits & IS_LEAF_MASK) != 0;
}

int oracle_load_model(const char *path, int *num_trees, int *depth, oracle_node **nodes_out)
{
    /* BaseTahoeTest.h:267-352.  One value per line; the value lines are read with unchecked
     * fgets(), so when the file ends early `buf` keeps the last line read (:298-307). */
    char buf[MAX_LINE];
    FILE *fp = fopen(path, "r");
    if (fp == NULL) return -1; /* :273-277 perror + exit(1) in the reference */
    buf[0] = '\0';
    if (fgets(buf, MAX_LINE, fp)) *num_trees = atoi(buf);  /* :279-280 */
    if (fgets(buf, MAX_LINE, fp)) *depth = atoi(buf) - 1;  /* :281-282 */

    size_t per_tree = (size_t)oracle_tree_num_nodes(*depth);
    size_t num_nodes = (size_t)(*num_trees) * per_tree;    /* :286 */
    oracle_node *nodes = (oracle_node *)malloc((num_nodes ? num_nodes : 1) * sizeof(oracle_node));
    if (!nodes) {
        fclose(fp);
        return -2;
    }
    for (size_t i = 0; i < num_nodes; ++i) {               /* :296-315, same order tree-major */
        char *r;
        r = fgets(buf, MAX_LINE, fp); (void)r;
        int fid = atoi(buf);                               /* :299 */
        r = fgets(buf, MAX_LINE, fp); (void)r;
        float value = (float)atof(buf);                    /* :301 double -> float */
        r = fgets(buf, MAX_LINE, fp); (void)r;
        int def_left = atoi(buf) != 0;                     /* :303 int -> bool */
        r = fgets(buf, MAX_LINE, fp); (void)r;
        float weight = (float)atof(buf);                   /* :305 */
        r = fgets(buf, MAX_LINE, fp); (void)r;
        int is_leaf = atoi(buf) != 0;                      /* :307 */
        oracle_encode_node(&nodes[i], fid, value, def_left, weight, is_leaf); /* :321-328 */
    }
    fclose(fp);
    *nodes_out = nodes;
    return 0;
}

int oracle_load_data(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out)
{
    /* BaseTahoeTest.h:354-402 */
    char buf[MAX_LINE];
    FILE *fp = fopen(path, "r");
    if (fp == NULL) return -1; /* :360-364 */
    buf[0] = '\0';
    if (fgets(buf, MAX_LINE, fp)) *num_rows = atoi(buf);       /* :366-367 */
    if (fgets(buf, MAX_LINE, fp)) *num_cols = atoi(buf);       /* :368-369 */
    if (fgets(buf, MAX_LINE, fp)) *missing = (float)atof(buf); /* :370-371 */

    size_t num_data = (size_t)(*num_rows) * (size_t)(*num_cols); /* :377 (size_t here, int there) */
    float *data = (float *)malloc((num_data ? num_data : 1) * sizeof(float));
    if (!data) {
        fclose(fp);
        return -2;
    }
    for (size_t i = 0; i < num_data; ++i) {                    /* :382-388 row-major */
        char *r = fgets(buf, MAX_LINE, fp); (void)r;
        data[i] = (float)atof(buf);
    }
    fclose(fp);
    *data_out = data;
    return 0;
}

void oracle_free(void *p) { free(p); }

float oracle_infer_one_tree(const oracle_node *root, const float *row, float missing, uint32_t *leaf_idx)
{
    /* BaseTahoeTest.h:440-456 */
    int curr = 0;
    float value = 0.0f, weight = 0.0f;
    int fid = 0, def_left = 0, is_leaf = 0;
    for (;;) {
        oracle_decode_node(&root[curr], &value, &weight, &fid, &def_left, &is_leaf); /* :447 */
        if (is_leaf) break;                                                          /* :449 */
        float val = row[fid];                                                        /* :450 */
        const float eps = 1.0e-6f;                                                   /* :451 */
        /* :452  fabs() of a float difference compared with a float eps; the float32 subtraction
         * happens first in either overload, so the test is on the float32 difference. */
        int cond = (fabsf(val - missing) <= eps) ? !def_left : (val >= value);
        curr = (curr << 1) + 1 + (cond ? 1 : 0);                                     /* :453 */
    }
    if (leaf_idx) *leaf_idx = (uint32_t)curr;
    return value;                                                                    /* :455 */
}

static float oracle_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); } /* Struct.h:13 */

void oracle_predict(const oracle_node *nodes, int num_trees, int depth, const float *data,
                    size_t row_begin, size_t row_end, int num_cols, float missing, int output,
                    float threshold, float global_bias, float *preds, uint32_t *leaf_idx)
{
    /* BaseTahoeTest.h:458-474 */
    size_t num_nodes = (size_t)oracle_tree_num_nodes(depth); /* :461 */
    for (size_t i = row_begin; i < row_end; ++i) {           /* :462 */
        float pred = 0.0f;                                   /* :463 */
        for (int j = 0; j < num_trees; ++j) {                /* :464 trees in file order, float32 += */
            uint32_t li;
            pred += oracle_infer_one_tree(&nodes[(size_t)j * num_nodes], &data[i * (size_t)num_cols],
                                          missing, &li);     /* :465 */
            if (leaf_idx) leaf_idx[i * (size_t)num_trees + (size_t)j] = li;
        }
        if ((output & ORACLE_OUT_AVG) != 0) pred = pred / num_trees;       /* :467 */
        pred += global_bias;                                               /* :468 */
        if ((output & ORACLE_OUT_SIGMOID) != 0) pred = oracle_sigmoid(pred); /* :469 */
        if ((output & ORACLE_OUT_THRESHOLD) != 0) pred = pred > threshold ? 1.0f : 0.0f; /* :470-472 */
        if (preds) preds[i] = pred;                                        /* :473 */
    }
}

void oracle_predict_f64(const oracle_node *nodes, int num_trees, int depth, const float *data,
                        size_t row_begin, size_t row_end, int num_cols, float missing, double *sums)
{
    size_t num_nodes = (size_t)oracle_tree_num_nodes(depth);
    for (size_t i = row_begin; i < row_end; ++i) {
        double s = 0.0;
        for (int j = 0; j < num_trees; ++j)
            s += (double)oracle_infer_one_tree(&nodes[(size_t)j * num_nodes],
                                               &data[i * (size_t)num_cols], missing, NULL);
        sums[i] = s;
    }
}

/* Checkers of the multi-GPU tree shards (no counterpart in the single-GPU reference).
 * oracle_predict_continue: the loop of BaseTahoeTest.h:462-466 started from sums[i] instead of 0.0f, i.e. what shard k
 * of a chained predict must produce from shard k-1's output; with sums = 0 it is oracle_predict with output = RAW.
 * oracle_abs_leaf_sum: sum over trees of |leaf value| in float64, the A of the error bound gamma(n) * A. */
void oracle_predict_continue(const oracle_node *nodes, int num_trees, int depth, const float *data,
                             size_t row_begin, size_t row_end, int num_cols, float missing, float *sums)
{
    size_t num_nodes = (size_t)oracle_tree_num_nodes(depth);
    for (size_t i = row_begin; i < row_end; ++i) {
        float pred = sums[i];
        for (int j = 0; j < num_trees; ++j)
            pred += oracle_infer_one_tree(&nodes[(size_t)j * num_nodes], &data[i * (size_t)num_cols], missing, NULL);
        sums[i] = pred;
    }
}

void oracle_abs_leaf_sum(const oracle_node *nodes, int num_trees, int depth, const float *data,
                         size_t row_begin, size_t row_end, int num_cols, float missing, double *sums)
{
    size_t num_nodes = (size_t)oracle_tree_num_nodes(depth);
    for (size_t i = row_begin; i < row_end; ++i) {
        double s = 0.0;
        for (int j = 0; j < num_trees; ++j)
            s += fabs((double)oracle_infer_one_tree(&nodes[(size_t)j * num_nodes], &data[i * (size_t)num_cols], missing, NULL));
        sums[i] = s;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Sparse forests (dead code in the reference; the only reference-defined format that can hold the
 * irregular config K5).  See tahoe_oracle.h for the choice of branch rule. */
float oracle_sparse_infer_one_tree(const oracle_sparse_node *root, const float *row, float missing, uint32_t *leaf_idx)
{
    /* Struct.h:2217-2250 with NITEMS = 1 */
    unsigned int curr = 0;                                  /* :2221 */
    for (;;) {
        const float n_val = root[curr].val;                 /* :2226 */
        const int32_t n_bits = root[curr].bits;             /* :2227 */
        const int n_fid = n_bits & FID_MASK;                /* :2230 */
        const int n_def_left = (n_bits & DEF_LEFT_MASK) != 0; /* :2231 */
        const int n_is_leaf = (n_bits & IS_LEAF_MASK) != 0; /* :2232 */
        if (n_is_leaf) break;                               /* :2234-2237 */
        const float val = row[n_fid];                       /* :2239 */
        const float eps = 1.0e-6f;
        const int cond = (fabsf(val - missing) <= eps) ? !n_def_left : (val >= n_val); /* live rule, BaseTahoeTest.h:452 */
        curr = (unsigned int)(root[curr].left_idx + cond);  /* :2244 */
    }
    if (leaf_idx) *leaf_idx = curr;
    return root[curr].val;                                  /* :2249 */
}

void oracle_sparse_predict(const oracle_sparse_node *nodes, const int32_t *trees, int num_trees, const float *data,
                           size_t row_begin, size_t row_end, int num_cols, float missing, float *preds,
                           uint32_t *leaf_idx)
{
    for (size_t i = row_begin; i < row_end; ++i) {
        float pred = 0.0f;
        for (int j = 0; j < num_trees; ++j) { /* trees in order, float32 += (infer_k, Struct.h:2271-2273, made sequential) */
            uint32_t li;
            pred += oracle_sparse_infer_one_tree(&nodes[trees[j]], &data[i * (size_t)num_cols], missing, &li);
            if (leaf_idx) leaf_idx[i * (size_t)num_trees + (size_t)j] = li;
        }
        if (preds) preds[i] = pred;
    }
}

typedef struct {
    oracle_sparse_node *nodes;
    size_t size, cap;
} sparse_vec;

static size_t sv_push(sparse_vec *v)
{
    if (v->size == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->nodes = (oracle_sparse_node *)realloc(v->nodes, v->cap * sizeof(oracle_sparse_node));
    }
    memset(&v->nodes[v->size], 0, sizeof(oracle_sparse_node));
    return v->size++;
}

static void d2s_node(const oracle_node *dense_root, size_t i_dense, size_t i_sparse_root, size_t i_sparse, sparse_vec *v)
{
    /* dense2sparse_node, BaseTahoeTest.h:728-752 */
    float value, weight;
    int fid, def_left, is_leaf;
    oracle_decode_node(&dense_root[i_dense], &value, &weight, &fid, &def_left, &is_leaf);
    if (is_leaf) {                                            /* :735-740 */
        v->nodes[i_sparse].val = value;
        v->nodes[i_sparse].bits = (fid & FID_MASK) | (def_left ? DEF_LEFT_MASK : 0) | IS_LEAF_MASK;
        v->nodes[i_sparse].left_idx = 0;
        return;
    }
    const size_t left_index = sv_push(v);                     /* :743-745 reserve both children */
    (void)sv_push(v);
    v->nodes[i_sparse].val = value;
    v->nodes[i_sparse].bits = (fid & FID_MASK) | (def_left ? DEF_LEFT_MASK : 0);
    v->nodes[i_sparse].left_idx = (int32_t)(left_index - i_sparse_root); /* :746-747 */
    d2s_node(dense_root, 2 * i_dense + 1, i_sparse_root, left_index, v);      /* :748 */
    d2s_node(dense_root, 2 * i_dense + 2, i_sparse_root, left_index + 1, v);  /* :749-750 */
}

size_t oracle_dense_to_sparse(const oracle_node *dense, int num_trees, int depth, oracle_sparse_node **nodes_out,
                              int32_t **trees_out)
{
    /* dense2sparse / dense2sparse_tree, BaseTahoeTest.h:754-764 */
    sparse_vec v = {NULL, 0, 0};
    int32_t *trees = (int32_t *)malloc((num_trees ? num_trees : 1) * sizeof(int32_t));
    const size_t per_tree = (size_t)oracle_tree_num_nodes(depth);
    for (int t = 0; t < num_trees; ++t) {
        const size_t root = sv_push(&v);
        d2s_node(&dense[(size_t)t * per_tree], 0, root, root, &v);
        trees[t] = (int32_t)root;
    }
    *nodes_out = v.nodes;
    *trees_out = trees;
    return v.size;
}
