// Host-side pieces of the Tahoe surface that need no device: node encoding, the two text file
// formats, deterministic synthetic inputs, error text.
//
// Reference behaviour restated here (file:line into sampathrg/Tahoe):
//   encode_node / dense_node_decode            Struct.h:103-117
//   generate_forest_from_file                  BaseTahoeTest.h:267-352
//   generate_data_from_file (host half)        BaseTahoeTest.h:354-402
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "common.h"

namespace tahoe {

static thread_local std::string g_last_error;

tahoe_status fail(tahoe_status code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
void clear_error() { g_last_error.clear(); }

namespace {

constexpr int32_t kFidMask = (int32_t)((1u << 30) - 1u);  // Struct.h:57
constexpr int32_t kDefLeftMask = (int32_t)(1u << 30);     // Struct.h:58
constexpr int32_t kIsLeafMask = (int32_t)(1u << 31);      // Struct.h:59

// Hands out the file one "fgets(buf, 1024, fp)" unit at a time: at most 1023 characters, ending
// after a newline.  At end of file the previous unit stays current, which is what the reference's
// unchecked fgets calls observe (BaseTahoeTest.h:298-307, :384).
class LineFeed {
   public:
    explicit LineFeed(FILE *fp) : fp_(fp), chunk_(1 << 22) { unit_[0] = '\0'; }
    // Advances to the next unit; returns false (unit unchanged) at end of file.
    bool next()
    {
        size_t n = 0;
        while (n < kMax - 1) {
            if (pos_ == len_) {
                len_ = fread(chunk_.data(), 1, chunk_.size(), fp_);
                pos_ = 0;
                if (len_ == 0) break;
            }
            char c = chunk_[pos_++];
            scratch_[n++] = c;
            if (c == '\n') break;
        }
        if (n == 0) return false;
        memcpy(unit_, scratch_, n);
        unit_[n] = '\0';
        return true;
    }
    int as_int() const { return (int)strtol(unit_, nullptr, 10); }  // atoi
    float as_float() const { return (float)strtod(unit_, nullptr); }  // atof, then double -> float

   private:
    static constexpr size_t kMax = 1024;  // MAX_LINE, BaseTahoeTest.h:269,356
    FILE *fp_;
    std::vector<char> chunk_;
    size_t pos_ = 0, len_ = 0;
    char scratch_[kMax];
    char unit_[kMax];
};

}  // namespace
}  // namespace tahoe

using namespace tahoe;

extern "C" {

const char *tahoe_last_error(void) { return g_last_error.c_str(); }
int tahoe_abi_version(void) { return TAHOE_AMD_ABI_VERSION; }

int tahoe_tree_num_nodes(int depth) { return (1 << (depth + 1)) - 1; }

void tahoe_encode_node(tahoe_dense_node *n, int fid, float value, int def_left, float weight, int is_leaf)
{
    n->weight = weight;
    n->val = value;
    n->bits = (fid & kFidMask) | (def_left ? kDefLeftMask : 0) | (is_leaf ? kIsLeafMask : 0);
}

void tahoe_decode_node(const tahoe_dense_node *n, float *value, float *weight, int *fid, int *def_left,
                       int *is_leaf)
{
    if (value) *value = n->val;
    if (weight) *weight = n->weight;
    if (fid) *fid = n->bits & kFidMask;
    if (def_left) *def_left = (n->bits & kDefLeftMask) != 0;
    if (is_leaf) *is_leaf = (n->bits & kIsLeafMask) != 0;
}

tahoe_status tahoe_load_model(const char *path, int *num_trees, int *depth, tahoe_dense_node **nodes_out)
{
    if (!path || !num_trees || !depth || !nodes_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_model: null argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(TAHOE_ERR_IO, "fail to read: %s: %s", path, strerror(errno));
    LineFeed in(fp);
    if (in.next()) *num_trees = in.as_int();
    if (in.next()) *depth = in.as_int() - 1;  // the file stores levels = depth + 1
    if (*num_trees < 0 || *depth < 0 || *depth > 30) {
        fclose(fp);
        return fail(TAHOE_ERR_INVALID_ARG, "model header out of range: num_trees=%d depth=%d", *num_trees,
                    *depth);
    }
    const size_t total = (size_t)*num_trees * (size_t)tahoe_tree_num_nodes(*depth);
    tahoe_dense_node *nodes = (tahoe_dense_node *)malloc((total ? total : 1) * sizeof(tahoe_dense_node));
    if (!nodes) {
        fclose(fp);
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_load_model: %zu nodes", total);
    }
    for (size_t i = 0; i < total; ++i) {
        in.next();
        const int fid = in.as_int();
        in.next();
        const float value = in.as_float();
        in.next();
        const bool def_left = in.as_int() != 0;
        in.next();
        const float weight = in.as_float();
        in.next();
        const bool is_leaf = in.as_int() != 0;
        tahoe_encode_node(&nodes[i], fid, value, def_left, weight, is_leaf);
    }
    fclose(fp);
    *nodes_out = nodes;
    return TAHOE_OK;
}

tahoe_status tahoe_load_data(const char *path, int *num_rows, int *num_cols, float *missing, float **data_out)
{
    if (!path || !num_rows || !num_cols || !missing || !data_out)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_load_data: null argument");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(TAHOE_ERR_IO, "fail to read: %s: %s", path, strerror(errno));
    LineFeed in(fp);
    if (in.next()) *num_rows = in.as_int();
    if (in.next()) *num_cols = in.as_int();
    if (in.next()) *missing = in.as_float();
    if (*num_rows < 0 || *num_cols < 0) {
        fclose(fp);
        return fail(TAHOE_ERR_INVALID_ARG, "data header out of range: rows=%d cols=%d", *num_rows, *num_cols);
    }
    const size_t total = (size_t)*num_rows * (size_t)*num_cols;  // 64-bit (int in the reference, :377)
    float *data = (float *)malloc((total ? total : 1) * sizeof(float));
    if (!data) {
        fclose(fp);
        return fail(TAHOE_ERR_NO_MEMORY, "tahoe_load_data: %zu values", total);
    }
    for (size_t i = 0; i < total; ++i) {
        in.next();
        data[i] = in.as_float();
    }
    fclose(fp);
    *data_out = data;
    return TAHOE_OK;
}

tahoe_status tahoe_write_model(const char *path, int num_trees, int depth, const tahoe_dense_node *nodes)
{
    if (!path || !nodes || num_trees < 0 || depth < 0)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_write_model: bad argument");
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(TAHOE_ERR_IO, "cannot write %s: %s", path, strerror(errno));
    fprintf(fp, "%d\n%d\n", num_trees, depth + 1);
    const size_t total = (size_t)num_trees * (size_t)tahoe_tree_num_nodes(depth);
    for (size_t i = 0; i < total; ++i) {
        int fid, def_left, is_leaf;
        float value, weight;
        tahoe_decode_node(&nodes[i], &value, &weight, &fid, &def_left, &is_leaf);
        // %.9g round-trips every float32 through strtod -> float.
        fprintf(fp, "%d\n%.9g\n%d\n%.9g\n%d\n", fid, value, def_left, weight, is_leaf);
    }
    if (fclose(fp) != 0) return fail(TAHOE_ERR_IO, "write error on %s", path);
    return TAHOE_OK;
}

tahoe_status tahoe_write_data(const char *path, int num_rows, int num_cols, float missing, const float *data)
{
    if (!path || (!data && num_rows * (size_t)num_cols) || num_rows < 0 || num_cols < 0)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_write_data: bad argument");
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(TAHOE_ERR_IO, "cannot write %s: %s", path, strerror(errno));
    fprintf(fp, "%d\n%d\n%.9g\n", num_rows, num_cols, missing);
    const size_t total = (size_t)num_rows * (size_t)num_cols;
    for (size_t i = 0; i < total; ++i) fprintf(fp, "%.9g\n", data[i]);
    if (fclose(fp) != 0) return fail(TAHOE_ERR_IO, "write error on %s", path);
    return TAHOE_OK;
}

void tahoe_free_host(void *p) { free(p); }

void tahoe_synth_forest(tahoe_dense_node *nodes, int num_trees, int depth, int num_cols, uint64_t seed,
                        float leaf_prob)
{
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(depth);
    const size_t first_bottom = ((size_t)1 << depth) - 1;
    const uint64_t cols = num_cols > 0 ? (uint64_t)num_cols : 1;
    for (size_t t = 0; t < (size_t)num_trees; ++t) {
        for (size_t j = 0; j < per_tree; ++j) {
            const uint64_t g = (t * per_tree + j) * 4;
            const uint64_t x0 = splitmix64_at(seed, g), x1 = splitmix64_at(seed, g + 1),
                           x2 = splitmix64_at(seed, g + 2), x3 = splitmix64_at(seed, g + 3);
            const int fid = (int)(x0 % cols);
            const float value = 2.0f * u01(x1) - 1.0f;
            const bool bottom = j >= first_bottom;
            const bool is_leaf = bottom || (u01(x3) < leaf_prob);
            tahoe_encode_node(&nodes[t * per_tree + j], is_leaf ? 0 : fid, value, (int)(x2 & 1), u01(x2), is_leaf);
        }
    }
}

void tahoe_synth_data(float *out, size_t first_row, size_t rows, int num_cols, uint64_t seed, float missing_prob,
                      float missing, float nan_prob)
{
    const size_t cols = (size_t)num_cols;
    for (size_t r = 0; r < rows; ++r) {
        for (size_t c = 0; c < cols; ++c) {
            const uint64_t e = ((first_row + r) * cols + c) * 2;
            float v = 2.0f * u01(splitmix64_at(seed, e)) - 1.0f;
            if (missing_prob > 0.0f || nan_prob > 0.0f) {
                const float p = u01(splitmix64_at(seed, e + 1));
                if (p < missing_prob)
                    v = missing;
                else if (p < missing_prob + nan_prob)
                    v = std::numeric_limits<float>::quiet_NaN();
            }
            out[r * cols + c] = v;
        }
    }
}

}  // extern "C"
