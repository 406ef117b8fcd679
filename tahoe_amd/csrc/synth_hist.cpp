// Deterministic synthetic inputs in the style of histogram-trained gradient-boosted models (XGBoost / LightGBM with
// max_bin <= 255) -- the model families the reference is run on (run_all_15_examples.sh:51-65) -- as a second generator
// beside tahoe_synth_forest, whose features and thresholds are uniform.  What differs, and why it matters here:
//   * every feature has at most `max_bins` distinct thresholds, the bin edges of a 4096-value sample of that feature
//     (quantiles: histogram training) -> QRING's rank codes fit 8 bits, the quantise pass takes its small-table form;
//   * features are used with Zipf-skewed frequencies (a few features carry most splits) -> LDS bank / cache-line reuse
//     differs from the uniform case;
//   * a node splits the part of the feature's range that its ancestors left, at a skewed position -> unbalanced branch
//     probabilities (`weight` = the node's reach probability under the data generator), no dead branches;
//   * nodes whose reach probability is small become leaves early (min_child_weight), plus `leaf_prob`;
//   * features live on different scales (`scale_decades` decades of spread) and have different shapes: uniform,
//     exponential, bell-shaped, small integer counts (thresholds at half-integers).
// tahoe_synth_data_hist draws rows from the same per-feature distributions (same feature_seed), so thresholds ARE
// quantiles of the data.  Everything is counter-based SplitMix64: element i of a stream never depends on element i-1.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"

namespace {

using tahoe::splitmix64_at;
using tahoe::u01;

struct FeatureModel {
    int kind;     // 0 uniform, 1 exponential, 2 bell (sum of four uniforms), 3 integer counts
    float loc, scale;
    int levels;   // kind 3: values 0 .. levels
};

FeatureModel feature_model(uint64_t feature_seed, int f, float scale_decades)
{
    FeatureModel m;
    const uint64_t a = splitmix64_at(feature_seed, (uint64_t)f * 4), b = splitmix64_at(feature_seed, (uint64_t)f * 4 + 1),
                   c = splitmix64_at(feature_seed, (uint64_t)f * 4 + 2);
    m.kind = (int)(a & 3);
    m.scale = std::pow(10.0f, scale_decades * (u01(b) - 0.5f));
    m.loc = m.kind == 1 ? 0.0f : m.scale * (2.0f * u01(c) - 1.0f);
    m.levels = 2 + (int)((c >> 8) % 60);  // 2 .. 61 distinct counts
    return m;
}

// value of feature f for the uniform draws (u, and three more for the bell shape)
float feature_value(const FeatureModel &m, float u, float u2, float u3, float u4)
{
    switch (m.kind) {
    case 0: return m.loc + m.scale * (2.0f * u - 1.0f);
    case 1: return m.scale * -std::log1p(-std::min(u, 0.99999994f));
    case 2: return m.loc + m.scale * ((u + u2 + u3 + u4) - 2.0f);
    default: return std::floor(u * u * (float)(m.levels + 1));  // skewed towards small counts; not scaled (counts are counts)
    }
}

// bin edges of feature f: quantiles of a deterministic 4096-value sample (distinct, ascending, at most max_bins)
std::vector<float> feature_edges(uint64_t feature_seed, int f, const FeatureModel &m, int max_bins)
{
    const int n = 4096;
    std::vector<float> sample((size_t)n);
    const uint64_t s = feature_seed ^ (0xA5A5A5A5ull + (uint64_t)f * 0x9E3779B97F4A7C15ull);
    for (int i = 0; i < n; ++i)
        sample[(size_t)i] = feature_value(m, u01(splitmix64_at(s, (uint64_t)i * 4)), u01(splitmix64_at(s, (uint64_t)i * 4 + 1)),
                                          u01(splitmix64_at(s, (uint64_t)i * 4 + 2)), u01(splitmix64_at(s, (uint64_t)i * 4 + 3)));
    std::sort(sample.begin(), sample.end());
    std::vector<float> edges;
    if (m.kind == 3) {  // counts: a threshold between every two neighbouring values that occur
        for (int i = 1; i < n; ++i)
            if (sample[(size_t)i] != sample[(size_t)i - 1]) edges.push_back(0.5f * (sample[(size_t)i] + sample[(size_t)i - 1]));
        if ((int)edges.size() > max_bins) edges.resize((size_t)max_bins);
    } else {
        for (int k = 1; k <= max_bins; ++k) {
            const float e = sample[(size_t)((long long)k * n / (max_bins + 1))];
            if (edges.empty() || e > edges.back()) edges.push_back(e);
        }
    }
    if (edges.empty()) edges.push_back(m.loc);
    return edges;
}

}  // namespace

extern "C" {

tahoe_status tahoe_synth_forest_hist(tahoe_dense_node *nodes, int num_trees, int depth, int num_cols, uint64_t seed,
                                     uint64_t feature_seed, int max_bins, float zipf_s, float leaf_prob, float scale_decades)
{
    using tahoe::fail;
    if (!nodes || num_trees < 0 || depth < 0 || depth > 30 || num_cols < 1 || max_bins < 1 || max_bins > 32767)
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_synth_forest_hist: bad argument");
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(depth);
    const size_t first_bottom = ((size_t)1 << depth) - 1;
    std::vector<FeatureModel> fm((size_t)num_cols);
    std::vector<std::vector<float>> edges((size_t)num_cols);
    for (int f = 0; f < num_cols; ++f) {
        fm[(size_t)f] = feature_model(feature_seed, f, scale_decades);
        edges[(size_t)f] = feature_edges(feature_seed, f, fm[(size_t)f], max_bins);
    }
    // Zipf over a seeded permutation of the features: P(rank k) ~ 1 / (k + 1)^s
    std::vector<int> by_rank((size_t)num_cols);
    for (int f = 0; f < num_cols; ++f) by_rank[(size_t)f] = f;
    for (int f = num_cols - 1; f > 0; --f)
        std::swap(by_rank[(size_t)f], by_rank[(size_t)(splitmix64_at(feature_seed ^ 0x5EEDull, (uint64_t)f) % (uint64_t)(f + 1))]);
    std::vector<double> cdf((size_t)num_cols);
    double tot = 0.0;
    for (int k = 0; k < num_cols; ++k) cdf[(size_t)k] = (tot += std::pow((double)(k + 1), -(double)zipf_s));
    auto pick_feature = [&](float u) {
        const double x = (double)u * tot;
        return by_rank[(size_t)(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin())];
    };
    struct Span { int fid, lo, hi; };  // edges [lo, hi) of feature fid are still inside the node's region
    tahoe::parallel_for((size_t)num_trees, 4, [&](size_t t_lo, size_t t_hi) {
        std::vector<float> reach(per_tree);
        std::vector<unsigned char> dead(per_tree);  // below a leaf: never reached, filled with harmless leaves
        std::vector<Span> path;
        for (size_t t = t_lo; t < t_hi; ++t) {
            tahoe_dense_node *tree = nodes + t * per_tree;
            std::fill(dead.begin(), dead.end(), 0);
            reach[0] = 1.0f;
            for (size_t j = 0; j < per_tree; ++j) {
                const uint64_t g = (t * per_tree + j) * 6;
                const uint64_t x0 = splitmix64_at(seed, g), x1 = splitmix64_at(seed, g + 1), x2 = splitmix64_at(seed, g + 2),
                               x3 = splitmix64_at(seed, g + 3), x4 = splitmix64_at(seed, g + 4);
                const float leaf_value = 0.1f * ((u01(x1) + u01(x4)) - 1.0f);
                const bool bottom = j >= first_bottom;
                if (dead[j]) {
                    tahoe_encode_node(&tree[j], 0, leaf_value, 0, 0.0f, 1);
                    if (!bottom) dead[2 * j + 1] = dead[2 * j + 2] = 1;
                    continue;
                }
                // the region of this node: what the ancestors' splits left of each feature they used
                path.clear();
                for (size_t c = j; c > 0;) {
                    const size_t p = (c - 1) >> 1;
                    float value, weight;
                    int fid, dl, leaf;
                    tahoe_decode_node(&tree[p], &value, &weight, &fid, &dl, &leaf);
                    const std::vector<float> &e = edges[(size_t)fid];
                    const int at = (int)(std::lower_bound(e.begin(), e.end(), value) - e.begin());  // the edge the parent split at
                    Span *s = nullptr;
                    for (Span &q : path)
                        if (q.fid == fid) s = &q;
                    if (!s) {
                        path.push_back({fid, 0, (int)e.size()});
                        s = &path.back();
                    }
                    if (c == 2 * p + 2) s->lo = std::max(s->lo, at + 1);  // right child: x >= edge[at]
                    else s->hi = std::min(s->hi, at);                     // left child: x < edge[at]
                    c = p;
                }
                // early leaves: the configured probability, and regions hardly any row reaches (min_child_weight)
                bool is_leaf = bottom || u01(x3) < leaf_prob || reach[j] < 0.5f / (float)(1u << std::min(depth, 20));
                int fid = 0, lo = 0, hi = 0;
                if (!is_leaf) {
                    bool found = false;
                    for (int attempt = 0; attempt < 4 && !found; ++attempt) {  // a feature with an edge left inside the region
                        fid = pick_feature(u01(splitmix64_at(seed, g + 5) + (uint64_t)attempt * 0x9E3779B97F4A7C15ull));
                        lo = 0;
                        hi = (int)edges[(size_t)fid].size();
                        for (const Span &q : path)
                            if (q.fid == fid) {
                                lo = q.lo;
                                hi = q.hi;
                            }
                        found = hi > lo;
                    }
                    is_leaf = !found;
                }
                if (is_leaf) {
                    tahoe_encode_node(&tree[j], 0, leaf_value, 0, reach[j], 1);
                    if (!bottom) dead[2 * j + 1] = dead[2 * j + 2] = 1;
                    continue;
                }
                // split position: skewed towards the lower end of what is left (unbalanced branch probabilities)
                const float u = u01(x0);
                const int at = lo + std::min(hi - lo - 1, (int)((float)(hi - lo) * u * u));
                // edges are (k + 1) / (n + 1) quantiles: the region holds the data between edge lo - 1 and edge hi
                const float p_left = (float)(at + 1 - lo) / (float)(hi + 1 - lo);
                tahoe_encode_node(&tree[j], fid, edges[(size_t)fid][(size_t)at], (int)(x2 & 1), reach[j], 0);
                reach[2 * j + 1] = reach[j] * std::min(std::max(p_left, 0.0f), 1.0f);
                reach[2 * j + 2] = reach[j] * std::min(std::max(1.0f - p_left, 0.0f), 1.0f);
            }
        }
    });
    return TAHOE_OK;
}

tahoe_status tahoe_synth_data_hist(float *out, size_t first_row, size_t rows, int num_cols, uint64_t seed, uint64_t feature_seed,
                                   float scale_decades, float missing_prob, float missing)
{
    if (!out || num_cols < 1) return tahoe::fail(TAHOE_ERR_INVALID_ARG, "tahoe_synth_data_hist: bad argument");
    const size_t cols = (size_t)num_cols;
    std::vector<FeatureModel> fm(cols);
    for (size_t f = 0; f < cols; ++f) fm[f] = feature_model(feature_seed, (int)f, scale_decades);
    tahoe::parallel_for(rows, 1024, [&](size_t r_lo, size_t r_hi) {
        for (size_t r = r_lo; r < r_hi; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const uint64_t e = ((first_row + r) * cols + c) * 5;
                float v = feature_value(fm[c], u01(splitmix64_at(seed, e)), u01(splitmix64_at(seed, e + 1)), u01(splitmix64_at(seed, e + 2)),
                                        u01(splitmix64_at(seed, e + 3)));
                if (missing_prob > 0.0f && u01(splitmix64_at(seed, e + 4)) < missing_prob) v = missing;
                out[r * cols + c] = v;
            }
    });
    return TAHOE_OK;
}

}  // extern "C"
