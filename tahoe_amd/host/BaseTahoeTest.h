// BaseTahoeTest -- the reference's C++ load / SetUp / predict surface, on top of libtahoe_amd.so.
//
// Source-compatible with the class in the reference's BaseTahoeTest.h (:45-907): same class name,
// constructor signature (:49), SetUp(float&) (:71), TearDown() (:117), Free() (:873), the public fields
// ps / nodes / data_h / data_d / preds_d / want_preds_d / stream (:889-906), tree_num_nodes() /
// forest_num_nodes() (:884-886) and the same stdout lines.  Plain C++: no HIP or CUDA header is needed
// to compile code that uses it; all device work goes through the C ABI in tahoe_amd.h.
//
// What SetUp does, in the reference's order (:71-115): load the model and the data from the two text
// files, predict on the CPU "to get standard results" (the harness's own single-threaded check, as in the
// reference :458-487 -- it is never used to produce predictions), time a baseline strategy and then every
// traversal strategy of the library (5 warm-ups, then 5 / 50 timed calls, µs per sample, :549-710),
// compare each against the CPU result with the reference's absolute 1e-3 tolerance (compare_GPU,
// cuda_base.h:98-111) and return the 1-based index of the fastest strategy plus speedup = baseline / best.
//   baseline ("FIL (baseline)")  = TAHOE_STRATEGY_DIRECT
//   strategy 1..5               = DIRECT, ROWTILE, TILEBLOCK, TILERING, QRING (the library's numbering)
#ifndef TAHOE_AMD_BASETAHOETEST_H
#define TAHOE_AMD_BASETAHOETEST_H

#include <sys/time.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tahoe_amd.h"

// names the reference's client code uses (Struct.h:23-48)
typedef tahoe_dense_node dense_node_t;
enum algo_t { NAIVE = TAHOE_ALGO_NAIVE, TREE_REORG = TAHOE_ALGO_TREE_REORG, BATCH_TREE_REORG = TAHOE_ALGO_BATCH_TREE_REORG };
enum strategy_t {
    SHARED_DATA = TAHOE_FIL_SHARED_DATA,
    SHARED_FOREST = TAHOE_FIL_SHARED_FOREST,
    SPLIT_FOREST = TAHOE_FIL_SPLIT_FOREST,
    SPLIT_FOREST_SHARED_DATA = TAHOE_FIL_SPLIT_FOREST_SHARED_DATA
};
enum output_t { RAW = TAHOE_OUT_RAW, AVG = TAHOE_OUT_AVG, SIGMOID = TAHOE_OUT_SIGMOID, THRESHOLD = TAHOE_OUT_THRESHOLD };

// TahoeTestParams, Struct.h:120-142
struct TahoeTestParams {
    int num_rows;
    int num_cols;
    float nan_prob;
    int depth;
    int num_trees;
    float leaf_prob;
    output_t output;
    float threshold;
    float global_bias;
    algo_t algo;
    int seed;
    float tolerance;
    strategy_t strategy;
    char input_model_file[1024];
    char input_data_file[1024];
    float missing;
};

static int epoch_new = 50;  // timed calls per strategy (BaseTahoeTest.h:43)

class BaseTahoeTest {
   public:
    BaseTahoeTest(std::string input_model_file, std::string input_data_file, int algorithm = 0, int num_rows = 1000,
                  int num_cols = 500, float nan_prob = 0.0, int depth = 20, int num_trees = 10, float leaf_prob = 0.0,
                  output_t output = output_t::RAW, float threshold = 0.0, float global_bias = 0.0,
                  algo_t algo = algo_t::NAIVE, int seed = 0, float tolerance = 1e-3f,
                  strategy_t strategy = strategy_t::SHARED_DATA)
    {
        memset(&ps, 0, sizeof(ps));
        ps.num_rows = num_rows;
        ps.num_cols = num_cols;
        ps.nan_prob = nan_prob;
        ps.depth = depth;
        ps.num_trees = num_trees;
        ps.leaf_prob = leaf_prob;
        ps.output = output;
        ps.threshold = threshold;
        ps.global_bias = global_bias;
        ps.algo = algo;
        ps.seed = seed;
        ps.tolerance = tolerance;
        ps.strategy = strategy;
        snprintf(ps.input_model_file, sizeof(ps.input_model_file), "%s", input_model_file.c_str());
        snprintf(ps.input_data_file, sizeof(ps.input_data_file), "%s", input_data_file.c_str());
        selected_algorithm = algorithm;
    }

    int SetUp(float &speedup)
    {
        float *acc = last_acc;
        check(tahoe_stream_create(&stream), "tahoe_stream_create");
        printf("Loading model...\n");
        generate_forest_from_file();
        printf("Loading data...\n");
        generate_data_from_file();
        printf("Predict on CPU to get standard results...\n");
        predict_on_cpu();
        printf("Test on GPU...\n");
        init_forest();
        const float baseline = predict_on_gpu_baseline();
        last_baseline = baseline;
        predict_on_gpu_strategies(acc);
        int algorithm = 0;
        float best = FLT_MAX;
        for (int i = 0; i < 5; ++i)
            if (acc[i] < best) {
                algorithm = i + 1;
                best = acc[i];
            }
        speedup = baseline / best;
        return algorithm;
    }

    void TearDown()
    {
        tahoe_device_free(preds_d);
        tahoe_device_free(want_preds_d);
        tahoe_device_free(data_d);
        preds_d = want_preds_d = data_d = nullptr;
    }

    void Free()
    {
        TearDown();
        tahoe_forest_destroy(forest);
        forest = nullptr;
        if (stream) tahoe_stream_destroy(stream);
        stream = nullptr;
    }

    int tree_num_nodes() { return tahoe_tree_num_nodes(ps.depth); }
    int forest_num_nodes() { return ps.num_trees * tree_num_nodes(); }

    // -- pieces of SetUp, public as in the reference --------------------------------------------
    void generate_forest_from_file()
    {
        tahoe_dense_node *loaded = nullptr;
        // TAHOE_BIN_CACHE set: keep "<file>.tbin" beside the text file and read that while it is current
        const tahoe_status st = getenv("TAHOE_BIN_CACHE")
                                    ? tahoe_load_model_cached(ps.input_model_file, &ps.num_trees, &ps.depth, &loaded, nullptr)
                                    : tahoe_load_model(ps.input_model_file, &ps.num_trees, &ps.depth, &loaded);
        if (st != TAHOE_OK) {
            fprintf(stderr, "%s\n", tahoe_last_error());  // the reference: perror("fail to read"); exit(1)
            exit(1);
        }
        nodes.assign(loaded, loaded + (size_t)forest_num_nodes());
        tahoe_free_host(loaded);
    }

    void generate_data_from_file()
    {
        float *loaded = nullptr;
        const tahoe_status st =
            getenv("TAHOE_BIN_CACHE")
                ? tahoe_load_data_cached(ps.input_data_file, &ps.num_rows, &ps.num_cols, &ps.missing, &loaded, nullptr)
                : tahoe_load_data(ps.input_data_file, &ps.num_rows, &ps.num_cols, &ps.missing, &loaded);
        if (st != TAHOE_OK) {
            fprintf(stderr, "%s\n", tahoe_last_error());
            exit(1);
        }
        const size_t n = (size_t)ps.num_rows * (size_t)ps.num_cols;
        data_h.assign(loaded, loaded + n);
        tahoe_free_host(loaded);
        check(tahoe_device_alloc((void **)&data_d, n * sizeof(float), 1), "tahoe_device_alloc(data_d)");
        check(tahoe_copy_to_device(data_d, data_h.data(), n * sizeof(float), stream), "tahoe_copy_to_device(data_d)");
    }

    // One tree, as the CPU check walks it: returns the leaf value.
    float infer_one_tree(const dense_node_t *root, const float *row)
    {
        int at = 0;
        for (;;) {
            float value = 0.f;
            int fid = 0, def_left = 0, is_leaf = 0;
            tahoe_decode_node(&root[at], &value, nullptr, &fid, &def_left, &is_leaf);
            if (is_leaf) return value;
            const float x = row[fid];
            const bool right = (std::fabs(x - ps.missing) <= 1.0e-6f) ? !def_left : (x >= value);
            at = 2 * at + (right ? 2 : 1);
        }
    }

    void predict_on_cpu()
    {
        std::vector<float> want(ps.num_rows);
        const size_t per_tree = (size_t)tree_num_nodes();
        for (int r = 0; r < ps.num_rows; ++r) {
            float pred = 0.0f;
            for (int t = 0; t < ps.num_trees; ++t)
                pred += infer_one_tree(&nodes[(size_t)t * per_tree], &data_h[(size_t)r * ps.num_cols]);
            if (ps.output & output_t::AVG) pred = pred / ps.num_trees;
            pred += ps.global_bias;
            if (ps.output & output_t::SIGMOID) pred = 1.0f / (1.0f + expf(-pred));
            if (ps.output & output_t::THRESHOLD) pred = pred > ps.threshold ? 1.0f : 0.0f;
            want[r] = pred;
        }
        check(tahoe_device_alloc((void **)&want_preds_d, want.size() * sizeof(float), 1), "tahoe_device_alloc");
        check(tahoe_copy_to_device(want_preds_d, want.data(), want.size() * sizeof(float), stream), "copy");
        check(tahoe_stream_synchronize(stream), "tahoe_stream_synchronize");
    }

    // Additions to the reference surface (machine-readable results, leaf-index dump).
    // Per-(row, tree) leaf indices in the reference's heap numbering, rows x trees uint32, row-major.
    bool dump_leaf_indices(const char *path)
    {
        const size_t n = (size_t)ps.num_rows * (size_t)ps.num_trees;
        uint32_t *leaf_d = nullptr;
        if (tahoe_device_alloc((void **)&leaf_d, n * sizeof(uint32_t), 0) != TAHOE_OK) return false;
        std::vector<uint32_t> leaf(n);
        bool ok = tahoe_forest_predict_leaf_idx(forest, leaf_d, nullptr, data_d, (size_t)ps.num_rows, stream) == TAHOE_OK &&
                  tahoe_copy_to_host(leaf.data(), leaf_d, n * sizeof(uint32_t), stream) == TAHOE_OK &&
                  tahoe_stream_synchronize(stream) == TAHOE_OK;
        tahoe_device_free(leaf_d);
        if (!ok) return false;
        FILE *fp = fopen(path, "wb");
        if (!fp) return false;
        ok = fwrite(leaf.data(), sizeof(uint32_t), n, fp) == n;
        return fclose(fp) == 0 && ok;
    }

    bool write_result_json(const char *path, int best_by_run, float speedup)
    {
        FILE *fp = fopen(path, "w");
        if (!fp) return false;
        fprintf(fp, "{\"model\": \"%s\", \"data\": \"%s\", \"num_trees\": %d, \"depth\": %d, \"num_rows\": %d, \"num_cols\": %d, ",
                ps.input_model_file, ps.input_data_file, ps.num_trees, ps.depth, ps.num_rows, ps.num_cols);
        fprintf(fp, "\"baseline_us_per_sample\": %.6f, \"strategy_us_per_sample\": [", last_baseline);
        for (int i = 0; i < 5; ++i) {
            if (last_acc[i] == FLT_MAX)
                fprintf(fp, "null%s", i < 4 ? ", " : "");
            else
                fprintf(fp, "%.6f%s", last_acc[i], i < 4 ? ", " : "");
        }
        fprintf(fp, "], \"best_strategy\": %d, \"auto_strategy\": %d, \"speedup\": %.4f}\n", best_by_run, auto_strategy, speedup);
        return fclose(fp) == 0;
    }

    TahoeTestParams ps;
    std::vector<dense_node_t> nodes;
    std::vector<float> data_h;
    float *data_d = nullptr;
    float *preds_d = nullptr;
    float *want_preds_d = nullptr;
    void *stream = nullptr;  // hipStream_t behind the C ABI
    tahoe_forest *forest = nullptr;
    int selected_algorithm = 0;
    float last_acc[6] = {0, 0, 0, 0, 0, 0};  // us/sample of strategies 1..5 from the last SetUp (FLT_MAX = not suitable)
    float last_baseline = 0.0f;              // us/sample of the baseline run
    int auto_strategy = 0;  // what TAHOE_STRATEGY_AUTO resolves to for this shape (1-based, as SetUp returns)

   private:
    static void check(tahoe_status s, const char *what)
    {
        // the reference prints and continues (CUDA_CHECK, cuda_base.h:19-25)
        if (s != TAHOE_OK) printf("FAIL: call='%s'. Reason:%s\n", what, tahoe_last_error());
    }

    void init_forest()
    {
        tahoe_forest_params fp;
        memset(&fp, 0, sizeof(fp));
        fp.depth = ps.depth;
        fp.num_trees = ps.num_trees;
        fp.num_cols = ps.num_cols;
        fp.algo = ps.algo;
        fp.output = ps.output;
        fp.threshold = ps.threshold;
        fp.global_bias = ps.global_bias;
        fp.strategy = ps.strategy;
        fp.missing = ps.missing;
        if (tahoe_forest_create(&forest, nodes.data(), &fp) != TAHOE_OK) {
            fprintf(stderr, "tahoe_forest_create: %s\n", tahoe_last_error());
            exit(1);
        }
        auto_strategy = tahoe_forest_get_strategy(forest, (size_t)ps.num_rows);
        check(tahoe_device_alloc((void **)&preds_d, (size_t)ps.num_rows * sizeof(float), 1), "tahoe_device_alloc(preds_d)");
    }

    // `warm` untimed + `timed` timed predicts on `stream`; returns µs per sample, or FLT_MAX when a predict failed or
    // a kernel raised the handle's error flag (tahoe_forest_check) -- preds_d is cleared first, so that a strategy
    // that produced nothing cannot pass the comparison on the previous strategy's output.
    float time_predicts(int warm, int timed)
    {
        bool ok = tahoe_device_memset(preds_d, 0, (size_t)ps.num_rows * sizeof(float), stream) == TAHOE_OK;
        for (int i = 0; i < warm && ok; ++i) ok = tahoe_forest_predict(forest, preds_d, data_d, (size_t)ps.num_rows, stream) == TAHOE_OK;
        tahoe_device_synchronize();
        struct timeval start, end;
        gettimeofday(&start, NULL);
        for (int i = 0; i < timed && ok; ++i) ok = tahoe_forest_predict(forest, preds_d, data_d, (size_t)ps.num_rows, stream) == TAHOE_OK;
        tahoe_device_synchronize();
        gettimeofday(&end, NULL);
        if (ok) ok = tahoe_forest_check(forest, stream) == TAHOE_OK;
        if (!ok) {
            printf("FAIL: predict. Reason:%s\n", tahoe_last_error());
            return FLT_MAX;
        }
        const float us = (end.tv_sec - start.tv_sec) * 1000000.0f + (end.tv_usec - start.tv_usec);
        return us / ps.num_rows / timed;
    }

    void report_compare()
    {
        size_t bad = 0;
        check(tahoe_compare_device(preds_d, want_preds_d, (size_t)ps.num_rows, 1e-3f, &bad, stream), "tahoe_compare_device");
        printf(bad == 0 ? "Results are correct\n" : "Results are incorrect\n");
    }

    float predict_on_gpu_baseline()
    {
        tahoe_forest_set_strategy(forest, TAHOE_STRATEGY_DIRECT);
        const float us = time_predicts(5, 5);
        printf("Exec.Time/Sample on FIL (baseline) is %f us\n", us);
        report_compare();
        return us;
    }

    void predict_on_gpu_strategies(float *acc)
    {
        static const int order[5] = {TAHOE_STRATEGY_DIRECT, TAHOE_STRATEGY_ROWTILE, TAHOE_STRATEGY_TILEBLOCK,
                                     TAHOE_STRATEGY_TILERING, TAHOE_STRATEGY_QRING};
        for (int loop = 0; loop <= 4; ++loop) {
            if (tahoe_forest_set_strategy(forest, order[loop]) != TAHOE_OK) {
                acc[loop] = FLT_MAX;
                printf("Strategy %d is not suitable for this case.\n", loop + 1);
                continue;
            }
            printf("Using strategy %d\n", loop + 1);
            acc[loop] = time_predicts(5, epoch_new);
            if (acc[loop] == FLT_MAX) {  // a failed launch / raised error flag: same line as an infeasible strategy
                printf("Strategy %d is not suitable for this case.\n", loop + 1);
                continue;
            }
            printf("Exec.Time/Sample on strategy %d is %f us\n", loop + 1, acc[loop]);
            report_compare();
        }
        tahoe_forest_set_strategy(forest, TAHOE_STRATEGY_AUTO);
    }
};

#endif  // TAHOE_AMD_BASETAHOETEST_H
