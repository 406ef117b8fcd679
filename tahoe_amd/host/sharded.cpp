// ./TahoeSharded <model> <data> [gpus] [--emulate K] -- a forest too large for one GPU's caches, split by trees
// (north_star): one process drives G devices; device g holds trees [T*g/G, T*(g+1)/G) and a copy of
// all rows, computes partial float32 sums with the single-GPU library, then ONE ncclAllReduce (RCCL over xGMI,
// 4 bytes per row) gives every device the total, and the output transform runs on it.  Plain C++ on the C ABI +
// rccl.h; the HIP types appear only as the stream argument of the collective.
// --emulate K: K shards on device 0, one after the other, partials added on the host in shard order -- the
// partition logic without a second GPU (what a 1-GPU box can check).
// The per-shard sums are bit-identical to the CPU's partial sums; the total differs from the sequential CPU sum only
// by the float32 association of the shard totals (checked to 1e-6 relative; exact for G = 1).
#include <rccl/rccl.h>
#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tahoe_amd.h"

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, tahoe_last_error());
    exit(1);
}
#define OK(call)                        \
    do {                                \
        if ((call) != TAHOE_OK) die(#call); \
    } while (0)

static float cpu_tree(const tahoe_dense_node *root, const float *row, float missing)
{
    int at = 0;
    for (;;) {
        float value = 0.f;
        int fid = 0, def_left = 0, is_leaf = 0;
        tahoe_decode_node(&root[at], &value, nullptr, &fid, &def_left, &is_leaf);
        if (is_leaf) return value;
        const float x = row[fid];
        const bool right = (std::fabs(x - missing) <= 1.0e-6f) ? !def_left : (x >= value);
        at = 2 * at + (right ? 2 : 1);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        printf("usage: %s <model> <data> [gpus] [--emulate K]\n", argv[0]);
        return 2;
    }
    int want_gpus = 0, emulate = 0;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--emulate") && i + 1 < argc)
            emulate = atoi(argv[++i]);
        else
            want_gpus = atoi(argv[i]);
    }
    int T = 10, D = 20, R = 1000, C = 500;
    float missing = 0.f;
    tahoe_dense_node *nodes = nullptr;
    float *data = nullptr;
    OK(tahoe_load_model(argv[1], &T, &D, &nodes));
    OK(tahoe_load_data(argv[2], &R, &C, &missing, &data));
    int ndev = 0;
    OK(tahoe_device_count(&ndev));
    int G = emulate > 0 ? emulate : (want_gpus > 0 ? want_gpus : ndev);
    if (emulate == 0 && G > ndev) G = ndev;
    if (G > T) G = T > 0 ? T : 1;
    if (G < 1) G = 1;
    printf("%d trees of depth %d, %d rows x %d cols, %d shard(s)%s\n", T, D, R, C, G, emulate ? " emulated on device 0" : "");
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(D);
    const size_t rows = (size_t)R, dbytes = rows * (size_t)C * sizeof(float);

    std::vector<tahoe_forest *> shard((size_t)G, nullptr);
    std::vector<void *> stream((size_t)G, nullptr);
    std::vector<float *> data_d((size_t)G, nullptr), sums_d((size_t)G, nullptr);
    std::vector<int> devs((size_t)G);
    for (int g = 0; g < G; ++g) {
        devs[(size_t)g] = emulate ? 0 : g;
        OK(tahoe_device_set(devs[(size_t)g]));
        const int lo = (int)((long long)T * g / G), hi = (int)((long long)T * (g + 1) / G);
        tahoe_forest_params p;
        memset(&p, 0, sizeof(p));
        p.depth = D;
        p.num_trees = hi - lo;
        p.num_cols = C;
        p.output = TAHOE_OUT_RAW;
        p.missing = missing;
        OK(tahoe_forest_create(&shard[(size_t)g], nodes + (size_t)lo * per_tree, &p));
        OK(tahoe_stream_create(&stream[(size_t)g]));
        if (emulate && g > 0) {
            data_d[(size_t)g] = data_d[0];
        } else {
            OK(tahoe_device_alloc((void **)&data_d[(size_t)g], dbytes, 0));
            OK(tahoe_copy_to_device(data_d[(size_t)g], data, dbytes, stream[(size_t)g]));
        }
        OK(tahoe_device_alloc((void **)&sums_d[(size_t)g], rows * sizeof(float), 1));
        OK(tahoe_forest_reserve(shard[(size_t)g], rows));
    }
    std::vector<ncclComm_t> comms((size_t)G);
    if (!emulate) {
        const ncclResult_t r = ncclCommInitAll(comms.data(), G, devs.data());
        if (r != ncclSuccess) {
            fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(r));
            return 1;
        }
    }
    std::vector<float> total(rows, 0.f), part(rows);
    auto predict = [&]() {
        for (int g = 0; g < G; ++g)  // every device walks its trees over all rows
            OK(tahoe_forest_predict_raw(shard[(size_t)g], sums_d[(size_t)g], data_d[(size_t)g], rows, stream[(size_t)g]));
        if (!emulate) {
            ncclGroupStart();
            for (int g = 0; g < G; ++g)
                ncclAllReduce(sums_d[(size_t)g], sums_d[(size_t)g], rows, ncclFloat32, ncclSum, comms[(size_t)g],
                              (hipStream_t)stream[(size_t)g]);
            ncclGroupEnd();
        }
    };
    for (int i = 0; i < 2; ++i) predict();
    for (int g = 0; g < G; ++g) OK(tahoe_stream_synchronize(stream[(size_t)g]));
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    const int reps = emulate ? 1 : 10;
    for (int i = 0; i < reps; ++i) predict();
    for (int g = 0; g < G; ++g) OK(tahoe_stream_synchronize(stream[(size_t)g]));
    gettimeofday(&t1, NULL);
    const double us = ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_usec - t0.tv_usec)) / reps;
    if (emulate) {
        for (int g = 0; g < G; ++g) {  // partials added in shard order
            OK(tahoe_copy_to_host(part.data(), sums_d[(size_t)g], rows * sizeof(float), stream[(size_t)g]));
            OK(tahoe_stream_synchronize(stream[(size_t)g]));
            for (size_t r = 0; r < rows; ++r) total[r] += part[r];
        }
    } else {
        OK(tahoe_device_set(devs[0]));
        OK(tahoe_copy_to_host(total.data(), sums_d[0], rows * sizeof(float), stream[0]));
        OK(tahoe_stream_synchronize(stream[0]));
    }
    printf("Exec.Time/Sample with %d tree shard(s) is %f us (%.3f ms per batch)\n", G, us / (double)R, us / 1e3);
    // check against the sequential CPU sum (BaseTahoeTest.h:458-474), 1e-6 relative (exact for one shard)
    size_t bad = 0;
    double worst = 0.0;
    for (size_t r = 0; r < rows; ++r) {
        float want = 0.0f;
        for (int t = 0; t < T; ++t) want += cpu_tree(nodes + (size_t)t * per_tree, data + r * (size_t)C, missing);
        const double err = std::fabs((double)total[r] - (double)want), tol = 1e-6 * std::fmax(std::fabs((double)want), 1e-30);
        if (G == 1 ? (memcmp(&total[r], &want, 4) != 0) : (err > tol && err > 1e-6)) ++bad;
        if (err > worst) worst = err;
    }
    printf("max abs difference to the CPU sum: %g\n", worst);
    printf(bad == 0 ? "Results are correct\n" : "Results are incorrect\n");
    for (int g = 0; g < G; ++g) {
        if (!emulate) ncclCommDestroy(comms[(size_t)g]);
        tahoe_forest_destroy(shard[(size_t)g]);
        if (!(emulate && g > 0)) tahoe_device_free(data_d[(size_t)g]);
        tahoe_device_free(sums_d[(size_t)g]);
        tahoe_stream_destroy(stream[(size_t)g]);
    }
    tahoe_free_host(nodes);
    tahoe_free_host(data);
    return bad == 0 ? 0 : 1;
}
