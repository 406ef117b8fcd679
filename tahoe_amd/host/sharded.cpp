// ./TahoeSharded <model> <data> [gpus] [--emulate K] [--mode rows|allreduce64|allreduce32|chain] [--chunk ROWS]
// A forest on the GPUs of one node (north_star): one process drives G devices.  Plain C++ on the C ABI +
// rccl.h; the HIP types appear only as the stream argument of the collective.
//   rows                   what choose_sharding (tahoe_amd/sharding.py) picks whenever the forest fits one GPU: every device
//                          holds the WHOLE forest and rows [R*g/G, R*(g+1)/G); no data-path collective, every sum is the
//                          sequential float32 sum of predict_on_cpu, bit for bit.
// The other three split the forest by trees: device g holds trees [T*g/G, T*(g+1)/G) and a copy of all rows; three ways to
// combine the per-device sums (tahoe_amd/sharding.py has the same three for one process per GPU):
//   allreduce64 (default)  per-device sequential float32 partial sums, widened to float64, ONE ncclAllReduce of 8 bytes
//                          per row over RCCL / xGMI, rounded to float32 once.  Not bit-identical to the CPU's single
//                          sequential float32 sum; checked against a float64 CPU sum with the bound
//                          gamma(T/G) * sum|leaf| + u * |sum|, and reported next to the CPU float32 sum's own error.
//   allreduce32            the same with float32 on the wire (4 bytes per row).
//   chain                  bit-exact: rows go in chunks; device g waits for device g-1's chunk, copies its running sums
//                          over xGMI (4 bytes per row, point to point), continues them through its own trees
//                          (tahoe_forest_predict_accumulate) while device g-1 works on the next chunk.  The last device
//                          ends up with THE sequential float32 sum of predict_on_cpu (BaseTahoeTest.h:462-466).
// --emulate K: K shards on device 0 (what a 1-GPU box can check): the partition and combination logic without a second
// GPU; the all-reduce is replaced by the same arithmetic on the host (float64 or float32 adds in shard order).
#include <rccl/rccl.h>
#include <sys/time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tahoe_amd.h"

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, tahoe_last_error());
    exit(1);
}
#define OK(call)                            \
    do {                                    \
        if ((call) != TAHOE_OK) die(#call); \
    } while (0)

// every RCCL call is checked: a failed collective ends the run with its own message, not with "Results are incorrect"
#define NCCL_OK(call)                                                                   \
    do {                                                                                \
        const ncclResult_t r__ = (call);                                                \
        if (r__ != ncclSuccess) {                                                       \
            fprintf(stderr, "%s: %s (%s:%d)\n", #call, ncclGetErrorString(r__), __FILE__, __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
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
        printf("usage: %s <model> <data> [gpus] [--emulate K] [--mode rows|allreduce64|allreduce32|chain] [--chunk ROWS]\n", argv[0]);
        return 2;
    }
    int want_gpus = 0, emulate = 0;
    size_t chunk = 32768;
    std::string mode = "allreduce64";
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--emulate") && i + 1 < argc)
            emulate = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--mode") && i + 1 < argc)
            mode = argv[++i];
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc)
            chunk = (size_t)atoll(argv[++i]);
        else
            want_gpus = atoi(argv[i]);
    }
    if (mode != "rows" && mode != "allreduce64" && mode != "allreduce32" && mode != "chain") {
        fprintf(stderr, "unknown --mode %s\n", mode.c_str());
        return 2;
    }
    if (chunk < 1) chunk = 1;
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
    const bool by_rows = mode == "rows";
    if (!by_rows && G > T) G = T > 0 ? T : 1;
    if (by_rows && G > R) G = R > 0 ? R : 1;
    if (G < 1) G = 1;
    printf("%d trees of depth %d, %d rows x %d cols, %d shard(s)%s, mode %s\n", T, D, R, C, G, emulate ? " emulated on device 0" : "",
           mode.c_str());
    const size_t per_tree = (size_t)tahoe_tree_num_nodes(D);
    const size_t rows = (size_t)R, dbytes = rows * (size_t)C * sizeof(float);
    const bool chain = mode == "chain", wide = mode == "allreduce64";

    std::vector<tahoe_forest *> shard((size_t)G, nullptr);
    std::vector<void *> stream((size_t)G, nullptr);
    std::vector<float *> data_d((size_t)G, nullptr), sums_d((size_t)G, nullptr);
    std::vector<double *> wide_d((size_t)G, nullptr);
    std::vector<int> devs((size_t)G), shard_trees((size_t)G);
    std::vector<size_t> row_lo((size_t)G, 0), row_n((size_t)G, rows);  // the rows device g predicts
    for (int g = 0; g < G; ++g) {
        devs[(size_t)g] = emulate ? 0 : g;
        OK(tahoe_device_set(devs[(size_t)g]));
        const int lo = by_rows ? 0 : (int)((long long)T * g / G), hi = by_rows ? T : (int)((long long)T * (g + 1) / G);
        if (by_rows) {
            row_lo[(size_t)g] = rows * (size_t)g / (size_t)G;
            row_n[(size_t)g] = rows * (size_t)(g + 1) / (size_t)G - row_lo[(size_t)g];
        }
        shard_trees[(size_t)g] = hi - lo;
        tahoe_forest_params p;
        memset(&p, 0, sizeof(p));
        p.depth = D;
        p.num_trees = hi - lo;
        p.num_cols = C;
        p.output = TAHOE_OUT_RAW;
        p.missing = missing;
        OK(tahoe_forest_create(&shard[(size_t)g], nodes + (size_t)lo * per_tree, &p));
        OK(tahoe_stream_create(&stream[(size_t)g]));
        if (by_rows) {  // only this device's rows travel
            const size_t nb = std::max<size_t>(row_n[(size_t)g], 1) * (size_t)C * sizeof(float);
            OK(tahoe_device_alloc((void **)&data_d[(size_t)g], nb, 0));
            OK(tahoe_copy_to_device(data_d[(size_t)g], data + row_lo[(size_t)g] * (size_t)C, row_n[(size_t)g] * (size_t)C * sizeof(float),
                                    stream[(size_t)g]));
        } else if (emulate && g > 0) {
            data_d[(size_t)g] = data_d[0];
        } else {
            OK(tahoe_device_alloc((void **)&data_d[(size_t)g], dbytes, 0));
            OK(tahoe_copy_to_device(data_d[(size_t)g], data, dbytes, stream[(size_t)g]));
        }
        OK(tahoe_device_alloc((void **)&sums_d[(size_t)g], rows * sizeof(float), 1));
        if (wide) OK(tahoe_device_alloc((void **)&wide_d[(size_t)g], rows * sizeof(double), 1));
        OK(tahoe_forest_reserve(shard[(size_t)g], chain ? std::min(chunk, rows) : by_rows ? row_n[(size_t)g] : rows));
    }
    std::vector<ncclComm_t> comms((size_t)G);
    const bool use_rccl = !emulate && !chain && !by_rows;
    if (use_rccl) {
        NCCL_OK(ncclCommInitAll(comms.data(), G, devs.data()));
        for (int g = 0; g < G; ++g) {  // every communicator knows its size, rank and device
            int count = 0, rank_of = -1, dev_of = -1;
            NCCL_OK(ncclCommCount(comms[(size_t)g], &count));
            NCCL_OK(ncclCommUserRank(comms[(size_t)g], &rank_of));
            NCCL_OK(ncclCommCuDevice(comms[(size_t)g], &dev_of));
            if (count != G || rank_of != g || dev_of != devs[(size_t)g]) {
                fprintf(stderr, "RCCL communicator %d: count %d (expected %d), rank %d, device %d (expected %d)\n", g, count, G, rank_of, dev_of,
                        devs[(size_t)g]);
                return 1;
            }
        }
        printf("RCCL communicator of %d rank(s)\n", G);
    }
    // chain: one event per (device, chunk)
    const size_t n_chunks = chain ? (rows + chunk - 1) / chunk : 0;
    // ... and one for "device g has copied chunk c out of device g-1's buffer" (a device may not start chunk c of the
    // NEXT batch before its successor has taken chunk c of this one)
    std::vector<std::vector<void *>> done((size_t)G), taken((size_t)G);
    if (chain)
        for (int g = 0; g < G; ++g) {
            OK(tahoe_device_set(devs[(size_t)g]));
            done[(size_t)g].resize(n_chunks, nullptr);
            taken[(size_t)g].resize(n_chunks, nullptr);
            for (size_t c = 0; c < n_chunks; ++c) {
                OK(tahoe_event_create(&done[(size_t)g][c]));
                OK(tahoe_event_create(&taken[(size_t)g][c]));
            }
        }

    auto predict = [&]() {
        if (chain) {
            // chunk-major issue order: device g's work on chunk c is queued behind device g-1's event for chunk c
            for (size_t c = 0; c < n_chunks; ++c) {
                const size_t lo = c * chunk, n = std::min(chunk, rows - lo);
                for (int g = 0; g < G; ++g) {
                    OK(tahoe_device_set(devs[(size_t)g]));
                    float *part = sums_d[(size_t)g] + lo;
                    // never recorded (first batch) = already complete
                    if (g + 1 < G) OK(tahoe_stream_wait_event(stream[(size_t)g], taken[(size_t)g + 1][c]));
                    if (g == 0) {
                        OK(tahoe_device_memset(part, 0, n * sizeof(float), stream[0]));
                    } else {
                        OK(tahoe_stream_wait_event(stream[(size_t)g], done[(size_t)g - 1][c]));
                        OK(tahoe_copy_peer(part, devs[(size_t)g], sums_d[(size_t)g - 1] + lo, devs[(size_t)g - 1], n * sizeof(float),
                                           stream[(size_t)g]));
                        OK(tahoe_event_record(taken[(size_t)g][c], stream[(size_t)g]));
                    }
                    OK(tahoe_forest_predict_accumulate(shard[(size_t)g], part, data_d[(size_t)g] + lo * (size_t)C, n, stream[(size_t)g]));
                    OK(tahoe_event_record(done[(size_t)g][c], stream[(size_t)g]));
                }
            }
            return;
        }
        if (by_rows) {  // every device walks the whole forest over its rows; nothing to combine
            for (int g = 0; g < G; ++g) {
                OK(tahoe_device_set(devs[(size_t)g]));
                if (row_n[(size_t)g])
                    OK(tahoe_forest_predict_raw(shard[(size_t)g], sums_d[(size_t)g], data_d[(size_t)g], row_n[(size_t)g], stream[(size_t)g]));
            }
            return;
        }
        for (int g = 0; g < G; ++g) {  // every device walks its trees over all rows
            OK(tahoe_device_set(devs[(size_t)g]));
            OK(tahoe_forest_predict_raw(shard[(size_t)g], sums_d[(size_t)g], data_d[(size_t)g], rows, stream[(size_t)g]));
            if (wide) OK(tahoe_widen_f32_to_f64(wide_d[(size_t)g], sums_d[(size_t)g], rows, stream[(size_t)g]));
        }
        if (use_rccl) {
            NCCL_OK(ncclGroupStart());
            for (int g = 0; g < G; ++g) {
                if (wide)
                    NCCL_OK(ncclAllReduce(wide_d[(size_t)g], wide_d[(size_t)g], rows, ncclFloat64, ncclSum, comms[(size_t)g],
                                          (hipStream_t)stream[(size_t)g]));
                else
                    NCCL_OK(ncclAllReduce(sums_d[(size_t)g], sums_d[(size_t)g], rows, ncclFloat32, ncclSum, comms[(size_t)g],
                                          (hipStream_t)stream[(size_t)g]));
            }
            NCCL_OK(ncclGroupEnd());
            if (wide)
                for (int g = 0; g < G; ++g) {
                    OK(tahoe_device_set(devs[(size_t)g]));
                    OK(tahoe_narrow_f64_to_f32(sums_d[(size_t)g], wide_d[(size_t)g], rows, stream[(size_t)g]));
                }
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
    for (int g = 0; g < G; ++g) OK(tahoe_forest_check(shard[(size_t)g], stream[(size_t)g]));
    if (use_rccl)  // an asynchronous failure of a collective (a peer lost, a link error) shows here, not in the sums
        for (int g = 0; g < G; ++g) {
            ncclResult_t async = ncclSuccess;
            NCCL_OK(ncclCommGetAsyncError(comms[(size_t)g], &async));
            NCCL_OK(async);
        }

    std::vector<float> total(rows, 0.f), part(rows);
    if (by_rows) {  // every device holds the finished sums of its rows
        for (int g = 0; g < G; ++g) {
            OK(tahoe_device_set(devs[(size_t)g]));
            if (row_n[(size_t)g])
                OK(tahoe_copy_to_host(total.data() + row_lo[(size_t)g], sums_d[(size_t)g], row_n[(size_t)g] * sizeof(float), stream[(size_t)g]));
        }
    } else if (chain) {  // the last device holds the result
        OK(tahoe_device_set(devs[(size_t)G - 1]));
        OK(tahoe_copy_to_host(total.data(), sums_d[(size_t)G - 1], rows * sizeof(float), stream[(size_t)G - 1]));
    } else if (emulate) {  // the all-reduce's arithmetic on the host, shard order
        std::vector<double> acc(rows, 0.0);
        for (int g = 0; g < G; ++g) {
            OK(tahoe_copy_to_host(part.data(), sums_d[(size_t)g], rows * sizeof(float), stream[(size_t)g]));
            for (size_t r = 0; r < rows; ++r) {
                if (wide)
                    acc[r] += (double)part[r];
                else
                    total[r] += part[r];
            }
        }
        if (wide)
            for (size_t r = 0; r < rows; ++r) total[r] = (float)acc[r];
    } else {
        OK(tahoe_device_set(devs[0]));
        OK(tahoe_copy_to_host(total.data(), sums_d[0], rows * sizeof(float), stream[0]));
    }
    printf("Exec.Time/Sample with %d %s shard(s) is %f us (%.3f ms per batch)\n", G, by_rows ? "row" : "tree", us / (double)R, us / 1e3);

    // ---- check.  Reference values per row: the CPU's sequential float32 sum (BaseTahoeTest.h:458-474) and the same
    // sum in float64.  chain (and G = 1): bit-identical to the float32 sum.  all-reduce modes: within the stated bound
    // of the float64 sum; the CPU float32 sum's own distance to it is printed beside ours.
    int max_shard = 0;
    for (int g = 0; g < G; ++g) max_shard = std::max(max_shard, shard_trees[(size_t)g]);
    const double u = std::ldexp(1.0, -24);
    const double n1 = (double)std::max(max_shard - 1, 0);
    const double gam = n1 * u / (1.0 - n1 * u);
    // float32 on the wire: the G partials are added in float32 too, in an order RCCL picks; every intermediate rounds a
    // prefix of the partials, so that step is bounded by gamma(G - 1) * sum_g |partial_g| <= gamma(G - 1) * sum |leaf|
    // (a bound in |exact| would reject correct runs on rows whose total nearly cancels)
    const double g1 = (double)std::max(G - 1, 0);
    const double gam_combine = wide ? 0.0 : g1 * u / (1.0 - g1 * u);
    size_t bad = 0;
    double worst = 0.0, worst_cpu = 0.0, worst_vs_cpu = 0.0;
    for (size_t r = 0; r < rows; ++r) {
        float want = 0.0f;
        double exact = 0.0, abs_sum = 0.0;
        for (int t = 0; t < T; ++t) {
            const float v = cpu_tree(nodes + (size_t)t * per_tree, data + r * (size_t)C, missing);
            want += v;
            exact += (double)v;
            abs_sum += std::fabs((double)v);
        }
        const double err = std::fabs((double)total[r] - exact);
        worst = std::fmax(worst, err);
        worst_cpu = std::fmax(worst_cpu, std::fabs((double)want - exact));
        worst_vs_cpu = std::fmax(worst_vs_cpu, std::fabs((double)total[r] - (double)want));
        if (chain || by_rows || G == 1) {
            if (memcmp(&total[r], &want, 4) != 0) ++bad;
        } else {
            const double bound = (gam + gam_combine) * (1.0 + u) * abs_sum + u * std::fabs(exact) + 1e-300;
            if (!(err <= bound)) ++bad;
        }
    }
    printf("max |ours - float64 sum| = %g; max |CPU float32 sum - float64 sum| = %g; max |ours - CPU float32 sum| = %g\n", worst,
           worst_cpu, worst_vs_cpu);
    printf(bad == 0 ? "Results are correct\n" : "Results are incorrect\n");
    for (int g = 0; g < G; ++g) {
        OK(tahoe_device_set(devs[(size_t)g]));
        if (use_rccl) NCCL_OK(ncclCommDestroy(comms[(size_t)g]));
        for (void *e : done[(size_t)g]) tahoe_event_destroy(e);
        for (void *e : taken[(size_t)g]) tahoe_event_destroy(e);
        tahoe_forest_destroy(shard[(size_t)g]);
        if (by_rows || !(emulate && g > 0)) tahoe_device_free(data_d[(size_t)g]);
        tahoe_device_free(sums_d[(size_t)g]);
        tahoe_device_free(wide_d[(size_t)g]);
        tahoe_stream_destroy(stream[(size_t)g]);
    }
    tahoe_free_host(nodes);
    tahoe_free_host(data);
    return bad == 0 ? 0 : 1;
}
