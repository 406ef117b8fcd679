// Host-resident batches: the step either side of the hot path.
//
// The reference uploads the whole data file once (generate_data_from_file, BaseTahoeTest.h:378-389:
// cudaMalloc + one cudaMemcpyAsync) and times only resident-data predicts.  A caller whose rows live in host
// memory pays that copy on every batch, and over PCIe the copy, not the traversal, is the long pole (K3: 1 GB of
// rows, ~20 ms on the link against 5.4 ms of traversal).  This file overlaps the two: rows go up in chunks through
// two device buffers on a copy stream while the previous chunk is traversed on a compute stream, and predictions
// come back the same way.  Pageable source memory is staged through two pinned buffers by a few host threads;
// a source that is already pinned (hipHostMalloc / hipHostRegister) is copied from directly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "common.h"
#include "forest_internal.h"

struct tahoe_pstate {
    size_t chunk_rows = 0;
    int cols = 0;
    float *d_data[2] = {nullptr, nullptr};
    float *d_preds[2] = {nullptr, nullptr};
    float *h_stage[2] = {nullptr, nullptr};  // pinned, for pageable sources
    float *h_preds[2] = {nullptr, nullptr};  // pinned
    hipStream_t copy = nullptr, compute = nullptr, back = nullptr;
    hipEvent_t uploaded[2] = {nullptr, nullptr};   // copy stream: chunk is in d_data[slot]
    hipEvent_t traversed[2] = {nullptr, nullptr};  // compute stream: d_data[slot] may be overwritten, d_preds[slot] is ready
    hipEvent_t returned[2] = {nullptr, nullptr};   // back stream: h_preds[slot] holds the chunk's predictions
};

namespace tahoe {

static void pipe_free(tahoe_pstate *ps)
{
    for (int s = 0; s < 2; ++s) {
        if (ps->d_data[s]) (void)hipFree(ps->d_data[s]);
        if (ps->d_preds[s]) (void)hipFree(ps->d_preds[s]);
        if (ps->h_stage[s]) (void)hipHostFree(ps->h_stage[s]);
        if (ps->h_preds[s]) (void)hipHostFree(ps->h_preds[s]);
        if (ps->uploaded[s]) (void)hipEventDestroy(ps->uploaded[s]);
        if (ps->traversed[s]) (void)hipEventDestroy(ps->traversed[s]);
        if (ps->returned[s]) (void)hipEventDestroy(ps->returned[s]);
    }
    if (ps->copy) (void)hipStreamDestroy(ps->copy);
    if (ps->compute) (void)hipStreamDestroy(ps->compute);
    if (ps->back) (void)hipStreamDestroy(ps->back);
    delete ps;
}

void pipeline_destroy(tahoe_forest *f)
{
    if (f->pipe) pipe_free(f->pipe);
    f->pipe = nullptr;
}

// Chunks of about 32 MiB of rows: long enough that a chunk's copy runs at link speed and its traversal fills the
// chip (>= 8192 rows = 64 tiles... at least a few waves per CU), short enough that the un-overlapped first upload
// and last traversal are a small part of the batch.  Multiples of 32768 rows (QRING's quantisation chunk).
static size_t auto_chunk_rows(const tahoe_forest *f, size_t rows)
{
    const size_t row_bytes = std::max<size_t>((size_t)f->p.num_cols * sizeof(float), 4);
    size_t r = (32u << 20) / row_bytes;
    r = std::max<size_t>(r / 32768 * 32768, 32768);
    return std::min(r, std::max<size_t>(rows, 1));
}

static tahoe_status pipe_prepare(tahoe_forest *f, size_t chunk_rows, bool need_stage)
{
    tahoe_pstate *ps = f->pipe;
    if (ps && (ps->chunk_rows < chunk_rows || ps->cols != f->p.num_cols)) {
        TAHOE_HIP_TRY(hipDeviceSynchronize());
        pipe_free(ps);
        f->pipe = ps = nullptr;
    }
    if (!ps) {
        // Built in a local state and published only when complete: a failed allocation half-way frees what was
        // made and leaves f->pipe null, so the next call starts over instead of reusing null buffers.
        ps = new (std::nothrow) tahoe_pstate();
        if (!ps) return fail(TAHOE_ERR_NO_MEMORY, "tahoe_forest_predict_host");
        ps->chunk_rows = chunk_rows;
        ps->cols = f->p.num_cols;
        const size_t dbytes = std::max<size_t>(chunk_rows * (size_t)f->p.num_cols * sizeof(float), 4);
        const size_t pbytes = std::max<size_t>(chunk_rows * sizeof(float), 4);
        hipError_t e = hipSuccess;
        auto ok = [&e](hipError_t r) { return e == hipSuccess && (e = r) == hipSuccess; };
        for (int s = 0; s < 2 && e == hipSuccess; ++s) {
            (void)(ok(hipMalloc(reinterpret_cast<void **>(&ps->d_data[s]), dbytes)) &&
                   ok(hipMalloc(reinterpret_cast<void **>(&ps->d_preds[s]), pbytes)) &&
                   ok(hipHostMalloc(reinterpret_cast<void **>(&ps->h_preds[s]), pbytes, hipHostMallocDefault)) &&
                   ok(hipEventCreateWithFlags(&ps->uploaded[s], hipEventDisableTiming)) &&
                   ok(hipEventCreateWithFlags(&ps->traversed[s], hipEventDisableTiming)) &&
                   ok(hipEventCreateWithFlags(&ps->returned[s], hipEventDisableTiming)));
        }
        (void)(ok(hipStreamCreateWithFlags(&ps->copy, hipStreamNonBlocking)) &&
               ok(hipStreamCreateWithFlags(&ps->compute, hipStreamNonBlocking)) &&
               ok(hipStreamCreateWithFlags(&ps->back, hipStreamNonBlocking)));
        if (e != hipSuccess) {
            pipe_free(ps);
            return fail(TAHOE_ERR_HIP, "tahoe_forest_predict_host: preparing the upload pipeline failed: %s", hipGetErrorString(e));
        }
        // the quantised workspace is sized once for a chunk: no allocation inside the loop
        const tahoe_status st = qring_reserve(f, chunk_rows);
        if (st != TAHOE_OK) {
            pipe_free(ps);
            return st;
        }
        f->pipe = ps;
    }
    if (need_stage && !ps->h_stage[0]) {
        const size_t dbytes = std::max<size_t>(ps->chunk_rows * (size_t)f->p.num_cols * sizeof(float), 4);
        for (int s = 0; s < 2; ++s) {
            const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ps->h_stage[s]), dbytes, hipHostMallocDefault);
            if (e != hipSuccess) {  // all or nothing: a lone first buffer would read as "staging is ready"
                if (ps->h_stage[0]) (void)hipHostFree(ps->h_stage[0]);
                ps->h_stage[0] = ps->h_stage[1] = nullptr;
                return fail(TAHOE_ERR_HIP, "tahoe_forest_predict_host: hipHostMalloc(stage) failed: %s", hipGetErrorString(e));
            }
        }
    }
    return TAHOE_OK;
}

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// memcpy split over a few threads: one core moves ~10 GB/s, the link takes ~50
static void staged_copy(float *dst, const float *src, size_t bytes)
{
    const size_t kMin = 4u << 20;
    unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(bytes / kMin, 1), 8);
    nt = std::min(nt, std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t part = (bytes / nt + 63) & ~(size_t)63;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = std::min(bytes, t * part), hi = std::min(bytes, lo + part);
        if (hi > lo)
            th.emplace_back([=] { std::memcpy(reinterpret_cast<char *>(dst) + lo, reinterpret_cast<const char *>(src) + lo, hi - lo); });
    }
    for (std::thread &t : th) t.join();
}

}  // namespace tahoe

using namespace tahoe;

extern "C" tahoe_status tahoe_forest_predict_host(tahoe_forest *f, float *preds_host, const float *data_host, size_t rows,
                                                  size_t chunk_rows)
{
    if (!f || (rows && (!preds_host || !data_host)))
        return fail(TAHOE_ERR_INVALID_ARG, "tahoe_forest_predict_host: null argument");
    if (rows == 0) return TAHOE_OK;
    if (chunk_rows == 0) chunk_rows = auto_chunk_rows(f, rows);
    chunk_rows = std::min(chunk_rows, rows);
    TAHOE_HIP_TRY(hipSetDevice(f->device));
    const bool pinned = is_pinned(data_host);
    tahoe_status st = pipe_prepare(f, chunk_rows, !pinned);
    if (st != TAHOE_OK) return st;
    tahoe_pstate *ps = f->pipe;
    const size_t cols = (size_t)f->p.num_cols;
    const size_t n_chunks = (rows + chunk_rows - 1) / chunk_rows;

    // Chunk i uses slot i & 1.  Per chunk: [host: stage] -> copy stream: H2D -> compute stream: traverse ->
    // back stream: D2H into pinned -> [host: memcpy out, one chunk late].
    auto collect = [&](size_t i) -> tahoe_status {  // predictions of chunk i: pinned -> caller
        const int s = (int)(i & 1);
        TAHOE_HIP_TRY(hipEventSynchronize(ps->returned[s]));
        const size_t lo = i * chunk_rows, n = std::min(chunk_rows, rows - lo);
        std::memcpy(preds_host + lo, ps->h_preds[s], n * sizeof(float));
        return TAHOE_OK;
    };
    for (size_t i = 0; i < n_chunks; ++i) {
        const int s = (int)(i & 1);
        const size_t lo = i * chunk_rows, n = std::min(chunk_rows, rows - lo);
        const size_t bytes = n * cols * sizeof(float);
        const float *src = data_host + lo * cols;
        if (i >= 2) {
            // slot reuse: chunk i-2's predictions must have left h_preds[s] (collect), which also implies its
            // traversal is over, so d_data[s], d_preds[s] and h_stage[s] are free
            if ((st = collect(i - 2)) != TAHOE_OK) return st;
        }
        if (!pinned && bytes) {
            staged_copy(ps->h_stage[s], src, bytes);
            src = ps->h_stage[s];
        }
        if (bytes) TAHOE_HIP_TRY(hipMemcpyAsync(ps->d_data[s], src, bytes, hipMemcpyHostToDevice, ps->copy));
        TAHOE_HIP_TRY(hipEventRecord(ps->uploaded[s], ps->copy));
        TAHOE_HIP_TRY(hipStreamWaitEvent(ps->compute, ps->uploaded[s], 0));
        if ((st = tahoe_forest_predict(f, ps->d_preds[s], ps->d_data[s], n, ps->compute)) != TAHOE_OK) return st;
        TAHOE_HIP_TRY(hipEventRecord(ps->traversed[s], ps->compute));
        TAHOE_HIP_TRY(hipStreamWaitEvent(ps->back, ps->traversed[s], 0));
        TAHOE_HIP_TRY(hipMemcpyAsync(ps->h_preds[s], ps->d_preds[s], n * sizeof(float), hipMemcpyDeviceToHost, ps->back));
        TAHOE_HIP_TRY(hipEventRecord(ps->returned[s], ps->back));
    }
    for (size_t i = n_chunks >= 2 ? n_chunks - 2 : 0; i < n_chunks; ++i)
        if ((st = collect(i)) != TAHOE_OK) return st;
    return tahoe_forest_check(f, ps->compute);
}
