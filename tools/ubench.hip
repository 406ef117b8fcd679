// Micro-benchmarks that price the primitives the traversal kernels are built from on gfx950:
// dependent per-lane gathers from global memory (by footprint, width, chains in flight) and from LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int W>
struct Vec;
template <>
struct Vec<1> { using T = uint32_t; };
template <>
struct Vec<2> { using T = uint2; };
template <>
struct Vec<4> { using T = uint4; };

__device__ __forceinline__ uint32_t first(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t first(uint2 v) { return v.x; }
__device__ __forceinline__ uint32_t first(uint4 v) { return v.x; }

// Each lane runs ILP dependent chains: next index = f(loaded value).  mask = elements - 1.
template <int W, int ILP>
__global__ void __launch_bounds__(256) gather_global(const typename Vec<W>::T *__restrict__ table, uint32_t mask,
                                                     int iters, uint32_t *out)
{
    uint32_t idx[ILP];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ILP; ++k) idx[k] = (tid * 2654435761u + k * 40503u) & mask;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const uint32_t v = first(table[idx[k]]);
            idx[k] = (idx[k] * 2u + 1u + v) & mask;  // value-dependent, tree-walk-like
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc += idx[k];
    out[tid] = acc;
}

// Same from LDS: table of `elems` W-dword records staged once, then dependent gathers.
template <int W, int ILP>
__global__ void __launch_bounds__(256) gather_lds(const typename Vec<W>::T *__restrict__ table, uint32_t mask, int iters,
                                                  uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using T = typename Vec<W>::T;
    T *lds = reinterpret_cast<T *>(smem);
    for (uint32_t i = threadIdx.x; i <= mask; i += blockDim.x) lds[i] = table[i];
    __syncthreads();
    uint32_t idx[ILP];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ILP; ++k) idx[k] = (tid * 2654435761u + k * 40503u) & mask;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const uint32_t v = first(lds[idx[k]]);
            idx[k] = (idx[k] * 2u + 1u + v) & mask;
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc += idx[k];
    out[tid] = acc;
}

static double time_ms(hipEvent_t a, hipEvent_t b)
{
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

template <int W, int ILP>
static void run_global(const void *table, size_t bytes, int blocks_per_cu, uint32_t *out, hipEvent_t e0, hipEvent_t e1)
{
    const uint32_t elems = (uint32_t)(bytes / (4 * W));
    const int iters = 2000 / ILP;
    const int grid = 256 * blocks_per_cu;
    using T = typename Vec<W>::T;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_global<W, ILP>), dim3(grid), dim3(256), 0, 0, (const T *)table, elems - 1, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    const double ms = time_ms(e0, e1);
    const double wave_loads_per_cu = (double)blocks_per_cu * 4 * iters * ILP;
    printf("global W=%d ILP=%d foot=%8zu KB waves/CU=%2d : %7.3f ms  %6.1f ns/wave-load/CU  (%5.1f clk @2.4GHz)\n", W, ILP,
           bytes >> 10, blocks_per_cu * 4, ms, ms * 1e6 / wave_loads_per_cu, ms * 1e6 / wave_loads_per_cu * 2.4);
}

template <int W, int ILP>
static void run_lds(const void *table, size_t bytes, int blocks_per_cu, uint32_t *out, hipEvent_t e0, hipEvent_t e1)
{
    const uint32_t elems = (uint32_t)(bytes / (4 * W));
    const int iters = 20000 / ILP;
    const int grid = 256 * blocks_per_cu;
    using T = typename Vec<W>::T;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_lds<W, ILP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_lds<W, ILP>), dim3(grid), dim3(256), bytes, 0, (const T *)table, elems - 1, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    const double ms = time_ms(e0, e1);
    const double wave_loads_per_cu = (double)blocks_per_cu * 4 * iters * ILP;
    printf("lds    W=%d ILP=%d foot=%8zu KB waves/CU=%2d : %7.3f ms  %6.1f ns/wave-load/CU  (%5.1f clk @2.4GHz)\n", W, ILP,
           bytes >> 10, blocks_per_cu * 4, ms, ms * 1e6 / wave_loads_per_cu, ms * 1e6 / wave_loads_per_cu * 2.4);
}

int main()
{
    const size_t max_bytes = 64u << 20;
    std::vector<uint32_t> h(max_bytes / 4);
    uint64_t s = 12345;
    for (auto &v : h) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        v = (uint32_t)(s >> 33);
    }
    void *table;
    uint32_t *out;
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    const size_t foots[] = {1u << 10, 16u << 10, 1u << 20, 64u << 20};
    for (size_t f : foots) {
        for (int bpc : {2, 4}) {
            run_global<1, 1>(table, f, bpc, out, e0, e1);
            run_global<2, 1>(table, f, bpc, out, e0, e1);
            run_global<4, 1>(table, f, bpc, out, e0, e1);
            run_global<2, 2>(table, f, bpc, out, e0, e1);
            run_global<2, 4>(table, f, bpc, out, e0, e1);
            run_global<4, 4>(table, f, bpc, out, e0, e1);
        }
    }
    for (int bpc : {2, 4}) {
        run_lds<1, 1>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<2, 1>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<1, 2>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<2, 2>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<1, 4>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<2, 4>(table, 32u << 10, bpc, out, e0, e1);
        run_lds<2, 8>(table, 32u << 10, bpc, out, e0, e1);
    }
    return 0;
}
