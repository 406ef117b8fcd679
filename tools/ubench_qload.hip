// Micro-benchmark of the memory side of the quantise pre-pass on gfx950 (tahoe_amd/csrc/quantize.hip) with the threshold
// search left out: 1 Mi rows x 256 float32 features (row-major, 1 KiB per row) in, one u16 per value out in the walk
// kernel's region layout xq[row / 64][feature][row % 64].  What does each way of assigning (rows, features) to a
// workgroup cost the texture path?
//   pair   workgroup = 2 features x 65536 rows, a lane reads 8 bytes of one row          (the product kernel's shape)
//   quad   workgroup = 4 features x 65536 rows, a lane reads 16 bytes of one row
//   line   workgroup = 16 features x 16384 rows, 4 adjacent lanes read one 64-byte piece of a row
//   rows   workgroup = all features x 64-row groups, a wave reads whole rows (the streaming bound; codes through LDS)
//   tpose  row-major -> pair-major float2 copy through LDS (the explicit transposition), into a full-size buffer or
//          into one 64 MiB slab that is overwritten again and again (does the Infinity Cache keep it?)
//   cpair  pair-major float2 in (coalesced), codes out
// LDS bytes per workgroup are a parameter (the product kernel holds 144 KiB of tables: one workgroup per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_qload tools/ubench_qload.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

constexpr int COLS = 256;
constexpr size_t ROWS = 1u << 20;
constexpr int THREADS = 1024;

__device__ __forceinline__ uint16_t fake_code(float x) { return (uint16_t)(int)(x * 16383.0f); }
__device__ __forceinline__ size_t xq_index(size_t r, int f) { return (r >> 6) * (size_t)(COLS * 64) + (size_t)f * 64 + (r & 63); }

// V floats per lane (2: pair, 4: quad); workgroup = V features x 2^cshift rows; rows loaded one iteration ahead, U per thread
typedef float nf2 __attribute__((ext_vector_type(2)));
typedef float nf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 ld(const float2 *p, bool nt)
{
    if (!nt) return *p;
    const nf2 v = __builtin_nontemporal_load(reinterpret_cast<const nf2 *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 ld(const float4 *p, bool nt)
{
    if (!nt) return *p;
    const nf4 v = __builtin_nontemporal_load(reinterpret_cast<const nf4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int V, int U = 4, bool NT = false>
__global__ void __launch_bounds__(THREADS) feat_kernel(const float *__restrict__ data, uint16_t *__restrict__ xq, int cshift)
{
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    using vec = typename std::conditional<V == 2, float2, float4>::type;
    const int groups = COLS / V;
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;  // XCD-contiguous, as the product
    const int f0 = (int)(vid % groups) * V;
    const size_t chunk = vid / groups;
    const size_t r0 = chunk << cshift, r1 = r0 + ((size_t)1 << cshift);
    vec nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r = min(r0 + threadIdx.x + (size_t)u * THREADS, r1 - 1);
        nx[u] = ld(reinterpret_cast<const vec *>(data + r * COLS + f0), NT);
    }
    for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)THREADS * U) {
        vec xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = nx[u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = min(rb + (size_t)(U + u) * THREADS, r1 - 1);
            nx[u] = ld(reinterpret_cast<const vec *>(data + r * COLS + f0), NT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = rb + (size_t)u * THREADS;
            if (r < r1) {
                const float *x = reinterpret_cast<const float *>(&xv[u]);
#pragma unroll
                for (int j = 0; j < V; ++j) xq[xq_index(r, f0 + j)] = fake_code(x[j]);
            }
        }
    }
}

// 16 features per workgroup: lane -> (row = i / 4, 16-byte piece = i % 4): 64-byte pieces of 16 rows per wave-instruction
__global__ void __launch_bounds__(THREADS) line_kernel(const float *__restrict__ data, uint16_t *__restrict__ xq, int cshift)
{
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    constexpr int U = 4, RPI = THREADS / 4;
    const int groups = COLS / 16;
    const unsigned nblk = gridDim.x;
    const unsigned vid = (nblk % 8u == 0u) ? (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u : blockIdx.x;
    const int f0 = (int)(vid % groups) * 16 + 4 * (threadIdx.x & 3);
    const size_t chunk = vid / groups;
    const size_t r0 = chunk << cshift, r1 = r0 + ((size_t)1 << cshift);
    const int rsub = threadIdx.x >> 2;
    float4 nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nx[u] = *reinterpret_cast<const float4 *>(data + min(r0 + rsub + (size_t)u * RPI, r1 - 1) * COLS + f0);
    for (size_t rb = r0 + rsub; rb < r1; rb += (size_t)RPI * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = nx[u];
#pragma unroll
        for (int u = 0; u < U; ++u) nx[u] = *reinterpret_cast<const float4 *>(data + min(rb + (size_t)(U + u) * RPI, r1 - 1) * COLS + f0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = rb + (size_t)u * RPI;
            if (r < r1) {
                const float *x = reinterpret_cast<const float *>(&xv[u]);
#pragma unroll
                for (int j = 0; j < 4; ++j) xq[xq_index(r, f0 + j)] = fake_code(x[j]);
            }
        }
    }
}

// whole rows: a wave converts 64-row groups; lane l holds features 4l..4l+3 of one row per load; codes go through LDS
// ([feature][row] u16, 32 KiB per wave-group) and leave as the contiguous 32 KiB region
__global__ void __launch_bounds__(256) rows_kernel(const float *__restrict__ data, uint16_t *__restrict__ xq, size_t groups64)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *t = reinterpret_cast<uint16_t *>(smem) + (size_t)wave * COLS * 64;
    for (size_t g = (size_t)blockIdx.x * 4 + wave; g < groups64; g += (size_t)gridDim.x * 4) {
        const float *src = data + g * 64 * COLS + 4 * lane;
#pragma unroll 4
        for (int r = 0; r < 64; r += 4) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4 *>(src + (size_t)(r + j) * COLS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[(4 * lane + 0) * 64 + r + j] = fake_code(v[j].x);
                t[(4 * lane + 1) * 64 + r + j] = fake_code(v[j].y);
                t[(4 * lane + 2) * 64 + r + j] = fake_code(v[j].z);
                t[(4 * lane + 3) * 64 + r + j] = fake_code(v[j].w);
            }
        }
        uint4 *dst = reinterpret_cast<uint4 *>(xq + g * (size_t)(COLS * 64));
        const uint4 *s4 = reinterpret_cast<const uint4 *>(t);
#pragma unroll 4
        for (int i = lane; i < COLS * 64 * 2 / 16; i += 64) dst[i] = s4[i];
    }
}

// transposition: rows [g*64, g*64+64) x 256 features -> out[(pair * slab_rows + row_in_slab)] float2; one wave per 64-row group
__global__ void __launch_bounds__(192) tpose_kernel(const float *__restrict__ data, float2 *__restrict__ out, size_t groups64,
                                                    size_t slab_rows)
{
    __shared__ float tile[3][64 * 65];  // one quarter (64 features) of a 64-row group at a time, padded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    for (size_t g = (size_t)blockIdx.x * 3 + wave; g < groups64; g += (size_t)gridDim.x * 3) {
        const size_t row0 = g * 64, srow0 = row0 % slab_rows;
        for (int q = 0; q < 4; ++q) {  // 64 features per pass
            // load: lane -> (row = i / 16, float4 piece = i % 16): 4 rows x 256 B per instruction
            const float *src = data + row0 * COLS + q * 64 + 4 * (lane & 15);
#pragma unroll 4
            for (int r = lane >> 4; r < 64; r += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)r * COLS);
                float *d = t + r * 65 + 4 * (lane & 15);
                d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
            }
            // store: for each pair of this quarter, 64 rows x float2 = 512 contiguous bytes
#pragma unroll 4
            for (int p = 0; p < 32; ++p) {
                const float2 v = make_float2(t[lane * 65 + 2 * p], t[lane * 65 + 2 * p + 1]);
                out[(size_t)(q * 32 + p) * slab_rows + srow0 + lane] = v;
            }
        }
    }
}

// pair-major in (coalesced), codes out: workgroup = pair x 2^cshift rows
__global__ void __launch_bounds__(THREADS) cpair_kernel(const float2 *__restrict__ in, uint16_t *__restrict__ xq, int cshift, size_t slab_rows,
                                                        size_t slab_row0)
{
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    constexpr int U = 4;
    const int pair = blockIdx.x % (COLS / 2);
    const size_t chunk = blockIdx.x / (COLS / 2);
    const size_t r0 = chunk << cshift, r1 = r0 + ((size_t)1 << cshift);
    const float2 *src = in + (size_t)pair * slab_rows;
    for (size_t rb = r0 + threadIdx.x; rb < r1; rb += (size_t)THREADS * U) {
        float2 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = src[min(rb + (size_t)u * THREADS, r1 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t r = rb + (size_t)u * THREADS;
            if (r < r1) {
                xq[xq_index(slab_row0 + r, 2 * pair)] = fake_code(xv[u].x);
                xq[xq_index(slab_row0 + r, 2 * pair + 1)] = fake_code(xv[u].y);
            }
        }
    }
}

template <typename F>
static float time_ms(F launch, int reps = 10)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    float *data;
    uint16_t *xq;
    float2 *tp;
    CHECK(hipMalloc(&data, ROWS * COLS * sizeof(float)));
    CHECK(hipMalloc(&xq, ROWS * COLS * sizeof(uint16_t)));
    CHECK(hipMalloc(&tp, ROWS * COLS * sizeof(float)));
    {
        std::vector<float> h(ROWS * COLS / 16);
        uint32_t s = 12345;
        for (auto &v : h) v = (float)((s = s * 1664525u + 1013904223u) >> 8) / 16777216.0f;
        for (int i = 0; i < 16; ++i) CHECK(hipMemcpy(data + (size_t)i * h.size(), h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    const double gb = (ROWS * COLS * 6.0) / 1e9;
    for (int lds : {147456})
        for (int cshift : {16, 15}) {
            CHECK(hipFuncSetAttribute((const void *)&feat_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            CHECK(hipFuncSetAttribute((const void *)&feat_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            CHECK(hipFuncSetAttribute((const void *)&line_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const unsigned chunks = (unsigned)(ROWS >> cshift);
            float ms = time_ms([&] { hipLaunchKernelGGL(feat_kernel<2>, dim3(chunks * COLS / 2), dim3(THREADS), lds, 0, data, xq, cshift); });
            printf("pair  lds %6d rows/wg 2^%d : %.3f ms  %.2f TB/s\n", lds, cshift, ms, gb / ms);
            ms = time_ms([&] { hipLaunchKernelGGL(feat_kernel<4>, dim3(chunks * COLS / 4), dim3(THREADS), lds, 0, data, xq, cshift); });
            printf("quad  lds %6d rows/wg 2^%d : %.3f ms  %.2f TB/s\n", lds, cshift, ms, gb / ms);
            ms = time_ms([&] { hipLaunchKernelGGL(line_kernel, dim3(chunks * COLS / 16), dim3(THREADS), lds, 0, data, xq, cshift); });
            printf("line  lds %6d rows/wg 2^%d : %.3f ms  %.2f TB/s\n", lds, cshift, ms, gb / ms);
        }
    {
        const int lds = 147456;
        auto run = [&](const char *name, auto kern, int v, int cshift) {
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const unsigned chunks = (unsigned)(ROWS >> cshift);
            float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(chunks * COLS / v), dim3(THREADS), lds, 0, data, xq, cshift); });
            printf("%s rows/wg 2^%d : %.3f ms\n", name, cshift, ms);
        };
        for (int cs : {15, 16}) {
            run("pair U=8      ", &feat_kernel<2, 8, false>, 2, cs);
            run("pair U=4 nt   ", &feat_kernel<2, 4, true>, 2, cs);
            run("pair U=8 nt   ", &feat_kernel<2, 8, true>, 2, cs);
            run("pair U=2      ", &feat_kernel<2, 2, false>, 2, cs);
            run("quad U=8      ", &feat_kernel<4, 8, false>, 4, cs);
            run("quad U=4 nt   ", &feat_kernel<4, 4, true>, 4, cs);
        }
    }
    for (int grid : {256, 512, 1024, 2048}) {
        CHECK(hipFuncSetAttribute((const void *)&rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * COLS * 64 * 2));
        float ms = time_ms([&] { hipLaunchKernelGGL(rows_kernel, dim3(grid), dim3(256), 4 * COLS * 64 * 2, 0, data, xq, ROWS / 64); });
        printf("rows  grid %4d : %.3f ms  %.2f TB/s\n", grid, ms, gb / ms);
    }
    for (int grid : {512, 1024, 2048, 4096}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(tpose_kernel, dim3(grid), dim3(192), 0, 0, data, tp, ROWS / 64, ROWS); });
        printf("tpose full buffer grid %4d : %.3f ms  %.2f TB/s (read + write)\n", grid, ms, ROWS * COLS * 8.0 / 1e9 / ms);
    }
    for (size_t slab : {(size_t)16384, (size_t)65536, (size_t)131072}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(tpose_kernel, dim3(2048), dim3(192), 0, 0, data, tp, ROWS / 64, slab); });
        printf("tpose slab %6zu rows (%3zu MiB, overwritten) : %.3f ms\n", slab, slab * COLS * 4 >> 20, ms);
    }
    for (int lds : {147456, 1024}) {
        CHECK(hipFuncSetAttribute((const void *)&cpair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        float ms = time_ms([&] { hipLaunchKernelGGL(cpair_kernel, dim3(16 * COLS / 2), dim3(THREADS), lds, 0, tp, xq, 16, ROWS, (size_t)0); });
        printf("cpair lds %6d : %.3f ms  %.2f TB/s\n", lds, ms, gb / ms);
    }
    // slabs: transpose a slab into the reused buffer, convert it, next slab (two launches per slab, one stream)
    for (size_t slab : {(size_t)65536, (size_t)131072, (size_t)262144}) {
        CHECK(hipFuncSetAttribute((const void *)&cpair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
        const int cs = slab >= 131072 ? 16 : 15;
        float ms = time_ms([&] {
            for (size_t s0 = 0; s0 < ROWS; s0 += slab) {
                hipLaunchKernelGGL(tpose_kernel, dim3(1024), dim3(192), 0, 0, data + s0 * COLS, tp, slab / 64, slab);
                hipLaunchKernelGGL(cpair_kernel, dim3((unsigned)(slab >> cs) * COLS / 2), dim3(THREADS), 147456, 0, tp, xq, cs, slab, s0);
            }
        });
        printf("slabs of %6zu rows: tpose + cpair per slab : %.3f ms\n", slab, ms);
    }
    return 0;
}
