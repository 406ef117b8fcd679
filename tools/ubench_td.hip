// What a divergent gather costs the texture path of one CU, by shape: bytes per lane (4 / 8 / 16), active lanes (64 / 16 / 4 / 1),
// the span of memory the lanes' addresses fall into (32 KiB: one K3 tree's bottom blocks; 1 MiB: K2's blocks; 32 MiB), and whether
// a second load re-reads the lines of the first.  Every wave keeps several independent gathers in flight (no dependent chain), 16
// waves per CU, one workgroup per CU: the figure is throughput, cycles per wave-instruction per CU at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_td tools/ubench_td.hip && tools/ubench_td
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int BYTES>
struct vec_of;
template <>
struct vec_of<4> {
    typedef uint32_t type;
};
template <>
struct vec_of<8> {
    typedef uint2 type;
};
template <>
struct vec_of<16> {
    typedef uint4 type;
};
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// MODE 0: random 32-byte-aligned record per lane; MODE 1: a second load of the next 16 bytes of the same record (same line);
// MODE 2: coalesced (lane-consecutive) loads for reference
template <int BYTES, int MODE>
__global__ void __launch_bounds__(1024) td_kernel(const unsigned char *__restrict__ buf, uint32_t span_records, int active, int iters,
                                                  uint32_t *__restrict__ out)
{
    typedef typename vec_of<BYTES>::type V;
    const int lane = threadIdx.x & 63;
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const bool on = lane < active;
    for (int it = 0; it < iters; ++it) {
        V v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four independent gathers in flight per wave
            h = h * 1664525u + 1013904223u;
            const uint32_t rec = MODE == 2 ? ((h >> 8) % (span_records / 64)) * 64 + lane : (h >> 8) % span_records;
            const unsigned char *p = buf + (size_t)rec * 32;
            if (on) {
                v[u] = *reinterpret_cast<const V *>(p);
                if (MODE == 1) w[u] = *reinterpret_cast<const V *>(p + 16);
            } else {
                v[u] = V{};
                w[u] = V{};
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc += fold(v[u]);
            if (MODE == 1) acc += fold(w[u]);
        }
        if (MODE == 2) h = __builtin_amdgcn_readfirstlane(h);  // keep the wave on one record group
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int BYTES, int MODE>
static void run(const unsigned char *buf, size_t span_bytes, int active, uint32_t *out, int cus, const char *what)
{
    const int iters = 400;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((td_kernel<BYTES, MODE>), dim3(cus), dim3(1024), 0, 0, buf, (uint32_t)(span_bytes / 32), active, iters, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    const double instr = 16.0 * iters * 4 * (MODE == 1 ? 2 : 1);  // wave-instructions per CU
    printf("%-34s %2d B/lane, %2d lanes, span %8zu KiB : %7.1f clk per wave-load per CU   (%.3f ms)\n", what, BYTES, active, span_bytes >> 10,
           ms * 1e-3 * 2.4e9 / instr, ms);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)32 << 20;
    std::vector<uint32_t> h(bytes / 4);
    uint64_t s = 99;
    for (auto &v : h) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        v = (uint32_t)(s >> 33);
    }
    unsigned char *d;
    uint32_t *out;
    CK(hipMalloc(&d, bytes));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)cus * 1024 * 4));
    for (size_t span : {(size_t)32 << 10, (size_t)1 << 20, (size_t)32 << 20}) {
        run<16, 2>(d, span, 64, out, cus, "coalesced 1-KiB load");
        run<16, 0>(d, span, 64, out, cus, "divergent gather");
        run<8, 0>(d, span, 64, out, cus, "divergent gather");
        run<4, 0>(d, span, 64, out, cus, "divergent gather");
        run<16, 1>(d, span, 64, out, cus, "gather + second piece of the line");
        run<16, 0>(d, span, 16, out, cus, "divergent gather, fewer lanes");
        run<16, 0>(d, span, 4, out, cus, "divergent gather, fewer lanes");
        run<4, 0>(d, span, 1, out, cus, "divergent gather, one lane");
    }
    return 0;
}
