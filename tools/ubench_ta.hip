// Micro-benchmark of the vector-memory side of the QRING walk on gfx950 (tahoe_amd/csrc/qring.hip), with the LDS walk
// and its VALU left out: per (tree, 128-row tile) a walker wave stages the tree's top with S coalesced 16-byte loads
// per lane (S = 4: 4 KiB, the 10-level top) and fetches, for each of its two 64-row chains, G adjacent 16-byte pieces
// of a randomly chosen bottom block.  256 workgroups of NW waves walk the trees in the same order (as the real kernel:
// the active trees stay L2-resident).  Prices design alternatives by what they cost the texture path:
//   S=4 G=2  the round-1 kernel (10-level top, 32-byte blocks: 3 nodes + 4 leaves)
//   S=8 G=1  11-level top, 16-byte blocks (1 node + 2 leaves)
//   S=2 G=4  9-level top, 64-byte blocks
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ta tools/ubench_ta.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

constexpr int TREES = 1000;
constexpr int BLOCK_BYTES_PER_TREE = 32768;

template <int NW, int S, int G, int K, bool WRITE_LDS>
__global__ void __launch_bounds__(NW * 64) ta_kernel(const uint4 *__restrict__ tops, const uint4 *__restrict__ blocks, int tiles,
                                                     uint32_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 *slot = reinterpret_cast<uint4 *>(smem + (size_t)wave * S * 1024);
    uint32_t acc = 0;
    uint32_t h = (blockIdx.x * 977u + tid) * 2654435761u;
    for (int tile = 0; tile < tiles; ++tile) {
        for (int t = wave; t < TREES; t += NW) {
            const uint4 *g = tops + (size_t)t * (S * 64);
            uint4 pf[S > 0 ? S : 1];
#pragma unroll
            for (int s = 0; s < S; ++s) pf[s] = g[s * 64 + lane];
            uint4 v[K][G > 0 ? G : 1];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                h = h * 1664525u + 1013904223u;
                const uint32_t b = (h >> 8) % (BLOCK_BYTES_PER_TREE / (16 * G));
                const uint4 *bp = blocks + ((size_t)t * (BLOCK_BYTES_PER_TREE / 16) + (size_t)b * G);
#pragma unroll
                for (int j = 0; j < G; ++j) v[k][j] = bp[j];
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (WRITE_LDS)
                    slot[s * 64 + lane] = pf[s];
                else
                    acc += pf[s].x ^ pf[s].w;
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < G; ++j) acc += v[k][j].x + v[k][j].w;
            if (WRITE_LDS) acc += reinterpret_cast<uint32_t *>(slot)[(lane * 17 + t) & (S * 256 - 1)];
        }
    }
    out[(size_t)blockIdx.x * NW * 64 + tid] = acc;
}

template <int NW, int S, int G, int K, bool WRITE_LDS>
static void run(const uint4 *tops, const uint4 *blocks, uint32_t *out, const char *what)
{
    const int lds = NW * S * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ta_kernel<NW, S, G, K, WRITE_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int tiles = 4;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((ta_kernel<NW, S, G, K, WRITE_LDS>), dim3(256), dim3(NW * 64), lds, 0, tops, blocks, tiles, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double units = (double)tiles * TREES;  // (tree, tile) units per CU
    const double clk = ms * 1e-3 * 2.4e9 / units;
    // K3: 7813 tiles of 128 rows (64 * K rows here) x 1000 trees over 256 CUs
    const double k3_ms = clk * (1.0e6 / (64.0 * K)) * 1000.0 / 256.0 / 2.4e9 * 1e3;
    printf("NW %2d S %d G %d K %d lds-write %d : %7.3f ms  %6.1f clk per (tree, %3d-row tile) per CU -> K3 %.2f ms   [%s]\n", NW, S, G, K,
           (int)WRITE_LDS, ms, clk, 64 * K, k3_ms, what);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main()
{
    const size_t top_bytes = (size_t)TREES * 8192, blk_bytes = (size_t)TREES * BLOCK_BYTES_PER_TREE;
    std::vector<uint32_t> h(blk_bytes / 4);
    uint64_t s = 7;
    for (auto &v : h) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        v = (uint32_t)(s >> 33);
    }
    uint4 *d_tops, *d_blocks;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_tops, top_bytes));
    CHECK(hipMalloc(&d_blocks, blk_bytes));
    CHECK(hipMalloc(&d_out, 256 * 16 * 64 * 4));
    CHECK(hipMemcpy(d_tops, h.data(), top_bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_blocks, h.data(), blk_bytes, hipMemcpyHostToDevice));

    run<15, 4, 2, 2, true>(d_tops, d_blocks, d_out, "round-1 kernel: 4 KiB top, 32-B blocks");
    run<15, 4, 2, 2, false>(d_tops, d_blocks, d_out, "same, no LDS writes");
    run<15, 4, 1, 2, true>(d_tops, d_blocks, d_out, "4 KiB top, one 16-B piece per chain (the 'no leaf gather' ablation)");
    run<15, 0, 2, 2, false>(d_tops, d_blocks, d_out, "gathers only");
    run<15, 0, 1, 2, false>(d_tops, d_blocks, d_out, "one-piece gathers only");
    run<15, 4, 0, 2, true>(d_tops, d_blocks, d_out, "staging only, 4 KiB");
    run<15, 8, 0, 2, true>(d_tops, d_blocks, d_out, "staging only, 8 KiB");
    run<10, 8, 1, 2, true>(d_tops, d_blocks, d_out, "11-level top (8 KiB), 16-B blocks, 10 walkers");
    run<15, 8, 1, 2, true>(d_tops, d_blocks, d_out, "11-level top (8 KiB), 16-B blocks, 15 walkers (LDS would not fit)");
    run<15, 2, 4, 2, true>(d_tops, d_blocks, d_out, "9-level top (2 KiB), 64-B blocks");
    run<13, 4, 2, 3, true>(d_tops, d_blocks, d_out, "192-row tile: 4 KiB top, 32-B blocks, 13 walkers");
    run<7, 4, 2, 4, true>(d_tops, d_blocks, d_out, "256-row tile: 4 KiB top, 32-B blocks, 7 walkers");
    run<8, 8, 1, 3, true>(d_tops, d_blocks, d_out, "192-row tile: 8 KiB top, 16-B blocks, 8 walkers");
    return 0;
}
