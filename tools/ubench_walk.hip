// Micro-benchmark of the LDS-resident part of a tree walk on gfx950: a 64-row feature-major tile and
// one 10-level tree top per wave in LDS; every wave repeats the 10-level walk.  Prices the per-level
// cost of different node encodings / prefetch schemes with no global-memory traffic in the loop.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_walk tools/ubench_walk.hip
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

constexpr int LEVELS = 10;
constexpr int NODES = 1 << LEVELS;  // 1-based positions 1..1023
constexpr int COLS = 256;
constexpr int ROWS = 64;
constexpr float EPS = 1.0e-6f;

__device__ __forceinline__ uint32_t go_right(float x, float thr, bool def_left, float missing)
{
    const bool is_missing = fabsf(x - missing) <= EPS;
    const bool cond = is_missing ? !def_left : (x >= thr);
    return cond ? 1u : 0u;
}

// VARIANT 0: SoA thr float | meta u16, sibling-pair prefetch (3 LDS reads per level, one round trip)
// VARIANT 1: SoA, no prefetch (read node, then feature: two round trips, 3 LDS reads)
// VARIANT 2: AoS {thr, meta32} 8 B, no prefetch (ds_read_b64 + ds_read_b32, two round trips)
// VARIANT 3: AoS 8 B, sibling pair by one ds_read_b128 (2 LDS reads per level, one round trip)
// VARIANT 4: like 3, features pre-marked: missing -> sentinel NaN, compare by bits (fewer VALU)
template <int VARIANT, int NW, int ILP>
__global__ void __launch_bounds__(NW * 64) walk_kernel(const float *__restrict__ tile_src, const uint2 *__restrict__ tree_src,
                                                       int iters, float missing, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tile = reinterpret_cast<float *>(smem);
    unsigned char *slots = smem + COLS * ROWS * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < COLS * ROWS; e += NW * 64) tile[e] = tile_src[e];
    unsigned char *slot = slots + wave * (NODES * 8);
    // both layouts are filled; a variant reads only its own
    float *s_thr = reinterpret_cast<float *>(slot);
    uint16_t *s_meta = reinterpret_cast<uint16_t *>(slot + NODES * 4);
    uint2 *s_node = reinterpret_cast<uint2 *>(slot);
    const uint2 *src = tree_src + (size_t)(blockIdx.x * NW + wave) % 64 * NODES;
    if (VARIANT <= 1) {
        for (int i = lane; i < NODES; i += 64) {
            s_thr[i] = __uint_as_float(src[i].x);
            s_meta[i] = (uint16_t)((src[i].y & 0x7fffu) | ((src[i].y >> 31) << 15));
        }
    } else {
        for (int i = lane; i < NODES; i += 64) s_node[i] = src[i];
    }
    __syncthreads();

    float acc = 0.f;
    uint32_t salt = lane * 7u;
    for (int it = 0; it < iters; ++it) {
        uint32_t i[ILP];
        uint32_t rowoff[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            i[k] = 1;
            rowoff[k] = (lane + k * 17 + salt) & 63;  // which row of the tile this chain reads
        }
        if (VARIANT == 0) {
            float thr[ILP];
            uint32_t m[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                thr[k] = s_thr[1];
                m[k] = s_meta[1];
            }
#pragma unroll 1
            for (int l = 0; l < LEVELS - 1; ++l) {
                float x[ILP];
                float2 t2[ILP];
                uint32_t m2[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    x[k] = tile[(m[k] & 0xffu) * ROWS + rowoff[k]];
                    t2[k] = *reinterpret_cast<const float2 *>(&s_thr[2 * i[k]]);
                    m2[k] = *reinterpret_cast<const uint32_t *>(&s_meta[2 * i[k]]);
                }
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    const uint32_t c = go_right(x[k], thr[k], (m[k] >> 15) != 0, missing);
                    i[k] = 2u * i[k] + c;
                    thr[k] = c ? t2[k].y : t2[k].x;
                    m[k] = c ? (m2[k] >> 16) : (m2[k] & 0xffffu);
                }
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const float x = tile[(m[k] & 0xffu) * ROWS + rowoff[k]];
                i[k] = 2u * i[k] + go_right(x, thr[k], (m[k] >> 15) != 0, missing);
            }
        } else if (VARIANT == 1) {
#pragma unroll 1
            for (int l = 0; l < LEVELS; ++l) {
                float thr[ILP];
                uint32_t m[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    thr[k] = s_thr[i[k]];
                    m[k] = s_meta[i[k]];
                }
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    const float x = tile[(m[k] & 0xffu) * ROWS + rowoff[k]];
                    i[k] = 2u * i[k] + go_right(x, thr[k], (m[k] >> 15) != 0, missing);
                }
            }
        } else if (VARIANT == 2) {
#pragma unroll 1
            for (int l = 0; l < LEVELS; ++l) {
                uint2 n[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) n[k] = s_node[i[k]];
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    const float x = tile[(n[k].y & 0xffu) * ROWS + rowoff[k]];
                    i[k] = 2u * i[k] + go_right(x, __uint_as_float(n[k].x), (n[k].y >> 31) != 0, missing);
                }
            }
        } else if (VARIANT == 3) {
            uint2 n[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) n[k] = s_node[1];
#pragma unroll 1
            for (int l = 0; l < LEVELS - 1; ++l) {
                float x[ILP];
                uint4 p[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    x[k] = tile[(n[k].y & 0xffu) * ROWS + rowoff[k]];
                    p[k] = *reinterpret_cast<const uint4 *>(&s_node[2 * i[k]]);
                }
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    const uint32_t c = go_right(x[k], __uint_as_float(n[k].x), (n[k].y >> 31) != 0, missing);
                    i[k] = 2u * i[k] + c;
                    n[k].x = c ? p[k].z : p[k].x;
                    n[k].y = c ? p[k].w : p[k].y;
                }
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const float x = tile[(n[k].y & 0xffu) * ROWS + rowoff[k]];
                i[k] = 2u * i[k] + go_right(x, __uint_as_float(n[k].x), (n[k].y >> 31) != 0, missing);
            }
        } else {
            // VARIANT 4: meta = byte offset of the feature column | def_left in bit 0; the tile holds a
            // sentinel bit pattern for "missing", so the band test is one integer compare.
            uint2 n[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) n[k] = s_node[1];
            const unsigned char *tbytes = reinterpret_cast<const unsigned char *>(tile);
#pragma unroll 1
            for (int l = 0; l < LEVELS - 1; ++l) {
                uint32_t xb[ILP];
                uint4 p[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    xb[k] = *reinterpret_cast<const uint32_t *>(tbytes + ((n[k].y & 0xffu) << 8) + rowoff[k] * 4);
                    p[k] = *reinterpret_cast<const uint4 *>(&s_node[2 * i[k]]);
                }
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    const bool ge = __uint_as_float(xb[k]) >= __uint_as_float(n[k].x);
                    const bool ms = xb[k] == 0x7fc00001u;
                    const uint32_t c = (ms ? ((n[k].y >> 31) == 0) : ge) ? 1u : 0u;
                    i[k] = 2u * i[k] + c;
                    n[k].x = c ? p[k].z : p[k].x;
                    n[k].y = c ? p[k].w : p[k].y;
                }
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const uint32_t xb = *reinterpret_cast<const uint32_t *>(tbytes + ((n[k].y & 0xffu) << 8) + rowoff[k] * 4);
                const bool ge = __uint_as_float(xb) >= __uint_as_float(n[k].x);
                const bool ms = xb == 0x7fc00001u;
                i[k] = 2u * i[k] + ((ms ? ((n[k].y >> 31) == 0) : ge) ? 1u : 0u);
            }
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            acc += (float)i[k];
            salt += i[k];
        }
    }
    out[blockIdx.x * NW * 64 + tid] = acc;
}

template <int VARIANT, int NW, int ILP>
static void run(const float *tile, const uint2 *trees, float *out, int blocks_per_cu)
{
    const int lds = COLS * ROWS * 4 + NW * NODES * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&walk_kernel<VARIANT, NW, ILP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 400 / ILP;
    const int grid = 256 * blocks_per_cu;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((walk_kernel<VARIANT, NW, ILP>), dim3(grid), dim3(NW * 64), lds, 0, tile, trees, iters, -999.0f, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double wave_walks_per_cu = (double)blocks_per_cu * NW * iters * ILP;
    const double clk = ms * 1e-3 * 2.4e9 / wave_walks_per_cu;
    printf("variant %d  waves/WG %2d  ILP %d  WG/CU %d : %7.3f ms  %6.1f clk per wave-walk per CU = %5.1f clk per level\n", VARIANT,
           NW, ILP, blocks_per_cu, ms, clk, clk / LEVELS);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main()
{
    std::vector<float> h_tile(COLS * ROWS);
    uint64_t s = 99;
    auto rnd = [&]() {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        return (uint32_t)(s >> 33);
    };
    for (auto &v : h_tile) v = (float)(rnd() & 0xffffff) / 8388608.0f - 1.0f;
    std::vector<uint2> h_trees(64 * NODES);
    for (auto &n : h_trees) {
        const float thr = (float)(rnd() & 0xffffff) / 8388608.0f - 1.0f;
        n.x = *reinterpret_cast<const uint32_t *>(&thr);
        n.y = (rnd() & 0xffu) | ((rnd() & 1u) << 31);
    }
    float *d_tile, *d_out;
    uint2 *d_trees;
    CHECK(hipMalloc(&d_tile, h_tile.size() * 4));
    CHECK(hipMalloc(&d_trees, h_trees.size() * 8));
    CHECK(hipMalloc(&d_out, 256 * 2 * 16 * 64 * 4));
    CHECK(hipMemcpy(d_tile, h_tile.data(), h_tile.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_trees, h_trees.data(), h_trees.size() * 8, hipMemcpyHostToDevice));

    run<0, 8, 1>(d_tile, d_trees, d_out, 1);
    run<1, 8, 1>(d_tile, d_trees, d_out, 1);
    run<2, 8, 1>(d_tile, d_trees, d_out, 1);
    run<3, 8, 1>(d_tile, d_trees, d_out, 1);
    run<4, 8, 1>(d_tile, d_trees, d_out, 1);
    run<0, 8, 2>(d_tile, d_trees, d_out, 1);
    run<2, 8, 2>(d_tile, d_trees, d_out, 1);
    run<3, 8, 2>(d_tile, d_trees, d_out, 1);
    run<4, 8, 2>(d_tile, d_trees, d_out, 1);
    run<3, 8, 4>(d_tile, d_trees, d_out, 1);
    run<4, 8, 4>(d_tile, d_trees, d_out, 1);
    run<2, 4, 1>(d_tile, d_trees, d_out, 1);
    run<3, 4, 2>(d_tile, d_trees, d_out, 1);
    run<3, 4, 4>(d_tile, d_trees, d_out, 1);
    run<4, 4, 4>(d_tile, d_trees, d_out, 1);
    run<3, 12, 1>(d_tile, d_trees, d_out, 1);
    run<4, 12, 1>(d_tile, d_trees, d_out, 1);
    run<4, 12, 2>(d_tile, d_trees, d_out, 1);
    return 0;
}
