// Micro-benchmark of the LDS-resident part of the QRING walk on gfx950 (tahoe_amd/csrc/qring.hip): a u16 code tile
// [256 cols][64*K rows] and one 10-level quantised top (4 KiB, u32 nodes, child pairs by ds_read_b64) per walker wave
// in LDS; every wave repeats the 10-level walk with K independent 64-row chains.  No global-memory traffic in the
// loop: prices the walk proper as a function of walker waves, chains per lane and loop form, and separates the LDS
// and VALU ceilings (MODE 1: the LDS reads only, MODE 2: the VALU only).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_qwalk tools/ubench_qwalk.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
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
constexpr int COLS = 256;
typedef const uint16_t __attribute__((address_space(3))) *lds_u16_ptr;
typedef const uint64_t __attribute__((address_space(3))) *lds_u64_ptr;

__device__ __forceinline__ uint32_t descend(uint32_t i, uint64_t m)
{
    uint32_t r;
    uint64_t co;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=&s"(co) : "v"(i), "s"(m));
    return r;
}

// Column stride 256 B (128-row regions): node = code << 16 | fid << 8, address = bfi(0xFF00, node, pos).
// Column stride 128 B (64-row regions, one region per chain): node = code << 16 | fid << 7, address = bfi(0x7F80, node, pos)
// with the region's base folded into pos.  COLB = 0: general form, 2 VALU (bfe + mad) with 384-byte columns.
template <int COLB>
__device__ __forceinline__ uint32_t xaddr(uint32_t node, uint32_t posb)
{
    uint32_t a;
    if (COLB == 256) {
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "s"(0xFF00u), "v"(node), "v"(posb));
    } else if (COLB == 128) {
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "s"(0x7F80u), "v"(node), "v"(posb));
    } else {
        asm("v_bfe_u32 %0, %1, 8, 8\n\tv_mad_u32_u24 %0, %0, %3, %2" : "=&v"(a) : "v"(node), "v"(posb), "s"(384u));
    }
    return a;
}

template <int NW, int K, int MODE, bool UNROLL, int COLB, bool DEP>
__global__ void __launch_bounds__(NW * 64) qwalk(const uint16_t *__restrict__ tile_src, const uint32_t *__restrict__ top_src, int iters,
                                                 uint32_t *__restrict__ out)
{
    constexpr int TILEB = COLS * 128 * K;  // tile bytes: 64 * K rows of u16 per column
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *tile = reinterpret_cast<uint16_t *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < TILEB / 2; e += NW * 64) tile[e] = tile_src[e % (COLS * 128)];
    uint32_t *slot = reinterpret_cast<uint32_t *>(smem + TILEB + wave * 4096);
    const uint32_t *src = top_src + (size_t)((blockIdx.x * NW + wave) % 64) * 1024;
    for (int i = lane; i < 1024; i += 64) slot[i] = src[i];
    __syncthreads();
    const uint32_t slot_a = (uint32_t)reinterpret_cast<uintptr_t>(slot);
    uint32_t pos[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        pos[k] = COLB == 128 ? (uint32_t)(k * COLS * 128) + (uint32_t)(((lane & 31) << 2) | ((lane >> 5) << 1))
                         : 2u * (uint32_t)(k * 64 + lane);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t i[K], node[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            i[k] = 1;
            node[k] = slot[1] ^ ((acc & 0xffu) << 8);  // data-dependent start: the iterations stay ordered
        }
        auto level_dep = [&]() {  // read only the chosen child, after the compare: 4 VALU + 2 LDS, two round trips
            uint32_t xc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) xc[k] = *reinterpret_cast<lds_u16_ptr>(xaddr<COLB>(node[k], pos[k]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                i[k] = descend(i[k], __builtin_amdgcn_uicmp(xc[k], node[k] >> 16, 35));
                node[k] = *reinterpret_cast<const uint32_t __attribute__((address_space(3))) *>(slot_a + 4u * i[k]);
            }
        };
        auto level = [&]() {
            if (DEP) {
                level_dep();
                return;
            }
            uint32_t xc[K];
            uint2 pr[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t xa = xaddr<COLB>(node[k], pos[k]);
                const uint32_t pa = slot_a + 8u * i[k];
                if (MODE != 2) {
                    xc[k] = *reinterpret_cast<lds_u16_ptr>(xa);
                    {
                        const uint64_t w = *reinterpret_cast<lds_u64_ptr>(pa);
                        pr[k] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                    }
                } else {
                    xc[k] = xa >> 3;
                    pr[k] = make_uint2(pa * 0x9E3779B1u, xa * 0x85EBCA6Bu);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 1) {  // LDS only: the next addresses do not depend on a compare
                    i[k] = ((i[k] * 2u) | (xc[k] & 1u)) & 511u;
                    node[k] = pr[k].x;
                } else {
                    const uint64_t cm = __builtin_amdgcn_uicmp(xc[k], node[k] >> 16, 35);
                    i[k] = descend(i[k], cm);
                    node[k] = __builtin_amdgcn_inverse_ballot_w64(cm) ? pr[k].y : pr[k].x;
                }
            }
        };
        if (UNROLL) {
#pragma unroll
            for (int l = 0; l < LEVELS - 1; ++l) level();
        } else {
#pragma unroll 1
            for (int l = 0; l < LEVELS - 1; ++l) level();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t xcl = (MODE != 2) ? (uint32_t)*reinterpret_cast<lds_u16_ptr>(xaddr<COLB>(node[k], pos[k])) : node[k] >> 5;
            i[k] = descend(i[k], __builtin_amdgcn_uicmp(xcl, node[k] >> 16, 35));
            acc += i[k];
        }
    }
    out[(size_t)blockIdx.x * NW * 64 + tid] = acc;
}

template <int NW, int K, int MODE, bool UNROLL, int COLB = (K == 2 ? 256 : 0), bool DEP = false>
static void run(const uint16_t *tile, const uint32_t *tops, uint32_t *out)
{
    const int lds = COLS * 128 * K + NW * 4096;
    if (lds > 160 * 1024) {
        printf("NW %2d K %d: %d B of LDS do not fit\n", NW, K, lds);
        return;
    }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&qwalk<NW, K, MODE, UNROLL, COLB, DEP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 600 / K;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((qwalk<NW, K, MODE, UNROLL, COLB, DEP>), dim3(256), dim3(NW * 64), lds, 0, tile, tops, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double wave_levels_per_cu = (double)NW * iters * K * LEVELS;
    const double clk = ms * 1e-3 * 2.4e9 / wave_levels_per_cu;
    printf("NW %2d K %d colb %3d %s mode %s %s : %7.3f ms  %5.2f clk per wave-level per CU  (K3 walk of 12 levels at this rate: %.2f ms)\n", NW, K, COLB,
           DEP ? "dep " : "pair", MODE == 0 ? "full" : MODE == 1 ? "lds " : "valu", UNROLL ? "unrolled" : "loop    ", ms, clk, clk * 1.875e8 / 256 / 2.4e9 * 1e3);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

// --kernel-shape: only the shape the product's K3 walk runs (14 walker waves x 3 chains on 64-row regions, child pairs,
// unrolled), best of 5 launches, as one JSON line: bench.py runs this as a child process for roofline.physical.walk.lds_walk_ceiling.
template <int NW, int K, bool DEP>
static float best_ms(const uint16_t *tile, const uint32_t *tops, uint32_t *out, int cus, int iters)
{
    const int lds = COLS * 128 * K + NW * 4096;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&qwalk<NW, K, 0, true, 128, DEP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        float ms = 0;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((qwalk<NW, K, 0, true, 128, DEP>), dim3(cus), dim3(NW * 64), lds, 0, tile, tops, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}
// Both forms of the level (both children beside the feature read: 5 VALU + 2 LDS; only the chosen child after the compare: 4 + 2);
// the ceiling is the faster one.
template <int NW, int K>
static void run_json(const uint16_t *tile, const uint32_t *tops, uint32_t *out)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount < 256 ? prop.multiProcessorCount : 256;  // (the output buffer holds 256 workgroups)
    const int iters = 2400 / K;
    const float pair = best_ms<NW, K, false>(tile, tops, out, cus, iters), dep = best_ms<NW, K, true>(tile, tops, out, cus, iters);
    const float best = pair < dep ? pair : dep;
    const double wave_levels_per_cu = (double)NW * iters * K * LEVELS;  // one workgroup per CU
    printf("{\"nw\": %d, \"k\": %d, \"cus\": %d, \"ms\": %.4f, \"ms_pair_reads\": %.4f, \"ms_chosen_child\": %.4f, \"wave_levels_per_cu\": %.0f, "
           "\"ns_per_wave_level_per_cu\": %.5f, \"clock_mhz\": %d}\n", NW, K, cus, best, pair, dep, wave_levels_per_cu,
           best * 1e6 / wave_levels_per_cu, prop.clockRate / 1000);
}

int main(int argc, char **argv)
{
    uint64_t s = 99;
    auto rnd = [&]() {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        return (uint32_t)(s >> 33);
    };
    std::vector<uint16_t> h_tile(COLS * 128);
    for (auto &v : h_tile) v = (uint16_t)(rnd() % 16000);
    std::vector<uint32_t> h_tops(64 * 1024);
    for (auto &n : h_tops) n = ((rnd() % 16000 + 1) << 16) | ((rnd() & 0xffu) << 8) | (rnd() & 1u);
    uint16_t *d_tile;
    uint32_t *d_tops, *d_out;
    CHECK(hipMalloc(&d_tile, h_tile.size() * 2));
    CHECK(hipMalloc(&d_tops, h_tops.size() * 4));
    CHECK(hipMalloc(&d_out, 256 * 16 * 64 * 4));
    CHECK(hipMemcpy(d_tile, h_tile.data(), h_tile.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tops, h_tops.data(), h_tops.size() * 4, hipMemcpyHostToDevice));
    if (argc > 1 && std::string(argv[1]) == "--kernel-shape") {
        run_json<14, 3>(d_tile, d_tops, d_out);
        return 0;
    }

    run<15, 2, 0, false>(d_tile, d_tops, d_out);
    run<15, 2, 0, true>(d_tile, d_tops, d_out);
    run<15, 2, 0, false, 256, true>(d_tile, d_tops, d_out);
    run<15, 2, 0, true, 256, true>(d_tile, d_tops, d_out);
    run<15, 2, 0, true, 128, false>(d_tile, d_tops, d_out);
    run<13, 3, 0, false, 128, false>(d_tile, d_tops, d_out);
    run<13, 3, 0, true, 128, false>(d_tile, d_tops, d_out);
    run<13, 3, 0, false, 128, true>(d_tile, d_tops, d_out);
    run<13, 3, 0, true, 128, true>(d_tile, d_tops, d_out);
    run<14, 3, 0, true, 128, false>(d_tile, d_tops, d_out);
    run<7, 4, 0, true, 128, false>(d_tile, d_tops, d_out);
    run<7, 4, 0, true, 128, true>(d_tile, d_tops, d_out);
    run<15, 2, 1, false>(d_tile, d_tops, d_out);
    run<15, 2, 2, false>(d_tile, d_tops, d_out);
    run<15, 1, 0, false>(d_tile, d_tops, d_out);
    run<16, 2, 0, false>(d_tile, d_tops, d_out);
    run<12, 2, 0, false>(d_tile, d_tops, d_out);
    run<8, 2, 0, false>(d_tile, d_tops, d_out);
    run<8, 2, 0, true>(d_tile, d_tops, d_out);
    run<4, 2, 0, true>(d_tile, d_tops, d_out);
    run<13, 3, 0, false>(d_tile, d_tops, d_out);
    run<13, 3, 0, true>(d_tile, d_tops, d_out);
    run<13, 3, 1, false>(d_tile, d_tops, d_out);
    run<13, 3, 2, false>(d_tile, d_tops, d_out);
    run<8, 3, 0, true>(d_tile, d_tops, d_out);
    run<7, 4, 0, false>(d_tile, d_tops, d_out);
    run<7, 4, 0, true>(d_tile, d_tops, d_out);
    run<4, 4, 0, true>(d_tile, d_tops, d_out);
    return 0;
}
