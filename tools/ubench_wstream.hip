// Micro-benchmark behind the row-streaming wide form (widef.hip, wstream_kernel): how fast can ONE persistent workgroup per CU
// pull whole float32 rows (K2: 3072 features = 12 KiB) through a small ring of LDS row slots by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs), with one loader wave and NW consumer waves that do a
// chain of dependent LDS reads per row before they release the slot?  Swept: row slots S, DMA pieces in flight Q, cache
// policy (default / nt), consumer work.  A verify mode checksums every row (also proves that LDS-DMA reaches LDS addresses
// above 64 KiB).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_wstream tools/ubench_wstream.hip && tools/ubench_wstream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ uint32_t fl(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void fs(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The loader's own flag accesses: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see while an LDS-DMA is in
// flight (the DMA is a pending LDS write), which drains the stream once per row; inline asm is invisible to that pass.
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }  // low half of a generic LDS address
__device__ __forceinline__ uint32_t afl(const uint32_t *p)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void afs(uint32_t *p, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory"); }

template <int Q>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory");
}

constexpr int kSpin = 1 << 24;

// LDS: [lds_off bytes unused (stands for the resident tops)] [S row slots of cols * 4 bytes] [flags]
template <int Q, int AUX, bool VERIFY>
__global__ void __launch_bounds__(1024) stream_kernel(const float *__restrict__ data, size_t rows, int cols, int S, int lds_off, int nwalk,
                                                      int nch, int work, uint32_t *__restrict__ out, int *__restrict__ err, int nload, int slp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_bytes = cols * 4;
    const int P = row_bytes / 1024;  // DMA pieces per row
    float *slots = reinterpret_cast<float *>(smem + lds_off);
    uint32_t *landed = reinterpret_cast<uint32_t *>(smem + lds_off + (size_t)S * row_bytes);
    uint32_t *walked = landed + 1;  // [S] monotone: chunk-walks finished in the slot
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per;
    const int n = r0 < rows ? (int)((rows - r0 < per) ? rows - r0 : per) : 0;
    if (tid == 0) *landed = 0u;
    if (tid < S) walked[tid] = 0u;
    __syncthreads();
    if (nch == 0) {
        // ---- free run: nload loader waves, wave l loads rows l, l + nload, ... into slot (row % S); nobody reads ----
        if (wave >= nload) return;
        for (int k = wave; k < n; k += nload) {
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + (r0 + k) * (size_t)cols) + lane * 16;
            unsigned char *dst = reinterpret_cast<unsigned char *>(slots) + (size_t)(k % S) * row_bytes;
            for (int p = 0; p < P; ++p) {
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + p * 1024),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, AUX);
                wait_vm<Q>();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (wave == 0) {
        // ---- loader: keeps up to Q pieces in flight; publishes the count of landed PIECES once per row (a wave issues one
        // instruction per ~5 clk: any per-piece bookkeeping beyond the load itself costs more than the load) ----
        int issued = 0, pub = 0;
        for (int k = 0; k < n; ++k) {
            if (k >= S) {
                const uint32_t need = (uint32_t)nch * (uint32_t)(k / S);
                if (afl(&walked[k % S]) < need) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // never block with unpublished pieces in flight
                    pub = issued;
                    afs(landed, (uint32_t)pub);
                    int spins = 0;
                    while (afl(&walked[k % S]) < need) {
                        if (++spins > kSpin) {
                            if (lane == 0) atomicOr(err, 1);
                            return;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + (r0 + k) * (size_t)cols) + lane * 16;
            unsigned char *dst = reinterpret_cast<unsigned char *>(slots) + (size_t)(k % S) * row_bytes;
            for (int p = 0; p < P; ++p)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + p * 1024),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, AUX);
            issued += P;
            wait_vm<Q>();  // all but the Q youngest pieces have landed
            if (issued - Q > pub) {
                pub = issued - Q;
                afs(landed, (uint32_t)pub);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        afs(landed, (uint32_t)issued);
        return;
    }
    if (wave > nwalk) return;
    // ---- consumers: item i = (row i / nch, chunk i % nch) ----
    const int w = wave - 1;
    for (int i = w; i < n * nch; i += nwalk) {
        const int k = i / nch, c = i - k * nch;
        int spins = 0;
        while (fl(landed) < (uint32_t)((k + 1) * P)) {
            if (++spins > kSpin) {
                if (lane == 0) atomicOr(err, 2);
                return;
            }
            if (slp == 1) __builtin_amdgcn_s_sleep(1);
            else if (slp == 8) __builtin_amdgcn_s_sleep(8);
            else if (slp == 32) __builtin_amdgcn_s_sleep(32);
            else __builtin_amdgcn_s_sleep(127);
        }
        asm volatile("" ::: "memory");
        const float *xr = slots + (size_t)(k % S) * cols;
        if (VERIFY) {
            if (c == 0) {  // checksum of the whole row
                uint32_t s = 0;
                for (int e = lane; e < cols; e += 64) s += __float_as_uint(xr[e]) * (uint32_t)(e + 1);
                for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
                if (lane == 0) out[r0 + k] = s;
            }
        } else {
            uint32_t idx = (uint32_t)(lane * 37 + c * 101) % (uint32_t)cols;
            uint32_t acc = 0;
            for (int l = 0; l < work; ++l) {  // dependent random reads of the row, like a walk
                const uint32_t v = __float_as_uint(xr[idx]);
                acc += v;
                idx = (idx * 1664525u + (v >> 9) + 1013904223u) % (uint32_t)cols;
            }
            if (acc == 0x12345678u) out[r0 + k] = acc;  // keeps the chain alive
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&walked[k % S], 1u);
    }
}

template <int Q, int AUX>
static float run(const float *d, size_t rows, int cols, int S, int lds_off, int nwalk, int nch, int work, uint32_t *out, int *err, int grid,
                 bool verify, int nload = 1, int slp = 1)
{
    const int lds = lds_off + S * cols * 4 + 256;
    auto kt = stream_kernel<Q, AUX, false>;
    auto kv = stream_kernel<Q, AUX, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int threads = (nwalk + 1) * 64;
    if (verify) {
        hipLaunchKernelGGL(kv, dim3(grid), dim3(threads), lds, 0, d, rows, cols, S, lds_off, nwalk, nch, work, out, err, nload, slp);
        CK(hipDeviceSynchronize());
        return 0.f;
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kt, dim3(grid), dim3(threads), lds, 0, d, rows, cols, S, lds_off, nwalk, nch, work, out, err, nload, slp);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kt, dim3(grid), dim3(threads), lds, 0, d, rows, cols, S, lds_off, nwalk, nch, work, out, err, nload, slp);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t rows = 100000;
    const int cols = 3072;
    std::vector<float> h(rows * cols);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        v = (float)((s >> 40) * (1.0 / 16777216.0)) * 2.f - 1.f;
    }
    float *d;
    uint32_t *out;
    int *err;
    CK(hipMalloc(&d, h.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, rows * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    const double gb = rows * (double)cols * 4 / 1e9;
    // verify: slots above 64 KiB of LDS
    {
        run<16, 0>(d, rows, cols, 4, 96 * 1024, 12, 8, 0, out, err, grid, true);
        std::vector<uint32_t> got(rows);
        CK(hipMemcpy(got.data(), out, rows * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t r = 0; r < rows; ++r) {
            uint32_t want = 0;
            for (int e = 0; e < cols; ++e) {
                uint32_t u;
                memcpy(&u, &h[r * cols + e], 4);
                want += u * (uint32_t)(e + 1);
            }
            bad += want != got[r];
        }
        int e;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("verify (slots at LDS offset 96 KiB, S 4, Q 16): %zu of %zu rows wrong, err flag %d\n", bad, rows, e);
    }
    printf("rows %zu x %d cols = %.3f GB, grid %d persistent workgroups; time per pass, GB/s\n", rows, cols, gb, grid);
    auto line = [&](const char *tag, int S, int off, int nw, int nch, int work, float ms) {
        printf("%-10s S %d lds_off %3d KiB walkers %2d chunks %d work %2d : %.3f ms  %.0f GB/s\n", tag, S, off / 1024, nw, nch, work, ms, gb / ms * 1e3);
        fflush(stdout);
    };
    for (int work : {0}) {
        for (int S : {4, 8}) {
            const int off = S <= 4 ? 96 * 1024 : 48 * 1024;
            line("Q8 dflt", S, off, 12, 8, work, run<8, 0>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
            line("Q16 dflt", S, off, 12, 8, work, run<16, 0>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
            line("Q24 dflt", S, off, 12, 8, work, run<24, 0>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
            line("Q32 dflt", S, off, 12, 8, work, run<32, 0>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
            line("Q16 nt", S, off, 12, 8, work, run<16, 2>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
            line("Q32 nt", S, off, 12, 8, work, run<32, 2>(d, rows, cols, S, off, 12, 8, work, out, err, grid, false));
        }
    }
    // fewer chunks per row (a consumer wave per row): the hand-over rate alone
    line("Q24 dflt", 4, 96 * 1024, 4, 1, 0, run<24, 0>(d, rows, cols, 4, 96 * 1024, 4, 1, 0, out, err, grid, false));
    for (int slp : {1, 8, 32, 127})
        for (int nw : {1, 4, 12}) {
            printf("consumer sleep %3d, %2d consumer waves (1 chunk per row), S 8 Q 32: %.3f ms\n", slp, nw,
                   run<32, 0>(d, rows, cols, 8, 0, nw, 1, 0, out, err, grid, false, 1, slp));
            fflush(stdout);
        }
    for (int nl : {1, 4}) {
        printf("free run, %d loader wave(s), S 8: Q16 %.3f ms  Q32 %.3f ms  Q48 %.3f ms  Q63 %.3f ms  Q32 nt %.3f ms\n", nl,
               run<16, 0>(d, rows, cols, 8, 0, 8, 0, 0, out, err, grid, false, nl), run<32, 0>(d, rows, cols, 8, 0, 8, 0, 0, out, err, grid, false, nl),
               run<48, 0>(d, rows, cols, 8, 0, 8, 0, 0, out, err, grid, false, nl), run<63, 0>(d, rows, cols, 8, 0, 8, 0, 0, out, err, grid, false, nl),
               run<32, 2>(d, rows, cols, 8, 0, 8, 0, 0, out, err, grid, false, nl));
        fflush(stdout);
    }
    int e;
    CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("err flag %d\n", e);
    return 0;
}
