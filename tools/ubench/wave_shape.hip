// micro-benchmark (round 6): the bf16 GEMM k loop is LDS-bandwidth bound -- a 64-deep stage of a 256 x 128 tile moves 48 KB INTO LDS (LDS-DMA) and 8 waves x 16 KB of
// fragments OUT of it (each wave's 64 x 64 block: (64 + 64) rows x 128 B) = 176 KB against 128 B/clk = 1375 cycles, where the matrix instructions of the stage need 1024
// per SIMD.  Fewer, FATTER waves read less: 4 waves of 128 x 64 read (128 + 64) x 128 B = 24 KB each = 96 KB, 144 KB per stage with the DMA = 1125 cycles.
// This program times the loop's ingredients for both wave shapes at the same work per CU (same DMA bytes, same number of matrix instructions per stage):
//   A: 8 waves x { 6 pieces, 16 ds_read_b128, 32 mfma_16x16x32_bf16 }      (ppo_bf16.hpp's gb_mainloop today)
//   B: 4 waves x { 12 pieces, 24 ds_read_b128, 64 mfma }                    one wave per SIMD
//   C: B with the second k-step's 12 reads issued before the first k-step's 32 matrix instructions (two fragment sets)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/wave_shape tools/ubench/wave_shape.hip ; run: tools/ubench/wave_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const char* src, char* dst_wave_uniform) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)dst_wave_uniform, 16, 0, 0);
}
// WAVES = 8 (MA = 4 row blocks per wave) or 4 (MA = 8); three 48 KB ring buffers; stage s + 2 requested at the head of stage s; 128-byte rows at 2 KB pitch
template <int WAVES, bool PIPE>
__global__ __launch_bounds__(64 * WAVES) void k(const char* __restrict__ base, int stages, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int MA = 32 / WAVES, NP = 48 / WAVES;                 // row blocks of 16 per wave; pieces per wave and stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* mine = base;                                        // every workgroup of an XCD walks the same window: it stays in L2
    const size_t lane_off = (size_t)(lane >> 3) * 2048 + (size_t)(lane & 7) * 16;
    f32x4 acc[MA][4];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int s) __attribute__((always_inline)) {
        char* buf = lds + (s % 3) * 49152;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int piece = wave + WAVES * q;
            dma16(mine + (size_t)(((s * 48 + piece) * 8) % 512) * 2048 + lane_off, buf + piece * 1024);
        }
    };
    const int wr = (WAVES == 8 ? (wave >> 1) * 64 : (wave >> 1) * 128), wc = (wave & 1) * 64;
    auto fragA = [&](const char* buf, int a, int ks) __attribute__((always_inline)) {
        const int r = wr + 16 * a + (lane & 15), c = 4 * ks + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(buf + r * 128 + ((c ^ (r & 7)) * 16));
    };
    auto fragB = [&](const char* buf, int b, int ks) __attribute__((always_inline)) {
        const int r = wc + 16 * b + (lane & 15), c = 4 * ks + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(buf + 32768 + r * 128 + ((c ^ (r & 7)) * 16));
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    stage(0); stage(1);
    for (int s = 0; s < stages; ++s) {
        if (NP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage(s + 2);
        const char* buf = lds + (s % 3) * 49152;
        bf16x8 af[2][MA], bf[2][4];
        if (!PIPE) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int a = 0; a < MA; ++a) af[ks][a] = fragA(buf, a, ks);
#pragma unroll
                for (int b = 0; b < 4; ++b) bf[ks][b] = fragB(buf, b, ks);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < MA; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][b], af[ks][a], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int a = 0; a < MA; ++a) af[0][a] = fragA(buf, a, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[0][b] = fragB(buf, b, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MA; ++a) af[1][a] = fragA(buf, a, 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[1][b] = fragB(buf, b, 1);
#pragma unroll
            for (int a = 0; a < MA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][b], af[0][a], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][b], af[1][a], acc[a][b], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) keep += acc[a][b][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (keep == 123.456f) cyc[256] = 1;
}

int main() {
    const size_t bytes = 64ull << 20;
    char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 258 * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int stages = 2048;
    auto run = [&](auto kern, int threads, const char* name) -> int {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 3 * 49152, 0, d, stages, cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        std::vector<unsigned long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("%-64s %7.1f us  %.3f us per stage  median %6.0f cycles per stage  (matrix instructions alone: 1024)\n", name, best * 1e3, best * 1e3 / stages, (double)h[128] / stages);
        return 0;
    };
    run(k<8, false>, 512, "A: 8 waves x (6 DMA, 16 reads, 32 MFMA) = today's loop");
    run(k<4, false>, 256, "B: 4 waves x (12 DMA, 24 reads, 64 MFMA)");
    run(k<4, true>, 256, "C: B, second k-step's reads under the first's MFMAs");
    run(k<8, true>, 512, "D: A with the k-step pipeline");
    return 0;
}
