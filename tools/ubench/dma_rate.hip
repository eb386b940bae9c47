// micro-benchmark: what a CU pulls from L2 into LDS with LDS-DMA (global_load_lds_dwordx4), per row width / row pitch / sharing.
// 256 workgroups x 8 waves (one per CU); a wave keeps two 6-piece stages in flight (counted vmcnt) over a window that stays in L2.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dma_rate tools/ubench/dma_rate.hip ; run: tools/ubench/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void dma16(const char* src, char* dst_wave_uniform) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)dst_wave_uniform, 16, 0, 0);
}

// ROWB: contiguous bytes per row taken by a piece (64 lanes x 16 B = 1 KB = 1024 / ROWB rows); PITCH: bytes between rows;
// window: rows of the region a workgroup walks; shared: every workgroup of an XCD reads the same region (like weights) or its own
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// BAR: one s_barrier per stage (all 8 waves in step, as in the GEMM loops); RD: every wave reads 16 x 16 bytes per lane of the ring per stage
// (the fragment reads of a 64 x 64 wave block); MF: 32 v_mfma_f32_16x16x32_bf16 per wave and stage on what was read; PP: two barriers per stage,
// waves 0-3 read while waves 4-7 multiply and vice versa (ping-pong)
template <int ROWB, bool BAR = false, bool RD = false, bool MF = false, bool PP = false>
__global__ __launch_bounds__(512) void k(const char* __restrict__ base, int pitch, int window_rows, size_t wg_stride, int stages, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = ROWB / 16, RPP = 1024 / ROWB, NPF = 6;
    const char* mine = base + (size_t)blockIdx.x * wg_stride;
    const size_t lane_off = (size_t)(lane / LPR) * pitch + (size_t)(lane % LPR) * 16;
    char* ring = lds + wave * 2 * NPF * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    int row = wave * RPP * NPF;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 sink = make_uint4(0, 0, 0, 0);
    uint4 f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            dma16(mine + (size_t)((row + q * RPP) % window_rows) * pitch + lane_off, ring + ((s & 1) * NPF + q) * 1024);
        }
        row += 8 * RPP * NPF;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        auto reads = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = *reinterpret_cast<const uint4*>(lds + ((wave * 12 + ((s & 1) ^ 1) * 6 + (r % 6)) * 1024 + ((lane * 16 + (r / 6) * 256) & 1023)));
        };
        auto mma = [&]() __attribute__((always_inline)) {
            if (MF) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[8 * ks + a]), __builtin_bit_cast(bf16x8, f[8 * ks + 4 + b]), acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { sink.x ^= f[r].x; sink.y ^= f[r].y; sink.z ^= f[r].z; sink.w ^= f[r].w; }
            }
        };
        if (PP) {
            if (wave < 4) {
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                reads();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                mma();
            } else {
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                if (s > 0) mma();
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                reads();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            if (BAR) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
            if (RD) { reads(); mma(); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) keep += acc[a][b][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (keep == 123.456f || sink.x == 0x12345678u) cyc[blockIdx.x + 1] = sink.y;      // keeps the reads / the products alive
}

int main() {
    const size_t bytes = 512ull << 20;
    char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 258 * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int stages = 2048;                                 // per wave: 2048 x 6 KB ; per workgroup 96 MB (about a millisecond: clocks settle)
    auto run = [&](auto kern, const char* name, int rowb, int pitch, int window_rows, size_t wg_stride) -> int {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 96 * 1024, 0, d, pitch, window_rows, wg_stride, stages, cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double kb_per_wg = 8.0 * stages * 6;           // KB
        printf("%-38s row %4d B pitch %5d window %5d rows (%6.0f KB) %s: %7.1f us  median %8llu cycles  %5.1f B/clk/CU  (%.1f TB/s chip)\n", name, rowb, pitch, window_rows,
               window_rows * (double)pitch / 1024, wg_stride ? "own " : "same", ms * 1e3, h[128], kb_per_wg * 1024 / (double)h[128], 256 * kb_per_wg * 1024 / (ms * 1e-3) / 1e12);
        return 0;
    };
    // contiguous 1 KB pieces, window small enough for L2 (per XCD: 32 workgroups)
    run(k<1024>, "contiguous", 1024, 1024, 96, 0);
    run(k<1024>, "contiguous", 1024, 1024, 96, 96 * 1024);
    run(k<256>, "256 B rows, 1 KB pitch", 256, 1024, 512, 0);
    run(k<256>, "256 B rows, 1 KB pitch", 256, 1024, 96, 96 * 1024);
    run(k<128>, "128 B rows, 1 KB pitch", 128, 1024, 512, 0);
    run(k<128>, "128 B rows, 1 KB pitch", 128, 1024, 96, 96 * 1024);
    run(k<128>, "128 B rows, contiguous", 128, 128, 768, 0);
    run(k<128>, "128 B rows, 2 KB pitch (bf16 GEMM)", 128, 2048, 512, 0);
    run(k<128>, "128 B rows, 2 KB pitch (bf16 GEMM)", 128, 2048, 48, 96 * 1024);
    run(k<64>, "64 B rows, 1 KB pitch", 64, 1024, 512, 0);
    run(k<64>, "64 B rows, 1 KB pitch", 64, 1024, 96, 96 * 1024);
    run(k<128, true>, "128 B / 2 KB + barrier per stage", 128, 2048, 512, 0);
    run(k<128, false, true>, "128 B / 2 KB + LDS reads", 128, 2048, 512, 0);
    run(k<128, true, true>, "128 B / 2 KB + barrier + LDS reads", 128, 2048, 512, 0);
    run(k<128, false, true, true>, "128 B / 2 KB + LDS reads + MFMA", 128, 2048, 512, 0);
    run(k<128, true, true, true>, "128 B / 2 KB + barrier + reads + MFMA", 128, 2048, 512, 0);
    run(k<128, true, true, true, true>, "128 B / 2 KB + ping-pong reads | MFMA", 128, 2048, 512, 0);
    run(k<512>, "512 B rows, 2 KB pitch", 512, 2048, 256, 0);
    run(k<256>, "256 B rows, 2 KB pitch", 256, 2048, 256, 0);
    return 0;
}
