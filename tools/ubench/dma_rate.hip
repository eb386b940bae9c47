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
template <int ROWB>
__global__ __launch_bounds__(512) void k(const char* __restrict__ base, int pitch, int window_rows, size_t wg_stride, int stages, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = ROWB / 16, RPP = 1024 / ROWB, NPF = 6;
    const char* mine = base + (size_t)blockIdx.x * wg_stride;
    const size_t lane_off = (size_t)(lane / LPR) * pitch + (size_t)(lane % LPR) * 16;
    char* ring = lds + wave * 2 * NPF * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    int row = wave * RPP * NPF;
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            dma16(mine + (size_t)((row + q * RPP) % window_rows) * pitch + lane_off, ring + ((s & 1) * NPF + q) * 1024);
        }
        row += 8 * RPP * NPF;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t bytes = 512ull << 20;
    char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * sizeof(unsigned long long)));
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
        printf("%-34s row %4d B pitch %5d window %5d rows (%6.0f KB) %s: %7.1f us  median %8llu cycles  %5.1f B/clk/CU  (%.1f TB/s chip)\n", name, rowb, pitch, window_rows,
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
    run(k<512>, "512 B rows, 2 KB pitch", 512, 2048, 256, 0);
    run(k<256>, "256 B rows, 2 KB pitch", 256, 2048, 256, 0);
    return 0;
}
