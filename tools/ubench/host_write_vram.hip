// Can the HOST store straight into device memory (large BAR)?  hipExtMallocWithFlags(hipDeviceMallocFinegrained) / hipMallocManaged with a device-preferred
// location; a kernel then spins on a word the host writes and reports how many polls it took.
// build: hipcc --offload-arch=gfx950 -O2 -o host_write_vram host_write_vram.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <csignal>
#include <csetjmp>
#include <chrono>
#include <thread>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void spin(volatile unsigned* w, unsigned* out, unsigned long long* cyc) {
    unsigned n = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__hip_atomic_load((unsigned*)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && n < 200000000u) ++n;
    *cyc = __builtin_readcyclecounter() - t0;
    *out = n;
}
int main() {
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    for (int mode = 0; mode < 3; ++mode) {
        unsigned* p = nullptr; hipError_t e;
        if (mode == 0) e = hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocFinegrained);
        else if (mode == 1) e = hipMalloc((void**)&p, 4096);
        else { e = hipMallocManaged((void**)&p, 4096); if (e == hipSuccess) { (void)hipMemAdvise(p, 4096, hipMemAdviseSetPreferredLocation, 0); (void)hipMemAdvise(p, 4096, hipMemAdviseSetCoarseGrain, 0); } }
        printf("mode %d alloc: %s\n", mode, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        (void)hipMemset(p, 0, 4096); (void)hipDeviceSynchronize();
        unsigned* out; unsigned long long* cyc; (void)hipMalloc((void**)&out, 8); (void)hipMalloc((void**)&cyc, 8);
        if (sigsetjmp(jb, 1)) { printf("   host store into this memory: SIGSEGV/SIGBUS\n"); continue; }
        if (mode == 1) { volatile unsigned probe = *(volatile unsigned*)p; (void)probe; }      // (plain hipMalloc: expected to fault)
        hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, 0, p, out, cyc);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        const auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(p, 1u, __ATOMIC_RELEASE);
        (void)hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        unsigned n = 0; unsigned long long c = 0; (void)hipMemcpy(&n, out, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("   host store worked; kernel saw it after %u polls (%.1f cycles per poll); store -> kernel end -> sync returned: %.1f us\n", n, n ? (double)c / n : 0.0, us);
    }
    return 0;
}
