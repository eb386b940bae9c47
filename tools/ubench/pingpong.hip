// Host <-> resident-kernel ping-pong floor (the host-Env protocol of narrow_rollout1_kernel<.., HOST>): the host stores a sequence word into DEVICE memory through the
// BAR, one wave polls it and answers into pinned HOST memory, the host polls that.  Three answers: (a) flag only, (b) 18 floats + wait for their acknowledgement + flag
// (what the kernel does), (c) 18 floats and the sequence number in the same 64-byte lines (self-validating lines, no wait).   us per round trip.
// build: hipcc --offload-arch=gfx950 -O2 -o pingpong pingpong.hip
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
__global__ void responder(const unsigned* in_word, float* out, unsigned* out_flag, int mode, int rounds) {
    const int lane = threadIdx.x;
    for (int t = 1; t <= rounds; ++t) {
        if (lane == 0) while (__hip_atomic_load(in_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)t) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
        if (mode == 0) { if (lane == 0) __hip_atomic_store(out_flag, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else if (mode == 1) {
            if (lane < 18) __hip_atomic_store(out + lane, (float)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(out_flag, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            // two 64-byte lines: [15 floats | seq] [3 floats | .. | seq]: lanes 0..31, one store instruction
            if (lane < 32) { const bool is_seq = (lane & 15) == 15; __hip_atomic_store(reinterpret_cast<unsigned*>(out) + lane, is_seq ? (unsigned)t : __float_as_uint((float)t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}
int main() {
    unsigned* in_word; (void)hipMalloc((void**)&in_word, 256); (void)hipMemset(in_word, 0, 256);
    float* out; unsigned* out_flag; (void)hipHostMalloc((void**)&out, 4096, hipHostMallocDefault); out_flag = reinterpret_cast<unsigned*>(out) + 512;
    const int rounds = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(in_word, 0, 256); for (int i = 0; i < 1024; ++i) reinterpret_cast<volatile unsigned*>(out)[i] = 0; (void)hipDeviceSynchronize();
        float* dout; unsigned* dflag; (void)hipHostGetDevicePointer((void**)&dout, out, 0); dflag = reinterpret_cast<unsigned*>(dout) + 512;
        hipLaunchKernelGGL(responder, dim3(1), dim3(64), 0, 0, in_word, dout, dflag, mode, rounds);
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 1; t <= rounds; ++t) {
            _mm_sfence(); *reinterpret_cast<volatile unsigned*>(in_word) = (unsigned)t; _mm_sfence();
            if (mode < 2) while (__atomic_load_n(out_flag, __ATOMIC_ACQUIRE) < (unsigned)t) { }
            else while (__atomic_load_n(reinterpret_cast<unsigned*>(out) + 15, __ATOMIC_ACQUIRE) < (unsigned)t || __atomic_load_n(reinterpret_cast<unsigned*>(out) + 31, __ATOMIC_ACQUIRE) < (unsigned)t) { }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
        (void)hipDeviceSynchronize();
        printf("mode %d (%s): %.2f us per round trip\n", mode, mode == 0 ? "flag only" : mode == 1 ? "18 floats, wait, flag" : "self-validating lines", us);
    }
    return 0;
}
