// wave_sum_lane0 / half_sum_lane0 (ppo_kernels.hpp) against the __shfl_xor butterflies they replace: same BITS in the lanes that are used (lane 0 of a wave; lanes 0 and 32
// of the two 32-lane halves), on random data incl. mixed signs and magnitudes.
// build: hipcc --offload-arch=gfx950 -O2 -I ppo_cpp_amd/csrc -I include -o lane0_sum tools/ubench/lane0_sum.hip
#include <hip/hip_runtime.h>
#include "ppo_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
__global__ void check(const float* x, unsigned* bad, int n_waves) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const float v = x[(size_t)w * 64 + lane];
    float a = v; for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    const float b = wave_sum_lane0(v);
    float c = v; for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o);
    const float d = half_sum_lane0(v);
    if (lane == 0 && __float_as_uint(a) != __float_as_uint(b)) atomicAdd(bad, 1u);
    if ((lane & 31) == 0 && __float_as_uint(c) != __float_as_uint(d)) atomicAdd(bad + 1, 1u);
}
int main() {
    const int n_waves = 1 << 18;
    std::vector<float> h((size_t)n_waves * 64);
    srand(1);
    for (auto& f : h) { const float m = (float)(rand() % 2000 - 1000) / 1000.f; const int e = rand() % 40 - 20; f = ldexpf(m, e); }
    float* x; unsigned* bad; (void)hipMalloc((void**)&x, h.size() * 4); (void)hipMalloc((void**)&bad, 8); (void)hipMemset(bad, 0, 8);
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(n_waves / 4), dim3(256), 0, 0, x, bad, n_waves);
    unsigned r[2]; (void)hipMemcpy(r, bad, 8, hipMemcpyDeviceToHost);
    printf("%d waves: wave_sum_lane0 mismatches %u, half_sum_lane0 mismatches %u\n", n_waves, r[0], r[1]);
    return r[0] || r[1];
}
