// micro-benchmark: 256 workgroups x 4 waves; every wave streams its 64-column slice of a [256][256] fp32 matrix from
// L2 (all workgroups read the SAME matrix, like the PPO layers) and feeds 16x16x4 MFMAs.  Variants isolate the limits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int RING, int KS>   // MODE 0: loads+mfma, 1: loads only, 2: mfma only, 3: loads+mfma interleaved 1 load / 4 mfma, A from LDS
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int reps, int nmat) {
    __shared__ __attribute__((aligned(16))) float xs[16 * 260];
    for (int i = threadIdx.x; i < 16 * 260; i += 256) xs[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int col = wave * 64 + 4 * c;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float4 w[RING][KS * 4];
    constexpr int KB = 16 * KS;
    float sink = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        const float* Wm = W + (size_t)(rep % nmat) * 65536;
        auto load = [&](float4* dst, int kb) __attribute__((always_inline)) {
            kb = kb < 256 ? kb : 256 - KB;
#pragma unroll
            for (int q = 0; q < KS * 4; ++q) dst[q] = *reinterpret_cast<const float4*>(Wm + (size_t)(kb + 16 * (q / 4) + 4 * g + (q & 3)) * 256 + col);
        };
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) load(w[i], i * KB);
        }
        for (int kb = 0; kb < 256; kb += RING * KB) {
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                if (MODE != 2) load(w[(i + RING - 1) % RING], kb + (i + RING - 1) * KB);
                if (MODE != 3) __builtin_amdgcn_sched_barrier(0);
                if (MODE == 3) {
                    if (kb + i * KB < 256) {
#pragma unroll
                        for (int q2 = 0; q2 < KS; ++q2) {
                            const float4 a4 = *reinterpret_cast<const float4*>(xs + c * 260 + kb + i * KB + 16 * q2 + 4 * g);
                            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                const float4 v = w[i][q2 * 4 + s4];
                                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], v.x, acc[0], 0, 0, 0);
                                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], v.y, acc[1], 0, 0, 0);
                                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], v.z, acc[2], 0, 0, 0);
                                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], v.w, acc[3], 0, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
#pragma unroll
                    for (int t = 0; t < KS * 4; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                } else if (kb + i * KB < 256) {
#pragma unroll
                    for (int q = 0; q < KS * 4; ++q) {
                        float4 v = (MODE == 2) ? make_float4(1.f, 2.f, 3.f, 4.f) : w[i][q];
                        if (MODE == 1) { sink += v.x + v.y + v.z + v.w; }
                        else {
                            const float a = (float)lane;
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v.x, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v.y, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v.z, acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v.w, acc[3], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = sink;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int RING, int KS>
int run(const char* name, const float* W, float* out, int nwg, int nmat) {
    const int reps = 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<MODE, RING, KS>), dim3(nwg), dim3(256), 0, 0, W, out, reps, nmat);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, RING, KS>), dim3(nwg), dim3(256), 0, 0, W, out, reps, nmat);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us_per_mat = 1e3 * ms / reps;
    printf("%-34s wgs %4d nmat %2d: %7.2f us per 256x256 matrix pass  (L2->CU %.2f TB/s, mfma-bound 3.6us)\n", name, nwg, nmat, us_per_mat,
           nwg * 262144.0 / (us_per_mat * 1e-6) / 1e12);
    return 0;
}

int main() {
    float *W, *out;
    const int NM = 16;
    CK(hipMalloc(&W, NM * 65536 * 4)); CK(hipMalloc(&out, 4096));
    std::vector<float> h(NM * 65536, 0.5f); CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int nmat : {1, 8}) {
        run<3, 2, 2>("interleaved ring2 ks2 +ldsA", W, out, 256, nmat);
        run<3, 3, 2>("interleaved ring3 ks2 +ldsA", W, out, 256, nmat);
        run<3, 3, 1>("interleaved ring3 ks1 +ldsA", W, out, 256, nmat);
        run<3, 4, 1>("interleaved ring4 ks1 +ldsA", W, out, 256, nmat);
        run<0, 2, 2>("loads+mfma ring2 ks2", W, out, 256, nmat);
        run<0, 3, 2>("loads+mfma ring3 ks2", W, out, 256, nmat);
        run<0, 4, 1>("loads+mfma ring4 ks1", W, out, 256, nmat);
        run<1, 2, 2>("loads only ring2 ks2", W, out, 256, nmat);
        run<1, 3, 2>("loads only ring3 ks2", W, out, 256, nmat);
        run<2, 2, 2>("mfma only", W, out, 256, nmat);
    }
    run<0, 3, 2>("loads+mfma ring3 ks2", W, out, 128, 8);
    run<1, 3, 2>("loads only ring3 ks2", W, out, 128, 8);
    run<0, 3, 2>("loads+mfma ring3 ks2", W, out, 512, 8);
    return 0;
}
