// micro-benchmark (round 6): bf16_heads_kernel of ppo_bf16.hpp at the configs[4] shape (4096 rows, K = 1024, 64 actions): time per launch (50 back to back) and the
// error of a few rows against a double-precision sum; TRIM = only the 16-byte chunks of W's columns that exist are requested (lanes masked out of the LDS-DMA).
// build: hipcc --offload-arch=gfx950 -O3 -I ppo_cpp_amd/csrc -o tools/ubench/heads tools/ubench/heads.hip ; run: tools/ubench/heads
#include "ppo_bf16.hpp"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int R = 4096, K = 1024, Ap = 128, A = 64;
    std::mt19937 rng(7); std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<bf16_t> hH(2 * (size_t)R * K), hW(2 * (size_t)K * Ap); std::vector<float> hb(2 * Ap);
    for (auto& x : hH) x = (bf16_t)u(rng);
    for (auto& x : hW) x = (bf16_t)(u(rng) * 0.05f);
    for (auto& x : hb) x = u(rng);
    bf16_t *dH, *dW; float *db, *dF; const int KS = 4;
    CK(hipMalloc(&dH, hH.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dF, 2 * (size_t)KS * R * Ap * 4));
    CK(hipMemcpy(dH, hH.data(), hH.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    HeadArgsB a{};
    for (int t = 0; t < 2; ++t) { a.H[t] = dH + (size_t)t * R * K; a.W[t] = dW + (size_t)t * K * Ap; a.bias[t] = db + t * Ap; a.F[t] = dF + (size_t)t * KS * R * Ap; }
    a.ldh = K; a.ldw = Ap; a.ldf = Ap; a.K = K; a.ksplit = KS; a.f_split = (size_t)R * Ap;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int towers, const char* name) -> int {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, BH_LDS_BYTES));
        CK(hipMemset(dF, 0, 2 * (size_t)KS * R * Ap * 4));
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0));
            for (int n = 0; n < 50; ++n) hipLaunchKernelGGL(kern, dim3(R / BH_ROWS * KS, towers), dim3(256), BH_LDS_BYTES, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 50);
        }
        std::vector<float> F(2 * (size_t)KS * R * Ap); CK(hipMemcpy(F.data(), dF, F.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < towers; ++t)
            for (int row : {0, 1, 17, 31, 32, 127, 128, 2049, 4095})
                for (int j = 0; j < (t ? 1 : A); ++j) {
                    double s = hb[t * Ap + j];
                    for (int k = 0; k < K; ++k) s += (double)(float)hH[((size_t)t * R + row) * K + k] * (double)(float)hW[((size_t)t * K + k) * Ap + j];
                    double f = 0; for (int q = 0; q < KS; ++q) f += F[(((size_t)t * KS + q) * R + row) * Ap + j];
                    worst = std::max(worst, std::abs(s - f));
                }
#ifdef PPO_STAMPS
        {
            unsigned long long* st; CK(hipMalloc(&st, 2 * 1024 * 8 * 8)); CK(hipMemset(st, 0, 2 * 1024 * 8 * 8));
            HeadArgsB b = a; b.stamps = st;
            hipLaunchKernelGGL(kern, dim3(R / BH_ROWS * KS, towers), dim3(256), BH_LDS_BYTES, 0, b);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(2 * 1024 * 8); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            const int nblk = R / BH_ROWS * KS * towers;
            printf("   stamps (cycles, median over %d workgroups): ", nblk);
            const char* nm[] = {"requests issued", "own requests landed", "barrier", "matrix work", "stores drained"};
            for (int k = 0; k < 5; ++k) { std::vector<long> d; for (int i = 0; i < nblk; ++i) d.push_back((long)(h[i * 8 + k + 1] - h[i * 8 + k])); std::sort(d.begin(), d.end()); printf("%s %ld | ", nm[k], d[nblk / 2]); }
            printf("\n");
            CK(hipFree(st));
        }
#endif
        printf("%-72s %6.2f us per launch   max |error| against a double sum %.2e\n", name, best * 1e3, worst);
        return 0;
    };
    if (run(bf16_heads_kernel<false, 4>, 2, "whole rows of W requested, both towers")) return 1;
    if (run(bf16_heads_kernel<false, 4>, 1, "whole rows of W requested, policy tower only")) return 1;
    if (run(bf16_heads_kernel<true, 4>, 2, "TRIM, both towers")) return 1;
    if (run(bf16_heads_kernel<true, 4>, 1, "TRIM, policy tower only")) return 1;
    return 0;
}
