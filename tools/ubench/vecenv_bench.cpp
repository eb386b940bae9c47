// Host-side micro-benchmark of the pooled VecEnv: N SeededEnvMock environments, S steps; prints us per step (median of 5 repeats).
// args: N [steps [workers [gap_us]]] -- gap_us: the caller idles that long between two steps (the policy's act call of a rollout); not counted.
// Build (either header): g++ -O2 -std=c++17 -pthread -I ppo_cpp_amd/host [-DVEC_ENV_HEADER='"path/to/other/vec_env.hpp"'] tools/ubench/vecenv_bench.cpp -o vecenv_bench
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#ifdef VEC_ENV_HEADER
#include "env/env.hpp"
#include VEC_ENV_HEADER
#else
#include "env/vec_env.hpp"
#endif
#include "env/env_mock.hpp"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 200, workers = argc > 3 ? atoi(argv[3]) : 0, gap_us = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<std::shared_ptr<Env>> envs;
    for (int i = 0; i < n; ++i) envs.push_back(std::make_shared<SeededEnvMock>(1234u, (uint32_t)i));
    VecEnv ve{envs, workers};
    Mat actions = Mat::Zero(n, 18);
    for (int i = 0; i < 20; ++i) ve.step(actions);
    std::vector<double> rep;
    for (int r = 0; r < 5; ++r) {
        double in_step = 0;
        for (int i = 0; i < steps; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            ve.step(actions);
            const auto t1 = std::chrono::steady_clock::now();
            in_step += std::chrono::duration<double, std::micro>(t1 - t0).count();
            while (gap_us > 0 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() < gap_us) {}
        }
        rep.push_back(in_step / steps);
    }
    std::sort(rep.begin(), rep.end());
    std::printf("n %d workers %d (pool %d, chunk %d, active in the last step %d) gap %d us: %.1f us per step (min %.1f max %.1f)\n", n, workers, ve.pool_workers(), ve.pool_chunk(), ve.pool_active(), gap_us, rep[2], rep[0], rep[4]);
    return 0;
}
