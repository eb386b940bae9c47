// Host-side micro-benchmark of the pooled VecEnv: N SeededEnvMock environments, S steps; prints us per step (median of 5 repeats).
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
    const int n = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 200, workers = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<std::shared_ptr<Env>> envs;
    for (int i = 0; i < n; ++i) envs.push_back(std::make_shared<SeededEnvMock>(1234u, (uint32_t)i));
    VecEnv ve{envs, workers};
    Mat actions = Mat::Zero(n, 18);
    for (int i = 0; i < 20; ++i) ve.step(actions);
    std::vector<double> rep;
    for (int r = 0; r < 5; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) ve.step(actions);
        rep.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps);
    }
    std::sort(rep.begin(), rep.end());
    std::printf("n %d workers %d: %.1f us per step (min %.1f max %.1f)\n", n, workers, rep[2], rep[0], rep[4]);
    return 0;
}
