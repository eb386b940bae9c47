// vec_env.hpp -- N sub-environments behind one Env, same class interface as the reference (env/vec_env.hpp:16-279).
//
// The reference spawns one std::thread per sub-environment with a mutex + condition variable per slot
// (env/vec_env.hpp:33-63, 108-154, 205-277); that cannot scale to the 4096 environments of the benchmark
// configuration.  Here a fixed pool of workers (<= hardware threads) owns contiguous ranges of environments and is
// driven by a generation counter: step() publishes the actions, bumps the generation and waits until every worker has
// finished its range.  Semantics kept from the reference: sub-envs are reset once from the worker threads at
// construction; reset() does NOT reset sub-envs but gathers get_original_obs() (env/vec_env.hpp:94-106);
// get_original_rew() returns the rewards of the last step; the caller's vector of environments is referenced, not
// copied (env/vec_env.hpp:190).  Conscious fix: get_observation_space_size() returns the OBSERVATION size (the
// reference returns the action size, env/vec_env.hpp:90-92 -- identical for every environment in its tree).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>
#if defined(__linux__)
#include <sched.h>
#endif

#include "env.hpp"

// CPUs this process may actually use: hardware threads, narrowed by the affinity mask and by the cgroup CPU quota
// (a container often sees every core of the host but is throttled to a few; a worker per visible core then only
// adds contention)
inline int usable_cpus() {
    int n = static_cast<int>(std::thread::hardware_concurrency());
    if (n < 1) n = 1;
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
    auto quota = [](const char* quota_file, const char* period_file) -> int {
        FILE* f = std::fopen(quota_file, "r");
        if (!f) return 0;
        char a[64] = {0}; long long period = 0, q = 0;
        int got = period_file ? std::fscanf(f, "%63s", a) : std::fscanf(f, "%63s %lld", a, &period);
        std::fclose(f);
        if (got < 1 || a[0] == 'm' || a[0] == '-') return 0;            // "max" / -1: unlimited
        q = std::atoll(a);
        if (period_file) { FILE* g = std::fopen(period_file, "r"); if (!g) return 0; if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
        if (q <= 0 || period <= 0) return 0;
        return static_cast<int>((q + period - 1) / period);
    };
    int q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                               // cgroup v2
    if (q == 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // v1
    if (q > 0) n = std::min(n, q);
#endif
    return n;
}

class VecEnv : public virtual Env {
public:
    explicit VecEnv(const std::vector<std::shared_ptr<Env>>& envs, int max_workers = 0)
        : envs_(envs), n_(static_cast<int>(envs.size())), generation_(0), pending_(0), terminate_(false), actions_(nullptr),
          observations_(Mat::Zero(n_, envs[0]->get_observation_space_size())), rewards_(Mat::Zero(n_, 1)), dones_(Mat::Zero(n_, 1)),
          original_rewards_(Mat::Zero(n_, 1)) {
        assert(!envs.empty());
        const int hw = usable_cpus();
        int workers = std::min(n_, max_workers > 0 ? max_workers : hw);
        const int per = (n_ + workers - 1) / workers;
        workers = (n_ + per - 1) / per;
        {
            std::lock_guard<std::mutex> l(m_);
            pending_ = workers;
        }
        for (int w = 0; w < workers; ++w) threads_.emplace_back(&VecEnv::worker, this, w * per, std::min(n_, (w + 1) * per));
        wait_all();                      // every sub-env has been reset once (env/vec_env.hpp:209)
    }
    VecEnv(const VecEnv&) = delete;
    VecEnv& operator=(const VecEnv&) = delete;

    ~VecEnv() override {
        {
            std::lock_guard<std::mutex> l(m_);
            terminate_ = true;
            ++generation_;
        }
        go_.notify_all();
        for (auto& t : threads_) t.join();
    }

    std::string get_action_space() override { return envs_[0]->get_action_space(); }
    std::string get_observation_space() override { return envs_[0]->get_observation_space(); }
    int get_action_space_size() override { return envs_[0]->get_action_space_size(); }
    int get_observation_space_size() override { return envs_[0]->get_observation_space_size(); }
    int get_num_envs() override { return n_; }

    Mat reset() override {
        Mat result = Mat::Zero(n_, get_observation_space_size());
        for (int i = 0; i < n_; ++i) { const Mat o = envs_[i]->get_original_obs(); mat_set_row(result, i, o.data()); }
        return result;
    }

    std::vector<Mat> step(const Mat& actions) override {
        assert(actions.rows() == n_);
        {
            std::lock_guard<std::mutex> l(m_);
            actions_ = &actions;
            pending_ = static_cast<int>(threads_.size());
            ++generation_;
        }
        go_.notify_all();
        wait_all();
        return {observations_, rewards_, dones_};
    }

    Mat get_original_obs() override { std::cout << "VecEnv::get_original_obs() not implemented\n"; return Mat::Zero(n_, get_observation_space_size()); }
    Mat get_original_rew() override { return original_rewards_; }
    void serialize(nlohmann::json&) override {}
    void deserialize(nlohmann::json&) override {}
    void render() override { std::cout << "VecEnv::render() not implemented\n"; }
    float get_time() override { std::cout << "VecEnv::get_time() not implemented\n"; return -1.f; }

private:
    void wait_all() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    void finish_one() {
        bool last;
        {
            std::lock_guard<std::mutex> l(m_);
            last = (--pending_ == 0);
        }
        if (last) done_.notify_one();
    }
    void worker(int begin, int end) {
        for (int i = begin; i < end; ++i) envs_[i]->reset();
        unsigned long seen = 0;
        Mat a(1, 1);
        finish_one();
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                go_.wait(l, [&] { return generation_ != seen; });
                seen = generation_;
                if (terminate_) return;
            }
            const int acols = static_cast<int>(actions_->cols());
            if (a.cols() != acols) a = Mat(1, acols);               // one action row per worker, reused (not one allocation per env step)
            for (int i = begin; i < end; ++i) {
                mat_set_row(a, 0, mat_row_ptr(*actions_, i));
                const std::vector<Mat> res = envs_[i]->step(a);
                mat_set_row(observations_, i, res[0].data());
                rewards_(i, 0) = res[1](0, 0);
                dones_(i, 0) = res[2](0, 0);
                original_rewards_(i, 0) = envs_[i]->get_original_rew()(0, 0);
            }
            finish_one();
        }
    }

    const std::vector<std::shared_ptr<Env>>& envs_;
    const int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable go_, done_;
    unsigned long generation_;
    int pending_;
    bool terminate_;
    const Mat* actions_;
    Mat observations_, rewards_, dones_, original_rewards_;
};
