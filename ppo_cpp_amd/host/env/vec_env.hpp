// vec_env.hpp -- N sub-environments behind one Env, same class interface as the reference (env/vec_env.hpp:16-279).
//
// The reference spawns one std::thread per sub-environment with a mutex + condition variable per slot
// (env/vec_env.hpp:33-63, 108-154, 205-277); that cannot scale to the 4096 environments of the benchmark
// configuration.  Here a fixed pool of helper threads (<= usable CPUs - 1) and the CALLING thread share the environments of a
// step as small chunks claimed from an atomic counter: step() publishes the actions, bumps the generation, starts claiming
// chunks itself (no wake-up latency in front of the first environment) and returns when the last chunk is finished.  A helper that
// wakes late or is descheduled simply claims fewer chunks -- the pool adapts to the cores the box really gives it instead of
// waiting for its slowest fixed share (round 3: 1.44 ms vs 2.70 ms per 16 steps of 4096 mock environments on two boxes that both
// report 16 cores).  The chunk size is calibrated on the first three steps from the measured time per environment (~25 us of work
// per chunk, at least 4 chunks per thread); pool_workers() / pool_chunk() / pool_active() report what was chosen and how many
// threads actually took part.  Between two steps of a rollout (~100 us: the policy's act call) a helper does not go back to sleep at once: it
// watches the claim word for up to PPO_VECENV_SPIN_US (default 250 us; 0 = never; only while rounds have lately followed each other within half of that) and starts on the
// next round by itself -- a futex wake-up per helper and step (5-10 us each, issued one after the other by the caller) is what a step of a few
// thousand cheap environments otherwise mostly consists of; step() then only wakes as many helpers as are really asleep.  Semantics kept from the reference: sub-envs are reset once from pool threads at
// construction; reset() does NOT reset sub-envs but gathers get_original_obs() (env/vec_env.hpp:94-106);
// get_original_rew() returns the rewards of the last step; the caller's vector of environments is referenced, not
// copied (env/vec_env.hpp:190).  Conscious fix: get_observation_space_size() returns the OBSERVATION size (the
// reference returns the action size, env/vec_env.hpp:90-92 -- identical for every environment in its tree).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
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
        : envs_(envs), n_(static_cast<int>(envs.size())), generation_(0), terminate_(false), actions_(nullptr),
          obs_dim_(envs[0]->get_observation_space_size()), original_rewards_(Mat::Zero(n_, 1)) {
        assert(!envs.empty());
        const int hw = usable_cpus();
        workers_ = std::max(1, std::min(n_, max_workers > 0 ? max_workers : hw));
        // first guess: 8 chunks per thread; recalibrated from measured time per environment after the first steps
        chunk_ = std::max(1, n_ / (8 * workers_));
        per_.reset(new PerThread[workers_]);
        if (const char* e = std::getenv("PPO_VECENV_SPIN_US")) spin_cap_ns_ = std::max(0ll, std::atoll(e)) * 1000;
        mode_ = RESET;
        begin_round();
        for (int w = 1; w < workers_; ++w) threads_.emplace_back(&VecEnv::helper, this, w);
        wake_helpers();
        drain(0);                        // every sub-env is reset once, from the pool (env/vec_env.hpp:209)
        wait_round();
        mode_ = STEP;
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
        const auto t0 = std::chrono::steady_clock::now();
        actions_ = &actions;
        // the results leave by MOVE: the threads fill fresh matrices (their storage is the block the caller's previous result gave back,
        // mat.hpp) -- copying 300 KB out of members was half of a step at 4096 environments
        if (observations_.rows() != n_) observations_ = Mat::Zero(n_, obs_dim_);
        if (rewards_.rows() != n_) rewards_ = Mat::Zero(n_, 1);
        if (dones_.rows() != n_) dones_ = Mat::Zero(n_, 1);
        begin_round();
        wake_helpers();
        drain(0);
        wait_round();
        if (steps_ < kCalibrationSteps) calibrate(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        ++steps_;
        std::vector<Mat> out;
        out.reserve(3);
        out.push_back(std::move(observations_)); out.push_back(std::move(rewards_)); out.push_back(std::move(dones_));
        observations_ = Mat(); rewards_ = Mat(); dones_ = Mat();             // (a moved-from matrix keeps its shape fields: make them empty again)
        return out;
    }

    Mat get_original_obs() override { std::cout << "VecEnv::get_original_obs() not implemented\n"; return Mat::Zero(n_, get_observation_space_size()); }
    Mat get_original_rew() override { return original_rewards_; }
    void serialize(nlohmann::json&) override {}
    void deserialize(nlohmann::json&) override {}
    void render() override { std::cout << "VecEnv::render() not implemented\n"; }
    float get_time() override { std::cout << "VecEnv::get_time() not implemented\n"; return -1.f; }

    // what the pool settled on (not part of the reference's interface): threads incl. the caller, environments per chunk, and how
    // many threads claimed at least one chunk in the last step
    int pool_workers() const { return workers_; }
    int pool_chunk() const { return chunk_; }
    int pool_active() const { int a = 0; for (int w = 0; w < workers_; ++w) a += per_[w].claimed.load(std::memory_order_relaxed) > 0; return a; }

private:
    enum Mode { RESET, STEP };
    static constexpr int kCalibrationSteps = 3;

    // A round = one pass over the environments.  Everything a claim needs travels in ONE atomic word, so a helper that wakes late
    // (after the round it was woken for has ended, or while the next one is being set up) can never act on a mix of two rounds:
    //   ticket = [round : 24][chunks in the round : 20][next chunk : 20]
    // A claim is valid iff next < chunks; a valid claim means its round is still open, so chunk_ / actions_ / mode_ (published
    // before the ticket by the release store) are stable while the chunk is processed.
    static constexpr int kFieldBits = 20;
    static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // only as many helpers as there are chunks beyond the caller's first one are woken (a helper that sleeps through a round simply joins
    // the next one it is woken for): a step of a few hundred cheap environments costs one or two wake-ups, not fifteen
    void wake_helpers() {
        const int want = std::min(workers_ - 1, n_chunks_ - 1);
        if (want <= 0) return;
        int asleep;
        {
            std::lock_guard<std::mutex> l(m_);
            ++generation_;
            asleep = sleepers_;
        }
        // helpers that are not asleep are watching the ticket and have the round already
        const int need = std::min(asleep, want - (workers_ - 1 - asleep));
        if (need <= 0) return;
        if (need >= asleep) go_.notify_all();
        else for (int i = 0; i < need; ++i) go_.notify_one();
    }
    void begin_round() {
        n_chunks_ = (n_ + chunk_ - 1) / chunk_;
        if (n_chunks_ >= (1 << (kFieldBits - 1))) { chunk_ = (n_ + (1 << (kFieldBits - 1)) - 2) / ((1 << (kFieldBits - 1)) - 1); n_chunks_ = (n_ + chunk_ - 1) / chunk_; }
        for (int w = 0; w < workers_; ++w) per_[w].claimed.store(0, std::memory_order_relaxed);
        remaining_.store(n_chunks_, std::memory_order_relaxed);
        ++round_;
        ticket_.store(((round_ & 0xFFFFFFull) << (2 * kFieldBits)) | (static_cast<unsigned long long>(n_chunks_) << kFieldBits), std::memory_order_release);
    }
    // claim chunks until none is left; `who` = 0 for the calling thread, 1.. for helpers
    void drain(int who) {
        Mat a(1, 1);
        int mine = 0;
        bool timing = false;
        for (;;) {
            const unsigned long long t = ticket_.fetch_add(1, std::memory_order_acq_rel);
            const int c = static_cast<int>(t & ((1ull << kFieldBits) - 1)), nc = static_cast<int>((t >> kFieldBits) & ((1ull << kFieldBits) - 1));
            if (c >= nc) break;
            // (read only behind a valid claim: steps_ / mode_ change between rounds, on the caller's thread, before the ticket's release store)
            if (++mine == 1) timing = steps_ < kCalibrationSteps && mode_ == STEP;
            // calibration steps: the time inside the environments only (claims and waiting excluded: with the first guess of tiny chunks they would
            // be most of it, and the chunk derived from it would stay tiny)
            const long long ta = timing ? now_ns() : 0;
            const int begin = c * chunk_, end = std::min(n_, begin + chunk_);
            if (mode_ == RESET) {
                for (int i = begin; i < end; ++i) envs_[i]->reset();
            } else {
                const int acols = static_cast<int>(actions_->cols());
                if (a.cols() != acols) a = Mat(1, acols);           // one action row per thread and step, reused for its environments
                for (int i = begin; i < end; ++i) {
                    mat_set_row(a, 0, mat_row_ptr(*actions_, i));
                    const std::vector<Mat> res = envs_[i]->step(a);
                    mat_set_row(observations_, i, res[0].data());
                    rewards_(i, 0) = res[1](0, 0);
                    dones_(i, 0) = res[2](0, 0);
                    original_rewards_(i, 0) = envs_[i]->get_original_rew()(0, 0);
                }
            }
            // this chunk's share is published BEFORE the decrement that may end the round: the caller reads the shares (calibrate, pool_active) after it
            // has seen remaining_ == 0, and the acq_rel decrement orders these relaxed adds in front of that read -- whichever thread finished last
            per_[who].claimed.fetch_add(1, std::memory_order_relaxed);
            if (timing) per_[who].busy_ns.fetch_add(now_ns() - ta, std::memory_order_relaxed);
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1 && who != 0) {
                std::lock_guard<std::mutex> l(m_);               // the round's last chunk: wake the caller if it may be asleep in wait_round()
                done_.notify_one();
            }
        }
    }
    // the caller has no chunk left to claim: the last ones are being finished by helpers (a few microseconds: spin, then sleep)
    void wait_round() {
        for (int spin = 0; spin < 2000; ++spin) {
            if (remaining_.load(std::memory_order_acquire) == 0) return;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return remaining_.load(std::memory_order_acquire) == 0; });
    }
    void helper(int who) {
        unsigned long seen = 0;
        unsigned long long last_round = 0;                      // round of this thread's last drain
        // `credit` > 0: the next round lately came within half the spin budget, so watching for it beats sleeping.  One long pause (the update
        // between two rollouts) costs one credit, not the habit; environments whose steps are always far apart run it down to 0 and sleep at once.
        int credit = 2;
        long long idle_since = now_ns();
        for (;;) {
            bool go = false;
            const long long budget = spin_cap_ns_;
            if (spin_cap_ns_ > 0 && credit > 0) {
                for (;;) {
                    if ((ticket_.load(std::memory_order_acquire) >> (2 * kFieldBits)) != last_round) { go = true; break; }
#if defined(__x86_64__)
                    for (int i = 0; i < 16; ++i) __builtin_ia32_pause();
#endif
                    if (now_ns() - idle_since > budget) break;
                }
            }
            if (!go) {
                std::unique_lock<std::mutex> l(m_);
                if (generation_ == seen) {
                    ++sleepers_;
                    go_.wait(l, [&] { return generation_ != seen; });
                    --sleepers_;
                }
                seen = generation_;
                if (terminate_) return;
            }
            credit = (now_ns() - idle_since) * 2 <= spin_cap_ns_ ? std::min(credit + 1, 4) : std::max(credit - 1, 0);
            last_round = ticket_.load(std::memory_order_acquire) >> (2 * kFieldBits);
            drain(who);
            idle_since = now_ns();
        }
    }
    // after each of the first steps: time per environment from the threads' own busy time (not the step's wall time, which is mostly wake-up
    // latency for cheap environments) -> ~12 us of work per chunk, at most 8 chunks per thread, at least one environment per chunk; rounds
    // with fewer chunks than threads wake fewer helpers
    void calibrate(double /*wall_s*/) {
        long long busy = 0;
        for (int w = 0; w < workers_; ++w) busy += per_[w].busy_ns.exchange(0, std::memory_order_relaxed);
        if (busy <= 0) return;
        const double per_env = 1e-9 * (double)busy / n_;
        int c = static_cast<int>(12e-6 / per_env);
        c = std::max(c, (n_ + 8 * workers_ - 1) / (8 * workers_));
        // never fewer than two chunks per thread: an estimate that came out low (a step timed while helpers were still waking) must not leave threads
        // without work for the lifetime of the pool
        c = std::min(c, (n_ + 2 * workers_ - 1) / (2 * workers_));
        chunk_ = std::max(1, std::min(c, n_));
    }

    const std::vector<std::shared_ptr<Env>>& envs_;
    const int n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable go_, done_;
    unsigned long generation_;
    int sleepers_ = 0;                 // helpers inside go_.wait (guarded by m_)
    long long spin_cap_ns_ = 250000;
    bool terminate_;
    const Mat* actions_;
    const int obs_dim_;
    Mat observations_, rewards_, dones_, original_rewards_;
    int workers_ = 1, chunk_ = 1, n_chunks_ = 1, steps_ = 0;
    Mode mode_ = RESET;
    // the two words every thread hammers live on cache lines of their own; so does each thread's share record
    alignas(64) std::atomic<unsigned long long> ticket_{0};
    alignas(64) std::atomic<int> remaining_{0};
    struct alignas(64) PerThread { std::atomic<int> claimed{0}; std::atomic<long long> busy_ns{0}; };
    std::unique_ptr<PerThread[]> per_;
    unsigned long long round_ = 0;
};
