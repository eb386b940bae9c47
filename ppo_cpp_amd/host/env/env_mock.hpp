// env_mock.hpp -- test doubles behind the Env interface.
//   EnvMock        behaviour of the reference's stub (env/env_mock.hpp:18-92): 18 obs / 18 act, observation and reward
//                  are the constant scaling_coeff, done on every 300th call, actions ignored.
//   SeededEnvMock  same shape, non-degenerate data: obs ~ U(-1,1)^18, reward ~ U(-1,1), done ~ Bernoulli(1/300) from
//                  the counter hash keyed by (seed, env id, step) that the device-side synthetic env also uses.
//   TargetEnv      a LEARNABLE task on SeededEnvMock's observation stream: reward = -mean_j (a_j - (W obs)_j)^2 for a fixed hashed
//                  matrix W, episodes of a fixed length.  The reference's only validation is that it learns (README.md:22-24: the
//                  hexapod's reward curves); this is the environment behind tests/test_learning.py, small enough for the oracle.
#pragma once
#include <cstdint>

#include "env.hpp"

class EnvMock : public Env {
public:
    explicit EnvMock(double scaling_coeff = 0.) : calls_(0), coeff_((float)scaling_coeff) {}

    std::string get_action_space() override { return Env::SPACE_CONTINOUS; }
    std::string get_observation_space() override { return Env::SPACE_CONTINOUS; }
    int get_action_space_size() override { return kDim; }
    int get_observation_space_size() override { return kDim; }

    Mat reset() override { return filled(get_num_envs(), kDim); }

    std::vector<Mat> step(const Mat& /*actions*/) override {
        ++calls_;
        Mat dones = Mat::Zero(get_num_envs(), 1);
        if (calls_ % 300 == 0) dones = Mat::Ones(get_num_envs(), 1);
        return {filled(get_num_envs(), kDim), filled(get_num_envs(), 1), dones};
    }

    Mat get_original_obs() override { return filled(get_num_envs(), kDim); }
    Mat get_original_rew() override { return filled(get_num_envs(), 1); }
    void serialize(nlohmann::json&) override {}
    void deserialize(nlohmann::json&) override {}
    void render() override {}
    float get_time() override { return 0.f; }

private:
    static constexpr int kDim = 18;
    Mat filled(int r, int c) const { Mat m = Mat::Zero(r, c); for (long i = 0; i < (long)r * c; ++i) m.data()[i] = coeff_; return m; }
    long calls_;
    float coeff_;
};

namespace ppo_detail {
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t ctr_key(uint32_t seed, uint32_t env) { return splitmix64(((uint64_t)seed << 32) | env); }      // the (seed, env) half of the hash
inline uint32_t ctr_hash_keyed(uint64_t key, uint32_t step, uint32_t lane) { return (uint32_t)(splitmix64(key ^ (((uint64_t)step << 32) | lane)) >> 32); }
inline uint32_t ctr_hash(uint32_t seed, uint32_t env, uint32_t step, uint32_t lane) { return ctr_hash_keyed(ctr_key(seed, env), step, lane); }
inline float sym_unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }
}  // namespace ppo_detail

class SeededEnvMock : public Env {
public:
    // obs_dim / act_dim: 18 / 18 like EnvMock; 36 / 18 stands in for the hexapod that also observes its velocities
    // (reference env/hexapod_closed_loop_env.hpp:20).  Hash lanes 0 .. obs_dim-1 = the observation, obs_dim = reward, obs_dim + 1 = done.
    SeededEnvMock(uint32_t seed, uint32_t env_id, int obs_dim = 18, int act_dim = 18)
        : seed_(seed), id_(env_id), step_(0), last_rew_(0.f), key_(ppo_detail::ctr_key(seed, env_id)), kDim(obs_dim), kAct(act_dim) {}
    std::string get_action_space() override { return Env::SPACE_CONTINOUS; }
    std::string get_observation_space() override { return Env::SPACE_CONTINOUS; }
    int get_action_space_size() override { return kAct; }
    int get_observation_space_size() override { return kDim; }
    Mat reset() override { step_ = 0; return obs_at(0); }
    std::vector<Mat> step(const Mat& /*actions*/) override {
        ++step_;
        Mat rew(1, 1), done(1, 1);
        last_rew_ = ppo_detail::sym_unit(ppo_detail::ctr_hash_keyed(key_, step_, kDim));
        rew(0, 0) = last_rew_;
        done(0, 0) = (ppo_detail::ctr_hash_keyed(key_, step_, kDim + 1) % 300u == 0u) ? 1.f : 0.f;
        std::vector<Mat> out;             // (a braced list would COPY the three matrices into the vector: three more allocations per env step)
        out.reserve(3);
        out.push_back(obs_at(step_)); out.push_back(std::move(rew)); out.push_back(std::move(done));
        return out;
    }
    Mat get_original_obs() override { return obs_at(step_); }
    Mat get_original_rew() override { Mat r(1, 1); r(0, 0) = last_rew_; return r; }
    void serialize(nlohmann::json&) override {}
    void deserialize(nlohmann::json&) override {}
    void render() override {}
    float get_time() override { return 0.f; }

private:
    Mat obs_at(uint32_t step) const { Mat m(1, kDim); for (int j = 0; j < kDim; ++j) m(0, j) = ppo_detail::sym_unit(ppo_detail::ctr_hash_keyed(key_, step, (uint32_t)j)); return m; }
    uint32_t seed_, id_, step_;
    float last_rew_;
    uint64_t key_;                  // splitmix64(seed, env id): the step-independent half of the counter hash
    int kDim, kAct;
};

class TargetEnv : public Env {
public:
    TargetEnv(uint32_t seed, uint32_t env_id, int obs_dim = 18, int act_dim = 18, int episode_len = 100)
        : step_(0), last_rew_(0.f), key_(ppo_detail::ctr_key(seed, env_id)), kDim(obs_dim), kAct(act_dim), len_(episode_len), w_((size_t)act_dim * obs_dim) {
        const uint64_t wkey = ppo_detail::splitmix64(((uint64_t)seed << 32) | 0xffffffffull);       // one W per seed, shared by every environment of the job
        for (int j = 0; j < kAct; ++j)
            for (int k = 0; k < kDim; ++k) w_[(size_t)j * kDim + k] = 0.5f * ppo_detail::sym_unit((uint32_t)(ppo_detail::splitmix64(wkey ^ (((uint64_t)j << 32) | (uint32_t)k)) >> 32));
    }
    std::string get_action_space() override { return Env::SPACE_CONTINOUS; }
    std::string get_observation_space() override { return Env::SPACE_CONTINOUS; }
    int get_action_space_size() override { return kAct; }
    int get_observation_space_size() override { return kDim; }
    Mat reset() override { step_ = 0; return obs_at(0); }
    std::vector<Mat> step(const Mat& actions) override {
        const Mat cur = obs_at(step_);                       // the observation the action answers
        float acc = 0.f;
        for (int j = 0; j < kAct; ++j) {
            float tgt = 0.f;
            for (int k = 0; k < kDim; ++k) tgt += w_[(size_t)j * kDim + k] * cur(0, k);
            const float d = actions(0, j) - tgt;
            acc += d * d;
        }
        last_rew_ = -acc / (float)kAct;
        ++step_;
        Mat rew(1, 1), done(1, 1);
        rew(0, 0) = last_rew_;
        done(0, 0) = (step_ % (uint32_t)len_ == 0u) ? 1.f : 0.f;
        std::vector<Mat> out;
        out.reserve(3);
        out.push_back(obs_at(step_)); out.push_back(std::move(rew)); out.push_back(std::move(done));
        return out;
    }
    Mat get_original_obs() override { return obs_at(step_); }
    Mat get_original_rew() override { Mat r(1, 1); r(0, 0) = last_rew_; return r; }
    void serialize(nlohmann::json&) override {}
    void deserialize(nlohmann::json&) override {}
    void render() override {}
    float get_time() override { return 0.f; }

private:
    Mat obs_at(uint32_t step) const { Mat m(1, kDim); for (int j = 0; j < kDim; ++j) m(0, j) = ppo_detail::sym_unit(ppo_detail::ctr_hash_keyed(key_, step, (uint32_t)j)); return m; }
    uint32_t step_;
    float last_rew_;
    uint64_t key_;
    int kDim, kAct, len_;
    std::vector<float> w_;          // [act][obs], row-major
};
