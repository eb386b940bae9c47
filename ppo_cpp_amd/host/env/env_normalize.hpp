// env_normalize.hpp -- the reference's VecNormalize port (env/env_normalize.hpp:16-164) as an Env wrapper whose
// arithmetic runs on the GPU: running mean/variance of observations and of the discounted return, normalise + clip
// (+-10) of both, through libppo_hip's normaliser (ppo_norm_* in include/ppo_hip.h).  Same constructor arguments and
// the same serialise format ({"obs_rms": {mean, var, count}, "ret_rms": {...}}, env_normalize.hpp:134-146) plus the
// handle that owns the device state.  There is no host arithmetic here and no CPU fallback.
#pragma once
#include <memory>
#include <stdexcept>

#include "../../../include/ppo_hip.h"
#include "env.hpp"

class EnvNormalize : public Env {
public:
    EnvNormalize(std::unique_ptr<Env> env, ppo_handle* handle, bool training, bool norm_obs = true, bool norm_reward = true,
                 float clip_reward = 10, float clip_obs = 10, float gamma = 0.99f, float epsilon = 1e-8f)
        : env_(std::move(env)), h_(handle), training_(training), norm_obs_(norm_obs), norm_reward_(norm_reward) {
        check(ppo_norm_init(h_, env_->get_num_envs(), gamma, clip_obs, clip_reward, epsilon));
        check(ppo_norm_set_flags(h_, norm_obs_ ? 1 : 0, norm_reward_ ? 1 : 0));     // honoured by the device-resident rollout too
    }

    std::string get_action_space() override { return env_->get_action_space(); }
    std::string get_observation_space() override { return env_->get_observation_space(); }
    int get_action_space_size() override { return env_->get_action_space_size(); }
    int get_observation_space_size() override { return env_->get_observation_space_size(); }
    int get_num_envs() override { return env_->get_num_envs(); }

    std::vector<Mat> step(const Mat& actions) override {
        std::vector<Mat> r = env_->step(actions);
        Mat obs = normalize_observation(r[0]);
        Mat rew = r[1];
        // always called: the discounted return accumulates whether or not rewards are scaled (env_normalize.hpp:66,91);
        // with norm_reward == false the library passes the rewards through and leaves ret_rms alone (:75)
        check(ppo_norm_reward(h_, r[1].data(), r[2].data(), get_num_envs(), training_ ? 1 : 0, rew.data()));
        return {obs, rew, r[2]};
    }

    Mat reset() override {
        const Mat obs = env_->reset();
        check(ppo_norm_reset_returns(h_));                        // ret = Zero (env_normalize.hpp:114); statistics kept
        return normalize_observation(obs);
    }

    void render() override { env_->render(); }
    float get_time() override { return env_->get_time(); }
    Mat get_original_obs() override { return env_->get_original_obs(); }
    Mat get_original_rew() override { return env_->get_original_rew(); }

    void serialize(nlohmann::json& json) override {
        write_stats(json["obs_rms"], 0, get_observation_space_size());
        write_stats(json["ret_rms"], 1, 1);
        env_->serialize(json);
    }
    void deserialize(nlohmann::json& json) override {
        read_stats(json["obs_rms"], 0);
        read_stats(json["ret_rms"], 1);
        env_->deserialize(json);
    }
    Env& inner() { return *env_; }
    bool training() const { return training_; }
    bool norm_obs() const { return norm_obs_; }
    bool norm_reward() const { return norm_reward_; }

private:
    Mat normalize_observation(const Mat& obs) {
        if (!norm_obs_) return obs;
        Mat out = obs;
        check(ppo_norm_obs(h_, obs.data(), get_num_envs(), training_ ? 1 : 0, out.data()));
        return out;
    }
    void write_stats(nlohmann::json& j, int which, int dim) {
        std::vector<float> mean(dim), var(dim);
        double count = 0;
        check(ppo_norm_get_stats(h_, which, mean.data(), var.data(), &count));
        j["var"] = var; j["mean"] = mean; j["count"] = count;
    }
    void read_stats(nlohmann::json& j, int which) {
        const std::vector<float> var = j["var"].get<std::vector<float>>(), mean = j["mean"].get<std::vector<float>>();
        check(ppo_norm_set_stats(h_, which, mean.data(), var.data(), j["count"].get<double>()));
    }
    void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("EnvNormalize: ") + ppo_last_error(h_)); }

    std::unique_ptr<Env> env_;
    ppo_handle* h_;
    bool training_, norm_obs_, norm_reward_;
};
