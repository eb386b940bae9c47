// runner.hpp -- rollout collector with the reference's Runner / MiniBatch surface (ppo2/runner.hpp:21-204).
//
// run() is the drop-in path: an arbitrary host Env, one MlpPolicy::step per env step, GAE and the env-major flatten
// (row = env * n_steps + t, runner.hpp:136-152) -- every array returned as a Mat exactly like the reference.  GAE
// runs on the device (ppo_gae).  PPO2::learn uses the device-resident variant (ppo_rollout_*), which keeps the same
// buffers in HBM and never flattens.
#pragma once
#include <memory>

#include "../env/env.hpp"
#include "policies.hpp"

struct MiniBatch {
    std::shared_ptr<Mat> obs, returns, dones, actions, values, neglogpacs, true_rewards, unnormalized_rewards;
    std::vector<std::shared_ptr<Mat>> get_train_input() const { return {obs, returns, dones, actions, values, neglogpacs}; }
    std::vector<std::shared_ptr<Mat>> get_1_dims() const { return {returns, dones, values, neglogpacs, true_rewards, unnormalized_rewards}; }
};

class Runner {
public:
    Runner(Env& env, MlpPolicy& model, int n_steps, float gamma, float lam)
        : env_(env), model_(model), n_steps_(n_steps), gamma_(gamma), lam_(lam), obs_(env.reset()), num_envs_(env.get_num_envs()),
          dones_(Mat::Zero(num_envs_, 1)) {}

    MiniBatch run() {
        const int E = num_envs_, T = n_steps_, O = env_.get_observation_space_size(), A = env_.get_action_space_size();
        // time-major staging [T, E, .]
        std::vector<float> obs((size_t)T * E * O), act((size_t)T * E * A);
        Mat values(T, E), neglogp(T, E), dones(T, E), rewards(T, E), raw_rewards(T, E);
        for (int t = 0; t < T; ++t) {
            std::memcpy(&obs[(size_t)t * E * O], obs_.data(), sizeof(float) * (size_t)E * O);
            Mat eps;                                                       // explicit exploration noise of this env step [E, A], if any
            if (noise) { eps = Mat(E, A); std::memcpy(eps.data(), noise + (size_t)t * E * A, sizeof(float) * (size_t)E * A); }
            const std::vector<Mat> s = model_.step(obs_, noise ? &eps : nullptr);
            assert(s[0].rows() == E && s[0].cols() == A && s[1].rows() == E && s[2].rows() == E);
            std::memcpy(&act[(size_t)t * E * A], s[0].data(), sizeof(float) * (size_t)E * A);
            mat_set_row(values, t, s[1].data());
            mat_set_row(neglogp, t, s[2].data());
            mat_set_row(dones, t, dones_.data());                         // the done flag that arrived WITH obs_t (runner.hpp:110)
            const std::vector<Mat> r = env_.step(s[0]);
            assert(r[0].rows() == E && r[0].cols() == O && r[1].rows() == E && r[2].rows() == E);
            obs_ = r[0];
            dones_ = r[2];
            mat_set_row(rewards, t, r[1].data());
            mat_set_row(raw_rewards, t, env_.get_original_rew().data());
        }
        // set_returns (runner.hpp:159-191): bootstrap value of the observation after the last step, GAE on the device
        const Mat last_values = model_.value(obs_);
        Mat returns(T, E);
        model_.gae(rewards, values, dones, last_values, dones_, gamma_, lam_, returns);
        MiniBatch mb;
        mb.obs = flatten(obs.data(), T, E, O);
        mb.actions = flatten(act.data(), T, E, A);
        mb.returns = flatten(returns.data(), T, E, 1);
        mb.dones = flatten(dones.data(), T, E, 1);
        mb.values = flatten(values.data(), T, E, 1);
        mb.neglogpacs = flatten(neglogp.data(), T, E, 1);
        mb.true_rewards = flatten(rewards.data(), T, E, 1);
        mb.unnormalized_rewards = flatten(raw_rewards.data(), T, E, 1);
        return mb;
    }

    // parity runs: [n_steps, n_envs, A] standard-normal draws used instead of the on-device generator (the reference draws from
    // TF's RandomStandardNormal with seed 0, G:5894, i.e. it is not reproducible; SURVEY 7 "noise is an explicit input")
    const float* noise = nullptr;

    const Mat& current_obs() const { return obs_; }
    const Mat& current_dones() const { return dones_; }

    // [T, E, W] time-major -> [E*T, W] with row = e*T + t  (runner.hpp:136-152)
    static std::shared_ptr<Mat> flatten(const float* src, int T, int E, int W) {
        auto m = std::make_shared<Mat>((long)E * T, W);
        for (int t = 0; t < T; ++t)
            for (int e = 0; e < E; ++e) std::memcpy(m->data() + ((size_t)e * T + t) * W, src + ((size_t)t * E + e) * W, sizeof(float) * (size_t)W);
        return m;
    }

private:
    Env& env_;
    MlpPolicy& model_;
    int n_steps_;
    float gamma_, lam_;
    Mat obs_;
    int num_envs_;
    Mat dones_;
};
