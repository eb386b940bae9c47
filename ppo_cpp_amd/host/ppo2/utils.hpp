// utils.hpp -- the episode-reward logger of the reference (ppo2/utils.hpp:75-114) without TensorFlow: finished
// episodes are reported through a callback (step, reward) instead of a TensorBoard event file.
#pragma once
#include <functional>

#include "../mat.hpp"

class Utils {
public:
    using ScalarSink = std::function<void(int step, const char* tag, float value)>;

    // rew_acc [n_envs,1] carries the running episode reward across updates; rewards / dones are [n_envs, n_steps]
    // (env-major views of the rollout).  A done flag at step k ends the episode BEFORE reward k is counted, exactly
    // like the reference: rewards [prev_done, k) are summed, then the accumulator restarts at k.
    static Mat total_episode_reward_logger(Mat rew_acc, const Mat& rewards, const Mat& dones, const ScalarSink& sink, int total_steps) {
        assert(rew_acc.rows() == rewards.rows() && rewards.rows() == dones.rows() && rewards.cols() == dones.cols());
        const int steps = static_cast<int>(rewards.cols());
        for (long e = 0; e < rew_acc.rows(); ++e) {
            int start = 0;
            bool any = false;
            for (int k = 0; k < steps; ++k) {
                if (dones(e, k) > .5f) {
                    float s = any ? 0.f : rew_acc(e, 0);
                    for (int j = start; j < k; ++j) s += rewards(e, j);
                    if (sink) sink(total_steps + k, "episode_reward", s);
                    start = k;
                    any = true;
                }
            }
            float s = any ? 0.f : rew_acc(e, 0);
            for (int j = start; j < steps; ++j) s += rewards(e, j);
            rew_acc(e, 0) = s;
        }
        return rew_acc;
    }
};
