// policies.hpp -- MlpPolicy with the reference's method names (ppo2/policies.hpp:26-82).  The reference wraps
// tensorflow::Session::Run with fixed feed/fetch names; this one wraps the C-ABI of libppo_hip (include/ppo_hip.h).
// Results come back as Mats: step -> {actions [n,A], values [n,1], neglogps [n,1]}.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ppo_hip.h"
#include "../mat.hpp"

class MlpPolicy {
public:
    explicit MlpPolicy(ppo_handle* handle, int act_dim) : h_(handle), act_dim_(act_dim) {}
    virtual ~MlpPolicy() {}

    // output/_action, output/_value_flat, output/_neglogp  (policies.hpp:33-46); noise == nullptr: on-device RNG
    virtual std::vector<Mat> step(const Mat& obs, const Mat* noise = nullptr) {
        const int n = static_cast<int>(obs.rows());
        Mat a(n, act_dim_), v(n, 1), nlp(n, 1);
        check(ppo_step(h_, obs.data(), n, noise ? noise->data() : nullptr, a.data(), v.data(), nlp.data()), "step");
        return {a, v, nlp};
    }
    // output/_deterministic_action (policies.hpp:49-62)
    virtual Mat get_deterministic_action(const Mat& obs) {
        Mat a(obs.rows(), act_dim_);
        check(ppo_act_deterministic(h_, obs.data(), static_cast<int>(obs.rows()), a.data()), "get_action()");
        return a;
    }
    // output/_value_flat (policies.hpp:64-77)
    virtual Mat value(const Mat& obs) {
        Mat v(obs.rows(), 1);
        check(ppo_value(h_, obs.data(), static_cast<int>(obs.rows()), v.data()), "value()");
        return v;
    }
    // Runner::set_returns' GAE scan (runner.hpp:159-191) on the device; [T,E] time-major
    virtual void gae(const Mat& rewards, const Mat& values, const Mat& dones, const Mat& last_values, const Mat& last_dones, float gamma,
                     float lam, Mat& returns) {
        check(ppo_gae(h_, rewards.data(), values.data(), dones.data(), last_values.data(), last_dones.data(), static_cast<int>(rewards.rows()),
                      static_cast<int>(rewards.cols()), gamma, lam, returns.data()), "gae");
    }
    ppo_handle* handle() const { return h_; }

protected:
    MlpPolicy() : h_(nullptr), act_dim_(0) {}
    void check(int rc, const char* what) { if (rc != 0) throw std::runtime_error(std::string(what) + " error: " + ppo_last_error(h_)); }
    ppo_handle* h_;
    int act_dim_;
};
