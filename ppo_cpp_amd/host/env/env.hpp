// env.hpp -- the environment-side boundary of the hot path, kept identical to the reference's abstract interface
// (reference env/env.hpp:16-56) so existing environments drop in: every method, its name, argument and return type
// match; `Mat` is the same row-major float matrix (host/mat.hpp).
#pragma once
// the reference's own include guard as well: a translation unit that pulls the reference's env/env.hpp first (through one
// of its environment headers) keeps that definition, and this file becomes a no-op
#ifndef PPO_CPP_ENV_HPP
#define PPO_CPP_ENV_HPP
#include <cassert>
#include <string>
#include <vector>

#include "../common/serializable.hpp"
#include "../mat.hpp"

class Env : public virtual ISerializable {
public:
    virtual ~Env() {}

    virtual std::string get_action_space() = 0;
    virtual std::string get_observation_space() = 0;
    virtual int get_action_space_size() = 0;
    virtual int get_observation_space_size() = 0;
    virtual int get_num_envs() { return 1; }

    virtual Mat reset() = 0;
    // returns {observations [n_envs, obs], rewards [n_envs, 1], dones [n_envs, 1]}
    virtual std::vector<Mat> step(const Mat& actions) = 0;

    virtual void render() = 0;
    virtual float get_time() = 0;
    virtual Mat get_original_obs() = 0;
    virtual Mat get_original_rew() = 0;

protected:
    template <class T>
    static constexpr const T& clamp(const T& v, const T& lo, const T& hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }

public:
    // spelling as in the reference (env/env.hpp:58-59); inline so the header can be included from several TUs
    static const std::string& space_continuous() { static const std::string s = "continous"; return s; }
    static const std::string& space_discrete() { static const std::string s = "discrete"; return s; }
    const static std::string SPACE_CONTINOUS;
    const static std::string SPACE_DISCRETE;
};
inline const std::string Env::SPACE_CONTINOUS = "continous";
inline const std::string Env::SPACE_DISCRETE = "discrete";
#endif  // PPO_CPP_ENV_HPP
