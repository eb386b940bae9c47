// ppo_host.cpp -- C entry points over the C++ host layer (env stack, Runner, PPO2) so that tests and bench.py can
// drive it with ctypes.  The wiring of ppo_host_learn mirrors the reference's main() (ppo2.cpp:188-250):
//   N x Env -> VecEnv -> EnvNormalize{training} -> PPO2{gamma .99, lam .95, vf .5, max_grad_norm .5, 32 minibatches}.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

#include "env/env_mock.hpp"
#include "env/env_normalize.hpp"
#include "env/vec_env.hpp"
#include "ppo2/checkpoint.hpp"
#include "ppo2/graph_spec.hpp"
#include "ppo2/ppo2.hpp"

extern "C" {

// Port of the reference's only test (test/vecenv_test.cpp:13-49): N x EnvMock(i+1) behind a VecEnv, `steps` steps of
// zero actions; every observation column and the reward column must equal [1..N]^T.  Returns 0 or the failing check.
int ppo_host_vecenv_check(int num_envs, int steps, int max_workers) {
    std::vector<std::shared_ptr<Env>> envs;
    Mat expect = Mat::Zero(num_envs, 1);
    for (int i = 0; i < num_envs; ++i) { envs.push_back(std::make_shared<EnvMock>(i + 1)); expect(i, 0) = (float)(i + 1); }
    VecEnv ve{envs, max_workers};
    for (int s = 0; s < steps; ++s) {
        const Mat actions = Mat::Zero(ve.get_num_envs(), ve.get_action_space_size());
        const std::vector<Mat> result = ve.step(actions);
        const Mat& obs = result[0];
        const Mat& rew = result[1];
        if (obs.rows() != ve.get_num_envs()) return 1;
        if (rew.rows() != ve.get_num_envs()) return 2;
        if (obs.cols() != ve.get_observation_space_size()) return 3;
        if (rew.cols() != 1) return 4;
        if ((rew - expect).squaredNorm() > 1e-2f) return 5;
        for (int j = 0; j < ve.get_observation_space_size(); ++j)
            if ((obs.col(j) - expect).squaredNorm() > 1e-2f) return 6;
        const Mat dones = result[2];
        for (int i = 0; i < num_envs; ++i) if (dones(i, 0) != ((s + 1) % 300 == 0 ? 1.f : 0.f)) return 7;
        if ((ve.get_original_rew() - expect).squaredNorm() > 1e-2f) return 8;
    }
    const Mat r = ve.reset();                         // gathers get_original_obs(), does not reset sub-envs
    if (r.rows() != num_envs || r.cols() != 18) return 9;
    for (int i = 0; i < num_envs; ++i) if (r(i, 0) != (float)(i + 1)) return 10;
    return 0;
}

// pure host checks of the small utilities (no GPU): Mat shim, JSON, episode logger, env-major flatten
int ppo_host_selftest() {
    {   // flatten: [T,E,W] -> row e*T+t
        const int T = 3, E = 2, W = 2;
        float src[T * E * W];
        for (int i = 0; i < T * E * W; ++i) src[i] = (float)i;
        auto m = Runner::flatten(src, T, E, W);
        for (int e = 0; e < E; ++e) for (int t = 0; t < T; ++t) for (int w = 0; w < W; ++w)
            if ((*m)(e * T + t, w) != src[(t * E + e) * W + w]) return 1;
    }
    {   // episode logger: env 0 has a done at k=2 -> episode reward = acc + r0 + r1, new accumulator r2 + r3
        Mat acc = Mat::Zero(2, 1), rew(2, 4), dn = Mat::Zero(2, 4);
        acc(0, 0) = 10.f;
        for (int e = 0; e < 2; ++e) for (int k = 0; k < 4; ++k) rew(e, k) = (float)(k + 1);
        dn(0, 2) = 1.f;
        std::vector<std::pair<int, float>> got;
        acc = Utils::total_episode_reward_logger(acc, rew, dn, [&](int s, const char*, float v) { got.push_back({s, v}); }, 100);
        if (got.size() != 1 || got[0].first != 102 || got[0].second != 13.f) return 2;
        if (acc(0, 0) != 7.f || acc(1, 0) != 10.f) return 3;
    }
    {   // JSON round trip in the reference's running-statistics format
        nlohmann::json j;
        j["obs_rms"]["mean"] = std::vector<float>{0.5f, -1.25f};
        j["obs_rms"]["count"] = 72001473.000001;
        const std::string text = j.dump();
#ifndef PPO_HAVE_NLOHMANN
        nlohmann::json k = nlohmann::json::parse(text);
        if (k["obs_rms"]["mean"].get<std::vector<float>>()[1] != -1.25f) return 4;
        if (std::fabs(k["obs_rms"]["count"].get<double>() - 72001473.000001) > 1e-6) return 5;
#endif
    }
    {   // seeded mock: deterministic, bounded
        SeededEnvMock a(1234, 7), b(1234, 7);
        const Mat o1 = a.reset(), o2 = b.reset();
        for (int j = 0; j < 18; ++j) if (o1(0, j) != o2(0, j) || o1(0, j) < -1.f || o1(0, j) >= 1.f) return 6;
    }
    return 0;
}

// read a reference checkpoint (<in_prefix>.index / .data-00000-of-00001) and write it back under out_prefix; returns the
// number of tensors, or -1 (message on stderr).  tests/test_checkpoint.py compares the output files byte for byte.
int ppo_host_bundle_roundtrip(const char* in_prefix, const char* out_prefix) {
    try {
        const ckpt::Bundle b = ckpt::load_bundle(in_prefix);
        ckpt::save_bundle(out_prefix, b);
        return (int)b.size();
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}

// tensor access for tests: copies tensor `name` (at most cap floats) and its shape (up to 4 dims); returns element count
int ppo_host_bundle_tensor(const char* prefix, const char* name, float* dst, int cap, long long shape[4]) {
    try {
        const ckpt::Bundle b = ckpt::load_bundle(prefix);
        auto it = b.find(name);
        if (it == b.end()) return -2;
        const int n = (int)it->second.data.size();
        if (n > cap) return -3;
        std::memcpy(dst, it->second.data.data(), sizeof(float) * (size_t)n);
        for (int i = 0; i < 4; ++i) shape[i] = i < (int)it->second.shape.size() ? it->second.shape[i] : 0;
        return n;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}

// PPO2::load of a reference checkpoint ([4,5] net, EnvMock behind EnvNormalize) -> deterministic action + value of the
// zero observation -> PPO2::save under out_prefix.  Returns 0; mu[18], value[1], obs_count out.
int ppo_host_checkpoint_eval(const char* in_prefix, const char* out_prefix, float* mu, double* obs_count) {
    ppo_handle* h = nullptr;
    try {
        ppo_config cfg; const int32_t hidden[2] = {4, 5};
        ppo_config_default(&cfg, 18, 18, 2, hidden);
        if (ppo_create(&cfg, &h) != 0) throw std::runtime_error(ppo_last_error(nullptr));
        {
            EnvNormalize env{std::unique_ptr<Env>(new EnvMock(1)), h, /*training=*/false};
            PPO2 algo{h, env};
            algo.load(in_prefix);
            const Mat a = algo.eval(Mat::Zero(1, 18));
            std::memcpy(mu, a.data(), sizeof(float) * 18);
            float m[18], v[18];
            if (ppo_norm_get_stats(h, 0, m, v, obs_count) != 0) throw std::runtime_error(ppo_last_error(h));
            algo.save(out_prefix);
        }
        ppo_destroy(h);
        return 0;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); if (h) ppo_destroy(h); return -1; }
}

// Graph-spec importer: parses a reference .meta.txt; fills cfg, the betas' initial powers, and copies
// variable `name`'s initial value (cap floats).  Returns the element count, -2 if absent, -1 on parse errors.
int ppo_host_graph_spec(const char* path, ppo_config* cfg, float pw0[2], const char* name, float* dst, int cap) {
    try {
        const graphspec::GraphSpec g = graphspec::load_graph_spec(path);
        *cfg = g.config; pw0[0] = g.beta1_power0; pw0[1] = g.beta2_power0;
        if (!name || !name[0]) return 0;
        auto it = g.initial.find(name);
        if (it == g.initial.end()) return -2;
        const int n = (int)it->second.data.size();
        if (n > cap) return -3;
        std::memcpy(dst, it->second.data.data(), sizeof(float) * (size_t)n);
        return n;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
}

// load_graph + Run("init") on the DEVICE (session_creator.hpp:40-58): create a handle from a graph file, assign the graph's
// initial weights, and evaluate deterministic action [n,A], value [n] and the beta powers for the given observations.
int ppo_host_graph_eval(const char* path, const float* obs, int n, float* actions, float* values, float pw[2]) {
    ppo_handle* h = nullptr;
    try {
        const graphspec::GraphSpec g = graphspec::load_graph_spec(path);
        h = graphspec::create_from_graph(g);
        if (ppo_act_deterministic(h, obs, n, actions) != 0 || ppo_value(h, obs, n, values) != 0 || ppo_get_beta_powers(h, pw) != 0)
            throw std::runtime_error(ppo_last_error(h));
        ppo_destroy(h);
        return 0;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); if (h) ppo_destroy(h); return -1; }
}

struct ppo_host_args {
    int n_envs, n_steps, n_hidden, hidden[8];
    int nminibatches, noptepochs, n_updates;
    float lr, cliprange, gamma, lam;
    int seeded_env;          // 0: EnvMock(i+1) (degenerate constant data, the reference's stub) ; 1: SeededEnvMock ; 2: TargetEnv (learnable: tests/test_learning.py)
    int device;
    int max_workers;
    int reference_loop;      // 1: force the literal reference loop (Runner::run + host shuffle + _train_step)
    int norm_obs, norm_reward;   // EnvNormalize constructor flags (env_normalize.hpp:24-27)
    unsigned long long seed;     // PPO2::seed (exploration noise + epoch shuffles)
    int obs_dim, act_dim;        // SeededEnvMock's shape (0 = 18): 36 / 18 is the hexapod with observed velocities (hexapod_closed_loop_env.hpp:20)
};
struct ppo_host_result {
    double env_steps_per_s, collect_ms, update_ms;
    float losses[5];
    int fps_last;
    char error[256];
    double obs_count, ret_count;     // running-statistics counts after the run (1e-6 = never updated)
    double phase_env_ms, phase_act_ms, phase_observe_ms;   // host-Env collect split per update: Env::step | ppo_rollout_act (kernel + D2H + sync) | ppo_rollout_observe (pack + H2D enqueue)
    int pool_workers, pool_chunk, pool_active;              // what the pooled VecEnv settled on: threads incl. the caller, environments per claimed chunk, threads that took part in the last step
};

// explicit inputs / extra outputs of a parity run (all optional)
struct ppo_host_explicit {
    const float* theta_in;       // [P] dense initial weights (null: ppo_init_orthogonal(0))
    const float* noise;          // [n_updates][n_steps][n_envs][A]
    const int32_t* perms;        // [n_updates][noptepochs][n_batch]
    float* losses_out;           // [n_updates][5] mean losses of every update (ppo2.hpp:335)
    float* theta_out;            // [P]
    float* obs_mean; float* obs_var; double* obs_count;      // obs_rms [18], [18], [1]
    float* ret_mean; float* ret_var; double* ret_count;      // ret_rms [1], [1], [1]
    float* reward_curve;         // [n_updates] mean un-normalised reward of every update's rollout
};

static int run_learn(const ppo_host_args* a, ppo_host_result* out, const ppo_host_explicit* x) {
    std::memset(out, 0, sizeof *out);
    ppo_handle* h = nullptr;
    try {
        ppo_config cfg;
        const int O = a->obs_dim > 0 ? a->obs_dim : 18, A = a->act_dim > 0 ? a->act_dim : 18;
        if ((O != 18 || A != 18) && !a->seeded_env) throw std::runtime_error("EnvMock (the reference's stub) is 18 / 18");
        ppo_config_default(&cfg, O, A, a->n_hidden, a->hidden);
        cfg.device = a->device;
        if (ppo_create(&cfg, &h) != 0) throw std::runtime_error(ppo_last_error(nullptr));
        if (ppo_init_orthogonal(h, 0) != 0) throw std::runtime_error(ppo_last_error(h));
        if (x && x->theta_in && ppo_set_flat(h, 0, x->theta_in, ppo_num_params(h)) != 0) throw std::runtime_error(ppo_last_error(h));
        std::vector<std::shared_ptr<Env>> envs;
        for (int i = 0; i < a->n_envs; ++i) {
            if (a->seeded_env == 2) envs.push_back(std::make_shared<TargetEnv>(1234u, (uint32_t)i, O, A));
            else if (a->seeded_env) envs.push_back(std::make_shared<SeededEnvMock>(1234u, (uint32_t)i, O, A));
            else envs.push_back(std::make_shared<EnvMock>(i + 1));
        }
        std::unique_ptr<Env> inner;
        VecEnv* pool = nullptr;
        if (a->n_envs > 1) inner.reset(pool = new VecEnv(envs, a->max_workers));
        else inner.reset(a->seeded_env == 2 ? static_cast<Env*>(new TargetEnv(1234u, 0, O, A)) : a->seeded_env ? static_cast<Env*>(new SeededEnvMock(1234u, 0, O, A)) : static_cast<Env*>(new EnvMock(1)));
        {
            EnvNormalize env{std::move(inner), h, /*training=*/true, a->norm_obs != 0, a->norm_reward != 0, 10.f, 10.f, a->gamma};
            PPO2 algorithm{h, env, a->gamma, a->n_steps, cfg.ent_coef, a->lr, 0.5f, 0.5f, a->lam, a->nminibatches, a->noptepochs, a->cliprange};
            algorithm.quiet = true;
            struct Plain : Env {       // hides the EnvNormalize type to force the reference loop
                Env& e; explicit Plain(Env& x) : e(x) {}
                std::string get_action_space() override { return e.get_action_space(); }
                std::string get_observation_space() override { return e.get_observation_space(); }
                int get_action_space_size() override { return e.get_action_space_size(); }
                int get_observation_space_size() override { return e.get_observation_space_size(); }
                int get_num_envs() override { return e.get_num_envs(); }
                Mat reset() override { return e.reset(); }
                std::vector<Mat> step(const Mat& x) override { return e.step(x); }
                void render() override {}
                float get_time() override { return 0; }
                Mat get_original_obs() override { return e.get_original_obs(); }
                Mat get_original_rew() override { return e.get_original_rew(); }
                void serialize(nlohmann::json& j) override { e.serialize(j); }
                void deserialize(nlohmann::json& j) override { e.deserialize(j); }
            } plain{env};
            PPO2 literal{h, plain, a->gamma, a->n_steps, cfg.ent_coef, a->lr, 0.5f, 0.5f, a->lam, a->nminibatches, a->noptepochs, a->cliprange};
            literal.quiet = true;
            PPO2& algo = a->reference_loop ? literal : algorithm;
            algo.seed = a->seed;
            if (x) { algo.explicit_noise = x->noise; algo.explicit_perms = x->perms; }
            const auto t0 = std::chrono::steady_clock::now();
            algo.learn(a->n_updates * a->n_envs * a->n_steps);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const auto& hist = algo.history();
            if (hist.empty()) throw std::runtime_error("no update ran");
            // the first update pays allocation + graph capture, and the first env step after that capture pays a one-off
            // ~8 ms in the runtime: report the steady state (from the third update on) when there is one
            size_t from = hist.size() > 2 ? 2 : hist.size() > 1 ? 1 : 0;
            double c = 0, u = 0;
            for (size_t i = from; i < hist.size(); ++i) { c += hist[i].collect_ms; u += hist[i].update_ms; }
            const double n = (double)(hist.size() - from);
            out->collect_ms = c / n; out->update_ms = u / n;
            out->env_steps_per_s = (double)a->n_envs * a->n_steps / ((c + u) / n / 1e3);
            std::memcpy(out->losses, hist.back().losses, sizeof out->losses);
            out->fps_last = hist.back().fps;
            (void)sec;
            out->phase_env_ms = algo.phase_env_ms / n; out->phase_act_ms = algo.phase_act_ms / n; out->phase_observe_ms = algo.phase_observe_ms / n;
            if (pool) { out->pool_workers = pool->pool_workers(); out->pool_chunk = pool->pool_chunk(); out->pool_active = pool->pool_active(); }
            nlohmann::json j;                                   // serialise round trip of the normaliser
            env.serialize(j);
            out->obs_count = j["obs_rms"]["count"].get<double>(); out->ret_count = j["ret_rms"]["count"].get<double>();
            env.deserialize(j);
            if (x) {
                if (x->losses_out) for (size_t i = 0; i < hist.size(); ++i) std::memcpy(x->losses_out + 5 * i, hist[i].losses, sizeof(float) * 5);
                if (x->reward_curve) for (size_t i = 0; i < hist.size(); ++i) x->reward_curve[i] = hist[i].mean_reward;
                if (x->theta_out && ppo_get_flat(h, 0, x->theta_out, ppo_num_params(h)) != 0) throw std::runtime_error(ppo_last_error(h));
                if (x->obs_mean && ppo_norm_get_stats(h, 0, x->obs_mean, x->obs_var, x->obs_count) != 0) throw std::runtime_error(ppo_last_error(h));
                if (x->ret_mean && ppo_norm_get_stats(h, 1, x->ret_mean, x->ret_var, x->ret_count) != 0) throw std::runtime_error(ppo_last_error(h));
            }
        }
        ppo_destroy(h);
        return 0;
    } catch (const std::exception& e) {
        std::snprintf(out->error, sizeof out->error, "%s", e.what());
        if (h) ppo_destroy(h);
        return -1;
    }
}

int ppo_host_learn(const ppo_host_args* a, ppo_host_result* out) { return run_learn(a, out, nullptr); }

// PPO2::learn with EXPLICIT exploration noise and epoch permutations on the reference's stack (SeededEnvMock x N -> VecEnv ->
// EnvNormalize -> PPO2; ppo2.cpp:188-250), through the HBM-resident loop or (reference_loop) the literal one: what
// tests/test_host_layer.py holds against oracle.collect + oracle.update update by update (ppo2.hpp:264-349).
int ppo_host_learn_explicit(const ppo_host_args* a, const ppo_host_explicit* x, ppo_host_result* out) { return run_learn(a, out, x); }

}  // extern "C"
