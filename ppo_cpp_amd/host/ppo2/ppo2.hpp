// ppo2.hpp -- the PPO2 host algorithm with the reference's surface (ppo2/ppo2.hpp:31-547): constructor arguments,
// learn(), eval(), _train_step() and the "fps,pg_loss,vf_loss,entropy,approxkl,clipfrac," line per update.
//
// Differences that follow from replacing the TensorFlow graph executor:
//   * the first constructor argument is the libppo_hip handle (network shape + graph-baked constants live in its
//     ppo_config) instead of a graph file name;
//   * learn() keeps the rollout and the whole minibatch-update phase in HBM when the environment is an EnvNormalize
//     (the reference's stack, ppo2.cpp:188-207): ppo_rollout_* per env step, one ppo_update per update;
//     any other Env goes through the literal reference loop (Runner::run, host shuffle, _train_step per minibatch);
//   * the epoch shuffle is a seeded on-device permutation (the reference uses std::random_shuffle over rand() seeded
//     from the clock, ppo2.hpp:288 / ppo2.cpp:159-162, i.e. it is not reproducible either way).
#pragma once
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>

#include "../env/env_normalize.hpp"
#include "checkpoint.hpp"
#include "runner.hpp"
#include "utils.hpp"

class PPO2 {
public:
    PPO2(ppo_handle* handle, Env& env, float gamma = 0.99f, int n_steps = 128, float ent_coef = 0.01f, float learning_rate = 2.5e-4f,
         float vf_coef = 0.5f, float max_grad_norm = 0.5f, float lam = 0.95f, int nminibatches = 4, int noptepochs = 4, float cliprange = 0.2f,
         float cliprange_vf = -1.f, std::string tensorboard_log = "")
        : h_(handle), env_(env), gamma_(gamma), n_steps_(n_steps), ent_coef_(ent_coef), learning_rate_(learning_rate), vf_coef_(vf_coef),
          max_grad_norm_(max_grad_norm), lam_(lam), nminibatches_(nminibatches), noptepochs_(noptepochs), cliprange_(cliprange),
          cliprange_vf_(cliprange_vf), tensorboard_log_(std::move(tensorboard_log)), n_envs_(env.get_num_envs()), num_timesteps_(0),
          act_model_(handle, env.get_action_space_size()), episode_reward_(Mat::Zero(n_envs_, 1)) {
        n_batch_ = n_envs_ * n_steps_;
    }

    struct UpdateLog { int fps; float losses[5]; double collect_ms, update_ms; float mean_reward; };     // mean_reward: the rollout's un-normalised rewards (the learning curve)
    // Data parallel (no reference counterpart): this PPO2 is rank `rank` of `world` processes, one per GPU, whose handles share a
    // communicator (ppo_dist_init, see dist.hpp).  `env` holds THIS rank's share of the environments; n_batch, total_timesteps, the fps of the
    // CSV line and the checkpoint's n_envs are job-wide quantities.  The replicas stay bit-identical, so rank 0 alone prints and saves
    // (replica_saves: the others also save, under <path>.rank<r>, for the tests that compare them byte for byte).
    void set_distributed(int world, int rank) { world_ = world < 1 ? 1 : world; rank_ = rank; if (rank_ != 0) quiet = true; }
    int world() const { return world_; }
    int rank() const { return rank_; }
    bool replica_saves = false;
    const std::vector<UpdateLog>& history() const { return history_; }
    std::vector<std::pair<int, float>>& episode_rewards() { return episodes_; }
    bool quiet = false;
    unsigned long long seed = 0;
    // parity runs (tests): explicit exploration noise [n_updates][n_steps][n_envs][A] and epoch permutations
    // [n_updates][noptepochs][n_batch] (out.row(perm[i]) = in.row(i), ppo2.hpp:291-296) replace the generators of both loops
    const float* explicit_noise = nullptr;
    const int32_t* explicit_perms = nullptr;
    // host-Env collect split, summed over the updates after the second (the first pays allocation + graph capture, the first
    // env step after that capture a one-off runtime hiccup)
    double phase_env_ms = 0, phase_act_ms = 0, phase_observe_ms = 0;

    // Checkpoint in the reference's on-disk format (ppo2.hpp:107-166): the 15 model tensors as a TF bundle
    // (<path>[.<id>].index / .data-00000-of-00001, names "model/<tensor>"; the untrained q/w, q/b ride along so that the
    // reference's restore_all finds every variable) and the JSON side-car with hyper-parameters + Env::serialize.
    void save(std::string save_path, int save_id = -1) {
        if (rank_ != 0) { if (!replica_saves) return; save_path += ".rank" + std::to_string(rank_); }
        if (save_id >= 0) save_path += "." + std::to_string(save_id);
        ckpt::Bundle b = extra_tensors_;
        const int nt = ppo_num_tensors(h_);
        for (int i = 0; i < nt; ++i) {
            char name[32]; int32_t rows = 0, cols = 0;
            check(ppo_tensor_info(h_, i, name, &rows, &cols));
            ckpt::Tensor t;
            t.shape.push_back(rows); if (cols) t.shape.push_back(cols);
            t.data.resize((size_t)rows * (cols ? cols : 1));
            check(ppo_get_tensor(h_, 0, i, t.data.data(), (int64_t)t.data.size()));
            b[std::string("model/") + name] = t;
        }
        if (!b.count("model/q/w")) {                               // graph variables without gradient
            const int A = env_.get_action_space_size();
            char name[32]; int32_t rows = 0, cols = 0;
            check(ppo_tensor_info(h_, nt - 3, name, &rows, &cols));                  // pi/w: [h_last, A]
            ckpt::Tensor qw; qw.shape = {rows, A}; qw.data.assign((size_t)rows * A, 0.f);
            ckpt::Tensor qb; qb.shape = {A}; qb.data.assign((size_t)A, 0.f);
            b["model/q/w"] = qw; b["model/q/b"] = qb;
        }
        ckpt::save_bundle(save_path, b);
        nlohmann::json json{};
        env_.serialize(json);
        json["gamma"] = gamma_; json["n_steps"] = n_steps_; json["vf_coef"] = vf_coef_; json["ent_coef"] = ent_coef_;
        json["max_grad_norm"] = max_grad_norm_; json["learning_rate"] = learning_rate_; json["lam"] = lam_;
        json["nminibatches"] = nminibatches_; json["noptepochs"] = noptepochs_; json["cliprange"] = cliprange_;
        json["cliprange_vf"] = cliprange_vf_; json["observation_space"] = env_.get_observation_space();
        json["action_space"] = env_.get_action_space(); json["n_envs"] = n_envs_ * world_; json["model_filename"] = model_filename;
        std::ofstream f(save_path + ".json");
        if (!f) throw std::runtime_error("PPO2::save: unable to open " + save_path + ".json");
        f << json.dump();
    }

    // restores weights + hyper-parameters + normaliser statistics (ppo2.hpp:169-223); like the reference it does not
    // restore optimiser state (the graph's saver holds no Adam slots, G:32396-32496)
    void load(const std::string& save_path) {
        nlohmann::json json = nlohmann::json::parse(ckpt::slurp(save_path + ".json"));
        env_.deserialize(json);
        gamma_ = json["gamma"].get<float>(); n_steps_ = json["n_steps"].get<int>(); vf_coef_ = json["vf_coef"].get<float>();
        ent_coef_ = json["ent_coef"].get<float>(); max_grad_norm_ = json["max_grad_norm"].get<float>();
        learning_rate_ = json["learning_rate"].get<float>(); lam_ = json["lam"].get<float>();
        nminibatches_ = json["nminibatches"].get<int>(); noptepochs_ = json["noptepochs"].get<int>();
        cliprange_ = json["cliprange"].get<float>(); cliprange_vf_ = json["cliprange_vf"].get<float>();
        n_batch_ = n_envs_ * n_steps_;
        const ckpt::Bundle b = ckpt::load_bundle(save_path);
        extra_tensors_.clear();
        const int nt = ppo_num_tensors(h_);
        for (int i = 0; i < nt; ++i) {
            char name[32]; int32_t rows = 0, cols = 0;
            check(ppo_tensor_info(h_, i, name, &rows, &cols));
            auto it = b.find(std::string("model/") + name);
            if (it == b.end()) throw std::runtime_error(std::string("PPO2::load: checkpoint lacks model/") + name);
            if ((int64_t)it->second.data.size() != (int64_t)rows * (cols ? cols : 1)) throw std::runtime_error(std::string("PPO2::load: shape mismatch for ") + name);
            check(ppo_set_tensor(h_, 0, i, it->second.data.data(), (int64_t)it->second.data.size()));
        }
        for (const char* q : {"model/q/w", "model/q/b"}) { auto it = b.find(q); if (it != b.end()) extra_tensors_[q] = it->second; }
    }
    std::string model_filename;

    // deterministic action for one observation row (ppo2.hpp:225-237)
    Mat eval(const Mat& obs) { return act_model_.get_deterministic_action(obs); }

    // save cadence of the reference (ppo2.hpp:256-262, 361-376): a save every ceil(n_updates / num_saves) updates with
    // ids 0, 1, ..., plus a trailing save when the interval does not divide the number of updates
    void learn(int total_timesteps, int num_saves = 0, const std::string& save_path = "") {
        num_timesteps_ = 0;
        updates_this_learn_ = 0;
        if (!seeded_ || seeded_with_ != seed) {                    // exploration noise AND epoch shuffles follow PPO2::seed / --seed; a repeated
            check(ppo_seed(h_, seed));                             // learn() with the same seed continues both generators instead of replaying
            seeded_ = true; seeded_with_ = seed;                   // their draws (the shuffle key counts updates since the seed was set)
            shuffle_updates_ = 0; shuffle_rng_.seed((unsigned)seed);
        }
        const int n_updates = total_timesteps / (n_batch_ * world_);
        save_interval_ = num_saves > 0 ? static_cast<int>(std::ceil(static_cast<float>(n_updates) / static_cast<float>(num_saves))) : -1;
        save_path_ = save_path;
        if (num_saves > 0 && save_path.empty()) throw std::runtime_error("PPO2::learn: num_saves > 0 needs a save path");
        if (n_batch_ % nminibatches_ != 0) throw std::runtime_error("PPO2: n_batch must be divisible by nminibatches");
        EnvNormalize* nz = dynamic_cast<EnvNormalize*>(&env_);
        if (nz && nz->training()) learn_resident(*nz, n_updates);
        else if (world_ > 1) throw std::runtime_error("PPO2::learn: data parallel needs the HBM-resident loop (an EnvNormalize in training mode): the literal loop's "
                                                      "host-side advantage normalisation sees one rank's rows only");
        else learn_reference_loop(n_updates);
        if (num_saves > 0 && save_interval_ > 0 && (n_updates % save_interval_) != 0) save(save_path, n_updates / save_interval_);
    }

    // PPO2::_train_step (ppo2.hpp:380-471): advantage normalisation over the minibatch, then the train op
    Mat _train_step(float lr, float cliprange, const Mat& obs, const Mat& returns, const Mat& /*masks*/, const Mat& actions, const Mat& values,
                    const Mat& neglogpacs) {
        const int n = static_cast<int>(obs.rows());
        Mat advs(n, 1), losses(1, 5);
        check(ppo_adv_normalize(h_, returns.data(), values.data(), n, advs.data()));
        check(ppo_train_step(h_, lr, cliprange, obs.data(), actions.data(), advs.data(), returns.data(), neglogpacs.data(), values.data(), n,
                             losses.data()));
        return losses;
    }

private:
    using clk = std::chrono::steady_clock;
    static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

    void learn_resident(EnvNormalize& nz, int n_updates) {
        Env& raw = nz.inner();
        const int E = n_envs_, T = n_steps_;
        check(ppo_rollout_alloc(h_, E, T));
        check(ppo_rollout_reset(h_, raw.reset().data()));                  // EnvNormalize::reset + Runner ctor
        // the logger's env-major [E, T] views are filled once per update from time-major rows (a row per env step is one copy; writing a
        // column of an [E, T] matrix touches E cache lines)
        Mat actions(E, raw.get_action_space_size()), rew_view(E, T), done_view(E, T), rew_tm(T, E), done_tm(T, E), dones = Mat::Zero(E, 1);
        for (int update = 1; update <= n_updates; ++update) {
            const auto t0 = clk::now();
            for (int t = 0; t < T; ++t) {
                std::memcpy(done_tm.data() + (size_t)t * E, dones.data(), sizeof(float) * (size_t)E);
                const auto p0 = clk::now();
                const float* eps = explicit_noise ? explicit_noise + ((size_t)(update - 1) * T + t) * E * actions.cols() : nullptr;
                check(ppo_rollout_act(h_, t, eps, actions.data()));
                const auto p1 = clk::now();
                std::vector<Mat> r = raw.step(actions);
                const auto p2 = clk::now();
                check(ppo_rollout_observe(h_, t, r[0].data(), r[1].data(), r[2].data()));
                const auto p3 = clk::now();
                if (update > 2) { phase_act_ms += ms(p0, p1); phase_env_ms += ms(p1, p2); phase_observe_ms += ms(p2, p3); }
                dones = std::move(r[2]);
                const Mat orig = raw.get_original_rew();
                std::memcpy(rew_tm.data() + (size_t)t * E, orig.data(), sizeof(float) * (size_t)E);
            }
            for (int e = 0; e < E; ++e)
                for (int t = 0; t < T; ++t) { rew_view(e, t) = rew_tm(t, e); done_view(e, t) = done_tm(t, e); }
            check(ppo_rollout_finish(h_, gamma_, lam_));
            const auto t1 = clk::now();
            num_timesteps_ += n_batch_ * world_;
            UpdateLog log{};
            const int32_t* perms = explicit_perms ? explicit_perms + (size_t)(update - 1) * noptepochs_ * n_batch_ : nullptr;
            // every rank shuffles its OWN rows (include/ppo_hip.h, ppo_dist_global_shuffle 0): the rank is part of the key
            check(ppo_update(h_, learning_rate_, cliprange_, noptepochs_, nminibatches_, perms,
                             seed + (unsigned long long)(++shuffle_updates_) + ((unsigned long long)rank_ << 40), nullptr, log.losses));
            const auto t2 = clk::now();
            finish_update(log, t0, t1, t2, rew_view, done_view);
        }
    }

    void learn_reference_loop(int n_updates) {
        Runner runner{env_, act_model_, n_steps_, gamma_, lam_};
        std::mt19937& rng = shuffle_rng_;                           // (a member: reset only when the seed changes, see learn())
        const int batch_size = n_batch_ / nminibatches_;
        for (int update = 1; update <= n_updates; ++update) {
            const auto t0 = clk::now();
            runner.noise = explicit_noise ? explicit_noise + (size_t)(update - 1) * n_batch_ * env_.get_action_space_size() : nullptr;
            const MiniBatch mb = runner.run();
            const auto t1 = clk::now();
            const auto all = mb.get_train_input();
            std::vector<int> perm(n_batch_);
            std::iota(perm.begin(), perm.end(), 0);
            num_timesteps_ += n_batch_;
            double acc[5] = {0, 0, 0, 0, 0};
            for (int epoch = 0; epoch < noptepochs_; ++epoch) {
                if (explicit_perms) std::memcpy(perm.data(), explicit_perms + ((size_t)(update - 1) * noptepochs_ + epoch) * n_batch_, sizeof(int) * (size_t)n_batch_);
                else std::shuffle(perm.begin(), perm.end(), rng);            // cumulative, like ppo2.hpp:288
                std::vector<Mat> shuffled;
                for (const auto& v : all) {                                  // out.row(perm[i]) = in.row(i)  (ppo2.hpp:291-296)
                    Mat o(v->rows(), v->cols());
                    for (int i = 0; i < n_batch_; ++i) mat_set_row(o, perm[i], mat_row_ptr(*v, i));
                    shuffled.push_back(o);
                }
                for (int start = 0; start < n_batch_; start += batch_size) {
                    std::vector<Mat> sl;
                    for (const Mat& v : shuffled) {
                        Mat s(batch_size, v.cols());
                        std::memcpy(s.data(), mat_row_ptr(v, start), sizeof(float) * (size_t)batch_size * v.cols());
                        sl.push_back(s);
                    }
                    const Mat l = _train_step(learning_rate_, cliprange_, sl[0], sl[1], sl[2], sl[3], sl[4], sl[5]);
                    for (int j = 0; j < 5; ++j) acc[j] += l(0, j);
                }
            }
            const auto t2 = clk::now();
            UpdateLog log{};
            for (int j = 0; j < 5; ++j) log.losses[j] = (float)(acc[j] / (noptepochs_ * nminibatches_));   // colwise().mean() (ppo2.hpp:335)
            Mat rew_view(n_envs_, n_steps_), done_view(n_envs_, n_steps_);
            std::memcpy(rew_view.data(), mb.unnormalized_rewards->data(), sizeof(float) * (size_t)n_batch_);
            std::memcpy(done_view.data(), mb.dones->data(), sizeof(float) * (size_t)n_batch_);
            finish_update(log, t0, t1, t2, rew_view, done_view);
        }
    }

    void finish_update(UpdateLog& log, clk::time_point t0, clk::time_point t1, clk::time_point t2, const Mat& rew_view, const Mat& done_view) {
        log.collect_ms = ms(t0, t1);
        log.update_ms = ms(t1, t2);
        { double acc = 0; const long cnt = (long)rew_view.rows() * rew_view.cols(); for (long i = 0; i < cnt; ++i) acc += rew_view.data()[i]; log.mean_reward = (float)(acc / (double)std::max(cnt, 1l)); }
        const double total = std::max(ms(t0, t2), 1e-3);
        // ppo2.hpp:337-341; data parallel: the job's env steps over THIS rank's wall time (the ranks meet in a collective every minibatch,
        // so their update times differ by less than one exchange)
        log.fps = static_cast<int>((double)n_batch_ * world_ * 1000.0 / total);
        if (!quiet) {
            std::printf("%d,", log.fps);
            for (int i = 0; i < 5; ++i) std::printf("%g,", log.losses[i]);
            std::printf("\n");
        }
        episode_reward_ = Utils::total_episode_reward_logger(
            episode_reward_, rew_view, done_view, [this](int step, const char*, float v) { episodes_.push_back({step, v}); }, num_timesteps_ - n_batch_ * world_);
        history_.push_back(log);
        const int update = ++updates_this_learn_;                   // save ids / cadence count from the start of THIS learn() call
        if (save_interval_ > 0 && update % save_interval_ == 0) save(save_path_, update / save_interval_ - 1);
    }

    void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("PPO2: ") + ppo_last_error(h_)); }

    ppo_handle* h_;
    Env& env_;
    int world_ = 1, rank_ = 0;
    float gamma_;
    int n_steps_;
    float ent_coef_, learning_rate_, vf_coef_, max_grad_norm_, lam_;
    int nminibatches_, noptepochs_;
    float cliprange_, cliprange_vf_;
    std::string tensorboard_log_;
    int n_envs_, n_batch_, num_timesteps_;
    MlpPolicy act_model_;
    Mat episode_reward_;
    int save_interval_ = -1;
    int updates_this_learn_ = 0;
    bool seeded_ = false; unsigned long long seeded_with_ = 0;
    unsigned long long shuffle_updates_ = 0;                       // updates since the seed was set: key of the on-device epoch shuffles
    std::mt19937 shuffle_rng_;                                     // the literal loop's std::shuffle generator (ppo2.hpp:288)
    std::string save_path_;
    ckpt::Bundle extra_tensors_;      // q/w, q/b carried through load -> save
    std::vector<UpdateLog> history_;
    std::vector<std::pair<int, float>> episodes_;
};
