// main.cpp -- command-line driver with the training flags of the reference's ppo2.cpp (ppo2.cpp:93-128) for the mock
// environments: builds the same stack (N x Env -> VecEnv -> EnvNormalize -> PPO2, ppo2.cpp:188-217), trains with periodic
// checkpoints in the reference's format, or (--path) loads a checkpoint and plays the deterministic policy.
// Physics (DART hexapod, --closed_loop / --bullet / --duration / --framerate) is out of scope: those flags are rejected.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

#include "env/env_mock.hpp"
#include "env/env_normalize.hpp"
#include "env/vec_env.hpp"
#include "ppo2/graph_spec.hpp"
#include "ppo2/ppo2.hpp"

namespace {
struct Flags {
    std::map<std::string, std::string> kv;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    std::string str(const std::string& k, const std::string& d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    double num(const std::string& k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : atof(it->second.c_str()); }
};
const char* kAliases[][2] = {{"-d", "dir"}, {"--dir", "dir"}, {"-p", "path"}, {"--path", "path"}, {"--id", "id"}, {"-s", "steps"}, {"--steps", "steps"},
                             {"-l", "lr"}, {"--lr", "lr"}, {"--learning_rate", "lr"}, {"-e", "ent"}, {"--ent", "ent"}, {"--entropy", "ent"},
                             {"-c", "cr"}, {"--cr", "cr"}, {"--clip_range", "cr"}, {"--cliprange", "cr"}, {"--saves", "saves"}, {"--num_saves", "saves"},
                             {"--epochs", "epochs"}, {"--num_epochs", "epochs"}, {"--batch_steps", "batch_steps"}, {"--n_steps", "batch_steps"},
                             {"-j", "threads"}, {"--threads", "threads"}, {"--jobs", "threads"}, {"--num_threads", "threads"}, {"--hidden", "hidden"},
                             {"--minibatches", "minibatches"}, {"--seed", "seed"}, {"-g", "graph"}, {"--graph", "graph"}, {"--graph_path", "graph"}, {"--obs", "obs"}};
const char* kSwitches[][2] = {{"-r", "resume"}, {"--resume", "resume"}, {"-v", "verbose"}, {"--verbose", "verbose"}, {"--seeded", "seeded"}};
}  // namespace

int main(int argc, char** argv) {
    Flags f;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        bool ok = false;
        if (a == "-h" || a == "--help") {
            std::printf("usage: ppo_cpp_hip [--steps N] [--lr X] [--ent X] [--cr X] [--epochs N] [--batch_steps N] [--threads N_ENVS] [--minibatches N]\n"
                        "                   [--hidden 256,256] [--saves N --dir DIR --id ID] [--path CKPT_PREFIX] [--resume] [--seeded] [--seed N]\n"
                        "                   [--obs 36   (with --seeded: observation width of the mock environment; 36 = the hexapod that observes its velocities)]\n");
            return 0;
        }
        for (auto& al : kAliases) if (a == al[0] && i + 1 < argc) { f.kv[al[1]] = argv[++i]; ok = true; break; }
        for (auto& sw : kSwitches) if (!ok && a == sw[0]) { f.kv[sw[1]] = "1"; ok = true; break; }
        if (!ok) { std::fprintf(stderr, "unsupported flag %s (physics / visualisation flags of the reference are out of scope)\n", a.c_str()); return 1; }
    }
    std::vector<int32_t> hidden;
    { std::stringstream ss(f.str("hidden", "64,64")); std::string tok; while (std::getline(ss, tok, ',')) hidden.push_back(atoi(tok.c_str())); }
    const int n_envs = (int)f.num("threads", 1), n_steps = (int)f.num("batch_steps", 2048);
    const bool training = !f.has("path") || f.has("resume");                  // ppo2.cpp:171
    ppo_handle* h = nullptr;
    ppo_config cfg;
    const int obs_dim = (int)f.num("obs", 18);                                  // (the reference's closed-loop hexapod: 18, or 36 with observe_velocities, hexapod_closed_loop_env.hpp:20)
    if (obs_dim != 18 && !f.has("seeded")) { std::fprintf(stderr, "--obs needs --seeded (EnvMock, the reference's stub, is 18 / 18)\n"); return 1; }
    ppo_config_default(&cfg, obs_dim, 18, (int)hidden.size(), hidden.data());
    cfg.ent_coef = (float)f.num("ent", 0.0);                                    // live here (the reference bakes it into the graph)
    int rc = 0;
    try {
        if (f.has("graph")) {                                                    // -g: shape, constants and initial weights from a reference graph file
            const graphspec::GraphSpec g = graphspec::load_graph_spec(f.str("graph", ""));
            cfg = g.config;
            h = graphspec::create_from_graph(g);
        } else {
            if (ppo_create(&cfg, &h) != 0) throw std::runtime_error(ppo_last_error(nullptr));
            if (ppo_init_orthogonal(h, (uint64_t)f.num("seed", 0)) != 0) throw std::runtime_error(ppo_last_error(h));
        }
        std::vector<std::shared_ptr<Env>> envs;
        for (int i = 0; i < n_envs; ++i) {
            if (f.has("seeded")) envs.push_back(std::make_shared<SeededEnvMock>(1234u, (uint32_t)i, obs_dim, 18));
            else envs.push_back(std::make_shared<EnvMock>(i + 1));
        }
        std::unique_ptr<Env> inner;
        if (n_envs > 1) inner.reset(new VecEnv(envs));                           // ppo2.cpp:188-201
        else inner.reset(f.has("seeded") ? static_cast<Env*>(new SeededEnvMock(1234u, 0, obs_dim, 18)) : static_cast<Env*>(new EnvMock(1)));
        EnvNormalize env{std::move(inner), h, training};                          // ppo2.cpp:207
        PPO2 algorithm{h, env, 0.99f, n_steps, cfg.ent_coef, (float)f.num("lr", 1e-3), 0.5f, 0.5f, 0.95f, (int)f.num("minibatches", 32),
                       (int)f.num("epochs", 10), (float)f.num("cr", 0.2)};         // ppo2.cpp:215-217
        algorithm.seed = (unsigned long long)f.num("seed", 0);
        if (f.has("path")) algorithm.load(f.str("path", ""));
        if (training) {
            const int steps = (int)f.num("steps", 2e7);
            const int saves = f.has("saves") ? (int)f.num("saves", 0) : 0;
            const std::string prefix = f.str("dir", ".") + "/" + f.str("id", "ppo_cpp_hip") + ".pkl";
            algorithm.learn(steps, saves, saves > 0 ? prefix : "");
        } else {                                                                   // playback without a renderer: print deterministic actions
            Mat obs = env.reset();
            for (int t = 0; t < 5; ++t) {
                const Mat a = algorithm.eval(obs);
                std::printf("t=%d action[0..3] = %g %g %g %g\n", t, a(0, 0), a(0, 1), a(0, 2), a(0, 3));
                obs = env.step(a)[0];
            }
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); rc = 3; }
    if (h) ppo_destroy(h);
    return rc;
}
