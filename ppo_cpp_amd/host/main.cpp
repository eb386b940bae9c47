// main.cpp -- command-line driver with the training flags of the reference's ppo2.cpp (ppo2.cpp:93-128) for the mock
// environments: builds the same stack (N x Env -> VecEnv -> EnvNormalize -> PPO2, ppo2.cpp:188-217), trains with periodic
// checkpoints in the reference's format, or (--path) loads a checkpoint and plays the deterministic policy.
// Physics (DART hexapod, --closed_loop / --bullet / --duration / --framerate) is out of scope: those flags are rejected.
//
// Data parallel (the reference has none): `--ranks N` turns this process into a LAUNCHER that makes no GPU call -- it starts N
// children of the same binary (`--rank r --world N --ctl_fd F`, ppo2/dist.hpp), relays rank 0's output and returns the worst exit code.
// Every rank builds threads / N environments with the global ids rank * E/N + i, joins the communicator (ppo_dist_init; rank 0's
// ncclUniqueId travels over the launcher's socket pairs) and runs the unchanged PPO2::learn; rank 0 prints the CSV line with the job's fps.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

#include "env/env_mock.hpp"
#include "env/env_normalize.hpp"
#include "env/vec_env.hpp"
#include "ppo2/dist.hpp"
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
                             {"--minibatches", "minibatches"}, {"--seed", "seed"}, {"-g", "graph"}, {"--graph", "graph"}, {"--graph_path", "graph"}, {"--obs", "obs"},
                             {"--ranks", "ranks"}, {"--rank", "rank"}, {"--world", "world"}, {"--ctl_fd", "ctl_fd"}, {"--devices", "devices"}, {"--collective", "collective"},
                             {"--explicit_dir", "explicit_dir"}, {"--dump_dir", "dump_dir"}};
const char* kSwitches[][2] = {{"-r", "resume"}, {"--resume", "resume"}, {"-v", "verbose"}, {"--verbose", "verbose"}, {"--seeded", "seeded"},
                              {"--replica_saves", "replica_saves"}, {"--ctl_selftest", "ctl_selftest"}};

// raw little-endian arrays: the parity hooks' on-disk format (--explicit_dir / --dump_dir; tests/test_host_dp.py writes and reads them with numpy)
template <typename T>
std::vector<T> read_raw(const std::string& path, bool required) {
    std::vector<T> v;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { if (required) throw std::runtime_error("cannot open " + path); return v; }
    std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (n > 0 && std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) { std::fclose(f); throw std::runtime_error("short read: " + path); }
    std::fclose(f);
    return v;
}
template <typename T>
void write_raw(const std::string& path, const T* p, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, sizeof(T), n, f) != n) { if (f) std::fclose(f); throw std::runtime_error("cannot write " + path); }
    std::fclose(f);
}
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
    // ---- launcher: no GPU call in this process (dist.hpp) --------------------------------------------------------------------
    if (f.has("ranks") && !f.has("rank")) {
        const int world = (int)f.num("ranks", 1);
        std::vector<std::string> args;
        for (int i = 1; i < argc; ++i) {
            if (std::string(argv[i]) == "--ranks") { ++i; continue; }
            args.push_back(argv[i]);
        }
        char self[4096];
        const ssize_t k = ::readlink("/proc/self/exe", self, sizeof self - 1);
        const std::string exe = k > 0 ? std::string(self, (size_t)k) : std::string(argv[0]);
        return ppodist::launch_ranks(exe, args, world);
    }
    ppodist::Context ctx;
    if (f.has("rank")) {
        ctx.world = (int)f.num("world", 1); ctx.rank = (int)f.num("rank", 0); ctx.ctl_fd = (int)f.num("ctl_fd", -1);
        if (ctx.world < 1 || ctx.rank < 0 || ctx.rank >= ctx.world || (ctx.world > 1 && ctx.ctl_fd < 0)) { std::fprintf(stderr, "bad --rank / --world / --ctl_fd\n"); return 1; }
    }
    if (f.has("ctl_selftest")) {
        // control plane alone, no GPU (tests/test_host_dp.py on CPU): a 128-byte all-gather where only rank 0 contributes (the unique id's
        // path), a 64-byte all-gather (the IPC handles'), a barrier; rank 0 prints what it saw
        try {
            char uid[128] = {0};
            if (ctx.rank == 0) std::snprintf(uid, sizeof uid, "uid-from-rank-0");
            const std::vector<char> u = ctx.allgather(uid, 128);
            char hd[64] = {0};
            std::snprintf(hd, sizeof hd, "handle-%d", ctx.rank);
            const std::vector<char> hs = ctx.allgather(hd, 64);
            ctx.barrier();
            bool ok = std::strcmp(u.data(), "uid-from-rank-0") == 0;
            for (int r = 0; r < ctx.world; ++r) { char want[64]; std::snprintf(want, sizeof want, "handle-%d", r); ok = ok && std::strcmp(hs.data() + 64 * (size_t)r, want) == 0; }
            std::printf("ctl_selftest rank %d of %d: %s\n", ctx.rank, ctx.world, ok ? "ok" : "MISMATCH");
            if (const char* die = std::getenv("PPO_CTL_SELFTEST_DIE")) if (atoi(die) == ctx.rank) return 7;       // failure drill: this rank leaves with an error
            if (std::getenv("PPO_CTL_SELFTEST_DIE")) ctx.barrier();                                                  // ... while the others wait for it
            return ok ? 0 : 5;
        } catch (const std::exception& e) { std::fprintf(stderr, "rank %d: %s\n", ctx.rank, e.what()); return 6; }
    }
    std::vector<int32_t> hidden;
    { std::stringstream ss(f.str("hidden", "64,64")); std::string tok; while (std::getline(ss, tok, ',')) hidden.push_back(atoi(tok.c_str())); }
    const int n_envs_job = (int)f.num("threads", 1), n_steps = (int)f.num("batch_steps", 2048);
    if (n_envs_job % ctx.world != 0) { std::fprintf(stderr, "--threads %d does not divide over %d ranks\n", n_envs_job, ctx.world); return 1; }
    const int n_envs = n_envs_job / ctx.world, env0 = ctx.rank * n_envs;          // this rank's environments: global ids env0 .. env0 + n_envs - 1
    const bool training = !f.has("path") || f.has("resume");                  // ppo2.cpp:171
    ppo_handle* h = nullptr;
    ppo_config cfg;
    const int obs_dim = (int)f.num("obs", 18);                                  // (the reference's closed-loop hexapod: 18, or 36 with observe_velocities, hexapod_closed_loop_env.hpp:20)
    if (obs_dim != 18 && !f.has("seeded")) { std::fprintf(stderr, "--obs needs --seeded (EnvMock, the reference's stub, is 18 / 18)\n"); return 1; }
    ppo_config_default(&cfg, obs_dim, 18, (int)hidden.size(), hidden.data());
    cfg.ent_coef = (float)f.num("ent", 0.0);                                    // live here (the reference bakes it into the graph)
    {   // --devices a,b,c: HIP ordinal per rank (default: rank r -> device r; one entry = every rank, e.g. a dry run of N ranks on one GPU)
        std::vector<int> devs;
        std::stringstream ss(f.str("devices", "")); std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) devs.push_back(atoi(tok.c_str()));
        if (!devs.empty()) cfg.device = devs[(size_t)ctx.rank % devs.size()];
        else if (ctx.world > 1) cfg.device = ctx.rank;
    }
    int rc = 0;
    try {
        if (f.has("graph")) {                                                    // -g: shape, constants and initial weights from a reference graph file
            graphspec::GraphSpec g = graphspec::load_graph_spec(f.str("graph", ""));
            g.config.device = cfg.device;
            cfg = g.config;
            h = graphspec::create_from_graph(g);
        } else {
            if (ppo_create(&cfg, &h) != 0) throw std::runtime_error(ppo_last_error(nullptr));
            if (ppo_init_orthogonal(h, (uint64_t)f.num("seed", 0)) != 0) throw std::runtime_error(ppo_last_error(h));     // same seed on every rank: replicated weights
        }
        const std::string xdir = f.str("explicit_dir", ""), ddir = f.str("dump_dir", "");
        if (!xdir.empty()) {
            const std::vector<float> theta = read_raw<float>(xdir + "/theta.f32", false);
            if (!theta.empty() && ppo_set_flat(h, 0, theta.data(), (int64_t)theta.size()) != 0) throw std::runtime_error(ppo_last_error(h));
        }
        const bool peer = ctx.init_handle(h, f.str("collective", "rccl") == "peer");                  // before the normaliser and the rollout buffers exist
        if (f.str("collective", "rccl") == "peer" && ctx.world > 1 && !peer) throw std::runtime_error("--collective peer: the peer exchange's probe failed");
        std::vector<std::shared_ptr<Env>> envs;
        for (int i = 0; i < n_envs; ++i) {
            if (f.has("seeded")) envs.push_back(std::make_shared<SeededEnvMock>(1234u, (uint32_t)(env0 + i), obs_dim, 18));
            else envs.push_back(std::make_shared<EnvMock>(env0 + i + 1));
        }
        std::unique_ptr<Env> inner;
        // (the pool sizes itself by the cores this process may use; N ranks on one node share them)
        if (n_envs > 1) inner.reset(new VecEnv(envs, ctx.world > 1 ? std::max(1, usable_cpus() / ctx.world) : 0));          // ppo2.cpp:188-201
        else inner.reset(f.has("seeded") ? static_cast<Env*>(new SeededEnvMock(1234u, (uint32_t)env0, obs_dim, 18)) : static_cast<Env*>(new EnvMock(env0 + 1)));
        EnvNormalize env{std::move(inner), h, training};                          // ppo2.cpp:207
        PPO2 algorithm{h, env, 0.99f, n_steps, cfg.ent_coef, (float)f.num("lr", 1e-3), 0.5f, 0.5f, 0.95f, (int)f.num("minibatches", 32),
                       (int)f.num("epochs", 10), (float)f.num("cr", 0.2)};         // ppo2.cpp:215-217
        algorithm.seed = (unsigned long long)f.num("seed", 0);
        algorithm.set_distributed(ctx.world, ctx.rank);
        algorithm.replica_saves = f.has("replica_saves");
        if (f.has("path")) algorithm.load(f.str("path", ""));
        if (training) {
            const int steps = (int)f.num("steps", 2e7);
            const int saves = f.has("saves") ? (int)f.num("saves", 0) : 0;
            const std::string prefix = f.str("dir", ".") + "/" + f.str("id", "ppo_cpp_hip") + ".pkl";
            // parity hooks: explicit exploration noise [U][T][E_job][A] and epoch permutations [world][U][epochs][B_rank] from raw files
            std::vector<float> noise_all, noise;
            std::vector<int32_t> perms_all;
            const int A = 18, B = n_envs * n_steps, epochs = (int)f.num("epochs", 10);
            const int U = steps / (B * ctx.world);
            if (!xdir.empty()) {
                noise_all = read_raw<float>(xdir + "/noise.f32", false);
                perms_all = read_raw<int32_t>(xdir + "/perms.i32", false);
                if (!noise_all.empty()) {
                    if (noise_all.size() != (size_t)U * n_steps * n_envs_job * A) throw std::runtime_error("noise.f32: expected [updates][steps][envs of the job][actions]");
                    noise.resize((size_t)U * n_steps * n_envs * A);
                    for (size_t ut = 0; ut < (size_t)U * n_steps; ++ut)                  // this rank's environment columns
                        std::memcpy(noise.data() + ut * n_envs * A, noise_all.data() + (ut * n_envs_job + env0) * A, sizeof(float) * (size_t)n_envs * A);
                    algorithm.explicit_noise = noise.data();
                }
                if (!perms_all.empty()) {
                    if (perms_all.size() != (size_t)ctx.world * U * epochs * B) throw std::runtime_error("perms.i32: expected [ranks][updates][epochs][rows of a rank]");
                    algorithm.explicit_perms = perms_all.data() + (size_t)ctx.rank * U * epochs * B;
                }
            }
            algorithm.learn(steps, saves, saves > 0 ? prefix : "");
            if (!ddir.empty()) {
                const std::string base = ddir + "/rank" + std::to_string(ctx.rank);
                const auto& hist = algorithm.history();
                std::vector<float> losses;
                for (const auto& u : hist) losses.insert(losses.end(), u.losses, u.losses + 5);
                write_raw(base + ".losses.f32", losses.data(), losses.size());
                const int P = ppo_num_params(h);
                std::vector<float> v((size_t)P);
                const char* names[3] = {".theta.f32", ".adam_m.f32", ".adam_v.f32"};
                for (int which = 0; which < 3; ++which) {
                    if (ppo_get_flat(h, which, v.data(), P) != 0) throw std::runtime_error(ppo_last_error(h));
                    write_raw(base + names[which], v.data(), v.size());
                }
                std::vector<float> st((size_t)2 * obs_dim + 2);
                double counts[2] = {0, 0};
                if (ppo_norm_get_stats(h, 0, st.data(), st.data() + obs_dim, &counts[0]) != 0 ||
                    ppo_norm_get_stats(h, 1, st.data() + 2 * obs_dim, st.data() + 2 * obs_dim + 1, &counts[1]) != 0) throw std::runtime_error(ppo_last_error(h));
                write_raw(base + ".rms.f32", st.data(), st.size());              // obs mean | obs var | ret mean | ret var
                write_raw(base + ".counts.f64", counts, 2);
                int32_t info[4] = {0, 0, ppo_dist_world(h), ppo_dist_peer_active(h)};
                if (ppo_dist_info(h, &info[0], &info[1], nullptr, nullptr) != 0) throw std::runtime_error(ppo_last_error(h));
                write_raw(base + ".dist.i32", info, 4);                           // communicator's rank count | HIP ordinal | world | peer path in use
            }
        } else {                                                                   // playback without a renderer: print deterministic actions
            Mat obs = env.reset();
            for (int t = 0; t < 5; ++t) {
                const Mat a = algorithm.eval(obs);
                std::printf("t=%d action[0..3] = %g %g %g %g\n", t, a(0, 0), a(0, 1), a(0, 2), a(0, 3));
                obs = env.step(a)[0];
            }
        }
        ctx.barrier();                                                            // every rank is past its last collective before any communicator goes away
    } catch (const std::exception& e) { std::fprintf(stderr, "error%s: %s\n", ctx.world > 1 ? (" (rank " + std::to_string(ctx.rank) + ")").c_str() : "", e.what()); rc = 3; }
    if (h) ppo_destroy(h);
    return rc;
}
