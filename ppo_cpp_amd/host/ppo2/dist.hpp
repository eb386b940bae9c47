// dist.hpp -- data-parallel bootstrap of the C++ host layer: one process per GPU, no Python, no torch.
//
// The reference has no counterpart (its only parallelism is a thread per environment, env/vec_env.hpp:51-53); this is the
// host side of SURVEY 8e.  `ppo_cpp_hip --ranks N` is a LAUNCHER: it makes no HIP call at all, starts N fresh child processes
// of the same binary (posix_spawn: `--rank r --world N --ctl_fd F`), serves their control plane, lets rank 0's standard output
// through (the other ranks' goes to /dev/null, standard error is shared) and returns the worst exit code.  No process that has
// touched the GPU ever execs or re-launches itself.
//
// Control plane: one AF_UNIX socket pair per rank between the launcher and the child.  The only operation is an ALL-GATHER in
// lock step: every rank sends [u32 length][payload], the launcher answers every rank with the N payloads in rank order.  That
// carries the 128-byte ncclUniqueId (made by rank 0), the 64-byte IPC handles of the one-shot peer exchange and barriers
// (length 0).  Nothing of the data path goes through it: gradients and statistics travel over RCCL / peer-mapped HBM inside
// libppo_hip (ppo_dist_init, ppo_dist_peer_attach).
#pragma once
#include <cerrno>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <fcntl.h>
#include <poll.h>
#include <spawn.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../../include/ppo_hip.h"

extern char** environ;

namespace ppodist {

inline bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}
inline bool read_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t k = ::recv(fd, c, n, 0);
        if (k == 0) return false;                                   // peer closed
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}

// What a rank knows about the job.  world == 1: single process, every call below is a no-op.
struct Context {
    int world = 1, rank = 0;
    int ctl_fd = -1;                  // this rank's end of its socket pair (inherited from the launcher)

    // every rank contributes n bytes; returns the world * n bytes of all ranks in rank order
    std::vector<char> allgather(const void* mine, uint32_t n) const {
        std::vector<char> all((size_t)world * n);
        if (world == 1) { if (n) std::memcpy(all.data(), mine, n); return all; }
        if (!write_all(ctl_fd, &n, sizeof n) || (n && !write_all(ctl_fd, mine, n))) throw std::runtime_error("control plane: the launcher is gone (send)");
        if (!all.empty() && !read_all(ctl_fd, all.data(), all.size())) throw std::runtime_error("control plane: the launcher is gone (a rank left the job?)");
        if (all.empty()) { char ack; if (!read_all(ctl_fd, &ack, 1)) throw std::runtime_error("control plane: the launcher is gone (barrier)"); }
        return all;
    }
    void barrier() const { (void)allgather(nullptr, 0); }

    // ppo_dist_init with the unique id made by rank 0, then (peer == true, world <= 8) the one-shot peer exchange: export, all-gather of
    // the IPC handles, collective attach.  Returns true when the peer path is in use afterwards.
    bool init_handle(ppo_handle* h, bool peer) const {
        if (world == 1) return false;
        char uid[128] = {0};
        if (rank == 0 && ppo_dist_unique_id(uid) != 0) throw std::runtime_error(std::string("ppo_dist_unique_id: ") + ppo_last_error(nullptr));
        const std::vector<char> uids = allgather(uid, 128);
        if (ppo_dist_init(h, world, rank, uids.data()) != 0) throw std::runtime_error(std::string("ppo_dist_init: ") + ppo_last_error(h));
        if (!peer || world > 8) return false;
        char handle[64];
        if (ppo_dist_peer_export(h, handle) != 0) throw std::runtime_error(std::string("ppo_dist_peer_export: ") + ppo_last_error(h));
        const std::vector<char> handles = allgather(handle, 64);
        if (ppo_dist_peer_attach(h, handles.data()) != 0) throw std::runtime_error(std::string("ppo_dist_peer_attach: ") + ppo_last_error(h));
        return ppo_dist_peer_active(h) != 0;
    }
};

// ---- launcher side ------------------------------------------------------------------------------------------------------------
// Starts `world` children of `exe` with `args` + {"--rank", r, "--world", N, "--ctl_fd", F}, serves the all-gather until every child has
// closed its socket, reaps them and returns the worst exit status (a child killed by signal s counts as 128 + s).  When a rank dies or
// leaves the protocol the others cannot finish their collectives: they get `grace_ms` to notice (their control-plane reads fail at once,
// a data-path wait inside libppo_hip is bounded by PPO_HIP_PEER_TIMEOUT_MS) and are then terminated by pid.
inline int launch_ranks(const std::string& exe, const std::vector<std::string>& args, int world, int grace_ms = 15000) {
    if (world < 1 || world > 64) { std::fprintf(stderr, "--ranks %d: expected 1..64\n", world); return 2; }
    std::vector<pid_t> pid((size_t)world, -1);
    std::vector<int> fd((size_t)world, -1), child_fd((size_t)world, -1), status((size_t)world, -1);
    for (int r = 0; r < world; ++r) {
        int sv[2];
        if (::socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { std::perror("socketpair"); return 2; }
        ::fcntl(sv[0], F_SETFD, FD_CLOEXEC);                       // the launcher's ends never reach a child
        fd[r] = sv[0]; child_fd[r] = sv[1];
    }
    int worst = 0;
    for (int r = 0; r < world; ++r) {
        std::vector<std::string> a;
        a.push_back(exe);
        a.insert(a.end(), args.begin(), args.end());
        a.push_back("--rank"); a.push_back(std::to_string(r));
        a.push_back("--world"); a.push_back(std::to_string(world));
        a.push_back("--ctl_fd"); a.push_back(std::to_string(child_fd[r]));
        std::vector<char*> argv;
        for (auto& s : a) argv.push_back(const_cast<char*>(s.c_str()));
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        for (int q = 0; q < world; ++q) if (q != r) posix_spawn_file_actions_addclose(&fa, child_fd[q]);
        if (r != 0) posix_spawn_file_actions_addopen(&fa, STDOUT_FILENO, "/dev/null", O_WRONLY, 0);     // rank 0 speaks for the job
        const int rc = ::posix_spawn(&pid[r], exe.c_str(), &fa, nullptr, argv.data(), environ);
        posix_spawn_file_actions_destroy(&fa);
        if (rc != 0) { std::fprintf(stderr, "posix_spawn(%s): %s\n", exe.c_str(), std::strerror(rc)); pid[r] = -1; worst = 2; break; }
    }
    for (int r = 0; r < world; ++r) { ::close(child_fd[r]); child_fd[r] = -1; }
    auto drop = [&](int r) { if (fd[r] >= 0) { ::close(fd[r]); fd[r] = -1; } };
    auto reap = [&](bool block) {
        for (int r = 0; r < world; ++r) {
            if (pid[r] <= 0 || status[r] >= 0) continue;
            int st = 0;
            const pid_t w = ::waitpid(pid[r], &st, block ? 0 : WNOHANG);
            if (w == pid[r]) status[r] = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        }
    };
    bool broken = worst != 0;
    // Event loop: a round completes when every rank's next message has arrived in full.  poll() over all sockets (never a blocking read on
    // one rank: a rank that dies while the others are busy on the GPU must be noticed), children reaped as they leave.
    std::vector<std::vector<char>> in((size_t)world);                  // bytes received and not yet consumed, per rank
    auto have_msg = [&](int r, uint32_t* n) {
        if (in[r].size() < sizeof(uint32_t)) return false;
        std::memcpy(n, in[r].data(), sizeof *n);
        return in[r].size() >= sizeof(uint32_t) + (size_t)*n;
    };
    while (!broken) {
        int open = 0;
        for (int r = 0; r < world; ++r) open += fd[r] >= 0;
        if (open == 0) break;
        std::vector<pollfd> pf;
        std::vector<int> who;
        for (int r = 0; r < world; ++r) if (fd[r] >= 0) { pf.push_back(pollfd{fd[r], POLLIN, 0}); who.push_back(r); }
        const int pr = ::poll(pf.data(), (nfds_t)pf.size(), 100);
        if (pr < 0 && errno != EINTR) { broken = true; break; }
        for (size_t i = 0; pr > 0 && i < pf.size(); ++i) {
            if (!(pf[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            const int r = who[i];
            char buf[4096];
            const ssize_t k = ::recv(fd[r], buf, sizeof buf, MSG_DONTWAIT);
            if (k > 0) in[r].insert(in[r].end(), buf, buf + k);
            else if (k == 0 || (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) drop(r);
        }
        reap(false);
        for (int r = 0; r < world; ++r) if (status[r] > 0) broken = true;                              // a rank failed: the job cannot finish
        // a complete round?
        int ready = 0, closed = 0;
        uint32_t len0 = 0;
        bool mismatch = false;
        for (int r = 0; r < world; ++r) {
            uint32_t n = 0;
            if (have_msg(r, &n)) { if (ready && n != len0) mismatch = true; len0 = n; ++ready; if (n > (1u << 20)) mismatch = true; }
            else if (fd[r] < 0) ++closed;
        }
        if (mismatch || (ready && closed)) { broken = true; break; }                                    // sizes differ, or a rank left while others wait for it
        if (ready == world) {
            std::vector<char> all;
            for (int r = 0; r < world; ++r) {
                all.insert(all.end(), in[r].begin() + sizeof(uint32_t), in[r].begin() + sizeof(uint32_t) + len0);
                in[r].erase(in[r].begin(), in[r].begin() + sizeof(uint32_t) + len0);
            }
            if (all.empty()) all.push_back(1);                                                          // barrier acknowledgement
            for (int r = 0; r < world; ++r) if (fd[r] < 0 || !write_all(fd[r], all.data(), all.size())) { drop(r); broken = true; }
        }
    }
    for (int r = 0; r < world; ++r) drop(r);                           // (broken: the survivors' next control-plane call fails at once)
    // reap; after a failure the survivors get a grace period, then SIGTERM / SIGKILL by pid
    for (int waited = 0; ; waited += 20) {
        reap(false);
        bool all_done = true, any_bad = broken;
        for (int r = 0; r < world; ++r) { if (pid[r] > 0 && status[r] < 0) all_done = false; if (status[r] > 0) any_bad = true; }
        if (all_done) break;
        if (any_bad && waited >= grace_ms) {
            for (int r = 0; r < world; ++r) if (pid[r] > 0 && status[r] < 0) ::kill(pid[r], waited >= grace_ms + 2000 ? SIGKILL : SIGTERM);
        }
        ::usleep(20000);
    }
    for (int r = 0; r < world; ++r) if (status[r] > worst) worst = status[r];
    if (broken && worst == 0) worst = 4;
    return worst;
}

}  // namespace ppodist
