// ppo_rollout1.hpp -- persistent rollout of ONE environment (the reference's shipped command line: one env, ppo2.cpp:114,215-217;
// BASELINE configs[1]) on the reference's network shape (18 obs / 18 act, [64,64]): the whole rollout in ONE launch of ONE WAVE.
//
// narrow_rollout_kernel (ppo_narrow.hpp) runs an env step of its 32-row group through LDS tiles, matrix instructions and nine
// workgroup barriers: 10.5 k cycles = 4.4 us per env step, for ONE live row.  Here the policy tower's forward weights live in
// registers (lane n holds column n of W0, W1 and W_mu: 160 registers), an activation vector is one register per lane, and a layer
// is a k-ordered chain of `v_readlane` + `v_fma` -- no LDS, no barrier, nothing but the rollout-row stores leaves the wave.
// The arithmetic is narrow_collect_kernel's statement for statement, so every rollout field, the running statistics and the state
// handed to the next rollout are bit-identical to the per-step launches (test_persistent_rollout_is_bitwise_the_per_step_launches):
//   * a dense layer reproduces nw_dense<CK>: four accumulator chains, chain q taking the k-steps s = q, q+4, ... of four k values
//     each in k order (an exact-fp32 16x16x4 matrix instruction IS that fmaf chain), combined as (c0 + c1) + (c2 + c3);
//   * the sums over actions reproduce the 16-lanes-per-row loop (elements j and j + 16 added in that order, then group16_sum's tree);
//   * EnvNormalize::step / RunningStatistics::update for a batch of one row (env_normalize.hpp:64-116, running_statistics.hpp:26-104).
// Two more waves run AHEAD of the main wave, on their own SIMDs: the noise wave draws the counter-RNG noise of step t + 1 (two 64-bit hashes,
// a log, a cos and a square root per draw), the env wave owns the whole environment side -- the transition that follows step t, the
// running-statistics recurrences (four dependent divisions and a square root per step), the reward branch and the normalised
// observation of step t + 1, which it also writes to the rollout -- while the main wave runs step t.  All of that depends on the step
// index and on earlier env-side state only, never on the policy's output (the synthetic env ignores the actions, env_mock.hpp:47-60).
// One s_barrier per env step orders the hand-over through a double-buffered LDS block.
// The value tower is not needed inside the loop: the host runs it afterwards, batched over the T rows (as before).
#pragma once
#include "ppo_narrow.hpp"

#define R1_CHAIN(ACC, XV, WREG, K)                                                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < (K); ++k_) {                                               \
        const float xk_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, XV), k_)); \
        ACC[(k_ >> 2) & 3] = fmaf(xk_, WREG[k_], ACC[(k_ >> 2) & 3]);                                   \
    }

// HOST (round 5; the reference's real setting: ONE environment stepped on the host, ppo2.cpp:215-217): the same three waves stay resident across the env
// steps of a host-Env rollout and speak narrow_rollout_kernel's protocol (NwRolloutArgs: host_act / ctl[PCTL_D2H], host_in / ctl[PCTL_H2D], bounded waits,
// ctl[PCTL_EXIT] = 1 + booked transitions, relaunch at t0 with a pending transition).  The env wave cannot run ahead any more -- the transition depends on the
// action -- so a step is: main wave forward + sample (~1 us) -> action into pinned memory -> host Env::step -> env wave reads the transition, books it (statistics,
// reward) and hands the next normalised observation over.  narrow_rollout_kernel spent 10.5 k cycles of LDS tiles and barriers on that one live row.
// KP0 = observation tile (32 or 64 columns); the dense widths O <= KP0 and A <= 32 are uniform run-time values
template <int KP0, bool HOST = false>
__global__ __launch_bounds__(192) void narrow_rollout1_kernel(NetDev net, NwLayout lay, NwRolloutArgs q) {
    __shared__ float s_hand[2][96];                          // [parity of t]: 0..A-1 noise of step t, 32..32+O-1 the normalised observation of step t
    __shared__ int s_stop;                                   // HOST: the env wave gave up waiting for the host (or was asked to stop): everybody leaves at the next barrier
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(NwRolloutArgs)>();
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int O = __builtin_amdgcn_readfirstlane(net.O), A = __builtin_amdgcn_readfirstlane(net.A);
    const bool lj = lane < O;                                 // observation element of this lane
    const bool la = lane < A;                                 // action element of this lane
    if (role == 1) {
        // ---- noise wave: one step ahead of the main wave ----------------------------------------------------------------------------------
        auto produce = [&](int t) __attribute__((always_inline)) {
            if (lane < net.A && !q.noise) s_hand[t & 1][lane] = ctr_normal(q.seed, (uint32_t)q.env0, q.step0 + (uint32_t)t, lane);
        };
        produce(q.t0);
        for (int t = q.t0; t < q.T; ++t) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // what was produced for step t is visible to the main wave
            if (HOST && *(volatile int*)&s_stop) break;
            if (t + 1 < q.T) produce(t + 1);
        }
        return;
    }
    if (role == 2) {
        // ---- env wave: transition, EnvNormalize::step / RunningStatistics::update for a batch of ONE row, rollout rows obs / rewards / dones ----
        float raw = lj ? q.st.raw_obs[lane] : 0.f;
        float mean = lj ? q.st.obs_mean[lane] : 0.f, var = lj ? q.st.obs_var[lane] : 1.f;
        float istd = 1.0f / sqrtf(var + q.eps);
        float done = q.st.done[0], ret = q.st.ret[0];
        float ret_mean = *q.st.ret_mean, ret_var = *q.st.ret_var;
        double obs_cnt = *q.st.obs_count, ret_cnt = *q.st.ret_count;
        auto merge = [&](float mean0, float var0, double cnt, float bmean, float bM2, float nbf, float& mean1, float& var1) __attribute__((always_inline)) {
            const double nb = (double)nbf, tot = cnt + nb;
            const float bvar = bM2 / (float)nb;                                        // running_statistics.hpp:51-54
            const float delta = bmean - mean0;                                         // :90
            mean1 = mean0 + (delta * (float)nb) / (float)tot;                          // :94
            const float m_a = var0 * (float)cnt, m_b = bvar * (float)nb;               // :97-98
            const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;   // :100
            var1 = M2 / (float)tot;                                                    // :101
        };
        // rollout row t: the flag that arrived with obs_t, the observation normalised with the statistics that already include it
        auto publish = [&](int t) __attribute__((always_inline)) {
            if (lane == 0) q.ro_done[t] = done;
            if (lj) {
                float x = raw;
                if (q.norm_obs) { x = (x - mean) * istd; x = tf_min(tf_max(x, -q.clip_obs), q.clip_obs); }   // env_normalize.hpp:99-104
                q.ro_obs[(size_t)t * O + lane] = x;
                s_hand[t & 1][32 + lane] = x;
            }
        };
        // HOST: the transition the host posted into the pinned block [O obs | reward | done] (system-scope loads: never served by a cache)
        auto read_host = [&](float& rew_out) __attribute__((always_inline)) {
            raw = lj ? __hip_atomic_load(q.host_in + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
            rew_out = __hip_atomic_load(q.host_in + O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            done = __hip_atomic_load(q.host_in + O + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        };
        int booked = q.t0;                                   // HOST: transitions whose bookkeeping is done (the exit report)
        if constexpr (!HOST) {
        publish(q.t0);
        for (int t = q.t0; t < q.T; ++t) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // block t is complete; the main wave is done with block t + 1's buffer
            // ---- env transition that follows step t (counter hash): new observation, reward, done ------------------------------------------------
            // (hash lanes 0..O-1 = the observation, O = the reward, O+1 = the done flag: with O up to 64 the last two need lanes of their
            // own, so lanes 0 and 1 draw them in a second call)
            float tv = 0.f, tw = 0.f;
            if (lj) tv = u32_to_sym_unit(ctr_hash(q.seed, (uint32_t)q.env0, q.step0 + (uint32_t)t + 1u, (uint32_t)lane));
            if (lane < 2) {
                const uint32_t hs = ctr_hash(q.seed, (uint32_t)q.env0, q.step0 + (uint32_t)t + 1u, (uint32_t)(O + lane));
                tw = lane == 0 ? u32_to_sym_unit(hs) : ((hs % 300u == 0u) ? 1.0f : 0.0f);
            }
            raw = tv;
            const float rew = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw), 0));
            done = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw), 1));
            // ---- EnvNormalize::step bookkeeping for a batch of ONE row ----------------------------------------------------------------------------
            if (q.norm_obs) {
                float sum = 0.f; sum += raw;
                const float bmean = sum / 1.0f;                                        // colwise().mean()
                float m2 = 0.f; { const float d = raw - bmean; m2 += d * d; }
                float m1, v1;
                merge(mean, var, obs_cnt, bmean, m2, 1.0f, m1, v1);
                mean = m1; var = v1; istd = 1.0f / sqrtf(v1 + q.eps);
                obs_cnt = (double)1.0f + obs_cnt;                                      // :103
            }
            {
                ret = ret * q.gamma + rew;                                             // env_normalize.hpp:66
                float sum = 0.f; sum += ret;
                float m1 = ret_mean, v1 = ret_var;
                if (q.norm_rew) {                                                      // :75-77 (training)
                    const float bmean = sum / 1.0f;
                    float m2 = 0.f; { const float d = ret - bmean; m2 += d * d; }
                    merge(ret_mean, ret_var, ret_cnt, bmean, m2, 1.0f, m1, v1);
                    ret_cnt = (double)1.0f + ret_cnt;
                }
                ret_mean = m1; ret_var = v1;
                const float inv = 1.0f / sqrtf(v1 + q.eps);                            // :79
                float y = rew;
                if (q.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -q.clip_rew), q.clip_rew); }
                if (lane == 0) q.ro_rew[t] = y;
                ret = ret * (1.0f - done);                                             // :88-91
            }
            if (t + 1 < q.T) publish(t + 1);
        }
        } else {
        // ---- HOST: the same statements, the transition from the host instead of the counter hash; t = t0 - 1 books a transition posted before this launch.
        // The main wave waits for the next NORMALISED OBSERVATION only: the observation statistics are merged and the row handed over first, the reward side of the
        // transition (return statistics: three dependent divisions and a square root) is booked behind the barrier, while the main wave runs the policy ----
        bool stopped = false, owe = false;
        float owe_rew = 0.f, owe_done = 0.f; int owe_t = 0;
        auto book_reward = [&]() __attribute__((always_inline)) {
            if (!owe) return;
            owe = false;
            const float rew = owe_rew;
            ret = ret * q.gamma + rew;                                             // env_normalize.hpp:66
            float sum = 0.f; sum += ret;
            float m1 = ret_mean, v1 = ret_var;
            if (q.norm_rew) {                                                      // :75-77 (training)
                const float bmean = sum / 1.0f;
                float m2 = 0.f; { const float d = ret - bmean; m2 += d * d; }
                merge(ret_mean, ret_var, ret_cnt, bmean, m2, 1.0f, m1, v1);
                ret_cnt = (double)1.0f + ret_cnt;
            }
            ret_mean = m1; ret_var = v1;
            const float inv = 1.0f / sqrtf(v1 + q.eps);                            // :79
            float y = rew;
            if (q.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -q.clip_rew), q.clip_rew); }
            if (lane == 0) q.ro_rew[owe_t] = y;
            ret = ret * (1.0f - owe_done);                                         // :88-91
        };
        if (lane == 0) s_stop = 0;
        for (int t = q.pending ? q.t0 - 1 : q.t0; t < q.T; ++t) {
            float rew;
            if (t < q.t0) read_host(rew);
            else {
                if (t == q.t0 && !q.pending) publish(q.t0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // block t is complete; the main wave is done with block t + 1's buffer
                book_reward();                               // (of transition t - 1: the main wave is busy with step t)
                if (stopped) break;
                int ok = 1;                                  // the host's transition after action t: bounded wait on its sequence word
                if (lane == 0) {
                    unsigned n = 0;
                    const unsigned* hw = q.h2d ? q.h2d : q.ctl + PCTL_H2D;
                    while (peer_ld_sys(hw) < (unsigned)(t + 1)) {
                        if (++n > q.poll_cap || peer_ld_sys(q.ctl + PCTL_STOP)) { ok = 0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                if (!ok) { if (lane == 0) s_stop = 1; stopped = true; continue; }       // (one more barrier: the other waves see the flag there)
                read_host(rew);
            }
            // ---- EnvNormalize::step bookkeeping for a batch of ONE row: the observation side now ... ------------------------------------------------
            if (q.norm_obs) {
                float sum = 0.f; sum += raw;
                const float bmean = sum / 1.0f;                                        // colwise().mean()
                float m2 = 0.f; { const float d = raw - bmean; m2 += d * d; }
                float m1, v1;
                merge(mean, var, obs_cnt, bmean, m2, 1.0f, m1, v1);
                mean = m1; var = v1; istd = 1.0f / sqrtf(v1 + q.eps);
                obs_cnt = (double)1.0f + obs_cnt;                                      // :103
            }
            owe = true; owe_rew = rew; owe_done = done; owe_t = t;                     // ... the reward side behind the next barrier (or at the exit)
            booked = t + 1;
            if (t + 1 < q.T) publish(t + 1);
        }
        book_reward();
        }
        // ---- exit: the state goes home ---------------------------------------------------------------------------------------------------------
        if (lj) { q.st.raw_obs[lane] = raw; q.st.obs_mean[lane] = mean; q.st.obs_var[lane] = var; }
        if (lane == 0) {
            q.st.done[0] = done; q.st.ret[0] = ret;
            *q.st.obs_count = obs_cnt; *q.st.ret_mean = ret_mean; *q.st.ret_var = ret_var; *q.st.ret_count = ret_cnt;
        }
        if (HOST && lane == 0) peer_st_sys(q.ctl + PCTL_EXIT, 1u + (unsigned)booked);      // (the state is read by the NEXT launch: the kernel boundary orders it)
        return;
    }
    // ---- main wave: this lane's weight columns and biases ------------------------------------------------------------------------------------
    float w0[KP0], w1[64], wm[64];
    {
        const float* img = q.img;
#pragma unroll
        for (int k = 0; k < KP0; ++k) w0[k] = img[lay.wf[0] + k * lay.wf_ld[0] + lane];
#pragma unroll
        for (int k = 0; k < 64; ++k) w1[k] = img[lay.wf[1] + k * lay.wf_ld[1] + lane];
#pragma unroll
        for (int k = 0; k < 64; ++k) wm[k] = lane < 32 ? img[lay.wh + k * lay.wh_ld + lane] : 0.f;
    }
    const float* par = q.img + lay.par;
    const float b0 = par[net.par_b[0] + lane], b1 = par[net.par_b[1] + lane];
    const float bmu = lane < 32 ? par[net.par_bmu + lane] : 0.f, lsj = lane < 32 ? par[net.par_ls + lane] : 0.f;
    for (int t = q.t0; t < q.T; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // the other waves' block of step t is complete (and they may overwrite the other one)
        if (HOST && *(volatile int*)&s_stop) break;
        const float* hb = s_hand[t & 1];
        float eps = 0.f, x = 0.f;
        if (la) eps = q.noise ? q.noise[(size_t)t * A + lane] : hb[lane];
        if (lj) x = hb[32 + lane];
        // ---- forward: lane n owns output column n --------------------------------------------------------------------------------------
        float h1, h2, mu;
        { float c[4] = {0.f, 0.f, 0.f, 0.f}; R1_CHAIN(c, x, w0, KP0); h1 = fast_tanh(((c[0] + c[1]) + (c[2] + c[3])) + b0); }
        { float c[4] = {0.f, 0.f, 0.f, 0.f}; R1_CHAIN(c, h1, w1, 64); h2 = fast_tanh(((c[0] + c[1]) + (c[2] + c[3])) + b1); }
        { float c[4] = {0.f, 0.f, 0.f, 0.f}; R1_CHAIN(c, h2, wm, 64); mu = ((c[0] + c[1]) + (c[2] + c[3])) + bmu; }
        // ---- sample + neglogp (G:5894-6672) -----------------------------------------------------------------------------------------------
        const float logstd = mu * 0.0f + lsj;
        const float sigma = expf(logstd);
        const float act = mu + sigma * eps;
        const float z = (act - mu) / sigma;
        if constexpr (HOST) {
            // the action goes out FIRST (system-scope write-through stores into pinned host memory, drained, then the sequence word): the host steps its Env
            // while this wave finishes the row
            if (la) __hip_atomic_store(q.host_act + lane, act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) peer_st_sys(q.ctl + PCTL_D2H, (unsigned)(t + 1));
        }
        if (la) q.ro_act[(size_t)t * A + lane] = act;
        // the per-step kernels add elements j and j + 16 on lane j of a 16-lane group, then group16_sum: same order here
        float zz = la ? z * z : 0.f, sl = la ? logstd : 0.f;
        const float zz_hi = __shfl_down(zz, 16), sl_hi = __shfl_down(sl, 16);
        float ssq = 0.f + zz, slog = 0.f + sl;
        if (lane + 16 < A) { ssq += zz_hi; slog += sl_hi; }
        ssq = group16_sum(ssq); slog = group16_sum(slog);
        if (lane == 0) q.ro_nlp[t] = 0.5f * ssq + HALF_LOG_2PI * (float)A + slog;
    }
}
