/*
 * ppo_hip.h  --  C-ABI of libppo_hip.so: the MI355X (gfx950) replacement for the TensorFlow graph
 * executor behind ppo_cpp's rollout-collect + minibatch-update hot path.
 *
 * The reference has no plugin/FFI layer for this path: the seam is tensorflow::Session::Run addressed by
 * tensor-name strings (reference ppo2/ppo2.hpp:521-544).  Every entry point below replaces one of those
 * call sites (cited per function, paths relative to the reference root).  Plain pointers and sizes only;
 * no torch / Eigen / TF types.
 *
 * Conventions
 *   - all matrices are row-major fp32 (reference env/env.hpp:14 `Mat`), all vectors contiguous fp32
 *   - every pointer argument is HOST memory unless the name ends in _dev; the caller owns it and it is
 *     only used for the duration of the call (the reference copies on every call too: ppo2/utils.hpp:42,53)
 *   - return value: 0 = ok, negative = error; ppo_last_error() gives the message.  The reference prints
 *     status.ToString() and assert(false)s (ppo2/policies.hpp:39-43, ppo2/ppo2.hpp:452-456)
 *   - one handle = one caller thread = one HIP device + stream (reference: every Session::Run is issued
 *     from the single main thread, env worker threads never touch the session: env/vec_env.hpp:247)
 *   - a non-finite gradient norm poisons the weights with NaN exactly like G:24493-24543; not an error
 *   - there is NO CPU fallback: every call fails loudly when no gfx950 device is usable
 *   - PPO_F32 arithmetic: exact-fp32 matrix instructions (a k-ordered fmaf chain), correctly rounded square root / division in
 *     the clip + Adam step -- in EVERY form of it since round 6, including the reference's own [64,64] shape, whose handle applies Adam inside the next train kernel's
 *     prologue (or, with minibatches of <= 64 rows, inside the resident epoch kernel) during ppo_update.  No stated deviation is left in the default build.
 *     PPO_HIP_ADAM_FAST=1 (read at ppo_create) opts a handle into the hardware's 1-ulp reciprocal and square root for the quotient m * alpha / (sqrt(v) + eps) in ALL its
 *     Adam steps (a 3-ulp error in an update term that is ~1e-3 of the weight; 5 - 9 % of that shape's train step: bench.py reports both);
 *     PPO_HIP_NO_LAZY_ADAM=1 switches the deferred / resident forms themselves off.
 */
#ifndef PPO_HIP_H
#define PPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_MAX_LAYERS 8
#define PPO_ABI_VERSION 3

typedef struct ppo_handle ppo_handle;

/* Everything the reference bakes into the graph file at generation time (SURVEY section 5 'config'):
 * network shape, ent_coef G:11323, vf_coef G:11395, max_grad_norm G:24370, Adam beta1/beta2/eps G:30430-30490. */
typedef struct ppo_config {
    int32_t obs_dim;                  /* O */
    int32_t act_dim;                  /* A */
    int32_t n_hidden;                 /* L, 1..PPO_MAX_LAYERS */
    int32_t hidden[PPO_MAX_LAYERS];   /* h0..h{L-1} */
    float ent_coef;
    float vf_coef;
    float max_grad_norm;
    float adam_beta1;
    float adam_beta2;
    float adam_eps;
    int32_t device;                   /* HIP device ordinal, -1 = current/LOCAL_RANK */
    int32_t max_rows;                 /* largest row count ever passed to step/value/train_step (0 = 65536) */
    int32_t compute_dtype;            /* PPO_F32 (0, default): the reference's arithmetic on the exact-fp32 matrix cores.
                                         PPO_BF16 (1): bf16 operands, fp32 accumulation, fp32 master weights / Adam, layer-by-layer
                                         128x128-tile GEMMs -- BASELINE configs[4]'s mode for wide nets; no reference counterpart */
} ppo_config;
#define PPO_F32 0
#define PPO_BF16 1

/* fills the graph-baked defaults of the reference's shipped graph (ent 0.00071602932, vf 0.5, clip 0.5,
 * Adam 0.9 / 0.999 / 1e-5) for the given shape */
void ppo_config_default(ppo_config* cfg, int32_t obs_dim, int32_t act_dim, int32_t n_hidden, const int32_t* hidden);

/* ---- lifecycle: SessionCreator::load_graph + Session::Run("init") (ppo2/session_creator.hpp:23-66) ---- */
int ppo_create(const ppo_config* cfg, ppo_handle** out);
void ppo_destroy(ppo_handle* h);
const char* ppo_last_error(const ppo_handle* h);   /* h may be NULL: error of the last failed ppo_create */
int ppo_abi_version(void);

/* ---- variables: initializer consts G:2249-5297 / saver G:32312-33437 (ppo2/ppo2.hpp:107-223) ----------
 * Tensors are addressed by index in TF trainable-variable order
 *   pi_fc0/w, pi_fc0/b, vf_fc0/w, vf_fc0/b, pi_fc1/w, ... , vf/w, vf/b, pi/w, pi/b, pi/logstd   (4L+5 tensors)
 * `which`: 0 = weights, 1 = Adam m slot, 2 = Adam v slot. */
int ppo_num_tensors(const ppo_handle* h);
int ppo_tensor_info(const ppo_handle* h, int index, char name[32], int32_t* rows, int32_t* cols); /* cols 0 = 1-D */
int ppo_num_params(const ppo_handle* h);                      /* dense count (146213 for 18/18/[256,256]) */
int ppo_get_tensor(ppo_handle* h, int which, int index, float* dst, int64_t count);
int ppo_set_tensor(ppo_handle* h, int which, int index, const float* src, int64_t count);
/* dense flat vector in the order above (the layout of the oracle's theta / m / v) */
int ppo_get_flat(ppo_handle* h, int which, float* dst, int64_t count);
int ppo_set_flat(ppo_handle* h, int which, const float* src, int64_t count);
int ppo_get_beta_powers(ppo_handle* h, float pw[2]);          /* beta1_power, beta2_power (G:25426,25579) */
int ppo_set_beta_powers(ppo_handle* h, const float pw[2]);
/* orthogonal init of the same family as the constants in G (gain sqrt2 hidden, 0.01 pi, 1.0 vf; biases,
 * logstd, Adam slots zero; beta powers = beta) */
int ppo_init_orthogonal(ppo_handle* h, uint64_t seed);
/* seed of the on-device action-noise generator that stands in for G:5894 RandomStandardNormal when no explicit noise
 * is passed (the reference seeds TF from the clock, ppo2.cpp:159-162; PPO2::seed / --seed end up here).  The draw for
 * a row is keyed by (seed, global row = rank * n_envs + row, call counter, action index), so data-parallel ranks
 * draw independent noise. */
int ppo_seed(ppo_handle* h, uint64_t seed);

/* ---- act model ------------------------------------------------------------------------------------------
 * MlpPolicy::step (ppo2/policies.hpp:33-46): feeds input/Ob:0, fetches output/_action, _value_flat, _neglogp.
 * noise [n,A] = the N(0,1) draw of G:5894 made explicit (parity mode); NULL = on-device counter RNG. */
int ppo_step(ppo_handle* h, const float* obs, int32_t n, const float* noise, float* action, float* value,
             float* neglogp);
/* MlpPolicy::value (ppo2/policies.hpp:64-77) */
int ppo_value(ppo_handle* h, const float* obs, int32_t n, float* value);
/* MlpPolicy::get_deterministic_action (ppo2/policies.hpp:49-62) */
int ppo_act_deterministic(ppo_handle* h, const float* obs, int32_t n, float* action);

/* ---- train op: PPO2::_train_step's Session::Run (ppo2/ppo2.hpp:430-468) ---------------------------------
 * feeds train_model/input/Ob, loss/{action,advs,rewards,old_neglog_pac,old_vpred,learning_rate,clip_range}_ph;
 * target ppo2/_train; losses = {pg_loss, vf_loss, entropy, approxkl, clipfrac}.  `advs` are ALREADY normalised
 * (the reference normalises on the host, ppo2/ppo2.hpp:401-406; ppo_adv_normalize does it on the device). */
int ppo_train_step(ppo_handle* h, float lr, float cliprange, const float* obs, const float* actions,
                   const float* advs, const float* returns, const float* old_neglogp, const float* old_values,
                   int32_t n, float losses[5]);
/* gradient of the last train step (of ppo_train_step, or the last one of ppo_update) BEFORE clipping (debug/parity), dense flat order.
 * PPO_BF16 handles on one GPU do not keep it: it is rebuilt here from the step's partial sums (same bits), which stay valid until the next train step. */
int ppo_get_last_grad(ppo_handle* h, float* dst, int64_t count, float* global_norm);

/* ---- fused host-loop numerics the north star moves onto the device ----------------------------------------
 * PPO2::_train_step prologue (ppo2/ppo2.hpp:401-406) */
int ppo_adv_normalize(ppo_handle* h, const float* returns, const float* values, int32_t n, float* advs);
/* Runner::set_returns (ppo2/runner.hpp:159-191); [T,E] time-major; last_dones = dones after the last step */
int ppo_gae(ppo_handle* h, const float* rewards, const float* values, const float* dones,
            const float* last_values, const float* last_dones, int32_t T, int32_t E, float gamma, float lam,
            float* returns);

/* EnvNormalize (env/env_normalize.hpp:20-116) + RunningStatistics (common/running_statistics.hpp). */
int ppo_norm_init(ppo_handle* h, int32_t n_envs, float gamma, float clip_obs, float clip_rew, float epsilon);
int ppo_norm_obs(ppo_handle* h, const float* raw_obs, int32_t n_envs, int training, float* out);
int ppo_norm_reward(ppo_handle* h, const float* raw_rew, const float* dones, int32_t n_envs, int training,
                    float* out);
/* the norm_obs / norm_reward constructor flags of EnvNormalize (env_normalize.hpp:24-27, honoured at :75 and :95): with
 * norm_obs == 0 observations pass through unscaled and obs_rms is never updated; with norm_reward == 0 rewards pass
 * through and ret_rms is never updated (the discounted return is still accumulated, :66).  Default 1, 1. */
int ppo_norm_set_flags(ppo_handle* h, int norm_obs, int norm_reward);
/* EnvNormalize::reset's `ret = Zero` (env_normalize.hpp:111-116) without touching the statistics */
int ppo_norm_reset_returns(ppo_handle* h);
/* which: 0 = obs_rms, 1 = ret_rms; serialise / deserialise of env_normalize.hpp:134-146 */
int ppo_norm_get_stats(ppo_handle* h, int which, float* mean, float* var, double* count);
int ppo_norm_set_stats(ppo_handle* h, int which, const float* mean, const float* var, double count);

/* ---- device-resident rollout: Runner::run (ppo2/runner.hpp:56-157) with buffers kept in HBM ---------------
 * Buffers are time-major [T,E,...]; the reference's env-major row r = e*T + t (runner.hpp:136-152) is mapped
 * at gather time, nothing is transposed. */
int ppo_rollout_alloc(ppo_handle* h, int32_t n_envs, int32_t n_steps);
/* host-Env path (an Env behind env/env.hpp steps on the host):
 *   reset   : EnvNormalize::reset + Runner ctor (runner.hpp:48-50): normalise raw obs, dones = 0
 *   act     : policy step on the current obs -> rollout[t]; actions copied back for Env::step (runner.hpp:75-116)
 *   observe : EnvNormalize::step on the raw step result; stores reward[t]; becomes the current obs/dones
 * Call pattern per rollout: act(0), observe(0), act(1), ... observe(T-1), finish.  With <= 64 environments on the narrow path
 * (the reference's setting is ONE) and no explicit noise, the library serves the whole rollout from one resident kernel: act /
 * observe then only exchange the actions and the transition through pinned memory.  That is invisible to the caller: any other
 * entry point called in between sees the state the step-by-step path would show (the kernel is retired and relaunched as
 * needed), values[t] are available after ppo_rollout_finish, and a host that pauses only costs the kernel a bounded wait. */
int ppo_rollout_reset(ppo_handle* h, const float* raw_obs);
int ppo_rollout_act(ppo_handle* h, int32_t t, const float* noise, float* actions_out);
int ppo_rollout_observe(ppo_handle* h, int32_t t, const float* raw_obs, const float* raw_rew, const float* dones);
/* bootstrap value + GAE (runner.hpp:134, 159-191) */
int ppo_rollout_finish(ppo_handle* h, float gamma, float lam);
/* device-env path: the whole T-step collect against the on-device seeded synthetic env (obs ~ U(-1,1)^O,
 * reward ~ U(-1,1), done ~ Bernoulli(1/300), keyed by (seed, global env id, step counter)); env ids start at
 * env0 (rank sharding).  first != 0 performs the reset (step counter step0), otherwise continues from the carried
 * obs/dones.  noise [T,E,A] or NULL.  Ends with bootstrap + GAE. */
int ppo_collect_synthetic(ppo_handle* h, uint32_t seed, int32_t env0, uint32_t step0, int first,
                          const float* noise, float gamma, float lam);
/* field: 0 obs[T,E,O] 1 actions[T,E,A] 2 values 3 neglogp 4 dones 5 rewards 6 returns (all [T,E]) */
int ppo_rollout_download(ppo_handle* h, int field, float* dst, int64_t count);
int ppo_rollout_upload(ppo_handle* h, int field, const float* src, int64_t count);

/* ---- the whole minibatch-update phase of PPO2::learn (ppo2/ppo2.hpp:274-335) on the resident rollout ------
 * perms [noptepochs, B] int32: perms[ep][i] = destination row of flattened source row i in epoch ep
 * (out.row(perm[i]) = in.row(i), ppo2.hpp:291-296), or NULL = fresh on-device pseudo-random permutation per
 * epoch keyed by (seed, epoch).  loss_rows [noptepochs*nminibatches, 5] (may be NULL); mean_losses = their
 * column means (ppo2.hpp:335).  One HIP launch sequence per minibatch, replayed from a hipGraph (the reference's own shape -- [64,64], minibatches of <= 64 rows --
 * runs all minibatches of an epoch inside ONE resident launch; same results bit for bit, PPO_HIP_NO_NARROW_EPOCH=1 keeps the launches). */
int ppo_update(ppo_handle* h, float lr, float cliprange, int32_t noptepochs, int32_t nminibatches,
               const int32_t* perms, uint64_t seed, float* loss_rows, float mean_losses[5]);

/* ---- data parallel: one process per GPU, RCCL over xGMI (new work; the reference has no collectives) -------
 * uid = 128-byte ncclUniqueId made by rank 0 (ppo_dist_unique_id) and broadcast by the launcher.  After init,
 * ppo_train_step / ppo_update all-reduce (sum) the flat gradient + loss sums across ranks between the backward
 * and the clip+Adam launches, and ppo_norm_* merge their batch moments across ranks. */
int ppo_dist_unique_id(char uid[128]);
int ppo_dist_init(ppo_handle* h, int32_t world_size, int32_t rank, const char uid[128]);
int ppo_dist_world(const ppo_handle* h);
/* What a reader of a scaling run needs in order to check the ranks (all out-pointers optional): the number of ranks the
 * COMMUNICATOR reports (ncclCommCount; -1 when the library has no such entry point, 0 without a communicator), this handle's HIP
 * device ordinal, its PCI bus id ("0000:05:00.0") and the path of the collective library that was actually loaded. */
int ppo_dist_info(ppo_handle* h, int32_t* comm_nranks, int32_t* device, char pci_bus_id[32], char library[256]);
/* 1 when ppo_update replays the collectives from its hipGraph (the communicator's library passed the capture probe of
 * ppo_dist_init, or PPO_HIP_GRAPH_RCCL=1), 0 when they are issued eagerly between the launches */
int ppo_dist_graph_collectives(const ppo_handle* h);
/* One-shot all-reduce over peer-mapped buffers (xGMI is point-to-point: every rank pushes its vector into its slot of every
 * peer's gather region, raises a flag there, and sums the slots of its own region in rank order -- one hop instead of a
 * ring's 2(W-1), same summation order on every rank).  One node, world_size <= 8.
 *   ppo_dist_peer_export: after ppo_dist_init; allocates this rank's gather region and returns its 64-byte IPC handle.
 *   ppo_dist_peer_attach: COLLECTIVE; handles = world_size x 64 bytes in rank order (all-gathered by the launcher).  Maps the
 *     peers' regions, runs a known-answer exchange and lets the ranks agree on the outcome over the communicator: every
 *     collective of ppo_update / ppo_rollout_* / ppo_norm_* whose payload fits then uses the peer kernels (plain launches,
 *     replayed from the update's hipGraph); if any rank fails to map a peer or fails the probe, all ranks stay on RCCL.
 *     PPO_HIP_PEER_REDUCE=0 keeps RCCL; PPO_HIP_PEER_TIMEOUT_MS bounds the wait on a peer's flag (default 10 s), after which
 *     the next ppo_update / ppo_rollout_finish / ppo_collect_synthetic returns an error instead of hanging the device.
 *     The call ends with a collective over the communicator, so no rank returns before every rank has finished its probe.
 *   ppo_dist_peer_active: 1 when the peer path is in use.   ppo_dist_peer_enable: switch between the two paths after a
 *     successful attach (collectively, at the same point on every rank).
 *   With the peer path on, a [256,256] handle pushes its gradient tiles from inside the weight-gradient kernel and adds the ranks up inside the Adam launch
 *     (no push / sum launches; PPO_HIP_NO_PEER_TILES=1 restores them).  Single-rank kernels whose workgroups wait for EACH OTHER while holding a CU (the bf16
 *     path's chained layers) are switched off when two ranks of the job share a device: ppo_dist_init compares the ranks' PCI ids over the communicator.
 *     The data-parallel forms that wait on the PEERS' flags (the tile push, adam_kernel's meeting) stay on there -- one device with several ranks is how this path
 *     is tested -- and every such wait is bounded (PPO_HIP_PEER_TIMEOUT_MS): a rank whose peers cannot become resident beside it reports an error, it does not hang. */
int ppo_dist_peer_export(ppo_handle* h, char handle[64]);
int ppo_dist_peer_attach(ppo_handle* h, const char* handles);
int ppo_dist_peer_active(const ppo_handle* h);
int ppo_dist_peer_enable(ppo_handle* h, int on);
/* Sampling under data parallelism.  Default (0): every rank shuffles its OWN B rows, global minibatch k = the union of the ranks'
 * local minibatches k (a stratified form of the reference's shuffle; nothing but gradients and statistics crosses ranks).
 * 1 = the reference's literal scheme (ppo2/ppo2.hpp:288-307): ONE permutation of the B * world rows of all ranks per
 * epoch -- the same explicit `perms` [epochs][B * world] on every rank, or the same `seed` -- after an all-gather of the rollout
 * rows (ncclAllGather, once per ppo_update); rank r trains rows [r M, (r + 1) M) of every global minibatch of M * world rows.
 * A row index is e_global * T + t with e_global = rank * n_envs + e (runner.hpp:136-152 over the environments of all ranks). */
int ppo_dist_global_shuffle(ppo_handle* h, int on);
/* bf16 path under a communicator of more than one rank (collective library, not the peer regions): the gradient of a train step leaves in BUCKETS -- last layer + heads
 * first -- each bucket's ncclAllReduce on a second stream under the remaining backward and weight-gradient launches (default on; 0 = one all-reduce of the whole vector behind
 * them; 2 = also under a one-rank communicator, for measuring what the per-layer launches cost a rank).  Collective: call it at the same point on every rank.
 * No reference counterpart (its job is one process); DESIGN.md section 6. */
int ppo_dist_bucketed(ppo_handle* h, int on);

/* ---- measurement hooks ----------------------------------------------------------------------------------
 * per-kernel device time (ms) accumulated with hipEvents on the handle's stream since the last reset;
 * names/values are parallel arrays; returns the number of timed kernel classes */
int ppo_prof_enable(ppo_handle* h, int on);
int ppo_prof_read(ppo_handle* h, int max, char names[][32], double* total_ms, int64_t* launches);
int ppo_sync(ppo_handle* h);
/* which kernel VARIANT the calls so far took: the library picks its kernels from the shape, and a caller or a
 * test can ask which ones were ENQUEUED since ppo_create (a hipGraph capture counts once, its replays do not).  names: e.g.
 * "train8_kernel", "weight_grad_assemble_kernel", "narrow_train_kernel<static>", "narrow_epoch_kernel", "narrow_rollout1_kernel"; returns the count of entries */
int ppo_kernel_counts(ppo_handle* h, int max, char names[][32], int64_t* enqueued);

/* ---- debug: a raw device buffer by name, padding included (the getters above copy the dense part of every tensor) ----------------
 * "theta" "adam_m" "adam_v" (padded), "thetaT" / "par" (the transposed and small-parameter mirrors the train kernels read), "grad" (+ its tail), "sumsq", "beta_pow", "hyper",
 * "norm_out", "dw2_parts", the last train step's workspaces ("x0g" "dmug" "h_pi_0" ... "slots_pi" "slabs"), the gathered epoch ("mb_obs" ... "gidx" "advstats" "keys"), the
 * narrow path's packed image and second weight set ("nw_img" "nw_theta1" ...), the normaliser's state.  *count = the buffer's length in 4-byte words (0: not used by this shape);
 * at most max_count words are copied.  Two runs that must agree bit for bit are compared buffer by buffer with it (tests/test_other_shapes.py).  No reference counterpart. */
int ppo_debug_buffer(ppo_handle* h, const char* name, float* dst, int64_t max_count, int64_t* count);
/* debug: leave `word` in every LDS word of every CU (a launch of whole-CU workgroups on the handle's stream, synchronised).  A kernel that reads LDS it never wrote sees
 * what the previous workgroup on its CU left there; with a NaN pattern in place such a read shows in the results (tests/test_race_guards.py).  No reference counterpart. */
int ppo_debug_poison_lds(ppo_handle* h, uint32_t word);
/* debug: the node types of the hipGraph the last ppo_update captured: counts = {kernel, memset, memcpy, other}; -1 when the handle holds no graph.  The library's rule is
 * kernel nodes only (a memset node replayed out of order on ROCm 7.0.2: DESIGN.md section 9). */
int ppo_debug_graph_nodes(ppo_handle* h, int32_t counts[4]);
/* debug: raise the error word of the bf16 path's chained launch as a failed hand-off would; the next call that chains its layers must report it (ppo_step / ppo_value /
 * ppo_act_deterministic repeat their pass layer by layer; the rollout calls and ppo_update return the error) and the handle launches layer by layer from then on. */
int ppo_debug_raise_chain_error(ppo_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* PPO_HIP_H */
