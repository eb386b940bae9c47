// ppo_hip.hip -- C-ABI (include/ppo_hip.h) of libppo_hip.so: handle, padded parameter layout, HIP launch
// sequences and hipGraph replay for the PPO rollout-collect + minibatch-update hot path on MI355X (gfx950).
//
// There is NO CPU fallback in this library: every entry point needs a HIP device and fails loudly without one.
#include "../../include/ppo_hip.h"
#include "ppo_kernels.hpp"
#include "ppo_bf16.hpp"
#include "ppo_narrow.hpp"
#include "ppo_peer.hpp"
#include "ppo_dw2.hpp"
#include "ppo_train8.hpp"
#include "ppo_rollout1.hpp"

#include <dlfcn.h>
#include <atomic>
#if defined(__x86_64__)
#include <immintrin.h>     // _mm_sfence (host side of the VRAM inbox: drains the write-combining buffers behind the posted stores)
#endif

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

std::string g_create_error;
#ifdef PPO_STAMPS
unsigned long long* g_stamps = nullptr;
#endif

int ru(int x, int m) { return (x + m - 1) / m * m; }

struct Tensor {
    char name[32];
    int rows, cols;          // dense shape (cols 0 = 1-D of length rows)
    int prow, pcol;          // padded shape as stored ([prow][pcol], pcol = leading dimension)
    int off_dense, off_pad;  // offsets into the dense flat vector / the padded device vector
    int count() const { return rows * (cols ? cols : 1); }
    int dcols() const { return cols ? cols : 1; }
};

enum ProfClass { PK_STEP = 0, PK_TRAIN_FB, PK_DW, PK_REDUCE, PK_ADAM, PK_EPOCH, PK_STATS, PK_ENV, PK_GAE, PK_COMM, PK_COUNT };
const char* kProfNames[PK_COUNT] = {"policy_step", "train_fwd_bwd", "weight_grad", "grad_reduce", "adam", "epoch_prepare",
                                    "running_stats", "seeded_env", "gae", "allreduce"};

// which kernel VARIANT a call took (ppo_kernel_counts): the fast paths are chosen by shape, and a test must be able to say which one ran
enum KernelVariant { KV_TRAIN8 = 0, KV_TRAIN_FB, KV_DW2, KV_DW, KV_GRAD_REDUCE, KV_NARROW_TRAIN_STATIC, KV_NARROW_TRAIN, KV_NARROW_STEP_STATIC, KV_NARROW_STEP,
                     KV_POLICY_STEP, KV_ROLLOUT1, KV_ROLLOUT_PERSISTENT, KV_ROLLOUT_COOP, KV_COLLECT_FUSED, KV_BF16_TRAIN, KV_BF16_STEP, KV_BF16_REDUCE_ADAM, KV_NARROW_EPOCH, KV_COUNT };
const char* kVariantNames[KV_COUNT] = {"train8_kernel", "train_fwd_bwd_kernel", "weight_grad_assemble_kernel", "weight_grad_kernel", "grad_reduce_kernel",
                                       "narrow_train_kernel<static>", "narrow_train_kernel<runtime>", "narrow_step_kernel<static>", "narrow_step_kernel<runtime>",
                                       "policy_step_kernel", "narrow_rollout1_kernel", "narrow_rollout_kernel", "narrow_rollout_coop_kernel", "narrow_collect_kernel",
                                       "bf16_train_sequence", "bf16_step_sequence", "bf16_reduce_adam_kernel", "narrow_epoch_kernel"};

// RCCL entry points resolved at run time (the single-GPU path must not depend on librccl being loadable)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, char[128], int) = nullptr;   // ncclUniqueId is a 128-byte struct passed by value
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

}  // namespace

struct ppo_handle {
    ppo_config cfg{};
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    int CT = 1;                       // column tiles per wave in the dense layers (4 for wide nets)
    int CTH = 0;                      // split-K policy head column tiles (2 when Ap == 32 on the wide path), 0 = generic
    bool early = false;               // train kernel keeps the small products' weights in registers from kernel entry (18-obs / [256, ...] shape)
    NetDev net{};
    std::vector<Tensor> tensors;
    int P_dense = 0, P_pad = 0, n_blocks = 0, PT = 0;
    float* thetaT = nullptr;          // transposed copies (layout: NetDev::wT_off / wmuT_off)
    float* par = nullptr;             // small-parameter mirror [2][par_total] (layout: NetDev::par_*)
    // parameters + optimiser state (padded layout)
    float *theta = nullptr, *adam_m = nullptr, *adam_v = nullptr, *grad = nullptr, *sumsq = nullptr, *sumsq2 = nullptr;
    float* beta_pow = nullptr;        // {cur b1, cur b2, next b1, next b2}
    float* hyper = nullptr;           // {lr, cliprange}
    float* norm_out = nullptr;        // [1]
    GradSrc* grad_src = nullptr;
    int n_tiled = 0; AdamArgs::Tiled tiled[ADAM_MAX_TILED]{};   // matrices whose transposed copy adam_kernel writes tile by tile
    // train workspaces (sized for ws_rows minibatch rows)
    int ws_rows = 0;
    float* x0g = nullptr;
    float* hg[2][PPO_MAX_LAYERS]{};
    float* dyg[2][PPO_MAX_LAYERS]{};
    float* dmug = nullptr;
    float* slots[2]{};
    float* slabs = nullptr;
    int max_split = 8;
    int lds_step_total = 0;           // floats of dynamic LDS the act kernel needs (the train layout minus the gradient tiles)
    DwWork* dw_tiles = nullptr;
    int n_dw_tiles = 0;
    bool dw_has_big = false;
    // weight gradients + gradient assembly in one launch (ppo_dw2.hpp; 18-obs / [256,256] shape)
    bool dw2 = false;
    bool t8 = false;                  // 8-wave train kernel with K-split wave pairs (ppo_train8.hpp; same shape as dw2)
    // data-parallel adam_kernel<.., MEET>: the meeting's epoch words (+ error word) and the workgroups' partial sums of squares; PPO_HIP_NO_ADAM_MEET=1:
    // the round-4 sequence (all-reduce + grad_sumsq_kernel / push + sum kernels, then the plain adam_kernel)
    unsigned* adam_meet_words = nullptr; float* adam_meet_parts = nullptr; bool adam_meet = false;
    unsigned* dw2_counters = nullptr; float* dw2_parts = nullptr; SlotJob* dw2_jobs = nullptr; int dw2_n_jobs = 0, dw2_jpw = 0;
    // staging for host-pointer calls
    int st_rows = 0;
    float *st_obs = nullptr, *st_act = nullptr, *st_noise = nullptr, *st_vec[6]{};
    float* st_loss = nullptr;         // [8]
    // normaliser
    int nz_envs = 0;
    float nz_gamma = 0.99f, nz_clip_obs = 10.f, nz_clip_rew = 10.f, nz_eps = 1e-8f;
    NormDev obs_rms{}, ret_rms{};
    float* nz_ret = nullptr;
    float* stats_part = nullptr;      // per-workgroup (n, mean, M2) chunks of norm_batch_kernel
    float* stats_counter = nullptr;   // its arrival counter (one unsigned)
    float* stats_xch = nullptr;       // [world][(1+2D)+3] cross-rank batch moments (data-parallel running statistics)
    int stats_xch_world = 0;
    float* adv_xch = nullptr;         // [2*nminibatches] cross-rank advantage sums
    // rollout
    int E = 0, T = 0;
    float *ro_obs = nullptr, *ro_act = nullptr, *ro_val = nullptr, *ro_nlp = nullptr, *ro_done = nullptr, *ro_rew = nullptr,
          *ro_ret = nullptr;
    int done_staged = -1;             // ro_done[done_staged] already holds cur_done (written by norm_batch_kernel)
    float *cur_done = nullptr, *raw_obs = nullptr, *raw_rew = nullptr, *raw_done = nullptr, *last_val = nullptr;
    float* ro_noise = nullptr;        // [T,E,A] staging for explicit noise
    // update
    int* d_perms = nullptr; int* d_inv = nullptr; int* d_gidx = nullptr; float* d_advstats = nullptr;
    float *mb_obs = nullptr, *mb_act = nullptr, *mb_adv = nullptr, *mb_ret = nullptr, *mb_val = nullptr, *mb_nlp = nullptr;
    uint32_t* d_keys = nullptr; float* d_loss_rows = nullptr; float* d_loss_mean = nullptr;
    int upd_cap_rows = 0, upd_cap_steps = 0;
    hipGraphExec_t upd_graph = nullptr;
    hipGraph_t upd_graph_tmpl = nullptr;      // the captured graph the executable one was instantiated from: kept until the executable one goes (drop_graph)
    int g_epochs = 0, g_nmb = 0, g_E = 0, g_T = 0, g_explicit = -1, g_world = 0;
    bool use_graph = true;
    uint32_t rng_calls = 0;
    uint32_t rng_seed = 0x5EEDu;      // ppo_seed
    int norm_obs_flag = 1, norm_rew_flag = 1;   // EnvNormalize's norm_obs / norm_reward (env_normalize.hpp:75,95)
    float* env_in = nullptr;          // [E*O | E | E] raw obs | raw reward | dones of the current env step, one block: one H2D per env step
    float* hyper_host = nullptr;      // pinned {lr, cliprange}: the source of set_hyper's asynchronous copy
    float* pin_in = nullptr;          // pinned host mirror of env_in (hipHostMalloc, owned by the handle)
    // ONE environment behind a host Env, resident three-wave kernel: the transition ALSO goes straight into device memory through the BAR (posted writes) with its own
    // sequence word, so the kernel polls and reads local memory instead of host memory over PCIe (large-BAR devices; PPO_HIP_NO_VRAM_INBOX=1 keeps the pinned block only)
    float* vram_in = nullptr; unsigned* vram_h2d = nullptr;
    float* pin_out = nullptr;         // pinned host landing buffer for the actions of one env step [E*A]
    size_t pin_in_n = 0, pin_out_n = 0;
    // fused host-Env step for <= 32 environments (narrow_host_step_kernel): the kernel reads pin_in / writes pin_out itself
    unsigned* pin_flag = nullptr;     // pinned host word the step kernel raises when the actions are in pin_out
    unsigned act_seq = 0;
    bool host_pending = false; int host_pending_t = 0;   // a transition sits in pin_in, its bookkeeping rides in the next launch
    // ... and the RESIDENT form (narrow_rollout_kernel in host mode): one launch serves many env steps, host and kernel talk
    // through sequence words in pinned memory (pin_flag[PCTL_*])
    bool opt_no_host_fused = false, opt_no_host_resident = false;     // PPO_HIP_NO_HOST_FUSED / _RESIDENT, read once in ppo_create
    float* pin_in_dev = nullptr; float* pin_out_dev = nullptr; unsigned* pin_flag_dev = nullptr;   // the pinned blocks as the device sees them
    // general host-Env path: policy_step_kernel's policy-tower workgroups store their 16 rows of actions into pin_out themselves and raise their
    // word of this pinned table to wg_seq; ppo_rollout_act watches the table (PPO_HIP_NO_DIRECT_ACT=1: D2H copy + stream query as in round 4)
    unsigned* pin_wgflag = nullptr; unsigned* pin_wgflag_dev = nullptr; int pin_wgflag_n = 0; unsigned wg_seq = 0; bool opt_no_direct_act = false;
    bool host_proto = false;          // the last ppo_rollout_act used the resident / fused form: observe only posts the transition
    bool hp_active = false;           // a resident kernel may be running
    int hp_posted = 0;                // transitions posted in this rollout
    bool pin_in_busy = false;         // an H2D copy out of pin_in may still be in flight (cleared by every stream synchronisation of the rollout calls)
    int upd_cap_epochs = 0;
    // bf16 matrix-core path (ppo_config::compute_dtype == PPO_BF16; kernels in ppo_bf16.hpp)
    struct Bf16 {
        bool on = false;
        int Rcap = 0;                                   // row capacity of the workspaces (multiple of 128)
        bf16_t* theta_bf = nullptr;                     // straight cast of theta (same padded offsets): B operand of the forward ([K][J], transposed LDS reads)
                                                        // and of the backward products ([J][K] as it lies); no transposed mirror since round 3
        bf16_t *x0 = nullptr;
        bf16_t *xe = nullptr; int xe_rows = 0; bool epoch_staged = false;   // observations of a whole epoch staged once (bf16)
        int db_tiles = 0;                               // row tiles the bias-gradient table is sized for
        bf16_t *hb[2][PPO_MAX_LAYERS]{}, *dy[2][PPO_MAX_LAYERS]{};     // [Rcap][Hp_l]: tanh outputs and the gradients w.r.t. the pre-activations
        float* head_out[2]{};                           // [GB_HEAD_SPLIT][Rcap][Ap] fp32 partial products of the head GEMM's reduction ranges
        int head_split = 1;                             // ranges the last bf16_forward used
        // the fused assembly + Adam launch does not write the assembled gradient: whoever asks for it (ppo_get_last_grad, ppo_debug_buffer) has it rebuilt
        // from the last train step's slabs first (bf16_materialize_grad).  lazy_last: the last train step ENQUEUED was of that kind (a replayed graph keeps
        // what it captured: g_lazy); grad_lazy: h->grad is behind the last train step that RAN
        bool lazy_last = false, g_lazy = false, grad_lazy = false;
        ReduceArgs lazy_ra{}; int lazy_n_old = 0;
        bf16_t *dhead[2]{};
        float* dbias = nullptr; int db_off[2][PPO_MAX_LAYERS]{}; int n_dbias = 0;
        DwTileB* dw_tiles = nullptr; int n_dw_tiles = 0; int dw_wm = 4;
        int tile0[2][PPO_MAX_LAYERS + 1]{};             // first weight-gradient tile of every matrix ([L] = the head), in the order of the tile table
        int layer_tile0[PPO_MAX_LAYERS + 2]{};          // ... and of every layer's pair of matrices ([L] = the heads, [L + 1] = the table's end): the table is layer-major
        // data parallel over the collective library: gradient buckets (last layer + heads first) all-reduced on a second stream under the remaining backward / weight-gradient launches
        bool bucketed = true, bucketed_any_world = false; hipStream_t comm_stream = nullptr; hipEvent_t bk_ev[PPO_MAX_LAYERS + 1]{}; hipEvent_t bk_join = nullptr;
        // the hidden layers of a pass as ONE launch (gemm_chain_bf16_kernel): its word tables (forward, backward); PPO_HIP_NO_BF16_CHAIN=1: a launch per layer
        bool chain = false; unsigned* chain_words[2]{}; int n_cu = 0;
        bool chain_used = false; unsigned* chain_err_host = nullptr;     // a chained launch since the last check; pinned landing words of its two error words (act paths)
        // gradient assembly + clip + Adam in one persistent launch (bf16_reduce_adam_kernel; single GPU): its meeting's table [BRA_GRID] + error word
        bool fuse_ra = false; unsigned long long* ra_ent = nullptr;
    } bf;
    bool dev_shared = false;          // data parallel: another rank of the job runs on this device (single-rank kernels whose workgroups wait for each other -- the bf16 chain, the fused train launch -- are not used then; the peer forms' waits on other ranks are time-bounded instead: include/ppo_hip.h)
    // narrow-network path (every hidden width <= 64; kernels in ppo_narrow.hpp)
    bool narrow = false;
    NwLayout nw{};
    bool nw_static = false;
    float* nw_img = nullptr;          // [2][w_total] packed weight images (kept current by adam_kernel / transpose_refresh_kernel)
    float* nw_partials = nullptr; int nw_groups_cap = 0; int nw_stride = 0;
    // deferred Adam inside ppo_update (reference shape): second parameter / moment set and the step whose clip + Adam is pending
    float *nw_theta1 = nullptr, *nw_m1 = nullptr, *nw_v1 = nullptr;
    bool nw_lazy = false;             // the path is available (static shape, PPO_HIP_NO_LAZY_ADAM unset)
    // all minibatches of an epoch in one resident launch (narrow_epoch_kernel; minibatches of <= 32 NW_EPOCH_MAX_G rows on the deferred-Adam shapes, single GPU)
    bool nw_epoch = false; unsigned* nw_epoch_words = nullptr;
    bool nw_epoch_xl = false;         // ... its XCD-local form (workgroups 0, 8, 16, ... of the launch; ordinary stores / loads through one L2; a partial buffer per step)
    float* nw_epoch_partials = nullptr; size_t nw_epoch_cap = 0;
    bool adam_fast = false;           // PPO_HIP_ADAM_FAST=1: every Adam step of the handle uses the hardware's 1-ulp reciprocal / square root in the quotient (opt-in since round 6)
    bool adam_exact = false;          // a deferred-Adam handle (nw_lazy) without that switch: its deferred / resident forms use the correctly rounded quotient (the default)
    int nw_cur = 0;                   // parameter set holding the current weights (0 outside ppo_update)
    bool nw_pending = false; float* nw_pending_loss = nullptr; int nw_pending_parts = 0;
    float* nw_coop = nullptr; int nw_coop_G = 0;      // cooperative persistent rollout: [2][G][NW_COOP_PW] chunk moments, then {arrive, err}
    float* nw_alt = nullptr; double* nw_alt_counts = nullptr; int nw_alt_envs = 0;   // second env/normaliser state set of the fused collect step
    // dist
    Rccl rccl;
    void* comm = nullptr;
    int world = 1, rank = 0;
    bool graph_rccl = false;          // the collectives can be captured into the update's hipGraph (probed in ppo_dist_init)
    // literal data-parallel sampling (ppo_dist_global_shuffle): the rollout rows of all ranks, all-gathered once per update
    bool global_shuffle = false;
    float *gs_obs = nullptr, *gs_act = nullptr, *gs_ret = nullptr, *gs_val = nullptr, *gs_nlp = nullptr; int gs_rows = 0;
    // one-shot all-reduce over peer-mapped gather regions (ppo_peer.hpp; ppo_dist_peer_export / ppo_dist_peer_attach)
    struct Peer {
        bool on = false;                                // every collective that fits `cap` goes through the peer kernels
        bool usable = false;                            // the probe passed on every rank (ppo_dist_peer_enable may switch `on`)
        bool coarse = false;                            // the region is plain hipMalloc memory: the peer path is refused
        void* region = nullptr;                         // mine: [flag block | slots[2][world][cap]] (exported over IPC)
        size_t cap = 0;                                 // floats per slot (multiple of PEER_CHUNK)
        int scap = 0;                                   // floats per slot of the statistics area behind the slots
        void* mapped[PEER_MAX_WORLD]{};                 // the other ranks' regions as this process sees them
        unsigned* local = nullptr;                      // {seq, arrive, err}
        PeerDev dev{};
    } peer;
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_used;
    size_t ev_next = 0;
    double prof_ms[PK_COUNT]{};
    int64_t prof_n[PK_COUNT]{};
    int64_t kv[KV_COUNT]{};           // enqueues per kernel variant since ppo_create (a hipGraph capture counts once, its replays do not)
};

namespace {

int fail(ppo_handle* h, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return -1;
}

#define HIP_OK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// every entry point that allocates or launches makes the handle's device current on the calling thread first (another
// handle, torch or another thread may have changed it)
#define ENTER(h)                                                                                      \
    do {                                                                                             \
        if (hipSetDevice((h)->device) != hipSuccess) return fail(h, "hipSetDevice(%d) failed", (h)->device); \
    } while (0)

// The update's executable graph AND the captured graph it came from go together.  (Rounds 1 - 5 destroyed the captured graph right after hipGraphInstantiate, which
// is allowed; it is kept now because a runtime that held pointers into the captured graph's node parameters was one suspect for the open observation of DESIGN.md
// section 9.  It was NOT the cause -- the observation is unchanged with the graph kept -- but a few hundred KB per handle cost nothing and rule that class out.)
// Zero a few words from INSIDE the launch sequence: a kernel, not hipMemsetAsync.  The update's sequence is captured into a hipGraph and replayed, and a memset NODE in
// that graph is not safe on the HIP runtime the PyTorch wheel bundles (HIP 7.0.51831; it serves any process that imports torch before loading this library -- every full
// pytest run, every rank of bench.py -- while the system's 7.2 runtime replays the node correctly): a replay ran the node's fill in the MIDDLE of the kernels behind
// it -- weight_grad_assemble_kernel's arrival counters were cleared while its workgroups were counting, no tile found its last arriver, and from then on every tile was
// "finished" by whoever brought a half-counted word to 4 (found in round 6 through the oracle leg of tests/test_other_shapes.py's interleaved-handles test: both the
// handle run beside others AND the handle run alone were wrong by 1e-2 after enough replays, with the counters non-zero behind the update; eager launches and a graph
// without the node were right).  Rule: nothing but kernel nodes (and the collective library's own) in a captured sequence.  DESIGN.md section 9.
__global__ void zero_words_kernel(unsigned* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static int zero_words(ppo_handle* h, unsigned* p, int n) {
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, h->stream, p, n);
    HIP_OK(h, hipGetLastError());
    return 0;
}

static void drop_graph(ppo_handle* h) {
    if (h->upd_graph) { (void)hipGraphExecDestroy(h->upd_graph); h->upd_graph = nullptr; }
    if (h->upd_graph_tmpl) { (void)hipGraphDestroy(h->upd_graph_tmpl); h->upd_graph_tmpl = nullptr; }
}

template <typename T>
int dev_alloc(ppo_handle* h, T** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_OK(h, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    HIP_OK(h, hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream));
    return 0;
}

// ---- profiling (hipEvents on the handle's stream) ---------------------------------------------------------
struct ProfScope {
    ppo_handle* h; int cls; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(ppo_handle* h_, int c) : h(h_), cls(c) {
        if (!h->prof) return;
        if (h->ev_next + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; h->ev_pool.push_back(e); }
        }
        a = h->ev_pool[h->ev_next++]; b = h->ev_pool[h->ev_next++];
        (void)hipEventRecord(a, h->stream);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, h->stream);
        h->ev_used.push_back({cls, {a, b}});
    }
};

void prof_collect(ppo_handle* h) {
    if (h->ev_used.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto& u : h->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, u.second.first, u.second.second) == hipSuccess) { h->prof_ms[u.first] += ms; h->prof_n[u.first] += 1; }
    }
    h->ev_used.clear();
    h->ev_next = 0;
}

// ---- layout -------------------------------------------------------------------------------------------------
void add_tensor(ppo_handle* h, const char* name, int rows, int cols, int prow, int pcol, int& od, int& op, bool tile_align = false) {
    // a matrix with a transposed copy starts on a 1024-element boundary when it can be walked in 32 x 32 tiles (adam_kernel's tiled form)
    if (tile_align && prow % 32 == 0 && pcol % 32 == 0) op = ru(op, 1024);
    Tensor t{};
    snprintf(t.name, sizeof t.name, "%s", name);
    t.rows = rows; t.cols = cols; t.prow = prow; t.pcol = pcol;
    t.off_dense = od; t.off_pad = op;
    od += t.count();
    op += ru(prow * pcol, 256);
    h->tensors.push_back(t);
}

int build_layout(ppo_handle* h) {
    const ppo_config& c = h->cfg;
    NetDev& n = h->net;
    memset(&n, 0, sizeof n);
    n.O = c.obs_dim; n.A = c.act_dim; n.L = c.n_hidden;
    int minH = 1 << 30;
    for (int l = 0; l < n.L; ++l) { n.H[l] = c.hidden[l]; n.Hp[l] = ru(c.hidden[l], 16); minH = std::min(minH, n.Hp[l]); }
    h->CT = 1;
    if (minH >= 256) {
        bool ok = true;
        for (int l = 0; l < n.L; ++l) ok = ok && (n.Hp[l] % 64 == 0);
        if (ok) h->CT = 4;
    }
    if (h->CT == 4) for (int l = 0; l < n.L; ++l) n.Hp[l] = ru(n.Hp[l], 64);
    const int kq = h->CT == 4 ? 32 : 16;                  // reduction dims must be whole pipeline stages (16*KS)
    n.Kp0 = ru(c.obs_dim, kq); n.Ap = ru(c.act_dim, kq);
    // the reference's own network family -- two hidden layers of 64 behind up to 64 observations and up to 32 actions (18 / 18 and, with
    // observed velocities, 36 / 18: env/hexapod_closed_loop_env.hpp:20) -- takes the observation tile as 32 or 64 columns and the action
    // tile as 32, the two shapes the narrow kernels are instantiated for at compile time (padding weights are zero and stay zero)
    if (!h->bf.on && n.L == 2 && n.Hp[0] == 64 && n.Hp[1] == 64 && c.obs_dim <= 64 && c.act_dim <= 32) { n.Kp0 = c.obs_dim <= 32 ? 32 : 64; n.Ap = 32; }
    const bool bf = h->bf.on;
    if (bf) {                                             // GEMM path: every dimension is a whole number of 128-wide tiles
        for (int l = 0; l < n.L; ++l) n.Hp[l] = ru(c.hidden[l], GB_PAD);
        n.Kp0 = ru(c.obs_dim, GB_PAD); n.Ap = ru(c.act_dim, GB_PAD);
        h->CT = 1;
    }
    // split-K policy head: each of the 4 waves takes K/4 of the reduction in whole 64-wide stages
    h->CTH = (h->CT == 4 && n.Ap == 32 && n.Hp[n.L - 1] % 256 == 0) ? 2 : 0;
    n.ent_coef = c.ent_coef; n.vf_coef = c.vf_coef;
    int od = 0, op = 0;
    char nm[32];
    for (int l = 0; l < n.L; ++l) {
        const int in = l ? n.H[l - 1] : n.O, inp = l ? n.Hp[l - 1] : n.Kp0;
        const bool ta = !bf && l >= 1;                          // (layers >= 1 and the policy head have transposed copies on the fp32 paths)
        snprintf(nm, sizeof nm, "pi_fc%d/w", l); add_tensor(h, nm, in, n.H[l], inp, n.Hp[l], od, op, ta); n.w_off[0][l] = h->tensors.back().off_pad;
        snprintf(nm, sizeof nm, "pi_fc%d/b", l); n.b_off[0][l] = op; add_tensor(h, nm, n.H[l], 0, 1, n.Hp[l], od, op);
        snprintf(nm, sizeof nm, "vf_fc%d/w", l); add_tensor(h, nm, in, n.H[l], inp, n.Hp[l], od, op, ta); n.w_off[1][l] = h->tensors.back().off_pad;
        snprintf(nm, sizeof nm, "vf_fc%d/b", l); n.b_off[1][l] = op; add_tensor(h, nm, n.H[l], 0, 1, n.Hp[l], od, op);
    }
    const int HL = n.H[n.L - 1], HpL = n.Hp[n.L - 1];
    // bf16 path: the value head is stored like the policy head, [HpL][Ap] with only column 0 in use, so that both towers
    // run through the same batched GEMM launches (the padding columns are zero and provably stay zero)
    n.wv_off = op;  add_tensor(h, "vf/w", HL, 1, HpL, bf ? n.Ap : 1, od, op);
    n.bv_off = op;  add_tensor(h, "vf/b", 1, 0, 1, bf ? n.Ap : 1, od, op);
    add_tensor(h, "pi/w", HL, n.A, HpL, n.Ap, od, op, !bf); n.wmu_off = h->tensors.back().off_pad;
    n.bmu_off = op; add_tensor(h, "pi/b", n.A, 0, 1, n.Ap, od, op);
    n.ls_off = op;  add_tensor(h, "pi/logstd", 1, n.A, 1, n.Ap, od, op);
    h->P_dense = od; h->P_pad = op; h->n_blocks = op / 256;
    // transposed copies streamed by the backward pass
    int ot = 0;
    for (int tw = 0; tw < 2; ++tw)
        for (int l = 1; l < n.L; ++l) { n.wT_off[tw][l] = ot; ot += n.Hp[l] * n.Hp[l - 1]; }
    n.wmuT_off = ot; ot += n.Ap * n.Hp[n.L - 1];
    h->PT = ot;
    n.n_theta = op; n.n_thetaT = ot;
    // small-parameter mirror layout (also the LDS copy): biases | b_mu | logstd | w_v | b_v
    {
        int po = 0;
        for (int l = 0; l < n.L; ++l) { n.par_b[l] = po; po += n.Hp[l]; }
        n.par_bmu = po; po += n.Ap;
        n.par_ls = po; po += n.Ap;
        n.par_wv = po; po += n.Hp[n.L - 1];
        n.par_bv = po; po += 4;
        n.par_total = po;
    }
    // LDS carve.  Regular form: one activation tile per layer (the backward pass reads tanh outputs from LDS).
    int hmax = std::max(n.Ap, n.Kp0);
    for (int l = 0; l < n.L; ++l) hmax = std::max(hmax, n.Hp[l]);
    auto carve = [&](bool wide) {
        int o = 0;
        n.wide = wide ? 1 : 0;
        if (!wide) {
            n.lds_h[0] = o; o += ROWS_PER_BLOCK * (n.Kp0 + LDS_PAD);
            for (int l = 0; l < n.L; ++l) { n.lds_h[l + 1] = o; o += ROWS_PER_BLOCK * (n.Hp[l] + LDS_PAD); }
            n.lds_head = 0; n.par_skip = 0;
        } else {
            // Wide form (nets whose per-layer tiles do not fit 160 KB): two ping-pong tiles; layer l reads tile l%2 and
            // writes tile (l+1)%2; the backward pass re-reads tanh outputs from HBM (they are stored for the weight
            // gradients anyway); biases are read from the global mirror; generic policy head (no split-K scratch).
            const int tile = ROWS_PER_BLOCK * (hmax + LDS_PAD);
            const int t0 = o; o += tile;
            const int t1 = o; o += tile;
            for (int l = 0; l <= n.L; ++l) n.lds_h[l] = (l % 2) ? t1 : t0;
            n.lds_d[1] = ((n.L + 1) % 2) ? t1 : t0;                        // first backward output: the tile h_L is not in
            n.par_skip = n.par_bmu;
        }
        n.lds_mu = o; o += ROWS_PER_BLOCK * (n.Ap + LDS_PAD);
        if (!wide) { n.lds_head = o; o += 4 * ROWS_PER_BLOCK * n.Ap; }
        n.lds_misc = o; o += 64 + 2 * ROWS_PER_BLOCK * n.Ap + 64;
        n.lds_par = o; o += n.par_total - n.par_skip;
        // the gradient tiles come LAST: the act kernel never touches them and is launched with the smaller size, which
        // lets two of its workgroups share a compute unit
        h->lds_step_total = o;
        if (!wide) {
            n.lds_d[0] = o; o += ROWS_PER_BLOCK * (hmax + LDS_PAD);
            n.lds_d[1] = o; o += ROWS_PER_BLOCK * (hmax + LDS_PAD);
        } else {
            n.lds_d[0] = o; o += ROWS_PER_BLOCK * (n.Ap + LDS_PAD);       // d mu tile
        }
        n.lds_total = o;
        return (size_t)o * sizeof(float) <= 160 * 1024;
    };
    if (bf) {
        int d = 0;
        for (int tw = 0; tw < 2; ++tw)
            for (int l = 0; l < n.L; ++l) { h->bf.db_off[tw][l] = d; d += n.Hp[l]; }
        h->bf.n_dbias = d;
        // weight-gradient tiles (needed by upload_grad_src, i.e. before bf16_create): 256 features tall when every reduction-side width
        // allows it; tile0 = first tile of every matrix in the order of the tile table (bf16_ensure_ws)
        ppo_handle::Bf16& b = h->bf;
        b.dw_wm = 4;
        if (n.Kp0 % 256) b.dw_wm = 2;
        for (int l = 0; l < n.L; ++l) if (n.Hp[l] % 256) b.dw_wm = 2;
        // LAYER-major since round 6 (rounds 2 - 5: tower-major): layer l's matrices of both towers, then layer l + 1's, the heads last -- the tiles of a layer (and of
        // "last layer + heads") are then one contiguous range of the table, which is what the bucketed data-parallel step launches by itself (bf16_train_bucketed)
        int t0 = 0;
        for (int l = 0; l <= n.L; ++l) {
            b.layer_tile0[l] = t0;
            for (int t = 0; t < 2; ++t) {
                b.tile0[t][l] = t0;
                t0 += l < n.L ? ((l ? n.Hp[l - 1] : n.Kp0) / GB_BM(b.dw_wm)) * (n.Hp[l] / GB_N) : (n.Hp[n.L - 1] / GB_BM(b.dw_wm)) * (n.Ap / GB_N);
            }
        }
        b.layer_tile0[n.L + 1] = t0;
        n.lds_total = 0; h->lds_step_total = 0; h->CTH = 0;
        n.slot_head = 0; n.slot_aux = n.Ap; n.slot_loss = 2 * n.Ap; n.slot_w = 2 * n.Ap + 8;
        return 0;
    }
    if (!carve(false)) {
        if (!carve(true))
            return fail(h, "network too wide even for the two-tile LDS layout (%zu bytes needed, 163840 available)", (size_t)n.lds_total * 4);
        h->CTH = 0;
    }
    // slot layout
    int s = 0;
    for (int l = 0; l < n.L; ++l) { n.slot_db[l] = s; s += n.Hp[l]; }
    n.slot_head = s; s += std::max(HpL, n.Ap);
    n.slot_aux = s; s += n.Ap;
    n.slot_loss = s; s += 8;
    n.slot_w = ru(s, 4);
    return 0;
}

// LDS image of the narrow path: the tower's weights (both directions) + small parameters, then NW_PIPES sets of tiles
void build_narrow_layout(ppo_handle* h) {
    const NetDev& n = h->net;
    h->narrow = false;
    if (h->bf.on || n.L > NW_MAXL || n.Kp0 > 64 || n.Ap > 64) return;
    for (int l = 0; l < n.L; ++l) if (n.Hp[l] > 64) return;
    const char* off = getenv("PPO_HIP_NO_NARROW");
    if (off && off[0] == '1') return;
    NwLayout lay{};
    int o = 0;
    auto region = [&](int rows, int cols, int pad, int& dst, int& ld) { dst = o; ld = cols + pad; o += rows * (cols + pad); };
    int dummy;
    for (int l = 0; l < n.L; ++l) region(l ? n.Hp[l - 1] : n.Kp0, n.Hp[l], NW_WPAD, lay.wf[l], lay.wf_ld[l]);
    region(n.Hp[n.L - 1], n.Ap, NW_WPAD, lay.wh, lay.wh_ld);
    region(1, n.par_total, 0, lay.par, dummy);
    lay.w_fwd = ru(o, 4); o = lay.w_fwd;
    for (int l = 1; l < n.L; ++l) region(n.Hp[l], n.Hp[l - 1], NW_WPAD, lay.wt[l], lay.wt_ld[l]);
    region(n.Ap, n.Hp[n.L - 1], NW_WPAD, lay.wht, lay.wht_ld);
    lay.w_total = ru(o, 4);
    int p = 0;
    lay.x[0] = p; lay.ldx[0] = n.Kp0 + NW_XPAD; p += 16 * lay.ldx[0];
    for (int l = 0; l < n.L; ++l) { lay.x[l + 1] = p; lay.ldx[l + 1] = n.Hp[l] + NW_XPAD; p += 16 * lay.ldx[l + 1]; }
    for (int l = 0; l < n.L; ++l) { lay.dy[l] = p; lay.ldy[l] = n.Hp[l] + NW_XPAD; p += 16 * lay.ldy[l]; }
    lay.ldm = n.Ap + NW_XPAD;
    lay.mu = p; p += 16 * lay.ldm;
    lay.dmu = p; p += 16 * lay.ldm;
    lay.acts = p; p += 16 * n.Ap;
    lay.dls = p; p += 16 * n.Ap;
    lay.rowv = p; p += 32;
    lay.misc = p; p += 64;
    lay.pipe_total = ru(p, 4);
    lay.lds_total = lay.w_total + NW_PIPES * lay.pipe_total;
    if ((size_t)lay.lds_total * sizeof(float) > 160 * 1024) return;
    h->nw = lay;
    h->nw_stride = ru(h->P_pad + 8, 64);
    // compile-time shape of the reference's own network (18 obs / 18 act padded to 32, [64,64]); anything else runs the
    // runtime-shape instantiation
    h->nw_static = n.L == 2 && (n.Kp0 == 32 || n.Kp0 == 64) && n.Ap == 32 && n.Hp[0] == 64 && n.Hp[1] == 64;
    h->narrow = true;
}

int upload_grad_src(ppo_handle* h) {
    const NetDev& n = h->net;
    // chunks no tensor covers (alignment gaps): no gradient, no mirrors
    GradSrc none{2, 0, 0, 0, 0, -1, 0, 0, -1, 0, -1, 0, -1, 0, -1, -1};
    std::vector<GradSrc> src(h->n_blocks, none);
    for (size_t b = 0; b < src.size(); ++b) src[b].base = (int)b * 256;
    h->n_tiled = 0;
    for (const Tensor& t : h->tensors) {
        GradSrc g{2, 0, 0, 0, t.off_pad, -1, t.prow, t.pcol, -1, 0, -1, 0, -1, 0, -1, -1};
        const std::string nm = t.name;
        int l = -1;
        if (nm.find("_fc") != std::string::npos) l = atoi(nm.c_str() + 5);
        const int tower = nm[0] == 'v' ? 1 : 0;
        g.tower = tower;
        if (nm.size() > 2 && nm.substr(nm.size() - 2) == "/w" && nm != "vf/w") {
            g.kind = 0;
            if (nm == "pi/w") g.t_off = n.wmuT_off;
            else if (l >= 1) g.t_off = n.wT_off[tower][l];
        }
        else if (l >= 0) { g.kind = 1; g.tower = tower; g.slot_off = n.slot_db[l]; g.count = n.Hp[l]; g.p_off = tower * n.par_total + n.par_b[l]; g.p_count = n.Hp[l]; }
        else if (nm == "vf/w") { g.kind = 1; g.tower = 1; g.slot_off = n.slot_head; g.count = n.Hp[n.L - 1]; g.p_off = n.par_total + n.par_wv; g.p_count = n.Hp[n.L - 1]; }
        else if (nm == "vf/b") { g.kind = 1; g.tower = 1; g.slot_off = n.slot_aux; g.count = 1; g.p_off = n.par_total + n.par_bv; g.p_count = 1; }
        else if (nm == "pi/b") { g.kind = 1; g.tower = 0; g.slot_off = n.slot_head; g.count = n.Ap; g.p_off = n.par_bmu; g.p_count = n.Ap; }
        else if (nm == "pi/logstd") { g.kind = 1; g.tower = 0; g.slot_off = n.slot_aux; g.count = n.Ap; g.p_off = n.par_ls; g.p_count = n.Ap; }
        if (h->bf.on) {
            // matrices (value head included) come from the split-K slabs of the grouped weight-gradient GEMM, hidden biases
            // from the row-sum kernel's vector (kind 3), head bias / logstd / value bias from the loss kernel's per-block slots;
            // the fp32 transposed copies and the small-parameter mirror do not exist on this path
            g.t_off = -1; g.p_off = -1; g.p_count = 0;
            if (nm.size() > 2 && nm.substr(nm.size() - 2) == "/w") { g.kind = 0; g.tile0 = h->bf.tile0[tower][l >= 0 ? l : n.L]; }
            else if (l >= 0) { g.kind = 3; g.slot_off = h->bf.db_off[tower][l]; g.count = n.Hp[l]; }
            else if (nm == "vf/b") { g.kind = 1; g.tower = 1; g.slot_off = n.slot_head; g.count = 1; }
            else if (nm == "pi/b") { g.kind = 1; g.tower = 0; g.slot_off = n.slot_head; g.count = n.Ap; }
            else if (nm == "pi/logstd") { g.kind = 1; g.tower = 0; g.slot_off = n.slot_aux; g.count = n.Ap; }
        }
        g.i_off = g.it_off = g.ip_off = -1; g.i_ld = g.it_ld = 0;
        if (h->narrow) {
            const NwLayout& lay = h->nw;
            const int ib = tower * lay.w_total;
            const bool is_w = nm.size() > 2 && nm.substr(nm.size() - 2) == "/w";
            if (is_w && l >= 0) {
                g.i_off = ib + lay.wf[l]; g.i_ld = lay.wf_ld[l];
                if (l >= 1) { g.it_off = ib + lay.wt[l]; g.it_ld = lay.wt_ld[l]; }
            } else if (nm == "pi/w") { g.i_off = lay.wh; g.i_ld = lay.wh_ld; g.it_off = lay.wht; g.it_ld = lay.wht_ld; }
            else if (g.p_off >= 0) g.ip_off = ib + lay.par + (g.p_off - tower * n.par_total);
        }
        if (g.t_off >= 0 && t.off_pad % 1024 == 0 && t.prow % 32 == 0 && t.pcol % 32 == 0 && h->n_tiled < ADAM_MAX_TILED)
            h->tiled[h->n_tiled++] = AdamArgs::Tiled{t.off_pad, t.prow * t.pcol, t.pcol, t.prow, g.t_off};
        const int nb = ru(t.prow * t.pcol, 256) / 256;
        for (int b = 0; b < nb; ++b) src[t.off_pad / 256 + b] = g;
    }
    if (dev_alloc(h, &h->grad_src, src.size())) return -1;
    HIP_OK(h, hipMemcpyAsync(h->grad_src, src.data(), src.size() * sizeof(GradSrc), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int bf16_ensure_ws(ppo_handle* h, int rows);
int ensure_train_ws(ppo_handle* h, int rows) {
    if (h->bf.on) return bf16_ensure_ws(h, rows);
    if (h->narrow) {
        const int groups = (rows + NW_ROWS - 1) / NW_ROWS;
        if (groups <= h->nw_groups_cap) return 0;
        drop_graph(h);
        HIP_OK(h, hipStreamSynchronize(h->stream));
        if (dev_alloc(h, &h->nw_partials, (size_t)4 * groups * h->nw_stride)) return -1;     // zero-filled: padding elements stay zero ([2 towers][groups]; twice: narrow_epoch_kernel alternates two sets)
        h->nw_groups_cap = groups;
        h->ws_rows = std::max(h->ws_rows, groups * NW_ROWS);
        return 0;
    }
    rows = ru(rows, h->dw2 ? DW2_CH : 16);                 // (weight_grad_assemble_kernel walks whole 64-row chunks: the train kernel's grid is padded to them)
    if (rows <= h->ws_rows) return 0;
    drop_graph(h);
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const NetDev& n = h->net;
    if (dev_alloc(h, &h->x0g, (size_t)rows * n.Kp0)) return -1;
    for (int t = 0; t < 2; ++t)
        for (int l = 0; l < n.L; ++l) {
            if (dev_alloc(h, &h->hg[t][l], (size_t)rows * n.Hp[l])) return -1;
            if (dev_alloc(h, &h->dyg[t][l], (size_t)rows * n.Hp[l])) return -1;
        }
    if (dev_alloc(h, &h->dmug, (size_t)rows * n.Ap)) return -1;
    for (int t = 0; t < 2; ++t)
        if (dev_alloc(h, &h->slots[t], (size_t)(rows / 16) * n.slot_w)) return -1;
    if (!h->slabs && dev_alloc(h, &h->slabs, (size_t)h->max_split * h->P_pad)) return -1;
    // weight-gradient work table
    std::vector<DwTile> big, rest, strips;
    h->dw_has_big = false;
    auto add = [&](const float* X, int ldx, const float* dY, int ldy, int Kp, int Np, int out_off) {
        if (Kp % 64 == 0 && Np % 64 == 0) {                       // 64x64 tiles
            for (int i = 0; i < Kp; i += 64) for (int j = 0; j < Np; j += 64) big.push_back(DwTile{X, dY, ldx, ldy, i, j, out_off, Np, 0});
        } else if (Kp == 32 && Np % 16 == 0) {                    // narrow first layer: [32 x 16] strips
            for (int j = 0; j < Np; j += 16) strips.push_back(DwTile{X, dY, ldx, ldy, 0, j, out_off, Np, 4});
        } else if (Np == 32 && Kp % 16 == 0) {                    // narrow head: [16 x 32] strips
            for (int i = 0; i < Kp; i += 16) strips.push_back(DwTile{X, dY, ldx, ldy, i, 0, out_off, Np, 5});
        } else {
            for (int i = 0; i < Kp; i += 16) for (int j = 0; j < Np; j += 16) rest.push_back(DwTile{X, dY, ldx, ldy, i, j, out_off, Np, 1});
        }
    };
    for (int t = 0; t < 2; ++t)
        for (int l = 0; l < n.L; ++l) {
            const float* X = l ? h->hg[t][l - 1] : h->x0g;
            const int Kp = l ? n.Hp[l - 1] : n.Kp0;
            add(X, Kp, h->dyg[t][l], n.Hp[l], Kp, n.Hp[l], n.w_off[t][l]);
        }
    add(h->hg[0][n.L - 1], n.Hp[n.L - 1], h->dmug, n.Ap, n.Hp[n.L - 1], n.Ap, n.wmu_off);
    std::vector<DwWork> work;
    for (const DwTile& b : big) { DwWork w{}; w.main = b; w.n_extra = 0; work.push_back(w); }
    h->dw_has_big = !big.empty();
    if (!big.empty() && strips.size() <= 2 * big.size()) {        // fold the strips into the big workgroups, round robin
        for (size_t i = 0; i < strips.size(); ++i) { DwWork& w = work[i % big.size()]; w.extra[w.n_extra++] = strips[i]; }
        for (DwWork& w : work)                                     // dw_main_with_strips expects {first-layer strip[, head strip]}
            w.fused = w.main.cls == 0 && ((w.n_extra == 0) || (w.n_extra == 1 && w.extra[0].cls == 4) ||
                                          (w.n_extra == 2 && w.extra[0].cls == 4 && w.extra[1].cls == 5));
    } else {
        for (const DwTile& s : strips) { DwWork w{}; w.main = s; work.push_back(w); }
    }
    for (const DwTile& r : rest) { DwWork w{}; w.main = r; work.push_back(w); }
    h->n_dw_tiles = (int)work.size();
    if (dev_alloc(h, &h->dw_tiles, work.size())) return -1;
    HIP_OK(h, hipMemcpyAsync(h->dw_tiles, work.data(), work.size() * sizeof(DwWork), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    h->ws_rows = rows;
    return 0;
}

int ensure_staging(ppo_handle* h, int rows) {
    if (rows <= h->st_rows) return 0;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const NetDev& n = h->net;
    if (dev_alloc(h, &h->st_obs, (size_t)rows * n.O)) return -1;
    if (dev_alloc(h, &h->st_act, (size_t)rows * n.A)) return -1;
    if (dev_alloc(h, &h->st_noise, (size_t)rows * n.A)) return -1;
    for (int i = 0; i < 6; ++i) if (dev_alloc(h, &h->st_vec[i], (size_t)rows)) return -1;
    h->st_rows = rows;
    return 0;
}

ObsNorm no_norm_fwd() { return ObsNorm{nullptr, nullptr, 0.f, 0.f, 0}; }

// ---- bf16 matrix-core path (ppo_bf16.hpp) ------------------------------------------------------------------------
// bf16 operand copy of the fp32 master weights: a cast of the whole padded vector (adam_kernel keeps it current afterwards)
int bf16_refresh_mirrors(ppo_handle* h) {
    ppo_handle::Bf16& b = h->bf;
    const size_t n4 = (size_t)h->P_pad / 4;
    hipLaunchKernelGGL(bf16_cast_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, h->stream, h->theta, b.theta_bf, n4);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int bf16_create(ppo_handle* h) {
    ppo_handle::Bf16& b = h->bf;
    if (dev_alloc(h, &b.theta_bf, (size_t)h->P_pad)) return -1;
    bool ok = true;
    auto lds_attr = [&](const void* f, int bytes) { ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; };
    lds_attr((const void*)gemm_nt_bf16_kernel<4, GEPI_TANH>, GB_LDS_BYTES(4)); lds_attr((const void*)gemm_nt_bf16_kernel<2, GEPI_TANH>, GB_LDS_BYTES(2));
    lds_attr((const void*)gemm_nt_bf16_kernel<4, GEPI_TANHGRAD>, GB_LDS_BYTES(4)); lds_attr((const void*)gemm_nt_bf16_kernel<2, GEPI_TANHGRAD>, GB_LDS_BYTES(2));
    lds_attr((const void*)bf16_heads_kernel<true, 1>, BH_LDS_BYTES); lds_attr((const void*)bf16_heads_kernel<true, 2>, BH_LDS_BYTES);
    lds_attr((const void*)bf16_heads_kernel<true, 4>, BH_LDS_BYTES); lds_attr((const void*)bf16_heads_kernel<true, 8>, BH_LDS_BYTES);
    lds_attr((const void*)gemm_dw_bf16_kernel<4>, GB_LDS_BYTES(4)); lds_attr((const void*)gemm_dw_bf16_kernel<2>, GB_LDS_BYTES(2));
    lds_attr((const void*)gemm_chain_bf16_kernel<GEPI_TANH>, GB_LDS_BYTES(4)); lds_attr((const void*)gemm_chain_bf16_kernel<GEPI_TANHGRAD>, GB_LDS_BYTES(4));
    if (!ok) return fail(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the bf16 GEMM kernels");
    { const char* e = getenv("PPO_HIP_NO_BF16_CHAIN");
      b.chain = !(e && e[0] == '1');
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) b.n_cu = prop.multiProcessorCount;
      for (int d = 0; d < 2 && b.chain; ++d) if (dev_alloc(h, &b.chain_words[d], (size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS + 32)) return -1; }
    // the persistent assembly + Adam launch needs BRA_GRID workgroups of 1024 threads resident at once: one per CU of a whole device
    b.fuse_ra = b.n_cu >= BRA_GRID && (h->n_blocks + 1 + BGR_WAVES - 1) / BGR_WAVES <= 8 * BRA_GRID;
    if (b.fuse_ra && dev_alloc(h, &b.ra_ent, BRA_GRID + 8)) return -1;
    return 0;
}

// The layers of a pass in one launch (gemm_chain_bf16_kernel): `n` links of the same output shape [I][J]; false when the shape does not qualify
// (256-row tiles, every row group of tiles_j workgroups on one XCD, all workgroups resident at once) -- the caller then launches layer by layer.
template <int EPI>
bool bf16_chain(ppo_handle* h, const GemmArgs* links, int n, int I, int J, int which) {
    ppo_handle::Bf16& b = h->bf;
    if (!b.chain || h->dev_shared || n < 2 || n > GB_CHAIN_MAX || I % 256 != 0 || J % GB_N != 0) return false;
    const int tiles_i = I / 256, tiles_j = J / GB_N, G = 2 * tiles_i;
    if (G % 8 != 0 || G > 64 || tiles_j > 16 || G * tiles_j > b.n_cu) return false;
    ChainArgs ca{};
    ca.n = n; ca.tiles_i = tiles_i; ca.tiles_j = tiles_j;
    ca.words = b.chain_words[which] + (size_t)(G / 8 - 1) * GB_CHAIN_WORDS;      // a table per number of row groups (ChainArgs::words)
    ca.err = b.chain_words[which] + (size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS;
    for (int l = 0; l < n; ++l) { ca.link[l] = links[l]; ca.link[l].tiles_i = tiles_i; }
    hipLaunchKernelGGL((gemm_chain_bf16_kernel<EPI>), dim3(G * tiles_j), dim3(GB_THREADS(4)), GB_LDS_BYTES(4), h->stream, ca);
    b.chain_used = true;
    return hipGetLastError() == hipSuccess;
}

// after a stream synchronisation: did a chained launch time out or find a row group spread over two XCDs?
int bf16_chain_check(ppo_handle* h) {
    ppo_handle::Bf16& b = h->bf;
    if (b.on && b.fuse_ra && b.ra_ent) {
        unsigned long long e = 0;
        HIP_OK(h, hipMemcpy(&e, b.ra_ent + BRA_GRID, sizeof e, hipMemcpyDeviceToHost));
        if (e) {
            (void)hipMemset(b.ra_ent, 0, (BRA_GRID + 8) * sizeof e);
            b.fuse_ra = false;
            drop_graph(h);
            return fail(h, "bf16_reduce_adam_kernel: its 256 workgroups were not resident together within ~1 s (is another process using this GPU?); this step's results "
                           "are invalid.  The handle now launches bf16_grad_reduce_kernel and adam_kernel (PPO_HIP_NO_REDUCE_ADAM=1 selects them from the start)");
        }
    }
    if (!b.on || !b.chain) return 0;
    b.chain_used = false;
    for (int d = 0; d < 2; ++d) {
        unsigned e = 0;
        if (!b.chain_words[d]) continue;
        HIP_OK(h, hipMemcpy(&e, b.chain_words[d] + (size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS, sizeof e, hipMemcpyDeviceToHost));
        if (e) {
            for (int k = 0; k < 2; ++k) (void)hipMemset(b.chain_words[k], 0, ((size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS + 32) * sizeof e);
            b.chain = false;
            drop_graph(h);
            return fail(h, "gemm_chain_bf16_kernel: %s; this call's results are invalid.  The handle now launches layer by layer (PPO_HIP_NO_BF16_CHAIN=1 selects "
                           "that from the start)", e == 2 ? "a row group's workgroups were not on one XCD (the launch's workgroup dealing is not round-robin over the XCDs here)"
                                                          : "its workgroups were not resident together within ~1 s (is another process using this GPU?)");
        }
    }
    return 0;
}

// The act paths (ppo_step / ppo_value / the rollout calls) chain their layers too: the error words ride out with the call's own results -- two 4-byte copies
// into pinned memory IN FRONT of the call's stream synchronisation (bf16_chain_err_async), looked at right behind it (bf16_chain_err_test: 1 = this call's
// results are invalid; the handle launches layer by layer from now on and says so) -- so a failed hand-off is an error of the call that produced the garbage,
// not of the next update.
int bf16_chain_err_async(ppo_handle* h) {
    ppo_handle::Bf16& b = h->bf;
    if (!b.on || !b.chain || !b.chain_used) return 0;
    if (!b.chain_err_host) { HIP_OK(h, hipHostMalloc((void**)&b.chain_err_host, 64, hipHostMallocDefault)); b.chain_err_host[0] = b.chain_err_host[1] = 0u; }
    for (int d = 0; d < 2; ++d)
        if (b.chain_words[d]) HIP_OK(h, hipMemcpyAsync(b.chain_err_host + d, b.chain_words[d] + (size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    return 0;
}
int bf16_chain_err_test(ppo_handle* h) {                        // (after the stream synchronisation that follows bf16_chain_err_async)
    ppo_handle::Bf16& b = h->bf;
    if (!b.on || !b.chain || !b.chain_used || !b.chain_err_host) return 0;
    b.chain_used = false;
    if (!(b.chain_err_host[0] | b.chain_err_host[1])) return 0;
    b.chain_err_host[0] = b.chain_err_host[1] = 0u;
    return bf16_chain_check(h) ? 1 : 0;                         // (reads the words again, resets them, switches the chain off, sets the message)
}

// the assembled gradient of the last train step, when the launch that applied it did not write it (bf16_reduce_adam_kernel): the slabs, slot rows and bias-gradient
// rows it was summed from are still the last step's, and bf16_grad_reduce_kernel adds them exactly as the fused launch did (same bits)
int bf16_materialize_grad(ppo_handle* h) {
    ppo_handle::Bf16& b = h->bf;
    if (!b.grad_lazy) return 0;
    hipLaunchKernelGGL(bf16_grad_reduce_kernel, dim3(b.lazy_n_old), dim3(64 * BGR_WAVES), 0, h->stream, b.lazy_ra);
    HIP_OK(h, hipGetLastError());
    b.grad_lazy = false;
    return 0;
}

int bf16_ensure_ws(ppo_handle* h, int rows) {
    ppo_handle::Bf16& b = h->bf;
    const int R = ru(rows, GB_PAD);
    if (R <= b.Rcap) return 0;
    if (bf16_materialize_grad(h)) return -1;                 // (the slot rows and bias-gradient rows it would be rebuilt from are about to be reallocated)
    drop_graph(h);
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const NetDev& n = h->net;
    if (dev_alloc(h, &b.x0, (size_t)R * n.Kp0)) return -1;
    for (int t = 0; t < 2; ++t) {
        for (int l = 0; l < n.L; ++l) {
            const size_t cnt = (size_t)R * n.Hp[l];
            if (dev_alloc(h, &b.hb[t][l], cnt) || dev_alloc(h, &b.dy[t][l], cnt)) return -1;
        }
        if (dev_alloc(h, &b.head_out[t], (size_t)GB_HEAD_SPLIT * R * n.Ap) || dev_alloc(h, &b.dhead[t], (size_t)R * n.Ap)) return -1;
        if (dev_alloc(h, &h->slots[t], (size_t)(R / 16) * n.slot_w)) return -1;
    }
    if (!h->slabs && dev_alloc(h, &h->slabs, (size_t)h->max_split * h->P_pad)) return -1;
    b.db_tiles = R / 128;
    if (dev_alloc(h, &b.dbias, (size_t)b.db_tiles * b.n_dbias)) return -1;
    // grouped weight-gradient tile table: dW = X^T dY for every layer and both heads, operands as they lie ([rows][features])
    std::vector<DwTileB> tiles;
    auto add = [&](const bf16_t* A, const bf16_t* B, int Kp, int Np, int out_off) {
        for (int i = 0; i < Kp; i += GB_BM(b.dw_wm)) for (int j = 0; j < Np; j += GB_N) tiles.push_back(DwTileB{A, B, Kp, Np, i, j, out_off, Np, A == b.x0 ? 1 : 0});
    };
    for (int l = 0; l <= n.L; ++l)                             // layer-major (tile0 / layer_tile0 above): both towers' matrices of a layer side by side, the heads last
        for (int t = 0; t < 2; ++t) {
            if (l < n.L) add(l ? b.hb[t][l - 1] : b.x0, b.dy[t][l], l ? n.Hp[l - 1] : n.Kp0, n.Hp[l], n.w_off[t][l]);
            else add(b.hb[t][n.L - 1], b.dhead[t], n.Hp[n.L - 1], n.Ap, t ? n.wv_off : n.wmu_off);
        }
    b.n_dw_tiles = (int)tiles.size();
    if (dev_alloc(h, &b.dw_tiles, tiles.size())) return -1;
    HIP_OK(h, hipMemcpyAsync(b.dw_tiles, tiles.data(), tiles.size() * sizeof(DwTileB), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    b.Rcap = R;
    return 0;
}

template <int EPI>
int bf16_gemm(ppo_handle* h, const GemmArgs& a, int I, int J) {
    GemmArgs g = a;
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    g.stamps = (I / 256) * (J / GB_N) <= 256 ? g_stamps : nullptr;
#endif
    if (I % 256 == 0) {                                      // 256 x 128 tiles, 8 waves
        g.tiles_i = I / 256;
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<4, EPI>), dim3((I / 256) * (J / GB_N), 2), dim3(GB_THREADS(4)), GB_LDS_BYTES(4), h->stream, g);
    } else {
        g.tiles_i = I / 128;
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<2, EPI>), dim3((I / 128) * (J / GB_N), 2), dim3(GB_THREADS(2)), GB_LDS_BYTES(2), h->stream, g);
    }
    HIP_OK(h, hipGetLastError());
    return 0;
}

// forward of both towers on `rows` staged rows: hidden layers (bias + tanh) and the padded heads (fp32 out)
int bf16_forward(ppo_handle* h, int Rp, const bf16_t* x0_override = nullptr) {
    ppo_handle::Bf16& b = h->bf;
    const NetDev& n = h->net;
    GemmArgs fl[PPO_MAX_LAYERS];
    bool same_j = true;
    for (int l = 0; l < n.L; ++l) {
        GemmArgs a{};
        const int Kp = l ? n.Hp[l - 1] : n.Kp0;
        for (int t = 0; t < 2; ++t) {
            a.A[t] = l ? b.hb[t][l - 1] : (x0_override ? x0_override : b.x0); a.B[t] = b.theta_bf + n.w_off[t][l]; a.bias[t] = h->theta + n.b_off[t][l];
            a.C[t] = b.hb[t][l];
        }
        a.lda = Kp; a.ldb = n.Hp[l]; a.K = Kp; a.ldc = n.Hp[l];
        fl[l] = a;
        same_j = same_j && n.Hp[l] == n.Hp[0];
    }
    // every hidden layer in ONE launch when the shape qualifies, else a launch per layer.  (The heads riding behind the last layer of the chained launch --
    // a workgroup's own column tile is one 128-deep reduction range -- was measured: -4.2 us on the pass, +1.8 us in the loss kernel for twice the
    // partial products, and +16 us per env step on the act path, which has to cut its heads the same way to give the same bits; not kept.)
    if (!(same_j && n.L <= GB_CHAIN_MAX && bf16_chain<GEPI_TANH>(h, fl, n.L, Rp, n.Hp[0], 0)))
        for (int l = 0; l < n.L; ++l) if (bf16_gemm<GEPI_TANH>(h, fl[l], Rp, n.Hp[l])) return -1;
    // the heads: 64 rows of one tower and one reduction range per workgroup (bf16_heads_kernel)
    const int HpL = n.Hp[n.L - 1];
    if (n.Ap != GB_N) return fail(h, "bf16 path: the head kernel's images are 128 columns wide (Ap = %d)", n.Ap);
    HeadArgsB ha{};
    for (int t = 0; t < 2; ++t) {
        ha.H[t] = b.hb[t][n.L - 1]; ha.W[t] = b.theta_bf + (t ? n.wv_off : n.wmu_off); ha.bias[t] = h->theta + (t ? n.bv_off : n.bmu_off); ha.F[t] = b.head_out[t];
    }
    ha.ldh = HpL; ha.ldw = n.Ap; ha.ldf = n.Ap; ha.K = HpL;
    // the reduction in GB_HEAD_SPLIT ranges whatever the row count (the act model and the train model cut it alike: same bits per row)
    int ks = GB_HEAD_SPLIT;
    while (ks > 1 && (HpL / ks) % 32) ks /= 2;
    ha.ksplit = ks; ha.f_split = (size_t)b.Rcap * n.Ap; b.head_split = ks;
    const dim3 grid((Rp / BH_ROWS) * ks, 2);
    const int ncb = (n.A + 15) / 16;
    if (ncb <= 1) hipLaunchKernelGGL((bf16_heads_kernel<true, 1>), grid, dim3(256), BH_LDS_BYTES, h->stream, ha);
    else if (ncb <= 2) hipLaunchKernelGGL((bf16_heads_kernel<true, 2>), grid, dim3(256), BH_LDS_BYTES, h->stream, ha);
    else if (ncb <= 4) hipLaunchKernelGGL((bf16_heads_kernel<true, 4>), grid, dim3(256), BH_LDS_BYTES, h->stream, ha);
    else hipLaunchKernelGGL((bf16_heads_kernel<true, 8>), grid, dim3(256), BH_LDS_BYTES, h->stream, ha);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int bf16_stage(ppo_handle* h, const float* obs, int nrows, int Rp, ObsNorm nz, float* obs_out) {
    ppo_handle::Bf16& b = h->bf;
    const NetDev& n = h->net;
    StageArgsB sa{obs, nrows, n.O, n.Kp0, Rp, nz, obs_out, b.x0};
    const size_t cnt = (size_t)Rp * n.Kp0;
    if (n.O % 4 == 0 && n.Kp0 % 4 == 0) hipLaunchKernelGGL(bf16_stage4_kernel, dim3(bf16_stage4_grid(n.Kp0, cnt / 4)), dim3(256), 0, h->stream, sa);
    else hipLaunchKernelGGL(bf16_stage_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, sa);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int launch_step_bf16(ppo_handle* h, const StepArgs& a) {
    if (bf16_ensure_ws(h, a.n)) return -1;
    ProfScope ps(h, PK_STEP);
    const NetDev& n = h->net;
    const int Rp = ru(a.n, GB_PAD);
    if (bf16_stage(h, a.obs, a.n, Rp, a.nz, a.obs_out) || bf16_forward(h, Rp)) return -1;
    SampleArgsB sa{};
    sa.head[0] = h->bf.head_out[0]; sa.head[1] = h->bf.head_out[1]; sa.ldh = n.Ap; sa.hsplit = h->bf.head_split; sa.hstride = (size_t)h->bf.Rcap * n.Ap; sa.logstd = h->theta + n.ls_off;
    sa.noise = a.noise; sa.action = a.action; sa.det_action = a.det_action; sa.value = a.value; sa.neglogp = a.neglogp;
    sa.n = a.n; sa.A = n.A; sa.seed = a.seed; sa.rng_step = a.rng_step; sa.row_base = a.row_base;
    if (n.A > 128) return fail(h, "bf16 path: more than 128 actions");
    hipLaunchKernelGGL(bf16_sample_kernel, dim3((a.n + BS_ROWS - 1) / BS_ROWS), dim3(64 * BS_ROWS), 0, h->stream, sa);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// rows of the epoch-staged observations this minibatch starts at, or -1 when the step stages its own rows
long bf16_epoch_row(ppo_handle* h, const TrainArgs& ta, int Rp) {
    ppo_handle::Bf16& b = h->bf;
    if (!b.epoch_staged || !h->mb_obs || ta.obs < h->mb_obs || Rp != ta.n) return -1;
    const long r0 = (long)(ta.obs - h->mb_obs) / h->net.O;
    return (r0 + ta.n <= b.xe_rows) ? r0 : -1;
}

bool bf16_backward_links(ppo_handle* h, GemmArgs (&bl)[PPO_MAX_LAYERS], int (&outw)[PPO_MAX_LAYERS]);
// links_only: forward + heads + loss only (the bucketed data-parallel step runs the backward links itself)
int bf16_train_fwd_bwd(ppo_handle* h, const TrainArgs& ta, int Rp, bool links_only = false) {
    ppo_handle::Bf16& b = h->bf;
    const NetDev& n = h->net;
    const long er = bf16_epoch_row(h, ta, Rp);
    if (er < 0 && bf16_stage(h, ta.obs, ta.n, Rp, no_norm_fwd(), nullptr)) return -1;
    if (bf16_forward(h, Rp, er >= 0 ? b.xe + (size_t)er * n.Kp0 : nullptr)) return -1;
    LossArgsB la{};
    for (int t = 0; t < 2; ++t) { la.head[t] = b.head_out[t]; la.dhead[t] = b.dhead[t]; la.slots[t] = h->slots[t]; }
    la.ldh = n.Ap; la.hsplit = b.head_split; la.hstride = (size_t)b.Rcap * n.Ap; la.logstd = h->theta + n.ls_off; la.actions = ta.actions; la.advs = ta.advs; la.returns = ta.returns; la.old_values = ta.old_values;
    la.old_neglogp = ta.old_neglogp; la.hyper = h->hyper; la.n = ta.n; la.A = n.A; la.Ap = n.Ap; la.rows_pad = b.Rcap; la.inv_n = ta.inv_n;
    la.ent_coef = n.ent_coef; la.vf_coef = n.vf_coef; la.slot_w = n.slot_w; la.slot_head = n.slot_head; la.slot_aux = n.slot_aux; la.slot_loss = n.slot_loss;
    if (n.A > 64 * BL_EPT) return fail(h, "bf16 path: more than %d actions", 64 * BL_EPT);
    hipLaunchKernelGGL(bf16_loss_kernel, dim3(Rp / BL_ROWS), dim3(64 * BL_ROWS), (size_t)(2 * BL_ROWS * n.Ap + 6 * BL_ROWS) * sizeof(float), h->stream, la);
    HIP_OK(h, hipGetLastError());
    if (links_only) return 0;
    GemmArgs bl[PPO_MAX_LAYERS];
    int outw[PPO_MAX_LAYERS];
    const bool same_j = bf16_backward_links(h, bl, outw);
    const int HpL = n.Hp[n.L - 1];
    if (!(same_j && n.L <= GB_CHAIN_MAX && bf16_chain<GEPI_TANHGRAD>(h, bl, n.L, Rp, HpL, 1)))
        for (int k = 0; k < n.L; ++k) if (bf16_gemm<GEPI_TANHGRAD>(h, bl[k], Rp, outw[k])) return -1;
    return 0;
}

// the backward pass as L links: link 0 = dY_{L-1} = (d head * W_head^T) .* (1 - h_L^2), link k = down one hidden layer; returns whether all outputs have one width
bool bf16_backward_links(ppo_handle* h, GemmArgs (&bl)[PPO_MAX_LAYERS], int (&outw)[PPO_MAX_LAYERS]) {
    ppo_handle::Bf16& b = h->bf;
    const NetDev& n = h->net;
    const int HpL = n.Hp[n.L - 1];
    bool same_j = true;
    {
        GemmArgs a{};
        for (int t = 0; t < 2; ++t) {
            a.A[t] = b.dhead[t]; a.B[t] = b.theta_bf + (t ? n.wv_off : n.wmu_off); a.H[t] = b.hb[t][n.L - 1]; a.C[t] = b.dy[t][n.L - 1];
        }
        a.lda = n.Ap; a.ldb = n.Ap; a.K = n.Ap; a.ldh = HpL; a.ldc = HpL;
        for (int t = 0; t < 2; ++t) a.bsum[t] = b.dbias + b.db_off[t][n.L - 1];
        a.bsum_ld = b.n_dbias;
        bl[0] = a; outw[0] = HpL;
    }
    for (int l = n.L - 1; l >= 1; --l) {
        GemmArgs a{};
        for (int t = 0; t < 2; ++t) {
            a.A[t] = b.dy[t][l]; a.B[t] = b.theta_bf + n.w_off[t][l]; a.H[t] = b.hb[t][l - 1]; a.C[t] = b.dy[t][l - 1];
        }
        a.lda = n.Hp[l]; a.ldb = n.Hp[l]; a.K = n.Hp[l]; a.ldh = n.Hp[l - 1]; a.ldc = n.Hp[l - 1];
        for (int t = 0; t < 2; ++t) a.bsum[t] = b.dbias + b.db_off[t][l - 1];
        a.bsum_ld = b.n_dbias;
        bl[n.L - l] = a; outw[n.L - l] = n.Hp[l - 1];
        same_j = same_j && n.Hp[l - 1] == HpL;
    }
    return same_j;
}

// work split of the weight-gradient GEMM for `Rp` rows: stages per workgroup (see DwArgsB).  One round on the 256 CUs when there is
// that much work; never more contributors per tile than there are slabs.
void bf16_dw_split(ppo_handle* h, int Rp, int& nst, int& per, int& groups) {
    nst = Rp / GB_K;
    const int total = h->bf.n_dw_tiles * nst;
    per = std::max((total + 255) / 256, (nst + h->max_split - 3) / (h->max_split - 2));
    per = std::min(std::max(per, 1), nst);
    groups = (total + per - 1) / per;
}

// tile0 / ntiles: a contiguous range of the (layer-major) tile table walked as a launch of its own (-1: the whole table); per_out: the split the assembly needs
int bf16_weight_grads(ppo_handle* h, const TrainArgs& ta, int Rp, int tile0 = -1, int ntiles = 0, int* per_out = nullptr) {
    ppo_handle::Bf16& b = h->bf;
    const long er = bf16_epoch_row(h, ta, Rp);
    int nst, per, groups;
    bf16_dw_split(h, Rp, nst, per, groups);
    const DwTileB* tiles = b.dw_tiles; int nt = b.n_dw_tiles;
    if (tile0 >= 0) {
        tiles += tile0; nt = ntiles;
        const int total = nt * nst;
        per = std::max((total + 255) / 256, (nst + h->max_split - 3) / (h->max_split - 2));
        per = std::min(std::max(per, 1), nst);
        groups = (total + per - 1) / per;
    }
    if (per_out) *per_out = per;
    DwArgsB da{tiles, nst, per, nt * nst, h->slabs, (size_t)h->P_pad, er >= 0 ? b.xe + (size_t)er * h->net.Kp0 : nullptr, h->net.Kp0};
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    da.stamps = groups <= 512 ? g_stamps + 4096 * 16 : nullptr;
#endif
    if (b.dw_wm == 4) hipLaunchKernelGGL(gemm_dw_bf16_kernel<4>, dim3(groups), dim3(GB_THREADS(4)), GB_LDS_BYTES(4), h->stream, da);
    else hipLaunchKernelGGL(gemm_dw_bf16_kernel<2>, dim3(groups), dim3(GB_THREADS(2)), GB_LDS_BYTES(2), h->stream, da);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// ---- launches -------------------------------------------------------------------------------------------------
// narrow kernels: compile-time shapes <32, 64, 32, 2> / <64, 64, 32, 2> (observation tile of 32 / 64 columns) or the runtime-shape form;
// X(KP0, HP, AP, L) is the launch statement
#define NW_DISPATCH(h, X) do { if (!(h)->nw_static) { X(0, 0, 0, 0); } else if ((h)->net.Kp0 == 32) { X(32, 64, 32, 2); } else { X(64, 64, 32, 2); } } while (0)

template <int CT, int KS, int CTH, bool WIDE>
void launch_step_t(ppo_handle* h, const StepArgs& a0) {
    StepArgs a = a0;
    dim3 grid((a.n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, 2);
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    a.stamps = grid.x <= 256 ? g_stamps + 4096 * 44 : nullptr;
#endif
    hipLaunchKernelGGL((policy_step_kernel<CT, KS, CTH, WIDE>), grid, dim3(BLOCK_THREADS), (size_t)h->lds_step_total * sizeof(float), h->stream, h->net, a);
}
int launch_step(ppo_handle* h, const StepArgs& a) {
    if (h->bf.on) { ++h->kv[KV_BF16_STEP]; return launch_step_bf16(h, a); }
    ProfScope ps(h, PK_STEP);
    ++h->kv[h->narrow ? (h->nw_static ? KV_NARROW_STEP_STATIC : KV_NARROW_STEP) : KV_POLICY_STEP];
    if (h->narrow) {
        dim3 grid((a.n + NW_ROWS - 1) / NW_ROWS, 2);
        StepArgs sa = a;
        sa.theta = h->nw_img;                                  // the packed weight image stands in for the padded parameter vector
        const size_t lds = (size_t)h->nw.lds_total * sizeof(float);
#define X(a, b, c, d) hipLaunchKernelGGL((narrow_step_kernel<a, b, c, d>), grid, dim3(NW_THREADS), lds, h->stream, h->net, h->nw, sa)
        NW_DISPATCH(h, X);
#undef X
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    if (h->net.wide) { if (h->CT == 4) launch_step_t<4, 2, 0, true>(h, a); else launch_step_t<1, 1, 0, true>(h, a); }
    else if (h->CT == 4 && h->CTH == 2) launch_step_t<4, 2, 2, false>(h, a);
    else if (h->CT == 4) launch_step_t<4, 2, 0, false>(h, a);
    else launch_step_t<1, 1, 0, false>(h, a);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// ---- data-parallel collectives: one-shot peer exchange when attached and the payload fits, RCCL otherwise ------------------
// sum-all-reduce of buf[0, count) in place; sumsq != null: also the per-256-element sums of squares of the first
// sumsq_chunks chunks of the RESULT (what grad_sumsq_kernel would compute)
int enqueue_allreduce(ppo_handle* h, float* buf, size_t count, float* sumsq = nullptr, int sumsq_chunks = 0) {
    ProfScope ps(h, PK_COMM);
    if (h->peer.on && count <= h->peer.cap) {
        if (count > 32768) {
            const unsigned grid = (unsigned)((count + 4095) / 4096);
            hipLaunchKernelGGL(peer_push_kernel<4>, dim3(grid), dim3(PEER_THREADS), 0, h->stream, h->peer.dev, (const float*)buf, (unsigned long long)count);
            // (the sum wants parallelism, not few fences: 4096-float pieces took 9.9 us here)
            hipLaunchKernelGGL(peer_sum_kernel<1>, dim3((unsigned)((count + 1023) / 1024)), dim3(PEER_THREADS), 0, h->stream, h->peer.dev, buf, (unsigned long long)count, sumsq, (unsigned)sumsq_chunks);
        } else {
            const unsigned grid = (unsigned)((count + 1023) / 1024);
            hipLaunchKernelGGL(peer_push_kernel<1>, dim3(grid), dim3(PEER_THREADS), 0, h->stream, h->peer.dev, (const float*)buf, (unsigned long long)count);
            hipLaunchKernelGGL(peer_sum_kernel<1>, dim3(grid), dim3(PEER_THREADS), 0, h->stream, h->peer.dev, buf, (unsigned long long)count, sumsq, (unsigned)sumsq_chunks);
        }
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    const int rc = h->rccl.AllReduce(buf, buf, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, "ncclAllReduce failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(rc) : "?");
    if (sumsq) {
        hipLaunchKernelGGL(grad_sumsq_kernel, dim3(sumsq_chunks), dim3(256), 0, h->stream, buf, sumsq);
        HIP_OK(h, hipGetLastError());
    }
    return 0;
}

// the padded gradient + loss sums + row count of one minibatch, then the sums of squares of the reduced gradient
int enqueue_grad_allreduce(ppo_handle* h) { return enqueue_allreduce(h, h->grad, (size_t)h->P_pad + 8, h->sumsq, h->n_blocks); }

// after a stream synchronisation: did a peer wait time out since the last check?
int peer_check(ppo_handle* h) {
    if (!h->peer.on) return 0;
    unsigned e = 0;
    HIP_OK(h, hipMemcpy(&e, h->peer.dev.err, sizeof e, hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(h->peer.dev.err, 0, sizeof e);
        return fail(h, "peer all-reduce: rank %d gave up waiting for rank %u's flag (dead or diverged peer)", h->rank, e - 1);
    }
    return 0;
}

// after a stream synchronisation: did a meeting of narrow_epoch_kernel time out?
int nw_epoch_check(ppo_handle* h) {
    if (!h->nw_epoch || !h->nw_epoch_words) return 0;
    unsigned e = 0;
    HIP_OK(h, hipMemcpy(&e, h->nw_epoch_words + NW_EPOCH_WORDS - 1, sizeof e, hipMemcpyDeviceToHost));
    if (e == 2) {
        (void)hipMemset(h->nw_epoch_words, 0, NW_EPOCH_WORDS * sizeof e);
        h->nw_epoch_xl = false;
        drop_graph(h);
        return fail(h, "narrow_epoch_kernel: its workgroups were not on one XCD (the launch's workgroup dealing is not round-robin over the XCDs here); this update's results are "
                       "invalid.  The handle now exchanges the partial gradients write-through (PPO_HIP_NO_NARROW_EPOCH_XL=1 selects that from the start)");
    }
    if (e) {
        (void)hipMemset(h->nw_epoch_words, 0, NW_EPOCH_WORDS * sizeof e);
        h->nw_epoch = false;
        drop_graph(h);
        return fail(h, "narrow_epoch_kernel: its workgroups were not resident together (is another process using this GPU?); this update's results are invalid.  "
                       "The handle now launches every train step (PPO_HIP_NO_NARROW_EPOCH=1 selects that from the start)");
    }
    return 0;
}

// after a stream synchronisation: did adam_kernel<.., MEET>'s grid-wide meeting time out?
int adam_meet_check(ppo_handle* h) {
    if (!h->adam_meet_words || !h->comm) return 0;
    unsigned e = 0;
    HIP_OK(h, hipMemcpy(&e, h->adam_meet_words + ADAM_MEET_MAX_GRID, sizeof e, hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(h->adam_meet_words, 0, (ADAM_MEET_MAX_GRID + 32) * sizeof e);
        h->adam_meet = false;
        drop_graph(h);
        return fail(h, "adam_kernel: the workgroups of its data-parallel form were not resident together within ~0.5 s; this step's results are invalid.  "
                       "The handle now uses the launches that need no meeting (PPO_HIP_NO_ADAM_MEET=1 selects them from the start)");
    }
    return 0;
}

int pick_split(ppo_handle* h, int n) {
    int s = h->max_split;
    while (s > 1 && (n % (16 * s) != 0)) s >>= 1;
    return s;
}

// data parallel: may adam_kernel take the sums of squares (and, over peer regions, the ranks' sum) into its own launch?  Every workgroup of the launch
// has to be resident at once; the bf16 path's 20 k workgroups are not.
bool adam_can_meet(const ppo_handle* h) { return h->comm && h->adam_meet && (h->n_blocks + 3) / 4 <= ADAM_MEET_MAX_GRID && !h->bf.on; }

// clip + Adam on the assembled gradient (after the optional all-reduce)
// meet (data parallel, see adam_kernel): 2 = the ranks' tiles sit in the peer slots (weight_grad_assemble_kernel<.., PEER> pushed them).  (1 = `grad` is
// reduced but its sums of squares are not formed -- measured on the RCCL path against the grad_sumsq_kernel launch it would replace: 43.1 vs 42.2 us per
// train step at configs[2], the meeting costs more than the launch; not instantiated)
int enqueue_adam(ppo_handle* h, float* loss_row, int n_sumsq = 0, const float* parts_from = nullptr, int meet = 0) {      // n_sumsq: entries of h->sumsq / parts_from (0 = one per 256-element chunk)
    ProfScope ps(h, PK_ADAM);
    const float* parts = parts_from ? parts_from : h->sumsq; int n_parts = n_sumsq ? n_sumsq : h->n_blocks;
    if (!n_sumsq && h->n_blocks > 2048) {                  // very large nets: fold the per-chunk partials first
        n_parts = (h->n_blocks + 1023) / 1024;
        hipLaunchKernelGGL(sumsq_fold_kernel, dim3(n_parts), dim3(256), 0, h->stream, h->sumsq, h->n_blocks, h->sumsq2);
        HIP_OK(h, hipGetLastError());
        parts = h->sumsq2;
    }
    AdamArgs aa{h->theta, h->adam_m, h->adam_v, h->grad, h->sumsq, h->n_blocks, h->thetaT, h->par, h->grad_src, h->hyper, h->beta_pow,
                h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->cfg.max_grad_norm, loss_row, h->norm_out, parts, n_parts,
                0, {}, h->bf.on ? h->bf.theta_bf : nullptr, h->narrow ? h->nw_img : nullptr, nullptr, nullptr, nullptr};
    aa.n_tiled = h->n_tiled; for (int q = 0; q < h->n_tiled; ++q) aa.tiled[q] = h->tiled[q];
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    aa.stamps = g_stamps + 4096 * 40;
#endif
    if (h->nw_cur == 1) { aa.theta_in = h->nw_theta1; aa.m_in = h->nw_m1; aa.v_in = h->nw_v1; h->nw_cur = 0; }   // (always writes set 0)
    const dim3 grid((h->n_blocks + 3) / 4);
    if (meet) {
        aa.meet_words = h->adam_meet_words; aa.meet_parts = h->adam_meet_parts; aa.meet_grid = (int)grid.x;
        aa.peer = h->peer.dev;
        if (h->adam_fast) hipLaunchKernelGGL((adam_kernel<true, 2>), grid, dim3(256), 0, h->stream, aa);
        else hipLaunchKernelGGL((adam_kernel<false, 2>), grid, dim3(256), 0, h->stream, aa);
        HIP_OK(h, hipGetLastError());
        return 0;
    }
    // (the handle whose train kernels may apply Adam in their prologue uses the same 1-ulp quotient in its launches: bit-identical forms)
    if (h->adam_fast) hipLaunchKernelGGL(adam_kernel<true>, grid, dim3(256), 0, h->stream, aa);
    else hipLaunchKernelGGL(adam_kernel<false>, grid, dim3(256), 0, h->stream, aa);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// the [256,256] pair: one instantiation per (observation tile, action tile) width
template <int KP0, int AP>
void launch_train8(ppo_handle* h, dim3 grid, const TrainArgs& ta) {
    const size_t lds = sizeof(float) * T8L<KP0, AP>::TOTAL;
    hipLaunchKernelGGL((train8_kernel<KP0, AP>), grid, dim3(T8_THREADS), lds, h->stream, h->net, ta);
}
template <int KP0, int AP>
void launch_dw2(ppo_handle* h, const Dw2Args& da) {
    const size_t lds = sizeof(float) * Dw2L<KP0, AP>::LDS_FLOATS;
    hipLaunchKernelGGL((weight_grad_assemble_kernel<KP0, AP>), dim3(DW2_GRID), dim3(DW2_THREADS), lds, h->stream, da);
}
template <int KP0, int AP>
void launch_dw2_peer(ppo_handle* h, const Dw2Args& da) {
    const size_t lds = sizeof(float) * Dw2L<KP0, AP>::LDS_FLOATS;
    hipLaunchKernelGGL((weight_grad_assemble_peer_kernel<KP0, AP>), dim3(DW2_GRID), dim3(DW2_THREADS), lds, h->stream, da, h->peer.dev);
}
template <int KP0, int AP>
bool set_lds_pair() {
    return hipFuncSetAttribute((const void*)weight_grad_assemble_peer_kernel<KP0, AP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Dw2L<KP0, AP>::LDS_FLOATS) == hipSuccess &&
           hipFuncSetAttribute((const void*)train8_kernel<KP0, AP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * T8L<KP0, AP>::TOTAL) == hipSuccess &&
           hipFuncSetAttribute((const void*)weight_grad_assemble_kernel<KP0, AP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Dw2L<KP0, AP>::LDS_FLOATS) == hipSuccess;
}

// Data parallel over the collective library, bf16 path (BASELINE configs[4]: 19 MB of gradient per train step): the gradient leaves in BUCKETS, last layer + heads first.
// Backward link k (a launch of its own here; the single-rank step chains them) is followed by the weight-gradient GEMM of exactly the matrices it completed (a contiguous
// range of the layer-major tile table, work-balanced over the chip by itself) and their assembly, and that bucket's ncclAllReduce goes to a SECOND stream, where it runs
// under link k + 1 and its weight gradients; the streams join in front of the sums of squares + clip + Adam.  With one exposed all-reduce of the whole vector behind a
// ~230 us step, an 8-GPU job would spend ~40 % of its time in it; whether the overlap delivers cannot be measured on one device -- what CAN be
// measured is what the per-layer launches cost one rank (profiles/r06_f_*), and that the result is the oracle's over the union at world 2 and 8
// (tests/test_dp_two_ranks.py::test_two_ranks_bf16_path).  ppo_dist_bucketed(h, 0) keeps the single all-reduce.  No reference counterpart.
int bf16_train_bucketed(ppo_handle* h, const TrainArgs& ta, int Rp, float* loss_row) {
    ppo_handle::Bf16& b = h->bf;
    const NetDev& n = h->net;
    if (!b.comm_stream) {
        HIP_OK(h, hipStreamCreateWithFlags(&b.comm_stream, hipStreamNonBlocking));
        for (int k = 0; k <= PPO_MAX_LAYERS; ++k) HIP_OK(h, hipEventCreateWithFlags(&b.bk_ev[k], hipEventDisableTiming));
        HIP_OK(h, hipEventCreateWithFlags(&b.bk_join, hipEventDisableTiming));
    }
    { ProfScope ps(h, PK_TRAIN_FB); if (bf16_train_fwd_bwd(h, ta, Rp, /*links_only*/ true)) return -1; }
    GemmArgs bl[PPO_MAX_LAYERS];
    int outw[PPO_MAX_LAYERS];
    (void)bf16_backward_links(h, bl, outw);
    int nst, per_all, groups_all;
    bf16_dw_split(h, Rp, nst, per_all, groups_all);
    for (int k = 0; k < n.L; ++k) {
        const int l = n.L - 1 - k;                           // the layer whose pre-activation gradient this link completes
        { ProfScope ps(h, PK_TRAIN_FB); if (bf16_gemm<GEPI_TANHGRAD>(h, bl[k], Rp, outw[k])) return -1; }
        const int t0 = b.layer_tile0[l], t1 = k == 0 ? b.layer_tile0[n.L + 1] : b.layer_tile0[l + 1];
        int per = 0;
        { ProfScope ps(h, PK_DW); if (bf16_weight_grads(h, ta, Rp, t0, t1 - t0, &per)) return -1; }
        const int c0 = n.w_off[0][l] / 256, c1 = k == 0 ? h->n_blocks : n.w_off[0][l + 1] / 256;
        {
            ProfScope ps(h, PK_REDUCE);
            ReduceArgs ra{};
            ra.sk_nst = nst; ra.sk_per = per; ra.sk_bm = GB_BM(b.dw_wm); ra.sk_tile_base = t0; ra.chunk_lo = c0; ra.chunk_hi = c1;
            ra.src = h->grad_src; ra.n_blocks = h->n_blocks; ra.slabs = h->slabs; ra.slab_stride = (size_t)h->P_pad; ra.nsplit = 0;
            ra.slots[0] = h->slots[0]; ra.slots[1] = h->slots[1]; ra.n_rowblocks = Rp / BL_ROWS; ra.slot_w = n.slot_w; ra.slot_loss = n.slot_loss;
            ra.grad = h->grad; ra.sumsq = h->sumsq; ra.n_local = (float)ta.n; ra.beta_pow = h->beta_pow; ra.direct = b.dbias;
            ra.n_direct = Rp / (Rp % 256 == 0 ? 256 : 128); ra.direct_stride = b.n_dbias;
            const int chunks = c1 - c0 + (k == 0 ? 1 : 0);     // (+ the tail block: loss sums, row count, the powers' cur <- next, with the bucket at the vector's end)
            hipLaunchKernelGGL(bf16_grad_reduce_kernel, dim3((chunks + BGR_WAVES - 1) / BGR_WAVES), dim3(64 * BGR_WAVES), 0, h->stream, ra);
            HIP_OK(h, hipGetLastError());
        }
        {
            ProfScope ps(h, PK_COMM);
            HIP_OK(h, hipEventRecord(b.bk_ev[k], h->stream));
            HIP_OK(h, hipStreamWaitEvent(b.comm_stream, b.bk_ev[k], 0));
            float* buf = h->grad + (size_t)c0 * 256;
            const size_t count = (size_t)(c1 - c0) * 256 + (k == 0 ? 8 : 0);
            const int rc = h->rccl.AllReduce(buf, buf, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, b.comm_stream);
            if (rc != 0) return fail(h, "ncclAllReduce (bucket %d) failed: %s", k, h->rccl.GetErrorString ? h->rccl.GetErrorString(rc) : "?");
        }
    }
    HIP_OK(h, hipEventRecord(b.bk_join, b.comm_stream));
    HIP_OK(h, hipStreamWaitEvent(h->stream, b.bk_join, 0));
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(h->n_blocks), dim3(256), 0, h->stream, h->grad, h->sumsq);
    HIP_OK(h, hipGetLastError());
    return enqueue_adam(h, loss_row);
}

// the per-minibatch launch sequence: fwd+loss+bwd -> weight grads -> reduce [-> all-reduce] -> clip+Adam
// defer (narrow reference shape, inside ppo_update only): leave this step's clip + Adam to the next train kernel's prologue
// (flush_pending_adam after the last step)
int enqueue_train(ppo_handle* h, TrainArgs ta, float* loss_row, bool defer = false) {
    const NetDev& n = h->net;
    ta.theta = h->theta; ta.thetaT = h->thetaT; ta.par = h->par; ta.hyper = h->hyper;
    ta.x0g = h->x0g; ta.dmug = h->dmug;
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    ta.stamps = g_stamps;
#endif
    for (int t = 0; t < 2; ++t) {
        ta.slots[t] = h->slots[t];
        for (int l = 0; l < n.L; ++l) { ta.hg[t][l] = h->hg[t][l]; ta.dyg[t][l] = h->dyg[t][l]; }
    }
    if (h->narrow) {
        const int groups = (ta.n + NW_ROWS - 1) / NW_ROWS;
        {
            ProfScope ps(h, PK_TRAIN_FB);
            NwTrainArgs na{h->nw_img, ta.obs, ta.actions, ta.advs, ta.returns, ta.old_values, ta.old_neglogp, h->hyper, ta.n, ta.inv_n,
                           h->nw_partials, groups, h->nw_stride, nullptr};
#ifdef PPO_STAMPS
            if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
            na.stamps = g_stamps;
#endif
            const size_t lds = (size_t)h->nw.lds_total * sizeof(float);
            NwLazyArgs z{};
            ++h->kv[h->nw_static ? KV_NARROW_TRAIN_STATIC : KV_NARROW_TRAIN];
            if (h->nw_pending) {
                // the previous step's clip + Adam rides in this launch: read set nw_cur, write the other one
                float* set[2][3] = {{h->theta, h->adam_m, h->adam_v}, {h->nw_theta1, h->nw_m1, h->nw_v1}};
                const int ci = h->nw_cur, co = ci ^ 1;
                z = NwLazyArgs{h->grad, h->sumsq, h->nw_pending_parts, set[ci][0], set[ci][1], set[ci][2], set[co][0], set[co][1], set[co][2], h->beta_pow,
                               h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->cfg.max_grad_norm, h->nw_pending_loss, h->norm_out};
                if (h->adam_exact) {
                    if (n.Kp0 == 32) hipLaunchKernelGGL((narrow_train_kernel<32, 64, 32, 2, true, true>), dim3(groups, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, na, z);
                    else hipLaunchKernelGGL((narrow_train_kernel<64, 64, 32, 2, true, true>), dim3(groups, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, na, z);
                }
                else if (n.Kp0 == 32) hipLaunchKernelGGL((narrow_train_kernel<32, 64, 32, 2, true>), dim3(groups, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, na, z);
                else hipLaunchKernelGGL((narrow_train_kernel<64, 64, 32, 2, true>), dim3(groups, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, na, z);
                h->nw_cur = co; h->nw_pending = false;
            }
            else {
#define X(a, b, c, d) hipLaunchKernelGGL((narrow_train_kernel<a, b, c, d>), dim3(groups, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, na, z)
                NW_DISPATCH(h, X);
#undef X
            }
            HIP_OK(h, hipGetLastError());
        }
        const int n_chunks = h->P_pad / 64;
        {
            ProfScope ps(h, PK_REDUCE);
            NwReduceArgs ra{h->grad_src, n_chunks, h->nw_partials, groups, h->nw_stride, h->P_pad, h->grad, h->sumsq, (float)ta.n, h->beta_pow};
            hipLaunchKernelGGL(narrow_reduce_kernel, dim3(n_chunks + 1), dim3(256), 0, h->stream, ra);
            HIP_OK(h, hipGetLastError());
        }
        if (h->comm && enqueue_grad_allreduce(h)) return -1;
        const int n_parts = h->comm ? 0 : n_chunks;               // after an all-reduce: one partial per 256-element chunk (grad_sumsq_kernel's)
        if (defer && h->nw_lazy) { h->nw_pending = true; h->nw_pending_loss = loss_row; h->nw_pending_parts = n_parts ? n_parts : h->n_blocks; return 0; }
        return enqueue_adam(h, loss_row, n_parts);
    }
    if (h->bf.on) {
        const int Rp = ru(ta.n, GB_PAD);
        ++h->kv[KV_BF16_TRAIN];
        h->bf.lazy_last = false;
        if (h->comm && !h->peer.on && h->net.L >= 2 && (h->bf.bucketed_any_world || (h->world > 1 && h->bf.bucketed))) return bf16_train_bucketed(h, ta, Rp, loss_row);
        { ProfScope ps(h, PK_TRAIN_FB); if (bf16_train_fwd_bwd(h, ta, Rp)) return -1; }
        { ProfScope ps(h, PK_DW); if (bf16_weight_grads(h, ta, Rp)) return -1; }
        {
            ProfScope ps(h, PK_REDUCE);
            ReduceArgs ra{};
            int groups;
            bf16_dw_split(h, Rp, ra.sk_nst, ra.sk_per, groups); ra.sk_bm = GB_BM(h->bf.dw_wm);
            ra.src = h->grad_src; ra.n_blocks = h->n_blocks; ra.slabs = h->slabs; ra.slab_stride = (size_t)h->P_pad; ra.nsplit = 0;
            ra.slots[0] = h->slots[0]; ra.slots[1] = h->slots[1]; ra.n_rowblocks = Rp / BL_ROWS; ra.slot_w = n.slot_w; ra.slot_loss = n.slot_loss;
            ra.grad = h->grad; ra.sumsq = h->sumsq; ra.n_local = (float)ta.n; ra.beta_pow = h->beta_pow; ra.direct = h->bf.dbias;
            ra.n_direct = Rp / (Rp % 256 == 0 ? 256 : 128); ra.direct_stride = h->bf.n_dbias;
            const int n_old = (h->n_blocks + 1 + BGR_WAVES - 1) / BGR_WAVES;
            const char* e1 = getenv("PPO_HIP_NO_REDUCE_ADAM");                  // (read per call: a test compares the forms in one process; a graph keeps what it captured)
            if (!h->comm && h->bf.fuse_ra && !h->adam_fast && !(e1 && e1[0] == '1')) {
                // ... and clip + Adam in the same persistent launch (ppo_bf16.hpp, bf16_reduce_adam_kernel)
                h->bf.lazy_ra = ra; h->bf.lazy_n_old = n_old; h->bf.lazy_last = true;
                ReduceAdamArgs fa{ra, h->theta, h->adam_m, h->adam_v, h->bf.theta_bf, h->hyper, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->cfg.max_grad_norm,
                                  loss_row, h->norm_out, h->bf.ra_ent, n_old, /*store_grad*/ 0};
                const int rounds = (n_old + BRA_GRID - 1) / BRA_GRID;
                const dim3 g(BRA_GRID), blk(64 * BGR_WAVES);
                ++h->kv[KV_BF16_REDUCE_ADAM];
                if (rounds <= 1) hipLaunchKernelGGL(bf16_reduce_adam_kernel<1>, g, blk, 0, h->stream, fa);
                else if (rounds <= 2) hipLaunchKernelGGL(bf16_reduce_adam_kernel<2>, g, blk, 0, h->stream, fa);
                else if (rounds <= 3) hipLaunchKernelGGL(bf16_reduce_adam_kernel<3>, g, blk, 0, h->stream, fa);
                else if (rounds <= 5) hipLaunchKernelGGL(bf16_reduce_adam_kernel<5>, g, blk, 0, h->stream, fa);
                else hipLaunchKernelGGL(bf16_reduce_adam_kernel<8>, g, blk, 0, h->stream, fa);
                HIP_OK(h, hipGetLastError());
                return 0;
            }
            hipLaunchKernelGGL(bf16_grad_reduce_kernel, dim3(n_old), dim3(64 * BGR_WAVES), 0, h->stream, ra);
            HIP_OK(h, hipGetLastError());
        }
        if (h->comm) return enqueue_grad_allreduce(h) ? -1 : enqueue_adam(h, loss_row);      // (the exchange recomputes the per-chunk sums of squares)
        return enqueue_adam(h, loss_row, (h->n_blocks + 1 + BGR_WAVES - 1) / BGR_WAVES);       // one partial per assembly workgroup (adam_kernel keeps the bf16 copy of the weights current)
    }
    // weight_grad_assemble_kernel walks the minibatch in 64-row chunks.  train8_kernel writes zeros for every row >= n of every tile
    // it is launched on, so its grid is simply padded to whole chunks (ANY row count stays on the fast pair); behind the round-2
    // train kernel the pair needs a minibatch that is a whole number of chunks by itself.
    const bool use_dw2 = h->dw2 && (h->t8 || ta.n % DW2_CH == 0) && ru(ta.n, DW2_CH) / ROWS_PER_BLOCK <= 32 * DW2_SLOTK;
    const int n_pad = use_dw2 ? ru(ta.n, DW2_CH) : ru(ta.n, ROWS_PER_BLOCK);            // the train kernel zero-fills the rows of its last partial tile
    const int n_rb = n_pad / ROWS_PER_BLOCK;
    ta.xcd_map = use_dw2 ? 1 : 0;
    Dw2Args da{};
    if (use_dw2) {
        da.x0g = h->x0g; da.h2pi = h->hg[0][1]; da.dmug = h->dmug;
        for (int t = 0; t < 2; ++t) { da.h1[t] = h->hg[t][0]; da.dy0[t] = h->dyg[t][0]; da.dy1[t] = h->dyg[t][1]; da.w0_off[t] = n.w_off[t][0]; da.w1_off[t] = n.w_off[t][1]; da.slots[t] = h->slots[t]; }
        da.wmu_off = n.wmu_off; da.n = n_pad; da.slabs = h->slabs; da.slab_stride = (unsigned long long)h->P_pad; da.counters = h->dw2_counters;
        da.grad = h->grad; da.parts = h->dw2_parts; da.jobs = h->dw2_jobs; da.n_jobs = h->dw2_n_jobs; da.jobs_per_wg = h->dw2_jpw;
#ifdef PPO_STAMPS
        da.stamps = g_stamps + 4096 * 16;
#endif
        da.n_rowblocks = n_rb; da.slot_w = n.slot_w; da.n_local = (float)ta.n; da.beta_pow = h->beta_pow; da.tail_off = h->P_pad;
    }
    {
        ProfScope ps(h, PK_TRAIN_FB);
        dim3 grid(n_rb, 2);
        const size_t lds_bytes = (size_t)n.lds_total * sizeof(float);
        const dim3 blk(BLOCK_THREADS);
        ++h->kv[(h->t8 && !n.wide) ? KV_TRAIN8 : KV_TRAIN_FB];
        if (n.wide) {
            if (h->CT == 4) hipLaunchKernelGGL((train_fwd_bwd_kernel<4, 2, 0, true>), grid, blk, lds_bytes, h->stream, n, ta);
            else hipLaunchKernelGGL((train_fwd_bwd_kernel<1, 1, 0, true>), grid, blk, lds_bytes, h->stream, n, ta);
        }
        else if (h->t8) {
            if (n.Kp0 == 32 && n.Ap == 32) launch_train8<32, 32>(h, grid, ta);
            else if (n.Kp0 == 64 && n.Ap == 32) launch_train8<64, 32>(h, grid, ta);
            else if (n.Kp0 == 32 && n.Ap == 64) launch_train8<32, 64>(h, grid, ta);
            else launch_train8<64, 64>(h, grid, ta);
        }
        else if (h->CT == 4 && h->CTH == 2 && h->early) hipLaunchKernelGGL((train_fwd_bwd_kernel<4, 2, 2, false, true>), grid, blk, lds_bytes, h->stream, n, ta);
        else if (h->CT == 4 && h->CTH == 2) hipLaunchKernelGGL((train_fwd_bwd_kernel<4, 2, 2, false>), grid, blk, lds_bytes, h->stream, n, ta);
        else if (h->CT == 4) hipLaunchKernelGGL((train_fwd_bwd_kernel<4, 2, 0, false>), grid, blk, lds_bytes, h->stream, n, ta);
        else hipLaunchKernelGGL((train_fwd_bwd_kernel<1, 1, 0, false>), grid, blk, lds_bytes, h->stream, n, ta);
        HIP_OK(h, hipGetLastError());
    }
    // data parallel over peer regions: the tiles' finishers push to the peers themselves and adam_kernel adds the ranks up (no push / sum launches)
    const char* npt = getenv("PPO_HIP_NO_PEER_TILES");           // (read per call: a test compares the two forms in one process; a graph keeps what it captured)
    const bool no_peer_tiles = npt && npt[0] == '1';
    if (use_dw2 && h->comm && h->peer.on && !no_peer_tiles && adam_can_meet(h) && (size_t)h->P_pad + 8 <= h->peer.cap) {
        {
            ProfScope ps(h, PK_DW);
            ++h->kv[KV_DW2];
            if (n.Kp0 == 32 && n.Ap == 32) launch_dw2_peer<32, 32>(h, da);
            else if (n.Kp0 == 64 && n.Ap == 32) launch_dw2_peer<64, 32>(h, da);
            else if (n.Kp0 == 32 && n.Ap == 64) launch_dw2_peer<32, 64>(h, da);
            else launch_dw2_peer<64, 64>(h, da);
            HIP_OK(h, hipGetLastError());
        }
        return enqueue_adam(h, loss_row, 0, nullptr, 2);
    }
    if (use_dw2) {
        // weight gradients + slab / slot sums + partial sums of squares in ONE launch (ppo_dw2.hpp): no grad_reduce_kernel
        {
            ProfScope ps(h, PK_DW);
            ++h->kv[KV_DW2];
            if (n.Kp0 == 32 && n.Ap == 32) launch_dw2<32, 32>(h, da);
            else if (n.Kp0 == 64 && n.Ap == 32) launch_dw2<64, 32>(h, da);
            else if (n.Kp0 == 32 && n.Ap == 64) launch_dw2<32, 64>(h, da);
            else launch_dw2<64, 64>(h, da);
            HIP_OK(h, hipGetLastError());
        }
        if (h->comm) { if (enqueue_grad_allreduce(h)) return -1; return enqueue_adam(h, loss_row); }
        return enqueue_adam(h, loss_row, DW2_TILES + DW2_GRID, h->dw2_parts);
    }
    const int split = pick_split(h, n_pad);
    {
        ProfScope ps(h, PK_DW);
        DwArgs da{h->dw_tiles, n_pad, split, h->slabs, (size_t)h->P_pad, nullptr};
#ifdef PPO_STAMPS
        da.stamps = g_stamps + 4096 * 16;
#endif
        const size_t lds = (h->dw_has_big ? 4 * (64 * 64 + 1024) : 4 * 32 * 32) * sizeof(float);     // 4 waves x (tile + strips)
        const int rows_per_wave = n_pad / split / 4;
        ++h->kv[KV_DW]; ++h->kv[KV_GRAD_REDUCE];
        if (rows_per_wave % 16 == 0) hipLaunchKernelGGL(weight_grad_kernel<4>, dim3(h->n_dw_tiles * split), dim3(BLOCK_THREADS), lds, h->stream, da);
        else hipLaunchKernelGGL(weight_grad_kernel<1>, dim3(h->n_dw_tiles * split), dim3(BLOCK_THREADS), lds, h->stream, da);
        HIP_OK(h, hipGetLastError());
    }
    {
        ProfScope ps(h, PK_REDUCE);
        ReduceArgs ra{};
        ra.src = h->grad_src; ra.n_blocks = h->n_blocks; ra.slabs = h->slabs; ra.slab_stride = (size_t)h->P_pad; ra.nsplit = split;
        ra.slots[0] = h->slots[0]; ra.slots[1] = h->slots[1]; ra.n_rowblocks = n_rb; ra.slot_w = n.slot_w; ra.slot_loss = n.slot_loss;
        ra.grad = h->grad; ra.sumsq = h->sumsq; ra.n_local = (float)ta.n; ra.beta_pow = h->beta_pow;
        hipLaunchKernelGGL(grad_reduce_kernel, dim3(h->n_blocks + 1), dim3(256), 0, h->stream, ra);
        HIP_OK(h, hipGetLastError());
    }
    if (h->comm && enqueue_grad_allreduce(h)) return -1;         // also with a 1-rank communicator: same code path as N ranks
    return enqueue_adam(h, loss_row);
}

// the clip + Adam of the last deferred step, as a launch of its own (also brings the weights home to set 0 and refreshes the
// packed image / mirrors that only adam_kernel writes)
int flush_pending_adam(ppo_handle* h) {
    if (!h->nw_pending) return 0;
    h->nw_pending = false;
    return enqueue_adam(h, h->nw_pending_loss, h->nw_pending_parts == h->n_blocks ? 0 : h->nw_pending_parts);
}

// host stores into device memory through the BAR: order them (and push them out of the write-combining buffers) before the word that publishes them
static inline void host_store_fence() {
#if defined(__x86_64__)
    _mm_sfence();
#else
    std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}
// May the HOST store into [p, p + bytes)?  The range must lie inside ONE mapping of this process with read AND write permission (/proc/self/maps).  (msync() is not
// a probe for this: it succeeds on the runtime's reserved PROT_NONE aperture as well, and the first host store into such a page is a SIGSEGV.)
static bool host_can_store(const void* p, size_t bytes) {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return false;
    const unsigned long long lo = (unsigned long long)(uintptr_t)p, hi = lo + bytes;
    char line[512];
    bool ok = false;
    while (fgets(line, sizeof line, f)) {
        unsigned long long a = 0, b = 0; char perm[8] = {0};
        if (sscanf(line, "%llx-%llx %7s", &a, &b, perm) != 3) continue;
        if (a <= lo && hi <= b) { ok = perm[0] == 'r' && perm[1] == 'w'; break; }
    }
    fclose(f);
    return ok;
}

int set_hyper(ppo_handle* h, float lr, float cr) {
    // The source of an ASYNCHRONOUS copy must outlive this function: the runtime may read it when the copy executes, not when it is enqueued.  (Rounds 1 - 5 passed a
    // stack array here; at the end of a long process the second update of a handle then trained with whatever lay on the stack -- found by
    // tests/test_other_shapes.py::test_two_handles_interleaved_equal_the_same_handles_run_alone, which only failed behind the rest of the suite.)  The two floats
    // live in pinned memory owned by the handle; every API call that sets them synchronises the stream before it returns, so they are never rewritten under a copy.
    if (!h->hyper_host) HIP_OK(h, hipHostMalloc((void**)&h->hyper_host, 64, hipHostMallocDefault));
    h->hyper_host[0] = lr; h->hyper_host[1] = cr;
    HIP_OK(h, hipMemcpyAsync(h->hyper, h->hyper_host, 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    return 0;
}

// dense <-> padded copies of one tensor
int copy_tensor(ppo_handle* h, float* base, const Tensor& t, float* host, bool to_device) {
    float* d = base + t.off_pad;
    const size_t w = (size_t)t.dcols() * sizeof(float);
    const int rows = t.cols ? t.rows : 1;
    const size_t width = t.cols ? w : (size_t)t.rows * sizeof(float);
    if (to_device) HIP_OK(h, hipMemcpy2D(d, (size_t)t.pcol * sizeof(float), host, width, width, rows, hipMemcpyHostToDevice));
    else HIP_OK(h, hipMemcpy2D(host, width, d, (size_t)t.pcol * sizeof(float), width, rows, hipMemcpyDeviceToHost));
    return 0;
}

float* which_buf(ppo_handle* h, int which) { return which == 0 ? h->theta : which == 1 ? h->adam_m : which == 2 ? h->adam_v : nullptr; }

ObsNorm no_norm() { return ObsNorm{nullptr, nullptr, 0.f, 0.f, 0}; }

// GAE scan: one lane per env (coalesced across envs); few envs x long rollouts take the LDS form (bit-identical)
void launch_gae(ppo_handle* h, const float* rew, const float* val, const float* done, const float* last_val, const float* last_done, int T, int E,
                float gamma, float lam, float* ret) {
    if (E <= 64 && T >= 128 && (size_t)3 * T * sizeof(float) <= 60 * 1024)
        hipLaunchKernelGGL(gae_long_kernel, dim3(E), dim3(GAE_LONG_THREADS), (size_t)3 * T * sizeof(float), h->stream, rew, val, done, last_val, last_done, T, E, gamma, lam, ret);
    else hipLaunchKernelGGL(gae_kernel, dim3((E + 255) / 256), dim3(256), 0, h->stream, rew, val, done, last_val, last_done, T, E, gamma, lam, ret);
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
// (defined with the host-Env rollout forms below) retire a resident rollout kernel and book a posted transition, so that the
// caller sees -- and changes -- the state of the step-by-step path (include/ppo_hip.h: entry points called mid-rollout)
static int host_quiesce(ppo_handle* h);
// entry points that synchronise the stream, read the rollout / normaliser, or change weights, seed or buffers
#define ENTER_Q(h) do { ENTER(h); if (host_quiesce(h)) return -1; } while (0)

extern "C" {

int ppo_abi_version(void) { return PPO_ABI_VERSION; }

void ppo_config_default(ppo_config* cfg, int32_t obs_dim, int32_t act_dim, int32_t n_hidden, const int32_t* hidden) {
    memset(cfg, 0, sizeof *cfg);
    cfg->obs_dim = obs_dim; cfg->act_dim = act_dim; cfg->n_hidden = n_hidden;
    for (int i = 0; i < n_hidden && i < PPO_MAX_LAYERS; ++i) cfg->hidden[i] = hidden[i];
    cfg->ent_coef = 0.0007160293171182275f;   // G:11323
    cfg->vf_coef = 0.5f;                      // G:11395
    cfg->max_grad_norm = 0.5f;                // G:24370
    cfg->adam_beta1 = 0.9f; cfg->adam_beta2 = 0.999f; cfg->adam_eps = 1e-5f;   // G:30430-30490
    cfg->device = -1; cfg->max_rows = 0;
}

const char* ppo_last_error(const ppo_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ppo_create(const ppo_config* cfg, ppo_handle** out) {
    if (!cfg || !out) return fail(nullptr, "ppo_create: null argument");
    *out = nullptr;
    if (cfg->n_hidden < 1 || cfg->n_hidden > PPO_MAX_LAYERS) return fail(nullptr, "ppo_create: n_hidden must be 1..%d", PPO_MAX_LAYERS);
    if (cfg->obs_dim < 1 || cfg->act_dim < 1) return fail(nullptr, "ppo_create: bad obs/act dims");
    for (int l = 0; l < cfg->n_hidden; ++l) if (cfg->hidden[l] < 1) return fail(nullptr, "ppo_create: bad hidden size");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, "ppo_create: no HIP device available (%s); libppo_hip has no CPU fallback", hipGetErrorString(e));
    ppo_handle* h = new ppo_handle();
    h->cfg = *cfg;
    int dev = cfg->device;
    if (dev < 0) { const char* lr = getenv("LOCAL_RANK"); dev = lr ? atoi(lr) % ndev : 0; }
    h->device = dev;
    auto bail = [&](int) { g_create_error = h->err; ppo_destroy(h); return -1; };
    if (hipSetDevice(dev) != hipSuccess) { fail(h, "hipSetDevice(%d) failed", dev); return bail(0); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { fail(h, "hipGetDeviceProperties failed"); return bail(0); }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fail(h, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName); return bail(0); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { fail(h, "hipStreamCreate failed"); return bail(0); }
    const char* ng = getenv("PPO_HIP_NO_GRAPH");
    h->use_graph = !(ng && ng[0] == '1');
    if (cfg->compute_dtype != PPO_F32 && cfg->compute_dtype != PPO_BF16) { fail(h, "ppo_create: compute_dtype must be PPO_F32 or PPO_BF16"); return bail(0); }
    h->bf.on = cfg->compute_dtype == PPO_BF16;
    if (build_layout(h)) return bail(0);
    // large dynamic LDS needs an explicit opt-in.  The attribute is per function, not per handle: it is set to the
    // hardware maximum (160 KB) so that a later, narrower handle cannot lower the limit under a live wider one.
    const int lds_bytes = 160 * 1024;
    bool attr_ok = true;
    auto set_lds = [&](const void* f) { attr_ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess; };
    set_lds((const void*)policy_step_kernel<4, 2, 2, false>); set_lds((const void*)train_fwd_bwd_kernel<4, 2, 2, false>);
    set_lds((const void*)train_fwd_bwd_kernel<4, 2, 2, false, true>);
    { const NetDev& nn = h->net;
      h->early = !nn.wide && h->CT == 4 && h->CTH == 2 && nn.L >= 2 && nn.Kp0 == 32 && nn.Ap == 32 && nn.Hp[0] == 256 && nn.Hp[nn.L - 1] == 256; }
    set_lds((const void*)policy_step_kernel<4, 2, 0, false>); set_lds((const void*)train_fwd_bwd_kernel<4, 2, 0, false>);
    set_lds((const void*)policy_step_kernel<1, 1, 0, false>); set_lds((const void*)train_fwd_bwd_kernel<1, 1, 0, false>);
    set_lds((const void*)policy_step_kernel<4, 2, 0, true>); set_lds((const void*)train_fwd_bwd_kernel<4, 2, 0, true>);
    set_lds((const void*)policy_step_kernel<1, 1, 0, true>); set_lds((const void*)train_fwd_bwd_kernel<1, 1, 0, true>);
    attr_ok &= hipFuncSetAttribute((const void*)weight_grad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (64 * 64 + 1024) * 4) == hipSuccess;
    attr_ok &= hipFuncSetAttribute((const void*)weight_grad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (64 * 64 + 1024) * 4) == hipSuccess;
    if (!attr_ok) { fail(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); return bail(0); }
    const size_t P = (size_t)h->P_pad;
    if (dev_alloc(h, &h->par, (size_t)2 * h->net.par_total) || dev_alloc(h, &h->thetaT, (size_t)h->PT) || dev_alloc(h, &h->theta, P) || dev_alloc(h, &h->adam_m, P) || dev_alloc(h, &h->adam_v, P) || dev_alloc(h, &h->grad, P + 256) ||
        dev_alloc(h, &h->sumsq, (size_t)4 * h->n_blocks) || dev_alloc(h, &h->sumsq2, (size_t)(h->n_blocks + 1023) / 1024) || dev_alloc(h, &h->beta_pow, 4) || dev_alloc(h, &h->hyper, 2) ||
        dev_alloc(h, &h->norm_out, 1) || dev_alloc(h, &h->st_loss, 8))
        return bail(0);
    build_narrow_layout(h);
    if (upload_grad_src(h)) return bail(0);
    { const char* e = getenv("PPO_HIP_NO_DW2"); const NetDev& nn = h->net;
      h->dw2 = !(e && e[0] == '1') && !nn.wide && h->CT == 4 && nn.L == 2 && nn.Hp[0] == 256 && nn.Hp[1] == 256 && nn.Kp0 <= 64 && nn.Ap <= 64; }
    { const char* e = getenv("PPO_HIP_NO_T8"); const NetDev& nn = h->net;
      // any observation / action width up to 64 (tiles of 32 or 64 columns) in front of hidden [256,256]
      h->t8 = !(e && e[0] == '1') && !nn.wide && h->CT == 4 && nn.L == 2 && nn.Hp[0] == 256 && nn.Hp[1] == 256 && nn.Kp0 <= 64 && nn.Ap <= 64;
      if ((h->t8 || h->dw2) && !(set_lds_pair<32, 32>() && set_lds_pair<64, 32>() && set_lds_pair<32, 64>() && set_lds_pair<64, 64>())) {
          fail(h, "hipFuncSetAttribute failed for train8_kernel / weight_grad_assemble_kernel"); return bail(0); } }
    if (h->dw2) {
        // slot jobs: every element the train kernel leaves as per-row-block partial sums (bias / logstd / value-head gradients), then the loss sums
        const NetDev& nn = h->net;
        std::vector<SlotJob> jobs;
        for (const Tensor& t : h->tensors) {
            const std::string nm = t.name;
            const int tower = nm[0] == 'v' ? 1 : 0;
            int l = -1;
            if (nm.find("_fc") != std::string::npos) l = atoi(nm.c_str() + 5);
            int so = -1, cnt = 0, po = 0;                 // po: the tensor's place in the small-parameter mirror (upload_grad_src's p_off)
            if (l >= 0 && nm.substr(nm.size() - 2) == "/b") { so = nn.slot_db[l]; cnt = nn.Hp[l]; po = tower * nn.par_total + nn.par_b[l]; }
            else if (nm == "vf/w") { so = nn.slot_head; cnt = nn.Hp[nn.L - 1]; po = nn.par_total + nn.par_wv; }
            else if (nm == "vf/b") { so = nn.slot_aux; cnt = 1; po = nn.par_total + nn.par_bv; }
            else if (nm == "pi/b") { so = nn.slot_head; cnt = nn.Ap; po = nn.par_bmu; }
            else if (nm == "pi/logstd") { so = nn.slot_aux; cnt = nn.Ap; po = nn.par_ls; }
            for (int e2 = 0; e2 < cnt; ++e2) jobs.push_back(SlotJob{tower, so + e2, t.off_pad + e2, 1 + po + e2});
        }
        for (int q = 0; q < 5; ++q) jobs.push_back(SlotJob{q == 1 ? 1 : 0, nn.slot_loss + (q <= 1 ? 0 : q - 1), h->P_pad + q, 0});   // {pg, vf, ent, kl, cf} sums
        h->dw2_n_jobs = (int)jobs.size();
        h->dw2_jpw = (h->dw2_n_jobs + DW2_GRID - 1) / DW2_GRID;
        if (h->dw2_jpw > DW2_THREADS / 32) h->dw2 = false;
        else {
            if (dev_alloc(h, &h->dw2_jobs, jobs.size()) || dev_alloc(h, &h->dw2_counters, (size_t)DW2_TILES) || dev_alloc(h, &h->dw2_parts, (size_t)DW2_TILES + DW2_GRID)) return bail(0);
            HIP_OK(h, hipMemcpyAsync(h->dw2_jobs, jobs.data(), jobs.size() * sizeof(SlotJob), hipMemcpyHostToDevice, h->stream));
            HIP_OK(h, hipStreamSynchronize(h->stream));
        }
    }
    if (h->bf.on && bf16_create(h)) return bail(0);
    if (h->narrow) {
        attr_ok = true;
        auto big_lds = [&](const void* f) { attr_ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
#define X(a, b, c, d) do { big_lds((const void*)narrow_train_kernel<a, b, c, d>); big_lds((const void*)narrow_step_kernel<a, b, c, d>); big_lds((const void*)narrow_collect_kernel<a, b, c, d>); \
                           big_lds((const void*)narrow_rollout_kernel<a, b, c, d, false>); big_lds((const void*)narrow_rollout_kernel<a, b, c, d, true>); \
                           big_lds((const void*)narrow_rollout_coop_kernel<a, b, c, d>); big_lds((const void*)narrow_host_step_kernel<a, b, c, d>); } while (0)
        X(0, 0, 0, 0); X(32, 64, 32, 2); X(64, 64, 32, 2);
#undef X
        big_lds((const void*)narrow_train_kernel<32, 64, 32, 2, true>); big_lds((const void*)narrow_train_kernel<64, 64, 32, 2, true>);
        big_lds((const void*)narrow_train_kernel<32, 64, 32, 2, true, true>); big_lds((const void*)narrow_train_kernel<64, 64, 32, 2, true, true>);
        big_lds((const void*)narrow_epoch_kernel<32, false>); big_lds((const void*)narrow_epoch_kernel<64, false>);
        big_lds((const void*)narrow_epoch_kernel<32, true>); big_lds((const void*)narrow_epoch_kernel<64, true>);
        big_lds((const void*)narrow_epoch_kernel<32, false, true>); big_lds((const void*)narrow_epoch_kernel<64, false, true>);
        big_lds((const void*)narrow_epoch_kernel<32, true, true>); big_lds((const void*)narrow_epoch_kernel<64, true, true>);
        if (!attr_ok) { fail(h, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the narrow kernels"); return bail(0); }
        { const char* e1 = getenv("PPO_HIP_NO_HOST_FUSED"); const char* e2 = getenv("PPO_HIP_NO_HOST_RESIDENT");
          h->opt_no_host_fused = e1 && e1[0] == '1'; h->opt_no_host_resident = e2 && e2[0] == '1'; }
        const char* nl = getenv("PPO_HIP_NO_LAZY_ADAM");
        if (h->nw_static && !(nl && nl[0] == '1')) {       // (the static shapes: [64,64] behind a 32- or 64-column observation tile -- 18 / 36 observations -- and 32 action columns)
            // second parameter / moment set of the deferred Adam (zero-filled: the padding elements are never written and must read 0)
            if (dev_alloc(h, &h->nw_theta1, P) || dev_alloc(h, &h->nw_m1, P) || dev_alloc(h, &h->nw_v1, P)) return bail(0);
            h->nw_lazy = true;
            // the deferred / resident forms compute TF's quotient with the correctly rounded square root and division by DEFAULT since round 6 (no deviation from the
            // reference's arithmetic; 6 - 9 % of this shape's train step); PPO_HIP_ADAM_FAST=1 (below) opts into the hardware's 1-ulp reciprocal / square root
            h->adam_exact = true; h->adam_fast = false;
            const char* ne = getenv("PPO_HIP_NO_NARROW_EPOCH");
            h->nw_epoch = !(ne && ne[0] == '1') && prop.multiProcessorCount >= 2 * 2 * NW_EPOCH_MAX_G;
            if (h->nw_epoch && dev_alloc(h, &h->nw_epoch_words, NW_EPOCH_WORDS)) return bail(0);
            { const char* nx = getenv("PPO_HIP_NO_NARROW_EPOCH_XL"); h->nw_epoch_xl = h->nw_epoch && !(nx && nx[0] == '1') && prop.multiProcessorCount >= 64; }
        }
        if (dev_alloc(h, &h->nw_img, (size_t)2 * h->nw.w_total)) return bail(0);
    }
    { const char* af = getenv("PPO_HIP_ADAM_FAST"); if (af && af[0] == '1') { h->adam_fast = true; h->adam_exact = false; } }      // every Adam step of the handle, so that its forms agree bit for bit
    const float pw[2] = {cfg->adam_beta1, cfg->adam_beta2};
    if (ppo_set_beta_powers(h, pw)) return bail(0);
    *out = h;
    return 0;
}

void ppo_destroy(ppo_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->hp_active && h->pin_flag) __atomic_store_n(h->pin_flag + 64 + PCTL_STOP, 1u, __ATOMIC_RELEASE);    // a resident rollout kernel: ask it to leave
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    if (h->comm && h->rccl.CommDestroy) h->rccl.CommDestroy(h->comm);
    for (void* m : h->peer.mapped) if (m) (void)hipIpcCloseMemHandle(m);
    if (h->peer.region) (void)hipFree(h->peer.region);
    if (h->peer.local) (void)hipFree(h->peer.local);
    for (float* p : {h->nw_theta1, h->nw_m1, h->nw_v1}) if (p) (void)hipFree(p);
    if (h->nw_partials) (void)hipFree(h->nw_partials);
    if (h->nw_epoch_words) (void)hipFree(h->nw_epoch_words);
    if (h->nw_epoch_partials) (void)hipFree(h->nw_epoch_partials);
    if (h->nw_img) (void)hipFree(h->nw_img);
    if (h->nw_alt) (void)hipFree(h->nw_alt);
    if (h->nw_coop) (void)hipFree(h->nw_coop);
    if (h->nw_alt_counts) (void)hipFree(h->nw_alt_counts);
    for (float* p : {h->gs_obs, h->gs_act, h->gs_ret, h->gs_val, h->gs_nlp}) if (p) (void)hipFree(p);
    if (h->dw2_jobs) (void)hipFree(h->dw2_jobs);
    if (h->dw2_counters) (void)hipFree(h->dw2_counters);
    for (int d = 0; d < 2; ++d) if (h->bf.chain_words[d]) (void)hipFree(h->bf.chain_words[d]);
    if (h->bf.chain_err_host) (void)hipHostFree(h->bf.chain_err_host);
    if (h->bf.comm_stream) { (void)hipStreamDestroy(h->bf.comm_stream); for (auto& e : h->bf.bk_ev) if (e) (void)hipEventDestroy(e); if (h->bf.bk_join) (void)hipEventDestroy(h->bf.bk_join); }
    if (h->bf.ra_ent) (void)hipFree(h->bf.ra_ent);
    if (h->adam_meet_words) (void)hipFree(h->adam_meet_words);
    if (h->adam_meet_parts) (void)hipFree(h->adam_meet_parts);
    if (h->dw2_parts) (void)hipFree(h->dw2_parts);
    void* ptrs[] = {h->par, h->thetaT, h->theta, h->adam_m, h->adam_v, h->grad, h->sumsq, h->sumsq2, h->beta_pow, h->hyper, h->norm_out, h->grad_src, h->x0g, h->dmug,
                    h->slots[0], h->slots[1], h->slabs, h->dw_tiles, h->st_obs, h->st_act, h->st_noise, h->st_loss, h->obs_rms.mean,
                    h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->stats_xch, h->stats_part, h->stats_counter, h->adv_xch, h->ro_obs, h->ro_act,
                    h->ro_val, h->ro_nlp, h->ro_done, h->ro_rew, h->ro_ret, h->env_in /* raw_obs, raw_rew, cur_done live inside */, h->raw_done,
                    h->last_val, h->ro_noise, h->mb_obs, h->mb_act, h->mb_adv, h->mb_ret, h->mb_val, h->mb_nlp, h->d_perms, h->d_inv, h->d_gidx, h->d_advstats, h->d_keys, h->d_loss_rows, h->d_loss_mean};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int t = 0; t < 2; ++t) for (int l = 0; l < PPO_MAX_LAYERS; ++l) { if (h->hg[t][l]) (void)hipFree(h->hg[t][l]); if (h->dyg[t][l]) (void)hipFree(h->dyg[t][l]); }
    for (int i = 0; i < 6; ++i) if (h->st_vec[i]) (void)hipFree(h->st_vec[i]);
    {
        ppo_handle::Bf16& b = h->bf;
        void* bp[] = {b.theta_bf, b.x0, b.xe, b.head_out[0], b.head_out[1], b.dhead[0], b.dhead[1], b.dbias, b.dw_tiles};
        for (void* p : bp) if (p) (void)hipFree(p);
        for (int t = 0; t < 2; ++t) for (int l = 0; l < PPO_MAX_LAYERS; ++l) for (bf16_t* p : {b.hb[t][l], b.dy[t][l]}) if (p) (void)hipFree(p);
    }
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->hyper_host) (void)hipHostFree(h->hyper_host);
    if (h->vram_in) (void)hipFree(h->vram_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_flag) (void)hipHostFree(h->pin_flag);
    if (h->pin_wgflag) (void)hipHostFree(h->pin_wgflag);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ppo_sync(ppo_handle* h) { ENTER(h); HIP_OK(h, hipStreamSynchronize(h->stream)); return 0; }

// ---- variables ----------------------------------------------------------------------------------------------------
int ppo_num_tensors(const ppo_handle* h) { return (int)h->tensors.size(); }
int ppo_num_params(const ppo_handle* h) { return h->P_dense; }

int ppo_tensor_info(const ppo_handle* h, int index, char name[32], int32_t* rows, int32_t* cols) {
    if (index < 0 || index >= (int)h->tensors.size()) return -1;
    const Tensor& t = h->tensors[index];
    if (name) snprintf(name, 32, "%s", t.name);
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return 0;
}

int ppo_get_tensor(ppo_handle* h, int which, int index, float* dst, int64_t count) {
    ENTER_Q(h);
    float* base = which_buf(h, which);
    if (!base || index < 0 || index >= (int)h->tensors.size()) return fail(h, "ppo_get_tensor: bad which/index");
    const Tensor& t = h->tensors[index];
    if (count != t.count()) return fail(h, "ppo_get_tensor(%s): count %lld != %d", t.name, (long long)count, t.count());
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return copy_tensor(h, base, t, dst, false);
}

int ppo_set_tensor(ppo_handle* h, int which, int index, const float* src, int64_t count) {
    ENTER_Q(h);
    float* base = which_buf(h, which);
    if (!base || index < 0 || index >= (int)h->tensors.size()) return fail(h, "ppo_set_tensor: bad which/index");
    const Tensor& t = h->tensors[index];
    if (count != t.count()) return fail(h, "ppo_set_tensor(%s): count %lld != %d", t.name, (long long)count, t.count());
    HIP_OK(h, hipStreamSynchronize(h->stream));
    if (copy_tensor(h, base, t, const_cast<float*>(src), true)) return -1;
    if (which == 0) {
        hipLaunchKernelGGL(transpose_refresh_kernel, dim3(h->n_blocks), dim3(256), 0, h->stream, h->theta, h->thetaT, h->par, h->grad_src, h->narrow ? h->nw_img : nullptr);
        HIP_OK(h, hipGetLastError());
        if (h->bf.on && bf16_refresh_mirrors(h)) return -1;
    }
    return 0;
}

int ppo_get_flat(ppo_handle* h, int which, float* dst, int64_t count) {
    if (count != h->P_dense) return fail(h, "ppo_get_flat: count %lld != %d", (long long)count, h->P_dense);
    for (size_t i = 0; i < h->tensors.size(); ++i)
        if (ppo_get_tensor(h, which, (int)i, dst + h->tensors[i].off_dense, h->tensors[i].count())) return -1;
    return 0;
}

int ppo_set_flat(ppo_handle* h, int which, const float* src, int64_t count) {
    if (count != h->P_dense) return fail(h, "ppo_set_flat: count %lld != %d", (long long)count, h->P_dense);
    for (size_t i = 0; i < h->tensors.size(); ++i)
        if (ppo_set_tensor(h, which, (int)i, src + h->tensors[i].off_dense, h->tensors[i].count())) return -1;
    return 0;
}

int ppo_get_beta_powers(ppo_handle* h, float pw[2]) {
    ENTER_Q(h);
    float v[4];
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(v, h->beta_pow, sizeof v, hipMemcpyDeviceToHost));
    pw[0] = v[2]; pw[1] = v[3];          // "next" = the value the following train step will use
    return 0;
}

int ppo_set_beta_powers(ppo_handle* h, const float pw[2]) {
    ENTER_Q(h);
    const float v[4] = {pw[0], pw[1], pw[0], pw[1]};
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(h->beta_pow, v, sizeof v, hipMemcpyHostToDevice));
    return 0;
}

int ppo_init_orthogonal(ppo_handle* h, uint64_t seed) {
    // orthogonal initialiser of the same family as the constants in G (a16): rows or columns orthonormal, scaled by gain
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next_u = [&]() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto next_n = [&]() { const double u1 = ((next_u() >> 11) + 1.0) / 9007199254740993.0, u2 = (next_u() >> 11) / 9007199254740992.0; return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2); };
    std::vector<float> flat(h->P_dense, 0.f);
    for (const Tensor& t : h->tensors) {
        const std::string nm = t.name;
        if (nm.size() < 2 || nm.substr(nm.size() - 2) != "/w") continue;
        const double gain = nm == "pi/w" ? 0.01 : nm == "vf/w" ? 1.0 : sqrt(2.0);
        const int R = t.rows, C = t.dcols();
        const bool tall = R >= C;
        const int nv = tall ? C : R, len = tall ? R : C;            // nv orthonormal vectors of length len
        std::vector<double> q((size_t)nv * len);
        for (int v = 0; v < nv; ++v) {
            double* qv = &q[(size_t)v * len];
            for (int i = 0; i < len; ++i) qv[i] = next_n();
            for (int pass = 0; pass < 2; ++pass)                      // modified Gram-Schmidt, twice for orthogonality
                for (int u = 0; u < v; ++u) {
                    const double* qu = &q[(size_t)u * len];
                    double d = 0; for (int i = 0; i < len; ++i) d += qv[i] * qu[i];
                    for (int i = 0; i < len; ++i) qv[i] -= d * qu[i];
                }
            double nn = 0; for (int i = 0; i < len; ++i) nn += qv[i] * qv[i];
            nn = 1.0 / sqrt(nn);
            for (int i = 0; i < len; ++i) qv[i] *= nn;
        }
        float* w = &flat[t.off_dense];
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < C; ++c) w[(size_t)r * C + c] = (float)(gain * (tall ? q[(size_t)c * len + r] : q[(size_t)r * len + c]));
    }
    if (ppo_set_flat(h, 0, flat.data(), h->P_dense)) return -1;
    std::vector<float> zero(h->P_dense, 0.f);
    if (ppo_set_flat(h, 1, zero.data(), h->P_dense) || ppo_set_flat(h, 2, zero.data(), h->P_dense)) return -1;
    const float pw[2] = {h->cfg.adam_beta1, h->cfg.adam_beta2};
    return ppo_set_beta_powers(h, pw);
}

int ppo_seed(ppo_handle* h, uint64_t seed) {
    ENTER_Q(h);                                                    // a resident rollout kernel captured the old seed: retire it first
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    h->rng_seed = (uint32_t)(z ^ (z >> 32));
    h->rng_calls = 0;
    return 0;
}

// ---- act model ----------------------------------------------------------------------------------------------------
static int step_common(ppo_handle* h, const float* obs, int n, const float* noise, bool sample, float* action, float* det_action,
                       float* value, float* neglogp) {
    ENTER_Q(h);
    if (n < 1) return fail(h, "step: n must be positive");
    if (ensure_staging(h, n)) return -1;
    const NetDev& net = h->net;
    HIP_OK(h, hipMemcpyAsync(h->st_obs, obs, (size_t)n * net.O * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (noise) HIP_OK(h, hipMemcpyAsync(h->st_noise, noise, (size_t)n * net.A * sizeof(float), hipMemcpyHostToDevice, h->stream));
    StepArgs a{};
    a.theta = h->theta; a.par = h->par; a.obs = h->st_obs; a.noise = noise ? h->st_noise : nullptr;
    a.action = (sample && action) ? h->st_act : nullptr;
    a.det_action = det_action ? h->st_act : nullptr;
    a.value = value ? h->st_vec[0] : nullptr;
    a.neglogp = neglogp ? h->st_vec[1] : nullptr;
    a.obs_out = nullptr; a.nz = no_norm(); a.n = n;
    // (only a call that actually draws from the counter RNG advances it: ppo_value / ppo_act_deterministic in the middle of a
    // rollout must not shift the rollout's noise)
    a.seed = h->rng_seed; a.rng_step = (sample && action && !noise) ? h->rng_calls++ : h->rng_calls; a.row_base = (uint32_t)h->rank * (uint32_t)(h->nz_envs > 0 ? h->nz_envs : n);
    if (launch_step(h, a)) return -1;
    if (action || det_action) HIP_OK(h, hipMemcpyAsync(action ? action : det_action, h->st_act, (size_t)n * net.A * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (value) HIP_OK(h, hipMemcpyAsync(value, h->st_vec[0], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (neglogp) HIP_OK(h, hipMemcpyAsync(neglogp, h->st_vec[1], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (bf16_chain_err_async(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    if (bf16_chain_err_test(h)) {
        // a stateless pass: run it again, layer by layer this time (the handle has switched), and keep the finding on stderr
        fprintf(stderr, "libppo_hip: %s -- the pass was repeated with a launch per layer\n", h->err.c_str());
        return step_common(h, obs, n, noise, sample, action, det_action, value, neglogp);
    }
    return 0;
}

int ppo_step(ppo_handle* h, const float* obs, int32_t n, const float* noise, float* action, float* value, float* neglogp) {
    return step_common(h, obs, n, noise, true, action, nullptr, value, neglogp);
}
int ppo_value(ppo_handle* h, const float* obs, int32_t n, float* value) {
    return step_common(h, obs, n, nullptr, false, nullptr, nullptr, value, nullptr);
}
int ppo_act_deterministic(ppo_handle* h, const float* obs, int32_t n, float* action) {
    return step_common(h, obs, n, nullptr, false, nullptr, action, nullptr, nullptr);
}

// ---- train op -------------------------------------------------------------------------------------------------------
int ppo_train_step(ppo_handle* h, float lr, float cliprange, const float* obs, const float* actions, const float* advs,
                   const float* returns, const float* old_neglogp, const float* old_values, int32_t n, float losses[5]) {
    ENTER_Q(h);
    if (n < 2) return fail(h, "ppo_train_step: n=%d (the reference asserts more than one row, ppo2.hpp:402)", n);
    h->bf.epoch_staged = false;
    if (ensure_staging(h, n) || ensure_train_ws(h, n)) return -1;
    const NetDev& net = h->net;
    const size_t fb = sizeof(float);
    HIP_OK(h, hipMemcpyAsync(h->st_obs, obs, (size_t)n * net.O * fb, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->st_act, actions, (size_t)n * net.A * fb, hipMemcpyHostToDevice, h->stream));
    const float* vecs[4] = {advs, returns, old_neglogp, old_values};
    for (int i = 0; i < 4; ++i) HIP_OK(h, hipMemcpyAsync(h->st_vec[2 + i], vecs[i], (size_t)n * fb, hipMemcpyHostToDevice, h->stream));
    if (set_hyper(h, lr, cliprange)) return -1;
    TrainArgs ta{};
    ta.obs = h->st_obs; ta.actions = h->st_act; ta.advs = h->st_vec[2]; ta.returns = h->st_vec[3]; ta.old_neglogp = h->st_vec[4];
    ta.old_values = h->st_vec[5]; ta.adv_stats = nullptr; ta.n = n;
    ta.inv_n = 1.0f / (float)((int64_t)n * h->world);
    if (h->dw2 && zero_words(h, h->dw2_counters, DW2_TILES)) return -1;
    if (enqueue_train(h, ta, h->st_loss)) return -1;
    h->bf.grad_lazy = h->bf.lazy_last;
    HIP_OK(h, hipMemcpyAsync(losses, h->st_loss, 5 * fb, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    prof_collect(h);
    return adam_meet_check(h) || bf16_chain_check(h) ? -1 : 0;
}

int ppo_get_last_grad(ppo_handle* h, float* dst, int64_t count, float* global_norm) {
    ENTER(h);
    if (count != h->P_dense) return fail(h, "ppo_get_last_grad: count %lld != %d", (long long)count, h->P_dense);
    if (bf16_materialize_grad(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    for (const Tensor& t : h->tensors) if (copy_tensor(h, h->grad, t, dst + t.off_dense, false)) return -1;
    if (global_norm) HIP_OK(h, hipMemcpy(global_norm, h->norm_out, sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---- host-loop numerics on the device -------------------------------------------------------------------------------
int ppo_adv_normalize(ppo_handle* h, const float* returns, const float* values, int32_t n, float* advs) {
    ENTER(h);
    if (n < 1) return fail(h, "ppo_adv_normalize: n must be positive");
    if (ensure_staging(h, n)) return -1;
    HIP_OK(h, hipMemcpyAsync(h->st_vec[0], returns, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->st_vec[1], values, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(adv_normalize_kernel, dim3(1), dim3(1024), 0, h->stream, h->st_vec[0], h->st_vec[1], n, h->st_vec[2]);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipMemcpyAsync(advs, h->st_vec[2], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_gae(ppo_handle* h, const float* rewards, const float* values, const float* dones, const float* last_values,
            const float* last_dones, int32_t T, int32_t E, float gamma, float lam, float* returns) {
    ENTER(h);
    if (T < 1 || E < 1) return fail(h, "ppo_gae: bad shape");
    const size_t n = (size_t)T * E;
    if (n > (size_t)1 << 30) return fail(h, "ppo_gae: too large");
    if (ensure_staging(h, (int)n)) return -1;
    const float* src[5] = {rewards, values, dones, last_values, last_dones};
    const size_t cnt[5] = {n, n, n, (size_t)E, (size_t)E};
    for (int i = 0; i < 5; ++i) HIP_OK(h, hipMemcpyAsync(h->st_vec[i], src[i], cnt[i] * sizeof(float), hipMemcpyHostToDevice, h->stream));
    {
        ProfScope ps(h, PK_GAE);
        launch_gae(h, h->st_vec[0], h->st_vec[1], h->st_vec[2], h->st_vec[3], h->st_vec[4], T, E, gamma, lam, h->st_vec[5]);
        HIP_OK(h, hipGetLastError());
    }
    HIP_OK(h, hipMemcpyAsync(returns, h->st_vec[5], n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int norm_alloc_stats(ppo_handle* h, NormDev& s, int dim) {
    if (dev_alloc(h, &s.mean, dim) || dev_alloc(h, &s.var, dim) || dev_alloc(h, &s.count, 1)) return -1;
    std::vector<float> ones(dim, 1.0f);
    const double c0 = 1e-6;                                    // running_statistics.hpp:17-21
    HIP_OK(h, hipMemcpyAsync(s.var, ones.data(), dim * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(s.count, &c0, sizeof c0, hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_norm_init(ppo_handle* h, int32_t n_envs, float gamma, float clip_obs, float clip_rew, float epsilon) {
    ENTER_Q(h);
    if (n_envs < 1) return fail(h, "ppo_norm_init: n_envs must be positive");
    h->nz_envs = n_envs; h->nz_gamma = gamma; h->nz_clip_obs = clip_obs; h->nz_clip_rew = clip_rew; h->nz_eps = epsilon;
    if (norm_alloc_stats(h, h->obs_rms, h->net.O) || norm_alloc_stats(h, h->ret_rms, 1)) return -1;
    if (dev_alloc(h, &h->nz_ret, n_envs) || dev_alloc(h, &h->stats_counter, 2 + 256) ||
        dev_alloc(h, &h->stats_part, std::max((size_t)NB_MAX_OBS_BLOCKS * ru(1 + 2 * h->net.O, 32), (size_t)NB_CG_MAX_WG * NB_CG_STRIDE) + (size_t)NB_MAX_REW_BLOCKS * NB_REW_STRIDE)) return -1;
    // raw observations | raw rewards | dones of the current env step in ONE block (one H2D copy per env step on the
    // host-Env path) with a pinned host mirror owned by the handle (replaces the pageable Utils::convert_* copies of
    // ppo2/utils.hpp:17-73)
    const size_t in_n = (size_t)n_envs * (h->net.O + 2);
    if (dev_alloc(h, &h->env_in, in_n) || dev_alloc(h, &h->raw_done, n_envs)) return -1;
    h->raw_obs = h->env_in; h->raw_rew = h->env_in + (size_t)n_envs * h->net.O; h->cur_done = h->raw_rew + n_envs;
    if (h->pin_in) { (void)hipHostFree(h->pin_in); h->pin_in = nullptr; }
    if (h->pin_out) { (void)hipHostFree(h->pin_out); h->pin_out = nullptr; }
    HIP_OK(h, hipHostMalloc((void**)&h->pin_in, in_n * sizeof(float), hipHostMallocDefault));
    HIP_OK(h, hipHostMalloc((void**)&h->pin_out, (size_t)n_envs * h->net.A * sizeof(float), hipHostMallocDefault));
    if (!h->pin_flag) { HIP_OK(h, hipHostMalloc((void**)&h->pin_flag, 1024, hipHostMallocDefault)); memset(h->pin_flag, 0, 1024); h->act_seq = 0; }
    if (h->pin_wgflag) { (void)hipHostFree(h->pin_wgflag); h->pin_wgflag = nullptr; }
    h->pin_wgflag_n = (n_envs + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    HIP_OK(h, hipHostMalloc((void**)&h->pin_wgflag, (size_t)h->pin_wgflag_n * sizeof(unsigned), hipHostMallocDefault));
    memset(h->pin_wgflag, 0, (size_t)h->pin_wgflag_n * sizeof(unsigned)); h->wg_seq = 0;
    { const char* e = getenv("PPO_HIP_NO_DIRECT_ACT"); h->opt_no_direct_act = e && e[0] == '1'; }
    {
        void* d = nullptr;
        HIP_OK(h, hipHostGetDevicePointer(&d, h->pin_wgflag, 0)); h->pin_wgflag_dev = (unsigned*)d;
        HIP_OK(h, hipHostGetDevicePointer(&d, h->pin_in, 0)); h->pin_in_dev = (float*)d;
        HIP_OK(h, hipHostGetDevicePointer(&d, h->pin_out, 0)); h->pin_out_dev = (float*)d;
        HIP_OK(h, hipHostGetDevicePointer(&d, h->pin_flag, 0)); h->pin_flag_dev = (unsigned*)d;
    }
    h->host_pending = false;
    h->pin_in_n = in_n; h->pin_out_n = (size_t)n_envs * h->net.A;
    {
        if (h->vram_in) { (void)hipFree(h->vram_in); h->vram_in = nullptr; h->vram_h2d = nullptr; }
        int large_bar = 0;
        const char* nv = getenv("PPO_HIP_NO_VRAM_INBOX");
        if (n_envs == 1 && h->narrow && h->nw_static && !(nv && nv[0] == '1') &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) == hipSuccess && large_bar) {
            const size_t words = ru((int)in_n, 32) + 64;
            float* box = nullptr;
            if (hipMalloc((void**)&box, words * sizeof(float)) == hipSuccess) {
                (void)hipMemset(box, 0, words * sizeof(float));
                (void)hipDeviceSynchronize();
                // is the allocation mapped read-write into this process (the host will store into it)?
                if (host_can_store(box, words * sizeof(float))) { h->vram_in = box; h->vram_h2d = reinterpret_cast<unsigned*>(box + ru((int)in_n, 32) + 32); }
                else (void)hipFree(box);
            }
        }
    }
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_norm_set_flags(ppo_handle* h, int norm_obs, int norm_reward) {
    h->norm_obs_flag = norm_obs ? 1 : 0;
    h->norm_rew_flag = norm_reward ? 1 : 0;
    return 0;
}

int ppo_norm_reset_returns(ppo_handle* h) {
    ENTER(h);
    if (!h->nz_envs) return fail(h, "ppo_norm_reset_returns: call ppo_norm_init first");
    if (host_quiesce(h)) return -1;
    HIP_OK(h, hipMemsetAsync(h->nz_ret, 0, (size_t)h->nz_envs * sizeof(float), h->stream));      // env_normalize.hpp:114
    return 0;
}

// device-side pieces shared by the host-pointer API and the rollout loop
static int allreduce_f32(ppo_handle* h, float* buf, size_t count) { return enqueue_allreduce(h, buf, count); }

// EnvNormalize::step's statistics + reward branch for one batch of envs: one multi-block launch; with a communicator
// the ranks' batch moments are exchanged with ONE all-reduce (slot table = all-gather) and a one-block finish, so the
// running statistics are over the environments of ALL ranks.  obs_dev / rew_dev may be null (job absent).
static int enqueue_norm_batch(ppo_handle* h, const float* obs_dev, int rows, const float* rew_dev, const float* done_dev, int training_rew,
                              float* rew_out, float* done_copy) {
    if (!obs_dev && !rew_dev) return 0;
    NormBatchArgs a{};
    const int D = h->net.O;
    a.D = D; a.obs = obs_dev; a.rows = rows; a.obs_st = h->obs_rms;
    if (obs_dev && D % 64 == 0 && D <= 64 * 254 && rows >= 256) {
        // wide observations: 64-column groups x row splits (obs_cgroup_job): splits of 128 rows (a thread's 8 rows stay in registers) where the table
        // of sets allows it (<= NB_CG_MAX_WG workgroups, <= 256 splits for the combine)
        a.n_strips = D / 64;
        int S = std::min((rows + 127) / 128, std::min(16 * NB_CG_KP, NB_CG_MAX_WG / a.n_strips));
        a.rows_per_obs_block = (rows + S - 1) / S;
        a.n_splits = (rows + a.rows_per_obs_block - 1) / a.rows_per_obs_block;
        a.g_obs = a.n_strips * a.n_splits;
    } else if (obs_dev) {
        // ~2048 values per workgroup: 8 independent loads per thread and pass
        const int cap = D <= 32 ? NB_MAX_OBS_BLOCKS : 64;       // wide rows: the final combine walks the chunks per column
        a.g_obs = std::max(1, std::min(cap, (int)(((int64_t)rows * D + 2047) / 2048)));
        a.g_obs = std::min(a.g_obs, rows);
        a.rows_per_obs_block = (rows + a.g_obs - 1) / a.g_obs;
        a.g_obs = (rows + a.rows_per_obs_block - 1) / a.rows_per_obs_block;
    }
    a.rew = rew_dev; a.dones = done_dev; a.ret = h->nz_ret; a.ret_st = h->ret_rms; a.rew_out = rew_out; a.done_copy = done_copy;
    a.rew_rows = rows; a.training_rew = (training_rew && h->norm_rew_flag) ? 1 : 0; a.scale_rew = h->norm_rew_flag;
    if (rew_dev) {
        a.g_rew = std::max(1, std::min(NB_MAX_REW_BLOCKS, (rows + 1023) / 1024));
        // whole multiples of 32 rows: a chunk owner rewrites its rows of `ret` for the job's last arriver to read, and no 128-byte line of
        // that hand-off may belong to two workgroups (NB_ST in ppo_kernels.hpp)
        a.rows_per_rew_block = ru((rows + a.g_rew - 1) / a.g_rew, 32);
        a.g_rew = (rows + a.rows_per_rew_block - 1) / a.rows_per_rew_block;
    }
    a.gamma = h->nz_gamma; a.clip_rew = h->nz_clip_rew; a.eps = h->nz_eps;
    a.part = h->stats_part; a.part_stride = ru(1 + 2 * D, 32); a.counter = reinterpret_cast<unsigned*>(h->stats_counter);
    a.world = h->world; a.rank = h->rank;
    const size_t xw = (size_t)(1 + 2 * D) + 3;
    // over peer-mapped regions the two statistics kernels exchange the table themselves (the last workgroup of each job writes this
    // rank's moments into every rank's gather area and raises a flag there; norm_finalize_kernel waits for the flags): no memset, no
    // push / sum launches between them.  PPO_HIP_PEER_STATS=0: the table goes through the general all-reduce as before.
    static const bool no_peer_stats = [] { const char* e = getenv("PPO_HIP_PEER_STATS"); return e && e[0] == '0'; }();
    a.use_peer = (h->comm && h->peer.on && h->peer.dev.scap >= (int)xw && !no_peer_stats) ? 1 : 0;
    if (a.use_peer) a.peer = h->peer.dev;
    if (h->comm) {
        if (!h->stats_xch || h->stats_xch_world != h->world) {
            if (dev_alloc(h, &h->stats_xch, xw * h->world)) return -1;
            h->stats_xch_world = h->world;
        }
        a.xch = h->stats_xch;
        if (!a.use_peer) HIP_OK(h, hipMemsetAsync(h->stats_xch, 0, xw * h->world * sizeof(float), h->stream));
    }
#ifdef PPO_STAMPS
    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
    a.stamps = g_stamps + 4096 * 32;
#endif
    { ProfScope ps(h, PK_STATS);
      hipLaunchKernelGGL(norm_batch_kernel, dim3(a.g_obs + a.g_rew), dim3(NB_THREADS), 0, h->stream, a);
      HIP_OK(h, hipGetLastError()); }
    if (h->comm) {
        if (!a.use_peer && allreduce_f32(h, h->stats_xch, xw * h->world)) return -1;
        ProfScope ps(h, PK_STATS);
        hipLaunchKernelGGL(norm_finalize_kernel, dim3(2), dim3(NB_THREADS), 0, h->stream, a);
        HIP_OK(h, hipGetLastError());
    }
    return 0;
}

int ppo_norm_obs(ppo_handle* h, const float* raw_obs, int32_t n_envs, int training, float* out) {
    ENTER(h);
    if (!h->nz_envs) return fail(h, "ppo_norm_obs: call ppo_norm_init first");
    if (host_quiesce(h)) return -1;
    if (n_envs != h->nz_envs) return fail(h, "ppo_norm_obs: n_envs %d != %d", n_envs, h->nz_envs);
    const size_t cnt = (size_t)n_envs * h->net.O;
    if (ensure_staging(h, n_envs)) return -1;
    HIP_OK(h, hipMemcpyAsync(h->raw_obs, raw_obs, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (!h->norm_obs_flag) {                                  // env_normalize.hpp:107-109: pass-through
        if (out != raw_obs) memcpy(out, raw_obs, cnt * sizeof(float));
        return 0;
    }
    if (training && enqueue_norm_batch(h, h->raw_obs, n_envs, nullptr, nullptr, 0, nullptr, nullptr)) return -1;
    hipLaunchKernelGGL(obs_normalize_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->raw_obs, n_envs, h->net.O, h->obs_rms,
                       h->nz_eps, h->nz_clip_obs, h->st_obs);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipMemcpyAsync(out, h->st_obs, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_norm_reward(ppo_handle* h, const float* raw_rew, const float* dones, int32_t n_envs, int training, float* out) {
    ENTER(h);
    if (!h->nz_envs) return fail(h, "ppo_norm_reward: call ppo_norm_init first");
    if (host_quiesce(h)) return -1;
    if (n_envs != h->nz_envs) return fail(h, "ppo_norm_reward: n_envs %d != %d", n_envs, h->nz_envs);
    if (ensure_staging(h, n_envs)) return -1;
    HIP_OK(h, hipMemcpyAsync(h->raw_rew, raw_rew, (size_t)n_envs * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemcpyAsync(h->raw_done, dones, (size_t)n_envs * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (enqueue_norm_batch(h, nullptr, n_envs, h->raw_rew, h->raw_done, training, h->st_vec[0], nullptr)) return -1;
    HIP_OK(h, hipMemcpyAsync(out, h->st_vec[0], (size_t)n_envs * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_norm_get_stats(ppo_handle* h, int which, float* mean, float* var, double* count) {
    ENTER(h);
    if (!h->nz_envs) return fail(h, "ppo_norm_get_stats: call ppo_norm_init first");
    if (host_quiesce(h)) return -1;
    NormDev& s = which == 0 ? h->obs_rms : h->ret_rms;
    const int dim = which == 0 ? h->net.O : 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(mean, s.mean, dim * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(h, hipMemcpy(var, s.var, dim * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(h, hipMemcpy(count, s.count, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int ppo_norm_set_stats(ppo_handle* h, int which, const float* mean, const float* var, double count) {
    ENTER(h);
    if (!h->nz_envs) return fail(h, "ppo_norm_set_stats: call ppo_norm_init first");
    if (host_quiesce(h)) return -1;
    NormDev& s = which == 0 ? h->obs_rms : h->ret_rms;
    const int dim = which == 0 ? h->net.O : 1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(s.mean, mean, dim * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(s.var, var, dim * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(s.count, &count, sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

// ---- rollout ----------------------------------------------------------------------------------------------------------
int ppo_rollout_alloc(ppo_handle* h, int32_t E, int32_t T) {
    ENTER(h);
    if (E < 1 || T < 1) return fail(h, "ppo_rollout_alloc: bad shape");
    if (!h->nz_envs && ppo_norm_init(h, E, 0.99f, 10.f, 10.f, 1e-8f)) return -1;
    if (h->nz_envs != E) return fail(h, "ppo_rollout_alloc: n_envs %d != normaliser's %d", E, h->nz_envs);
    if (host_quiesce(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    const NetDev& n = h->net;
    const size_t B = (size_t)E * T;
    if (dev_alloc(h, &h->ro_obs, B * n.O) || dev_alloc(h, &h->ro_act, B * n.A) || dev_alloc(h, &h->ro_val, B) || dev_alloc(h, &h->ro_nlp, B) ||
        dev_alloc(h, &h->ro_done, B) || dev_alloc(h, &h->ro_rew, B) || dev_alloc(h, &h->ro_ret, B) ||
        dev_alloc(h, &h->last_val, E) || dev_alloc(h, &h->ro_noise, B * n.A))
        return -1;
    h->E = E; h->T = T;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// The current observations stay RAW in raw_obs: the statistics update runs when they arrive (update-then-normalise,
// env_normalize.hpp:94-105) and the scale + clip happens in the policy step's input staging, which also writes the
// normalised rows into the rollout buffer -- no separate normalise kernel, no device-to-device copy.
static ObsNorm obs_norm(ppo_handle* h) { return ObsNorm{h->obs_rms.mean, h->obs_rms.var, h->nz_eps, h->nz_clip_obs, h->norm_obs_flag}; }

// policy step on the current observations -> rollout[t]
// direct: the policy tower also publishes the actions to the host itself (StepArgs::host_action)
static int enqueue_rollout_act(ppo_handle* h, int t, const float* noise_dev, uint32_t seed, uint32_t rng_step, uint32_t row_base, bool direct = false) {
    const NetDev& n = h->net;
    const size_t E = h->E;
    if (h->done_staged != t) HIP_OK(h, hipMemcpyAsync(h->ro_done + t * E, h->cur_done, E * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    StepArgs a{};
    a.theta = h->theta; a.par = h->par; a.obs = h->raw_obs; a.noise = noise_dev; a.action = h->ro_act + t * E * n.A; a.det_action = nullptr;
    a.value = h->ro_val + t * E; a.neglogp = h->ro_nlp + t * E; a.obs_out = h->ro_obs + t * E * n.O; a.nz = obs_norm(h); a.n = (int)E;
    a.seed = seed; a.rng_step = rng_step; a.row_base = row_base;
    if (direct) { a.host_action = h->pin_out_dev; a.host_flags = h->pin_wgflag_dev; a.host_seq = ++h->wg_seq; }
    return launch_step(h, a);
}

// statistics of the observations that just arrived in raw_obs, the reward branch, and the dones of the next rollout row
static int enqueue_observe(ppo_handle* h, int t) {
    const size_t E = h->E;
    float* done_next = t + 1 < h->T ? h->ro_done + (size_t)(t + 1) * E : nullptr;
    if (enqueue_norm_batch(h, h->norm_obs_flag ? h->raw_obs : nullptr, (int)E, h->raw_rew, h->cur_done, 1, h->ro_rew + (size_t)t * E, done_next)) return -1;
    h->done_staged = done_next ? t + 1 : -1;
    return 0;
}

static int enqueue_finish(ppo_handle* h, float gamma, float lam) {
    StepArgs a{};
    a.theta = h->theta; a.par = h->par; a.obs = h->raw_obs; a.value = h->last_val; a.nz = obs_norm(h); a.n = h->E;
    if (launch_step(h, a)) return -1;
    ProfScope ps(h, PK_GAE);
    launch_gae(h, h->ro_rew, h->ro_val, h->ro_done, h->last_val, h->cur_done, h->T, h->E, gamma, lam, h->ro_ret);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// Small host-Env rollouts (narrow path, one rank): up to NW_RO_MAX_E environments are served by the resident kernel
// (narrow_rollout_kernel in host mode), up to NW_ROWS also by one fused launch per env step (narrow_host_step_kernel).  While
// either is in use ("protocol" mode) ppo_rollout_observe only fills the pinned block: the transition is booked by the next
// launch / the resident kernel.
static bool host_small(const ppo_handle* h) {
    if (h->opt_no_host_fused) return false;
    const NetDev& n = h->net;
    return h->narrow && !h->comm && !h->bf.on && h->E <= NW_RO_MAX_E && n.O <= 64 && n.A <= 64 && h->pin_flag &&
           ((size_t)h->nw.lds_total + std::max(nw_ro_extra(h->E, n.O), NW_RO_EXTRA)) * sizeof(float) <= 160 * 1024;
}
static bool host_fused(const ppo_handle* h) { return host_small(h) && h->E <= NW_ROWS; }

// one launch of narrow_host_step_kernel: the pending transition's bookkeeping (if any) and, with act, the policy step of row t
static int enqueue_host_step(ppo_handle* h, int t, bool act, const float* noise_dev, uint32_t rng_step) {
    const NetDev& n = h->net;
    const size_t E = h->E;
    NwHostStepArgs q{};
    q.img = h->nw_img;
    q.st = NwEnvState{h->raw_obs, h->obs_rms.mean, h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->cur_done};
    q.host_in = h->pin_in_dev; q.host_act = h->pin_out_dev; q.host_flag = h->pin_flag_dev; q.flag_value = act ? ++h->act_seq : 0u;
    q.noise = noise_dev;
    if (act) { q.ro_obs = h->ro_obs + t * E * n.O; q.ro_act = h->ro_act + t * E * n.A; q.ro_nlp = h->ro_nlp + t * E; q.ro_done = h->ro_done + t * E; }
    q.has_transition = h->host_pending ? 1 : 0;
    q.ro_rew_prev = h->host_pending ? h->ro_rew + (size_t)h->host_pending_t * E : nullptr;
    q.E = (int)E; q.act = act ? 1 : 0;
    q.seed = h->rng_seed; q.rng_step = rng_step; q.row_base = (uint32_t)(h->rank * h->E);
    q.gamma = h->nz_gamma; q.clip_rew = h->nz_clip_rew; q.clip_obs = h->nz_clip_obs; q.eps = h->nz_eps; q.norm_obs = h->norm_obs_flag; q.norm_rew = h->norm_rew_flag;
    const size_t lds = ((size_t)h->nw.lds_total + NW_RO_EXTRA) * sizeof(float);
    ProfScope ps(h, PK_STEP);
#define X(a, b, c, d) hipLaunchKernelGGL((narrow_host_step_kernel<a, b, c, d>), dim3(1), dim3(NW_THREADS), lds, h->stream, n, h->nw, q)
    NW_DISPATCH(h, X);
#undef X
    HIP_OK(h, hipGetLastError());
    h->host_pending = false;
    return 0;
}

static void launch_rollout_kernel(ppo_handle* h, const NwRolloutArgs& q, size_t lds) {
    const NetDev& n = h->net;
    const bool multi = q.E > NW_ROWS;
    // ONE environment on the device env, reference shape: the whole rollout in one wave, weights in registers (ppo_rollout1.hpp)
    const char* e1 = getenv("PPO_HIP_NO_ROLLOUT1");               // (read per call: the tests compare both forms in one process)
    const bool no_r1 = e1 && e1[0] == '1';
    if (h->nw_static && q.E == 1 && !no_r1) {                      // (any observation width up to 64, any action width up to 32)
        ++h->kv[KV_ROLLOUT1];
        if (q.host_mode) {                                          // the same three waves resident behind a host Env (round 5)
            if (n.Kp0 == 32) hipLaunchKernelGGL((narrow_rollout1_kernel<32, true>), dim3(1), dim3(192), 0, h->stream, n, h->nw, q);
            else hipLaunchKernelGGL((narrow_rollout1_kernel<64, true>), dim3(1), dim3(192), 0, h->stream, n, h->nw, q);
        }
        else if (n.Kp0 == 32) hipLaunchKernelGGL((narrow_rollout1_kernel<32, false>), dim3(1), dim3(192), 0, h->stream, n, h->nw, q);
        else hipLaunchKernelGGL((narrow_rollout1_kernel<64, false>), dim3(1), dim3(192), 0, h->stream, n, h->nw, q);
        return;
    }
    ++h->kv[KV_ROLLOUT_PERSISTENT];
#define X(a, b, c, d) do { if (multi) hipLaunchKernelGGL((narrow_rollout_kernel<a, b, c, d, true>), dim3(1), dim3(NW_THREADS), lds, h->stream, n, h->nw, q); \
                           else hipLaunchKernelGGL((narrow_rollout_kernel<a, b, c, d, false>), dim3(1), dim3(NW_THREADS), lds, h->stream, n, h->nw, q); } while (0)
    NW_DISPATCH(h, X);
#undef X
}

// the VRAM inbox's sequence word, stored by the host through the BAR (write-combined: fenced on both sides)
static void vram_word(ppo_handle* h, unsigned v) {
    if (!h->vram_h2d) return;
    host_store_fence();
    *reinterpret_cast<volatile unsigned*>(h->vram_h2d) = v;
    host_store_fence();
}

// ---- resident host-Env rollout kernel (see NwRolloutArgs) -----------------------------------------------------------------------
// control words live at pin_flag + 64 (the first line is the per-launch completion word of narrow_host_step_kernel)
static unsigned* hp_ctl(ppo_handle* h) { return h->pin_flag + 64; }
static bool host_resident(const ppo_handle* h) { return host_small(h) && !h->opt_no_host_resident; }
static int hp_launch(ppo_handle* h, int t0, uint32_t rng_step_t0) {
    const NetDev& n = h->net;
    NwRolloutArgs q{};
    q.img = h->nw_img;
    q.st = NwEnvState{h->raw_obs, h->obs_rms.mean, h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->cur_done};
    q.ro_obs = h->ro_obs; q.ro_act = h->ro_act; q.ro_nlp = h->ro_nlp; q.ro_rew = h->ro_rew; q.ro_done = h->ro_done;
    q.E = h->E; q.T = h->T; q.seed = h->rng_seed; q.step0 = rng_step_t0 - (uint32_t)t0; q.env0 = h->rank * h->E;
    q.gamma = h->nz_gamma; q.clip_rew = h->nz_clip_rew; q.clip_obs = h->nz_clip_obs; q.eps = h->nz_eps; q.norm_obs = h->norm_obs_flag; q.norm_rew = h->norm_rew_flag;
    q.host_mode = 1; q.t0 = t0; q.pending = h->host_pending ? 1 : 0;
    q.host_in = h->pin_in_dev; q.host_act = h->pin_out_dev; q.ctl = h->pin_flag_dev + 64;
    { const char* e1 = getenv("PPO_HIP_NO_ROLLOUT1");
      if (h->vram_in && h->E == 1 && h->nw_static && !(e1 && e1[0] == '1')) { q.host_in = h->vram_in; q.h2d = h->vram_h2d; } }     // (narrow_rollout1_kernel<.., HOST>: system-scope loads only)
    const char* pc = getenv("PPO_HIP_HOST_POLLS");
    q.poll_cap = pc ? (unsigned)atol(pc) : 150000u;                 // ~2 us per poll over PCIe: a fraction of a second, then the kernel parks itself
    __atomic_store_n(hp_ctl(h) + PCTL_EXIT, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(hp_ctl(h) + PCTL_STOP, 0u, __ATOMIC_RELEASE);
    const size_t lds = ((size_t)h->nw.lds_total + (h->E > NW_ROWS ? nw_ro_extra(h->E, n.O) : NW_RO_EXTRA)) * sizeof(float);
    launch_rollout_kernel(h, q, lds);
    HIP_OK(h, hipGetLastError());
    h->hp_active = true; h->host_pending = false;
    return 0;
}
// the resident kernel has left (or is asked to and waited for): fold its report into the host's bookkeeping
static int hp_retire(ppo_handle* h, bool ask) {
    if (!h->hp_active) return 0;
    unsigned* ctl = hp_ctl(h);
    if (ask) __atomic_store_n(ctl + PCTL_STOP, 1u, __ATOMIC_RELEASE);
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const unsigned ex = __atomic_load_n(ctl + PCTL_EXIT, __ATOMIC_ACQUIRE);
    h->hp_active = false;
    if (!ex) return fail(h, "host-Env rollout: the resident kernel ended without its exit report");
    const int booked = (int)ex - 1;
    h->host_pending = booked < h->hp_posted;                       // a posted transition the kernel did not get to: the next launch books it
    h->host_pending_t = h->hp_posted - 1;
    return 0;
}

// a posted transition that no kernel has booked yet, booked now (outside the resident kernel)
static int host_flush_pending(ppo_handle* h) {
    if (!h->host_pending) return 0;
    if (h->E <= NW_ROWS) {
        if (enqueue_host_step(h, 0, false, nullptr, 0)) return -1;
        h->pin_in_busy = true;                                      // that launch reads the pinned block in place: nobody rewrites it before a synchronisation
        return 0;
    }
    const size_t E = h->E, on = E * h->net.O;                       // more than one row group: the general path's copy + statistics kernel
    HIP_OK(h, hipMemcpyAsync(h->env_in, h->pin_in, (on + 2 * E) * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->pin_in_busy = true;
    h->host_pending = false;
    return enqueue_observe(h, h->host_pending_t);
}

static int host_quiesce(ppo_handle* h) {
    if (!h->hp_active && !h->host_pending) return 0;
    if (hp_retire(h, true)) return -1;
    return host_flush_pending(h);
}

int ppo_rollout_reset(ppo_handle* h, const float* raw_obs) {
    if (!h->E) return fail(h, "ppo_rollout_reset: call ppo_rollout_alloc first");
    ENTER(h);
    const size_t on = (size_t)h->E * h->net.O;
    if (host_quiesce(h)) return -1;                             // (a transition observed before the reset is still booked, as on the general path)
    if (h->pin_in_busy) HIP_OK(h, hipStreamSynchronize(h->stream));
    h->pin_in_busy = false; h->host_pending = false; h->hp_posted = 0;
    if (h->pin_flag) { __atomic_store_n(hp_ctl(h) + PCTL_H2D, 0u, __ATOMIC_RELEASE); __atomic_store_n(hp_ctl(h) + PCTL_D2H, 0u, __ATOMIC_RELEASE); vram_word(h, 0u); }
    memcpy(h->pin_in, raw_obs, on * sizeof(float));
    HIP_OK(h, hipMemcpyAsync(h->raw_obs, h->pin_in, on * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_OK(h, hipMemsetAsync(h->nz_ret, 0, (size_t)h->E * sizeof(float), h->stream));          // env_normalize.hpp:114
    HIP_OK(h, hipMemsetAsync(h->cur_done, 0, (size_t)h->E * sizeof(float), h->stream));        // runner.hpp:50
    h->done_staged = -1;
    if (h->norm_obs_flag && enqueue_norm_batch(h, h->raw_obs, h->E, nullptr, nullptr, 0, nullptr, nullptr)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int ppo_rollout_act(ppo_handle* h, int32_t t, const float* noise, float* actions_out) {
    if (!h->E || t < 0 || t >= h->T) return fail(h, "ppo_rollout_act: bad step %d", t);
    ENTER(h);
    const NetDev& n = h->net;
    const size_t cnt = (size_t)h->E * n.A;
    float* nd = nullptr;
    if (noise) { nd = h->ro_noise; HIP_OK(h, hipMemcpyAsync(nd, noise, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream)); }
    if (host_resident(h) && !noise) {
        // the kernel stays resident over the rollout: actions and transitions travel through the pinned blocks, sequence words
        // order them; (re)launched here when it is not running (first step, or it parked itself after a long host pause)
        unsigned* ctl = hp_ctl(h);
        const uint32_t rng_step = h->rng_calls++;
        if (t == 0) {
            // a rollout abandoned without ppo_rollout_finish / _reset (an env exception, say) leaves its kernel resident and its
            // sequence words raised: retire it, book what it posted, and start this rollout from clean words -- stale actions
            // must never be handed out for step 0
            if (h->hp_active || __atomic_load_n(ctl + PCTL_D2H, __ATOMIC_ACQUIRE) != 0u || __atomic_load_n(ctl + PCTL_H2D, __ATOMIC_ACQUIRE) != 0u) {
                if (host_quiesce(h)) return -1;
                HIP_OK(h, hipStreamSynchronize(h->stream));
                __atomic_store_n(ctl + PCTL_H2D, 0u, __ATOMIC_RELEASE);
                __atomic_store_n(ctl + PCTL_D2H, 0u, __ATOMIC_RELEASE);
                vram_word(h, 0u);
            }
            h->hp_posted = 0;
        }
        for (int attempt = 0; ; ++attempt) {
            if (!h->hp_active) { if (attempt > 3) return fail(h, "ppo_rollout_act: the resident kernel keeps leaving before step %d", t); if (hp_launch(h, t, rng_step)) return -1; }
            bool have = false;
            for (unsigned long spins = 1; ; ++spins) {
                if (__atomic_load_n(ctl + PCTL_D2H, __ATOMIC_ACQUIRE) >= (unsigned)(t + 1)) { have = true; break; }
                if (__atomic_load_n(ctl + PCTL_EXIT, __ATOMIC_ACQUIRE)) { have = __atomic_load_n(ctl + PCTL_D2H, __ATOMIC_ACQUIRE) >= (unsigned)(t + 1); break; }
                if ((spins & 0xfffff) == 0) {                       // every million polls: is the stream still alive?
                    const hipError_t qe = hipStreamQuery(h->stream);
                    if (qe != hipSuccess && qe != hipErrorNotReady) return fail(h, "ppo_rollout_act: %s", hipGetErrorString(qe));
                    if (qe == hipSuccess && !__atomic_load_n(ctl + PCTL_EXIT, __ATOMIC_ACQUIRE) && __atomic_load_n(ctl + PCTL_D2H, __ATOMIC_ACQUIRE) < (unsigned)(t + 1))
                        return fail(h, "ppo_rollout_act: the resident kernel is gone without a report");
                }
            }
            if (have) break;
            if (hp_retire(h, false)) return -1;                     // it parked itself before producing row t: relaunch from here
        }
        h->pin_in_busy = false; h->done_staged = -1; h->host_proto = true;
        memcpy(actions_out, h->pin_out, cnt * sizeof(float));
        return 0;
    }
    if (host_small(h)) { if (hp_retire(h, true)) return -1; if (!host_fused(h) && host_flush_pending(h)) return -1; }
    h->host_proto = host_fused(h);
    if (host_fused(h)) {
        // <= 32 environments on the narrow path: ONE launch does the pending transition's EnvNormalize bookkeeping (read from the
        // pinned block), the policy tower and the action store into pinned memory; the host spins on the completion word
        if (enqueue_host_step(h, t, true, nd, h->rng_calls++)) return -1;
        const unsigned want = h->act_seq;
        unsigned long spins = 0;
        while (__atomic_load_n(h->pin_flag, __ATOMIC_ACQUIRE) != want) {
            if ((++spins & 0x3ffff) == 0) {                     // every ~quarter million polls: is the stream still alive?
                const hipError_t qe = hipStreamQuery(h->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) return fail(h, "ppo_rollout_act: %s", hipGetErrorString(qe));
                if (qe == hipSuccess && __atomic_load_n(h->pin_flag, __ATOMIC_ACQUIRE) != want) return fail(h, "ppo_rollout_act: the step kernel finished without publishing");
            }
        }
        h->pin_in_busy = false; h->done_staged = -1;
        memcpy(actions_out, h->pin_out, cnt * sizeof(float));
        return 0;
    }
    // policy_step_kernel (the fp32 families wider than 64) can hand the actions over itself: no copy command, no wait for the value tower.  Used for
    // up to 64 environments only: shader stores into host memory cost ~1 us per 16-row block once more than a few workgroups publish (measured per
    // env step, direct | copy engine: 64 envs 25.4 | 31.6 us, 256: 45 | 43, 1024: 55 | 49, 4096: 80 | 66 -- profiles/r05_b_*).
    // PPO_HIP_DIRECT_ACT_MAX_BLOCKS overrides the limit (tests run the form at 256 blocks).
    const char* dme = getenv("PPO_HIP_DIRECT_ACT_MAX_BLOCKS");                   // (read per call: a test switches it inside one process)
    const int direct_max = dme ? atoi(dme) : 4;
    const bool direct = !h->opt_no_direct_act && !h->narrow && !h->bf.on && h->pin_wgflag && (h->E + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK <= direct_max;
    if (enqueue_rollout_act(h, t, nd, h->rng_seed, h->rng_calls++, (uint32_t)(h->rank * h->E), direct)) return -1;
    if (direct) {
        // Watch the table: block b's 16 rows are copied out as soon as its word shows this call's sequence number (the copy of the early blocks
        // hides under the kernel's tail).  The policy kernel is stream-ordered behind the H2D copy of the last observation and the statistics
        // kernel, so once its last block has published, the pinned input block is free again -- without a stream synchronisation.
        const unsigned want = h->wg_seq;
        const int G = (h->E + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
        const size_t blk = (size_t)ROWS_PER_BLOCK * n.A;
        unsigned long spins = 0;
        for (int b = 0; b < G; ) {
            if (__atomic_load_n(h->pin_wgflag + b, __ATOMIC_ACQUIRE) == want) {
                const size_t off = (size_t)b * blk, cntb = std::min(blk, cnt - off);
                memcpy(actions_out + off, h->pin_out + off, cntb * sizeof(float));
                ++b;
                continue;
            }
            if ((++spins & 0x3ffff) == 0) {                     // every ~quarter million polls: is the stream still alive?
                const hipError_t qe = hipStreamQuery(h->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) return fail(h, "ppo_rollout_act: %s", hipGetErrorString(qe));
                if (qe == hipSuccess && __atomic_load_n(h->pin_wgflag + b, __ATOMIC_ACQUIRE) != want) return fail(h, "ppo_rollout_act: the policy kernel finished without publishing block %d", b);
            }
        }
        h->pin_in_busy = false;
        return 0;
    }
    // one D2H into the handle's pinned landing buffer and the ONLY stream synchronisation of an env step: the statistics
    // kernel of the previous ppo_rollout_observe, this policy step and the copy drain together.  (Round 6 tried a copy-out KERNEL in its place -- 32 workgroups storing
    // 16 bytes per lane into the pinned buffer, a pinned word per slice, the host polling the words instead of the stream: no gain at 4096 environments, 1.25 vs 1.17 ms
    // per 16 env steps on one box, profiles/r06_c_*: this phase is the H2D copy of the last transition + the statistics kernel + the policy kernel in front of the copy,
    // not the copy command.)
    HIP_OK(h, hipMemcpyAsync(h->pin_out, h->ro_act + (size_t)t * cnt, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (bf16_chain_err_async(h)) return -1;
    // busy-wait on the stream: the blocking synchronise parks the thread on an interrupt and wakes it ~100-200 us late,
    // several times the whole GPU-side cost of an env step; this thread has nothing else to do until the actions are here
    {
        hipError_t q;
        int spins = 0;
        while ((q = hipStreamQuery(h->stream)) == hipErrorNotReady) { if (++spins > 200000) { HIP_OK(h, hipStreamSynchronize(h->stream)); q = hipSuccess; break; } }
        if (q != hipSuccess) return fail(h, "hipStreamQuery failed: %s", hipGetErrorString(q));
    }
    h->pin_in_busy = false;
    if (bf16_chain_err_test(h)) return -1;                       // (row t of the rollout is invalid; the message says what happened and what the handle does from now on)
    memcpy(actions_out, h->pin_out, cnt * sizeof(float));
    return 0;
}

int ppo_rollout_observe(ppo_handle* h, int32_t t, const float* raw_obs, const float* raw_rew, const float* dones) {
    if (!h->E || t < 0 || t >= h->T) return fail(h, "ppo_rollout_observe: bad step %d", t);
    ENTER(h);
    const size_t E = h->E, on = E * h->net.O;
    // obs | reward | dones packed into the pinned mirror, ONE H2D copy; no synchronisation here: nothing of this call is
    // read by the host, and the pinned block is not rewritten before the next ppo_rollout_act has drained the stream
    if (h->pin_in_busy) HIP_OK(h, hipStreamSynchronize(h->stream));
    memcpy(h->pin_in, raw_obs, on * sizeof(float));
    memcpy(h->pin_in + on, raw_rew, E * sizeof(float));
    memcpy(h->pin_in + on + E, dones, E * sizeof(float));
    if (host_small(h) && h->host_proto) {                       // the (next or resident) launch reads the block in place
        h->host_pending = true; h->host_pending_t = t; h->hp_posted = t + 1;
        if (h->vram_in) { memcpy(h->vram_in, h->pin_in, (on + 2 * E) * sizeof(float)); vram_word(h, (unsigned)(t + 1)); }      // (posted writes; the data first)
        __atomic_store_n(hp_ctl(h) + PCTL_H2D, (unsigned)(t + 1), __ATOMIC_RELEASE);
        return 0;
    }
    HIP_OK(h, hipMemcpyAsync(h->env_in, h->pin_in, (on + 2 * E) * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->pin_in_busy = true;
    if (enqueue_observe(h, t)) return -1;
    return 0;
}

int ppo_rollout_finish(ppo_handle* h, float gamma, float lam) {
    ENTER(h);
    if (!h->E) return fail(h, "ppo_rollout_finish: call ppo_rollout_alloc first");
    if (host_small(h)) {
        if (h->hp_active) {
            // after its last row the resident kernel waits for the last transition, books it and leaves by itself; if the host
            // stopped early it is asked to leave
            if (hp_retire(h, h->hp_posted < h->T)) return -1;
        }
        if (h->pin_flag) { __atomic_store_n(hp_ctl(h) + PCTL_H2D, 0u, __ATOMIC_RELEASE); __atomic_store_n(hp_ctl(h) + PCTL_D2H, 0u, __ATOMIC_RELEASE); vram_word(h, 0u); }
        if (host_flush_pending(h)) return -1;                                              // the last transition's bookkeeping
        StepArgs va{};                                                                     // values of all T x E normalised rows, batched
        va.obs = h->ro_obs; va.value = h->ro_val; va.n = h->E * h->T; va.nz = no_norm();
        if (launch_step(h, va)) return -1;
    }
    if (enqueue_finish(h, gamma, lam) || bf16_chain_err_async(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    prof_collect(h);
    if (bf16_chain_err_test(h)) return -1;
    return peer_check(h);
}

int ppo_collect_synthetic(ppo_handle* h, uint32_t seed, int32_t env0, uint32_t step0, int first, const float* noise, float gamma, float lam) {
    ENTER(h);
    if (!h->E) return fail(h, "ppo_collect_synthetic: call ppo_rollout_alloc first");
    const NetDev& n = h->net;
    const int E = h->E, T = h->T;
    const int envW = E * (n.O + 2);
    if (noise) HIP_OK(h, hipMemcpyAsync(h->ro_noise, noise, (size_t)E * T * n.A * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (first) {
        { ProfScope ps(h, PK_ENV);
          hipLaunchKernelGGL(seeded_env_kernel, dim3((envW + 255) / 256), dim3(256), 0, h->stream, seed, env0, E, step0, n.O, h->raw_obs, (float*)nullptr, (float*)nullptr);
          HIP_OK(h, hipGetLastError()); }
        HIP_OK(h, hipMemsetAsync(h->nz_ret, 0, (size_t)E * sizeof(float), h->stream));
        HIP_OK(h, hipMemsetAsync(h->cur_done, 0, (size_t)E * sizeof(float), h->stream));
        h->done_staged = -1;
        if (enqueue_norm_batch(h, h->raw_obs, E, nullptr, nullptr, 0, nullptr, nullptr)) return -1;
    }
    // small environment counts on the narrow path: one fused launch per env step (policy step + env + EnvNormalize bookkeeping)
    const bool fused = h->narrow && !h->comm && E <= NW_ROWS && n.O <= 64;
    // ... and on top of that the whole rollout in ONE launch of one persistent workgroup (narrow_rollout_kernel): state in LDS,
    // only stores leave the CU; the value tower runs afterwards, batched over the T x E normalised rows
    const char* npe = getenv("PPO_HIP_NO_PERSISTENT_COLLECT");
    const bool no_persist = npe && npe[0] == '1';
    // (up to NW_RO_MAX_E environments: the one workgroup walks them in groups of 32 rows)
    const size_t ro_lds = ((size_t)h->nw.lds_total + (E > NW_ROWS ? nw_ro_extra(E, n.O) : NW_RO_EXTRA)) * sizeof(float);
    const bool persistent = h->narrow && !h->comm && !no_persist && E <= NW_RO_MAX_E && n.O <= 64 && n.A <= 64 && ro_lds <= 160 * 1024;
    if (persistent) {
        NwRolloutArgs q{};
        q.img = h->nw_img;
        q.st = NwEnvState{h->raw_obs, h->obs_rms.mean, h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->cur_done};
        q.noise = noise ? h->ro_noise : nullptr;
        q.ro_obs = h->ro_obs; q.ro_act = h->ro_act; q.ro_nlp = h->ro_nlp; q.ro_rew = h->ro_rew; q.ro_done = h->ro_done;
        q.E = E; q.T = T; q.seed = seed; q.step0 = step0; q.env0 = env0;
        q.gamma = h->nz_gamma; q.clip_rew = h->nz_clip_rew; q.clip_obs = h->nz_clip_obs; q.eps = h->nz_eps; q.norm_obs = h->norm_obs_flag; q.norm_rew = h->norm_rew_flag;
#ifdef PPO_STAMPS
        if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
        q.stamps = g_stamps;
#endif
        { ProfScope ps(h, PK_STEP);
          launch_rollout_kernel(h, q, ro_lds);
          HIP_OK(h, hipGetLastError()); }
        StepArgs va{};                                         // values of all T x E rows: the rows are already normalised
        va.obs = h->ro_obs; va.value = h->ro_val; va.n = E * T; va.nz = no_norm();
        if (launch_step(h, va)) return -1;
        h->done_staged = -1;
    }
    // 65..2048 environments: G = ceil(E / 32) resident workgroups, one per CU, meeting once per env step for the statistics
    const int coopG = (E + NW_ROWS - 1) / NW_ROWS;
    const bool coop = !persistent && h->narrow && !h->comm && !no_persist && E > NW_RO_MAX_E && coopG <= NW_COOP_MAX_G && n.O <= 64 && n.A <= 64 &&
                      ((size_t)h->nw.lds_total + NW_RO_EXTRA) * sizeof(float) <= 160 * 1024 &&
                      coopG * (2 * n.O + 3) <= h->nw.w_total - h->nw.w_fwd;             // the chunks of a step are staged in the image's unused backward half
    if (coop) {
        const size_t pf = (size_t)2 * coopG * NW_COOP_PW;
        if (h->nw_coop_G != coopG) {
            HIP_OK(h, hipStreamSynchronize(h->stream));
            if (dev_alloc(h, &h->nw_coop, pf + 16 * (size_t)(coopG + 1))) return -1;
            h->nw_coop_G = coopG;
        }
        unsigned* ctl = reinterpret_cast<unsigned*>(h->nw_coop + pf);       // [G][16] step words, then the error word
        HIP_OK(h, hipMemsetAsync(ctl, 0, 16 * (size_t)(coopG + 1) * sizeof(unsigned), h->stream));
        NwCoopArgs q{};
        q.img = h->nw_img;
        q.st = NwEnvState{h->raw_obs, h->obs_rms.mean, h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->cur_done};
        q.noise = noise ? h->ro_noise : nullptr;
        q.ro_obs = h->ro_obs; q.ro_act = h->ro_act; q.ro_nlp = h->ro_nlp; q.ro_rew = h->ro_rew; q.ro_done = h->ro_done;
        q.E = E; q.T = T; q.G = coopG; q.seed = seed; q.step0 = step0; q.env0 = env0;
        q.gamma = h->nz_gamma; q.clip_rew = h->nz_clip_rew; q.clip_obs = h->nz_clip_obs; q.eps = h->nz_eps; q.norm_obs = h->norm_obs_flag; q.norm_rew = h->norm_rew_flag;
        q.part = h->nw_coop; q.arrive = ctl; q.err = ctl + 16 * (size_t)coopG; q.spin_limit = 4000000u;
        const size_t lds = ((size_t)h->nw.lds_total + NW_RO_EXTRA) * sizeof(float);
        ++h->kv[KV_ROLLOUT_COOP];
        { ProfScope ps(h, PK_STEP);
#define X(a, b, c, d) hipLaunchKernelGGL((narrow_rollout_coop_kernel<a, b, c, d>), dim3(coopG), dim3(NW_THREADS), lds, h->stream, n, h->nw, q)
          NW_DISPATCH(h, X);
#undef X
          HIP_OK(h, hipGetLastError()); }
        StepArgs va{};
        va.obs = h->ro_obs; va.value = h->ro_val; va.n = E * T; va.nz = no_norm();
        if (launch_step(h, va)) return -1;
        h->done_staged = -1;
    }
    if (fused && !persistent) {
        if (h->nw_alt_envs != E) {
            HIP_OK(h, hipStreamSynchronize(h->stream));
            if (dev_alloc(h, &h->nw_alt, (size_t)E * n.O + 2 * n.O + 2 + 2 * (size_t)E) || dev_alloc(h, &h->nw_alt_counts, 2)) return -1;
            h->nw_alt_envs = E;
        }
        float* alt = h->nw_alt;
        NwEnvState st[2];
        st[0] = NwEnvState{h->raw_obs, h->obs_rms.mean, h->obs_rms.var, h->obs_rms.count, h->ret_rms.mean, h->ret_rms.var, h->ret_rms.count, h->nz_ret, h->cur_done};
        st[1] = NwEnvState{alt, alt + (size_t)E * n.O, alt + (size_t)E * n.O + n.O, h->nw_alt_counts, alt + (size_t)E * n.O + 2 * n.O, alt + (size_t)E * n.O + 2 * n.O + 1,
                           h->nw_alt_counts + 1, alt + (size_t)E * n.O + 2 * n.O + 2, alt + (size_t)E * n.O + 2 * n.O + 2 + E};
        const size_t lds = (size_t)h->nw.lds_total * sizeof(float);
        for (int t = 0; t < T; ++t) {
            const NwEnvState& in = st[t & 1];
            StepArgs a{};
            a.theta = h->nw_img; a.obs = in.raw_obs; a.noise = noise ? h->ro_noise + (size_t)t * E * n.A : nullptr;
            a.action = h->ro_act + (size_t)t * E * n.A; a.value = h->ro_val + (size_t)t * E; a.neglogp = h->ro_nlp + (size_t)t * E;
            a.obs_out = h->ro_obs + (size_t)t * E * n.O; a.nz = ObsNorm{in.obs_mean, in.obs_var, h->nz_eps, h->nz_clip_obs, h->norm_obs_flag}; a.n = E;
            a.seed = seed; a.rng_step = step0 + (uint32_t)t; a.row_base = (uint32_t)env0;
            NwCollectArgs c{in, st[(t + 1) & 1], seed, step0 + (uint32_t)t + 1u, env0, h->nz_gamma, h->nz_clip_rew, h->nz_eps, h->norm_obs_flag, h->norm_rew_flag,
                            h->ro_rew + (size_t)t * E, h->ro_done + (size_t)t * E};
            ProfScope ps(h, PK_STEP);
            ++h->kv[KV_COLLECT_FUSED];
#define X(p, q_, r, s_) hipLaunchKernelGGL((narrow_collect_kernel<p, q_, r, s_>), dim3(1, 2), dim3(NW_THREADS), lds, h->stream, n, h->nw, a, c)
            NW_DISPATCH(h, X);
#undef X
            HIP_OK(h, hipGetLastError());
        }
        if (T & 1) {                                           // the live state sits in the second set: bring it home
            const NwEnvState& s1 = st[1]; const NwEnvState& s0 = st[0];
            const size_t fb = sizeof(float);
            HIP_OK(h, hipMemcpyAsync(s0.raw_obs, s1.raw_obs, (size_t)E * n.O * fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.obs_mean, s1.obs_mean, n.O * fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.obs_var, s1.obs_var, n.O * fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.obs_count, s1.obs_count, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.ret_mean, s1.ret_mean, fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.ret_var, s1.ret_var, fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.ret_count, s1.ret_count, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.ret, s1.ret, (size_t)E * fb, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(h, hipMemcpyAsync(s0.done, s1.done, (size_t)E * fb, hipMemcpyDeviceToDevice, h->stream));
        }
        h->done_staged = -1;
    }
    for (int t = 0; t < T && !fused && !persistent && !coop; ++t) {
        if (enqueue_rollout_act(h, t, noise ? h->ro_noise + (size_t)t * E * n.A : nullptr, seed, step0 + t, (uint32_t)env0)) return -1;
        { ProfScope ps(h, PK_ENV);
          hipLaunchKernelGGL(seeded_env_kernel, dim3((envW + 255) / 256), dim3(256), 0, h->stream, seed, env0, E, step0 + (uint32_t)t + 1u, n.O, h->raw_obs, h->raw_rew, h->cur_done);
          HIP_OK(h, hipGetLastError()); }
        if (enqueue_observe(h, t)) return -1;
    }
    if (enqueue_finish(h, gamma, lam) || bf16_chain_err_async(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    prof_collect(h);
    if (bf16_chain_err_test(h)) return -1;                       // (the rollout is invalid)
    if (coop) {
        unsigned e = 0;
        HIP_OK(h, hipMemcpy(&e, reinterpret_cast<unsigned*>(h->nw_coop + (size_t)2 * coopG * NW_COOP_PW) + 16 * (size_t)coopG, sizeof e, hipMemcpyDeviceToHost));
        if (e) return fail(h, "ppo_collect_synthetic: workgroup %u of the cooperative rollout gave up waiting for the others", e - 1);
    }
    return peer_check(h);
}

static float* rollout_field(ppo_handle* h, int field, size_t* count) {
    const size_t B = (size_t)h->E * h->T;
    switch (field) {
        case 0: *count = B * h->net.O; return h->ro_obs;
        case 1: *count = B * h->net.A; return h->ro_act;
        case 2: *count = B; return h->ro_val;
        case 3: *count = B; return h->ro_nlp;
        case 4: *count = B; return h->ro_done;
        case 5: *count = B; return h->ro_rew;
        case 6: *count = B; return h->ro_ret;
    }
    return nullptr;
}

int ppo_rollout_download(ppo_handle* h, int field, float* dst, int64_t count) {
    ENTER_Q(h);
    size_t c = 0;
    float* p = h->E ? rollout_field(h, field, &c) : nullptr;
    if (!p || (size_t)count != c) return fail(h, "ppo_rollout_download: bad field/count");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(dst, p, c * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int ppo_rollout_upload(ppo_handle* h, int field, const float* src, int64_t count) {
    ENTER_Q(h);
    size_t c = 0;
    float* p = h->E ? rollout_field(h, field, &c) : nullptr;
    if (!p || (size_t)count != c) return fail(h, "ppo_rollout_upload: bad field/count");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    HIP_OK(h, hipMemcpy(p, src, c * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// ---- update -----------------------------------------------------------------------------------------------------------
// every device buffer of a handle by name (ppo_debug_buffer): pointer (null = this shape does not use it) and length in 4-byte words
struct DbgEnt { const char* name; const void* p; size_t words; };
static std::vector<DbgEnt> debug_table(ppo_handle* h) {
    const NetDev& n = h->net;
    const size_t P = (size_t)h->P_pad, R = (size_t)h->ws_rows, U = (size_t)h->upd_cap_rows;
    const bool ws = !(h->narrow || h->bf.on);                 // the fp32 wide path's train workspaces
    return {
        {"theta", h->theta, P}, {"adam_m", h->adam_m, P}, {"adam_v", h->adam_v, P}, {"thetaT", h->thetaT, (size_t)h->PT}, {"par", h->par, (size_t)2 * n.par_total},
        {"grad", h->grad, P + 256}, {"sumsq", h->sumsq, (size_t)4 * h->n_blocks}, {"beta_pow", h->beta_pow, 4}, {"hyper", h->hyper, 2}, {"norm_out", h->norm_out, 1},
        {"dw2_parts", h->dw2_parts, (size_t)DW2_TILES + DW2_GRID}, {"dw2_counters", h->dw2_counters, (size_t)DW2_TILES},
        {"x0g", ws ? h->x0g : nullptr, R * n.Kp0}, {"dmug", ws ? h->dmug : nullptr, R * n.Ap},
        {"h_pi_0", ws ? h->hg[0][0] : nullptr, R * n.Hp[0]}, {"h_vf_0", ws ? h->hg[1][0] : nullptr, R * n.Hp[0]},
        {"dy_pi_0", ws ? h->dyg[0][0] : nullptr, R * n.Hp[0]}, {"dy_vf_0", ws ? h->dyg[1][0] : nullptr, R * n.Hp[0]},
        {"h_pi_1", ws && n.L > 1 ? h->hg[0][1] : nullptr, R * n.Hp[n.L > 1 ? 1 : 0]}, {"dy_pi_1", ws && n.L > 1 ? h->dyg[0][1] : nullptr, R * n.Hp[n.L > 1 ? 1 : 0]},
        {"dy_vf_1", ws && n.L > 1 ? h->dyg[1][1] : nullptr, R * n.Hp[n.L > 1 ? 1 : 0]},
        {"slots_pi", ws ? h->slots[0] : nullptr, (R / 16) * n.slot_w}, {"slots_vf", ws ? h->slots[1] : nullptr, (R / 16) * n.slot_w},
        {"slabs", h->narrow ? nullptr : h->slabs, (size_t)h->max_split * P},
        {"mb_obs", h->mb_obs, U * n.O}, {"mb_act", h->mb_act, U * n.A}, {"mb_adv", h->mb_adv, U}, {"mb_ret", h->mb_ret, U}, {"mb_val", h->mb_val, U}, {"mb_nlp", h->mb_nlp, U},
        {"gidx", h->d_gidx, U}, {"advstats", h->d_advstats, (size_t)2 * h->upd_cap_steps}, {"keys", h->d_keys, (size_t)2 * h->upd_cap_steps}, {"loss_rows", h->d_loss_rows, (size_t)5 * h->upd_cap_steps},
        {"nw_img", h->nw_img, h->narrow ? (size_t)2 * h->nw.w_total : 0}, {"nw_partials", h->nw_partials, (size_t)4 * h->nw_groups_cap * h->nw_stride},
        {"nw_theta1", h->nw_theta1, P}, {"nw_m1", h->nw_m1, P}, {"nw_v1", h->nw_v1, P}, {"nw_epoch_words", h->nw_epoch_words, NW_EPOCH_WORDS},
        {"obs_mean", h->obs_rms.mean, (size_t)n.O}, {"obs_var", h->obs_rms.var, (size_t)n.O}, {"nz_ret", h->nz_ret, (size_t)h->nz_envs}, {"cur_done", h->cur_done, (size_t)h->nz_envs},
    };
}

static int enqueue_update(ppo_handle* h, int epochs, int nmb, bool explicit_perms) {
    const int B = h->E * h->T, M = B / nmb;
    // the per-tile arrival counters of weight_grad_assemble_kernel are reset by their last arriver; an update that was cut short
    // (a failed launch) must not leave them half-counted for the next one: zeroed here, a memset node of the replayed graph
    if (h->dw2 && zero_words(h, h->dw2_counters, DW2_TILES)) return -1;
    h->nw_pending = false; h->nw_cur = 0;                      // outside an update the weights always live in set 0
    uint32_t bits = 1;
    while ((1u << bits) < (uint32_t)B) ++bits;
    for (int ep = 0; ep < epochs; ++ep) {
        if (explicit_perms) {
            const int Bp = (h->global_shuffle && h->comm && h->world > 1) ? B * h->world : B;
            hipLaunchKernelGGL(invert_perm_kernel, dim3((Bp + 255) / 256), dim3(256), 0, h->stream, h->d_perms + (size_t)ep * Bp, h->d_inv, Bp);
            HIP_OK(h, hipGetLastError());
        }
        bool merged = false;
        {
            ProfScope ps(h, PK_EPOCH);
            EpochArgs ea{};
            ea.inv_perm = explicit_perms ? h->d_inv : nullptr; ea.keys = h->d_keys + 2 * ep; ea.bits = bits;
            ea.B = B; ea.M = M; ea.T = h->T; ea.E = h->E; ea.returns = h->ro_ret; ea.values = h->ro_val; ea.gidx = h->d_gidx; ea.stats = h->d_advstats;
            ea.xch = h->adv_xch; ea.xch2 = ru(nmb, 4); ea.n_global = (float)((int64_t)M * h->world);
            const bool gs = h->global_shuffle && h->comm && h->world > 1;
            if (gs) {
                // ONE permutation over the rows of all ranks; the gathered returns / values are local, so the statistics of the whole
                // minibatch need no exchange
                uint32_t gb = 1;
                while ((1u << gb) < (uint32_t)B * (uint32_t)h->world) ++gb;
                ea.bits = gb; ea.world = h->world; ea.rank = h->rank; ea.returns = h->gs_ret; ea.values = h->gs_val; ea.phase = 0;
                hipLaunchKernelGGL(epoch_prepare_kernel, dim3(nmb), dim3(EP_THREADS), 0, h->stream, ea);
                HIP_OK(h, hipGetLastError());
            } else if (!h->comm) {
                ea.phase = 0;
                if (!h->bf.on && M <= EPG_MAX_M) {
                    // index map, advantage statistics AND the gather of the epoch in one launch (fp32 paths; the bf16 path stages its epoch separately)
                    GatherArgs ga{h->d_gidx, h->d_advstats, B, M, h->net.O, h->net.A, h->ro_obs, h->ro_act, h->ro_ret, h->ro_val, h->ro_nlp,
                                  h->mb_obs, h->mb_act, h->mb_adv, h->mb_ret, h->mb_val, h->mb_nlp};
#ifdef PPO_STAMPS
                    if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
                    ga.stamps = g_stamps + 4096 * 32;
#endif
                    hipLaunchKernelGGL(epoch_prepare_gather_kernel, dim3(nmb * EPG_SPLIT), dim3(EP_THREADS), 0, h->stream, ea, ga);
                    HIP_OK(h, hipGetLastError());
                    merged = true;
                } else {
                    hipLaunchKernelGGL(epoch_prepare_kernel, dim3(nmb), dim3(EP_THREADS), 0, h->stream, ea);
                    HIP_OK(h, hipGetLastError());
                }
            } else {
                // the advantage statistics are over the whole (all-rank) minibatch (ppo2.hpp:401-406)
                for (int phase = 1; phase <= 3; ++phase) {
                    ea.phase = phase;
                    hipLaunchKernelGGL(epoch_prepare_kernel, dim3(nmb), dim3(EP_THREADS), 0, h->stream, ea);
                    HIP_OK(h, hipGetLastError());
                    if (phase == 1 && allreduce_f32(h, h->adv_xch, (size_t)nmb)) return -1;
                    if (phase == 2 && allreduce_f32(h, h->adv_xch + ru(nmb, 4), (size_t)nmb)) return -1;
                }
            }
        }
        if (!merged) {
            ProfScope ps(h, PK_EPOCH);
            GatherArgs ga{h->d_gidx, h->d_advstats, B, M, h->net.O, h->net.A, h->ro_obs, h->ro_act, h->ro_ret, h->ro_val, h->ro_nlp,
                          h->mb_obs, h->mb_act, h->mb_adv, h->mb_ret, h->mb_val, h->mb_nlp};
            if (h->global_shuffle && h->comm && h->world > 1) { ga.obs = h->gs_obs; ga.act = h->gs_act; ga.ret = h->gs_ret; ga.val = h->gs_val; ga.nlp = h->gs_nlp; }
            const bool wide4 = h->net.O % 4 == 0 && h->net.A % 4 == 0;
            // bf16 path: the epoch's observations become bf16 once; a minibatch is then a row slice.  With 16-byte rows the gather writes them itself
            const bool fuse_stage = wide4 && h->bf.on && M % GB_PAD == 0 && h->bf.xe_rows >= B && h->net.Kp0 % 4 == 0 && !(h->global_shuffle && h->comm && h->world > 1);
            if (wide4) {
                Gather4Args g4{ga, fuse_stage ? h->bf.xe : nullptr, h->net.Kp0};
                hipLaunchKernelGGL(epoch_gather4_kernel, dim3((B + 15) / 16), dim3(256), 0, h->stream, g4);
            }
            else hipLaunchKernelGGL(epoch_gather_kernel, dim3((B + 15) / 16), dim3(256), 0, h->stream, ga);
            HIP_OK(h, hipGetLastError());
            if (h->bf.on && fuse_stage) h->bf.epoch_staged = true;
            else if (h->bf.on) {
                ppo_handle::Bf16& bb = h->bf;
                bb.epoch_staged = M % GB_PAD == 0 && bb.xe_rows >= B;
                if (bb.epoch_staged) {
                    StageArgsB sa{h->mb_obs, B, h->net.O, h->net.Kp0, B, no_norm(), nullptr, bb.xe};
                    const size_t cnt = (size_t)B * h->net.Kp0;
                    if (h->net.O % 4 == 0 && h->net.Kp0 % 4 == 0) hipLaunchKernelGGL(bf16_stage4_kernel, dim3(bf16_stage4_grid(h->net.Kp0, cnt / 4)), dim3(256), 0, h->stream, sa);
                    else hipLaunchKernelGGL(bf16_stage_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, sa);
                    HIP_OK(h, hipGetLastError());
                }
            }
        }
        const int egroups = (M + NW_ROWS - 1) / NW_ROWS;
        const char* ne = getenv("PPO_HIP_NO_NARROW_EPOCH");     // (read when the update is captured: the test compares both forms)
        if (h->narrow && h->nw_lazy && h->nw_epoch && !h->comm && egroups <= NW_EPOCH_MAX_G && egroups <= h->nw_groups_cap && !(ne && ne[0] == '1')) {
            // every minibatch of the epoch in ONE resident launch (ppo_narrow.hpp, narrow_epoch_kernel)
            ProfScope ps(h, PK_TRAIN_FB);
            const NetDev& n = h->net;
            const bool xl = h->nw_epoch_xl && h->nw_epoch_partials && (size_t)nmb * 2 * egroups * h->nw_stride <= h->nw_epoch_cap;
            NwEpochArgs ea{h->mb_obs, h->mb_act, h->mb_adv, h->mb_ret, h->mb_val, h->mb_nlp, M, nmb, 1.0f / (float)M, h->nw_img, xl ? h->nw_epoch_partials : h->nw_partials, h->nw_stride,
                           h->theta, h->adam_m, h->adam_v, h->grad, h->hyper, h->beta_pow, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->cfg.max_grad_norm,
                           h->d_loss_rows + (size_t)ep * nmb * 5, h->norm_out, h->nw_epoch_words, h->P_pad / 64, nullptr};
#ifdef PPO_STAMPS
            if (!g_stamps) (void)hipMalloc((void**)&g_stamps, 4096 * 48 * sizeof(unsigned long long));
            ea.stamps = g_stamps;
#endif
            const size_t lds = (size_t)h->nw.lds_total * sizeof(float);
            ++h->kv[KV_NARROW_EPOCH];
#define EPOCH_LAUNCH(K, X, EX, GRID) hipLaunchKernelGGL((narrow_epoch_kernel<K, X, EX>), GRID, dim3(NW_THREADS), lds, h->stream, n, h->nw, ea)
            const dim3 gx(16 * egroups), gw(egroups, 2);
            if (xl) {
                if (h->adam_exact) { if (n.Kp0 == 32) EPOCH_LAUNCH(32, true, true, gx); else EPOCH_LAUNCH(64, true, true, gx); }
                else if (n.Kp0 == 32) EPOCH_LAUNCH(32, true, false, gx); else EPOCH_LAUNCH(64, true, false, gx);
            } else {
                if (h->adam_exact) { if (n.Kp0 == 32) EPOCH_LAUNCH(32, false, true, gw); else EPOCH_LAUNCH(64, false, true, gw); }
                else if (n.Kp0 == 32) EPOCH_LAUNCH(32, false, false, gw); else EPOCH_LAUNCH(64, false, false, gw);
            }
#undef EPOCH_LAUNCH
            HIP_OK(h, hipGetLastError());
            continue;
        }
        for (int k = 0; k < nmb; ++k) {
            TrainArgs ta{};
            const size_t r0 = (size_t)k * M;
            ta.obs = h->mb_obs + r0 * h->net.O; ta.actions = h->mb_act + r0 * h->net.A; ta.returns = h->mb_ret + r0; ta.old_values = h->mb_val + r0;
            ta.old_neglogp = h->mb_nlp + r0; ta.advs = h->mb_adv + r0; ta.adv_stats = nullptr; ta.n = M;
            ta.inv_n = 1.0f / (float)((int64_t)M * h->world);
            if (enqueue_train(h, ta, h->d_loss_rows + (size_t)(ep * nmb + k) * 5, /*defer*/ true)) return -1;
        }
    }
    if (flush_pending_adam(h)) return -1;
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(320), 0, h->stream, h->d_loss_rows, epochs * nmb, h->d_loss_mean);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int ppo_update(ppo_handle* h, float lr, float cliprange, int32_t epochs, int32_t nmb, const int32_t* perms, uint64_t seed, float* loss_rows,
               float mean_losses[5]) {
    ENTER_Q(h);
    if (!h->E) return fail(h, "ppo_update: no rollout (ppo_rollout_alloc + collect first)");
    const int B = h->E * h->T;
    if (epochs < 1 || nmb < 1 || B % nmb) return fail(h, "ppo_update: n_batch %d not divisible by nminibatches %d", B, nmb);
    const int M = B / nmb;
    if (ensure_train_ws(h, M)) return -1;
    const int steps = epochs * nmb;
    const bool gs = h->global_shuffle && h->comm && h->world > 1;
    const int Bp = gs ? B * h->world : B;                      // rows one permutation covers
    if (gs && Bp > h->gs_rows) {
        HIP_OK(h, hipStreamSynchronize(h->stream));
        drop_graph(h);
        if (dev_alloc(h, &h->gs_obs, (size_t)Bp * h->net.O) || dev_alloc(h, &h->gs_act, (size_t)Bp * h->net.A) || dev_alloc(h, &h->gs_ret, Bp) ||
            dev_alloc(h, &h->gs_val, Bp) || dev_alloc(h, &h->gs_nlp, Bp)) return -1;
        h->gs_rows = Bp;
    }
    if (Bp > h->upd_cap_rows || steps > h->upd_cap_steps || !h->d_keys) {
        HIP_OK(h, hipStreamSynchronize(h->stream));
        drop_graph(h);
        const int cr = std::max(Bp, h->upd_cap_rows), cs = std::max(steps, h->upd_cap_steps);
        if (dev_alloc(h, &h->mb_obs, (size_t)cr * h->net.O) || dev_alloc(h, &h->mb_act, (size_t)cr * h->net.A) || dev_alloc(h, &h->mb_adv, cr) ||
            dev_alloc(h, &h->mb_ret, cr) || dev_alloc(h, &h->mb_val, cr) || dev_alloc(h, &h->mb_nlp, cr)) return -1;
        if (h->d_perms) { (void)hipFree(h->d_perms); h->d_perms = nullptr; h->upd_cap_epochs = 0; }      // sized on demand below
        if (dev_alloc(h, &h->d_inv, cr) || dev_alloc(h, &h->d_gidx, cr) ||
            dev_alloc(h, &h->d_advstats, (size_t)2 * cs) || dev_alloc(h, &h->d_keys, (size_t)2 * cs) || dev_alloc(h, &h->d_loss_rows, (size_t)5 * cs) ||
            dev_alloc(h, &h->d_loss_mean, 8) || dev_alloc(h, &h->adv_xch, (size_t)2 * ru(cs, 4)))
            return -1;
        h->upd_cap_rows = cr; h->upd_cap_steps = cs;
        if (h->bf.on) {
            if (dev_alloc(h, &h->bf.xe, (size_t)cr * h->net.Kp0)) return -1;
            h->bf.xe_rows = cr;
        }
    }
    if (h->narrow && h->nw_epoch && h->nw_epoch_xl && !h->comm && (M + NW_ROWS - 1) / NW_ROWS <= NW_EPOCH_MAX_G) {
        // the XCD-local epoch kernel's partial gradient vectors: one set per minibatch of an epoch (zero-filled: padding elements are never written and must read 0)
        const size_t need = (size_t)nmb * 2 * ((M + NW_ROWS - 1) / NW_ROWS) * h->nw_stride;
        if (need > h->nw_epoch_cap && need <= ((size_t)256 << 20) / sizeof(float)) {      // (thousands of tiny minibatches: the write-through form's two sets instead)
            HIP_OK(h, hipStreamSynchronize(h->stream));
            drop_graph(h);
            if (dev_alloc(h, &h->nw_epoch_partials, need)) return -1;
            h->nw_epoch_cap = need;
        }
    }
    if (set_hyper(h, lr, cliprange)) return -1;
    const bool explicit_perms = perms != nullptr;
    if (explicit_perms) {
        // every epoch's row must be a permutation of [0, Bp): invert_perm_kernel scatters inv[perm[i]] = i
        std::vector<unsigned char> seen((size_t)Bp);
        for (int ep = 0; ep < epochs; ++ep) {
            std::fill(seen.begin(), seen.end(), 0);
            const int32_t* pe = perms + (size_t)ep * Bp;
            for (int i = 0; i < Bp; ++i) {
                const int32_t d = pe[i];
                if (d < 0 || d >= Bp || seen[(size_t)d]) return fail(h, "ppo_update: perms[%d] is not a permutation of [0,%d) (entry %d = %d)", ep, Bp, i, (int)d);
                seen[(size_t)d] = 1;
            }
        }
        // [epochs, Bp] ints, allocated only when explicit permutations are used (on-device shuffles need none)
        if (!h->d_perms || epochs > h->upd_cap_epochs) {
            HIP_OK(h, hipStreamSynchronize(h->stream));
            drop_graph(h);
            if (dev_alloc(h, &h->d_perms, (size_t)epochs * h->upd_cap_rows)) return -1;
            h->upd_cap_epochs = epochs;
        }
        HIP_OK(h, hipMemcpyAsync(h->d_perms, perms, (size_t)epochs * Bp * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    if (gs) {
        // the rollout rows of all ranks, rank-major [world][T][E][.], once per update (10 MB per rank at config 3), outside the graph
        if (!h->rccl.AllGather) return fail(h, "ppo_update: the collective library has no ncclAllGather (needed by ppo_dist_global_shuffle)");
        const size_t rows = (size_t)B;
        struct { const float* src; float* dst; size_t w; } gl[5] = {{h->ro_obs, h->gs_obs, (size_t)h->net.O}, {h->ro_act, h->gs_act, (size_t)h->net.A},
                                                                    {h->ro_ret, h->gs_ret, 1}, {h->ro_val, h->gs_val, 1}, {h->ro_nlp, h->gs_nlp, 1}};
        for (auto& x : gl) {
            const int rc = h->rccl.AllGather(x.src, x.dst, rows * x.w, /*ncclFloat32*/ 7, h->comm, h->stream);
            if (rc != 0) return fail(h, "ncclAllGather failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(rc) : "?");
        }
    }
    std::vector<uint32_t> keys(2 * (size_t)epochs);
    for (int ep = 0; ep < epochs; ++ep) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(ep + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        keys[2 * ep] = (uint32_t)z; keys[2 * ep + 1] = (uint32_t)(z >> 32);
    }
    HIP_OK(h, hipMemcpyAsync(h->d_keys, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    // hipGraph replay of the whole update; RCCL calls and event-bracketed profiling run eagerly
    // With a communicator the sequence runs eagerly by default (the host enqueues a step faster than the GPU runs it);
    // PPO_HIP_GRAPH_RCCL=1 opts into capturing the ncclAllReduce calls as well.
    const bool graph_ok = h->use_graph && !h->prof && (!h->comm || h->graph_rccl || h->peer.on);
    if (graph_ok) {
        const bool same = h->upd_graph && h->g_epochs == epochs && h->g_nmb == nmb && h->g_E == h->E && h->g_T == h->T &&
                          h->g_explicit == (int)explicit_perms && h->g_world == (gs ? -h->world : h->world);
        if (!same) {
            drop_graph(h);
            HIP_OK(h, hipStreamSynchronize(h->stream));
            hipGraph_t graph = nullptr;
            // a runtime that cannot capture or instantiate this sequence is not fatal: the same launches run eagerly
            // (nothing has executed yet -- capture only records)
            bool ok = hipStreamBeginCapture(h->stream, h->comm ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int rc = enqueue_update(h, epochs, nmb, explicit_perms);
                const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
                ok = rc == 0 && ce == hipSuccess && graph != nullptr && hipGraphInstantiate(&h->upd_graph, graph, nullptr, nullptr, 0) == hipSuccess;
                if (ok) { h->upd_graph_tmpl = graph; h->bf.g_lazy = h->bf.lazy_last; }            // (kept alive beside the executable graph: drop_graph)
                else if (graph) (void)hipGraphDestroy(graph);
            }
            if (!ok) {
                (void)hipGetLastError();
                h->upd_graph = nullptr;
                h->use_graph = false;
                fprintf(stderr, "libppo_hip: hipGraph capture of the update failed (%s); continuing with eager launches\n", h->err.c_str());
            } else {
                h->g_epochs = epochs; h->g_nmb = nmb; h->g_E = h->E; h->g_T = h->T; h->g_explicit = (int)explicit_perms; h->g_world = gs ? -h->world : h->world;
            }
        }
        if (h->upd_graph) { HIP_OK(h, hipGraphLaunch(h->upd_graph, h->stream)); h->bf.lazy_last = h->bf.g_lazy; }
        else if (enqueue_update(h, epochs, nmb, explicit_perms)) return -1;
    } else if (enqueue_update(h, epochs, nmb, explicit_perms)) return -1;
    h->bf.grad_lazy = h->bf.lazy_last;
    if (loss_rows) HIP_OK(h, hipMemcpyAsync(loss_rows, h->d_loss_rows, (size_t)steps * 5 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipMemcpyAsync(mean_losses, h->d_loss_mean, 5 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(h, hipStreamSynchronize(h->stream));
    prof_collect(h);
    if (adam_meet_check(h) || bf16_chain_check(h) || nw_epoch_check(h)) return -1;
    return peer_check(h);
}

// ---- data parallel -------------------------------------------------------------------------------------------------------
static int load_rccl(Rccl& r, std::string& err) {
    if (r.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (const char* pick = getenv("PPO_RCCL_LIBRARY")) {       // a specific RCCL build (or the tests' shared-memory stand-in)
        r.lib = dlopen(pick, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { err = std::string("PPO_RCCL_LIBRARY: cannot load ") + pick + ": " + dlerror(); return -1; }
    }
    if (!r.lib) for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }   // prefer a copy torch already mapped
    if (!r.lib) for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
    if (!r.lib) { err = std::string("cannot load librccl: ") + dlerror(); return -1; }
    r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, char[128], int))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.lib, "ncclAllGather");
    r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
    r.CommCount = (int (*)(void*, int*))dlsym(r.lib, "ncclCommCount");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { err = "librccl lacks the expected nccl* symbols"; return -1; }
    return 0;
}

int ppo_dist_unique_id(char uid[128]) {
    static Rccl r;
    std::string err;
    if (load_rccl(r, err)) { g_create_error = err; return -1; }
    const int rc = r.GetUniqueId(uid);
    if (rc) { g_create_error = "ncclGetUniqueId failed"; return -1; }
    return 0;
}

struct UidByValue { char b[128]; };

int ppo_dist_init(ppo_handle* h, int32_t world, int32_t rank, const char uid[128]) {
    if (world < 1 || rank < 0 || rank >= world) return fail(h, "ppo_dist_init: bad world/rank");
    if (load_rccl(h->rccl, h->err)) return -1;
    HIP_OK(h, hipSetDevice(h->device));
    UidByValue u;
    memcpy(u.b, uid, 128);
    typedef int (*init_fn)(void**, int, UidByValue, int);
    init_fn init = (init_fn)dlsym(h->rccl.lib, "ncclCommInitRank");
    const int rc = init(&h->comm, world, u, rank);
    if (rc) return fail(h, "ncclCommInitRank failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(rc) : "?");
    h->world = world; h->rank = rank;
    drop_graph(h);
    { const char* e = getenv("PPO_HIP_NO_ADAM_MEET");
      h->adam_meet = !(e && e[0] == '1');
      if (h->adam_meet && (dev_alloc(h, &h->adam_meet_words, ADAM_MEET_MAX_GRID + 32) || dev_alloc(h, &h->adam_meet_parts, ADAM_MEET_MAX_GRID))) return -1;
      HIP_OK(h, hipStreamSynchronize(h->stream)); }
    if (world > 1) {
        // Do two ranks share a device (N processes on one GPU: the tests' dry runs)?  Then kernels whose workgroups wait for each other while
        // holding a CU each (train8_dw2_fused_kernel's grid-wide meeting) could starve one another: every rank publishes its device's PCI
        // location in its slot of a table, one all-reduce, and the handles of a job with a shared device fall back to the two-launch form.
        int dom = 0, bus = 0, dv = 0;
        (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->device);
        (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->device);
        (void)hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, h->device);
        std::vector<float> tab((size_t)world, 0.f);
        tab[(size_t)rank] = (float)(1 + (((dom & 0xff) << 16) | ((bus & 0xff) << 8) | (dv & 0xff)));      // exact in fp32 (< 2^24)
        float* dt = nullptr;
        bool shared = true;                                     // (unknown counts as shared)
        if (hipMalloc((void**)&dt, world * sizeof(float)) == hipSuccess &&
            hipMemcpy(dt, tab.data(), world * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
            h->rccl.AllReduce(dt, dt, (size_t)world, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream) == 0 &&
            hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(tab.data(), dt, world * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) {
            shared = false;
            for (int i = 0; i < world; ++i) for (int j = i + 1; j < world; ++j) if (tab[(size_t)i] == tab[(size_t)j]) shared = true;
        }
        if (dt) (void)hipFree(dt);
        h->dev_shared = shared;
    }
    // Collectives inside the update's hipGraph: with a communicator the eager sequence is 5-6 launches and up to three
    // collectives per train step issued from the host, which at narrow networks is slower than the GPU runs them.  Whether
    // this RCCL build can be stream-captured is PROBED here, collectively (every rank runs the same probe in the same order):
    // capture one small all-reduce, instantiate, replay twice, check the sum.  PPO_HIP_GRAPH_RCCL=0 / 1 overrides the probe.
    const char* ge = getenv("PPO_HIP_GRAPH_RCCL");
    if (ge && (ge[0] == '0' || ge[0] == '1')) h->graph_rccl = ge[0] == '1';
    else {
        // on a stream of its own: a library that cannot be captured (e.g. one that synchronises the stream inside the call)
        // invalidates the capture, and the handle's stream must not be left in that state
        h->graph_rccl = false;
        float* buf = nullptr;
        hipStream_t ps = nullptr;
        if (hipMalloc((void**)&buf, 64 * sizeof(float)) == hipSuccess && hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) == hipSuccess) {
            hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, ps, buf, 1.0f, (size_t)64);
            (void)hipStreamSynchronize(ps);
            hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(ps, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                const int arc = h->rccl.AllReduce(buf, buf, 64, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, ps);
                const hipError_t ce = hipStreamEndCapture(ps, &graph);
                ok = arc == 0 && ce == hipSuccess && graph != nullptr && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            }
            if (ok) ok = hipGraphLaunch(exec, ps) == hipSuccess && hipGraphLaunch(exec, ps) == hipSuccess && hipStreamSynchronize(ps) == hipSuccess;
            float got = 0.f;
            if (ok) ok = hipMemcpy(&got, buf, sizeof got, hipMemcpyDeviceToHost) == hipSuccess && got == (float)world * (float)world;
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            h->graph_rccl = ok;
        }
        if (ps) (void)hipStreamDestroy(ps);
        (void)hipGetLastError();
        if (buf) (void)hipFree(buf);
    }
    return 0;
}

int ppo_dist_world(const ppo_handle* h) { return h->world; }

int ppo_dist_info(ppo_handle* h, int32_t* comm_nranks, int32_t* device, char pci_bus_id[32], char library[256]) {
    ENTER(h);
    if (comm_nranks) {
        *comm_nranks = 0;
        if (h->comm) {
            int n = -1;
            if (!h->rccl.CommCount || h->rccl.CommCount(h->comm, &n) != 0) n = -1;
            *comm_nranks = n;
        }
    }
    if (device) *device = h->device;
    if (pci_bus_id) { pci_bus_id[0] = 0; HIP_OK(h, hipDeviceGetPCIBusId(pci_bus_id, 32, h->device)); }
    if (library) {
        library[0] = 0;
        Dl_info di;
        if (h->rccl.lib && h->rccl.AllReduce && dladdr((void*)h->rccl.AllReduce, &di) && di.dli_fname) snprintf(library, 256, "%s", di.dli_fname);
    }
    return 0;
}
int ppo_dist_graph_collectives(const ppo_handle* h) { return h->comm && (h->graph_rccl || h->peer.on) && h->use_graph ? 1 : 0; }

// ---- one-shot peer all-reduce (ppo_peer.hpp) ---------------------------------------------------------------------------------
// Region = [4 KB flag block | slots[2][world][cap]]; allocated uncached / fine-grained where the runtime offers it (the flags
// and slots are written by other devices), plain device memory otherwise (the kernels' system-scope fences do not depend on it).
static const size_t kPeerFlagBytes = 4096;

int ppo_dist_peer_export(ppo_handle* h, char handle[64]) {
    ENTER(h);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!h->comm) return fail(h, "ppo_dist_peer_export: call ppo_dist_init first");
    if (h->world > PEER_MAX_WORLD) return fail(h, "ppo_dist_peer_export: world %d > %d (one node)", h->world, PEER_MAX_WORLD);
    ppo_handle::Peer& P = h->peer;
    if (!P.region) {
        P.cap = (size_t)ru(std::max(h->P_pad + 8, 4096), PEER_CHUNK_MAX);
        P.scap = ru(2 * h->net.O + 4, 64);                        // a rank's batch moments of an env step: (n, mean[O], M2[O]) + (n, mean, M2) of the returns
        const size_t bytes = kPeerFlagBytes + (size_t)2 * h->world * (P.cap + P.scap) * sizeof(float);
        // memory kind of the region (other devices write into it): fine-grained = coherent at system scope with the kernels' fences, cached in L2; failing that, uncached
        // (every access goes to memory: slower, needs no fence to be seen).  Plain hipMalloc is NOT an option: a remote write may leave a stale line in this device's L2
        // that even sc1 loads can hit -- then the region exists (so that the collective attach can still run and agree) but the peer path is refused.
        hipError_t e = hipExtMallocWithFlags(&P.region, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&P.region, bytes, hipDeviceMallocUncached); }
        P.coarse = false;
        if (e != hipSuccess) {
            (void)hipGetLastError(); P.region = nullptr; HIP_OK(h, hipMalloc(&P.region, bytes));
            P.coarse = true;
            fprintf(stderr, "libppo_hip: no fine-grained / uncached device memory for the peer region (rank %d): the peer all-reduce stays off, RCCL is used\n", h->rank);
        }
        HIP_OK(h, hipMemset(P.region, 0, bytes));
        static_assert((PEER_SFLAG_OFF + 2 * 2 * PEER_MAX_WORLD * PEER_FLAG_STRIDE) * sizeof(unsigned) <= 4096, "statistics flags fit the flag block");
        HIP_OK(h, hipMalloc((void**)&P.local, 64));
        HIP_OK(h, hipMemset(P.local, 0, 64));
        HIP_OK(h, hipDeviceSynchronize());
    }
    hipIpcMemHandle_t ipc;
    HIP_OK(h, hipIpcGetMemHandle(&ipc, P.region));
    memcpy(handle, &ipc, 64);
    return 0;
}

// One peer all-reduce of a known pattern per parity; true when every element came back as the sum over the ranks
static bool peer_probe(ppo_handle* h) {
    ppo_handle::Peer& P = h->peer;
    const int n = 3000;                                            // not a multiple of the chunk: the tail path runs too
    std::vector<float> host(n), want(n);
    float* buf = nullptr;
    if (hipMalloc((void**)&buf, n * sizeof(float)) != hipSuccess) return false;
    bool ok = true;
    const PeerDev saved = P.dev;
    P.dev.spin_limit = 400000;                                     // a fraction of a second: a rank whose mapping failed never pushes
    P.on = true;
    for (int round = 0; round < 2 && ok; ++round) {
        for (int i = 0; i < n; ++i) {
            host[i] = (float)((h->rank + 1) * (i % 7 + 1 + round));
            want[i] = (float)((h->world * (h->world + 1) / 2) * (i % 7 + 1 + round));
        }
        ok = hipMemcpy(buf, host.data(), n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess && enqueue_allreduce(h, buf, n) == 0 &&
             hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(host.data(), buf, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
        for (int i = 0; i < n && ok; ++i) ok = host[i] == want[i];
    }
    unsigned e = 0;
    if (hipMemcpy(&e, P.dev.err, sizeof e, hipMemcpyDeviceToHost) != hipSuccess || e) ok = false;
    (void)hipMemset(P.dev.err, 0, sizeof e);
    P.on = false;
    P.dev.spin_limit = saved.spin_limit;
    (void)hipFree(buf);
    (void)hipGetLastError();
    return ok;
}

int ppo_dist_peer_attach(ppo_handle* h, const char* handles) {
    ENTER(h);
    ppo_handle::Peer& P = h->peer;
    if (!h->comm || !P.region) return fail(h, "ppo_dist_peer_attach: call ppo_dist_init and ppo_dist_peer_export first");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    P.on = false;
    bool mapped = true;
    for (int r = 0; r < h->world; ++r) {
        if (r == h->rank) continue;
        if (P.mapped[r]) { (void)hipIpcCloseMemHandle(P.mapped[r]); P.mapped[r] = nullptr; }
        hipIpcMemHandle_t ipc;
        memcpy(&ipc, handles + (size_t)r * 64, 64);
        if (hipIpcOpenMemHandle(&P.mapped[r], ipc, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); P.mapped[r] = nullptr; mapped = false; }
    }
    // collective reset: every rank zeroes its own sequence / arrival / error words and its own flag block, then all ranks meet
    // before anybody pushes -- a probe that failed on one rank in an earlier attach (that rank skipped a round) must not leave
    // the sequence numbers diverged for this one
    {
        HIP_OK(h, hipMemset(P.local, 0, 64));
        HIP_OK(h, hipMemset(P.region, 0, kPeerFlagBytes));
        HIP_OK(h, hipDeviceSynchronize());
        float* tok = nullptr;
        HIP_OK(h, hipMalloc((void**)&tok, sizeof(float)));
        const bool met = hipMemset(tok, 0, sizeof(float)) == hipSuccess && h->rccl.AllReduce(tok, tok, 1, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream) == 0 &&
                         hipStreamSynchronize(h->stream) == hipSuccess;
        (void)hipFree(tok);
        if (!met) return fail(h, "ppo_dist_peer_attach: the ranks could not meet before the probe");
    }
    PeerDev d{};
    for (int r = 0; r < h->world; ++r) {
        char* base = (char*)(r == h->rank ? P.region : P.mapped[r]);
        if (!base) base = (char*)P.region;                         // unmapped peer: the probe fails, nothing is ever sent there afterwards
        d.flags[r] = (unsigned*)base;
        d.slots[r] = (float*)(base + kPeerFlagBytes);
        d.sslots[r] = d.slots[r] + (size_t)2 * h->world * P.cap;
    }
    d.seq = P.local; d.arrive = P.local + 1; d.err = P.local + 2; d.sseq = P.local + 4; d.scap = P.scap;
    d.cap = P.cap; d.world = h->world; d.rank = h->rank;
    const char* tm = getenv("PPO_HIP_PEER_TIMEOUT_MS");
    const double ms = tm ? atof(tm) : 10000.0;
    d.spin_limit = (unsigned)std::min(4.0e9, std::max(1000.0, ms * 2000.0));       // ~0.5 us per wait iteration
    P.dev = d;
    const char* en = getenv("PPO_HIP_PEER_REDUCE");
    const bool wanted = !(en && en[0] == '0');
    // the verdict must be COMMON: a rank that could not map a peer, or whose probe failed, takes everybody back to RCCL
    bool ok = wanted && mapped && !P.coarse;
    if (wanted) ok = peer_probe(h) && ok;
    // My slots go back to zero before I join the agreement below: the probe's patterns must not stay in elements that a later collective leaves
    // unwritten (the padding of the parameter vector, which weight_grad_assemble_kernel<.., PEER>'s tile pushes never touch and adam_kernel<.., 2>
    // adds up with everything else).  Nobody writes into my slots between my probe's last sum and that agreement: every peer's next push comes
    // after IT has passed the agreement, which needs me.
    HIP_OK(h, hipMemset((char*)P.region + kPeerFlagBytes, 0, (size_t)2 * h->world * P.cap * sizeof(float)));
    HIP_OK(h, hipDeviceSynchronize());
    float* flag = nullptr;
    HIP_OK(h, hipMalloc((void**)&flag, sizeof(float)));
    const float mine = ok ? 1.f : 0.f;
    float got = 0.f;
    const bool agreed = hipMemcpy(flag, &mine, sizeof mine, hipMemcpyHostToDevice) == hipSuccess &&
                        h->rccl.AllReduce(flag, flag, 1, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream) == 0 &&
                        hipStreamSynchronize(h->stream) == hipSuccess && hipMemcpy(&got, flag, sizeof got, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(flag);
    if (!agreed) return fail(h, "ppo_dist_peer_attach: the ranks could not agree on the probe result");
    P.on = got == (float)h->world;
    P.usable = P.on;
    // No reset of seq / flags here: after a probe that passed everywhere every rank stands at the same sequence number, and a
    // rank zeroing its flags now could erase the first flag of a peer that has already left this call.
    if (wanted && !P.on) fprintf(stderr, "libppo_hip: peer all-reduce probe failed on some rank (this rank: %s); using RCCL\n", ok ? "ok" : "failed");
    return 0;
}

int ppo_dist_peer_active(const ppo_handle* h) { return h->peer.on ? 1 : 0; }

int ppo_dist_global_shuffle(ppo_handle* h, int on) {
    ENTER_Q(h);
    // refused at toggle time, not at the first ppo_update: without a communicator the flag would be silently ignored, and the
    // literal scheme needs ncclAllGather (with ONE rank the global permutation is the local one: accepted, nothing to gather)
    if (on && !h->comm) return fail(h, "ppo_dist_global_shuffle: call ppo_dist_init first (no communicator)");
    if (on && h->world > 1 && !h->rccl.AllGather) return fail(h, "ppo_dist_global_shuffle: the collective library has no ncclAllGather");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->global_shuffle = on != 0;
    return 0;
}

int ppo_dist_bucketed(ppo_handle* h, int on) {
    ENTER(h);
    HIP_OK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->bf.bucketed = on != 0;
    h->bf.bucketed_any_world = on == 2;                         // (2: also under a ONE-rank communicator -- tools/peer_overhead.py measures what the per-layer launches cost one rank)
    return 0;
}

int ppo_dist_peer_enable(ppo_handle* h, int on) {
    ENTER(h);
    if (on && !h->peer.usable) return fail(h, "ppo_dist_peer_enable: not attached, or the probe failed");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->peer.on = on != 0;
    return 0;
}

// ---- measurement ----------------------------------------------------------------------------------------------------------
int ppo_prof_enable(ppo_handle* h, int on) {
    ENTER(h);
    prof_collect(h);
    h->prof = on != 0;
    for (int i = 0; i < PK_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
    return 0;
}

int ppo_prof_read(ppo_handle* h, int max, char names[][32], double* total_ms, int64_t* launches) {
    ENTER(h);
    prof_collect(h);
    int n = 0;
    for (int i = 0; i < PK_COUNT && n < max; ++i) {
        snprintf(names[n], 32, "%s", kProfNames[i]);
        total_ms[n] = h->prof_ms[i];
        launches[n] = h->prof_n[i];
        ++n;
    }
    return n;
}

int ppo_kernel_counts(ppo_handle* h, int max, char names[][32], int64_t* enqueued) {
    int n = 0;
    for (int i = 0; i < KV_COUNT && n < max; ++i) { snprintf(names[n], 32, "%s", kVariantNames[i]); enqueued[n] = h->kv[i]; ++n; }
    return n;
}

// Raw device buffers by name, PADDING INCLUDED (ppo_get_flat and ppo_get_last_grad copy the dense part of every tensor only): the transposed mirrors, the small-parameter
// mirrors, the assembled gradient with its tail, the workspaces of the last train step, the gathered epoch.  For bitwise comparisons of two runs (tests/test_other_shapes.py,
// the interleaved-handles test's report); not part of the reference's interface.  *count = the buffer's length in 4-byte words; at most max_count are copied.
int ppo_debug_buffer(ppo_handle* h, const char* name, float* dst, int64_t max_count, int64_t* count) {
    ENTER_Q(h);
    if (bf16_materialize_grad(h)) return -1;
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const void* p = nullptr; size_t words = 0; bool found = false;
    for (const DbgEnt& e : debug_table(h)) if (!strcmp(e.name, name)) { found = true; p = e.p; words = e.words; }
    if (!found) return fail(h, "ppo_debug_buffer: no buffer named '%s'", name);
    if (!p || !words) { *count = 0; return 0; }               // this handle's shape does not use the buffer
    *count = (int64_t)words;
    const size_t c = std::min<size_t>(words, (size_t)std::max<int64_t>(max_count, 0));
    if (c) HIP_OK(h, hipMemcpy(dst, p, c * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// Raise the error word of the bf16 path's chained launch as a failed hand-off would (a row group spread over two XCDs, or a time-out): the NEXT call that chains its
// layers must report it -- ppo_step / ppo_value / ppo_act_deterministic repeat their pass with a launch per layer, the rollout calls and ppo_update return the error --
// and the handle launches layer by layer from then on (tests/test_bf16_path.py).  Returns -1 when the handle has no chained launch to fail.
int ppo_debug_raise_chain_error(ppo_handle* h) {
    ENTER(h);
    ppo_handle::Bf16& b = h->bf;
    if (!b.on || !b.chain || !b.chain_words[0]) return fail(h, "ppo_debug_raise_chain_error: this handle does not chain its layers");
    HIP_OK(h, hipStreamSynchronize(h->stream));
    const unsigned one = 1u;
    HIP_OK(h, hipMemcpy(b.chain_words[0] + (size_t)GB_CHAIN_SHAPES * GB_CHAIN_WORDS, &one, sizeof one, hipMemcpyHostToDevice));
    return 0;
}

// What the update's captured graph is made of: counts[0] kernel nodes, [1] memset nodes, [2] memcpy nodes, [3] anything else; returns -1 when no graph is held (eager
// handles, or before the first ppo_update).  The rule since round 6 is "kernel nodes only" (zero_words above): tests/test_race_guards.py holds every shape to it.
int ppo_debug_graph_nodes(ppo_handle* h, int32_t counts[4]) {
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (!h->upd_graph_tmpl) return -1;
    size_t n = 0;
    HIP_OK(h, hipGraphGetNodes(h->upd_graph_tmpl, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) HIP_OK(h, hipGraphGetNodes(h->upd_graph_tmpl, nodes.data(), &n));
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        HIP_OK(h, hipGraphNodeGetType(nodes[i], &ty));
        ++counts[ty == hipGraphNodeTypeKernel ? 0 : ty == hipGraphNodeTypeMemset ? 1 : ty == hipGraphNodeTypeMemcpy ? 2 : 3];
    }
    return 0;
}

// Fill the LDS of every CU with one word (and leave it there): a kernel whose result depends on LDS it never wrote -- what the previous workgroup on its CU left behind --
// shows it as soon as that word is a NaN pattern.  Every workgroup takes a CU's whole 160 KB and holds it ~30 us, so the launch's first 256 workgroups land one per CU;
// two more rounds behind them for CUs the dispatcher served late.  tests/test_race_guards.py (results must not depend on what the LDS held); no reference counterpart.
__global__ void __launch_bounds__(256) debug_poison_lds_kernel(unsigned word, unsigned* sink) {
    extern __shared__ unsigned poison_lds[];
    for (int i = threadIdx.x; i < 160 * 256; i += 256) poison_lds[i] = word;
    __syncthreads();
    for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(127);
    if (poison_lds[(threadIdx.x * 131) % (160 * 256)] != word) sink[0] = 1u;       // (keeps the stores)
}
int ppo_debug_poison_lds(ppo_handle* h, uint32_t word) {
    ENTER_Q(h);
    static bool attr_set = false;
    if (!attr_set) { HIP_OK(h, hipFuncSetAttribute((const void*)debug_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    hipDeviceProp_t prop;
    HIP_OK(h, hipGetDeviceProperties(&prop, h->device));
    hipLaunchKernelGGL(debug_poison_lds_kernel, dim3(3 * prop.multiProcessorCount), dim3(256), 160 * 1024, h->stream, word, reinterpret_cast<unsigned*>(h->norm_out));
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipStreamSynchronize(h->stream));
    return 0;
}

#ifdef PPO_STAMPS
int ppo_debug_read_stamps(ppo_handle* h, unsigned long long* dst, int n) {
    HIP_OK(h, hipStreamSynchronize(h->stream));
    // n > 0: the train kernels' area; n < 0: the second area (weight gradients); n < -(1 << 20): the third (statistics kernel), count = -n - (1 << 20)
    // ... n < -(2 << 20): the fourth (adam_kernel), count = -n - (2 << 20); n < -(3 << 20): the fifth (policy_step_kernel), count = -n - (3 << 20)
    const bool fifth = n < -(3 << 20), fourth = !fifth && n < -(2 << 20), third = !fifth && !fourth && n < -(1 << 20);
    const int cnt = fifth ? -n - (3 << 20) : fourth ? -n - (2 << 20) : third ? -n - (1 << 20) : (n < 0 ? -n : n);
    HIP_OK(h, hipMemcpy(dst, g_stamps + (fifth ? 4096 * 44 : fourth ? 4096 * 40 : third ? 4096 * 32 : (n < 0 ? 4096 * 16 : 0)), (size_t)cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

}  // extern "C"
