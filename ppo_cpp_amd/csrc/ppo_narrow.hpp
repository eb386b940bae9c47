// ppo_narrow.hpp -- dedicated kernels for NARROW networks (every hidden width <= 64: the reference's real shape
// [64,64], ppo2.cpp:114; BASELINE configs[1] and [3]).  gfx950 only.
//
// At these widths a whole tower (both directions: W_l and the transposed copies the backward pass multiplies by) is
// 45 KB: it is copied into LDS ONCE per workgroup with every load issued before the first LDS store (one memory round
// trip), and every matrix instruction of the step then takes its B operand from LDS.  Nothing streams, nothing waits on
// memory between layers, and the activations / gradients a row tile produces never leave LDS, so the weight gradients are
// formed in the same workgroup (dW = X^T dY over its 32 rows, 8 k-steps of v_mfma_f32_16x16x4_f32 per 16x16 tile) and
// leave as ONE partial gradient vector per workgroup.  The four-kernel sequence of the wide nets (A: forward/backward with
// X_l / dY_l workspaces, B: weight gradients, C1: slabs + slots, C2: Adam) becomes: narrow_train_kernel ->
// narrow_reduce_kernel -> adam_kernel.  Same arithmetic, same operation order per row as train_fwd_bwd_kernel (exact fp32
// MFMA, k-ordered); the reduction over row tiles is again a fixed-order sum, so results are bitwise reproducible.
//
// Reference arithmetic replaced: G:6889-23699 (train forward, loss, backward incl. the .../MatMul_grad/MatMul_1 and
// .../Add_grad/Sum_1 nodes), G:1859-6866 (act model), ppo2/policies.hpp:33-77.
#pragma once

#include "ppo_kernels.hpp"
#include "ppo_peer.hpp"      // system-scope word load / store helpers

#define NW_MAXL 4                  // hidden layers the narrow path accepts
#define NW_PIPES 2                 // 16-row tiles per workgroup (4 waves each); one pipe per workgroup (256 workgroups at M = 2048) measured
                                   // slower: 7.5e6 against 7.9e6 env-steps/s at BASELINE configs[3] (twice the partial vectors to add up)
#define NW_THREADS (256 * NW_PIPES)
#define NW_ROWS (16 * NW_PIPES)
#define NW_WPAD 16                 // LDS row padding of the weight images: B-operand reads of the 4 k-groups hit disjoint banks
#define NW_XPAD 4                  // row padding of the activation tiles

// LDS layout.  The weight image (identical layout for both towers) is kept PACKED in global memory by adam_kernel /
// transpose_refresh_kernel (GradSrc::i_off ...), so the prologue is one flat copy: [forward matrices | policy head |
// small parameters | transposed matrices | transposed head].  The act kernel copies the first `w_fwd` floats only.
struct NwLayout {
    int wf[NW_MAXL], wf_ld[NW_MAXL];         // forward weights  [K_l][Hp_l + pad]
    int wt[NW_MAXL], wt_ld[NW_MAXL];         // transposed copies W_l^T [Hp_l][Hp_{l-1} + pad], l >= 1
    int wh, wh_ld, wht, wht_ld;              // policy head [HpL][Ap + pad] and its transpose [Ap][HpL + pad]
    int par;                                  // small parameters, indexed with NetDev::par_* offsets
    int w_fwd;                                // floats the act kernel needs (multiple of 4)
    int w_total;                              // floats of the whole image (multiple of 4) = image stride between the towers
    // per-pipe tiles (offsets relative to the pipe's base)
    int x[NW_MAXL + 1]; int ldx[NW_MAXL + 1]; // x[0] = input tile, x[l+1] = h_{l+1}
    int dy[NW_MAXL]; int ldy[NW_MAXL];        // dLoss/d(pre-activation of layer l)
    int mu, ldm, dmu, acts, dls, rowv, misc;
    int pipe_total;
    int lds_total;                            // floats: w_total + NW_PIPES * pipe_total
};

struct NwTrainArgs {
    const float* img;              // [2][w_total] packed weight images
    const float* obs; const float* actions; const float* advs; const float* returns; const float* old_values; const float* old_neglogp;
    const float* hyper;            // {lr, cliprange}
    int n; float inv_n;
    float* partials;               // [2 towers][n_groups][part_stride]; a workgroup writes its tower's tensors + 8 tail floats
    int n_groups; int part_stride;
    unsigned long long* stamps;    // diagnostic builds only (-DPPO_STAMPS): [workgroups][32] cycle stamps
};
#ifdef PPO_STAMPS
#define NSTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

// Compile-time shape of a kernel instantiation: <KP0, HP, AP, L> (every hidden layer HP wide); all zero = runtime shape
// (any widths <= 64 that are multiples of 16).  With constants every loop below unrolls, the LDS reads of a product are
// all issued before its first matrix instruction and tile offsets fold into immediates.
template <int KP0_, int HP_, int AP_, int L_>
struct NwShape {
    static constexpr bool fixed = L_ > 0;
    __device__ static __forceinline__ int L(const NetDev& n) { return fixed ? L_ : n.L; }
    __device__ static __forceinline__ int Kp0(const NetDev& n) { return fixed ? KP0_ : n.Kp0; }
    __device__ static __forceinline__ int Ap(const NetDev& n) { return fixed ? AP_ : n.Ap; }
    __device__ static __forceinline__ int Hp(const NetDev& n, int l) { return fixed ? HP_ : n.Hp[l]; }
};

// Y[16 x Np] (this pipe) = X[16 x K] * W[K x Np], W in LDS; wave w of the pipe takes the 16-column tiles w, w+4, ...
// acc register r of lane (g = lane >> 4, c = lane & 15) = Y[4g + r][n0 + c].  CK > 0: compile-time depth, every operand
// read issued first, four interleaved accumulator chains (the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 is
// hidden and the result is ((c0 + c1) + (c2 + c3)): fp32 reassociation of the k-ordered sum, nothing else).
template <int CK, class Ep>
__device__ __forceinline__ void nw_dense(const float* Xs, int ldx, int K, const float* Ws, int ldw, int Np, Ep&& ep, const int tidx = (int)threadIdx.x) {
    // (tidx: the thread index; narrow_epoch_kernel passes a per-iteration opaque copy so that the address arithmetic is not hoisted out of its minibatch loop)
    const int lane = tidx & 63, w = uni((tidx >> 6) & 3);            // (wave-uniform: said explicitly, an opaque tidx hides it)
    const int g = lane >> 4, c = lane & 15;
    for (int n0 = 16 * w; n0 < Np; n0 += 64) {
        const float* xp = Xs + c * ldx + g;
        const float* wp = Ws + g * ldw + n0 + c;
        f32x4 acc;
        if constexpr (CK > 0) {
            float xa[CK / 4], wb[CK / 4];
#pragma unroll
            for (int s = 0; s < CK / 4; ++s) { xa[s] = xp[4 * s]; wb[s] = wp[4 * s * ldw]; }
            f32x4 ch[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ch[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < CK / 4; ++s) ch[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wb[s], ch[s & 3], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (ch[0][r] + ch[1][r]) + (ch[2][r] + ch[3][r]);
        } else {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            int kb = 0;
            for (; kb + 8 <= K; kb += 8) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[kb], wp[kb * ldw], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[kb + 4], wp[(kb + 4) * ldw], a1, 0, 0, 0);
            }
            for (; kb < K; kb += 4) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[kb], wp[kb * ldw], a0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = a0[r] + a1[r];
        }
        ep(acc, g, n0 + c);
    }
}

// block prologue: the tower's packed weight image (flat copy) + this workgroup's rows, every global load issued before the
// first LDS store: one memory round trip
template <class S>
__device__ __forceinline__ void nw_stage(const NetDev& net, const NwLayout& lay, const float* __restrict__ img, int n_img, float* lds,
                                         const float* __restrict__ obs, int row0, int nrows, ObsNorm nz, float* __restrict__ obs_out, int tower,
                                         const float* __restrict__ actions, const float* __restrict__ v0, const float* __restrict__ v1, int mode, const int tidx = (int)threadIdx.x) {
    const int tid = tidx;
    constexpr int WV = 16 / NW_PIPES;                       // float4 loads per thread: 64 KB in flight covers the image
    float4 wv[WV];
    const int n4 = n_img / 4;
#pragma unroll
    for (int k = 0; k < WV; ++k) {
        const int e = tid + NW_THREADS * k;
        wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < n4) wv[k] = reinterpret_cast<const float4*>(img)[e];
    }
    constexpr int OV = 2;
    float ov[OV], av[OV], r0 = 0.f, r1 = 0.f;
    const int O = net.O, A = net.A, Kp0 = S::Kp0(net), Ap = S::Ap(net);
#pragma unroll
    for (int k = 0; k < OV; ++k) {
        const int i = tid + NW_THREADS * k;
        const int r = i / Kp0, j = i - r * Kp0, row = row0 + r;
        ov[k] = 0.f;
        if (i < NW_ROWS * Kp0 && row < nrows && j < O) ov[k] = obs[(size_t)row * O + j];
    }
    if (mode == 1) {
#pragma unroll
        for (int k = 0; k < OV; ++k) {
            const int i = tid + NW_THREADS * k;
            const int r = i / Ap, j = i - r * Ap, row = row0 + r;
            av[k] = 0.f;
            if (i < NW_ROWS * Ap && row < nrows && j < A) av[k] = actions[(size_t)row * A + j];
        }
    }
    if (mode && tid < NW_ROWS && row0 + tid < nrows) { r0 = v0[row0 + tid]; r1 = v1[row0 + tid]; }
    // ---- consume ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < WV; ++k) {
        const int e = tid + NW_THREADS * k;
        if (e < n4) reinterpret_cast<float4*>(lds)[e] = wv[k];
    }
    for (int e = tid + NW_THREADS * WV; e < n4; e += NW_THREADS) reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(img)[e];
    auto put_obs = [&](int i, float x) __attribute__((always_inline)) {
        const int r = i / Kp0, j = i - r * Kp0, row = row0 + r;
        if (row < nrows && j < O) {
            if (nz.enabled) {                                // env_normalize.hpp:99-104
                x = (x - nz.mean[j]) * (1.0f / sqrtf(nz.var[j] + nz.eps));
                x = tf_min(tf_max(x, -nz.clip), nz.clip);
            }
            if (obs_out && tower == 0) obs_out[(size_t)row * O + j] = x;
        }
        lds[lay.w_total + (r >> 4) * lay.pipe_total + lay.x[0] + (r & 15) * lay.ldx[0] + j] = x;
    };
#pragma unroll
    for (int k = 0; k < OV; ++k) { const int i = tid + NW_THREADS * k; if (i < NW_ROWS * Kp0) put_obs(i, ov[k]); }
    for (int i = tid + NW_THREADS * OV; i < NW_ROWS * Kp0; i += NW_THREADS) {
        const int r = i / Kp0, j = i - r * Kp0, row = row0 + r;
        put_obs(i, (row < nrows && j < O) ? obs[(size_t)row * O + j] : 0.f);
    }
    if (mode == 1) {
        auto put_act = [&](int i, float x) __attribute__((always_inline)) {
            const int r = i / Ap, j = i - r * Ap;
            lds[lay.w_total + (r >> 4) * lay.pipe_total + lay.acts + (r & 15) * Ap + j] = x;
        };
#pragma unroll
        for (int k = 0; k < OV; ++k) { const int i = tid + NW_THREADS * k; if (i < NW_ROWS * Ap) put_act(i, av[k]); }
        for (int i = tid + NW_THREADS * OV; i < NW_ROWS * Ap; i += NW_THREADS) {
            const int r = i / Ap, j = i - r * Ap, row = row0 + r;
            put_act(i, (row < nrows && j < A) ? actions[(size_t)row * A + j] : 0.f);
        }
    }
    if (mode && tid < NW_ROWS) {
        const bool live = row0 + tid < nrows;
        float* rv = lds + lay.w_total + (tid >> 4) * lay.pipe_total + lay.rowv;
        rv[2 * (tid & 15)] = live ? r0 : 0.f;
        rv[2 * (tid & 15) + 1] = live ? r1 : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Deferred Adam (reference shape only: 18/18 -> 32, [64,64]).  Inside ppo_update the clip + Adam launch of train step k does
// not run as a kernel of its own: the train kernel of step k+1 applies it in its prologue.  Every workgroup loads its tower's
// 8.4 K parameters with their moments and the assembled gradient of step k (one memory round trip, requested together with
// the rows), derives the global norm from the per-chunk sums of squares in adam_kernel's order, applies clip_by_global_norm +
// ApplyAdam with adam_kernel's expression (bit-identical results) and writes the UPDATED weights straight into its LDS image,
// forward and transposed copies.  The parameter / moment vectors ping-pong between two sets (read set i, write set i^1: no
// workgroup reads what another one writes); workgroup g of a tower writes back every n_groups-th 16-byte piece.  One launch
// and one dependent round trip per train step less (21.8 -> 18 us at M = 2048); the last step of an update runs adam_kernel
// itself, which also refreshes the packed image the act kernels read.
// ------------------------------------------------------------------------------------------------------------------------
struct NwLazyArgs {
    const float* grad; const float* parts; int n_parts;          // assembled gradient (+ tail at n_theta) and its sums of squares
    const float* th_in; const float* m_in; const float* v_in;
    float* th_out; float* m_out; float* v_out;
    float* beta_pow;                                             // {cur b1, cur b2, next b1, next b2}
    float beta1, beta2, eps, max_norm;
    float* loss_row; float* norm_out;
};

// pieces per thread: 5 behind a 32-column observation tile, 6 behind a 64-column one (the first-layer matrix [64][64] is two 512-piece blocks)
template <int KP0> struct NwLazyN { static constexpr int N = KP0 == 64 ? 6 : 5; };
template <int NP> struct NwLazyRegs { float4 g[NP], m[NP], v[NP], t[NP]; float part, b1p, b2p, lr; };

// Thread -> 16-byte piece.  Pieces 0,1 = second-layer matrix [64][64], 2 = first-layer matrix rows 0..31 ([32][64]), 3 = policy head matrix
// [64][32], 4 = one float4 of a vector (threads 0..48), 5 (64-column observation tile only: the 36-observation hexapod,
// env/hexapod_closed_loop_env.hpp:20) = first-layer matrix rows 32..63.  Matrix pieces in memory order (fully coalesced loads: the prologue is
// bound by the bytes it pulls through the L1, 160 KB per workgroup; a row-spread dealing that would make the transposed LDS
// copy conflict-free costs more in the loads than it saves in LDS).  The transposed copy (element (r, c) -> [c][r], row pitch
// = 16 mod 64 banks) is written with lane-rotated elements -- the j-th write of a lane stores its element (j + c4) & 3 -- which
// brings the bank conflicts from 16-way to 4-way.
struct NwPiece { int off; int r, c; };       // element offset in the padded parameter vector (-1: none); row / first column of the float4

__device__ __forceinline__ NwPiece nw_lazy_piece(const NetDev& net, int tower, int k, int tid) {
    NwPiece p{-1, 0, 0};
    if (k < 3) {
        const int idx = k < 2 ? tid + 512 * k : tid;              // W1: 1024 float4 (16 per row) ; W0: 512 float4 (16 per row)
        const int c4 = idx & 15, r = idx >> 4 & 63;
        if (k < 2) { p.r = r; p.c = 4 * c4; p.off = net.w_off[tower][1] + 64 * r + 4 * c4; }
        else { p.r = idx >> 4; p.c = 4 * (idx & 15); p.off = net.w_off[tower][0] + 4 * idx; }
        return p;
    }
    if (k == 3) {
        if (tower != 0) return p;
        p.r = tid >> 3; p.c = 4 * (tid & 7); p.off = net.wmu_off + 4 * tid;    // [64][32]: 8 float4 per row
        return p;
    }
    if (k == 5) {                                                               // W0 rows 32..63 of a [64][64] first layer
        const int idx = tid + 512;
        p.r = idx >> 4; p.c = 4 * (idx & 15); p.off = net.w_off[tower][0] + 4 * idx;
        return p;
    }
    if (tid < 16) p.off = net.b_off[tower][0] + 4 * tid;
    else if (tid < 32) p.off = net.b_off[tower][1] + 4 * (tid - 16);
    else if (tower == 0) { if (tid < 40) p.off = net.bmu_off + 4 * (tid - 32); else if (tid < 48) p.off = net.ls_off + 4 * (tid - 40); }
    else if (tid < 48) p.off = net.wv_off + 4 * (tid - 32);
    else if (tid == 48) p.off = net.bv_off;
    return p;
}

template <int NP>
__device__ __forceinline__ void nw_lazy_issue(const NetDev& net, const NwLazyArgs& z, const float* hyper, int tower, NwLazyRegs<NP>& R) {
    const int tid = threadIdx.x;
    R.b1p = z.beta_pow[0]; R.b2p = z.beta_pow[1]; R.lr = hyper[0];             // requested with everything else: one round trip
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int off = nw_lazy_piece(net, tower, k, tid).off;
        R.g[k] = R.m[k] = R.v[k] = R.t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (off >= 0) {
            R.g[k] = *reinterpret_cast<const float4*>(z.grad + off);
            R.m[k] = *reinterpret_cast<const float4*>(z.m_in + off);
            R.v[k] = *reinterpret_cast<const float4*>(z.v_in + off);
            R.t[k] = *reinterpret_cast<const float4*>(z.th_in + off);
        }
    }
    // the norm partials: adam_kernel's order (thread t adds parts[t], parts[t + 256], ...); the first one is only REQUESTED here
    // (an addition would wait for it: a whole memory round trip before the rows are even asked for)
    R.part = (tid < 256 && tid < z.n_parts) ? z.parts[tid] : 0.f;
}

// `red`: 4 floats of LDS scratch.  Ends with the image complete in LDS (caller synchronises).
// RESIDENT (narrow_epoch_kernel): the moments and the weights stay in R from step to step (R.m / R.v / R.t receive the results), the pieces go back to
// memory only when `write_back` says so (the epoch's last step), and the caller keeps the powers, the loss row and the norm.
// EXACT (the default since round 6; PPO_HIP_ADAM_FAST=1 selects the other instantiation): correctly rounded square root and division instead of the 1-ulp instructions -- a template parameter, not a run-time flag: with both
// sequences in one kernel the default form lost 6 % (measured).
template <int NP, bool RESIDENT = false, bool EXACT = false>
__device__ __forceinline__ void nw_lazy_apply(const NetDev& net, const NwLayout& lay, const NwLazyArgs& z, int tower, int grp, int n_groups,
                                              NwLazyRegs<NP>& R, float* lds, float* red, unsigned long long* st = nullptr, bool write_back = true, float* norm_ret = nullptr,
                                              const int tidx = (int)threadIdx.x) {
    const int tid = tidx;
#ifdef PPO_STAMPS
#define LSTAMP(i) do { if (st && tid == 0) st[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif
    const float b1p = R.b1p, b2p = R.b2p, lr = R.lr;
    float s = 0.f + R.part;
    if (tid < 256) for (int i = tid + 256; i < z.n_parts; i += 256) s += z.parts[i];
    s = wave_sum_lane0(s);
    if (tid < 256 && (tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    float scale = z.max_norm * tf_min(1.0f / norm, 1.0f / z.max_norm);          // G:24289-24472
    if (!isfinite(norm)) scale = __builtin_nanf("");                            // G:24493-24543
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    LSTAMP(20);
    float* par = lds + lay.par;
    // write-back ownership: workgroup g of the tower stores the pieces [g * per, (g + 1) * per) of each 512-piece block (one
    // wave-uniform division per kernel; a per-piece modulo by the runtime group count is ~40 vector instructions each)
    const int per = (512 + n_groups - 1) / n_groups, own_lo = grp * per, own_hi = own_lo + per;
    const bool own = tid >= own_lo && tid < own_hi;
    // transposed copy of one float4: the j-th write stores element (j + rot) & 3, rot = the lane's float4 column (see above)
    auto put_t = [&](float* base, int ld, int r, int c, const float (&to)[4], int rot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (j + rot) & 3;
            const float x = i == 0 ? to[0] : i == 1 ? to[1] : i == 2 ? to[2] : to[3];
            base[(c + i) * ld + r] = x;
        }
    };
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const NwPiece pc = nw_lazy_piece(net, tower, k, tid);
        if (pc.off < 0) continue;
        const float gv[4] = {R.g[k].x, R.g[k].y, R.g[k].z, R.g[k].w}, mv[4] = {R.m[k].x, R.m[k].y, R.m[k].z, R.m[k].w};
        const float vv[4] = {R.v[k].x, R.v[k].y, R.v[k].z, R.v[k].w}, tv[4] = {R.t[k].x, R.t[k].y, R.t[k].z, R.t[k].w};
        float mo[4], vo[4], to[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) adam_element<!EXACT>(gv[i] * scale, mv[i], vv[i], tv[i], 1.0f - z.beta1, 1.0f - z.beta2, alpha, z.eps, mo[i], vo[i], to[i]);
        const float4 t4 = make_float4(to[0], to[1], to[2], to[3]);
        bool mine;                                                              // who writes this piece back
        if (k < 2) {                                                            // W1 [64][64]: forward + transposed copies
            *reinterpret_cast<float4*>(lds + lay.wf[1] + pc.r * lay.wf_ld[1] + pc.c) = t4;
            put_t(lds + lay.wt[1], lay.wt_ld[1], pc.r, pc.c, to, tid & 3);
            mine = own;
        } else if (k == 2 || k == 5) {                                          // W0 rows 0..31 / 32..63 (forward copy only: no dX of the first layer)
            *reinterpret_cast<float4*>(lds + lay.wf[0] + pc.r * lay.wf_ld[0] + pc.c) = t4;
            mine = own;
        } else if (k == 3) {                                                    // W_mu [64][32]
            *reinterpret_cast<float4*>(lds + lay.wh + pc.r * lay.wh_ld + pc.c) = t4;
            put_t(lds + lay.wht, lay.wht_ld, pc.r, pc.c, to, tid & 3);
            mine = own;
        } else {
            int po;
            if (tid < 16) po = net.par_b[0] + 4 * tid;
            else if (tid < 32) po = net.par_b[1] + 4 * (tid - 16);
            else if (tower == 0) po = tid < 40 ? net.par_bmu + 4 * (tid - 32) : net.par_ls + 4 * (tid - 40);
            else po = tid < 48 ? net.par_wv + 4 * (tid - 32) : net.par_bv;
            *reinterpret_cast<float4*>(par + po) = t4;                          // (par_bv: 4 floats reserved; the 3 behind b_v are padding zeros)
            mine = grp == 0;
        }
        if constexpr (RESIDENT) { R.t[k] = t4; R.m[k] = make_float4(mo[0], mo[1], mo[2], mo[3]); R.v[k] = make_float4(vo[0], vo[1], vo[2], vo[3]); }
        if (mine && write_back) {
            *reinterpret_cast<float4*>(z.th_out + pc.off) = t4;
            *reinterpret_cast<float4*>(z.m_out + pc.off) = make_float4(mo[0], mo[1], mo[2], mo[3]);
            *reinterpret_cast<float4*>(z.v_out + pc.off) = make_float4(vo[0], vo[1], vo[2], vo[3]);
        }
    }
    LSTAMP(21);
    if constexpr (RESIDENT) { if (norm_ret) *norm_ret = norm; return; }
    if (tower == 0 && grp == 0) {                                               // adam_kernel's block 0
        if (tid == 0) {
            z.beta_pow[2] = b1p * z.beta1;                                      // G:31217-31342
            z.beta_pow[3] = b2p * z.beta2;
            if (z.norm_out) *z.norm_out = norm;
        }
        if (z.loss_row && tid < 5) {
            const float* tail = z.grad + net.n_theta;
            const float n = tail[5];
            float r = tail[tid] / n;
            if (tid == 1 || tid == 3) r = 0.5f * r;
            z.loss_row[tid] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Train step, first launch: forward + loss + backward + weight gradients of 32 rows of ONE tower (blockIdx.y).
// ------------------------------------------------------------------------------------------------------------------------
// The train step of one workgroup behind its prologue (weight image and this workgroup's rows in LDS): forward, loss, backward, weight gradients ->
// one partial gradient vector.  Called by narrow_train_kernel and, once per minibatch, by narrow_epoch_kernel.
template <int KP0, int HP, int AP, int LL, bool WT = true>
__device__ __forceinline__ void nw_train_body(const NetDev& net, const NwLayout& lay, const NwTrainArgs& a, float* lds, const int tower, const int grp,
                                              const int tidx = (int)threadIdx.x) {
    typedef NwShape<KP0, HP, AP, LL> S;
    const int tid = tidx, pipe = uni(tid >> 8), ptid = tid & 255;
    const int row0 = grp * NW_ROWS;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    __syncthreads();
    NSTAMP(1);
    float* P = lds + lay.w_total + pipe * lay.pipe_total;            // this pipe's tiles
    const float* par = lds + lay.par;
    // ---- forward (G:6889-9187) -----------------------------------------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const float* bias = par + net.par_b[l];
        float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
        auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) Ys[(4 * g + r) * ldy + col] = fast_tanh(acc[r] + b);
        };
        if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep, tid);
        else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep, tid);
        __syncthreads();
        NSTAMP(2 + l);
    }
    const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
    const float cr = a.hyper[1];
    const int r = ptid >> 4, part = ptid & 15;
    const int row = row0 + 16 * pipe + r;
    const bool live = row < a.n;
    float* misc = P + lay.misc;
    float* dls = P + lay.dls; float* acts = P + lay.acts; float* rowv = P + lay.rowv;
    float* dYtop = P + lay.dy[L - 1]; const int ldt = lay.ldy[L - 1];
    if (tower == 0) {
        // ---- policy head + surrogate loss (G:9428-11290) and its gradient (G:12609-22656) ----------------------------
        float* mus = P + lay.mu; const int ldm = lay.ldm;
        nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = par[net.par_bmu + col];
#pragma unroll
            for (int q = 0; q < 4; ++q) mus[(4 * g + q) * ldm + col] = acc[q] + b;
        }, tid);
        __syncthreads();
        NSTAMP(6);
        // every lane keeps its (at most four: A <= 64) elements' z and sigma from the first pass: the second pass used to recompute both
        // (an exp and a division per element) -- same values, same results
        float ssq = 0.f, slog = 0.f, sent = 0.f;
        float zk[4], sk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                 // (compile-time k: the kept values stay in registers)
            const int j = part + 16 * k;
            zk[k] = 0.f; sk[k] = 1.f;
            if (j < net.A) {
                const float mu = mus[r * ldm + j];
                const float logstd = mu * 0.0f + par[net.par_ls + j];
                const float act = live ? acts[r * Ap + j] : mu;
                const float sigma = expf(logstd);
                const float z = (act - mu) / sigma;
                zk[k] = z; sk[k] = sigma;
                ssq += z * z; slog += logstd; sent += logstd + HALF_LOG_2PIE;
            }
        }
        ssq = group16_sum(ssq); slog = group16_sum(slog); sent = group16_sum(sent);
        const float nlp = 0.5f * ssq + HALF_LOG_2PI * (float)net.A + slog;
        const float adv = live ? rowv[2 * r] : 0.f;
        const float old_nlp = live ? rowv[2 * r + 1] : nlp;
        const float lo = 1.0f - cr, hi = 1.0f + cr;
        const float ratio = expf(old_nlp - nlp);
        const float rmin = tf_min(ratio, hi);
        const float rclip = tf_max(rmin, lo);
        const float m1 = -adv * ratio, m2 = -adv * rclip;
        const float gg = a.inv_n;
        const float sel = (m1 >= m2) ? 1.0f : 0.0f;                                       // G:12609
        const float pass = ((rmin >= lo) ? 1.0f : 0.0f) * ((ratio <= hi) ? 1.0f : 0.0f);  // G:15357, 16113
        float d_ratio = (-adv) * gg * sel;
        d_ratio += (-adv) * gg * (1.0f - sel) * pass;
        const float d_nlp = live ? -(d_ratio * ratio) : 0.0f;
        if (part == 0) {
            const float dk = nlp - old_nlp;
            misc[r * 4 + 0] = live ? tf_max(m1, m2) : 0.f;
            misc[r * 4 + 1] = live ? sent : 0.f;
            misc[r * 4 + 2] = live ? dk * dk : 0.f;
            misc[r * 4 + 3] = (live && fabsf(ratio - 1.0f) > cr) ? 1.0f : 0.f;
        }
        float* dmu_t = P + lay.dmu;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = part + 16 * k;
            if (j < Ap) {
                float dmu = 0.f, dl = 0.f;
                if (j < net.A && live) {
                    const float z = zk[k], sigma = sk[k];
                    dl = d_nlp * (1.0f - z * z) - net.ent_coef * gg;                          // AddN_2 G:21299
                    dmu = d_nlp * (-(z / sigma)) + dl * 0.0f;                                 // AddN_3 G:22656
                }
                dmu_t[r * ldm + j] = dmu;
                dls[r * Ap + j] = dl;
            }
        }
        __syncthreads();
        NSTAMP(7);
        // dh_L = (dmu * W_mu^T) .* (1 - h_L^2)
        nw_dense<AP>(dmu_t, ldm, Ap, lds + lay.wht, lay.wht_ld, HpL, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float h = hL[(4 * g + q) * ldh + col]; dYtop[(4 * g + q) * ldt + col] = acc[q] * (1.0f - h * h); }
        }, tid);
    } else {
        // ---- value head + clipped value loss (G:10213-10837) and its gradient (G:14975-19571) ------------------------
        const float* wv = par + net.par_wv;
        float s = 0.f;
        for (int k = part; k < HpL; k += 16) s = fmaf(hL[r * ldh + k], wv[k], s);
        s = group16_sum(s);
        const float v = s + par[net.par_bv];
        float dv = 0.f, lossv = 0.f;
        if (live) {
            const float R = rowv[2 * r], vo = rowv[2 * r + 1];
            const float dvo = v - vo;
            const float vmin = tf_min(dvo, cr);
            const float vclip = vo + tf_max(vmin, -cr);
            const float e1 = v - R, e2 = vclip - R;
            const float s1 = e1 * e1, s2 = e2 * e2;
            lossv = tf_max(s1, s2);
            const float gv = net.vf_coef * 0.5f * a.inv_n;
            const float selv = (s1 >= s2) ? 1.0f : 0.0f;                                       // G:14975
            const float passv = ((vmin >= -cr) ? 1.0f : 0.0f) * ((dvo <= cr) ? 1.0f : 0.0f);   // G:17477, 18071
            dv = gv * selv * (2.0f * e1) + gv * (1.0f - selv) * (2.0f * e2) * passv;           // AddN_1 G:19571
        }
        if (part == 0) { misc[r] = dv; misc[16 + r] = lossv; }
        __syncthreads();
        for (int i = ptid; i < 16 * HpL; i += 256) {                  // dh_L = dv (x) w_v .* (1 - h_L^2)
            const int q = i / HpL, k = i - q * HpL;
            const float h = hL[q * ldh + k];
            dYtop[q * ldt + k] = (misc[q] * wv[k]) * (1.0f - h * h);
        }
    }
    __syncthreads();
    NSTAMP(8);
    // ---- hidden layers, top down ---------------------------------------------------------------------------------------
    for (int l = L - 1; l >= 1; --l) {
        const float* hl = P + lay.x[l]; const int ldhl = lay.ldx[l];
        float* dn = P + lay.dy[l - 1]; const int ldn = lay.ldy[l - 1];
        nw_dense<HP>(P + lay.dy[l], lay.ldy[l], S::Hp(net, l), lds + lay.wt[l], lay.wt_ld[l], S::Hp(net, l - 1),
                     [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
#pragma unroll
                         for (int q = 0; q < 4; ++q) { const float h = hl[(4 * g + q) * ldhl + col]; dn[(4 * g + q) * ldn + col] = acc[q] * (1.0f - h * h); }
                     }, tid);
        __syncthreads();
    }
    NSTAMP(9);
    // ---- gradients of this workgroup's 32 rows: one partial vector ------------------------------------------------------
    float* out = a.partials + ((size_t)tower * a.n_groups + grp) * a.part_stride;
    // partial-vector stores are agent-scope WRITE-THROUGH: the next kernel reads them on other XCDs, and as ordinary stores they
    // sit dirty in this XCD's L2 until the end-of-kernel write-back (26 KB per workgroup, 16 workgroups per XCD): 20.5 -> 19.5 us
    // per train step at M = 2048
    // (WT = false, narrow_epoch_kernel's XCD-local form: the readers share this XCD's L2 -- ordinary stores)
    auto pst = [](float* p, float v) __attribute__((always_inline)) { if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; };
    const float* PB = lds + lay.w_total;                              // pipe q's tiles at PB + q * pipe_total
    const int lane = tid & 63, wave = uni(tid >> 6), g = lane >> 4, c = lane & 15;
    constexpr int NWV = NW_THREADS / 64;
    // dW = X^T dY: tile (i0, j0), reduction over the NW_ROWS rows: k-step s covers rows 4s .. 4s+3 (pipe = s >> 2).  A wave
    // walks tiles t = wave, wave + 8, ... of the concatenated tile list of all matrices (balanced: 32 tiles at [64,64]).
    auto dw_mat = [&](int xoff, int ldx_, int Kd, int yoff, int ldy_, int Nd, int out_off, int ldo, int first_tile) __attribute__((always_inline)) {
        const int tj = Nd / 16, nt = (Kd / 16) * tj;
        for (int t = (wave + NWV - (first_tile % NWV)) % NWV; t < nt; t += NWV) {
            const int i0 = (t / tj) * 16, j0 = (t % tj) * 16;
            float xa[4 * NW_PIPES], yb[4 * NW_PIPES];
#pragma unroll
            for (int s = 0; s < 4 * NW_PIPES; ++s) {
                const float* p = PB + (s >> 2) * lay.pipe_total;
                const int m = 4 * (s & 3) + g;
                xa[s] = p[xoff + m * ldx_ + i0 + c]; yb[s] = p[yoff + m * ldy_ + j0 + c];
            }
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4 * NW_PIPES; s += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], yb[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s + 1], yb[s + 1], a1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pst(out + (out_off + (i0 + 4 * g + q) * ldo + j0 + c), a0[q] + a1[q]);
        }
        return first_tile + nt;
    };
    int ft = 0;
    for (int l = 0; l < L; ++l)
        ft = dw_mat(lay.x[l], lay.ldx[l], l ? S::Hp(net, l - 1) : Kp0, lay.dy[l], lay.ldy[l], S::Hp(net, l), net.w_off[tower][l], S::Hp(net, l), ft);
    if (tower == 0) ft = dw_mat(lay.x[L], lay.ldx[L], HpL, lay.dmu, lay.ldm, Ap, net.wmu_off, Ap, ft);
    NSTAMP(10);
    // vectors: one per WAVE, lanes = elements, the NW_ROWS rows added in index order (fixed order, all reads independent)
    auto colsum = [&](int off, int ld_, int j) __attribute__((always_inline)) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NW_ROWS; ++q) s += PB[(q >> 4) * lay.pipe_total + off + (q & 15) * ld_ + j];
        return s;
    };
    const int vec = (wave + NWV - (ft % NWV)) % NWV;                  // continue the round robin after the matrix tiles
    if (vec < L) { if (lane < S::Hp(net, vec)) pst(out + (net.b_off[tower][vec] + lane), colsum(lay.dy[vec], lay.ldy[vec], lane)); }
    if (tower == 0) {
        if (vec == L) { if (lane < Ap) pst(out + (net.bmu_off + lane), colsum(lay.dmu, lay.ldm, lane)); }
        if (vec == L + 1) { if (lane < Ap) pst(out + (net.ls_off + lane), colsum(lay.dls, Ap, lane)); }
        if (vec == L + 2 && lane < 4) {                               // pg, entropy, kl, clipfrac sums
            const float s4 = colsum(lay.misc, 4, lane);
            pst(out + (net.n_theta + lane), s4);
        }
    } else {
        if (vec == L && lane < HpL) {                                 // dW_v[k] = sum_rows h_L[row,k] * dv[row]
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NW_ROWS; ++q) { const float* pq = PB + (q >> 4) * lay.pipe_total; s = fmaf(pq[lay.x[L] + (q & 15) * ldh + lane], pq[lay.misc + (q & 15)], s); }
            pst(out + (net.wv_off + lane), s);
        }
        if (vec == L + 1 && lane < 2) {                               // lane 0: db_v = sum dv ; lane 1: sum of max((v-R)^2, (vclip-R)^2)
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NW_ROWS; ++q) s += PB[(q >> 4) * lay.pipe_total + lay.misc + 16 * lane + (q & 15)];
            pst(out + (lane == 0 ? net.bv_off : net.n_theta), s);
        }
    }
    NSTAMP(11);
}

template <int KP0, int HP, int AP, int LL, bool LAZY = false, bool EXACT = false>
__global__ __launch_bounds__(NW_THREADS) void narrow_train_kernel(NetDev net, NwLayout lay, NwTrainArgs a, NwLazyArgs z) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(NwTrainArgs) + sizeof(NwLazyArgs)>();
    const int tower = blockIdx.y, grp = blockIdx.x;
    const int row0 = grp * NW_ROWS;
    NSTAMP(0);
    if constexpr (LAZY) {
        static_assert((KP0 == 32 || KP0 == 64) && HP == 64 && AP == 32 && LL == 2, "deferred Adam: the reference's shapes only (18 or 36 observations)");
        float* lazy_red = lds + lay.w_total + lay.misc;      // 4 floats of pipe 0's loss scratch (no static LDS: the launch may ask for all 160 KB)
        NwLazyRegs<NwLazyN<KP0>::N> R;
        nw_lazy_issue(net, z, a.hyper, tower, R);
        NSTAMP(12);
        nw_stage<S>(net, lay, nullptr, 0, lds, a.obs, row0, a.n, ObsNorm{nullptr, nullptr, 0.f, 0.f, 0}, nullptr, tower,
                    a.actions, tower == 0 ? a.advs : a.returns, tower == 0 ? a.old_neglogp : a.old_values, tower == 0 ? 1 : 2);
        NSTAMP(13);
        nw_lazy_apply<NwLazyN<KP0>::N, false, EXACT>(net, lay, z, tower, grp, (int)gridDim.x, R, lds, lazy_red
#ifdef PPO_STAMPS
                      , a.stamps ? a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr
#endif
                      );
        NSTAMP(14);
    } else {
        nw_stage<S>(net, lay, a.img + (size_t)tower * lay.w_total, lay.w_total, lds, a.obs, row0, a.n, ObsNorm{nullptr, nullptr, 0.f, 0.f, 0}, nullptr, tower,
                    a.actions, tower == 0 ? a.advs : a.returns, tower == 0 ? a.old_neglogp : a.old_values, tower == 0 ? 1 : 2);
    }
    nw_train_body<KP0, HP, AP, LL>(net, lay, a, lds, tower, grp);
}

// ------------------------------------------------------------------------------------------------------------------------
// Gradient assembly for the narrow path: 64 consecutive elements per block, 4 threads per element (each adds a quarter of
// the row groups in index order, the four meet in a fixed order).  Emits the sum of squares per 64-element chunk for the
// global norm (adam_kernel sums these `norm_parts`) and, in the last block, the loss tail.
// ------------------------------------------------------------------------------------------------------------------------
struct NwReduceArgs {
    const GradSrc* src; int n_chunks;          // chunks of 64 covering P_pad ; block n_chunks = loss tail
    const float* partials; int n_groups; int part_stride; int n_theta;
    float* grad; float* sumsq; float n_local; float* beta_pow;
};

__global__ __launch_bounds__(256) void narrow_reduce_kernel(NwReduceArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (blk == a.n_chunks) {
        // tail = {pg, vf, ent, kl, cf, rows}: policy tower holds pg, ent, kl, cf at n_theta + 0..3, value tower vf at n_theta
        if (tid < 5 * 32) {
            const int q = tid >> 5, ln = tid & 31;
            const int tower = (q == 1) ? 1 : 0;
            const int off = a.n_theta + (q <= 1 ? 0 : q - 1);
            float s = 0.f;
            for (int gidx = ln; gidx < a.n_groups; gidx += 32) s += a.partials[((size_t)tower * a.n_groups + gidx) * a.part_stride + off];
            s = half_sum_lane0(s);
            if (ln == 0) a.grad[a.n_theta + q] = s;
        }
        if (tid == 160) a.grad[a.n_theta + 5] = a.n_local;
        if (tid == 161) { a.beta_pow[0] = a.beta_pow[2]; a.beta_pow[1] = a.beta_pow[3]; }
        return;
    }
    const int e = tid >> 2, sub = tid & 3;
    const int idx = blk * 64 + e;
    const GradSrc s = a.src[idx >> 8];
    float sum = 0.f;
    if (s.kind != 2) {
        const float* p = a.partials + (size_t)s.tower * a.n_groups * a.part_stride + idx;
        const int per = (a.n_groups + 3) / 4, g0 = sub * per, g1 = min(a.n_groups, g0 + per);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int gi = g0;
        for (; gi + 4 <= g1; gi += 4) {
            s0 += p[(size_t)gi * a.part_stride]; s1 += p[(size_t)(gi + 1) * a.part_stride];
            s2 += p[(size_t)(gi + 2) * a.part_stride]; s3 += p[(size_t)(gi + 3) * a.part_stride];
        }
        for (; gi < g1; ++gi) s0 += p[(size_t)gi * a.part_stride];
        sum = (s0 + s1) + (s2 + s3);
    }
    sum += __shfl_xor(sum, 1);                  // (x0 + x1) + (x2 + x3): identical in all four lanes
    sum += __shfl_xor(sum, 2);
    if (sub == 0) a.grad[idx] = sum;
    float q = (sub == 0) ? sum * sum : 0.f;
    q = wave_sum_lane0(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) a.sumsq[blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------------------------------
// ALL minibatches of an epoch in ONE launch, for minibatches of <= 64 rows on the deferred-Adam shapes (the reference's own command line: 1 environment x
// 2048 steps, 32 minibatches of 64 rows, ppo2.cpp:114-128): 2 G workgroups (G = row groups of 32, 2 towers) stay resident, keep their tower's weight image in
// LDS and its Adam moments and weights in registers, and per minibatch run nw_train_body, publish their partial gradient vector (write-through, two buffers
// by step parity), MEET (one word per workgroup), add the partial vectors up themselves in narrow_reduce_kernel's order -- both towers': the global norm needs
// every chunk's sum of squares, and reading the other tower's 2 x 26 KB is cheaper than a second meeting -- and apply clip + Adam with nw_lazy_apply's code to
// the LDS image.  Same arithmetic in the same order as narrow_train_kernel<.., LAZY> + narrow_reduce_kernel per step: bit-identical
// (tests/test_other_shapes.py), without 2 launches and 2 dependent prologues per step.  (Round 3's resident form put a whole tower on ONE CU and was issue-
// bound, profiles/r03_d_*; this one keeps the launch form's 2 CUs per tower.)  At exit the owners write weights, moments and the packed image back.
// ------------------------------------------------------------------------------------------------------------------------
#define NW_EPOCH_MAX_G 2
#define NW_EPOCH_WORDS 16                // meeting table: [2 G] step counters ... [NW_EPOCH_BASE] steps completed by earlier launches, [NW_EPOCH_WORDS - 1] raised when a wait timed out
#define NW_EPOCH_BASE 8                  // (ONE base for all workgroups: a word of its own per workgroup breaks when the number of row groups changes between launches)
struct NwEpochArgs {
    const float* obs; const float* actions; const float* advs; const float* returns; const float* old_values; const float* old_neglogp;   // minibatch k = rows [k M, (k + 1) M)
    int M, nmb; float inv_n;
    float* img;                          // [2][w_total]: read at entry, written back at exit
    float* partials; int part_stride;    // [nmb steps (XL) | 2 step parities][2 towers][G][part_stride]
    float* theta; float* m; float* v;    // in place
    float* grad;                         // the LAST step's assembled gradient + tail (ppo_get_last_grad)
    const float* hyper; float* beta_pow; float beta1, beta2, eps, max_norm;
    float* loss_rows;                    // [nmb][5]
    float* norm_out;
    unsigned* words;
    int n_chunks;                        // 64-element chunks of the padded parameter vector
    unsigned long long* stamps;          // diagnostic builds only
};

// XL (the default): the 2 G workgroups are the launch's workgroups 0, 8, 16, ... -- dealt to ONE XCD (workgroup b runs on XCD b % 8; the others leave at once) --
// so a partial vector written with ordinary stores (complete in that XCD's L2 once acknowledged: s_waitcnt vmcnt(0)) is what the other workgroups read through the
// same L2 with ordinary loads, at L2 latency and bandwidth instead of a write-through trip to the memory side (4.4 k -> ~1.6 k cycles for the 104 KB a workgroup
// reads per step).  Every step has its own partial buffers: no address is read twice in a launch, so no CU's L1 holds a stale line.  The placement is CHECKED (every
// word carries its writer's hardware XCC id, as in gemm_chain_bf16_kernel); a mismatch raises error word 2 and the handle goes back to the write-through form.
template <int KP0, bool XL, bool EXACT = false>
__global__ __launch_bounds__(NW_THREADS) void narrow_epoch_kernel(NetDev net, NwLayout lay, NwEpochArgs e) {
    constexpr int HP = 64, AP = 32, LL = 2, NP = NwLazyN<KP0>::N;
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (XL && (blockIdx.x & 7u) != 0) return;
    const int G = XL ? (int)(gridDim.x >> 4) : (int)gridDim.x, nwg = 2 * G;
    const int wid = XL ? (int)(blockIdx.x >> 3) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int tower = wid / G, grp = wid - tower * G;
    const int tid0 = threadIdx.x;
    const int row0 = grp * NW_ROWS;
    unsigned xcc = 0;
    if constexpr (XL) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u; }
    // steps completed by earlier launches: written by workgroup 0 at its exit only (behind the last meeting, so every workgroup of THIS launch has read it by then); a
    // workgroup's word holds (absolute step count << 4) | XCC id, compared modulo 2^28
    const unsigned e0 = __hip_atomic_load(e.words + NW_EPOCH_BASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x0FFFFFFFu;
    NwLazyRegs<NP> R;
    R.b1p = e.beta_pow[2]; R.b2p = e.beta_pow[3]; R.lr = e.hyper[0];           // `next` = the powers the first step applies (the launch form copies them to `cur` first)
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int off = nw_lazy_piece(net, tower, k, tid0).off;
        R.g[k] = R.m[k] = R.v[k] = R.t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (off >= 0) { R.m[k] = *reinterpret_cast<const float4*>(e.m + off); R.v[k] = *reinterpret_cast<const float4*>(e.v + off); R.t[k] = *reinterpret_cast<const float4*>(e.theta + off); }
    }
    const float* v0 = tower == 0 ? e.advs : e.returns; const float* v1 = tower == 0 ? e.old_neglogp : e.old_values;
    nw_stage<S>(net, lay, e.img + (size_t)tower * lay.w_total, lay.w_total, lds, e.obs, row0, e.M, ObsNorm{nullptr, nullptr, 0.f, 0.f, 0}, nullptr, tower,
                e.actions, v0, v1, tower == 0 ? 1 : 2);
    float* parts = lds + lay.w_total + lay.dy[0];            // [512] chunk sums of squares: over pipe 0's dY tile, dead between a step's last product and the next step
    float* red = lds + lay.w_total + lay.misc;               // 4 floats of pipe 0's loss scratch (not touched by the row staging)
    NwTrainArgs ta{};
    ta.hyper = e.hyper; ta.n = e.M; ta.inv_n = e.inv_n; ta.n_groups = G; ta.part_stride = e.part_stride; ta.stamps = e.stamps;
    NwLazyArgs z{};
    z.n_parts = e.n_chunks; z.parts = parts; z.th_out = e.theta; z.m_out = e.m; z.v_out = e.v; z.beta1 = e.beta1; z.beta2 = e.beta2; z.eps = e.eps; z.max_norm = e.max_norm;
    float norm = 0.f;
    // the pieces' offsets of both towers, once (inside the loop the thread index is opaque and they would be recomputed every step)
    int offs[2][NP];
#pragma unroll
    for (int side = 0; side < 2; ++side)
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) offs[side][kk] = nw_lazy_piece(net, side == 0 ? tower : 1 - tower, kk, tid0).off;
#ifdef PPO_STAMPS
#define EPSTAMP(i) do { if (e.stamps && threadIdx.x == 0 && k == e.nmb - 2) e.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define EPSTAMP(i) do { } while (0)
#endif
    for (int k = 0; k < e.nmb; ++k) {
        // an OPAQUE copy of the thread index per iteration: with the plain one the compiler hoists every thread-dependent LDS address of the step out of this
        // loop (hundreds of live registers: 256 VGPRs + 190 spilled, measured)
        int tid = tid0, lane = tid0 & 63;
        asm volatile("" : "+v"(tid), "+v"(lane));
        __syncthreads();                                      // image (first step: the copy; later: Adam's writes) and this minibatch's rows are in LDS
        const bool last = k == e.nmb - 1;
        // the NEXT minibatch's rows are requested now and wait in registers under this step (nw_stage's loads, element for element; 2.5 k cycles of exposed
        // latency when requested after the step)
        constexpr int OVN = NW_ROWS * KP0 / NW_THREADS, AVN = NW_ROWS * AP / NW_THREADS;
        float nxo[OVN], nxa[AVN], nx0 = 0.f, nx1 = 0.f;
        {
            const size_t ro = (size_t)(k + 1) * e.M;
#pragma unroll
            for (int q = 0; q < OVN; ++q) {
                const int i = tid + NW_THREADS * q, r = i / KP0, j = i - r * KP0, row = row0 + r;
                nxo[q] = (!last && row < e.M && j < net.O) ? e.obs[(ro + row) * net.O + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < AVN; ++q) {
                const int i = tid + NW_THREADS * q, r = i / AP, j = i - r * AP, row = row0 + r;
                nxa[q] = (!last && tower == 0 && row < e.M && j < net.A) ? e.actions[(ro + row) * net.A + j] : 0.f;
            }
            if (!last && tid < NW_ROWS && row0 + tid < e.M) { nx0 = v0[ro + row0 + tid]; nx1 = v1[ro + row0 + tid]; }
        }
        float* pb = e.partials + (size_t)(XL ? k : (k & 1)) * nwg * e.part_stride;
        ta.partials = pb;
        ta.stamps = (k == e.nmb - 2) ? e.stamps : nullptr;
        EPSTAMP(0);
        nw_train_body<KP0, HP, AP, LL, !XL>(net, lay, ta, lds, tower, grp, tid);
        // ---- arrival: the partial vector was stored write-through; every wave drains, one word says "step k of this workgroup is out" ---------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        EPSTAMP(16);
        const unsigned target = (e0 + (unsigned)k + 1u) & 0x0FFFFFFFu;
        if (tid == 0) __hip_atomic_store(e.words + wid, (target << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        parts[tid] = 0.f;                                     // (NW_THREADS = 512 entries; chunks without a piece -- alignment gaps -- stay zero)
        if (tid < 64) {
            unsigned polls = 0;
            for (;;) {
                const unsigned w = lane < nwg ? __hip_atomic_load(e.words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((target << 4) | xcc);
                if (__all((((w >> 4) - target) & 0x0FFFFFFFu) < 0x08000000u)) {
                    if (XL && (w & 15u) != xcc) __hip_atomic_store(e.words + NW_EPOCH_WORDS - 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1u << 22)) { if (lane == 0) __hip_atomic_store(e.words + NW_EPOCH_WORDS - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        EPSTAMP(17);
        // ---- assembly: (p0 + p1) + (0 + 0) per element as narrow_reduce_kernel adds two row groups; per 64-element chunk the sum of squares in ITS tree --------
        // (16 consecutive threads hold a chunk's 16 pieces: levels 1-2 across pieces c ^ 2, c ^ 1 = elements e ^ 8, e ^ 4; levels 3-4 inside the piece; then the
        // kernel's four waves as piece groups: (r0 + r1) + (r2 + r3))
        // every load first -- both towers' pieces of the partial vectors (write-through loads: ~2 k cycles to the memory side and back) and, behind them, the
        // NEXT minibatch's rows (nw_stage: its wait covers them all) -- then the arithmetic
        f32x4 p0[2][NP], p1[2][NP];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int t2 = side == 0 ? tower : 1 - tower;
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                const float* src = pb + (size_t)t2 * G * e.part_stride + (offs[side][kk] >= 0 ? offs[side][kk] : 0);
                if constexpr (XL) {
                    const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + (G > 1 ? e.part_stride : 0));
                    p0[side][kk] = (f32x4){x0.x, x0.y, x0.z, x0.w}; p1[side][kk] = (f32x4){x1.x, x1.y, x1.z, x1.w};
                } else {
                    p0[side][kk] = nb_ld4_sc1(src);
                    p1[side][kk] = nb_ld4_sc1(src + (G > 1 ? e.part_stride : 0));
                }
            }
        }
        if (!last) {                                          // the next minibatch's rows: registers -> the tiles nw_stage fills (the step's last reads of them are done)
#pragma unroll
            for (int q = 0; q < OVN; ++q) {
                const int i = tid + NW_THREADS * q, r = i / KP0, j = i - r * KP0;
                lds[lay.w_total + (r >> 4) * lay.pipe_total + lay.x[0] + (r & 15) * lay.ldx[0] + j] = nxo[q];
            }
            if (tower == 0) {
#pragma unroll
                for (int q = 0; q < AVN; ++q) {
                    const int i = tid + NW_THREADS * q, r = i / AP, j = i - r * AP;
                    lds[lay.w_total + (r >> 4) * lay.pipe_total + lay.acts + (r & 15) * AP + j] = nxa[q];
                }
            }
            if (tid < NW_ROWS) {
                float* rv = lds + lay.w_total + (tid >> 4) * lay.pipe_total + lay.rowv;
                rv[2 * (tid & 15)] = nx0; rv[2 * (tid & 15) + 1] = nx1;
            }
        }
        if constexpr (!XL)
#pragma unroll
        for (int side = 0; side < 2; ++side) {   // (the loads are inline asm: the compiler does not count them)
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0;
            if constexpr (NP == 5) { nb_wait8(p0[side][0], p0[side][1], p0[side][2], p0[side][3], p0[side][4], d0, d1, d2); nb_wait8(p1[side][0], p1[side][1], p1[side][2], p1[side][3], p1[side][4], d0, d1, d2); }
            else { nb_wait8(p0[side][0], p0[side][1], p0[side][2], p0[side][3], p0[side][4], p0[side][5], d0, d1); nb_wait8(p1[side][0], p1[side][1], p1[side][2], p1[side][3], p1[side][4], p1[side][5], d0, d1); }
        }
        EPSTAMP(23);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int t2 = side == 0 ? tower : 1 - tower;
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                const bool has = offs[side][kk] >= 0;
                float g4[4], q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a0 = has ? p0[side][kk][i] : 0.f, a1 = (has && G > 1) ? p1[side][kk][i] : 0.f;
                    g4[i] = (a0 + a1) + (0.f + 0.f);
                    q[i] = g4[i] * g4[i];
                    q[i] += dpp_move<0x4E>(q[i]);            // lane ^ 2 (quad_perm [2,3,0,1]): one VALU instruction each (a __shfl_xor is a trip through the LDS crossbar:
                    q[i] += dpp_move<0xB1>(q[i]);            // lane ^ 1 (quad_perm [1,0,3,2])   100 of them per step and wave made this phase 8 k cycles)
                }
                if (side == 0) R.g[kk] = make_float4(g4[0], g4[1], g4[2], g4[3]);
                const float r = (q[0] + q[2]) + (q[1] + q[3]);
                const float t = r + dpp_move<0x141>(r);      // r is uniform over 4 lanes: row_half_mirror (lane i <-> 7 - i) reaches the other group of 4
                const float t8 = dpp_move<0x140>(t);         // t is uniform over 8 lanes: row_mirror (i <-> 15 - i) reaches the other half
                const bool half_chunk = kk == 4 && t2 == 0 && tid >= 32 && tid < 48;          // b_mu | logstd: 8 pieces each, neighbours in the lanes but not in memory
                const float u = t + (half_chunk ? 0.f : t8);
                if (has && ((tid & 15) == 0 || (half_chunk && tid == 40))) parts[offs[side][kk] >> 6] = u;
            }
        }
        // the loss sums {pg, vf, ent, kl, cf} of the minibatch: narrow_reduce_kernel's 32-lane tree over the row groups = (p0 + p2) + (p1 + p3)
        float tail = 0.f;
        if (wid == 0 && tid < 5) {
            const int tw = tid == 1 ? 1 : 0, off = net.n_theta + (tid <= 1 ? 0 : tid - 1);
            const float* src = pb + (size_t)tw * G * e.part_stride + off;
            float a0, a1 = 0.f;
            if constexpr (XL) { a0 = src[0]; if (G > 1) a1 = src[e.part_stride]; }               // (XL: the producers' ordinary stores may still be dirty in the shared L2)
            else { a0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (G > 1) a1 = __hip_atomic_load(src + e.part_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            tail = (a0 + 0.f) + (a1 + 0.f);
        }
        __syncthreads();                                      // parts complete
        EPSTAMP(18);
        R.part = (tid < 256 && tid < e.n_chunks) ? parts[tid] : 0.f;
        nw_lazy_apply<NP, true, EXACT>(net, lay, z, tower, grp, G, R, lds, red, nullptr, last, &norm, tid);
        EPSTAMP(19);
        if (wid == 0 && tid < 5) {
            float r = tail / (float)e.M;
            if (tid == 1 || tid == 3) r = 0.5f * r;           // vf_loss, approxkl carry the 0.5
            e.loss_rows[(size_t)k * 5 + tid] = r;
            if (last) e.grad[net.n_theta + tid] = tail;
        }
        if (last && grp == 0) {                               // the assembled gradient of the last step (what the launch form leaves in `grad`)
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) { const int off = nw_lazy_piece(net, tower, kk, tid).off; if (off >= 0) *reinterpret_cast<float4*>(e.grad + off) = R.g[kk]; }
        }
        if (!last) { R.b1p = R.b1p * e.beta1; R.b2p = R.b2p * e.beta2; }                         // G:31217-31342: the next step's powers
        EPSTAMP(22);
    }
    __syncthreads();
    const int tid = tid0;
    // ---- exit: the packed image of this tower (its LDS copy IS the layout), the powers, the norm --------------------------------------------------------
    if (grp == 0) {
        float* img = e.img + (size_t)tower * lay.w_total;
        for (int i = tid; i < lay.w_total / 4; i += NW_THREADS) reinterpret_cast<float4*>(img)[i] = reinterpret_cast<const float4*>(lds)[i];
    }
    if (wid == 0 && tid == 0) {
        __hip_atomic_store(e.words + NW_EPOCH_BASE, (e0 + (unsigned)e.nmb) & 0x0FFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e.beta_pow[0] = R.b1p; e.beta_pow[1] = R.b2p;                                              // cur = what the last step applied
        e.beta_pow[2] = R.b1p * e.beta1; e.beta_pow[3] = R.b2p * e.beta2;
        if (e.norm_out) *e.norm_out = norm;
        e.grad[net.n_theta + 5] = (float)e.M;
    }
}

// (Round 5 also had this resident epoch for LARGER minibatches -- narrow_epoch_dist_kernel, the assembly dealt over up to 128 workgroups with two meetings per
// minibatch -- opt-in and measured slower than the launches at configs[3]: 22.2 vs 19.0 us per step, profiles/r05_i_narrow_epoch_kernel.txt.  Removed in round 6;
// branch experiments-r05.)
// ------------------------------------------------------------------------------------------------------------------------
// Act model for narrow nets: the same forward on 32 rows per workgroup and tower, weights from the packed image.
// `img` rides in StepArgs::theta (the narrow launch passes the image instead of the padded parameter vector).
// ------------------------------------------------------------------------------------------------------------------------
template <int KP0, int HP, int AP, int LL>
__global__ __launch_bounds__(NW_THREADS) void narrow_step_kernel(NetDev net, NwLayout lay, StepArgs a) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(StepArgs)>();
    const int tower = blockIdx.y;
    if (tower == 1 && !a.value) return;
    if (tower == 0 && !a.action && !a.det_action && !a.neglogp && !a.obs_out) return;
    const int tid = threadIdx.x, pipe = tid >> 8, ptid = tid & 255;
    const int row0 = blockIdx.x * NW_ROWS;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    nw_stage<S>(net, lay, a.theta + (size_t)tower * lay.w_total, lay.w_fwd, lds, a.obs, row0, a.n, a.nz, a.obs_out, tower, nullptr, nullptr, nullptr, 0);
    __syncthreads();
    float* P = lds + lay.w_total + pipe * lay.pipe_total;
    const float* par = lds + lay.par;
    for (int l = 0; l < L; ++l) {
        const float* bias = par + net.par_b[l];
        float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
        auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) Ys[(4 * g + r) * ldy + col] = fast_tanh(acc[r] + b);
        };
        if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep);
        else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep);
        __syncthreads();
    }
    const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
    const int r = ptid >> 4, part = ptid & 15;
    const int row = row0 + 16 * pipe + r;
    if (tower == 1) {
        const float* wv = par + net.par_wv;
        float s = 0.f;
        for (int k = part; k < HpL; k += 16) s = fmaf(hL[r * ldh + k], wv[k], s);
        s = group16_sum(s);
        if (part == 0 && row < a.n) a.value[row] = s + par[net.par_bv];
        return;
    }
    float* mus = P + lay.mu; const int ldm = lay.ldm;
    nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
        const float b = par[net.par_bmu + col];
#pragma unroll
        for (int q = 0; q < 4; ++q) mus[(4 * g + q) * ldm + col] = acc[q] + b;
    });
    __syncthreads();
    float ssq = 0.f, slog = 0.f;
    for (int j = part; j < net.A; j += 16) {
        const float mu = mus[r * ldm + j];
        const float logstd = mu * 0.0f + par[net.par_ls + j];
        const float sigma = expf(logstd);
        float eps = 0.f;
        if (row < a.n) eps = a.noise ? a.noise[(size_t)row * net.A + j] : ctr_normal(a.seed, a.row_base + row, a.rng_step, j);
        const float act = mu + sigma * eps;
        const float z = (act - mu) / sigma;
        ssq += z * z; slog += logstd;
        if (row < a.n) {
            if (a.action) a.action[(size_t)row * net.A + j] = act;
            if (a.det_action) a.det_action[(size_t)row * net.A + j] = mu;
        }
    }
    ssq = group16_sum(ssq); slog = group16_sum(slog);
    if (part == 0 && row < a.n && a.neglogp) a.neglogp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)net.A + slog;
}

// ------------------------------------------------------------------------------------------------------------------------
// Fused collect step for small environment counts (n_envs <= 32: BASELINE configs[1], one environment) against the
// on-device seeded env: ONE launch per env step instead of three (policy step, env, running statistics).
//   both workgroups : act on the CURRENT raw observations (state S_in: normalise with S_in's statistics, forward, tower 0
//                     samples and writes rollout row t, tower 1 writes the values)
//   tower-0 workgroup, afterwards: the env transition (counter hash, step `env_step`), EnvNormalize::step's statistics
//                     (obs_rms.update, ret = ret*gamma + r, ret_rms.update, reward scale + clip, ret *= 1 - done;
//                     env/env_normalize.hpp:64-116, common/running_statistics.hpp:26-104) -> state S_out
// S_in is only read and S_out only written during a launch (the host alternates two state sets), so the two workgroups
// never race.  The batch moments are the reference's two passes over the E rows, one thread per column.
// ------------------------------------------------------------------------------------------------------------------------
struct NwEnvState { float* raw_obs; float* obs_mean; float* obs_var; double* obs_count; float* ret_mean; float* ret_var; double* ret_count; float* ret; float* done; };
struct NwCollectArgs {
    NwEnvState in, out;
    uint32_t seed, env_step; int env0;
    float gamma, clip_rew, eps; int norm_obs, norm_rew;
    float* rew_out;              // rollout rewards row t
    float* done_row;             // rollout dones row t (the flags that arrived with obs_t)
};

template <int KP0, int HP, int AP, int LL>
__global__ __launch_bounds__(NW_THREADS) void narrow_collect_kernel(NetDev net, NwLayout lay, StepArgs a, NwCollectArgs c) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(StepArgs) + sizeof(NwCollectArgs)>();
    const int tower = blockIdx.y;
    const int tid = threadIdx.x, pipe = tid >> 8, ptid = tid & 255;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    const int E = a.n, O = net.O;
    nw_stage<S>(net, lay, a.theta + (size_t)tower * lay.w_total, lay.w_fwd, lds, a.obs, 0, E, a.nz, a.obs_out, tower, nullptr, nullptr, nullptr, 0);
    __syncthreads();
    float* P = lds + lay.w_total + pipe * lay.pipe_total;
    const float* par = lds + lay.par;
    for (int l = 0; l < L; ++l) {
        const float* bias = par + net.par_b[l];
        float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
        auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) Ys[(4 * g + r) * ldy + col] = fast_tanh(acc[r] + b);
        };
        if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep);
        else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep);
        __syncthreads();
    }
    const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
    const int r = ptid >> 4, part = ptid & 15;
    const int row = 16 * pipe + r;
    if (tower == 1) {
        const float* wv = par + net.par_wv;
        float s = 0.f;
        for (int k = part; k < HpL; k += 16) s = fmaf(hL[r * ldh + k], wv[k], s);
        s = group16_sum(s);
        if (part == 0 && row < E) a.value[row] = s + par[net.par_bv];
        return;
    }
    float* mus = P + lay.mu; const int ldm = lay.ldm;
    nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
        const float b = par[net.par_bmu + col];
#pragma unroll
        for (int q = 0; q < 4; ++q) mus[(4 * g + q) * ldm + col] = acc[q] + b;
    });
    __syncthreads();
    float ssq = 0.f, slog = 0.f;
    for (int j = part; j < net.A; j += 16) {
        const float mu = mus[r * ldm + j];
        const float logstd = mu * 0.0f + par[net.par_ls + j];
        const float sigma = expf(logstd);
        float eps = 0.f;
        if (row < E) eps = a.noise ? a.noise[(size_t)row * net.A + j] : ctr_normal(a.seed, a.row_base + row, a.rng_step, j);
        const float act = mu + sigma * eps;
        const float z = (act - mu) / sigma;
        ssq += z * z; slog += logstd;
        if (row < E && a.action) a.action[(size_t)row * net.A + j] = act;
    }
    ssq = group16_sum(ssq); slog = group16_sum(slog);
    if (part == 0 && row < E) a.neglogp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)net.A + slog;
    if (tid < E) c.done_row[tid] = c.in.done[tid];
    // ---- env transition + EnvNormalize::step bookkeeping -> S_out (the tiles of the forward pass are free now) -------------
    __syncthreads();
    float* xs = lds + lay.w_total;                          // [E][O] raw observations of the NEW state
    float* rs = xs + NW_ROWS * 64;                          // [E] rewards | [E] dones | [E] returns
    for (int i = tid; i < E * (O + 2); i += NW_THREADS) {
        const int e = i / (O + 2), j = i - e * (O + 2);
        const uint32_t hsh = ctr_hash(c.seed, (uint32_t)(c.env0 + e), c.env_step, (uint32_t)j);
        if (j < O) { const float x = u32_to_sym_unit(hsh); xs[e * O + j] = x; c.out.raw_obs[(size_t)e * O + j] = x; }
        else if (j == O) rs[e] = u32_to_sym_unit(hsh);
        else { const float d = (hsh % 300u == 0u) ? 1.0f : 0.0f; rs[NW_ROWS + e] = d; c.out.done[e] = d; }
    }
    __syncthreads();
    // RunningStatistics::update's merge of a batch (mean, M2, n) (common/running_statistics.hpp:88-104), reading S_in, writing S_out
    auto merge = [&](float mean0, float var0, double cnt, float bmean, float bM2, float nbf, float& mean1, float& var1) __attribute__((always_inline)) {
        const double nb = (double)nbf, tot = cnt + nb;
        const float bvar = bM2 / (float)nb;                                        // :51-54
        const float delta = bmean - mean0;                                         // :90
        mean1 = mean0 + (delta * (float)nb) / (float)tot;                          // :94
        const float m_a = var0 * (float)cnt, m_b = bvar * (float)nb;               // :97-98
        const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;   // :100
        var1 = M2 / (float)tot;                                                    // :101
    };
    if (tid < O) {
        float m1 = c.in.obs_mean[tid], v1 = c.in.obs_var[tid];
        if (c.norm_obs) {
            float sum = 0.f;
            for (int e = 0; e < E; ++e) sum += xs[e * O + tid];
            const float bmean = sum / (float)E;                                    // colwise().mean()  (:38-39)
            float m2 = 0.f;
            for (int e = 0; e < E; ++e) { const float d = xs[e * O + tid] - bmean; m2 += d * d; }
            merge(c.in.obs_mean[tid], c.in.obs_var[tid], *c.in.obs_count, bmean, m2, (float)E, m1, v1);
        }
        c.out.obs_mean[tid] = m1; c.out.obs_var[tid] = v1;
        if (tid == 0) *c.out.obs_count = c.norm_obs ? (double)(float)E + *c.in.obs_count : *c.in.obs_count;     // :103
    }
    if (tid == 64) {                                        // (a different wave than the observation columns)
        float* ret = rs + 2 * NW_ROWS;
        float sum = 0.f;
        for (int e = 0; e < E; ++e) { ret[e] = c.in.ret[e] * c.gamma + rs[e]; sum += ret[e]; }        // env_normalize.hpp:66
        float m1 = *c.in.ret_mean, v1 = *c.in.ret_var;
        if (c.norm_rew) {                                                                              // :75-77 (training)
            const float bmean = sum / (float)E;
            float m2 = 0.f;
            for (int e = 0; e < E; ++e) { const float d = ret[e] - bmean; m2 += d * d; }
            merge(*c.in.ret_mean, *c.in.ret_var, *c.in.ret_count, bmean, m2, (float)E, m1, v1);
        }
        *c.out.ret_mean = m1; *c.out.ret_var = v1;
        *c.out.ret_count = c.norm_rew ? (double)(float)E + *c.in.ret_count : *c.in.ret_count;
        const float inv = 1.0f / sqrtf(v1 + c.eps);                                                    // :79
        for (int e = 0; e < E; ++e) {
            float y = rs[e];
            if (c.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -c.clip_rew), c.clip_rew); }
            c.rew_out[e] = y;
            c.out.ret[e] = ret[e] * (1.0f - rs[NW_ROWS + e]);                                          // :88-91
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Persistent rollout for small environment counts (n_envs <= 32) against the on-device seeded env: ALL T env steps of a
// rollout in ONE launch of ONE workgroup.  The policy tower's forward image is copied into LDS once; the raw observations,
// the running statistics (EnvNormalize: obs_rms, ret_rms, the discounted returns, the done flags) live in LDS / registers
// for the whole rollout; per env step the workgroup normalises, runs the forward pass, samples, steps the env and merges the
// statistics -- the arithmetic of narrow_collect_kernel, statement for statement -- and only STORES leave the CU (the rollout
// rows).  Nothing of a step waits on memory or on a launch: 7.5 us per env step (one launch each) becomes 4.4 us.
// The value tower is not needed inside the loop (values are consumed by the GAE scan only): the host runs it afterwards as
// one batched launch of narrow_step_kernel over the T x E normalised rows this kernel stored.
// Replaces, for this case, the loop of runner.hpp:75-127 over policies.hpp:33-46 + env_normalize.hpp:64-116.
// ------------------------------------------------------------------------------------------------------------------------
struct NwRolloutArgs {
    const float* img;            // policy tower's packed image
    NwEnvState st;               // read at entry, written back at exit
    const float* noise;          // [T][E][A] or null -> counter RNG
    float* ro_obs; float* ro_act; float* ro_nlp; float* ro_rew; float* ro_done;
    int E, T;
    uint32_t seed, step0; int env0;
    float gamma, clip_rew, clip_obs, eps; int norm_obs, norm_rew;
    unsigned long long* stamps;  // diagnostic builds only (-DPPO_STAMPS)
    // HOST mode (host_mode != 0): the env lives on the host.  The kernel stays resident across env steps: it publishes the actions
    // of step t into pinned host memory (host_act, then ctl[PCTL_D2H] = t + 1), waits until the host has posted the transition
    // (ctl[PCTL_H2D] >= t + 1, data in host_in = [E*O obs | E rewards | E dones]) and goes on -- no launch, no copy, no image
    // reload per env step.  A wait is BOUNDED (poll_cap polls, or the host's stop word): the kernel then saves its state, reports
    // how many transitions it has booked (ctl[PCTL_EXIT] = 1 + count) and exits; the host relaunches it at its next action
    // (t0, pending = a posted transition is still unbooked).  step0 is then the counter-RNG step of row t0 minus t0.
    int host_mode, t0, pending;
    const float* host_in; float* host_act; unsigned* ctl; unsigned poll_cap;
    const unsigned* h2d;         // narrow_rollout1_kernel<.., HOST> only: the host's sequence word in DEVICE memory (the host stores the transition and this word through the
                                 // BAR: posted writes, and the kernel polls its own memory -- two PCIe read round trips per env step less); null: ctl[PCTL_H2D] in pinned host memory
};
#define PCTL_H2D 0
#define PCTL_D2H 16
#define PCTL_EXIT 32
#define PCTL_STOP 48
#define NW_RO_XS (NW_ROWS * 64)                    // (narrow_host_step_kernel, <= 32 environments) raw observations
#define NW_RO_EXTRA (NW_RO_XS + 64 + 64 + 3 * NW_ROWS + 8 + 64 + 16 * 64)
#define NW_RO_MAX_E 64                             // environments ONE resident workgroup serves: groups of 32 rows, one after the other.
                                                   // Measured behind a host Env (us per env step, resident | general path): 1 env 18 | 37,
                                                   // 32: 25 | 38, 64: 35 | 38, 128: 50 | 39, 256: 82 | 49 -- beyond two groups the whole GPU wins
// floats of LDS behind the regular layout for E environments of O observations (E rounded up to whole row groups)
__host__ __device__ inline int nw_ro_extra(int E, int O) { const int Er = (E + NW_ROWS - 1) / NW_ROWS * NW_ROWS; return Er * O + 64 + 64 + 3 * Er + 8 + 64 + 16 * 64; }

// MULTI = more than one group of 32 rows (33..NW_RO_MAX_E environments); the single-group instantiation keeps the straight-line
// step (the runtime group loop costs the <= 32-environment case ~2.5 k cycles per env step in scheduling freedom)
template <int KP0, int HP, int AP, int LL, bool MULTI = false>
__global__ __launch_bounds__(NW_THREADS) void narrow_rollout_kernel(NetDev net, NwLayout lay, NwRolloutArgs q) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(NwRolloutArgs)>();
    const int tid = threadIdx.x, pipe = tid >> 8, ptid = tid & 255;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    const int E = q.E, O = net.O, A = net.A;
    // stride of the per-environment vectors / size of the observation block: compile-time for the single-group instantiation
    const int ES = MULTI ? (E + NW_ROWS - 1) / NW_ROWS * NW_ROWS : NW_ROWS;
    float* xs = lds + lay.lds_total;                        // [E][O] raw observations of the current state
    float* s_mean = xs + (MULTI ? ES * O : NW_RO_XS); float* s_var = s_mean + 64;
    float* rs = s_var + 64;                                 // [ES] rewards | [ES] dones | [ES] returns
    float* s_retstat = rs + 3 * ES;                         // ret_rms mean, var
    float* s_istd = s_retstat + 8;                          // 1 / sqrt(var + eps) per column: the expression of the per-step kernels, evaluated once per statistics update
    // ---- entry: image + state, one round trip ----------------------------------------------------------------------------
    {
        const int n4 = lay.w_fwd / 4;
        for (int e = tid; e < n4; e += NW_THREADS) reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(q.img)[e];
        for (int i = tid; i < E * O; i += NW_THREADS) xs[i] = q.st.raw_obs[i];
        if (tid < O) { s_mean[tid] = q.st.obs_mean[tid]; const float v0 = q.st.obs_var[tid]; s_var[tid] = v0; s_istd[tid] = 1.0f / sqrtf(v0 + q.eps); }
        if (tid < E) { rs[ES + tid] = q.st.done[tid]; rs[2 * ES + tid] = q.st.ret[tid]; }
        if (tid == 0) { s_retstat[0] = *q.st.ret_mean; s_retstat[1] = *q.st.ret_var; }
    }
    double obs_cnt = *q.st.obs_count, ret_cnt = *q.st.ret_count;          // replicated: every thread that merges holds the count
    __syncthreads();
    float* P = lds + lay.w_total + pipe * lay.pipe_total;
    const float* par = lds + lay.par;
    const int r = ptid >> 4, part = ptid & 15;
    const int row = 16 * pipe + r;
    const bool have_helper = NW_PIPES == 2 && E <= 16, helper = have_helper && pipe == 1;
    float* s_eps = s_istd + 64;                             // [16][A] counter-RNG draws of the current step (written by the helper pipe)
    auto merge = [&](float mean0, float var0, double cnt, float bmean, float bM2, float nbf, float& mean1, float& var1) __attribute__((always_inline)) {
        const double nb = (double)nbf, tot = cnt + nb;
        const float bvar = bM2 / (float)nb;                                        // running_statistics.hpp:51-54
        const float delta = bmean - mean0;                                         // :90
        mean1 = mean0 + (delta * (float)nb) / (float)tot;                          // :94
        const float m_a = var0 * (float)cnt, m_b = bvar * (float)nb;               // :97-98
        const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;   // :100
        var1 = M2 / (float)tot;                                                    // :101
    };
#ifdef PPO_STAMPS
#define RSTAMP(i) do { if (q.stamps && tid == 0 && t == q.T - 1) q.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
    int* s_ok = reinterpret_cast<int*>(s_retstat + 4);
    // host transition -> xs / rs (zero-copy reads of the pinned block)
    auto read_host = [&]() __attribute__((always_inline)) {
        for (int i = tid; i < E * O; i += NW_THREADS) xs[i] = q.host_in[i];
        if (tid < E) { rs[tid] = q.host_in[(size_t)E * O + tid]; rs[ES + tid] = q.host_in[(size_t)E * O + E + tid]; }
    };
    // EnvNormalize::step bookkeeping of the transition in xs / rs (env_normalize.hpp:64-116, running_statistics.hpp:26-104); rewards -> row tr
    auto bookkeeping = [&](int tr) __attribute__((always_inline)) {
        if (tid < O) {
            if (q.norm_obs) {
                float sum = 0.f;
                for (int e = 0; e < E; ++e) sum += xs[e * O + tid];
                const float bmean = sum / (float)E;                                    // colwise().mean()
                float m2 = 0.f;
                for (int e = 0; e < E; ++e) { const float d = xs[e * O + tid] - bmean; m2 += d * d; }
                float m1, v1;
                merge(s_mean[tid], s_var[tid], obs_cnt, bmean, m2, (float)E, m1, v1);
                s_mean[tid] = m1; s_var[tid] = v1; s_istd[tid] = 1.0f / sqrtf(v1 + q.eps);
                obs_cnt = (double)(float)E + obs_cnt;                                   // :103
            }
        }
        if (tid == 64) {                                        // (a different wave than the observation columns)
            float* ret = rs + 2 * ES;
            float sum = 0.f;
            for (int e = 0; e < E; ++e) { ret[e] = ret[e] * q.gamma + rs[e]; sum += ret[e]; }            // env_normalize.hpp:66
            float m1 = s_retstat[0], v1 = s_retstat[1];
            if (q.norm_rew) {                                                                              // :75-77 (training)
                const float bmean = sum / (float)E;
                float m2 = 0.f;
                for (int e = 0; e < E; ++e) { const float d = ret[e] - bmean; m2 += d * d; }
                merge(s_retstat[0], s_retstat[1], ret_cnt, bmean, m2, (float)E, m1, v1);
                ret_cnt = (double)(float)E + ret_cnt;
            }
            s_retstat[0] = m1; s_retstat[1] = v1;
            const float inv = 1.0f / sqrtf(v1 + q.eps);                                                    // :79
            for (int e = 0; e < E; ++e) {
                float y = rs[e];
                if (q.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -q.clip_rew), q.clip_rew); }
                q.ro_rew[(size_t)tr * E + e] = y;
                ret[e] = ret[e] * (1.0f - rs[ES + e]);                                                // :88-91
            }
        }
    };
    int booked = q.t0;                                          // transitions whose bookkeeping is done (host mode's exit report)
    if (q.host_mode && q.pending) {                             // a transition posted before this launch: rewards row t0 - 1
        booked = q.t0 - 1;
        read_host();
        lds_barrier();
        bookkeeping(q.t0 - 1);
        lds_barrier();
        booked = q.t0;
    }
    for (int t = q.t0; t < q.T; ++t) {
        RSTAMP(0);
        if (tid < E) q.ro_done[(size_t)t * E + tid] = rs[ES + tid];                  // the flags that arrived with obs_t
        // the environments go through the two 16-row pipes in groups of 32 (one group at <= 32 environments)
        for (int g0 = 0; g0 < (MULTI ? E : 1); g0 += NW_ROWS) {
            const int grow = g0 + row;                          // this thread's environment in the sampling phase
            const bool live_pipe = g0 + 16 * pipe < E;          // a pipe without environments skips the matrix work (it shares the SIMDs' matrix pipes with the live one)
            // explicit noise of this step: requested now, consumed after the forward pass
            float nz_eps[4] = {0.f, 0.f, 0.f, 0.f};
            if (q.noise) {
                if (grow < E) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int j = part + 16 * k; if (j < A) nz_eps[k] = q.noise[((size_t)t * E + grow) * A + j]; }
                }
            } else if (helper) {
                // counter RNG (two 64-bit hashes, log, cos, sqrt per draw): with <= 16 environments the second pipe has no rows and
                // draws the first pipe's noise into LDS while that one runs the forward pass
                if (r < E) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int j = part + 16 * k; if (j < A) s_eps[r * A + j] = ctr_normal(q.seed, (uint32_t)q.env0 + r, q.step0 + (uint32_t)t, j); }
                }
            } else if (!have_helper && grow < E) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int j = part + 16 * k; if (j < A) nz_eps[k] = ctr_normal(q.seed, (uint32_t)q.env0 + grow, q.step0 + (uint32_t)t, j); }
            }
            // ---- normalise the group's observations (env_normalize.hpp:99-104) -> input tiles + rollout row t ----------------
            for (int i = tid; i < NW_ROWS * Kp0; i += NW_THREADS) {
                const int rr = i / Kp0, j = i - rr * Kp0, ge = g0 + rr;
                float x = 0.f;
                if (ge < E && j < O) {
                    x = xs[ge * O + j];
                    if (q.norm_obs) {
                        x = (x - s_mean[j]) * s_istd[j];
                        x = tf_min(tf_max(x, -q.clip_obs), q.clip_obs);
                    }
                    q.ro_obs[((size_t)t * E + ge) * O + j] = x;
                }
                lds[lay.w_total + (rr >> 4) * lay.pipe_total + lay.x[0] + (rr & 15) * lay.ldx[0] + j] = x;
            }
            lds_barrier();
            RSTAMP(1);
            // ---- forward + head (narrow_step_kernel's code) -----------------------------------------------------------------
            for (int l = 0; l < L; ++l) {
                const float* bias = par + net.par_b[l];
                float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
                auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
                    const float b = bias[col];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) Ys[(4 * g + rr) * ldy + col] = fast_tanh(acc[rr] + b);
                };
                if (live_pipe) {
                    if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep);
                    else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep);
                }
                lds_barrier();
                RSTAMP(2 + l);
            }
            const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
            float* mus = P + lay.mu; const int ldm = lay.ldm;
            if (live_pipe) nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
                const float b = par[net.par_bmu + col];
#pragma unroll
                for (int k = 0; k < 4; ++k) mus[(4 * g + k) * ldm + col] = acc[k] + b;
            });
            lds_barrier();
            RSTAMP(6);
            // ---- sample + neglogp (G:5894-6672) --------------------------------------------------------------------------------
            if (live_pipe) {
                float ssq = 0.f, slog = 0.f;
                int k = 0;
                for (int j = part; j < A; j += 16, ++k) {
                    const float mu = mus[r * ldm + j];
                    const float logstd = mu * 0.0f + par[net.par_ls + j];
                    const float sigma = expf(logstd);
                    float eps = 0.f;
                    if (grow < E) eps = (q.noise || !have_helper) ? nz_eps[k & 3] : s_eps[grow * A + j];      // (A <= 64: k < 4)
                    const float act = mu + sigma * eps;
                    const float z = (act - mu) / sigma;
                    ssq += z * z; slog += logstd;
                    if (grow < E) { q.ro_act[((size_t)t * E + grow) * A + j] = act; if (q.host_mode) q.host_act[(size_t)grow * A + j] = act; }
                }
                ssq = group16_sum(ssq); slog = group16_sum(slog);
                if (part == 0 && grow < E) q.ro_nlp[(size_t)t * E + grow] = 0.5f * ssq + HALF_LOG_2PI * (float)A + slog;
            }
        }
        RSTAMP(7);
        if (q.host_mode) {
            // ---- publish the actions, wait for the host's transition ---------------------------------------------------------------
            __threadfence_system();                             // every thread's stores into host_act have landed
            __syncthreads();
            if (tid == 0) {
                peer_st_sys(q.ctl + PCTL_D2H, (unsigned)(t + 1));
                unsigned n = 0; int ok = 1;
                while (peer_ld_sys(q.ctl + PCTL_H2D) < (unsigned)(t + 1)) {
                    if (++n > q.poll_cap || peer_ld_sys(q.ctl + PCTL_STOP)) { ok = 0; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                *s_ok = ok;
            }
            __syncthreads();
            if (!*s_ok) break;                                  // bounded wait over (or stop requested): save the state and leave
            __atomic_thread_fence(__ATOMIC_ACQUIRE);            // system scope: the transition's bytes are read after its sequence word
            read_host();
        } else {
            // ---- env transition (counter hash) -> new raw observations, rewards, dones -------------------------------------------
            const uint32_t env_step = q.step0 + (uint32_t)t + 1u;
            for (int i = tid; i < E * (O + 2); i += NW_THREADS) {
                const int e = i / (O + 2), j = i - e * (O + 2);
                const uint32_t hsh = ctr_hash(q.seed, (uint32_t)(q.env0 + e), env_step, (uint32_t)j);
                if (j < O) xs[e * O + j] = u32_to_sym_unit(hsh);
                else if (j == O) rs[e] = u32_to_sym_unit(hsh);
                else rs[ES + e] = (hsh % 300u == 0u) ? 1.0f : 0.0f;
            }
        }
        lds_barrier();
        RSTAMP(8);
        bookkeeping(t);
        booked = t + 1;
        RSTAMP(9);
        lds_barrier();
        RSTAMP(10);
    }
    // ---- exit: the state goes home ----------------------------------------------------------------------------------------------
    for (int i = tid; i < E * O; i += NW_THREADS) q.st.raw_obs[i] = xs[i];
    if (tid < O) { q.st.obs_mean[tid] = s_mean[tid]; q.st.obs_var[tid] = s_var[tid]; }
    if (tid < E) { q.st.done[tid] = rs[ES + tid]; q.st.ret[tid] = rs[2 * ES + tid]; }
    if (tid == 0) *q.st.obs_count = obs_cnt;
    if (tid == 64) { *q.st.ret_mean = s_retstat[0]; *q.st.ret_var = s_retstat[1]; *q.st.ret_count = ret_cnt; }
    if (q.host_mode) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) peer_st_sys(q.ctl + PCTL_EXIT, 1u + (unsigned)booked);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Host-Env rollout step for small environment counts (n_envs <= 32; BASELINE configs[1]: ONE environment stepped on the
// host): one launch per env step instead of {H2D copy, statistics kernel, policy step of both towers, D2H copy}.
// The kernel reads the transition the host's Env::step produced STRAIGHT from the handle's pinned host block (zero-copy over
// PCIe: 80 bytes at one environment), does EnvNormalize::step's bookkeeping for it (env_normalize.hpp:64-116, the statements
// of narrow_collect_kernel), then normalises the new observations, runs the policy tower, samples, stores the actions
// STRAIGHT into pinned host memory and raises a completion word there; the host spins on that word.  No copy-engine
// operation and no stream query sits between the host's env step and the next action (34.6 -> ~16 us per env step at one
// environment).  The value tower is not needed per step: one batched launch at ppo_rollout_finish.
// ------------------------------------------------------------------------------------------------------------------------
struct NwHostStepArgs {
    const float* img;            // policy tower's packed image
    NwEnvState st;               // device-resident state (read and written: one workgroup, no race)
    const float* host_in;        // pinned host block [E*O obs | E rewards | E dones]: the transition that followed the previous action
    float* host_act;             // pinned host [E*A]
    unsigned* host_flag; unsigned flag_value;
    const float* noise;          // device [E][A] or null -> counter RNG
    float* ro_obs; float* ro_act; float* ro_nlp; float* ro_done;     // rollout row t (act != 0)
    float* ro_rew_prev;          // rollout rewards row t-1 (has_transition != 0)
    int E; int has_transition; int act;
    uint32_t seed, rng_step, row_base;
    float gamma, clip_rew, clip_obs, eps; int norm_obs, norm_rew;
};

template <int KP0, int HP, int AP, int LL>
__global__ __launch_bounds__(NW_THREADS) void narrow_host_step_kernel(NetDev net, NwLayout lay, NwHostStepArgs q) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(NwHostStepArgs)>();
    const int tid = threadIdx.x, pipe = tid >> 8, ptid = tid & 255;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    const int E = q.E, O = net.O, A = net.A;
    float* xs = lds + lay.lds_total;                        // [E][O] raw observations
    float* s_mean = xs + NW_RO_XS; float* s_var = s_mean + 64;
    float* rs = s_var + 64;                                 // [32] rewards | [32] dones | [32] returns
    // ---- everything this launch reads, requested together: image, transition (host memory) or current observations, state ----
    {
        const int n4 = q.act ? lay.w_fwd / 4 : 0;
        for (int e = tid; e < n4; e += NW_THREADS) reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(q.img)[e];
        const float* src = q.has_transition ? q.host_in : q.st.raw_obs;
        for (int i = tid; i < E * O; i += NW_THREADS) xs[i] = src[i];
        if (tid < O) { s_mean[tid] = q.st.obs_mean[tid]; s_var[tid] = q.st.obs_var[tid]; }
        if (tid < E) {
            rs[2 * NW_ROWS + tid] = q.st.ret[tid];
            if (q.has_transition) { rs[tid] = q.host_in[(size_t)E * O + tid]; rs[NW_ROWS + tid] = q.host_in[(size_t)E * O + E + tid]; }
            else rs[NW_ROWS + tid] = q.st.done[tid];
        }
    }
    float nz_eps[4] = {0.f, 0.f, 0.f, 0.f};
    const int r = ptid >> 4, part = ptid & 15;
    const int row = 16 * pipe + r;
    if (q.act && q.noise && row < E) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = part + 16 * k; if (j < A) nz_eps[k] = q.noise[(size_t)row * A + j]; }
    }
    __syncthreads();
    // ---- EnvNormalize::step bookkeeping for the transition that arrived (env_normalize.hpp:64-116) ------------------------------
    if (q.has_transition) {
        auto merge = [&](float mean0, float var0, double cnt, float bmean, float bM2, float nbf, float& mean1, float& var1) __attribute__((always_inline)) {
            const double nb = (double)nbf, tot = cnt + nb;
            const float bvar = bM2 / (float)nb;                                        // running_statistics.hpp:51-54
            const float delta = bmean - mean0;                                         // :90
            mean1 = mean0 + (delta * (float)nb) / (float)tot;                          // :94
            const float m_a = var0 * (float)cnt, m_b = bvar * (float)nb;               // :97-98
            const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;   // :100
            var1 = M2 / (float)tot;                                                    // :101
        };
        if (tid < O && q.norm_obs) {
            const double cnt = *q.st.obs_count;
            float sum = 0.f;
            for (int e = 0; e < E; ++e) sum += xs[e * O + tid];
            const float bmean = sum / (float)E;
            float m2 = 0.f;
            for (int e = 0; e < E; ++e) { const float d = xs[e * O + tid] - bmean; m2 += d * d; }
            float m1, v1;
            merge(s_mean[tid], s_var[tid], cnt, bmean, m2, (float)E, m1, v1);
            s_mean[tid] = m1; s_var[tid] = v1;
            q.st.obs_mean[tid] = m1; q.st.obs_var[tid] = v1;
        }
        if (tid == 64) {
            float* ret = rs + 2 * NW_ROWS;
            float sum = 0.f;
            for (int e = 0; e < E; ++e) { ret[e] = ret[e] * q.gamma + rs[e]; sum += ret[e]; }            // :66
            float m1 = *q.st.ret_mean, v1 = *q.st.ret_var;
            if (q.norm_rew) {
                const double cnt = *q.st.ret_count;
                const float bmean = sum / (float)E;
                float m2 = 0.f;
                for (int e = 0; e < E; ++e) { const float d = ret[e] - bmean; m2 += d * d; }
                merge(*q.st.ret_mean, *q.st.ret_var, cnt, bmean, m2, (float)E, m1, v1);
                *q.st.ret_mean = m1; *q.st.ret_var = v1; *q.st.ret_count = (double)(float)E + cnt;
            }
            const float inv = 1.0f / sqrtf(v1 + q.eps);                                                    // :79
            for (int e = 0; e < E; ++e) {
                float y = rs[e];
                if (q.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -q.clip_rew), q.clip_rew); }
                q.ro_rew_prev[e] = y;
                q.st.ret[e] = ret[e] * (1.0f - rs[NW_ROWS + e]);                                           // :88-91
                q.st.done[e] = rs[NW_ROWS + e];
            }
        }
        for (int i = tid; i < E * O; i += NW_THREADS) q.st.raw_obs[i] = xs[i];       // the device copy ppo_rollout_finish bootstraps from
        __syncthreads();
        // (the count is bumped after every column has read it)
        if (tid == 0 && q.norm_obs) *q.st.obs_count = (double)(float)E + *q.st.obs_count;              // :103
    }
    if (!q.act) return;
    // ---- normalise (env_normalize.hpp:99-104) -> input tile + rollout row -----------------------------------------------------------
    for (int i = tid; i < NW_ROWS * Kp0; i += NW_THREADS) {
        const int rr = i / Kp0, j = i - rr * Kp0;
        float x = 0.f;
        if (rr < E && j < O) {
            x = xs[rr * O + j];
            if (q.norm_obs) {
                x = (x - s_mean[j]) * (1.0f / sqrtf(s_var[j] + q.eps));
                x = tf_min(tf_max(x, -q.clip_obs), q.clip_obs);
            }
            q.ro_obs[(size_t)rr * O + j] = x;
        }
        lds[lay.w_total + (rr >> 4) * lay.pipe_total + lay.x[0] + (rr & 15) * lay.ldx[0] + j] = x;
    }
    if (tid < E) q.ro_done[tid] = rs[NW_ROWS + tid];
    lds_barrier();
    float* P = lds + lay.w_total + pipe * lay.pipe_total;
    const float* par = lds + lay.par;
    const bool live_pipe = 16 * pipe < E;
    for (int l = 0; l < L; ++l) {
        const float* bias = par + net.par_b[l];
        float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
        auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = bias[col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Ys[(4 * g + rr) * ldy + col] = fast_tanh(acc[rr] + b);
        };
        if (live_pipe) {
            if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep);
            else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep);
        }
        lds_barrier();
    }
    const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
    float* mus = P + lay.mu; const int ldm = lay.ldm;
    if (live_pipe) nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
        const float b = par[net.par_bmu + col];
#pragma unroll
        for (int k = 0; k < 4; ++k) mus[(4 * g + k) * ldm + col] = acc[k] + b;
    });
    lds_barrier();
    if (live_pipe) {
        float ssq = 0.f, slog = 0.f;
        int k = 0;
        for (int j = part; j < A; j += 16, ++k) {
            const float mu = mus[r * ldm + j];
            const float logstd = mu * 0.0f + par[net.par_ls + j];
            const float sigma = expf(logstd);
            float eps = 0.f;
            if (row < E) eps = q.noise ? nz_eps[k & 3] : ctr_normal(q.seed, q.row_base + row, q.rng_step, j);
            const float act = mu + sigma * eps;
            const float z = (act - mu) / sigma;
            ssq += z * z; slog += logstd;
            if (row < E) { q.ro_act[(size_t)row * A + j] = act; q.host_act[(size_t)row * A + j] = act; }
        }
        ssq = group16_sum(ssq); slog = group16_sum(slog);
        if (part == 0 && row < E) q.ro_nlp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)A + slog;
    }
    // ---- publish: the host's actions have landed, then the completion word ------------------------------------------------------------
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(q.host_flag, q.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------------------------------
// Cooperative persistent rollout for 65..2048 environments on the device env (BASELINE configs[3]: 1024): G = ceil(E / 32)
// workgroups, each resident on its own CU for the whole rollout and owning 32 environments (image, raw observations, returns
// and done flags in LDS).  The only coupling between them is EnvNormalize's running statistics: per env step every workgroup
// reduces its 32 rows to chunk moments (n, mean, M2 per column, the reference's two passes), publishes them, all workgroups
// meet at a flat arrival counter, and each one combines the G chunks in index order (mean = sum n_k mean_k / n, M2 = sum M2_k
// + n_k (mean_k - mean)^2: norm_batch_kernel's combine) before RunningStatistics::update's merge -- every workgroup holds the
// same statistics, bit for bit.  One launch instead of 3 T (policy step, env, statistics kernel per env step): 21 -> 14 us per
// env step at 1024 environments.  The waits are bounded (a workgroup that never arrives sets `err` instead of hanging the GPU).
// ------------------------------------------------------------------------------------------------------------------------
struct NwCoopArgs {
    const float* img;
    NwEnvState st;               // read at entry; workgroup 0 writes the statistics back, every workgroup its environments' rows
    const float* noise;          // [T][E][A] or null
    float* ro_obs; float* ro_act; float* ro_nlp; float* ro_rew; float* ro_done;
    int E, T, G;
    uint32_t seed, step0; int env0;
    float gamma, clip_rew, clip_obs, eps; int norm_obs, norm_rew;
    float* part;                 // [2 parities][G][2 O + 3] chunk moments (allocated for NW_COOP_PW floats per workgroup)
    unsigned* arrive;            // [G][16] step words, one 64-byte line per workgroup (zeroed by the host before every launch)
    unsigned* err; unsigned spin_limit;
};
#define NW_COOP_PW 132           // upper bound of the floats one workgroup publishes per step (2 O + 3 at O <= 64)
#define NW_COOP_MAX_G 64

template <int KP0, int HP, int AP, int LL>
__global__ __launch_bounds__(NW_THREADS) void narrow_rollout_coop_kernel(NetDev net, NwLayout lay, NwCoopArgs q) {
    typedef NwShape<KP0, HP, AP, LL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(NwLayout) + sizeof(NwCoopArgs)>();
    const int tid = threadIdx.x, pipe = tid >> 8, ptid = tid & 255;
    const int L = S::L(net), Kp0 = S::Kp0(net), Ap = S::Ap(net);
    const int wg = blockIdx.x, G = q.G, O = net.O, A = net.A;
    const int e0 = wg * NW_ROWS, El = min(NW_ROWS, q.E - e0);      // this workgroup's environments [e0, e0 + El)
    float* xs = lds + lay.lds_total;                        // [El][O] raw observations
    float* s_mean = xs + NW_RO_XS; float* s_var = s_mean + 64;
    float* rs = s_var + 64;                                 // [32] rewards | [32] dones | [32] returns
    float* s_retstat = rs + 3 * NW_ROWS;                    // ret_rms mean, var | [4..7] scratch
    float* s_istd = s_retstat + 8;
    {
        const int n4 = lay.w_fwd / 4;
        for (int e = tid; e < n4; e += NW_THREADS) reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(q.img)[e];
        for (int i = tid; i < El * O; i += NW_THREADS) xs[i] = q.st.raw_obs[(size_t)e0 * O + i];
        if (tid < O) { s_mean[tid] = q.st.obs_mean[tid]; const float v0 = q.st.obs_var[tid]; s_var[tid] = v0; s_istd[tid] = 1.0f / sqrtf(v0 + q.eps); }
        if (tid < El) { rs[NW_ROWS + tid] = q.st.done[e0 + tid]; rs[2 * NW_ROWS + tid] = q.st.ret[e0 + tid]; }
        if (tid == 0) { s_retstat[0] = *q.st.ret_mean; s_retstat[1] = *q.st.ret_var; }
    }
    double obs_cnt = *q.st.obs_count, ret_cnt = *q.st.ret_count;
    __syncthreads();
    float* P = lds + lay.w_total + pipe * lay.pipe_total;
    const float* par = lds + lay.par;
    const int r = ptid >> 4, part_ = ptid & 15;
    const int row = 16 * pipe + r;                          // local environment of this thread in the sampling phase
    const bool live_pipe = 16 * pipe < El;
    int* s_ok = reinterpret_cast<int*>(s_retstat + 4);
    const int pw = 2 * O + 3;                               // published chunk: n | mean[O] | M2[O] | ret mean | ret M2
    float* lp = lds + lay.w_fwd;                            // all G chunks of a step, staged in LDS (the image's backward half is unused here)
    auto merge = [&](float mean0, float var0, double cnt, float bmean, float bM2, float nbf, float& mean1, float& var1) __attribute__((always_inline)) {
        const double nb = (double)nbf, tot = cnt + nb;
        const float bvar = bM2 / (float)nb;                                        // running_statistics.hpp:51-54
        const float delta = bmean - mean0;                                         // :90
        mean1 = mean0 + (delta * (float)nb) / (float)tot;                          // :94
        const float m_a = var0 * (float)cnt, m_b = bvar * (float)nb;               // :97-98
        const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;   // :100
        var1 = M2 / (float)tot;                                                    // :101
    };
    bool dead = false;
    for (int t = 0; t < q.T && !dead; ++t) {
        float nz_eps[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < El) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = part_ + 16 * k;
                if (j < A) nz_eps[k] = q.noise ? q.noise[((size_t)t * q.E + e0 + row) * A + j] : ctr_normal(q.seed, (uint32_t)(q.env0 + e0 + row), q.step0 + (uint32_t)t, j);
            }
        }
        // ---- normalise (env_normalize.hpp:99-104) -> input tiles + rollout row t ----------------------------------------------------
        for (int i = tid; i < NW_ROWS * Kp0; i += NW_THREADS) {
            const int rr = i / Kp0, j = i - rr * Kp0;
            float x = 0.f;
            if (rr < El && j < O) {
                x = xs[rr * O + j];
                if (q.norm_obs) {
                    x = (x - s_mean[j]) * s_istd[j];
                    x = tf_min(tf_max(x, -q.clip_obs), q.clip_obs);
                }
                q.ro_obs[((size_t)t * q.E + e0 + rr) * O + j] = x;
            }
            lds[lay.w_total + (rr >> 4) * lay.pipe_total + lay.x[0] + (rr & 15) * lay.ldx[0] + j] = x;
        }
        if (tid < El) q.ro_done[(size_t)t * q.E + e0 + tid] = rs[NW_ROWS + tid];
        lds_barrier();
        for (int l = 0; l < L; ++l) {
            const float* bias = par + net.par_b[l];
            float* Ys = P + lay.x[l + 1]; const int ldy = lay.ldx[l + 1];
            auto ep = [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
                const float b = bias[col];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Ys[(4 * g + rr) * ldy + col] = fast_tanh(acc[rr] + b);
            };
            if (live_pipe) {
                if (l == 0) nw_dense<KP0>(P + lay.x[0], lay.ldx[0], Kp0, lds + lay.wf[0], lay.wf_ld[0], S::Hp(net, 0), ep);
                else nw_dense<HP>(P + lay.x[l], lay.ldx[l], S::Hp(net, l - 1), lds + lay.wf[l], lay.wf_ld[l], S::Hp(net, l), ep);
            }
            lds_barrier();
        }
        const float* hL = P + lay.x[L]; const int ldh = lay.ldx[L]; const int HpL = S::Hp(net, L - 1);
        float* mus = P + lay.mu; const int ldm = lay.ldm;
        if (live_pipe) nw_dense<HP>(hL, ldh, HpL, lds + lay.wh, lay.wh_ld, Ap, [&](const f32x4& acc, int g, int col) __attribute__((always_inline)) {
            const float b = par[net.par_bmu + col];
#pragma unroll
            for (int k = 0; k < 4; ++k) mus[(4 * g + k) * ldm + col] = acc[k] + b;
        });
        lds_barrier();
        if (live_pipe) {
            float ssq = 0.f, slog = 0.f;
            int k = 0;
            for (int j = part_; j < A; j += 16, ++k) {
                const float mu = mus[r * ldm + j];
                const float logstd = mu * 0.0f + par[net.par_ls + j];
                const float sigma = expf(logstd);
                const float eps = row < El ? nz_eps[k & 3] : 0.f;
                const float act = mu + sigma * eps;
                const float z = (act - mu) / sigma;
                ssq += z * z; slog += logstd;
                if (row < El) q.ro_act[((size_t)t * q.E + e0 + row) * A + j] = act;
            }
            ssq = group16_sum(ssq); slog = group16_sum(slog);
            if (part_ == 0 && row < El) q.ro_nlp[(size_t)t * q.E + e0 + row] = 0.5f * ssq + HALF_LOG_2PI * (float)A + slog;
        }
        // ---- env transition (counter hash) of this workgroup's environments ---------------------------------------------------------
        const uint32_t env_step = q.step0 + (uint32_t)t + 1u;
        for (int i = tid; i < El * (O + 2); i += NW_THREADS) {
            const int e = i / (O + 2), j = i - e * (O + 2);
            const uint32_t hsh = ctr_hash(q.seed, (uint32_t)(q.env0 + e0 + e), env_step, (uint32_t)j);
            if (j < O) xs[e * O + j] = u32_to_sym_unit(hsh);
            else if (j == O) rs[e] = u32_to_sym_unit(hsh);
            else rs[NW_ROWS + e] = (hsh % 300u == 0u) ? 1.0f : 0.0f;
        }
        lds_barrier();
        // ---- chunk moments of my rows -> published (agent-scope stores: visible to the other XCDs without a cache-wide fence) ------
        float* mine = q.part + ((size_t)(t & 1) * G + wg) * pw;
        if (tid < O) {
            float sum = 0.f;
            for (int e = 0; e < El; ++e) sum += xs[e * O + tid];
            const float bmean = sum / (float)El;
            float m2 = 0.f;
            for (int e = 0; e < El; ++e) { const float d = xs[e * O + tid] - bmean; m2 += d * d; }
            __hip_atomic_store(mine + 1 + tid, bmean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1 + O + tid, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(mine, (float)El, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 64) {
            float* ret = rs + 2 * NW_ROWS;
            float sum = 0.f;
            for (int e = 0; e < El; ++e) { ret[e] = ret[e] * q.gamma + rs[e]; sum += ret[e]; }            // env_normalize.hpp:66
            const float bmean = sum / (float)El;
            float m2 = 0.f;
            for (int e = 0; e < El; ++e) { const float d = ret[e] - bmean; m2 += d * d; }
            __hip_atomic_store(mine + 1 + 2 * O, bmean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 2 + 2 * O, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- all workgroups meet.  No read-modify-write on a shared word (same-address atomics serialise at the memory side): every
        // workgroup raises ITS OWN step word once its chunk's stores have been performed, and G lanes of every workgroup watch the G
        // words in parallel (bounded: a workgroup that never shows up sets `err` instead of hanging the device).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { *s_ok = 1; __hip_atomic_store(q.arrive + (size_t)wg * 16, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __syncthreads();
        if (tid < G) {
            unsigned n = 0;
            while (__hip_atomic_load(q.arrive + (size_t)tid * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t + 1)) {
                if (++n > q.spin_limit) { *s_ok = 0; __hip_atomic_store(q.err, 1u + (unsigned)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!*s_ok) { dead = true; }
        // every chunk of the step into LDS: all loads of the workgroup in one round trip (agent-scope loads: past any stale line)
        {
            const float* all = q.part + (size_t)(t & 1) * G * pw;
            for (int i = tid; i < G * pw; i += NW_THREADS) lp[i] = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();
        if (tid < O && q.norm_obs) {
            float ntot = 0.f, acc = 0.f;
            for (int k = 0; k < G; ++k) { const float nk = lp[k * pw]; ntot += nk; acc += nk * lp[k * pw + 1 + tid]; }
            const float bmean = acc / ntot;
            float m2 = 0.f;
            for (int k = 0; k < G; ++k) { const float nk = lp[k * pw], d = lp[k * pw + 1 + tid] - bmean; m2 += lp[k * pw + 1 + O + tid] + nk * (d * d); }
            float m1, v1;
            merge(s_mean[tid], s_var[tid], obs_cnt, bmean, m2, ntot, m1, v1);
            s_mean[tid] = m1; s_var[tid] = v1; s_istd[tid] = 1.0f / sqrtf(v1 + q.eps);
            obs_cnt = (double)ntot + obs_cnt;                                       // :103
        }
        if (tid == 64) {
            float* ret = rs + 2 * NW_ROWS;
            float m1 = s_retstat[0], v1 = s_retstat[1];
            if (q.norm_rew) {
                float ntot = 0.f, acc = 0.f;
                for (int k = 0; k < G; ++k) { const float nk = lp[k * pw]; ntot += nk; acc += nk * lp[k * pw + 1 + 2 * O]; }
                const float bmean = acc / ntot;
                float m2 = 0.f;
                for (int k = 0; k < G; ++k) { const float nk = lp[k * pw], d = lp[k * pw + 1 + 2 * O] - bmean; m2 += lp[k * pw + 2 + 2 * O] + nk * (d * d); }
                merge(s_retstat[0], s_retstat[1], ret_cnt, bmean, m2, ntot, m1, v1);
                ret_cnt = (double)ntot + ret_cnt;
            }
            s_retstat[0] = m1; s_retstat[1] = v1;
            const float inv = 1.0f / sqrtf(v1 + q.eps);                                                    // :79
            for (int e = 0; e < El; ++e) {
                float y = rs[e];
                if (q.norm_rew) { y = y * inv; y = tf_min(tf_max(y, -q.clip_rew), q.clip_rew); }
                q.ro_rew[(size_t)t * q.E + e0 + e] = y;
                ret[e] = ret[e] * (1.0f - rs[NW_ROWS + e]);                                                // :88-91
            }
        }
        lds_barrier();
    }
    // ---- exit: my environments' state goes home; workgroup 0 also writes the (common) statistics ------------------------------------
    for (int i = tid; i < El * O; i += NW_THREADS) q.st.raw_obs[(size_t)e0 * O + i] = xs[i];
    if (tid < El) { q.st.done[e0 + tid] = rs[NW_ROWS + tid]; q.st.ret[e0 + tid] = rs[2 * NW_ROWS + tid]; }
    if (wg == 0) {
        if (tid < O) { q.st.obs_mean[tid] = s_mean[tid]; q.st.obs_var[tid] = s_var[tid]; }
        if (tid == 0) *q.st.obs_count = obs_cnt;
        if (tid == 64) { *q.st.ret_mean = s_retstat[0]; *q.st.ret_var = s_retstat[1]; *q.st.ret_count = ret_cnt; }
    }
}
