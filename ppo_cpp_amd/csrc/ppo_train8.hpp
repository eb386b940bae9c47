// ppo_train8.hpp -- train model forward + loss + backward of one 16-row tile, 8 waves per workgroup, for hidden [256,256] with any
// observation / action width up to 64 (BASELINE configs[2] is 18 / 18; the reference's other hexapod shape is 36 / 18,
// env/hexapod_closed_loop_env.hpp:20).  Same arithmetic, slots and workspaces as train_fwd_bwd_kernel (G:6889-23699 minus the weight
// gradients); what changes is who does what inside the workgroup:
//   * M = 2048 rows give exactly one row tile per CU and tower, so the only latency hiding available is a second wave per SIMD:
//     the two 256 x 256 products (second layer forward, its transpose backward) are split over K between wave w and wave w + 4
//     (same SIMD, same 64 output columns, k in [0,128) / [128,256)): one wave's weight stream, LDS reads and waits are covered by
//     the other's matrix instructions.  The two halves meet in LDS: each wave hands the other the 8 accumulator values it does
//     not finish, and finishes (bias + tanh / TanhGrad, LDS + workspace stores) the other 8 -- rows {0,1} / {2,3} of every
//     4-row group, so that the workspace stores stay 16 bytes per lane;
//   * the three thin products (first layer, policy head transposed; the policy head itself is an 8-way K split) give every
//     wave 32 output columns: half the epilogue per wave;
//   * the loss phase has one lane per (row, action) (two actions per lane when the action tile is 64 wide) instead of two serial
//     passes over 16 lanes per row.
// The padded widths of the observation and action tiles (KP0, AP: 32 or 64) are template parameters -- every loop is compile-time;
// the dense widths O and A are uniform run-time values.  Weights of the thin products and the first two ring stages of the second
// layer are requested at kernel entry behind the input loads (one workgroup of 8 waves per CU: 256 registers per wave).
#pragma once
#include "ppo_kernels.hpp"

#define T8_THREADS 512
#define T8_LD 260                 // 256 + LDS_PAD
// LDS carve (floats) for an observation tile of KP0 and an action tile of AP columns
template <int KP0, int AP>
struct T8L {
    static constexpr int LD0 = KP0 + LDS_PAD, LDM = AP + LDS_PAD;
    static constexpr int NPAR = 3 * 256 + 2 * AP + 4;                      // biases (2 x 256) | b_mu | logstd | w_v | b_v (NetDev::par_*)
    static constexpr int X0 = 0, H1 = X0 + 16 * LD0, H2 = H1 + 16 * T8_LD, D2 = H2 + 16 * T8_LD, D1 = D2 + 16 * T8_LD,
                         MU = D1 + 16 * T8_LD, PAR = MU + 16 * LDM, ACT = PAR + NPAR + 4, DLS = ACT + 16 * AP, ROWV = DLS + 16 * AP,
                         MISC = ROWV + 32, TOTAL = MISC + 128;
};

#ifndef T8_PRO
#define T8_PRO 1                  // where the second layer's first two ring stages are requested: 0 both in the prologue, 1 the second one behind the prologue's
                                  // barrier (it lands while the first layer runs; the prologue's request burst is 32 KB per wave shorter: 41.7 -> 40.6 us per
                                  // train step), 2 both behind the barrier (42.6 us: the first layer's end then waits for them)
#endif
#ifndef T8_W1T
#define T8_W1T 0                  // where the backward product's first two W1^T stages are requested: 0 before the K-halves' exchange, 1 after it, 2 one after it + one behind the epilogue's barrier
#endif
#ifndef T8_WT
#define T8_WT 1                   // workspace stores write-through (sc1): drain while the kernel computes instead of at its end
#endif
__device__ __forceinline__ void t8_st_wt2(float* p, float x, float y) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x, y};
    if constexpr (T8_WT != 0) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<float2*>(p) = make_float2(x, y);
}
// sum over the 32 lanes of a half-wave
__device__ __forceinline__ float t8_sum32(float v) { v = group16_sum(v); v += __shfl_xor(v, 16); return v; }

// one 32-deep stage of a [16 x 64] output chunk: 8 k-steps x 4 column tiles
__device__ __forceinline__ void t8_mma_stage(const WFrag<4, 2>& f, f32x4 (&acc)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float av[4] = {f.a[q].x, f.a[q].y, f.a[q].z, f.a[q].w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], f.v[4 * q + s][j], acc[j], 0, 0, 0);
    }
}
// a (16 KS)-deep product into 16 CT output columns per wave (CT interleaved column tiles)
template <int CT, int KS>
__device__ __forceinline__ void t8_mma_small(const WFrag<CT, KS>& f, const float* Xs, int ldx, int c, int g, f32x4 (&acc)[CT]) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        const float4 a4 = *reinterpret_cast<const float4*>(Xs + c * ldx + 16 * q + 4 * g);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], f.v[4 * q + s][j], acc[j], 0, 0, 0);
    }
}

// The K-split 256 x 256 product of wave (p = column block, kh = K half): Y[16, 64p..64p+63] partial over k in [128 kh, 128 kh + 128).
// fr[0], fr[1] hold the first two stages (requested by the caller); returns the partial accumulators.  `rot` (wave-uniform, 0..3)
// rotates the order in which the four 32-deep stages are taken: the 32 workgroups of an XCD stream the SAME weights, and in
// lockstep they would all ask one 32 KB block of L2 at a time; rotated by row tile they spread over the whole matrix.
#ifndef T8_ROTATE
#define T8_ROTATE 0                // measured: 42.0 us per train step rotated against 41.2 (no L2 hot spot to avoid; dynamic stage addresses cost)
#endif
__device__ __forceinline__ int t8_stage_row(int st, int rot) { return T8_ROTATE ? 32 * ((st + rot) & 3) : 32 * st; }
__device__ __forceinline__ void t8_big_product(WFrag<4, 2> (&fr)[3], const float* Wk /* W + 128 kh rows */, const WOff<2>& off, const float* Xs /* tile + 128 kh */,
                                               int c, int g, int rot, f32x4 (&acc)[4]) {
    auto load_a = [&](WFrag<4, 2>& f, int st) __attribute__((always_inline)) {
        const int k0 = t8_stage_row(st, rot);
#pragma unroll
        for (int q = 0; q < 2; ++q) f.a[q] = *reinterpret_cast<const float4*>(Xs + c * T8_LD + k0 + 16 * q + 4 * g);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_a(fr[0], 0); load_a(fr[1], 1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        __builtin_amdgcn_sched_barrier(0);
#ifdef T8_NOLOAD
        if (st + 2 < 4) load_a(fr[(st + 2) % 3], st + 2);      // timing experiment only: results are wrong
#else
        if (st + 2 < 4) { load_a(fr[(st + 2) % 3], st + 2); load_w_stage<4, 2>(fr[(st + 2) % 3], Wk + (size_t)t8_stage_row(st + 2, rot) * 256, off); }
#endif
        t8_mma_stage(fr[st % 3], acc);
#ifndef T8_NOSCHED
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the two K halves meet: wave kh hands rows {2,3} (kh = 0) / {0,1} (kh = 1) of every 4-row group to its partner and returns the
// finished values of the rows it keeps: out[j][i] = row 4g + 2 kh + i, column 64 p + 4 c + j  (kh0 + kh1 in that order on both sides)
__device__ __forceinline__ void t8_exchange(const f32x4 (&acc)[4], float* xch, int p, int kh, int lane, float (&out)[4][2]) {
    float* mine = xch + ((p * 2 + kh) * 64 + lane) * 8;
    const float* theirs = xch + ((p * 2 + (kh ^ 1)) * 64 + lane) * 8;
    const int give = kh ? 0 : 2;                    // first of the two rows handed over
    *reinterpret_cast<float4*>(mine) = make_float4(acc[0][give], acc[1][give], acc[2][give], acc[3][give]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(acc[0][give + 1], acc[1][give + 1], acc[2][give + 1], acc[3][give + 1]);
    lds_barrier();
    const float4 t0 = *reinterpret_cast<const float4*>(theirs), t1 = *reinterpret_cast<const float4*>(theirs + 4);
    const float th[2][4] = {{t0.x, t0.y, t0.z, t0.w}, {t1.x, t1.y, t1.z, t1.w}};
    const int keep = kh ? 2 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) out[j][i] = kh ? th[i][j] + acc[j][keep + i] : acc[j][keep + i] + th[i][j];
}

// The kernel's body.  The big workspaces leave write-through (T8_WT: they drain while the kernel computes); the small outputs -- x0g, dmug, slots -- are plain
// stores left to the kernel boundary.  (Round 5 also called this body as the first phase of a fused train + weight-gradient launch; measured slower twice and removed
// in round 6: profiles/r05_a_*, branch experiments-r05.)  bx / by / gx: the launch's block index (x, y) and grid width.
template <int KP0, int AP>
__device__ __forceinline__ void train8_body(const NetDev& net, const TrainArgs& a, float* lds, const unsigned bx, const unsigned by, const unsigned gx) {
    typedef T8L<KP0, AP> L8;
    constexpr int KS0 = KP0 / 16, KSA = AP / 16;       // k-steps of 16 in the first layer / in the head's backward product
    constexpr int CTA = AP / 16;                       // column tiles of the policy head (all AP columns in every wave)
    constexpr int NX = KP0 / 32, NA = AP / 32;         // observation / action tile elements per thread ([16][KP0] and [16][AP] over 512 threads)
    int tower = (int)by;
    int rb = (gx % 8 == 0) ? (int)((bx % 8) * (gx / 8) + bx / 8) : (int)bx;
    if (a.xcd_map == 1 && gx % 4 == 0) {      // see train_fwd_bwd_kernel: XCD x = tower x & 1, row tiles of split x >> 1
        const int lid = (int)(bx + gx * by), x = lid & 7, q = lid >> 3;
        tower = x & 1;
        rb = (x >> 1) * (int)(gx / 4) + q;
    }
    tower = uni(tower); rb = uni(rb);
    const int row0 = rb * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int p = wave & 3, kh = wave >> 2;
    const int rot = uni(rb & 3);
    const int nO = uni(net.O), nA = uni(net.A);
    float* slot = a.slots[tower] + (size_t)rb * net.slot_w;
    float* par = lds + L8::PAR;
    STAMP(0);
#ifdef PPO_STAMPS
    { const float probe = a.hyper[0]; asm volatile("" :: "v"(probe)); }      // (diagnostic builds: the kernarg round trip ends here)
    STAMP(20);
#endif
    // ---- prologue: every input load, then the weights the thin products need and the first two ring stages -------------------------
    const float* par_src = a.par + tower * net.par_total;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < L8::NPAR / 4) pv = reinterpret_cast<const float4*>(par_src)[tid];         // biases | b_mu | logstd | w_v | b_v
    const int er = tid >> 5, ej = tid & 31;                                             // this thread's element(s) of a [16][32 q + ej] tile
    const bool erow_live = row0 + er < a.n;
    float ov[NX], av_[NA];
#pragma unroll
    for (int q = 0; q < NX; ++q) { ov[q] = 0.f; if (erow_live && ej + 32 * q < nO) ov[q] = a.obs[(size_t)(row0 + er) * nO + ej + 32 * q]; }
#pragma unroll
    for (int q = 0; q < NA; ++q) { av_[q] = 0.f; if (tower == 0 && erow_live && ej + 32 * q < nA) av_[q] = a.actions[(size_t)(row0 + er) * nA + ej + 32 * q]; }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, s0 = 0.f, s1 = 1.f;
    const bool explicit_adv = a.advs != nullptr;
    if (tid < 16 && row0 + tid < a.n) {
        const int src = row0 + tid;
        if (tower == 0) {
            r0 = explicit_adv ? a.advs[src] : a.returns[src]; r2 = a.old_neglogp[src];
            if (!explicit_adv) { r1 = a.old_values[src]; s0 = a.adv_stats[0]; s1 = a.adv_stats[1]; }
        } else { r0 = a.returns[src]; r2 = a.old_values[src]; }
    }
    const float* W0 = uni(a.theta + net.w_off[tower][0]);
    const float* W1 = uni(a.theta + net.w_off[tower][1]) + (size_t)(128 * kh) * 256;
    const float* W1T = uni(a.thetaT + net.wT_off[tower][1]) + (size_t)(128 * kh) * 256;
    const WOff<2> off_big = make_woff<2>(256, g, 64 * p + 4 * c);
    WFrag<2, KS0> wl0;
    WFrag<CTA, 2> whd;
    WFrag<2, KSA> whT;
    WFrag<4, 2> fr[3];
    load_w_stage<2, KS0>(wl0, W0, make_woff<KS0>(256, g, 32 * wave + 2 * c));
#if T8_PRO < 2
    load_w_stage<4, 2>(fr[0], W1 + (size_t)t8_stage_row(0, rot) * 256, off_big);
#endif
#if T8_PRO == 0
    load_w_stage<4, 2>(fr[1], W1 + (size_t)t8_stage_row(1, rot) * 256, off_big);
#endif
    if (tower == 0) {
        load_w_stage<CTA, 2>(whd, uni(a.theta + net.wmu_off) + (size_t)(32 * wave) * AP, make_woff<2>(AP, g, CTA * c));   // rows 32w .. 32w+31 of W_mu [256][AP]
        load_w_stage<2, KSA>(whT, uni(a.thetaT + net.wmuT_off), make_woff<KSA>(256, g, 32 * wave + 2 * c));               // W_mu^T [AP][256]
    }
    STAMP(21);
    // consume the inputs
    if (tid < L8::NPAR / 4) *reinterpret_cast<float4*>(par + 4 * tid) = pv;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        lds[L8::X0 + er * L8::LD0 + ej + 32 * q] = ov[q];
        if (tower == 0) st_wt<false>(a.x0g + (size_t)(row0 + er) * KP0 + ej + 32 * q, ov[q]);         // rows >= n and padding columns: zeros
    }
    if (tower == 0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) lds[L8::ACT + er * AP + ej + 32 * q] = av_[q];
    }
    if (tid < 16) {
        float v0 = r0;
        if (tower == 0 && !explicit_adv) v0 = ((r0 - r1) - s0) / s1;                   // ppo2.hpp:401-406
        const bool live = row0 + tid < a.n;
        lds[L8::ROWV + 2 * tid] = live ? v0 : 0.f;
        lds[L8::ROWV + 2 * tid + 1] = live ? r2 : 0.f;
    }
    STAMP(22);
    lds_barrier();
    STAMP(1);
#if T8_PRO >= 2
    load_w_stage<4, 2>(fr[0], W1 + (size_t)t8_stage_row(0, rot) * 256, off_big);
#endif
#if T8_PRO >= 1
    load_w_stage<4, 2>(fr[1], W1 + (size_t)t8_stage_row(1, rot) * 256, off_big);       // behind the barrier: lands while the first layer runs
#endif
    // ---- first layer: h1 = tanh(x W0 + b0), 32 columns per wave ---------------------------------------------------------------------
    {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        t8_mma_small<2, KS0>(wl0, lds + L8::X0, L8::LD0, c, g, acc);
        const int col = 32 * wave + 2 * c;
        const float b0 = par[net.par_b[0] + col], b1 = par[net.par_b[0] + col + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float y0 = fast_tanh(acc[0][r] + b0), y1 = fast_tanh(acc[1][r] + b1);
            if (row0 + row >= a.n) { y0 = 0.f; y1 = 0.f; }
            *reinterpret_cast<float2*>(lds + L8::H1 + row * T8_LD + col) = make_float2(y0, y1);
            t8_st_wt2(a.hg[tower][0] + (size_t)(row0 + row) * 256 + col, y0, y1);
        }
    }
    lds_barrier();
    STAMP(2);
    // ---- second layer: h2 = tanh(h1 W1 + b1), K split over the wave pair -------------------------------------------------------------
    {
        f32x4 acc[4];
        t8_big_product(fr, W1, off_big, lds + L8::H1 + 128 * kh, c, g, rot, acc);
        STAMP(11);
        // the backward product's transposed weights: two stages per wave, in flight through the head and the loss
#if T8_W1T == 0
        load_w_stage<4, 2>(fr[0], W1T + (size_t)t8_stage_row(0, rot) * 256, off_big);
        load_w_stage<4, 2>(fr[1], W1T + (size_t)t8_stage_row(1, rot) * 256, off_big);
#endif
        STAMP(12);
        float out[4][2];
        t8_exchange(acc, lds + L8::D2, p, kh, lane, out);
        STAMP(13);
#if T8_W1T >= 1
        load_w_stage<4, 2>(fr[0], W1T + (size_t)t8_stage_row(0, rot) * 256, off_big);
#endif
#if T8_W1T == 1
        load_w_stage<4, 2>(fr[1], W1T + (size_t)t8_stage_row(1, rot) * 256, off_big);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const int col = 64 * p + 4 * c;
        const float4 b4 = *reinterpret_cast<const float4*>(par + net.par_b[1] + col);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * g + 2 * kh + i;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (row0 + row < a.n) ? fast_tanh(out[j][i] + bb[j]) : 0.f;
            *reinterpret_cast<float4*>(lds + L8::H2 + row * T8_LD + col) = make_float4(y[0], y[1], y[2], y[3]);
            if (tower == 0) st_wt4<T8_WT != 0>(a.hg[0][1] + (size_t)(row0 + row) * 256 + col, make_float4(y[0], y[1], y[2], y[3]));   // (the value head's weight gradient is formed here: nobody reads a copy of its input)
        }
    }
    lds_barrier();
    STAMP(3);
#if T8_W1T == 2
    load_w_stage<4, 2>(fr[1], W1T + (size_t)t8_stage_row(1, rot) * 256, off_big);
    __builtin_amdgcn_sched_barrier(0);
#endif
    const float cr = a.hyper[1];
    const float* h2 = lds + L8::H2;
    float* d2 = lds + L8::D2;
    float* misc = lds + L8::MISC;
    if (tower == 0) {
        // ---- policy head: mu = h2 W_mu + b_mu, K split 8 ways, partial tiles meet in LDS (the d2 and d1 tiles are free until the
        // backward products: the exchange of the second layer is behind the barrier above)
        float* scratch = lds + L8::D2;                                                 // [8][16][AP]: 16 or 32 KB of the 33 KB the two tiles span
        static_assert(8 * 16 * AP <= 2 * 16 * T8_LD, "split-K scratch of the policy head");
        {
            f32x4 acc[CTA];
#pragma unroll
            for (int j = 0; j < CTA; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            t8_mma_small<CTA, 2>(whd, h2 + 32 * wave, T8_LD, c, g, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = scratch + wave * (16 * AP) + (4 * g + r) * AP + CTA * c;
                if constexpr (CTA == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[0][r], acc[1][r]);
                else *reinterpret_cast<float4*>(dst) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            }
        }
        lds_barrier();
        float mu[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            float s = scratch[er * AP + ej + 32 * q];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += scratch[w * (16 * AP) + er * AP + ej + 32 * q];
            mu[q] = s + par[net.par_bmu + ej + 32 * q];
        }
        STAMP(6);                                                                      // (the scratch lies over d2, which the head's backward product writes: behind the barrier that follows the d mu tile)
        // ---- surrogate loss (G:9428-11290) and its gradient (G:12609-22656): one lane per (row, action) ---------------------------
        const bool live = erow_live;
        float z[NA], sigma[NA], zz = 0.f, sl_ = 0.f, se_ = 0.f;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const bool lj = ej + 32 * q < nA;
            const float logstd = mu[q] * 0.0f + par[net.par_ls + ej + 32 * q];
            sigma[q] = expf(logstd);
            const float act = live ? lds[L8::ACT + er * AP + ej + 32 * q] : mu[q];
            z[q] = (act - mu[q]) / sigma[q];
            zz += lj ? z[q] * z[q] : 0.f; sl_ += lj ? logstd : 0.f; se_ += lj ? logstd + HALF_LOG_2PIE : 0.f;
        }
        const float ssq = t8_sum32(zz), slog = t8_sum32(sl_), sent = t8_sum32(se_);
        const float nlp = 0.5f * ssq + HALF_LOG_2PI * (float)nA + slog;
        const float adv = live ? lds[L8::ROWV + 2 * er] : 0.f;
        const float old_nlp = live ? lds[L8::ROWV + 2 * er + 1] : nlp;
        const float lo = 1.0f - cr, hi = 1.0f + cr;
        const float ratio = expf(old_nlp - nlp);
        const float rmin = tf_min(ratio, hi);
        const float rclip = tf_max(rmin, lo);
        const float m1 = -adv * ratio, m2 = -adv * rclip;
        const float gi = a.inv_n;
        const float sel = (m1 >= m2) ? 1.0f : 0.0f;                                   // Maximum tie rule G:12609
        const float pass = ((rmin >= lo) ? 1.0f : 0.0f) * ((ratio <= hi) ? 1.0f : 0.0f); // G:15357, 16113
        float d_ratio = (-adv) * gi * sel;
        d_ratio += (-adv) * gi * (1.0f - sel) * pass;
        const float d_nlp = live ? -(d_ratio * ratio) : 0.0f;
        if (ej == 0) {
            const float dk = nlp - old_nlp;
            misc[er * 4 + 0] = live ? tf_max(m1, m2) : 0.f;
            misc[er * 4 + 1] = live ? sent : 0.f;
            misc[er * 4 + 2] = live ? dk * dk : 0.f;
            misc[er * 4 + 3] = (live && fabsf(ratio - 1.0f) > cr) ? 1.0f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            float dmu = 0.f, dl = 0.f;
            if (ej + 32 * q < nA && live) {
                dl = d_nlp * (1.0f - z[q] * z[q]) - net.ent_coef * gi;                 // AddN_2 G:21299
                dmu = d_nlp * (-(z[q] / sigma[q])) + dl * 0.0f;                        // AddN_3 G:22656
            }
            lds[L8::MU + er * L8::LDM + ej + 32 * q] = dmu;                            // the d mu tile: A operand of the head's backward product
            lds[L8::DLS + er * AP + ej + 32 * q] = dl;
            st_wt<false>(a.dmug + (size_t)(row0 + er) * AP + ej + 32 * q, dmu);                      // dead rows / padding columns: zeros
        }
        lds_barrier();
        if (tid < AP) {
            float sb = 0.f, sl = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { sb += lds[L8::MU + q * L8::LDM + tid]; sl += lds[L8::DLS + q * AP + tid]; }
            st_wt<false>(slot + net.slot_head + tid, sb);
            st_wt<false>(slot + net.slot_aux + tid, sl);
        } else if (tid < AP + 4) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += misc[q * 4 + (tid - AP)];
            st_wt<false>(slot + net.slot_loss + (tid - AP), s);
        }
        STAMP(7);
        // ---- dY1 = (d mu W_mu^T) .* (1 - h2^2), 32 columns per wave -----------------------------------------------------------------
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            t8_mma_small<2, KSA>(whT, lds + L8::MU, L8::LDM, c, g, acc);
            const int col = 32 * wave + 2 * c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const float2 hh = *reinterpret_cast<const float2*>(h2 + row * T8_LD + col);
                const float y0 = acc[0][r] * (1.0f - hh.x * hh.x), y1 = acc[1][r] * (1.0f - hh.y * hh.y);     // TanhGrad; dead rows: d mu = 0
                *reinterpret_cast<float2*>(d2 + row * T8_LD + col) = make_float2(y0, y1);
                t8_st_wt2(a.dyg[0][1] + (size_t)(row0 + row) * 256 + col, y0, y1);
            }
        }
    } else {
        // ---- value head + clipped value loss (G:10213-10837) and its gradient (G:14975-19571): 32 lanes per row ---------------------
        const float* wv = par + net.par_wv;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(h2[er * T8_LD + ej + 32 * k], wv[ej + 32 * k], s);
        const float v = t8_sum32(s) + par[net.par_bv];
        float dv = 0.f, lossv = 0.f;
        if (erow_live) {
            const float R = lds[L8::ROWV + 2 * er], vo = lds[L8::ROWV + 2 * er + 1];
            const float dvo = v - vo;
            const float vmin = tf_min(dvo, cr);
            const float vclip = vo + tf_max(vmin, -cr);
            const float e1 = v - R, e2 = vclip - R;
            const float q1 = e1 * e1, q2 = e2 * e2;
            lossv = tf_max(q1, q2);
            const float gv = net.vf_coef * 0.5f * a.inv_n;
            const float selv = (q1 >= q2) ? 1.0f : 0.0f;                                       // G:14975
            const float passv = ((vmin >= -cr) ? 1.0f : 0.0f) * ((dvo <= cr) ? 1.0f : 0.0f);   // G:17477, 18071
            dv = gv * selv * (2.0f * e1) + gv * (1.0f - selv) * (2.0f * e2) * passv;           // AddN_1 G:19571
        }
        if (ej == 0) { misc[er] = dv; misc[16 + er] = lossv; }
        lds_barrier();
        STAMP(7);
        if (tid == 0) {
            float sb = 0.f, sl = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { sb += misc[q]; sl += misc[16 + q]; }
            st_wt<false>(slot + net.slot_aux, sb);            // db_v
            st_wt<false>(slot + net.slot_loss, sl);           // sum of max((v-R)^2, (vclip-R)^2)
        }
        // dW_v[k] = sum_rows h2[row,k] dv[row] ; dY1 = dv (x) w_v .* (1 - h2^2): thread (k = tid & 255, rows 8 (tid >> 8) .. + 7)
        {
            const int k = tid & 255, q0 = (tid >> 8) * 8;
            const float w = wv[k];
            if (tid < 256) {
                float sw = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) sw = fmaf(h2[q * T8_LD + k], misc[q], sw);
                st_wt<false>(slot + net.slot_head + k, sw);
            }
#pragma unroll
            for (int q = q0; q < q0 + 8; ++q) {
                const float h = h2[q * T8_LD + k];
                const float d = (misc[q] * w) * (1.0f - h * h);
                d2[q * T8_LD + k] = d;
                st_wt<T8_WT != 0>(a.dyg[1][1] + (size_t)(row0 + q) * 256 + k, d);             // dead rows: d == 0
            }
        }
    }
    lds_barrier();
    STAMP(8);
    // ---- dY0 = (dY1 W1^T) .* (1 - h1^2), K split over the wave pair (the h2 tile is free now: exchange scratch) ------------------------
    {
        f32x4 acc[4];
        t8_big_product(fr, W1T, off_big, d2 + 128 * kh, c, g, rot, acc);
        STAMP(14);
        float out[4][2];
        t8_exchange(acc, lds + L8::H2, p, kh, lane, out);
        STAMP(15);
        const int col = 64 * p + 4 * c;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * g + 2 * kh + i;
            const float4 hh = *reinterpret_cast<const float4*>(lds + L8::H1 + row * T8_LD + col);
            const float4 y = make_float4(out[0][i] * (1.0f - hh.x * hh.x), out[1][i] * (1.0f - hh.y * hh.y), out[2][i] * (1.0f - hh.z * hh.z), out[3][i] * (1.0f - hh.w * hh.w));
            *reinterpret_cast<float4*>(lds + L8::D1 + row * T8_LD + col) = y;
            st_wt4<T8_WT != 0>(a.dyg[tower][0] + (size_t)(row0 + row) * 256 + col, y);
        }
    }
    lds_barrier();
    STAMP(9);
    // ---- bias gradients: db1 = sum_rows dY1 (threads 0..255), db0 = sum_rows dY0 (threads 256..511)  (.../Add_grad/Sum_1) -------------
    {
        const int k = tid & 255;
        const float* src = (tid < 256) ? d2 : lds + L8::D1;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += src[q * T8_LD + k];
        st_wt<false>(slot + net.slot_db[tid < 256 ? 1 : 0] + k, s);
    }
    STAMP(10);
}

template <int KP0, int AP>
__global__ __launch_bounds__(T8_THREADS) void train8_kernel(NetDev net, TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(TrainArgs)>();
    train8_body<KP0, AP>(net, a, lds, blockIdx.x, blockIdx.y, gridDim.x);
}
