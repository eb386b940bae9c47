// ppo_kernels.hpp -- gfx950 (MI355X, CDNA4) device code of libppo_hip.so.
//
// Hand-written HIP for the PPO rollout-collect + minibatch-update hot path of ppo_cpp.  All dense layers run on
// the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, bit-for-bit fp32) with the
// activation tile of a 16-row block resident in LDS and the weights streamed straight from L2 into VGPRs
// (each weight element is used by exactly one wave of a block, so an LDS round trip would be pure overhead).
//
// Kernel inventory (reference arithmetic each one replaces, paths relative to the reference root; "G" is the
// TF graph resources/ppo_cl/graphs/ppo_cpp_[4_5]_lr_0.0004_cr_0.1610_ent_0.0007.meta.txt):
//   policy_step_kernel   act model forward + sampling + neglogp        G:1859-6866, ppo2/policies.hpp:33-77
//   train_fwd_bwd_kernel train model forward, loss, dLoss/dactivations G:6889-23699 (all but the weight grads)
//   weight_grad_kernel   dW = X^T dY for every layer (split-K slabs)    G: .../MatMul_grad/MatMul_1 nodes
//   grad_reduce_kernel   slab/slot reduction, bias/logstd grads, losses G: .../Add_grad/Sum_1, loss/* Mean nodes
//   adam_kernel          global-norm clip + TF-1.14 ApplyAdam           G:23738-25392, 25426-25704, 30430-31383
//   epoch_prepare_kernel shuffle -> gather index, advantage statistics  ppo2/ppo2.hpp:274-307, 401-406
//   gae_kernel           GAE(lambda) / returns scan                     ppo2/runner.hpp:159-191
//   running_stats_kernel / reward_norm_kernel  VecNormalize numerics    env/env_normalize.hpp:64-116,
//                                                                       common/running_statistics.hpp:26-104
//   seeded_env_kernel    on-device synthetic env (bench / parity data)  stands in for env/env_mock.hpp
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ppo_peer.hpp"      // PeerDev: the data-parallel statistics exchange rides in norm_batch_kernel / norm_finalize_kernel

#define PPO_MAX_LAYERS 8
#define ROWS_PER_BLOCK 16          // one 16x16x4 MFMA row tile per workgroup
#define BLOCK_THREADS 256          // 4 waves, one per SIMD
#define LDS_PAD 4                  // row padding (floats): keeps float4 alignment, spreads rows over banks

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// Device-visible network description (padded internal layout).
// ------------------------------------------------------------------------------------------------------------
struct NetDev {
    int O, A, L;
    int Kp0;                        // obs dim padded to 16
    int Ap;                         // act dim padded to 16
    int H[PPO_MAX_LAYERS];
    int Hp[PPO_MAX_LAYERS];         // hidden dims padded to 16
    int w_off[2][PPO_MAX_LAYERS];   // [tower][layer] offsets into the padded parameter vector (tower 0 = pi, 1 = vf)
    int b_off[2][PPO_MAX_LAYERS];
    int wv_off, bv_off, wmu_off, bmu_off, ls_off;
    int wT_off[2][PPO_MAX_LAYERS];  // offsets into the transposed-copy buffer: W_l^T [Hp_l][Hp_{l-1}] for l >= 1
    int wmuT_off;                   // W_mu^T [Ap][Hp_{L-1}]
    int n_theta, n_thetaT;          // float counts of the padded parameter vector / the transposed-copy buffer
    // LDS carve (floats)
    int lds_h[PPO_MAX_LAYERS + 1];  // [0] = input tile, [l+1] = h_{l+1}
    int lds_d[2];                   // ping-pong gradient tiles
    int lds_mu;                     // head tile [16][Ap+PAD]
    int lds_head;                   // split-K scratch of the policy head [4][16][Ap]
    int lds_par;                    // small parameters staged once per block: biases | b_mu | logstd | w_v | b_v
    int par_b[PPO_MAX_LAYERS], par_bmu, par_ls, par_wv, par_bv, par_total;
    int par_skip;                   // first mirror index kept in LDS (wide form: biases stay in the global mirror)
    int wide;                       // two ping-pong LDS tiles instead of one tile per layer
    int lds_misc;                   // loss scratch: 64 + 16*Ap (dlogstd) + 16*Ap (actions) + 64
    int lds_total;
    // per-workgroup slot layout (floats) -- partial sums a row block contributes to non-matrix gradients
    int slot_db[PPO_MAX_LAYERS];    // bias grads of layer l        [Hp[l]]
    int slot_head;                  // pi: db_mu [Ap] ; vf: dW_v [Hp[L-1]]
    int slot_aux;                   // pi: dlogstd [Ap] ; vf: db_v [1]
    int slot_loss;                  // pi: pg, entropy, kl, clipfrac ; vf: vf_loss
    int slot_w;                     // slot width
    float ent_coef, vf_coef;
};

struct NormDev {                    // EnvNormalize state for one statistics object
    float* mean;                    // [D]
    float* var;                     // [D]
    double* count;                  // [1]
};

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tf_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float tf_max(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// counter-based hash shared with the oracle's seeded env (oracle/ppo_oracle.c orc_hash)
__device__ __forceinline__ uint32_t ctr_hash(uint32_t seed, uint32_t env, uint32_t step, uint32_t lane) {
    const uint64_t a = ((uint64_t)seed << 32) | (uint64_t)env;
    const uint64_t b = ((uint64_t)step << 32) | (uint64_t)lane;
    return (uint32_t)(splitmix64_dev(splitmix64_dev(a) ^ b) >> 32);
}
__device__ __forceinline__ float u32_to_sym_unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }

// N(0,1) from two counter hashes (Box-Muller); used only when the caller passes no explicit noise
__device__ __forceinline__ float ctr_normal(uint32_t seed, uint32_t row, uint32_t step, uint32_t j) {
    const uint32_t h1 = ctr_hash(seed ^ 0xA5A5A5A5u, row, step, 2u * j);
    const uint32_t h2 = ctr_hash(seed ^ 0xA5A5A5A5u, row, step, 2u * j + 1u);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0,1]
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);              // [0,1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

// tanh on the hardware exp2 / rcp units: 1 - 2/(e^{2|x|}+1) with the sign restored, and the odd Taylor polynomial for
// |x| < 1/16 where that form would cancel.  Absolute error <= 2 ulp of 1.0 (~1.2e-7) everywhere, ~10 instructions
// (ocml tanhf costs ~100 per element; a 256-wide layer epilogue has 16 per lane).
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x);
    const float x2 = x * x;
    const float p = fmaf(x2 * x, fmaf(x2, 0.13333333333333333f, -0.33333333333333333f), x);   // x - x^3/3 + 2x^5/15
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);                          // e^{2|x|}
    const float t = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f), x);
    return ax < 0.0625f ? p : t;
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also fences global memory, i.e. emits s_waitcnt vmcnt(0):
// it would drain the weight prefetch that was just issued for the NEXT layer and wait for this layer's activation
// stores (which only the next kernel reads).  LDS traffic is ordered by lgkmcnt alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Write-through (agent scope, sc1) stores for data that only the NEXT kernel reads: as ordinary stores they would sit dirty
// in this XCD's L2 until the end-of-kernel write-back, which then delays the kernel boundary; written through they drain
// while the kernel computes.  It pays where a workgroup's output is a trickle beside its compute (the fused train kernels'
// workspaces: -3 us for 17 MB; the narrow path's partial vectors: -1 us) and COSTS where storing is most of the kernel: the
// slab stores of the weight-gradient kernel (+0.4 us), Adam's parameter / moment stores (+2.2 us), the bf16 GEMM's epilogue
// (+8 us per GEMM).  Selected per kernel below.
template <bool WT> __device__ __forceinline__ void st_wt(float* p, float v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool WT> __device__ __forceinline__ void st_wt4(float* p, float4 v) {
    if constexpr (WT) { const f32x4 x = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory"); }   // (s_nop: the data registers stay untouched until the store has read them)
    else *reinterpret_cast<float4*>(p) = v;
}
#ifndef PPO_WT_B
#define PPO_WT_B 0
#endif
#ifndef PPO_WT_C1
#define PPO_WT_C1 1                 // measured at config 3 (us per train step): none 45.67, B 46.05, C1 45.49, C2 47.86
#endif
#ifndef PPO_WT_C2
#define PPO_WT_C2 0
#endif

// Kernel arguments live in memory and the compiler fetches them on demand: a kernel with ~1 KB of by-value arguments
// (NetDev + the args struct) otherwise starts with 5-7 DEPENDENT scalar-load round trips (~700 cycles each, cold) before
// its first vector load goes out.  Touching one dword of every 64-byte line of the kernarg segment in one batch costs
// one round trip and leaves the segment in the scalar cache for the loads the compiler emits later.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    typedef const __attribute__((address_space(4))) unsigned* kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) acc |= ka[i * 16 < BYTES / 4 ? i * 16 : BYTES / 4 - 1];
    asm volatile("" :: "s"(acc));
}

// reduce over the 16 lanes that share (lane >> 4)
// DPP lane permutes (vector-ALU data path, ~4 cycles each) instead of __shfl_xor's ds_bpermute (an LDS-crossbar round
// trip each): pairs, quads, then the two mirrors -- every lane of the 16 ends with the same total.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// The 64-lane (32-lane) xor butterfly `for (o = 32 (16); o > 0; o >>= 1) v += __shfl_xor(v, o)` where only LANE 0 of the wave (lanes 0 and 32) uses the result: the
// levels 8, 4, 2, 1 as DPP row operations instead of trips through the LDS crossbar (~100 cycles of latency each, on the critical path of a kernel's tail).  Lane 0
// receives the same partners' partial sums in the same order -- row_shl:4 hands lane i the value of lane i + 4, which is the xor partner for the lanes below 4 of every
// 8 and garbage elsewhere, and everything lane 0 depends on afterwards comes from such lanes -- so it holds the same BITS (tools/ubench/lane0_sum.hip); other lanes do not.
__device__ __forceinline__ float half_sum_lane0(float v) {
    v += __shfl_xor(v, 16);
    v += dpp_move<0x128>(v);     // row_ror:8  (lane i <-> i ^ 8, exact in every lane)
    v += dpp_move<0x104>(v);     // row_shl:4  (lane i <- lane i + 4)
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_sum_lane0(float v) { v += __shfl_xor(v, 32); return half_sum_lane0(v); }
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror: lane i <-> 7 - i within each 8
    v += dpp_move<0x140>(v);     // row_mirror:      lane i <-> 15 - i within each 16
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// Dense product on one 16-row tile:   Y[16,Np] = epilogue( X[16,K] * W[K,Np] )
//   X  : LDS, row-major, leading dimension ldx (multiple of 4)
//   W  : global, row-major [K][ldw]   (K multiple of 16*KS, Np multiple of 16*CT)
//   Y  : LDS (ldy) and optionally global (gy, row stride ldg; rows >= nrows of the last tile are written as ZEROS so
//        that the weight-gradient kernel can run over whole 16-row tiles for any minibatch size)
// The forward layers use W = the layer's weights (epilogue bias + tanh); the backward pass uses W = the
// TRANSPOSED copy the Adam kernel keeps (epilogue TanhGrad), so both directions stream weights identically:
// each wave owns 16*CT output columns; MFMA j covers columns n0 + CT*c + j (c = lane & 15), so ONE 16-byte load per
// lane feeds CT matrix instructions and the epilogue writes 16-byte vectors.  The reduction index of instruction s
// for lane group g = lane >> 4 is k = kb + 4g + s: the A operand is ONE 16-byte LDS read per 16 k values.
// Weights are pipelined through two named register stages (ping-pong; a copy would make the compiler wait for the
// loads it has just issued); stage i+1's loads are pinned above stage i's MFMAs by sched_barrier.  The first stage
// of a layer is loaded by dense_prefetch() BEFORE the previous layer's epilogue and barrier (weights do not depend
// on activations), which takes one L2 round trip per layer off the critical path.
// ------------------------------------------------------------------------------------------------------------
#ifndef PPO_STAMP_LAYER
#define PPO_STAMP_LAYER 1
#endif
#ifdef PPO_STAMPS
#define DSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

template <int CT, int KS>
struct WFrag { float v[4 * KS][CT]; float4 a[KS]; };   // one pipeline stage: weights + the matching A-operand vectors

// Addressing: a stage's weights are rows kb + 16q + 4g + s of W.  The per-lane part (4g + 16q + s)*ldw + col is a 32-bit
// element offset computed ONCE per column chunk (WOff); the per-stage part kb*ldw is wave-uniform and lives in scalar
// registers, so a stage's loads are `global_load_dwordx4 v, voff, s[base]` with no vector address arithmetic between the
// matrix instructions (64-bit per-lane multiplies there cost ~700 cycles per stage, un-overlapped with the MFMAs).
template <int KS>
struct WOff { unsigned e[4 * KS]; };          // BYTE offsets

// make a value / pointer provably wave-uniform for the compiler (it then lives in SGPRs)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ const float* uni(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

template <int KS>
__device__ __forceinline__ WOff<KS> make_woff(int ldw, int g, int col) {
    WOff<KS> o;
#pragma unroll
    for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s) o.e[4 * q + s] = (unsigned)(((4 * g + 16 * q + s) * ldw + col) * 4);
    return o;
}

// Wk = W + kb*ldw (wave-uniform)
template <int CT, int KS>
__device__ __forceinline__ void load_w_stage(WFrag<CT, KS>& w, const float* __restrict__ Wk, const WOff<KS>& off) {
#pragma unroll
    for (int i = 0; i < 4 * KS; ++i) {
        // global address space + uniform base + 32-bit lane offset -> `global_load_dwordx4 v, voff, s[base:base+1]`
        typedef const __attribute__((address_space(1))) char* gchar;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) f32x4* gfloat4;
        typedef const __attribute__((address_space(1))) f32x2_t* gfloat2;
        typedef const __attribute__((address_space(1))) float* gfloat;
        gchar p = (gchar)(reinterpret_cast<unsigned long long>(Wk)) + off.e[i];
        if constexpr (CT == 4) {
            const f32x4 t = *(gfloat4)p;
            w.v[i][0] = t.x; w.v[i][1] = t.y; w.v[i][2] = t.z; w.v[i][3] = t.w;
        } else if constexpr (CT == 2) {
            const f32x2_t t = *(gfloat2)p;
            w.v[i][0] = t.x; w.v[i][1] = t.y;
        } else {
            w.v[i][0] = *(gfloat)p;
        }
    }
}

// rows krow + 16q + s (q < KS, s < 4) of W, CT consecutive columns starting at col  (head / one-off loads)
template <int CT, int KS>
__device__ __forceinline__ void load_w_rows(WFrag<CT, KS>& w, const float* __restrict__ W, int ldw, int krow, int col) {
#pragma unroll
    for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* p = W + (size_t)(krow + 16 * q + s) * ldw + col;
            if constexpr (CT == 4) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                w.v[4 * q + s][0] = t.x; w.v[4 * q + s][1] = t.y; w.v[4 * q + s][2] = t.z; w.v[4 * q + s][3] = t.w;
            } else if constexpr (CT == 2) {
                const float2 t = *reinterpret_cast<const float2*>(p);
                w.v[4 * q + s][0] = t.x; w.v[4 * q + s][1] = t.y;
            } else {
                w.v[4 * q + s][0] = *p;
            }
        }
}

#ifndef PPO_INTERLEAVE
#define PPO_INTERLEAVE 1
#endif
#ifndef PPO_WT_STORES
#define PPO_WT_STORES 1
#endif
#ifndef PPO_RING
#define PPO_RING 3                  // register stages in flight per wave: PPO_RING-1 stages (each 16*KS k deep) ahead
#endif
template <int CT, int KS>
struct WRing { WFrag<CT, KS> s[PPO_RING]; };

// the first PPO_RING-1 k stages of this wave's first column chunk (a wave without a chunk loads chunk 0: harmless);
// K = reduction depth of the layer (addresses clamped to the last stage)
template <int CT, int KS>
__device__ __forceinline__ void dense_prefetch(WRing<CT, KS>& w, const float* W, int ldw, int Np, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    W = uni(W); ldw = uni(ldw); K = uni(K);
    int n0 = wave * 16 * CT;
    if (n0 >= Np) n0 = 0;
    const WOff<KS> off = make_woff<KS>(ldw, lane >> 4, n0 + CT * (lane & 15));
#pragma unroll
    for (int i = 0; i < PPO_RING - 1; ++i)
        if (i * 16 * KS < K) load_w_stage<CT, KS>(w.s[i], W + (size_t)(i * 16 * KS) * ldw, off);
}

enum { EP_BIAS_TANH = 0, EP_TANHGRAD = 1 };

// epilogue of one 16 x (16*CT) output chunk: accumulator register r of MFMA j holds Y[row 4g + r][col + j], col = n0 + CT*c
template <int CT, int EP>
__device__ __forceinline__ void tile_epilogue(const f32x4 (&acc)[CT], int g, int col, const float* __restrict__ bias, const float* Hs, int ldh,
                                              float* Ys, int ldy, float* __restrict__ gy, int ldg, int row0, int nrows) {
    float bv[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) bv[j] = (EP == EP_BIAS_TANH) ? bias[col + j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float y[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if constexpr (EP == EP_BIAS_TANH) y[j] = fast_tanh(acc[j][r] + bv[j]);
            else { const float h = Hs[row * ldh + col + j]; y[j] = acc[j][r] * (1.0f - h * h); }   // TanhGrad
        }
        float* ys = Ys + row * ldy + col;
        float* yg = gy ? gy + (size_t)(row0 + row) * ldg + col : nullptr;
#ifdef PPO_NO_WS_STORES
        const bool wr = false;           // timing experiment only: results are wrong
#else
        const bool wr = gy != nullptr;
#endif
        if (wr && (row0 + row) >= nrows) {
#pragma unroll
            for (int j = 0; j < CT; ++j) y[j] = 0.f;
        }
        if constexpr (CT == 4) {
            *reinterpret_cast<float4*>(ys) = make_float4(y[0], y[1], y[2], y[3]);
            if (wr) {
#if PPO_WT_STORES
                // write-through (sc1) 16-byte store: the activations are only read by the NEXT kernel, so they should
                // drain to memory while this kernel computes instead of piling up as dirty L2 lines that the
                // end-of-kernel release has to write back (~3 us at the boundary for 17 MB)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const f32x4 fv = {y[0], y[1], y[2], y[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fv), __builtin_amdgcn_make_buffer_rsrc(gy, 0, 0x7fffffff, 0x00020000),
                                                       (int)(((size_t)(row0 + row) * ldg + col) * 4), 0, 16);
#else
                *reinterpret_cast<float4*>(yg) = make_float4(y[0], y[1], y[2], y[3]);
#endif
            }
        } else if constexpr (CT == 2) {
            *reinterpret_cast<float2*>(ys) = make_float2(y[0], y[1]);
            if (wr) *reinterpret_cast<float2*>(yg) = make_float2(y[0], y[1]);
        } else {
            *ys = y[0];
            if (wr) *yg = y[0];
        }
    }
}

// A product whose reduction is exactly ONE pipeline stage deep (16*KS), computed from ring slot SLOT.  With SLOT =
// PPO_RING-1 the slots 0..PPO_RING-2 are free for the NEXT product's first stages, which can then be requested long
// before this one runs instead of in a burst right after its matrix instructions (every CU of an XCD asks for its first
// two stages at the same moment: ~2.6 k cycles of L2-bandwidth-bound issue).  Used for the policy head's backward
// product (K = padded action width): the transposed second-layer weights stream in under the loss arithmetic.
template <int CT, int KS, int EP, int SLOT>
__device__ __forceinline__ void dense_tile_one_stage(WRing<CT, KS>& w, const float* W, int ldw, const float* __restrict__ bias,
                                                     const float* Xs, int ldx, float* Ys, int ldy, int Np, const float* Hs, int ldh,
                                                     float* __restrict__ gy, int ldg, int row0, int nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int CW = 16 * CT;
    constexpr int NSTRIDE = (BLOCK_THREADS / 64) * CW;
    W = uni(W); ldw = uni(ldw); Np = uni(Np);
    bool first = true;
    for (int n0 = wave * CW; n0 < Np; n0 += NSTRIDE) {
        f32x4 acc[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int col = n0 + CT * c;
        WFrag<CT, KS>& wf = w.s[SLOT];
        if (!first) load_w_stage<CT, KS>(wf, W, make_woff<KS>(ldw, g, col));
        first = false;
#pragma unroll
        for (int q = 0; q < KS; ++q) wf.a[q] = *reinterpret_cast<const float4*>(Xs + c * ldx + 16 * q + 4 * g);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const float av[4] = {wf.a[q].x, wf.a[q].y, wf.a[q].z, wf.a[q].w};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], wf.v[4 * q + s_][j], acc[j], 0, 0, 0);
        }
        tile_epilogue<CT, EP>(acc, g, col, bias, Hs, ldh, Ys, ldy, gy, ldg, row0, nrows);
    }
}

// The same from a fragment OUTSIDE the ring that the caller loaded earlier (kernel entry): exactly one 16*CT-column chunk per
// wave (Np == 4 waves * 16 * CT), nothing is fetched here.
template <int CT, int KS, int EP>
__device__ __forceinline__ void dense_tile_frag(WFrag<CT, KS>& wf, const float* __restrict__ bias, const float* Xs, int ldx, float* Ys, int ldy,
                                                const float* Hs, int ldh, float* __restrict__ gy, int ldg, int row0, int nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int col = wave * 16 * CT + CT * c;
    f32x4 acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KS; ++q) wf.a[q] = *reinterpret_cast<const float4*>(Xs + c * ldx + 16 * q + 4 * g);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        const float av[4] = {wf.a[q].x, wf.a[q].y, wf.a[q].z, wf.a[q].w};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], wf.v[4 * q + s_][j], acc[j], 0, 0, 0);
    }
    tile_epilogue<CT, EP>(acc, g, col, bias, Hs, ldh, Ys, ldy, gy, ldg, row0, nrows);
}

// this wave's chunk of a one-stage product's weights -> an explicit fragment
template <int CT, int KS>
__device__ __forceinline__ void load_frag(WFrag<CT, KS>& wf, const float* W, int ldw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    W = uni(W); ldw = uni(ldw);
    load_w_stage<CT, KS>(wf, W, make_woff<KS>(ldw, lane >> 4, wave * 16 * CT + CT * (lane & 15)));
}

// this wave's first-chunk weights of a one-stage product -> ring slot SLOT
template <int CT, int KS, int SLOT>
__device__ __forceinline__ void prefetch_one_stage(WRing<CT, KS>& w, const float* W, int ldw, int Np) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    W = uni(W); ldw = uni(ldw);
    int n0 = wave * 16 * CT;
    if (n0 >= Np) n0 = 0;
    load_w_stage<CT, KS>(w.s[SLOT], W, make_woff<KS>(ldw, lane >> 4, n0 + CT * (lane & 15)));
}

template <int CT, int KS, int EP, class Between>
__device__ __forceinline__ void dense_tile(WRing<CT, KS>& w, const float* W, int ldw, const float* __restrict__ bias,
                                           const float* Xs, int ldx, int K, float* Ys, int ldy, int Np, const float* Hs, int ldh,
                                           float* __restrict__ gy, int ldg, int row0, int nrows, Between&& between,
                                           unsigned long long* dbg = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int CW = 16 * CT;
    constexpr int KB = 16 * KS;
    constexpr int NSTRIDE = (BLOCK_THREADS / 64) * CW;
    W = uni(W); ldw = uni(ldw); K = uni(K); Np = uni(Np);
    bool first = true;
    bool called = false;
    DSTAMP(0);
    for (int n0 = wave * CW; n0 < Np; n0 += NSTRIDE) {
        f32x4 acc[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int col = n0 + CT * c;
        auto compute = [&](const WFrag<CT, KS>& wf) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const float av[4] = {wf.a[q].x, wf.a[q].y, wf.a[q].z, wf.a[q].w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wf.v[4 * q + s][j], acc[j], 0, 0, 0);
                }
            }
        };
        // the A operand (this tile's activations, LDS) travels in the same ring as the weights: read PPO_RING-1 stages
        // ahead of its MFMAs, so no LDS latency sits between matrix instructions
        auto load_a = [&](WFrag<CT, KS>& wf, int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < KS; ++q) wf.a[q] = *reinterpret_cast<const float4*>(Xs + c * ldx + kb + 16 * q + 4 * g);
        };
        const WOff<KS> off = make_woff<KS>(ldw, g, col);
        if (!first) {
#pragma unroll
            for (int i = 0; i < PPO_RING - 1; ++i)
                if (i * KB < K) load_w_stage<CT, KS>(w.s[i], W + (size_t)(i * KB) * ldw, off);
        }
#pragma unroll
        for (int i = 0; i < PPO_RING - 1; ++i)
            if (i * KB < K) load_a(w.s[i], i * KB);
        first = false;
        // ring of PPO_RING named register stages (compile-time indices, no copies).  Steady state: whole rounds of
        // PPO_RING steps whose loads are all in range and UNCONDITIONAL, so the compiler waits with counted vmcnt;
        // sched_barrier pins each step's loads above its MFMAs.  The tail (at most 2*PPO_RING-2 steps) is straight-
        // line code that issues only the loads that exist: no stage is fetched twice.
        int kb = 0;
        for (; kb + (2 * PPO_RING - 1) * KB <= K; kb += PPO_RING * KB) {
#pragma unroll
            for (int i = 0; i < PPO_RING; ++i) {
                DSTAMP(4 + (kb / KB + i < 11 ? kb / KB + i : 11));
                const int kn = kb + (i + PPO_RING - 1) * KB;                          // wave-uniform, < K
                __builtin_amdgcn_sched_barrier(0);
                load_a(w.s[(i + PPO_RING - 1) % PPO_RING], kn);
                load_w_stage<CT, KS>(w.s[(i + PPO_RING - 1) % PPO_RING], W + (size_t)kn * ldw, off);
#if PPO_INTERLEAVE
                compute(w.s[i]);
                // one weight load after every CT matrix instructions: the wave never sits in a burst of VMEM issue
                // while its MFMA pipe runs dry (the loads are for stage i+RING-1, independent of these MFMAs)
                __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
#pragma unroll
                for (int t = 0; t < 4 * KS; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#else
                __builtin_amdgcn_sched_barrier(0);
                compute(w.s[i]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * PPO_RING - 2; ++i) {
            if (kb + i * KB < K) {
                const int kn = kb + (i + PPO_RING - 1) * KB;
                if (kn < K) {
                    load_a(w.s[(i + PPO_RING - 1) % PPO_RING], kn);
                    load_w_stage<CT, KS>(w.s[(i + PPO_RING - 1) % PPO_RING], W + (size_t)kn * ldw, off);
                }
                __builtin_amdgcn_sched_barrier(0);
                compute(w.s[i % PPO_RING]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DSTAMP(1);
        if (n0 + NSTRIDE >= Np) { between(); called = true; }       // next layer's first stage goes out before this epilogue
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(2);
        tile_epilogue<CT, EP>(acc, g, col, bias, Hs, ldh, Ys, ldy, gy, ldg, row0, nrows);
    }
    DSTAMP(3);
    if (!called) between();                                          // waves without a column chunk
}

// ------------------------------------------------------------------------------------------------------------
// Policy head  mu[16,Ap] = h[16,K] * Wmu[K,Ap] + b  with Ap = 16*CTH: too narrow to split over waves by columns, so
// the 4 waves split K instead (each K/4 deep, ALL its weights prefetched in one go), partial tiles meet in LDS.
// ------------------------------------------------------------------------------------------------------------
#define HEAD_KS 4          // 64 k values per wave and stage

template <int CTH>
__device__ __forceinline__ void head_prefetch(WFrag<CTH, HEAD_KS>& w, const float* __restrict__ Wmu, int Ap, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_w_rows<CTH, HEAD_KS>(w, Wmu, Ap, wave * (K / 4) + 4 * (lane >> 4), CTH * (lane & 15));
}

template <int CTH>
__device__ __forceinline__ void head_splitk(WFrag<CTH, HEAD_KS>& w, const float* __restrict__ Wmu, const float* __restrict__ bmu,
                                            const float* Hs, int ldh, int K, int Ap, float* scratch /* [4][16][Ap] */,
                                            float* mus, int ldm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int kq = K / 4, kbeg = wave * kq;
    f32x4 acc[CTH];
#pragma unroll
    for (int j = 0; j < CTH; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < kq; kb += 16 * HEAD_KS) {
        if (kb > 0) load_w_rows<CTH, HEAD_KS>(w, Wmu, Ap, kbeg + kb + 4 * g, CTH * c);
#pragma unroll
        for (int q = 0; q < HEAD_KS; ++q) {
            const float4 av4 = *reinterpret_cast<const float4*>(Hs + c * ldh + kbeg + kb + 16 * q + 4 * g);
            const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < CTH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], w.v[4 * q + s][j], acc[j], 0, 0, 0);
        }
    }
    float* mine = scratch + wave * (ROWS_PER_BLOCK * Ap);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < CTH; ++j) mine[(4 * g + r) * Ap + CTH * c + j] = acc[j][r];
    lds_barrier();
    constexpr int APC = 16 * CTH;                       // == Ap on this path: compile-time row / column split (no division)
    for (int i = threadIdx.x; i < ROWS_PER_BLOCK * APC; i += BLOCK_THREADS) {
        const int row = i / APC, col = i - row * APC;
        const float sum = ((scratch[i] + scratch[ROWS_PER_BLOCK * APC + i]) + scratch[2 * ROWS_PER_BLOCK * APC + i]) + scratch[3 * ROWS_PER_BLOCK * APC + i];
        mus[row * ldm + col] = sum + bmu[col];
    }
}

// generic (narrow-net) head: one 16-column chunk per wave, K serial
__device__ __forceinline__ void head_generic(const float* __restrict__ Wmu, const float* __restrict__ bmu, const float* Hs, int ldh, int K,
                                             int Ap, float* mus, int ldm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    for (int n0 = wave * 16; n0 < Ap; n0 += 64) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < K; kb += 16) {
            const float4 av4 = *reinterpret_cast<const float4*>(Hs + c * ldh + kb + 4 * g);
            const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], Wmu[(size_t)(kb + 4 * g + s) * Ap + n0 + c], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mus[(4 * g + r) * ldm + n0 + c] = acc[r] + bmu[n0 + c];
    }
}

// value head: v[row] = h[row,:] . w + b    (N = 1: VALU dot products, 16 lanes per row)
__device__ __forceinline__ float value_head(const float* Hs, int ldh, int Kp, const float* __restrict__ wv, float bv) {
    const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
    float s = 0.f;
    for (int k = part; k < Kp; k += 16) s = fmaf(Hs[row * ldh + k], wv[k], s);
    s = group16_sum(s);
    return s + bv;
}

struct ObsNorm { const float* mean; const float* var; float eps; float clip; int enabled; };

// Block prologue: everything a 16-row block reads that is not a weight matrix -- its input rows, the small parameters
// (biases, logstd, value head; a contiguous per-tower mirror `par_src` kept current by the Adam kernel) and, for the
// train kernel, the rows' actions and scalars -- is fetched with ALL loads issued before the first LDS store, i.e. in
// ONE memory round trip (separate load->store loops cost one dependent round trip each, ~1 us apiece, and each wait
// also drains the in-order queue behind the weight prefetch).
struct RowScalars { const float* actions; const float* v0a; const float* v0b; const float* v1; const float* stats; int mode; };
// mode 0: none ; 1 (policy): v0 = advs[src] (v0b null) or ((v0a - v0b) - stats[0]) / stats[1], v1 = old_neglogp
// mode 2 (value) : v0 = returns, v1 = old_values

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_issue` runs right after this block's loads have been issued and before the first of them is consumed: loads the
// caller issues there queue BEHIND the inputs (the inputs are needed first) but ahead of the wait, so they are in flight
// during the staging round trip.
template <class Hook = NoHook>
__device__ __forceinline__ void stage_block_inputs(const NetDev& net, const float* __restrict__ par_src, float* par, float* Xs, int ldx,
                                                   const float* __restrict__ obs, int row0, int nrows,
                                                   ObsNorm nz, float* __restrict__ obs_out, float* __restrict__ x0g, RowScalars rs,
                                                   float* acts, float* rowv, Hook&& after_issue = Hook()) {
    const int tid = threadIdx.x;
    const int Kp0 = net.Kp0, O = net.O, A = net.A, Ap = net.Ap;
    constexpr int PK = 4, OK = 2, AK = 2;
    float pv[PK], ov[OK], av[AK], r0 = 0.f, r1 = 0.f, r2 = 0.f, s0 = 0.f, s1 = 1.f;
    // element i = tid + 256 k of a [16][W] tile is (row, col) = (i / W, i % W): one division per tile width instead of
    // one per element (a 32-bit division is ~40 vector instructions, and this prologue is pure latency)
    const int orow0 = tid / Kp0, ocol0 = tid - orow0 * Kp0, odq = BLOCK_THREADS / Kp0, odr = BLOCK_THREADS - odq * Kp0;
    const int arow0 = tid / Ap, acol0 = tid - arow0 * Ap, adq = BLOCK_THREADS / Ap, adr = BLOCK_THREADS - adq * Ap;
    auto tile_rc = [](int row0_, int col0_, int dq, int dr, int W, int k, int& r, int& j) __attribute__((always_inline)) {
        r = row0_ + dq * k; j = col0_ + dr * k;
#pragma unroll
        for (int t = 0; t < k; ++t) if (j >= W) { j -= W; ++r; }
    };
    // ---- issue every load -----------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < PK; ++k) { const int i = net.par_skip + tid + BLOCK_THREADS * k; pv[k] = i < net.par_total ? par_src[i] : 0.f; }
#pragma unroll
    for (int k = 0; k < OK; ++k) {
        const int i = tid + BLOCK_THREADS * k;
        int r, j; tile_rc(orow0, ocol0, odq, odr, Kp0, k, r, j);
        const int row = row0 + r;
        ov[k] = 0.f;
        if (i < ROWS_PER_BLOCK * Kp0 && row < nrows && j < O) ov[k] = obs[(size_t)row * O + j];
    }
    if (rs.mode == 1) {
#pragma unroll
        for (int k = 0; k < AK; ++k) {
            const int i = tid + BLOCK_THREADS * k;
            int r, j; tile_rc(arow0, acol0, adq, adr, Ap, k, r, j);
            const int row = row0 + r;
            av[k] = 0.f;
            if (i < ROWS_PER_BLOCK * Ap && row < nrows && j < A) av[k] = rs.actions[(size_t)row * A + j];
        }
    }
    if (rs.mode && tid < ROWS_PER_BLOCK && row0 + tid < nrows) {
        const int src = row0 + tid;
        r0 = rs.v0a[src]; r2 = rs.v1[src];
        if (rs.v0b) { r1 = rs.v0b[src]; s0 = rs.stats[0]; s1 = rs.stats[1]; }
    }
    after_issue();
    // ---- consume ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < PK; ++k) { const int i = net.par_skip + tid + BLOCK_THREADS * k; if (i < net.par_total) par[i] = pv[k]; }
    for (int i = net.par_skip + tid + BLOCK_THREADS * PK; i < net.par_total; i += BLOCK_THREADS) par[i] = par_src[i];
    auto put_obs = [&](int r, int j, float x) __attribute__((always_inline)) {
        const int row = row0 + r;
        if (row < nrows && j < O) {
            if (nz.enabled) {        // env_normalize.hpp:99-104: (x - mean) * 1/sqrt(var + eps), then clamp
                x = (x - nz.mean[j]) * (1.0f / sqrtf(nz.var[j] + nz.eps));
                x = tf_min(tf_max(x, -nz.clip), nz.clip);
            }
            if (obs_out) obs_out[(size_t)row * O + j] = x;
        }
        Xs[r * ldx + j] = x;
        if (x0g) x0g[(size_t)row * Kp0 + j] = x;                    // rows >= nrows: zeros (whole tiles for the dW kernel)
    };
#pragma unroll
    for (int k = 0; k < OK; ++k) {
        const int i = tid + BLOCK_THREADS * k;
        int r, j; tile_rc(orow0, ocol0, odq, odr, Kp0, k, r, j);
        if (i < ROWS_PER_BLOCK * Kp0) put_obs(r, j, ov[k]);
    }
    for (int i = tid + BLOCK_THREADS * OK; i < ROWS_PER_BLOCK * Kp0; i += BLOCK_THREADS) {
        const int r = i / Kp0, j = i - r * Kp0, row = row0 + r;
        float x = 0.f;
        if (row < nrows && j < O) x = obs[(size_t)row * O + j];
        put_obs(r, j, x);
    }
    if (rs.mode == 1) {
#pragma unroll
        for (int k = 0; k < AK; ++k) { const int i = tid + BLOCK_THREADS * k; if (i < ROWS_PER_BLOCK * Ap) acts[i] = av[k]; }
        for (int i = tid + BLOCK_THREADS * AK; i < ROWS_PER_BLOCK * Ap; i += BLOCK_THREADS) {
            const int r = i / Ap, j = i - r * Ap, row = row0 + r;
            float x = 0.f;
            if (row < nrows && j < A) x = rs.actions[(size_t)row * A + j];
            acts[i] = x;
        }
    }
    if (rs.mode && tid < ROWS_PER_BLOCK) {
        float v0 = r0;
        if (rs.v0b) v0 = ((r0 - r1) - s0) / s1;                                  // ppo2.hpp:401-406
        const bool live = row0 + tid < nrows;
        rowv[2 * tid] = live ? v0 : 0.f;
        rowv[2 * tid + 1] = live ? r2 : 0.f;
    }
}

#ifdef PPO_STAMPS
#define STAMP(i)                                                                                        \
    do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)(tower * gridDim.x + rb) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

#define HALF_LOG_2PI 0.9189385175704956f   /* G:6531 */
#define HALF_LOG_2PIE 1.4189385175704956f  /* G:10021-10180 */

// policy head dispatch shared by the act and train kernels (CTH = Ap/16 for the split-K form, 0 = generic):
// fills mus[16][Ap+PAD]; ends with a barrier
template <int CTH> struct HeadFrag { WFrag<(CTH > 0 ? CTH : 1), HEAD_KS> w; };

template <int CTH>
__device__ __forceinline__ void policy_head(const NetDev& net, const float* __restrict__ theta, HeadFrag<CTH>& hpre, const float* hL, int ldh,
                                            int K, float* lds) {
    float* mus = lds + net.lds_mu;
    const int ldm = net.Ap + LDS_PAD;
    if constexpr (CTH > 0) head_splitk<CTH>(hpre.w, theta + net.wmu_off, lds + net.lds_par - net.par_skip + net.par_bmu, hL, ldh, K, net.Ap, lds + net.lds_head, mus, ldm);
    else head_generic(theta + net.wmu_off, lds + net.lds_par - net.par_skip + net.par_bmu, hL, ldh, K, net.Ap, mus, ldm);
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------------------
// Act model: blockIdx.y = 0 policy tower (mu, sample, neglogp), = 1 value tower.
// ------------------------------------------------------------------------------------------------------------
struct StepArgs {
    const float* theta;
    const float* par;        // small-parameter mirror [2][par_total]
    const float* obs;        // [n,O] raw or already normalised
    const float* noise;      // [n,A] or null -> counter RNG (seed, rng_step)
    float* action;           // [n,A] (null: skip)   -- sampled action
    float* det_action;       // [n,A] (null: skip)   -- mu
    float* value;            // [n]   (null: value tower blocks exit)
    float* neglogp;          // [n]
    float* obs_out;          // [n,O] normalised obs copy (rollout buffer) or null
    ObsNorm nz;
    int n;
    uint32_t seed, rng_step, row_base;
    // host-Env rollouts (ppo_rollout_act): the policy tower's workgroup ALSO stores its 16 rows of actions into the handle's pinned landing buffer
    // (16-byte stores) and then raises its own word of a pinned flag table to host_seq; the host watches the table -- no D2H copy command, no
    // stream synchronisation, and the value tower's workgroups are not waited for.  null: off
    float* host_action;      // pinned host memory as the device sees it, [n,A]
    unsigned* host_flags;    // pinned, one word per 16-row block
    unsigned host_seq;
#ifdef PPO_STAMPS
    unsigned long long* stamps;   // diagnostic builds only: [tower][block][8]
#endif
};
#ifdef PPO_STAMPS
#define PSTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

template <int CT, int KS, int CTH, bool WIDE>
__global__ __launch_bounds__(BLOCK_THREADS, 2) void policy_step_kernel(NetDev net, StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    PSTAMP(0);
    warm_kernargs<sizeof(NetDev) + sizeof(StepArgs)>();
    const int tower = blockIdx.y;
    if (tower == 1 && !a.value) return;
    if (tower == 0 && !a.action && !a.det_action && !a.neglogp && !a.obs_out) return;
    const int row0 = blockIdx.x * ROWS_PER_BLOCK;
    const int ld0 = net.Kp0 + LDS_PAD;
    WRing<CT, KS> wpre;
    HeadFrag<CTH> hpre;
    dense_prefetch<CT, KS>(wpre, a.theta + net.w_off[tower][0], net.Hp[0], net.Hp[0], net.Kp0);
    float* par = lds + net.lds_par - net.par_skip;      // indexed with absolute mirror offsets
    stage_block_inputs(net, a.par + tower * net.par_total, par, lds + net.lds_h[0], ld0, a.obs, row0, a.n, a.nz,
                       tower == 0 ? a.obs_out : nullptr, nullptr, RowScalars{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, nullptr, nullptr);
    lds_barrier();
    PSTAMP(1);
    int K = net.Kp0, ldx = ld0;
    for (int l = 0; l < net.L; ++l) {
        const int Np = net.Hp[l], ldy = Np + LDS_PAD;
        dense_tile<CT, KS, EP_BIAS_TANH>(wpre, a.theta + net.w_off[tower][l], Np, WIDE ? a.par + tower * net.par_total + net.par_b[l] : par + net.par_b[l], lds + net.lds_h[l], ldx, K,
                                         lds + net.lds_h[l + 1], ldy, Np, nullptr, 0, nullptr, 0, row0, a.n, [&]() __attribute__((always_inline)) {
                                             if (l + 1 < net.L) dense_prefetch<CT, KS>(wpre, a.theta + net.w_off[tower][l + 1], net.Hp[l + 1], net.Hp[l + 1], Np);
                                             else if constexpr (CTH > 0) head_prefetch<CTH>(hpre.w, a.theta + net.wmu_off, net.Ap, Np);   // (both towers: uniform code)
                                         });
        lds_barrier();
        if (l == 0) PSTAMP(2);
        K = Np; ldx = ldy;
    }
    PSTAMP(3);
    const float* hL = lds + net.lds_h[net.L];
    if (tower == 1) {
        const float v = value_head(hL, ldx, K, par + net.par_wv, par[net.par_bv]);
        const int row = row0 + (threadIdx.x >> 4);
        if ((threadIdx.x & 15) == 0 && row < a.n) a.value[row] = v;
        PSTAMP(5);
        return;
    }
    policy_head<CTH>(net, a.theta, hpre, hL, ldx, K, lds);
    PSTAMP(4);
    const float* mus = lds + net.lds_mu;
    const int ldm = net.Ap + LDS_PAD;
    // sampling + neglogp (G:5894-6672): 16 lanes per row, each lane owns actions j = part, part+16, ...
    const int r = threadIdx.x >> 4, part = threadIdx.x & 15;
    const int row = row0 + r;
    float ssq = 0.f, slog = 0.f;
    for (int j = part; j < net.A; j += 16) {
        const float mu = mus[r * ldm + j];
        const float logstd = mu * 0.0f + par[net.par_ls + j];
        const float sigma = expf(logstd);
        float eps = 0.f;
        if (row < a.n) eps = a.noise ? a.noise[(size_t)row * net.A + j] : ctr_normal(a.seed, a.row_base + row, a.rng_step, j);
        const float act = mu + sigma * eps;
        const float z = (act - mu) / sigma;
        ssq += z * z;
        slog += logstd;
        if (row < a.n) {
            if (a.action) a.action[(size_t)row * net.A + j] = act;
            if (a.det_action) a.det_action[(size_t)row * net.A + j] = mu;
        }
        if (a.host_action) lds[net.lds_mu + r * ldm + j] = act;      // (this lane owns element (r, j): mu has been read)
    }
    ssq = group16_sum(ssq);
    slog = group16_sum(slog);
    if (part == 0 && row < a.n && a.neglogp) a.neglogp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)net.A + slog;
    if (a.host_action) {
        // The block's rows are contiguous in the landing buffer ([16][A] floats from byte 64 A blockIdx.x): whole 16-byte pieces, the last rows' odd
        // elements one by one.  Pinned host memory is uncached on the device: a store leaves for the host at once.  Every storing thread waits for
        // its stores to be acknowledged, the barrier collects the waves, then ONE word: the host reads the block only after that word shows host_seq.
        lds_barrier();
        const int live = min(ROWS_PER_BLOCK, a.n - row0) * net.A;       // elements of this block
        float* dst = a.host_action + (size_t)row0 * net.A;
        const float* tile = lds + net.lds_mu;
        for (int k = threadIdx.x; 4 * k < live; k += BLOCK_THREADS) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = min(4 * k + i, live - 1); v[i] = tile[(e / net.A) * ldm + e % net.A]; }
            if (4 * k + 3 < live) {
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t q = {v[0], v[1], v[2], v[3]};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst + 4 * k), "v"(q) : "memory");
            } else {
                for (int i = 0; 4 * k + i < live; ++i) __hip_atomic_store(dst + 4 * k + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.host_flags + blockIdx.x, a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    PSTAMP(5);
}

// ------------------------------------------------------------------------------------------------------------
// Train model forward + loss + backward down to the pre-activation gradients of every layer, for one 16-row
// tile of the minibatch and one tower.  Everything a row needs stays in LDS; what the weight-gradient kernel
// needs (layer inputs X_l and pre-activation gradients dY_l) is written to HBM/L2 once.
// ------------------------------------------------------------------------------------------------------------
struct TrainArgs {
    const float* theta;
    const float* thetaT;         // transposed copies of the matrices the backward pass streams (kept by adam_kernel)
    const float* par;            // small-parameter mirror [2][par_total] (kept by adam_kernel)
    // minibatch sources (the epoch gather has already put the rows in minibatch order)
    const float* obs; const float* actions; const float* returns; const float* old_values; const float* old_neglogp;
    const float* advs;           // explicit normalised advantages (indexed like the others) or null
    const float* adv_stats;      // {mean, denom} of this minibatch when advs == null (ppo2.hpp:401-406)
    const float* hyper;          // {lr, cliprange}
    int n;                       // rows in this minibatch on this rank
    float inv_n;                 // 1 / (global minibatch rows): gradient of the Mean nodes
    // workspaces, row-major with padded leading dimensions
    float* x0g;                  // [n][Kp0]
    float* hg[2][PPO_MAX_LAYERS];   // [tower][l] = h_{l+1} [n][Hp[l]]   (input of layer l+1 / of the head)
    float* dyg[2][PPO_MAX_LAYERS];  // [tower][l] = dLoss/d(pre-activation of layer l) [n][Hp[l]]
    float* dmug;                 // [n][Ap]
    float* slots[2];             // [tower][n_blocks][slot_w]
    unsigned long long* stamps;  // diagnostic builds only (-DPPO_STAMPS): [blocks][16] s_memtime stamps
    int xcd_map;                 // workgroup -> (tower, row tile) placement, see the kernel (speed only)
};

// EARLY (18-obs / [256, 256-multiple] shape only: Kp0 == Ap == 16*KS, Hp[0] == Hp[L-1] == 256, L >= 2): the weights of the three
// SMALL products -- first layer, policy head, policy head transposed, 32 registers each -- are requested at kernel entry
// behind the input loads and stay in registers (the kernel runs one workgroup per CU: the whole 512-entry register file of
// a SIMD belongs to one wave), and the second layer's first ring stages follow them.  The small phases then pay neither
// their own weight round trip nor the burst that requests the next product's weights.
template <int CT, int KS, int CTH, bool WIDE, bool EARLY = false>
__global__ __launch_bounds__(BLOCK_THREADS) void train_fwd_bwd_kernel(NetDev net, TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(TrainArgs)>();
    // XCD-aware row mapping: workgroups are dealt round-robin over the 8 XCDs; giving XCD x the CONTIGUOUS row tiles
    // [x*G/8, (x+1)*G/8) makes the activations / gradients this kernel leaves in that XCD's L2 exactly the rows the
    // weight-gradient kernel's row split x (also on XCD x) streams next.  xcd_map 1 (weight_grad_assemble_kernel follows, 4 row
    // splits): XCD x takes tower x & 1 and the row tiles of split x >> 1 -- what that kernel's workgroups on XCD x read, and
    // nobody else's (write-through stores to lines another XCD's L2 has read since cost this kernel 2 us).  Speed only.
    int tower = blockIdx.y;
    int rb = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
    if (a.xcd_map == 1 && gridDim.x % 4 == 0) {
        const int lid = (int)(blockIdx.x + gridDim.x * blockIdx.y), x = lid & 7, q = lid >> 3;
        tower = x & 1;
        rb = (x >> 1) * (int)(gridDim.x / 4) + q;
    }
    const int row0 = rb * ROWS_PER_BLOCK;
    const int tid = threadIdx.x;
    const int ld0 = net.Kp0 + LDS_PAD;
    float* misc = lds + net.lds_misc;          // [0,64) per-row loss terms | [64, 64+16Ap) dlogstd rows | then actions | row scalars
    float* dls = misc + 64;
    float* acts = dls + ROWS_PER_BLOCK * net.Ap;
    float* rowv = acts + ROWS_PER_BLOCK * net.Ap;   // [16][2]: pi {adv, old_neglogp} ; vf {return, old_value}
    float* slot = a.slots[tower] + (size_t)rb * net.slot_w;
    STAMP(0);
    WRing<CT, KS> wpre;
    HeadFrag<CTH> hpre;
    WFrag<CT, KS> wl0, whT;                              // EARLY: first-layer and transposed-head weights of this wave's 64 columns
    if constexpr (!EARLY) dense_prefetch<CT, KS>(wpre, a.theta + net.w_off[tower][0], net.Hp[0], net.Hp[0], net.Kp0);
    float* par = lds + net.lds_par - net.par_skip;      // indexed with absolute mirror offsets
    ObsNorm nz = {nullptr, nullptr, 0.f, 0.f, 0};
    RowScalars rs;
    if (tower == 0) rs = RowScalars{a.actions, a.advs ? a.advs : a.returns, a.advs ? nullptr : a.old_values, a.old_neglogp, a.adv_stats, 1};
    else rs = RowScalars{nullptr, a.returns, nullptr, a.old_values, nullptr, 2};
    stage_block_inputs(net, a.par + tower * net.par_total, par, lds + net.lds_h[0], ld0, a.obs, row0, a.n, nz, nullptr,
                       tower == 0 ? a.x0g : nullptr, rs, acts, rowv, [&]() __attribute__((always_inline)) {
                           if constexpr (EARLY) {
                               load_frag<CT, KS>(wl0, a.theta + net.w_off[tower][0], net.Hp[0]);
                               dense_prefetch<CT, KS>(wpre, a.theta + net.w_off[tower][1], net.Hp[1], net.Hp[1], net.Hp[0]);
                               if (tower == 0) {
                                   if constexpr (CTH > 0) head_prefetch<CTH>(hpre.w, a.theta + net.wmu_off, net.Ap, net.Hp[net.L - 1]);
                                   load_frag<CT, KS>(whT, a.thetaT + net.wmuT_off, net.Hp[net.L - 1]);
                               }
                           }
                       });
    lds_barrier();
    STAMP(1);
    // ---- forward (G:6889-9187) -------------------------------------------------------------------------------
    int K = net.Kp0, ldx = ld0;
    if constexpr (EARLY) {
        const int Np = net.Hp[0], ldy = Np + LDS_PAD;
        dense_tile_frag<CT, KS, EP_BIAS_TANH>(wl0, par + net.par_b[0], lds + net.lds_h[0], ld0, lds + net.lds_h[1], ldy, nullptr, 0, a.hg[tower][0], Np, row0, a.n);
        lds_barrier();
        STAMP(2);
        K = Np; ldx = ldy;
    }
    for (int l = EARLY ? 1 : 0; l < net.L; ++l) {
        const int Np = net.Hp[l], ldy = Np + LDS_PAD;
        dense_tile<CT, KS, EP_BIAS_TANH>(wpre, a.theta + net.w_off[tower][l], Np, WIDE ? a.par + tower * net.par_total + net.par_b[l] : par + net.par_b[l], lds + net.lds_h[l], ldx, K,
                                         lds + net.lds_h[l + 1], ldy, Np, nullptr, 0, (tower == 1 && l == net.L - 1) ? nullptr : a.hg[tower][l] /* the value head's weight gradient is formed in this kernel: nobody reads that copy */, Np, row0, a.n, [&]() __attribute__((always_inline)) {
                                             if (l + 1 < net.L) dense_prefetch<CT, KS>(wpre, a.theta + net.w_off[tower][l + 1], net.Hp[l + 1], net.Hp[l + 1], Np);
                                             else {
                                                 if constexpr (CTH > 0 && !EARLY) head_prefetch<CTH>(hpre.w, a.theta + net.wmu_off, net.Ap, Np);
                                                 // the backward pass's transposed second-layer weights: the ring is idle from here to the backward layer
                                                 if ((tower == 1 || EARLY) && net.L > 1) dense_prefetch<CT, KS>(wpre, a.thetaT + net.wT_off[tower][net.L - 1], net.Hp[net.L - 2], net.Hp[net.L - 2], Np);
                                             }
                                         }
#ifdef PPO_STAMPS
                                         , a.stamps ? (l == PPO_STAMP_LAYER ? a.stamps + (size_t)(tower * gridDim.x + rb) * 32 + 16 : nullptr) : nullptr
#endif
                                         );
        lds_barrier();
        STAMP(2 + l);
        K = Np; ldx = ldy;
    }
    const float* hL = lds + net.lds_h[net.L];
    const int HpL = K, ldhL = ldx;
    const float cr = a.hyper[1];
    const int r = tid >> 4, part = tid & 15;
    const int row = row0 + r;
    const bool live = row < a.n;
    float* dcur = lds + net.lds_d[0];
    float* dnext = lds + net.lds_d[1];

    if (tower == 0) {
        // ---- policy head + surrogate loss (G:9428-11290) and its gradient (G:12609-22656) -------------------
        policy_head<CTH>(net, a.theta, hpre, hL, ldhL, HpL, lds);
        // the head's backward weights (transposed copy) stream in under the loss arithmetic; when that product is one
        // pipeline stage deep (Ap == 16*KS) it goes to the last ring slot and the transposed weights of the layer
        // below take slots 0.. right away (see dense_tile_one_stage)
        constexpr bool ONE = CTH > 0 && CTH == KS;
        if constexpr (EARLY) {
            // nothing to request: the head's transposed weights have been in registers since kernel entry, the layer below's
            // since the end of the last forward layer
        } else if constexpr (ONE) {
            prefetch_one_stage<CT, KS, PPO_RING - 1>(wpre, a.thetaT + net.wmuT_off, HpL, HpL);
            if (net.L > 1) dense_prefetch<CT, KS>(wpre, a.thetaT + net.wT_off[0][net.L - 1], net.Hp[net.L - 2], net.Hp[net.L - 2], HpL);
        } else dense_prefetch<CT, KS>(wpre, a.thetaT + net.wmuT_off, HpL, HpL, net.Ap);
        STAMP(6);
        const float* mus = lds + net.lds_mu;
        const int ldm = net.Ap + LDS_PAD;
        float ssq = 0.f, slog = 0.f, sent = 0.f;
        for (int j = part; j < net.A; j += 16) {
            const float mu = mus[r * ldm + j];
            const float logstd = mu * 0.0f + par[net.par_ls + j];
            const float act = live ? acts[r * net.Ap + j] : mu;
            const float z = (act - mu) / expf(logstd);
            ssq += z * z;
            slog += logstd;
            sent += logstd + HALF_LOG_2PIE;
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
        const float g = a.inv_n;
        const float sel = (m1 >= m2) ? 1.0f : 0.0f;                                   // Maximum tie rule G:12609
        const float pass = ((rmin >= lo) ? 1.0f : 0.0f) * ((ratio <= hi) ? 1.0f : 0.0f); // G:15357, 16113
        float d_ratio = (-adv) * g * sel;
        d_ratio += (-adv) * g * (1.0f - sel) * pass;
        const float d_nlp = live ? -(d_ratio * ratio) : 0.0f;
        if (part == 0) {
            const float dk = nlp - old_nlp;
            misc[r * 4 + 0] = live ? tf_max(m1, m2) : 0.f;
            misc[r * 4 + 1] = live ? sent : 0.f;
            misc[r * 4 + 2] = live ? dk * dk : 0.f;
            misc[r * 4 + 3] = (live && fabsf(ratio - 1.0f) > cr) ? 1.0f : 0.f;
        }
        // d mu (-> dcur tile, also the dY operand of the head weight gradient) and d logstd rows
        for (int j = part; j < net.Ap; j += 16) {
            float dmu = 0.f, dl = 0.f;
            if (j < net.A && live) {
                const float mu = mus[r * ldm + j];
                const float sigma = expf(mu * 0.0f + par[net.par_ls + j]);
                const float z = (acts[r * net.Ap + j] - mu) / sigma;
                dl = d_nlp * (1.0f - z * z) - net.ent_coef * g;                    // AddN_2 G:21299
                dmu = d_nlp * (-(z / sigma)) + dl * 0.0f;                          // AddN_3 G:22656
            }
            dcur[r * ldm + j] = dmu;
            dls[r * net.Ap + j] = dl;
            a.dmug[(size_t)row * net.Ap + j] = dmu;                // dead rows of the last tile: zeros
        }
        lds_barrier();
        // per-block partial sums: db_mu, dlogstd (over the 16 rows, fixed order), loss terms
        for (int j = tid; j < net.Ap; j += BLOCK_THREADS) {
            float sb = 0.f, sl = 0.f;
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) { sb += dcur[q * ldm + j]; sl += dls[q * net.Ap + j]; }
            slot[net.slot_head + j] = sb;
            slot[net.slot_aux + j] = sl;
        }
        if (tid < 4) {
            float s = 0.f;
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) s += misc[q * 4 + tid];
            slot[net.slot_loss + tid] = s;
        }
        STAMP(7);
        // dh_L = (dmu * W_mu^T) .* (1 - h_L^2): a dense product against the transposed head weights [Ap][HpL]
        if constexpr (EARLY)
            dense_tile_frag<CT, KS, EP_TANHGRAD>(whT, nullptr, dcur, ldm, dnext, HpL + LDS_PAD, hL, ldhL, a.dyg[0][net.L - 1], HpL, row0, a.n);
        else if constexpr (ONE)
            dense_tile_one_stage<CT, KS, EP_TANHGRAD, PPO_RING - 1>(wpre, a.thetaT + net.wmuT_off, HpL, nullptr, dcur, ldm, dnext, HpL + LDS_PAD, HpL, hL, ldhL,
                                                                    a.dyg[0][net.L - 1], HpL, row0, a.n);
        else
        dense_tile<CT, KS, EP_TANHGRAD>(wpre, a.thetaT + net.wmuT_off, HpL, nullptr, dcur, ldm, net.Ap, dnext, HpL + LDS_PAD, HpL, hL, ldhL,
                                        a.dyg[0][net.L - 1], HpL, row0, a.n, [&]() __attribute__((always_inline)) {
                                            if (net.L > 1) dense_prefetch<CT, KS>(wpre, a.thetaT + net.wT_off[0][net.L - 1], net.Hp[net.L - 2], net.Hp[net.L - 2], HpL);
                                        });
        lds_barrier();
    } else {
        // ---- value head + clipped value loss (G:10213-10837) and its gradient (G:14975-19571) ---------------
        const float* wv = par + net.par_wv;
        const float v = value_head(hL, ldhL, HpL, wv, par[net.par_bv]);
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
        lds_barrier();
        STAMP(7);
        if (tid == 0) {
            float sb = 0.f, sl = 0.f;
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) { sb += misc[q]; sl += misc[16 + q]; }
            slot[net.slot_aux] = sb;            // db_v
            slot[net.slot_loss] = sl;           // sum of max((v-R)^2, (vclip-R)^2)
        }
        // dW_v[k] = sum_rows h_L[row,k] * dv[row] ; dh_L = dv (x) w_v .* (1 - h_L^2)
        const int ldd = HpL + LDS_PAD;
        for (int k = tid; k < HpL; k += BLOCK_THREADS) {
            float s = 0.f;
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) s = fmaf(hL[q * ldhL + k], misc[q], s);
            slot[net.slot_head + k] = s;
            const float w = wv[k];
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) {
                const float h = hL[q * ldhL + k];
                const float d = (misc[q] * w) * (1.0f - h * h);
                dnext[q * ldd + k] = d;
                a.dyg[1][net.L - 1][(size_t)(row0 + q) * HpL + k] = d;   // dead rows: d == 0
            }
        }
        lds_barrier();
    }
    STAMP(8);
    // ---- hidden layers, top down: dnext holds dLoss/d(pre-activation of layer l) -------------------------------
    if (WIDE) dcur = lds + net.lds_h[net.L];           // wide form: h_L's tile is free now and becomes the other ping-pong tile
    for (int l = net.L - 1; l >= 0; --l) {
        float* t = dcur; dcur = dnext; dnext = t;            // dcur = dY_l
        const int Np = net.Hp[l], ldd = Np + LDS_PAD;
        for (int j = tid; j < Np; j += BLOCK_THREADS) {      // db_l = sum_rows dY_l  (…/Add_grad/Sum_1)
            float s = 0.f;
            for (int q = 0; q < ROWS_PER_BLOCK; ++q) s += dcur[q * ldd + j];
            slot[net.slot_db[l] + j] = s;
        }
        if (l > 0) {                                         // first-layer dX is never needed (obs is a placeholder)
            const int Kp = net.Hp[l - 1];
            // dY_{l-1} = (dY_l * W_l^T) .* (1 - h_l^2), W_l^T = transposed copy [Hp_l][Hp_{l-1}]
            dense_tile<CT, KS, EP_TANHGRAD>(wpre, a.thetaT + net.wT_off[tower][l], Kp, nullptr, dcur, ldd, Np, dnext, Kp + LDS_PAD, Kp,
                                            WIDE ? a.hg[tower][l - 1] + (size_t)row0 * Kp : lds + net.lds_h[l], WIDE ? Kp : Kp + LDS_PAD,
                                            a.dyg[tower][l - 1], Kp, row0, a.n, [&]() __attribute__((always_inline)) {
                                                if (l > 1) dense_prefetch<CT, KS>(wpre, a.thetaT + net.wT_off[tower][l - 1], net.Hp[l - 2], net.Hp[l - 2], Kp);
                                            });
        }
        lds_barrier();
        STAMP(9 + (net.L - 1 - l));
    }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradients dW = X^T dY (reduction over the minibatch rows), all layers of both towers in one launch.
// One workgroup = one (TI*16 x TJ*16) output tile x one K-split; its 4 waves take quarter row-slices and are
// summed through LDS.  Operands stream L2 -> VGPR as 16-byte vectors: lane (g, c) of MFMA (a, b) covers output
// row i0 + TI*c' ... see the index algebra in the body.
// ------------------------------------------------------------------------------------------------------------
struct DwTile {
    const float* X; const float* dY;   // [n][ldx], [n][ldy]
    int ldx, ldy;
    int i0, j0;                        // tile origin in the [Kp x Np] gradient
    int out_off;                       // offset of the tensor inside the padded parameter vector
    int ldo;                           // = Np
    int cls;                           // 0: 64x64 tile (TI=TJ=4) ; 1: 16x16 ; 2: 32x64 (TI=2,TJ=4) ; 3: 64x32 (TI=4,TJ=2)
                                       // 4: 32x16 strip (TI=2,TJ=1) ; 5: 16x32 strip (TI=1,TJ=2)
};

// One workgroup's work: a main tile plus up to two thin strips of the narrow matrices (first layer, policy head).
// Folding the strips into the 64x64 workgroups keeps the launch at one balanced workgroup per CU (extra workgroups
// would double up on some CUs and set the kernel's critical path).
struct DwWork { DwTile main; int n_extra; int fused; DwTile extra[2]; };   // fused: main is 64x64, extra[0] cls 4 (if any), extra[1] cls 5

#ifndef PPO_DW_RING
#define PPO_DW_RING 4              // register stages per wave in the weight-gradient kernel (operands come from the
#endif                             // Infinity Cache / HBM: the producer kernel ran on other XCDs)

template <int N>
__device__ __forceinline__ void gload_vec(float (&dst)[N], const float* base, unsigned byte_off) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) char* gchar;
    gchar p = (gchar)(reinterpret_cast<unsigned long long>(base)) + byte_off;
    if constexpr (N == 4) { const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)p; dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    else if constexpr (N == 2) { const f32x2_t v = *(const __attribute__((address_space(1))) f32x2_t*)p; dst[0] = v[0]; dst[1] = v[1]; }
    else { dst[0] = *(const __attribute__((address_space(1))) float*)p; }
}

template <int TI, int TJ, int KQ>
__device__ __forceinline__ void dw_tile_body(const DwTile& t, int n, int nsplit, float* __restrict__ slabs, size_t slab_stride,
                                             float* lds, unsigned long long* dbg = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so id % nsplit puts every tile of one row
    // split on one XCD (nsplit = 8): the X / dY slabs of that split are fetched from the fabric once per XCD and
    // shared through its L2 by the tiles that need them (speed only, never correctness)
    const int split = blockIdx.x % nsplit;
    DSTAMP(0);
    const int rows_per_split = n / nsplit;
    const int rows_per_wave = rows_per_split / 4;
    const int rbeg = uni(split * rows_per_split + wave * rows_per_wave);
    const int ldx = uni(t.ldx), ldy = uni(t.ldy);
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // MFMA (a,b): A[i][k] = X[row k][i0 + TI*i + a], B[k][j] = dY[row k][j0 + TJ*j + b], k = lane group g.
    // One pipeline stage = KQ k-steps (4*KQ minibatch rows): 2*KQ vector loads per lane feed TI*TJ*KQ MFMAs.
    // Addresses = wave-uniform stage base (scalar registers) + per-lane 32-bit byte offsets computed once.
    const float* Xb = uni(t.X) + (size_t)rbeg * ldx;
    const float* Yb = uni(t.dY) + (size_t)rbeg * ldy;
    unsigned xo[KQ], yo[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        xo[q] = (unsigned)(((g + 4 * q) * ldx + t.i0 + TI * c) * 4);
        yo[q] = (unsigned)(((g + 4 * q) * ldy + t.j0 + TJ * c) * 4);
    }
    struct Stage { float x[KQ][TI]; float y[KQ][TJ]; };
    Stage st[PPO_DW_RING];
    constexpr int RS = 4 * KQ;                                      // minibatch rows per pipeline stage
    auto ld = [&](Stage& s_, int koff) __attribute__((always_inline)) {
        koff = koff < rows_per_wave ? koff : rows_per_wave - RS;   // clamped: unconditional loads, counted vmcnt
        const float* xs = Xb + (size_t)koff * ldx;
        const float* ys = Yb + (size_t)koff * ldy;
#pragma unroll
        for (int q = 0; q < KQ; ++q) { gload_vec<TI>(s_.x[q], xs, xo[q]); gload_vec<TJ>(s_.y[q], ys, yo[q]); }
    };
    auto compute = [&](const Stage& s_) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_.x[q][a], s_.y[q][b], acc[a][b], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < PPO_DW_RING - 1; ++i) ld(st[i], i * RS);
    DSTAMP(1);
    for (int k = 0; k < rows_per_wave; k += PPO_DW_RING * RS) {
#pragma unroll
        for (int i = 0; i < PPO_DW_RING; ++i) {
            ld(st[(i + PPO_DW_RING - 1) % PPO_DW_RING], k + (i + PPO_DW_RING - 1) * RS);
            __builtin_amdgcn_sched_barrier(0);
            if (k + i * RS < rows_per_wave) compute(st[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    DSTAMP(2);
    // in-workgroup split-K: every wave parks its tile in LDS, then all threads add the 4 copies in wave order
    constexpr int TW = 16 * TJ, TH = 16 * TI;
    float* mine = lds + wave * (TH * TW);
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = TI * (4 * g + r) + a;          // accumulator row 4g+r of MFMA (a,*) = output row i0 + orow
#pragma unroll
            for (int b = 0; b < TJ; ++b) mine[orow * TW + TJ * c + b] = acc[a][b][r];
        }
    __syncthreads();
    DSTAMP(3);
    float* out = slabs + (size_t)split * slab_stride + t.out_off;
    for (int i = threadIdx.x; i < TH * TW; i += BLOCK_THREADS) {
        const float s = ((lds[i] + lds[TH * TW + i]) + lds[2 * TH * TW + i]) + lds[3 * TH * TW + i];
        const int orow = i / TW, ocol = i - orow * TW;
        st_wt<PPO_WT_B>(out + (size_t)(t.i0 + orow) * t.ldo + t.j0 + ocol, s);
    }
    DSTAMP(4);
}

// Balanced layout body: the main 64x64 tile plus NE thin strips (extra[0] = [32x16] first-layer strip, extra[1] = [16x32]
// head strip) over the SAME minibatch rows.  The strips' operands (a few registers) are requested before the main loop
// so their memory latency hides under it; their matrix instructions run right after it and their partial tiles are
// parked, reduced and stored together with the main tile (one LDS pass, one barrier).  Run one after the other, each
// strip costs a latency-bound prologue + reduction of ~2.5 us.
#define DW_STRIP_STEPS 16          // k-steps (of 4 rows) a wave can hold for the strips: rows_per_wave <= 64

template <int KQ, int NE>
__device__ __forceinline__ void dw_main_with_strips(const DwWork& w, int n, int nsplit, float* __restrict__ slabs, size_t slab_stride, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int split = blockIdx.x % nsplit;
    const int rows_per_split = n / nsplit;
    const int rows_per_wave = rows_per_split / 4;
    const int rbeg = uni(split * rows_per_split + wave * rows_per_wave);
    const int ksteps = rows_per_wave / 4;                              // <= DW_STRIP_STEPS (checked by the caller)
    const DwTile& t = w.main;
    const DwTile& e0 = w.extra[0];
    const DwTile& e1 = w.extra[1];
    // ---- strips: every operand of this wave's rows, issued now, consumed after the main loop ---------------------
    float x0[DW_STRIP_STEPS][2], y0[DW_STRIP_STEPS][1], x1[DW_STRIP_STEPS][1], y1[DW_STRIP_STEPS][2];
    if constexpr (NE >= 1) {
        const int lx = uni(e0.ldx), ly = uni(e0.ldy);
        const float* xb = uni(e0.X) + (size_t)rbeg * lx;
        const float* yb = uni(e0.dY) + (size_t)rbeg * ly;
#pragma unroll
        for (int q = 0; q < DW_STRIP_STEPS; ++q) {
            const int qq = q < ksteps ? q : ksteps - 1;
            gload_vec<2>(x0[q], xb, (unsigned)(((g + 4 * qq) * lx + e0.i0 + 2 * c) * 4));
            gload_vec<1>(y0[q], yb, (unsigned)(((g + 4 * qq) * ly + e0.j0 + c) * 4));
        }
    }
    if constexpr (NE >= 2) {
        const int lx = uni(e1.ldx), ly = uni(e1.ldy);
        const float* xb = uni(e1.X) + (size_t)rbeg * lx;
        const float* yb = uni(e1.dY) + (size_t)rbeg * ly;
#pragma unroll
        for (int q = 0; q < DW_STRIP_STEPS; ++q) {
            const int qq = q < ksteps ? q : ksteps - 1;
            gload_vec<1>(x1[q], xb, (unsigned)(((g + 4 * qq) * lx + e1.i0 + c) * 4));
            gload_vec<2>(y1[q], yb, (unsigned)(((g + 4 * qq) * ly + e1.j0 + 2 * c) * 4));
        }
    }
    // ---- main 64x64 tile: the ring-pipelined loop of dw_tile_body<4,4,KQ> ------------------------------------------
    const int ldx = uni(t.ldx), ldy = uni(t.ldy);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* Xb = uni(t.X) + (size_t)rbeg * ldx;
    const float* Yb = uni(t.dY) + (size_t)rbeg * ldy;
    unsigned xo[KQ], yo[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        xo[q] = (unsigned)(((g + 4 * q) * ldx + t.i0 + 4 * c) * 4);
        yo[q] = (unsigned)(((g + 4 * q) * ldy + t.j0 + 4 * c) * 4);
    }
    struct Stage { float x[KQ][4]; float y[KQ][4]; };
    Stage st[PPO_DW_RING];
    constexpr int RS = 4 * KQ;
    auto ld = [&](Stage& s_, int koff) __attribute__((always_inline)) {
        koff = koff < rows_per_wave ? koff : rows_per_wave - RS;
        const float* xs = Xb + (size_t)koff * ldx;
        const float* ys = Yb + (size_t)koff * ldy;
#pragma unroll
        for (int q = 0; q < KQ; ++q) { gload_vec<4>(s_.x[q], xs, xo[q]); gload_vec<4>(s_.y[q], ys, yo[q]); }
    };
    auto compute = [&](const Stage& s_) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_.x[q][a], s_.y[q][b], acc[a][b], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < PPO_DW_RING - 1; ++i) ld(st[i], i * RS);
    for (int k = 0; k < rows_per_wave; k += PPO_DW_RING * RS) {
#pragma unroll
        for (int i = 0; i < PPO_DW_RING; ++i) {
            ld(st[(i + PPO_DW_RING - 1) % PPO_DW_RING], k + (i + PPO_DW_RING - 1) * RS);
            __builtin_amdgcn_sched_barrier(0);
            if (k + i * RS < rows_per_wave) compute(st[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- strip matrix instructions ----------------------------------------------------------------------------------
    f32x4 a0[2], a1[2];
    a0[0] = a0[1] = a1[0] = a1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < DW_STRIP_STEPS; ++q) {
        if (q < ksteps) {
            if constexpr (NE >= 1) {
                a0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[q][0], y0[q][0], a0[0], 0, 0, 0);
                a0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[q][1], y0[q][0], a0[1], 0, 0, 0);
            }
            if constexpr (NE >= 2) {
                a1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[q][0], y1[q][0], a1[0], 0, 0, 0);
                a1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[q][0], y1[q][1], a1[1], 0, 0, 0);
            }
        }
    }
    // ---- park [64x64 main | 32x16 strip 0 | 16x32 strip 1] per wave, reduce over the 4 waves, store -------------------
    constexpr int WSZ = 64 * 64 + 512 + 512;
    float* mine = lds + wave * WSZ;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)       // the 4 column tiles of a lane are 4 consecutive floats: one 16-byte LDS write
            *reinterpret_cast<float4*>(mine + (4 * (4 * g + r) + a) * 64 + 4 * c) = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
    if constexpr (NE >= 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[4096 + (2 * (4 * g + r) + a) * 16 + c] = a0[a][r];
    }
    if constexpr (NE >= 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[4608 + (4 * g + r) * 32 + 2 * c + b] = a1[b][r];
    }
    __syncthreads();
    float* slab = slabs + (size_t)split * slab_stride;
    for (int v = threadIdx.x; v < 1024; v += BLOCK_THREADS) {          // 16-byte LDS reads, 16-byte slab stores
        const int i = 4 * v;
        const float4 p0 = *reinterpret_cast<const float4*>(lds + i), p1 = *reinterpret_cast<const float4*>(lds + WSZ + i);
        const float4 p2 = *reinterpret_cast<const float4*>(lds + 2 * WSZ + i), p3 = *reinterpret_cast<const float4*>(lds + 3 * WSZ + i);
        const float4 s_ = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                                      ((p0.w + p1.w) + p2.w) + p3.w);
        st_wt4<PPO_WT_B>(slab + t.out_off + (size_t)(t.i0 + (i >> 6)) * t.ldo + t.j0 + (i & 63), s_);
    }
    if constexpr (NE >= 1) {
        for (int i = threadIdx.x; i < 512; i += BLOCK_THREADS) {
            const int j = 4096 + i;
            const float s_ = ((lds[j] + lds[WSZ + j]) + lds[2 * WSZ + j]) + lds[3 * WSZ + j];
            st_wt<PPO_WT_B>(slab + e0.out_off + (size_t)(e0.i0 + (i >> 4)) * e0.ldo + e0.j0 + (i & 15), s_);
        }
    }
    if constexpr (NE >= 2) {
        for (int i = threadIdx.x; i < 512; i += BLOCK_THREADS) {
            const int j = 4608 + i;
            const float s_ = ((lds[j] + lds[WSZ + j]) + lds[2 * WSZ + j]) + lds[3 * WSZ + j];
            st_wt<PPO_WT_B>(slab + e1.out_off + (size_t)(e1.i0 + (i >> 5)) * e1.ldo + e1.j0 + (i & 31), s_);
        }
    }
}

struct DwArgs {
    const DwWork* tiles;
    int n;                 // minibatch rows
    int nsplit;            // row splits; grid = n_tiles * nsplit workgroups, split = id % nsplit
    float* slabs;          // [nsplit][P_pad]
    size_t slab_stride;
    unsigned long long* stamps;   // diagnostic builds only
};

template <int KQ>
__global__ __launch_bounds__(BLOCK_THREADS) void weight_grad_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef PPO_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
    const DwWork& w = a.tiles[blockIdx.x / a.nsplit];
    if (uni(w.fused) && a.n / a.nsplit / 4 <= 4 * DW_STRIP_STEPS) {     // balanced layout: strips ride with the main tile
        const int ne = uni(w.n_extra);
        if (ne == 2) dw_main_with_strips<KQ, 2>(w, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else if (ne == 1) dw_main_with_strips<KQ, 1>(w, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else dw_main_with_strips<KQ, 0>(w, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
#ifdef PPO_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
        return;
    }
    const int n_sub = 1 + uni(w.n_extra);
    for (int e = 0; e < n_sub; ++e) {
        const DwTile t = e == 0 ? w.main : w.extra[e - 1];
        const int cls = uni(t.cls);
        if (e) __syncthreads();                        // LDS of the previous tile's split-K reduction is reused
        if (cls == 0) dw_tile_body<4, 4, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds
#ifdef PPO_STAMPS
            , a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr
#endif
            );
        else if (cls == 2) dw_tile_body<2, 4, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else if (cls == 3) dw_tile_body<4, 2, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else if (cls == 4) dw_tile_body<2, 1, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else if (cls == 5) dw_tile_body<1, 2, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
        else dw_tile_body<1, 1, KQ>(t, a.n, a.nsplit, a.slabs, a.slab_stride, lds);
    }
#ifdef PPO_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Gradient assembly.  One 256-thread block covers 256 consecutive elements of the padded parameter vector; the
// host table says where the block's elements come from (block-uniform):
//   kind 0: matrix -> sum of the split-K slabs (fixed order)
//   kind 1: slot   -> sum over the row blocks' slots (fixed order), `count` valid elements, rest are padding
//   kind 2: padding / untrained -> 0
//   kind 3: finished vector (bf16 path: bias gradients from its row-sum kernel)
// Also emits the block's sum of squares for the global norm, and (last block) the five loss scalars.
// ------------------------------------------------------------------------------------------------------------
struct GradSrc { int kind; int tower; int slot_off; int count; int base;   // base = first element of the tensor
                 int t_off, prow, pcol;                                       // transposed copy: thetaT[t_off + c*prow + r], t_off < 0: none
                 int p_off, p_count;                                          // small-parameter mirror: par[p_off + e], e < p_count ; p_off < 0: none
                 // narrow path: packed LDS images (one per tower, [2][img_stride]) of the weights in exactly the layout its kernels
                 // copy into LDS; a tensor's element (r, c) lives at img[i_off + r*i_ld + c] (forward copy), img[it_off + c*it_ld + r]
                 // (transposed copy) and a small parameter e at img[ip_off + e]; < 0: none
                 int i_off, i_ld, it_off, it_ld, ip_off;
                 int tile0; };                                                // bf16 path: index of the tensor's first weight-gradient tile (tiles row-major over [prow / BM][pcol / 128])

struct ReduceArgs {
    const GradSrc* src;          // [n_blocks]
    int n_blocks;                // blocks covering P_pad ; block n_blocks = loss block
    const float* slabs; size_t slab_stride; int nsplit;
    int sk_nst, sk_per, sk_bm;   // bf16 path (work-balanced weight-gradient GEMM): stages per tile, stages per workgroup, tile rows; a tile's
                                 // partial sums are slabs 0 .. (last - first) of the workgroups first = t*nst/per .. last = (t*nst+nst-1)/per
    int sk_tile_base;            // ... t = the tile's index in the launch's OWN sequence: its index in the table minus this (0: the launch walked the whole table)
    int chunk_lo, chunk_hi;      // bf16_grad_reduce_kernel on a BUCKET (bucketed data-parallel step): chunks [chunk_lo, chunk_hi) only; 0, 0 = the whole vector + the tail block
    const float* slots[2]; int n_rowblocks; int slot_w;
    int slot_loss;
    const float* direct;         // kind 3: per-row-tile sums (bias gradients from the bf16 path's TanhGrad epilogues), [n_direct][direct_stride], indexed slot_off + e
    int n_direct; int direct_stride;
    float* grad;                 // [P_pad]  (+ 8 tail floats: 5 loss sums, row count)
    float* sumsq;                // [n_blocks]
    float n_local;               // rows summed on this rank
    float* beta_pow;             // {cur b1, cur b2, next b1, next b2}: cur <- next here (adam writes next)
};

__global__ __launch_bounds__(256) void grad_reduce_kernel(ReduceArgs a) {
    __shared__ float red[4];
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk == a.n_blocks) {
        // loss partial sums over the row blocks: tail = {pg, vf, ent, kl, cf, rows}; 32 lanes per quantity
        if (tid < 160) {
            const int q = tid >> 5, ln = tid & 31;
            const int tower = (q == 1) ? 1 : 0;
            const int off = a.slot_loss + (q <= 1 ? 0 : q - 1);
            float s = 0.f;
            for (int b = ln; b < a.n_rowblocks; b += 32) s += a.slots[tower][(size_t)b * a.slot_w + off];
            s = half_sum_lane0(s);
            if (ln == 0) a.grad[(size_t)a.n_blocks * 256 + q] = s;
        }
        if (tid == 160) a.grad[(size_t)a.n_blocks * 256 + 5] = a.n_local;
        if (tid == 161) { a.beta_pow[0] = a.beta_pow[2]; a.beta_pow[1] = a.beta_pow[3]; }
        return;
    }
    const GradSrc s = a.src[blk];
    const size_t idx = (size_t)blk * 256 + tid;
    float gsum = 0.f;
    if (s.kind == 0) {
        for (int k = 0; k < a.nsplit; ++k) gsum += a.slabs[(size_t)k * a.slab_stride + idx];
    } else if (s.kind == 1) {
        const int e = (int)(idx - (size_t)s.base);
        if (e < s.count) {
            // 4 independent chains x unroll 8 = 32 loads in flight (a serial chain of n_rowblocks L2 round trips
            // costs ~30 us); the summation order is fixed, so the result is reproducible
            const float* p = a.slots[s.tower] + s.slot_off + e;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int b = 0;
#pragma unroll 8
            for (; b + 4 <= a.n_rowblocks; b += 4) {
                s0 += p[(size_t)b * a.slot_w];
                s1 += p[(size_t)(b + 1) * a.slot_w];
                s2 += p[(size_t)(b + 2) * a.slot_w];
                s3 += p[(size_t)(b + 3) * a.slot_w];
            }
            for (; b < a.n_rowblocks; ++b) s0 += p[(size_t)b * a.slot_w];
            gsum = (s0 + s1) + (s2 + s3);
        }
    } else if (s.kind == 3) {
        const int e = (int)(idx - (size_t)s.base);
        if (e < s.count) for (int t = 0; t < a.n_direct; ++t) gsum += a.direct[(size_t)t * a.direct_stride + s.slot_off + e];
    }
    st_wt<PPO_WT_C1>(a.grad + idx, gsum);
    float q = gsum * gsum;
    q = wave_sum_lane0(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) st_wt<PPO_WT_C1>(a.sumsq + blk, (red[0] + red[1]) + (red[2] + red[3]));
}

// rebuild every transposed copy from theta (after parameters were written from the host)
__device__ __forceinline__ void write_images(const GradSrc& gs, float* img, int e, float x) {
    if (gs.ip_off >= 0) { if (e < gs.p_count) img[gs.ip_off + e] = x; return; }
    if ((gs.i_off >= 0 || gs.it_off >= 0) && e < gs.prow * gs.pcol) {
        const int r = e / gs.pcol, c = e - r * gs.pcol;
        if (gs.i_off >= 0) img[gs.i_off + r * gs.i_ld + c] = x;
        if (gs.it_off >= 0) img[gs.it_off + c * gs.it_ld + r] = x;
    }
}

__global__ __launch_bounds__(256) void transpose_refresh_kernel(const float* theta, float* thetaT, float* par, const GradSrc* src, float* img) {
    const GradSrc gs = src[blockIdx.x];
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(idx - (size_t)gs.base);
    if (gs.t_off >= 0 && e < gs.prow * gs.pcol) { const int r = e / gs.pcol, c = e - r * gs.pcol; thetaT[gs.t_off + c * gs.prow + r] = theta[idx]; }
    if (gs.p_off >= 0 && e < gs.p_count) par[gs.p_off + e] = theta[idx];
    if (img) write_images(gs, img, e, theta[idx]);
}

// after a cross-rank all-reduce of grad the per-block sums of squares must be recomputed
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* grad, float* sumsq) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float gv = grad[(size_t)blockIdx.x * 256 + tid];
    float q = gv * gv;
    q = wave_sum_lane0(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------------------
// clip_by_global_norm (G:23738-25392) + ApplyAdam (TF 1.14; G:30430-31383) on the padded flat vector.
// Every block re-derives the norm from the per-block partials in the same fixed order -> identical on all
// blocks (and on all ranks after the all-reduce).  Block 0 also writes the loss row and the next beta powers.
// ------------------------------------------------------------------------------------------------------------
#define ADAM_MAX_TILED 8
struct AdamArgs {
    float* theta; float* m; float* v; const float* grad; const float* sumsq; int n_blocks;
    float* thetaT; float* par; const GradSrc* src;
    const float* hyper;          // {lr, cliprange}
    float* beta_pow;             // {cur b1, cur b2, next b1, next b2}
    float beta1, beta2, eps, max_norm;
    float* loss_row;             // [5] destination for this train step (may be null)
    float* norm_out;             // [1] (may be null)
    const float* norm_parts; int n_parts;   // what the norm is summed from: sumsq itself, or its 1024-wide folds for very large nets
    // matrices whose transposed copy is written as whole 32 x 32 tiles (their region starts on a 1024-element boundary and both padded
    // dimensions are multiples of 32): a block of adam_kernel then takes ONE tile of the matrix instead of 1024 consecutive elements,
    // and the transposed tile leaves through LDS as 128-byte rows.  (Element by element the transposed copy is a scatter of 4-byte
    // stores 1 KB apart -- 64 cache lines per store instruction: 1.7 us of a 40 us train step at BASELINE configs[2].)
    int n_tiled; struct Tiled { int base, count, pcol, prow, t_off; } tiled[ADAM_MAX_TILED];
    __bf16* theta_bf;            // bf16 path: straight bf16 copy of theta kept current here (null otherwise)
    float* img;                  // narrow path: packed LDS images kept current here (null otherwise)
    const float* theta_in; const float* m_in; const float* v_in;   // read from another parameter set (narrow path's deferred Adam); null = in place
    // data parallel, MEET > 0 (see adam_kernel): the grid-wide meeting's table -- one epoch word and one partial sum of squares per workgroup, an
    // error word behind the epoch words -- and, MEET == 2, the peer regions whose slots this kernel adds up itself
    unsigned* meet_words; float* meet_parts; int meet_grid;
    PeerDev peer;
#ifdef PPO_STAMPS
    unsigned long long* stamps;  // diagnostic builds only: [block][8]
#endif
};
#ifdef PPO_STAMPS
#define ASTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

// second-level partial sums of the per-chunk sums of squares: with millions of parameters every Adam block re-reading
// every chunk's partial is hundreds of MB of L2 traffic per step; 1024-wide folds in a fixed order keep the norm
// reproducible and the re-read a few dozen floats
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* sumsq, int n, float* parts) {
    __shared__ float red[4];
    const int tid = threadIdx.x, base = blockIdx.x * 1024;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + tid + 256 * k; if (i < n) s += sumsq[i]; }
    s = wave_sum_lane0(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) parts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// TF-1.14 ApplyAdam on one element (G:30430-31383): m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); theta -= alpha m / (sqrt(v) + eps),
// g already scaled by the clip factor.
//   FAST = false: correctly rounded square root and division, the arithmetic of the reference's CPU kernel.  Every path but the one
//     below (adam_kernel on the fused fp32 families and on the bf16 path's fp32 master weights).
//   FAST = true: the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32.  ONLY the narrow path's deferred Adam (ppo_narrow.hpp), where this
//     expression runs redundantly in every workgroup and the exact sequences cost 2.2 k cycles of a 25 k-cycle kernel, and the
//     adam_kernel launches of the same handle (so that the deferred and the launched form stay bit-identical).  The update term is
//     ~1e-3 of the weight, so its 3-ulp error is 4e-10 absolute, a twentieth of the weight's own ulp -- a stated deviation from
//     "the reference's arithmetic" on that path (include/ppo_hip.h).
template <bool FAST>
__device__ __forceinline__ void adam_element(float gscaled, float m, float v, float t, float one_m_b1, float one_m_b2, float alpha, float eps,
                                             float& mo, float& vo, float& to) {
    mo = m + (gscaled - m) * one_m_b1;
    vo = v + (gscaled * gscaled - v) * one_m_b2;
    if constexpr (FAST) to = t - (mo * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vo) + eps);
    else to = t - (mo * alpha) / (sqrtf(vo) + eps);
}

// One block = 1024 consecutive parameters (4 per thread, 16-byte accesses); n_blocks counts the 256-element chunks the
// gradient-source table and the partial sums of squares are indexed by.
// MEET (data parallel only; the single-GPU launch is MEET = 0 and unchanged):
//   2: nobody has added the ranks' gradients yet: the finishers of weight_grad_assemble_kernel<.., PEER> pushed this rank's tiles straight into
//      every peer's slot (ppo_dw2.hpp).  Every workgroup waits for the W flags, adds the W slots of its elements in RANK order (the same order on
//      every rank: bit-identical replicas), writes the sum to `grad`, adds up the squares of ITS elements, publishes the partial, the workgroups
//      meet, and everybody derives the norm from the partials in workgroup order -- no push launch, no sum launch (tools/peer_overhead.py, one
//      rank, configs[2]: 54.9 -> 50.1 us per train step; 39.4 without a communicator).
//   1: `grad` already holds the all-reduced gradient, only its sums of squares are missing (what the RCCL path launches grad_sumsq_kernel for).
//      Measured: 43.1 us per train step against 42.2 with the launch -- the meeting costs more than the launch it replaces.  Not instantiated.
// The meeting: every workgroup of the launch must be resident together (the host uses these forms up to ADAM_MEET_MAX_GRID workgroups of this
// 256-thread, 4 KB kernel).  No read-modify-write on a shared word: workgroup b reads its own epoch word e at entry, stores e + 1 behind its partial,
// and 256 threads watch the table.  The wait is bounded: on a time-out the error word is raised, the step's result is garbage and the next
// synchronising call reports it.
#define ADAM_MEET_MAX_GRID 1024
__device__ __forceinline__ f32x4 adam_ld4_sys(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <bool FAST, int MEET = 0>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    __shared__ float red[4];
    __shared__ float tt[32][33];
    ASTAMP(0);
    const int tid = threadIdx.x;
    unsigned epoch = 0;
    if constexpr (MEET > 0) epoch = __hip_atomic_load(a.meet_words + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (only this workgroup writes its word)
    size_t idx = ((size_t)blockIdx.x * 256 + tid) * 4;
    const int chunk = (int)(idx >> 8);
    const bool live = chunk < a.n_blocks;
    // a block inside a tiled matrix takes tile (ti, tj): thread t = row t / 8, columns 4 (t % 8) .. + 3 of the tile (wave-uniform lookup in the
    // kernel arguments: no memory round trip in front of the element loads)
    int tiled = -1, ti = 0, tj = 0, t_base = 0, t_pcol = 32, t_prow = 0, t_toff = 0;
    {
        const int b0 = (int)blockIdx.x * 1024;
#pragma unroll
        for (int q = 0; q < ADAM_MAX_TILED; ++q)
            if (q < a.n_tiled && b0 >= a.tiled[q].base && b0 < a.tiled[q].base + a.tiled[q].count) {
                tiled = q; t_base = a.tiled[q].base; t_pcol = a.tiled[q].pcol; t_prow = a.tiled[q].prow; t_toff = a.tiled[q].t_off;
            }
        if (tiled >= 0) {
            const int k = (b0 - t_base) >> 10, tpr = t_pcol >> 5;
            ti = k / tpr; tj = k - ti * tpr;
            idx = (size_t)t_base + (size_t)(32 * ti + (tid >> 3)) * t_pcol + 32 * tj + 4 * (tid & 7);
        }
    }
    // issue this thread's element loads BEFORE the norm reduction: both memory round trips overlap
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = g4, v4 = g4, t4 = g4;
    GradSrc gs{};
    gs.t_off = -1; gs.p_off = -1;
    if (live) {
        if constexpr (MEET != 2) g4 = *reinterpret_cast<const float4*>(a.grad + idx);
        m4 = *reinterpret_cast<const float4*>((a.m_in ? a.m_in : a.m) + idx);
        v4 = *reinterpret_cast<const float4*>((a.v_in ? a.v_in : a.v) + idx);
        t4 = *reinterpret_cast<const float4*>((a.theta_in ? a.theta_in : a.theta) + idx);
        gs = a.src[chunk];
    }
    ASTAMP(1);
    const float b1p = a.beta_pow[0], b2p = a.beta_pow[1], lr = a.hyper[0];
    if constexpr (MEET == 2) {
        // the ranks' slots of this workgroup's elements, added in rank order (peer_sum_kernel's arithmetic) and left in `grad`
        __shared__ unsigned s_seq;
        if (tid == 0) s_seq = __hip_atomic_load(a.peer.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned sq = s_seq, par = sq & 1u;
        if (tid < (unsigned)a.peer.world) {
            const unsigned* f = a.peer.flags[a.peer.rank] + ((size_t)par * PEER_MAX_WORLD + tid) * PEER_FLAG_STRIDE;
            unsigned it = 0;
            while (peer_ld_sys(f) != sq) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > a.peer.spin_limit) { __hip_atomic_store(a.peer.err, 1u + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        if (tid < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        const float* mine = a.peer.slots[a.peer.rank] + (unsigned long long)par * a.peer.world * a.peer.cap;
        f32x4 sv[PEER_MAX_WORLD];
#pragma unroll
        for (int r = 0; r < PEER_MAX_WORLD; ++r) sv[r] = adam_ld4_sys(mine + (unsigned long long)(r < a.peer.world ? r : 0) * a.peer.cap + (live ? idx : 0));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]) :: "memory");
        f32x4 acc = sv[0];
#pragma unroll
        for (int r = 1; r < PEER_MAX_WORLD; ++r) if (r < a.peer.world) acc += sv[r];
        g4 = live ? make_float4(acc[0], acc[1], acc[2], acc[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) *reinterpret_cast<float4*>(const_cast<float*>(a.grad) + idx) = g4;       // (ppo_get_last_grad reads it)
        if (blockIdx.x == 0 && tid < 8) {                                                   // the loss sums + row count ride behind the gradient
            float t = 0.f;
            for (int r = 0; r < a.peer.world; ++r) { const float x = __hip_atomic_load(mine + (unsigned long long)r * a.peer.cap + (size_t)a.n_blocks * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); t = r ? t + x : x; }
            const_cast<float*>(a.grad)[(size_t)a.n_blocks * 256 + tid] = t;
        }
    }
    // global norm from the per-chunk partial sums, same fixed order in every block (and on every rank)
    float s = 0.f;
    if constexpr (MEET > 0) {
        // this workgroup's partial: the squares of its 1024 elements in a fixed tree, published behind a write-through store; then the meeting
        float q = (g4.x * g4.x + g4.y * g4.y) + (g4.z * g4.z + g4.w * g4.w);
        q = wave_sum_lane0(q);
        if ((tid & 63) == 0) red[tid >> 6] = q;
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(a.meet_parts + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.meet_words + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned polls = 0;
        for (;;) {
            bool ok = true;
            for (int i = tid; i < a.meet_grid; i += 256) ok = ok && (int)(__hip_atomic_load(a.meet_words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1u << 20)) { if (tid == 0) __hip_atomic_store(a.meet_words + ADAM_MEET_MAX_GRID, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        for (int i = tid; i < a.meet_grid; i += 256) s += __hip_atomic_load(a.meet_parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
    if (a.n_parts > 512) {

        // many partials (the bf16 path's one per assembly workgroup): a thread's first eight are requested together -- one memory round
        // trip, not one per partial -- and added in the same index order
        float pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = tid + 256 * k; pv[k] = i < a.n_parts ? a.norm_parts[i] : 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (tid + 256 * k < a.n_parts) s += pv[k];
        for (int i = tid + 2048; i < a.n_parts; i += 256) s += a.norm_parts[i];
    } else {
        for (int i = tid; i < a.n_parts; i += 256) s += a.norm_parts[i];
    }
    ASTAMP(2);
    s = wave_sum_lane0(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    ASTAMP(3);
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    float scale = a.max_norm * tf_min(1.0f / norm, 1.0f / a.max_norm);          // G:24289-24472
    if (!isfinite(norm)) scale = __builtin_nanf("");                            // G:24493-24543
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    if (live) {
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
        float mo[4], vo[4], to[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_element<FAST>(gv[k] * scale, mv[k], vv[k], tv[k], 1.0f - a.beta1, 1.0f - a.beta2, alpha, a.eps, mo[k], vo[k], to[k]);
        st_wt4<PPO_WT_C2>(a.m + idx, make_float4(mo[0], mo[1], mo[2], mo[3]));
        st_wt4<PPO_WT_C2>(a.v + idx, make_float4(vo[0], vo[1], vo[2], vo[3]));
        st_wt4<PPO_WT_C2>(a.theta + idx, make_float4(to[0], to[1], to[2], to[3]));
        if (a.theta_bf) {
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            bf16x4_t o4; o4[0] = (__bf16)to[0]; o4[1] = (__bf16)to[1]; o4[2] = (__bf16)to[2]; o4[3] = (__bf16)to[3];
            *reinterpret_cast<bf16x4_t*>(a.theta_bf + idx) = o4;
        }
        const int e0 = (int)(idx - (size_t)gs.base);
        if (gs.t_off >= 0 && tiled < 0) {                  // keep the backward pass's transposed copy current (element by element: small / unaligned matrices)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k;
                if (e < gs.prow * gs.pcol) { const int r = e / gs.pcol, c = e - r * gs.pcol; st_wt<PPO_WT_C2>(a.thetaT + gs.t_off + c * gs.prow + r, to[k]); }
            }
        }
        if (gs.p_off >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (e0 + k < gs.p_count) a.par[gs.p_off + e0 + k] = to[k];
        }
        if (a.img) {
#pragma unroll
            for (int k = 0; k < 4; ++k) write_images(gs, a.img, e0 + k, to[k]);
        }
        if (tiled >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[4 * (tid & 7) + k][tid >> 3] = to[k];
        }
    }
    ASTAMP(4);
    if (tiled >= 0) {                                      // (block-uniform) the transposed tile: row r' = column 32 tj + r' of the matrix, 128 bytes per row
        __syncthreads();
        const int rp = tid >> 3, q4 = 4 * (tid & 7);
        const float4 o = make_float4(tt[rp][q4], tt[rp][q4 + 1], tt[rp][q4 + 2], tt[rp][q4 + 3]);
        st_wt4<PPO_WT_C2>(a.thetaT + t_toff + (size_t)(32 * tj + rp) * t_prow + 32 * ti + q4, o);
    }
    ASTAMP(5);
#ifdef PPO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(6);
#endif
    if (blockIdx.x == 0) {
        if (tid == 0) {
            a.beta_pow[2] = b1p * a.beta1;                                      // G:31217-31342 (after the applies)
            a.beta_pow[3] = b2p * a.beta2;
            if (a.norm_out) *a.norm_out = norm;
        }
        if (a.loss_row && tid < 5) {
            const float* tail = a.grad + (size_t)a.n_blocks * 256;
            const float n = tail[5];
            const float sum = tail[tid];
            float r = sum / n;
            if (tid == 1 || tid == 3) r = 0.5f * r;                             // vf_loss, approxkl carry the 0.5
            a.loss_row[tid] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Epoch preparation (ppo2.hpp:274-307 + 401-406).  Block k handles minibatch k of the epoch:
//   r    = flattened env-major source row that lands at permuted position k*M + i   (inverse permutation)
//   gidx = its time-major storage row (runner.hpp:136-152: r = e*T + t  ->  t*E + e)
//   stats[k] = { mean(adv), sqrt(mean((adv-mean)^2)) + 1e-8 }  with adv = returns - values
// inv_perm == null: a keyed bijection on [0,B) (multiply-xorshift rounds + cycle walking) plays the shuffle.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t keyed_bijection(uint32_t x, uint32_t bits, uint32_t mask, uint32_t k0, uint32_t k1) {
    const uint32_t sh = bits > 1 ? bits / 2 : 1;
    x = (x * (k0 | 1u) + k1) & mask;  x ^= x >> sh;
    x = (x * 0x9E3779B1u + (k0 >> 7)) & mask;  x ^= x >> sh;
    x = (x * (k1 | 1u) + 0x85EBCA6Bu) & mask;  x ^= x >> sh;
    x = (x * 0xC2B2AE35u + k0) & mask;  x ^= x >> sh;
    return x;
}

struct EpochArgs {
    const int* inv_perm;     // [B] or null
    const uint32_t* keys;    // {k0, k1} of this epoch (device memory: a replayed graph must see fresh keys)
    uint32_t bits;
    int B, M, T, E;
    const float* returns; const float* values;   // [T*E] storage order
    int* gidx;               // [B]
    float* stats;            // [B/M][2]
    int phase;               // 0: single rank, everything ; data parallel: 1 index + local sums, 2 local squared deviations, 3 finish
    float* xch;              // two vectors of B/M partial sums all-reduced by the host between the phases; the second starts at xch2
    int xch2;                // (a multiple of 4 floats: the peer all-reduce reads its source as 16-byte vectors)
    float n_global;          // minibatch rows over all ranks
    // literal data-parallel sampling (ppo_dist_global_shuffle): ONE permutation of the B * world rows of all ranks; returns / values
    // are the all-gathered [world][T][E] arrays, this rank trains rows [k Mg + rank M, k Mg + (rank + 1) M) of global minibatch k
    int world, rank;         // world > 1 selects this mode (phase 0: the statistics of the whole minibatch are computed locally)
};

#define EP_THREADS 1024             // one workgroup per minibatch: its M rows spread over 16 waves (256 threads left 16 rows per thread
                                    // in two dependent passes: 13.6 us per epoch at M = 2048)
__global__ __launch_bounds__(EP_THREADS) void epoch_prepare_kernel(EpochArgs a) {
    __shared__ float red[EP_THREADS / 64];
    __shared__ float s_mean;
    const int k = blockIdx.x, tid = threadIdx.x;
    if (a.phase == 3) {
        if (tid == 0) {
            a.stats[2 * k] = a.xch[k] / a.n_global;
            a.stats[2 * k + 1] = (float)((double)sqrtf(a.xch[a.xch2 + k] / a.n_global) + 1e-8);
        }
        return;
    }
    const uint32_t mask = (a.bits >= 32) ? 0xFFFFFFFFu : ((1u << a.bits) - 1u);
    if (a.world > 1) {
        // one global permutation (ppo2.hpp:288-307 applied to the rows of ALL ranks): position p of the permuted order holds flattened
        // env-major row r = e_global * T + t; rank e_global / E owns it, stored at [rank][t][e] of the gathered arrays
        const int Bg = a.B * a.world, Mg = a.M * a.world;
        auto src_of = [&](int posg) __attribute__((always_inline)) {
            int r;
            if (a.inv_perm) r = a.inv_perm[posg];
            else {
                uint32_t x = (uint32_t)posg;
                do { x = keyed_bijection(x, a.bits, mask, a.keys[0], a.keys[1]); } while (x >= (uint32_t)Bg);
                r = (int)x;
            }
            const int eg = r / a.T, t = r - eg * a.T, rs = eg / a.E, e = eg - rs * a.E;
            return rs * (a.T * a.E) + t * a.E + e;
        };
        float sum = 0.f;
        for (int i = tid; i < Mg; i += EP_THREADS) {
            const int s = src_of(k * Mg + i);
            const int li = i - a.rank * a.M;
            if (li >= 0 && li < a.M) a.gidx[k * a.M + li] = s;
            sum += a.returns[s] - a.values[s];
        }
        sum = wave_sum_lane0(sum);
        if ((tid & 63) == 0) red[tid >> 6] = sum;
        __syncthreads();
        if (tid == 0) { float tot = 0.f; for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w]; s_mean = tot / (float)Mg; }
        __syncthreads();
        const float mean = s_mean;
        float sq = 0.f;
        for (int i = tid; i < Mg; i += EP_THREADS) {
            const int s = src_of(k * Mg + i);
            const float d = (a.returns[s] - a.values[s]) - mean;
            sq += d * d;
        }
        sq = wave_sum_lane0(sq);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];
            a.stats[2 * k] = mean;
            a.stats[2 * k + 1] = (float)((double)sqrtf(tot / (float)Mg) + 1e-8);
        }
        return;
    }
    float sum = 0.f;
    if (a.phase != 2) {
        for (int i = tid; i < a.M; i += EP_THREADS) {
            const int pos = k * a.M + i;
            int r;
            if (a.inv_perm) r = a.inv_perm[pos];
            else {
                uint32_t x = (uint32_t)pos;
                do { x = keyed_bijection(x, a.bits, mask, a.keys[0], a.keys[1]); } while (x >= (uint32_t)a.B);
                r = (int)x;
            }
            const int s = (r % a.T) * a.E + (r / a.T);
            a.gidx[pos] = s;
            sum += a.returns[s] - a.values[s];
        }
        sum = wave_sum_lane0(sum);
        if ((tid & 63) == 0) red[tid >> 6] = sum;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];           // fixed order
            s_mean = tot / (float)a.M;
            if (a.phase == 1) a.xch[k] = tot;
        }
        __syncthreads();
        if (a.phase == 1) return;
    } else {
        if (tid == 0) s_mean = a.xch[k] / a.n_global;      // global mean of this minibatch
        __syncthreads();
    }
    const float mean = s_mean;
    float sq = 0.f;
    for (int i = tid; i < a.M; i += EP_THREADS) {
        const int s = a.gidx[k * a.M + i];
        const float d = (a.returns[s] - a.values[s]) - mean;
        sq += d * d;
    }
    sq = wave_sum_lane0(sq);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];
        if (a.phase == 2) a.xch[a.xch2 + k] = tot;
        else {
            const float var = tot / (float)a.M;
            a.stats[2 * k] = mean;
            a.stats[2 * k + 1] = (float)((double)sqrtf(var) + 1e-8);
        }
    }
}

// Materialise the permuted epoch (the reference's `perm * v` copies, ppo2.hpp:291-296) in minibatch order so that the
// train kernel reads contiguous rows with no index indirection: 16 rows per block; advantages are normalised here
// with the minibatch statistics of epoch_prepare_kernel.
struct GatherArgs {
    const int* gidx; const float* stats; int B, M, O, A;
    const float* obs; const float* act; const float* ret; const float* val; const float* nlp;
    float* mb_obs; float* mb_act; float* mb_adv; float* mb_ret; float* mb_val; float* mb_nlp;
#ifdef PPO_STAMPS
    unsigned long long* stamps;
#endif
};

__global__ __launch_bounds__(256) void epoch_gather_kernel(GatherArgs a) {
    __shared__ int src[16];
    const int pos0 = blockIdx.x * 16, tid = threadIdx.x;
    if (tid < 16 && pos0 + tid < a.B) src[tid] = a.gidx[pos0 + tid];
    __syncthreads();
    const int W = a.O + a.A;
    for (int i = tid; i < 16 * W; i += 256) {
        const int r = i / W, j = i - r * W;
        if (pos0 + r >= a.B) continue;
        if (j < a.O) a.mb_obs[(size_t)(pos0 + r) * a.O + j] = a.obs[(size_t)src[r] * a.O + j];
        else a.mb_act[(size_t)(pos0 + r) * a.A + (j - a.O)] = a.act[(size_t)src[r] * a.A + (j - a.O)];
    }
    if (tid < 16 && pos0 + tid < a.B) {
        const int p = pos0 + tid, s = src[tid], k = p / a.M;
        const float R = a.ret[s], V = a.val[s];
        a.mb_ret[p] = R; a.mb_val[p] = V; a.mb_nlp[p] = a.nlp[s];
        a.mb_adv[p] = ((R - V) - a.stats[2 * k]) / a.stats[2 * k + 1];          // ppo2.hpp:401-406
    }
}

// The same gather for observation / action widths that are multiples of 4 (configs[4]: 256 / 64), 16 bytes per access (a wave instruction of the element-wise form
// touches 64 x 4 bytes of 1 - 2 rows; this one 64 x 16), and -- bf16 path -- the observations written ONCE, as the bf16 operand rows the GEMMs read (X [B][Kp0]; the
// padding columns of a row are never written and stay zero): the fp32 copy of the epoch and the staging launch behind it (read it again, write bf16) go away.
// 16 rows per workgroup as above; same values, same statistics.
struct Gather4Args { GatherArgs g; __bf16* xe; int Kp0; };
__global__ __launch_bounds__(256) void epoch_gather4_kernel(Gather4Args q) {
    const GatherArgs& a = q.g;
    __shared__ int src[16];
    const int pos0 = blockIdx.x * 16, tid = threadIdx.x;
    if (tid < 16 && pos0 + tid < a.B) src[tid] = a.gidx[pos0 + tid];
    __syncthreads();
    const int QO = a.O >> 2, QW = (a.O + a.A) >> 2;
    typedef __bf16 bf16q_t __attribute__((ext_vector_type(4)));
    for (int i = tid; i < 16 * QW; i += 256) {
        const int r = i / QW, jq = i - r * QW;
        if (pos0 + r >= a.B) continue;
        if (jq < QO) {
            const float4 v = *reinterpret_cast<const float4*>(a.obs + (size_t)src[r] * a.O + 4 * jq);
            if (q.xe) { bf16q_t o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w; *reinterpret_cast<bf16q_t*>(q.xe + (size_t)(pos0 + r) * q.Kp0 + 4 * jq) = o; }
            else *reinterpret_cast<float4*>(a.mb_obs + (size_t)(pos0 + r) * a.O + 4 * jq) = v;
        } else {
            const int ja = 4 * (jq - QO);
            *reinterpret_cast<float4*>(a.mb_act + (size_t)(pos0 + r) * a.A + ja) = *reinterpret_cast<const float4*>(a.act + (size_t)src[r] * a.A + ja);
        }
    }
    if (tid < 16 && pos0 + tid < a.B) {
        const int p = pos0 + tid, s = src[tid], k = p / a.M;
        const float R = a.ret[s], V = a.val[s];
        a.mb_ret[p] = R; a.mb_val[p] = V; a.mb_nlp[p] = a.nlp[s];
        a.mb_adv[p] = ((R - V) - a.stats[2 * k]) / a.stats[2 * k + 1];          // ppo2.hpp:401-406
    }
}

// epoch_prepare_kernel (single rank, local shuffle) AND epoch_gather_kernel in one launch: EPG_SPLIT workgroups per minibatch each derive the
// minibatch's index map and advantage statistics (the same statements in the same order: every one of them holds the same bits; the redundant
// work is a few thousand integer hashes) and then copy their 1 / EPG_SPLIT share of its rows.  One launch and one dependent round trip through
// memory (the index map) per epoch less: the map never leaves the workgroup's LDS except as this epoch's `gidx` record.
#define EPG_SPLIT 8
#define EPG_MAX_M 8192              // rows of a minibatch the LDS copy of the index map holds (32 KB)
__global__ __launch_bounds__(EP_THREADS) void epoch_prepare_gather_kernel(EpochArgs a, GatherArgs ga) {
    __shared__ float red[EP_THREADS / 64];
    __shared__ float s_mean, s_den;
    __shared__ int s_idx[EPG_MAX_M];
    __shared__ float s_rv[2 * ((EPG_MAX_M + EPG_SPLIT - 1) / EPG_SPLIT)];       // returns | values of this workgroup's own rows (met in the first pass)
#ifdef PPO_STAMPS
#define ESTAMP(i) do { if (ga.stamps && threadIdx.x == 0) ga.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif
    ESTAMP(0);
    const int k = blockIdx.x / EPG_SPLIT, part = blockIdx.x % EPG_SPLIT, tid = threadIdx.x;
    const uint32_t mask = (a.bits >= 32) ? 0xFFFFFFFFu : ((1u << a.bits) - 1u);
    const int per = (a.M + EPG_SPLIT - 1) / EPG_SPLIT, r0 = part * per, r1 = min(a.M, r0 + per);
    // In-kernel stamps (tools/stamps_epoch.py): of this kernel's 36 k cycles, 6.6 k were the second pass gathering again what the first had
    // seen (a scattered 4-byte gather is 64 cache-line lookups per wave instruction) and 3 k the scalar fields' own gathers.  A thread's
    // advantages now stay in registers between the passes and the rows this workgroup copies leave their returns / values in LDS when the
    // first pass meets them: 25 k cycles, 17.5 -> 13 us.  (Measured without effect on the 9 k-cycle row copy: all its loads ahead of its
    // stores, 8-byte elements, shifts instead of the divisions -- it waits for scattered rows, not for instructions.)
    constexpr int RU = EPG_MAX_M / EP_THREADS;                                  // advantages per thread held in registers
    float adv[RU];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int i = tid + EP_THREADS * u;
        adv[u] = 0.f;
        if (i < a.M) {
            const int pos = k * a.M + i;
            int r;
            if (a.inv_perm) r = a.inv_perm[pos];
            else {
                uint32_t x = (uint32_t)pos;
                do { x = keyed_bijection(x, a.bits, mask, a.keys[0], a.keys[1]); } while (x >= (uint32_t)a.B);
                r = (int)x;
            }
            const int s = (r % a.T) * a.E + (r / a.T);
            s_idx[i] = s;
            if (part == 0) a.gidx[pos] = s;
            const float R = a.returns[s], V = a.values[s];
            if (i >= r0 && i < r1) { s_rv[2 * (i - r0)] = R; s_rv[2 * (i - r0) + 1] = V; }
            adv[u] = R - V;
            sum += adv[u];
        }
    }
    ESTAMP(1);
    sum = wave_sum_lane0(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];           // fixed order
        s_mean = tot / (float)a.M;
    }
    __syncthreads();
    ESTAMP(2);
    const float mean = s_mean;
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        if (tid + EP_THREADS * u < a.M) { const float d = adv[u] - mean; sq += d * d; }
    }
    ESTAMP(3);
    sq = wave_sum_lane0(sq);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];
        const float var = tot / (float)a.M;
        s_den = (float)((double)sqrtf(var) + 1e-8);
        if (part == 0) { a.stats[2 * k] = mean; a.stats[2 * k + 1] = s_den; }
    }
    __syncthreads();
    ESTAMP(4);
    const float den = s_den;
    // ---- this workgroup's rows of the minibatch: [r0, r1) --------------------------------------------------------------------------------
    const int W = ga.O + ga.A, total = (r1 - r0) * W;
    int rr = tid / W, jj = tid - rr * W;                                        // element tid of the [rows][W] block; then + EP_THREADS per sweep
    const int dq = EP_THREADS / W, dr = EP_THREADS - dq * W;
    for (int i = tid; i < total; i += EP_THREADS) {
        const int r = r0 + rr, p = k * a.M + r, s = s_idx[r];
        if (jj < ga.O) ga.mb_obs[(size_t)p * ga.O + jj] = ga.obs[(size_t)s * ga.O + jj];
        else ga.mb_act[(size_t)p * ga.A + (jj - ga.O)] = ga.act[(size_t)s * ga.A + (jj - ga.O)];
        rr += dq; jj += dr;
        if (jj >= W) { jj -= W; ++rr; }
    }
    ESTAMP(5);
    for (int r = r0 + tid; r < r1; r += EP_THREADS) {
        const int p = k * a.M + r, s = s_idx[r];
        const float R = s_rv[2 * (r - r0)], V = s_rv[2 * (r - r0) + 1];
        ga.mb_ret[p] = R; ga.mb_val[p] = V; ga.mb_nlp[p] = ga.nlp[s];
        ga.mb_adv[p] = ((R - V) - mean) / den;                                  // ppo2.hpp:401-406
    }
    ESTAMP(6);
#ifdef PPO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ESTAMP(7);
#endif
}

// inv[perm[i]] = i   (out.row(perm[i]) = in.row(i), ppo2.hpp:291-296)
__global__ void invert_perm_kernel(const int* perm, int* inv, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) inv[perm[i]] = i;
}

// standalone advantage normalisation of an explicit minibatch (ppo2.hpp:401-406), single block
__global__ __launch_bounds__(1024) void adv_normalize_kernel(const float* returns, const float* values, int n, float* advs) {
    __shared__ float red[16];
    __shared__ float s_val;
    const int tid = threadIdx.x;
    float sum = 0.f;
    for (int i = tid; i < n; i += 1024) sum += returns[i] - values[i];
    sum = wave_sum_lane0(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int w = 0; w < 16; ++w) s += red[w]; s_val = s / (float)n; }
    __syncthreads();
    const float mean = s_val;
    float sq = 0.f;
    for (int i = tid; i < n; i += 1024) { const float d = (returns[i] - values[i]) - mean; sq += d * d; }
    sq = wave_sum_lane0(sq);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int w = 0; w < 16; ++w) s += red[w]; s_val = (float)((double)sqrtf(s / (float)n) + 1e-8); }
    __syncthreads();
    const float denom = s_val;
    for (int i = tid; i < n; i += 1024) advs[i] = ((returns[i] - values[i]) - mean) / denom;
}

// column means of the loss rows (ppo2.hpp:335)
__global__ void loss_mean_kernel(const float* rows, int n, float* mean) {      // launch with 5 * 64 threads
    const int j = threadIdx.x >> 6, ln = threadIdx.x & 63;
    float s = 0.f;
    for (int i = ln; i < n; i += 64) s += rows[i * 5 + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (ln == 0) mean[j] = s / (float)n;
}

// ------------------------------------------------------------------------------------------------------------
// GAE(lambda) (runner.hpp:159-191): one thread per env, serial in t, coalesced across envs (time-major buffers)
// ------------------------------------------------------------------------------------------------------------
__global__ void gae_kernel(const float* rewards, const float* values, const float* dones, const float* last_values,
                           const float* last_dones, int T, int E, float gamma, float lam, float* returns) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float last = 0.f;
    float nv = last_values[e];
    float nnt = 1.0f - last_dones[e];
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * E + e;
        const float v = values[i];
        const float delta = rewards[i] + gamma * (nv * nnt) - v;
        last = delta + (gamma * lam) * (nnt * last);
        returns[i] = last + v;
        nv = v;
        nnt = 1.0f - dones[i];
    }
}

// Few environments, long rollouts (the reference's own command line: ONE env x 2048 steps): the scan above is 2048 dependent
// iterations of global loads on one lane (280 us).  One workgroup per env: every temporal-difference term delta_t and every
// continuation flag are formed by all threads in parallel (the same expressions on the same operands) and parked in LDS; one lane then
// runs the recurrence last = delta_t + (gamma lam) (nnt_t last) -- two dependent operations per step, its operands read ahead.
// Bit-identical to gae_kernel (same operations in the same order per element).
#define GAE_LONG_THREADS 256
__global__ __launch_bounds__(GAE_LONG_THREADS) void gae_long_kernel(const float* rewards, const float* values, const float* dones, const float* last_values,
                                                                   const float* last_dones, int T, int E, float gamma, float lam, float* returns) {
    extern __shared__ __attribute__((aligned(16))) float gl_lds[];
    float* s_delta = gl_lds; float* s_nnt = gl_lds + T; float* s_v = gl_lds + 2 * T;
    const int e = blockIdx.x;
    for (int t = threadIdx.x; t < T; t += GAE_LONG_THREADS) {
        const size_t i = (size_t)t * E + e;
        const float v = values[i];
        const float nv = t == T - 1 ? last_values[e] : values[i + E];
        const float nnt = 1.0f - (t == T - 1 ? last_dones[e] : dones[i + E]);
        s_delta[t] = rewards[i] + gamma * (nv * nnt) - v;
        s_nnt[t] = nnt; s_v[t] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float last = 0.f;
        const float gl = gamma * lam;
        int t = T - 1;
        for (; t >= 7; t -= 8) {
            float d[8], n[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = s_delta[t - k]; n[k] = s_nnt[t - k]; v[k] = s_v[t - k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { last = d[k] + gl * (n[k] * last); s_delta[t - k] = last + v[k]; }
        }
        for (; t >= 0; --t) { last = s_delta[t] + gl * (s_nnt[t] * last); s_delta[t] = last + s_v[t]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += GAE_LONG_THREADS) returns[(size_t)t * E + e] = s_delta[t];
}

// ------------------------------------------------------------------------------------------------------------
// EnvNormalize::step for a whole batch of envs (env_normalize.hpp:64-116) in ONE multi-block launch:
//   obs job   : obs_rms.update(raw_obs)   (:94-98; scale + clip are fused into the next policy step's input staging)
//   reward job: ret = ret*gamma + r ; ret_rms.update(ret) ; out = clip(r / sqrt(var + eps)) ; ret *= 1 - done   (:64-92)
// Every workgroup reduces a chunk of rows to (n, mean, M2) with the reference's two passes (mean, then squared
// deviations: common/running_statistics.hpp:26-54).  The workgroup that finishes LAST (device-scope counter; write-through
// partials, see NB_ST) combines the chunks in index order with M2 = sum_k M2_k + sum_k n_k (mean_k - mean)^2 --
// algebraically the two-pass result over the whole batch, deterministic, and better conditioned than one long fp32
// sum -- and applies RunningStatistics::update's merge (:88-104; count is double, every matrix op fp32).
// Data-parallel: the last workgroup instead publishes this rank's (n, mean, M2) in its slot of `xch`; ONE all-reduce
// of the zero-padded slot table (= an all-gather) later, norm_finalize_kernel combines the ranks in rank order, so the
// statistics are over the environments of all ranks and bit-identical on every rank.
// ------------------------------------------------------------------------------------------------------------
#define NB_THREADS 256
#define NB_MAX_OBS_BLOCKS 256
#define NB_MAX_REW_BLOCKS 64
#define NB_REW_STRIDE 32      // floats between the reward chunks' (n, mean, M2) sets: one 128-byte line each

struct NormBatchArgs {
    const float* obs; int rows; int D; NormDev obs_st; int g_obs; int rows_per_obs_block;      // obs == null: no obs job
    const float* rew; const float* dones; float* ret; NormDev ret_st; float* rew_out; float* done_copy;   // rew == null: no reward job
    int rew_rows; int training_rew; int g_rew; int rows_per_rew_block;
    int scale_rew;        // EnvNormalize::norm_reward (env_normalize.hpp:75): 0 = rewards pass through unscaled and unclipped
    float gamma, clip_rew, eps;
    float* part;          // [g_obs][part_stride] then [g_rew][NB_REW_STRIDE]: every chunk's set on 128-byte lines of its own
    int part_stride;      // 1 + 2D rounded up to 32 floats
    unsigned* counter;    // zero between launches
    float* xch;           // data-parallel: [world][(1 + 2D) + 3] slot table (this rank's slot is written here), else null
    int world, rank;
    int use_peer;         // data-parallel over peer-mapped regions: the table is exchanged by these two kernels themselves (ppo_peer.hpp)
    PeerDev peer;
    // wide observations (D a multiple of 64: BASELINE configs[4] has 256): the obs job is dealt as n_strips = D / 64 column groups x n_splits row
    // splits (g_obs = n_strips * n_splits workgroups of rows_per_obs_block rows), see obs_cgroup_job.  0: the row-chunk form above
    int n_strips, n_splits;
#ifdef PPO_STAMPS
    unsigned long long* stamps;   // diagnostic builds only: [block][8] cycle stamps
#endif
};
#ifdef PPO_STAMPS
#define NSTAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// The chunks' partial sets travel from their workgroups to the last arriver of the job like weight_grad_assemble_kernel's slabs: stored
// write-through (agent-scope relaxed atomic stores = sc1), every storing thread drains its stores before the workgroup barrier in front of the
// ONE relaxed arrival, and the last arriver reads them with agent-scope loads (sc1: lines it has not touched in this launch, never served
// from its own XCD's L2).  The language model's pair -- a release fence in every workgroup, an acquire in the last -- costs a write-back of the
// whole L2 per workgroup: 3.3 - 3.8 k cycles of this 28 k-cycle kernel (tools/stamps_norm.py).
// "Has not touched" is made true by the layout, not assumed: no 128-byte line of the hand-off is shared between two workgroups -- the sets are
// part_stride / NB_REW_STRIDE floats apart (multiples of 32 on a 256-byte-aligned base) and the reward chunks are whole multiples of 32 rows of
// `ret` (enqueue_norm_batch), so the plain load a chunk owner makes of its OWN rows of `ret` never brings a neighbour's rows into its XCD's L2.
__device__ __forceinline__ void NB_ST(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float NB_LD(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (n, mean[D], M2[D]) of rows [r0, r1) of a row-major [*, D] matrix -> out[0], out[1 .. D], out[1+D .. 2D].
// A thread's first 8 values stay in registers between the two passes (one memory round trip for the usual chunk size).
// pre != null (D == 1 only): the thread's first 8 values, rows r0 + tid + 256 u, are handed over in registers instead of being re-read.
__device__ __forceinline__ void chunk_moments(const float* __restrict__ x, int r0, int r1, int D, float* out, float* sh /* 2*NB_THREADS */, const float* pre = nullptr) {
    const int tid = threadIdx.x;
    const float n = (float)(r1 - r0);
    if (tid == 0) NB_ST(out, n);
    float* cm = sh + NB_THREADS;                         // chunk means of the current column group
    constexpr int U = 8;
    for (int cbase = 0; cbase < D; cbase += NB_THREADS) {
        const int Dg = min(NB_THREADS, D - cbase);
        const int rpp = NB_THREADS / Dg;                 // rows per sweep: consecutive threads read consecutive addresses
        const int col = cbase + tid % Dg, rsub = tid / Dg;
        const bool act = rsub < rpp;
        const int top = pow2_ceil(rpp) >> 1;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int r = r0 + rsub + u * rpp; v[u] = (act && r < r1) ? (pre ? pre[u] : x[(size_t)r * D + col]) : 0.f; }
        for (int pass = 0; pass < 2; ++pass) {
            const float bm = pass ? cm[tid % Dg] : 0.f;
            float t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float a0 = v[u] - bm;
                t[u] = (act && r0 + rsub + u * rpp < r1) ? (pass ? a0 * a0 : a0) : 0.f;
            }
            float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            if (act) for (int r = r0 + rsub + U * rpp; r < r1; r += rpp) { const float a0 = x[(size_t)r * D + col] - bm; s += pass ? a0 * a0 : a0; }
            __syncthreads();                             // previous users of sh are done
            sh[tid] = s;                                 // == sh[rsub * Dg + tid % Dg]
            __syncthreads();
            for (int h = top; h > 0; h >>= 1) {          // fixed-shape tree over the row sweeps
                if (act && rsub < h && rsub + h < rpp) sh[tid] += sh[tid + h * Dg];
                __syncthreads();
            }
            if (tid < Dg) {
                if (pass == 0) cm[tid] = sh[tid] / n;    // colwise().mean()
                else { NB_ST(out + 1 + cbase + tid, cm[tid]); NB_ST(out + 1 + D + cbase + tid, sh[tid]); }
            }
            __syncthreads();
        }
    }
}

// Whole block: combine K <= NB_THREADS (n, mean[D], M2[D]) sets (stride floats apart) for the column group
// [cbase, cbase + Dg).  NB_THREADS / Dg threads share a column (strided over the sets) and meet in a fixed-shape tree,
// so the result does not depend on scheduling.  Threads tid < Dg return their column's batch (n, mean, M2).
// One column, K <= 64 sets (the reward job: its chunks are at most NB_MAX_REW_BLOCKS): the same sums in ONE wave.  The block-wide form puts set
// k in thread k and adds in a tree of strides 128 .. 1 over 256 slots of which only the first K are non-zero; strides 32 .. 1 over the 64 lanes
// of a wave pair the same values in the same order (the upper levels only ever add zeros), so the bits are the same -- without its 20
// workgroup barriers.  Every thread of the block returns the result (through sh).
__device__ __forceinline__ void combine_single_column(const float* sets, int K, int stride, float* sh, float& n, float& mean, float& M2) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid < 64) {
        const bool have = tid < K;
        const float nk = have ? NB_LD(sets + (size_t)tid * stride) : 0.f;
        const float mk = have ? NB_LD(sets + (size_t)tid * stride + 1) : 0.f;
        const float qk = have ? NB_LD(sets + (size_t)tid * stride + 2) : 0.f;
        float nn = 0.f;
        for (int k = 0; k < K; ++k) nn += __shfl(nk, k);                         // index order, like the block-wide form
        float s = have ? nk * mk : 0.f;
        for (int h = 32; h > 0; h >>= 1) s += __shfl_down(s, h);
        const float bm = __shfl(s, 0) / nn;
        const float d = mk - bm;
        float q = have ? qk + nk * (d * d) : 0.f;
        for (int h = 32; h > 0; h >>= 1) q += __shfl_down(q, h);
        if (tid == 0) { sh[0] = nn; sh[1] = bm; sh[2] = q; }
    }
    __syncthreads();
    n = sh[0]; mean = sh[1]; M2 = sh[2];
}

__device__ __forceinline__ void combine_group(const float* sets, int K, int stride, int D, int cbase, int Dg, float* sh /* 3*NB_THREADS */,
                                              float& n, float& mean, float& M2) {
    const int tid = threadIdx.x;
    float* cm = sh + NB_THREADS;
    float* shn = sh + 2 * NB_THREADS;
    const int rpp = NB_THREADS / Dg;
    const int cg = tid % Dg, col = cbase + cg, ksub = tid / Dg;
    const bool act = ksub < rpp;
    const int top = pow2_ceil(rpp) >> 1;
    // everything this thread will need of its first CG sets is requested up front, together with the sets' row counts: ONE trip to the
    // memory side (the sets were written through by other workgroups) instead of one per pass.  (16: the column-group form of wide observations
    // combines 64 sets over 4 threads per column; with 4 the other 12 were 48 dependent round trips, 21.8 k cycles -- tools/stamps_norm_wide.py)
    constexpr int CG = 16;
    float pm[CG], pq[CG];
#pragma unroll
    for (int j = 0; j < CG; ++j) {
        const int k = ksub + j * rpp;
        const bool ok = act && k < K;
        pm[j] = ok ? NB_LD(sets + (size_t)k * stride + 1 + col) : 0.f;
        pq[j] = ok ? NB_LD(sets + (size_t)k * stride + 1 + D + col) : 0.f;
    }
    __syncthreads();
    if (tid < K) shn[tid] = NB_LD(sets + (size_t)tid * stride);
    __syncthreads();
    float nn = 0.f;
    for (int k = 0; k < K; ++k) nn += shn[k];
    float q = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const float bm = pass ? cm[cg] : 0.f;
        float s = 0.f;
        if (act) {
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const int k = ksub + j * rpp;
                if (k < K) {
                    const float nk = shn[k], mk = pm[j];
                    if (pass == 0) s += nk * mk;
                    else { const float d = mk - bm; s += pq[j] + nk * (d * d); }
                }
            }
            for (int k = ksub + CG * rpp; k < K; k += rpp) {
                const float nk = shn[k], mk = NB_LD(sets + (size_t)k * stride + 1 + col);
                if (pass == 0) s += nk * mk;
                else { const float d = mk - bm; s += NB_LD(sets + (size_t)k * stride + 1 + D + col) + nk * (d * d); }
            }
        }
        __syncthreads();
        sh[tid] = s;
        __syncthreads();
        for (int h = top; h > 0; h >>= 1) {
            if (act && ksub < h && ksub + h < rpp) sh[tid] += sh[tid + h * Dg];
            __syncthreads();
        }
        if (tid < Dg) { if (pass == 0) cm[tid] = sh[tid] / nn; else q = sh[tid]; }
        __syncthreads();
    }
    n = nn; mean = cm[cg]; M2 = q;
}

// RunningStatistics::update's merge of a batch (mean, M2, n) into column c (common/running_statistics.hpp:88-104)
// (old_mean / old_var: the column's statistics before the merge, read by the caller ahead of the combine)
__device__ __forceinline__ float merge_column(NormDev st, int c, double cnt, float bmean, float bM2, float nbf, float old_mean, float old_var) {
    const double nb = (double)nbf;
    const double tot = cnt + nb;
    const float bvar = bM2 / (float)nb;                                        // :51-54
    const float delta = bmean - old_mean;                                      // :90
    const float new_mean = old_mean + (delta * (float)nb) / (float)tot;        // :94
    const float m_a = old_var * (float)cnt;                                    // :97
    const float m_b = bvar * (float)nb;                                        // :98
    const float M2 = m_a + m_b + (((delta * delta) * (float)cnt) * (float)nb) / (float)tot;  // :100
    const float v = M2 / (float)tot;                                           // :101
    st.mean[c] = new_mean;
    st.var[c] = v;
    return v;
}

// Whole block.  which = 0: obs statistics, 1: reward branch.  sets: K sets (chunks of this rank, or the ranks' slots).
// publish != null: write the combined local set there instead of merging (data-parallel first half).
__device__ __forceinline__ void norm_finish(const NormBatchArgs& a, int which, const float* sets, int K, int stride, float* publish,
                                            float* sh /* 3*NB_THREADS */) {
    const int tid = threadIdx.x;
    const int D = a.D;
    if (which == 0) {
        const double cnt = *a.obs_st.count;
        float ntot = 0.f;
        for (int cbase = 0; cbase < D; cbase += NB_THREADS) {
            const int Dg = min(NB_THREADS, D - cbase);
            float n, mean, M2, om = 0.f, ov = 0.f;
            if (!publish && tid < Dg) { om = a.obs_st.mean[cbase + tid]; ov = a.obs_st.var[cbase + tid]; }      // in flight during the combine
            combine_group(sets, K, stride, D, cbase, Dg, sh, n, mean, M2);
            if (tid < Dg) {
                const int c = cbase + tid;
                if (publish) { publish[0] = n; publish[1 + c] = mean; publish[1 + D + c] = M2; }
                else merge_column(a.obs_st, c, cnt, mean, M2, n, om, ov);
            }
            ntot = n;
        }
        __syncthreads();
        if (tid == 0 && !publish) *a.obs_st.count = (double)ntot + cnt;        // :103
        return;
    }
    // reward branch: the first batch of the apply pass (4096 rows) is in flight while the statistics are combined
    constexpr int U = 16;
    float dn[U], rw[U], rt[U];
    if (!publish) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + NB_THREADS * u;
            const bool ok = i < a.rew_rows;
            dn[u] = ok ? a.dones[i] : 0.f; rw[u] = ok ? a.rew[i] : 0.f; rt[u] = ok ? NB_LD(a.ret + i) : 0.f;
        }
    }
    float var = a.ret_st.var[0];
    const float old_ret_mean = a.ret_st.mean[0];
    if (a.training_rew) {
        const double cnt = *a.ret_st.count;
        float n, mean, M2;
        if (K <= 64) combine_single_column(sets, K, stride, sh, n, mean, M2);
        else combine_group(sets, K, stride, 1, 0, 1, sh, n, mean, M2);
        if (tid == 0) {
            if (publish) { publish[0] = n; publish[1] = mean; publish[2] = M2; }
            else {
                var = merge_column(a.ret_st, 0, cnt, mean, M2, n, old_ret_mean, var);
                *a.ret_st.count = (double)n + cnt;
            }
        }
    }
    if (publish) return;
    __syncthreads();
    if (tid == 0) sh[0] = var;
    __syncthreads();
    const float inv = 1.0f / sqrtf(sh[0] + a.eps);                             // env_normalize.hpp:76-79
    for (int base = 0; base < a.rew_rows; base += NB_THREADS * U) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + tid + NB_THREADS * u;
                const bool ok = i < a.rew_rows;
                dn[u] = ok ? a.dones[i] : 0.f; rw[u] = ok ? a.rew[i] : 0.f; rt[u] = ok ? NB_LD(a.ret + i) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + tid + NB_THREADS * u;
            if (i >= a.rew_rows) continue;
            float y = rw[u];
            if (a.scale_rew) { y = y * inv; y = tf_min(tf_max(y, -a.clip_rew), a.clip_rew); }
            a.rew_out[i] = y;
            a.ret[i] = rt[u] * (1.0f - dn[u]);                                 // :84-90
            if (a.done_copy) a.done_copy[i] = dn[u];
        }
    }
}

// ---- obs job for WIDE observations ----------------------------------------------------------------------------------------------------
// The row-chunk form gives a workgroup whole rows: at D = 256 a sweep of 256 threads is ONE row, a thread walks its chunk row by row, and the
// last arriver combines every chunk's 2 D floats through one CU (hence at most 64 chunks for wide rows): 75.7 us for 8192 x 256 observations,
// 0.22 TB/s (profiles/r04_z_kernel_stats_cfg5.csv).  Here (D a multiple of 64) workgroup b owns the 64-COLUMN GROUP b % n_cg of the rows of split
// b / n_cg: thread (q, rs) = 16-byte strip q of the group in rows rs, rs + 16, ... -- a wave reads 4 rows x 256 contiguous bytes per instruction
// (a first version with one 4-column strip per workgroup had every lane of a load on its own cache line: 23.7 k cycles for 8 loads per thread,
// tools/stamps_norm_wide.py) -- 8 rows per thread stay in registers (ONE trip to memory for splits of up to 128 rows), the two passes of
// common/running_statistics.hpp:26-54 run out of registers, the 16 row-subs of a strip meet in a fixed tree (two lane exchanges inside a wave,
// then the four waves through LDS).  What crosses workgroups is a (n, mean[64], M2[64]) set per (group, split); the LAST split of a group combines
// the group's sets with the row-chunk form's Chan combine (combine_group: index order) and merges / publishes its 64 columns; the last GROUP to
// finish writes the count (every group's merge has read the old one by then).  Same hand-off as the row-chunk form: write-through sets on lines of
// their own (NB_CG_STRIDE floats apart), drained stores, barrier, one relaxed arrival, agent-scope loads.  Fixed orders: bitwise reproducible.
// counter[2 + group]: the groups' arrival words.
#define NB_CG_STRIDE 160                  // 1 + 2 * 64 floats rounded up to whole 128-byte lines
#define NB_CG_MAX_WG 1024                 // workgroups of the column-group form (the table of sets is sized for it)
#define NB_CG_KP 8                        // sets a thread of the group's last arriver combines: at most 16 * NB_CG_KP row splits
// 16-byte write-through load of bytes another workgroup of this launch stored write-through; the caller waits with nb_wait8 before the first use
__device__ __forceinline__ f32x4 nb_ld4_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void nb_wait8(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f, f32x4& g, f32x4& h) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "memory");
}
__device__ __forceinline__ void obs_cgroup_job(const NormBatchArgs& a, int b, float* sh, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_cg = a.n_strips, cg = b % n_cg, split = b / n_cg, D = a.D;
    const int q = tid & 15, rs = tid >> 4, c0 = 64 * cg + 4 * q;
    const int r0 = split * a.rows_per_obs_block, r1 = min(a.rows, r0 + a.rows_per_obs_block);
    const float n = (float)(r1 - r0);
    constexpr int U = 8;
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int r = r0 + rs + 16 * u; x[u] = r < r1 ? *reinterpret_cast<const float4*>(a.obs + (size_t)r * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f); }
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, M2[4];
    float* red = sh;                                               // [4 waves][16 strips][4]
    for (int pass = 0; pass < 2; ++pass) {
        float t[4][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r0 + rs + 16 * u < r1;
            const float v[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mean[j]; t[j][u] = ok ? (pass ? d * d : d) : 0.f; }
        }
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = ((t[j][0] + t[j][1]) + (t[j][2] + t[j][3])) + ((t[j][4] + t[j][5]) + (t[j][6] + t[j][7]));
        for (int r = r0 + rs + 16 * U; r < r1; r += 16) {                                  // splits longer than 128 rows: the rest from memory, both passes
            const float4 w = *reinterpret_cast<const float4*>(a.obs + (size_t)r * D + c0);
            const float v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mean[j]; s[j] += pass ? d * d : d; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += __shfl_xor(s[j], 16); s[j] += __shfl_xor(s[j], 32); }      // the wave's four row-subs: (0 + 1) + (2 + 3)
        __syncthreads();                                                                   // (red is free again)
        if (lane < 16) { *reinterpret_cast<float4*>(red + (wave * 16 + q) * 4) = make_float4(s[0], s[1], s[2], s[3]); }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tot = (red[(0 * 16 + q) * 4 + j] + red[(1 * 16 + q) * 4 + j]) + (red[(2 * 16 + q) * 4 + j] + red[(3 * 16 + q) * 4 + j]);
            if (pass == 0) mean[j] = tot / n; else M2[j] = tot;                            // colwise().mean() ; sum of squared deviations
        }
    }
    NSTAMP(1);
    float* out = a.part + (size_t)b * NB_CG_STRIDE;                 // set layout of this form: [mean 64 | M2 64]; its row count follows from the split index
    if (tid < 16) {
        st_wt4<true>(out + 4 * q, make_float4(mean[0], mean[1], mean[2], mean[3]));
        st_wt4<true>(out + 64 + 4 * q, make_float4(M2[0], M2[1], M2[2], M2[3]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    NSTAMP(2);
    if (tid == 0) *flag = (__hip_atomic_fetch_add(a.counter + 2 + cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.n_splits - 1u) ? 1 : 0;
    __syncthreads();
    NSTAMP(3);
    if (!*flag) return;
    // ---- last split of this column group: combine the splits (index order), then merge (or publish) columns 64 cg .. 64 cg + 63 ----------
    float* publish = a.xch ? a.xch + (size_t)a.rank * (1 + 2 * D + 3) : nullptr;
    const double cnt = *a.obs_st.count;
    float om = 0.f, ov = 0.f;
    if (!publish && tid < 64) { om = a.obs_st.mean[64 * cg + tid]; ov = a.obs_st.var[64 * cg + tid]; }       // in flight during the combine
    // Thread (q, ks): strip q of the group, sets ks, ks + 16, ... (at most NB_CG_KP of them: n_splits <= 16 NB_CG_KP), every 16-byte piece of them requested
    // in ONE trip to the memory side (write-through loads; 4-byte agent-scope loads of the same 64 sets took 18 k cycles: ~75 per wave instruction).
    // Chan's combine in a fixed shape: a thread adds its sets in index order, the 16 set-subs of a strip meet like the row-subs above.
    const int ks = tid >> 4;
    f32x4 pm[NB_CG_KP], pq[NB_CG_KP];
    float nk[NB_CG_KP];
#pragma unroll
    for (int j = 0; j < NB_CG_KP; ++j) {
        const int k = min(ks + 16 * j, a.n_splits - 1);              // (past the end: a valid address, weight 0)
        const float* st = a.part + ((size_t)k * n_cg + cg) * NB_CG_STRIDE;
        pm[j] = nb_ld4_sc1(st + 4 * q); pq[j] = nb_ld4_sc1(st + 64 + 4 * q);
        const int kk = ks + 16 * j;
        nk[j] = kk < a.n_splits ? (float)(min(a.rows, (kk + 1) * a.rows_per_obs_block) - kk * a.rows_per_obs_block) : 0.f;
    }
    nb_wait8(pm[0], pm[1], pm[2], pm[3], pm[4], pm[5], pm[6], pm[7]);
    nb_wait8(pq[0], pq[1], pq[2], pq[3], pq[4], pq[5], pq[6], pq[7]);
    const float nn = (float)a.rows;
    float bmv[4] = {0.f, 0.f, 0.f, 0.f}, bqv[4];
    for (int pass = 0; pass < 2; ++pass) {
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NB_CG_KP; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (pass == 0) sc[c] += nk[j] * pm[j][c];
                else { const float d = pm[j][c] - bmv[c]; sc[c] += nk[j] > 0.f ? pq[j][c] + nk[j] * (d * d) : 0.f; }
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) { sc[c] += __shfl_xor(sc[c], 16); sc[c] += __shfl_xor(sc[c], 32); }
        __syncthreads();
        if (lane < 16) *reinterpret_cast<float4*>(red + (wave * 16 + q) * 4) = make_float4(sc[0], sc[1], sc[2], sc[3]);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tot = (red[(0 * 16 + q) * 4 + c] + red[(1 * 16 + q) * 4 + c]) + (red[(2 * 16 + q) * 4 + c] + red[(3 * 16 + q) * 4 + c]);
            if (pass == 0) bmv[c] = tot / nn; else bqv[c] = tot;
        }
    }
    NSTAMP(7);
    __syncthreads();
    if (tid < 16) { *reinterpret_cast<float4*>(sh + 64 + 4 * tid) = make_float4(bmv[0], bmv[1], bmv[2], bmv[3]); *reinterpret_cast<float4*>(sh + 128 + 4 * tid) = make_float4(bqv[0], bqv[1], bqv[2], bqv[3]); }
    __syncthreads();
    if (tid < 64) {
        const int c = 64 * cg + tid;
        const float bm = sh[64 + tid], bq = sh[128 + tid];
        if (publish) { if (c == 0) NB_ST(publish, nn); NB_ST(publish + 1 + c, bm); NB_ST(publish + 1 + D + c, bq); }
        else merge_column(a.obs_st, c, cnt, bm, bq, nn, om, ov);
    }
    NSTAMP(4);
    if (tid == 0) __hip_atomic_store(a.counter + 2 + cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // ready for the next launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the old count has been read, the published columns are out
    __syncthreads();
    NSTAMP(5);
    if (tid == 0) *flag = (__hip_atomic_fetch_add(a.counter + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)n_cg - 1u) ? 1 : 0;
    __syncthreads();
    NSTAMP(6);
    if (!*flag) return;
    // ---- last group: every group has read the old count ------------------------------------------------------------------------------------
    if (tid == 0) {
        if (!publish) *a.obs_st.count = (double)(float)a.rows + cnt; // :103
        a.counter[0] = 0u;
    }
    if (a.use_peer && publish) { __syncthreads(); peer_stats_publish(a.peer, 0, publish, 0, 1 + 2 * D); }
}

__global__ __launch_bounds__(NB_THREADS) void norm_batch_kernel(NormBatchArgs a) {
    __shared__ float sh[3 * NB_THREADS];
    __shared__ int is_last;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int so = 1 + 2 * a.D, ps = a.part_stride;
    float* part_rew = a.part + (size_t)a.g_obs * (a.n_strips ? NB_CG_STRIDE : ps);
    const int which = b < a.g_obs ? 0 : 1;
    NSTAMP(0);
#ifdef PPO_STAMPS
    if (a.stamps && threadIdx.x == 0) { a.stamps[(size_t)blockIdx.x * 8 + 4] = 0; a.stamps[(size_t)blockIdx.x * 8 + 5] = 0; a.stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)which; }
#endif
    if (which == 0 && a.n_strips) { obs_cgroup_job(a, b, sh, &is_last); return; }
    if (which == 0) {
        const int r0 = b * a.rows_per_obs_block, r1 = min(a.rows, r0 + a.rows_per_obs_block);
        chunk_moments(a.obs, r0, r1, a.D, a.part + (size_t)b * ps, sh);
    } else {
        const int k = b - a.g_obs;
        const int r0 = k * a.rows_per_rew_block, r1 = min(a.rew_rows, r0 + a.rows_per_rew_block);
        // ret = ret * gamma + r (:66).  A thread's first 8 rows (r0 + tid + 256 u) stay in registers for the statistics below.
        float nv[8];
        {
            float rt[8], rw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = r0 + tid + NB_THREADS * u; const bool ok = i < r1; rt[u] = ok ? a.ret[i] : 0.f; rw[u] = ok ? a.rew[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = r0 + tid + NB_THREADS * u;
                nv[u] = rt[u] * a.gamma + rw[u];
                if (i < r1) NB_ST(a.ret + i, nv[u]);                                   // (written through: the job's last arriver reads every chunk's rows)
            }
        }
        constexpr int U = 4;
        for (int base = r0 + 8 * NB_THREADS; base < r1; base += NB_THREADS * U) {
            float rt[U], rw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int i = base + tid + NB_THREADS * u; const bool ok = i < r1; rt[u] = ok ? a.ret[i] : 0.f; rw[u] = ok ? a.rew[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < U; ++u) { const int i = base + tid + NB_THREADS * u; if (i < r1) NB_ST(a.ret + i, rt[u] * a.gamma + rw[u]); }
        }
        if (a.training_rew) chunk_moments(a.ret, r0, r1, 1, part_rew + (size_t)k * NB_REW_STRIDE, sh, nv);
    }
    // last-block-done, one counter per job: the partials (and the reward job's returns) are written through, every thread drains its stores, the
    // barrier orders them before the ONE relaxed arrival, and the job's last arriver reads them with agent-scope loads (NB_ST / NB_LD above).
    // The two jobs finish independently (two workgroups run the two tails side by side).
    NSTAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores (partials; the reward job's `ret`) are complete
    __syncthreads();
    NSTAMP(2);
    const unsigned total = which == 0 ? (unsigned)a.g_obs : (unsigned)a.g_rew;
    if (tid == 0) is_last = (__hip_atomic_fetch_add(a.counter + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) ? 1 : 0;
    __syncthreads();
    NSTAMP(3);
    if (!is_last) return;
    NSTAMP(4);
    float* publish = a.xch ? a.xch + (size_t)a.rank * (so + 3) + (which ? so : 0) : nullptr;
    if (which == 0) norm_finish(a, 0, a.part, a.g_obs, ps, publish, sh);
    else norm_finish(a, 1, part_rew, a.g_rew, NB_REW_STRIDE, publish, sh);
    NSTAMP(5);
    if (tid == 0) a.counter[which] = 0u;
    if (a.use_peer && publish) {
        // this rank's batch moments go straight into every rank's gather area: no all-reduce launches between the two statistics kernels.
        // (a reward job that is not training publishes nothing: its three floats are not read either)
        __syncthreads();
        peer_stats_publish(a.peer, which, publish, which ? so : 0, which ? 3 : so);
    }
}

// data-parallel second half: combine the ranks' slots (after the all-reduce) and finish; one block
__global__ __launch_bounds__(NB_THREADS) void norm_finalize_kernel(NormBatchArgs a) {      // grid = 2: block 0 obs, block 1 reward
    __shared__ float sh[3 * NB_THREADS];
    const int so = 1 + 2 * a.D;
    if (a.use_peer) {                                          // the ranks' slots arrive here: wait for the flags, copy into the local table
        const int job = blockIdx.x;
        if ((job == 0 && !a.obs) || (job == 1 && !a.rew)) return;
        if (!peer_stats_collect(a.peer, job, a.xch, so + 3, job ? so : 0, job ? 3 : so)) return;
    }
    if (blockIdx.x == 0) { if (a.obs) norm_finish(a, 0, a.xch, a.world, so + 3, nullptr, sh); }
    else if (a.rew) norm_finish(a, 1, a.xch + so, a.world, so + 3, nullptr, sh);
}

// normalise + clip an [rows, D] batch with frozen statistics (env_normalize.hpp:99-104)
__global__ void obs_normalize_kernel(const float* in, int rows, int D, NormDev st, float eps, float clip, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int j = i % D;
    float x = (in[i] - st.mean[j]) * (1.0f / sqrtf(st.var[j] + eps));
    x = tf_min(tf_max(x, -clip), clip);
    out[i] = x;
}

// on-device seeded synthetic env (oracle/ppo_oracle.c orc_seeded_env_step): thread per (env, lane)
__global__ void seeded_env_kernel(uint32_t seed, int env0, int n_envs, uint32_t step, int O, float* obs, float* rew, float* dones) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = O + 2;
    if (i >= n_envs * W) return;
    const int e = i / W, j = i - e * W;
    const uint32_t h = ctr_hash(seed, (uint32_t)(env0 + e), step, (uint32_t)j);
    if (j < O) obs[(size_t)e * O + j] = u32_to_sym_unit(h);
    else if (j == O) { if (rew) rew[e] = u32_to_sym_unit(h); }
    else if (dones) dones[e] = (h % 300u == 0u) ? 1.0f : 0.0f;
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
