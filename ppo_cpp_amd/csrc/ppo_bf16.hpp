// ppo_bf16.hpp -- the bf16 matrix-core path of libppo_hip.so for wide networks (BASELINE configs[4]: 256 obs / 64 act,
// MLP [1024,1024,1024], 8192 envs).  gfx950 only.
//
// The fused 16-row-tile kernels of ppo_kernels.hpp stream every weight once per 16 rows: 16 FLOP per weight byte, fine
// against the fp32 matrix rate, a quarter of what v_mfma_f32_16x16x32_bf16 needs (64 FLOP/B per CU against the L1 fill
// rate).  Wide nets at thousands of rows per minibatch are therefore run layer by layer as 128x128-tile GEMMs
// (both towers batched in one launch), activations round-tripping through HBM/L2 as bf16 (8 MB per layer at
// 4096 x 1024, microseconds at HBM rates).  Master weights, gradients, the global-norm clip and Adam stay fp32
// (adam_kernel of ppo_kernels.hpp, unchanged); bf16_mirror_kernel refreshes the bf16 operand copies after every step.
//
// ONE GEMM form serves every product of the train step.  Both operands are stored with the REDUCTION index contiguous
// ("NT": C[i][j] = sum_k A[i][k] * B[j][k]), which is exactly what the 16x16x32 operand map wants (8 consecutive k per
// lane = one 16-byte LDS read) and what LDS-DMA can stage without a transpose.  Every activation / gradient matrix is
// therefore written in two layouts by the epilogue that produces it, [rows][features] and [features][rows]:
//     forward   H_{l+1} = tanh(X_l W_l + b)        A = X_l   [M][K]      B = W_l^T  [N][K]   (transposed bf16 mirror)
//     backward  dY_{l-1} = (dY_l W_l^T) .* (1-H^2)  A = dY_l  [M][N]      B = W_l    [K][N]   (straight bf16 mirror)
//     weights   dW_l = X_l^T dY_l                   A = X_l^T [K][M]      B = dY_l^T [N][M]   (split over M, fp32 slabs)
// Reference arithmetic replaced: G:6889-9187 (train forward), G:11773-23699 (backward), with bf16 operands and fp32
// accumulation; there is no reference counterpart for the precision (cfg 5 tolerance ~1e-2 on losses).
#pragma once

#include "ppo_kernels.hpp"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GB_N 128                   // output tile columns (index j)
#define GB_K 64                    // reduction depth per LDS stage: 128-byte rows
#define GB_STAGES 3                // LDS ring: stage t+2 is in flight while stage t is multiplied
#define GB_PAD 128                 // every dimension of the bf16 path is padded to this
// WM = waves along i (2 or 4): tile rows BM = 64 WM, 2 WM waves (WM x 2), each a 64 x 64 accumulator block (4 x 4 MFMA tiles)
#define GB_BM(WM) (64 * (WM))
#define GB_THREADS(WM) (128 * (WM))
#define GB_STAGE_BYTES(WM) ((GB_BM(WM) + GB_N) * GB_K * 2)         // A tile then B tile: 48 KB (WM 4) / 32 KB (WM 2)
#define GB_LDS_BYTES(WM) (GB_STAGES * GB_STAGE_BYTES(WM))          // 144 KB / 96 KB: one workgroup per CU
#define GB_PIECES(WM) ((GB_BM(WM) + GB_N) / 8 / (2 * (WM)))        // 1-KB LDS-DMA pieces per wave and stage: 6 / 8

enum { GEPI_TANH = 0, GEPI_TANHGRAD = 1, GEPI_F32 = 2, GEPI_DW = 3 };

struct GemmArgs {
    const bf16_t* A[2]; const bf16_t* B[2];   // per tower; row-major, reduction index contiguous
    int lda, ldb;
    int K;                                    // reduction length
    int tiles_i;                              // tiles along i (block id -> (ti, tj) = (id % tiles_i, id / tiles_i))
    const float* bias[2];                     // [J] fp32 (TANH, F32) or null
    const bf16_t* HT[2]; int ldht;            // TANHGRAD: tanh outputs of this layer in the [features][rows] layout
    bf16_t* C[2]; int ldc;                    // out [I][J] bf16 (may be null)
    bf16_t* CT[2]; int ldct;                  // out [J][I] bf16 (may be null)
    float* F[2]; int ldf;                     // F32: out [I][J] fp32
    float* bsum[2]; int bsum_ld;              // TANHGRAD: per row-tile column sums of the fp32 outputs, [tiles_i][bsum_ld] (bias gradients; may be null)
};

// 16-byte chunk c (0..7) of row r of a [rows][64 bf16] LDS tile lives at chunk c ^ (r & 7): a 16-lane group of a
// ds_read_b128 (16 different rows, same k range) then spreads over 8 slots of the 256-byte bank row instead of 2.
// LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address and again on the read.
__device__ __forceinline__ int gb_swz(int row, int chunk) { return chunk ^ (row & 7); }

// One stage = the [BM][64] A tile followed by the [128][64] B tile, (BM + 128) / 8 pieces of 1 KB (8 rows x 128 B), dealt
// to the waves round robin; lane l of a piece lands on (row 8p + l/8, chunk l%8) and fetches the chunk that belongs there.
template <int WM>
__device__ __forceinline__ void gb_stage(const bf16_t* __restrict__ A, int lda, int i0, const bf16_t* __restrict__ B, int ldb, int j0, int k0, char* st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PA = GB_BM(WM) / 8;
#pragma unroll
    for (int q = 0; q < GB_PIECES(WM); ++q) {
        const int piece = wave + 2 * WM * q;                 // wave-uniform
        const bool isA = piece < PA;
        const int r = (isA ? piece : piece - PA) * 8 + (lane >> 3), c = lane & 7;
        const bf16_t* src = isA ? A + (size_t)(i0 + r) * lda + k0 + gb_swz(r, c) * 8 : B + (size_t)(j0 + r) * ldb + k0 + gb_swz(r, c) * 8;
        typedef const __attribute__((address_space(1))) void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(st + piece * 1024), 16, 0, 0);
    }
}

// fragment of MFMA k-step ks (0/1) for the 16 rows starting at r0: lane (r = lane & 15, g = lane >> 4) holds
// k = 32 ks + 8 g .. + 7 of row r0 + r
__device__ __forceinline__ bf16x8 gb_frag(const char* lds_tile, int r0, int ks) {
    const int lane = threadIdx.x & 63;
    const int r = r0 + (lane & 15), c = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + gb_swz(r, c) * 16);
}

struct GemmAcc { f32x4 v[4][4]; };           // [mi][ni]: rows 16 mi + 4 g + reg, column 16 ni + (lane & 15) of the wave's 64 x 64 block

template <int N> __device__ __forceinline__ void gb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Main loop.  Per step ONE barrier: [wait until this wave's pieces of stage t have landed: all but the youngest stage's
// loads] -> barrier (every wave's pieces of stage t are in LDS; every wave is done multiplying stage t-1) -> issue stage
// t+2 into the buffer stage t-1 used -> read fragments of stage t, 32 MFMAs per wave.  A __syncthreads() would drain the
// LDS-DMA queue (it waits vmcnt(0)), so the barrier is the raw instruction and the wait is counted.
template <int WM>
__device__ __forceinline__ void gb_mainloop(GemmAcc& acc, const bf16_t* __restrict__ A, int lda, int i0, const bf16_t* __restrict__ B, int ldb, int j0,
                                            int kbeg, int K, char* lds) {
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    constexpr int SB = GB_STAGE_BYTES(WM), NP = GB_PIECES(WM);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc.v[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = K / GB_K;
    gb_stage<WM>(A, lda, i0, B, ldb, j0, kbeg, lds);
    if (nt > 1) gb_stage<WM>(A, lda, i0, B, ldb, j0, kbeg + GB_K, lds + SB);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gb_wait_vm<NP>(); else gb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) {
            int nb = cur + 2; if (nb >= GB_STAGES) nb -= GB_STAGES;
            gb_stage<WM>(A, lda, i0, B, ldb, j0, kbeg + (t + 2) * GB_K, lds + nb * SB);
        }
        const char* at = lds + cur * SB;
        const char* bt = at + GB_BM(WM) * GB_K * 2;
        // all 16 fragment reads of the stage go out first; the first k-step's MFMAs start when its 8 have landed while the
        // second k-step's are still in flight (two register sets: 64 accumulator + 64 fragment VGPRs per lane)
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a) af[ks][a] = gb_frag(at, wm + 16 * a, ks);
#pragma unroll
            for (int b = 0; b < 4; ++b) bfr[ks][b] = gb_frag(bt, wn + 16 * b, ks);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc.v[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][a], bfr[ks][b], acc.v[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (++cur == GB_STAGES) cur = 0;
    }
    __syncthreads();                                         // the epilogue reuses the ring as its staging tile
}

// Epilogue through LDS: the accumulator layout has 4 consecutive ROWS per lane for one column, i.e. 8 contiguous bytes
// of the [column][row] image; that image is parked in LDS (row stride BM + 8 elements: conflict-light 8-byte writes),
// streamed out as the [J][I] output with 16-byte stores, and gathered column-wise for the [I][J] output.
template <int WM, int EPI>
__device__ __forceinline__ void gb_epilogue_bf16(GemmAcc& acc, const GemmArgs& a, int tw, int i0, int j0, char* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int g = lane >> 4, c = lane & 15;
    constexpr int BM = GB_BM(WM), TLD = BM + 8, NT = GB_THREADS(WM);
    bf16_t* tt = reinterpret_cast<bf16_t*>(lds);              // [128 cols j][TLD rows i]
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int j = wn + 16 * nb + c;
        float bj = 0.f;
        if constexpr (EPI == GEPI_TANH) bj = a.bias[tw][j0 + j];
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) {
            const int i = wm + 16 * ma + 4 * g;
            float y[4];
            if constexpr (EPI == GEPI_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fast_tanh(acc.v[ma][nb][r] + bj);
            } else {                                          // TanhGrad: dY = dX .* (1 - h^2), h from the [features][rows] copy
                const bf16x4 h4 = *reinterpret_cast<const bf16x4*>(a.HT[tw] + (size_t)(j0 + j) * a.ldht + i0 + i);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float h = (float)h4[r]; y[r] = acc.v[ma][nb][r] * (1.0f - h * h); }
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)y[r];
            *reinterpret_cast<bf16x4*>(tt + j * TLD + i) = o;
            if constexpr (EPI == GEPI_TANHGRAD) csum[nb] += (y[0] + y[1]) + (y[2] + y[3]);
        }
    }
    // bias gradients ride along: db[j] = sum over rows of dY[:, j].  Each wave adds its 64 rows (16 values per lane, then the
    // four lane groups), the WM waves that share the columns meet in LDS behind the parked image, and the tile's sums go
    // to row `ti` of a small [row tiles][features] table the gradient assembly adds up in tile order (fixed order).
    float* cs = reinterpret_cast<float*>(lds + GB_N * TLD * 2);           // [WM][128]
    if constexpr (EPI == GEPI_TANHGRAD) {
        if (a.bsum[tw]) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float v = csum[nb];
                v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                if (g == 0) cs[(wave >> 1) * GB_N + wn + 16 * nb + c] = v;
            }
        }
    }
    __syncthreads();
    if constexpr (EPI == GEPI_TANHGRAD) {
        if (a.bsum[tw] && tid < GB_N) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < WM; ++q) v += cs[q * GB_N + tid];
            a.bsum[tw][(size_t)(i0 / BM) * a.bsum_ld + j0 + tid] = v;
        }
    }
    if (a.CT[tw]) {                                           // [J][I]: rows of the parked image, BM/8 chunks of 16 B per row
        constexpr int CH = BM / 8;
#pragma unroll
        for (int q = 0; q < GB_N * CH / NT; ++q) {
            const int id = tid + NT * q;
            const int j = id / CH, ch = id % CH;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(tt + j * TLD + ch * 8);
            *reinterpret_cast<bf16x8*>(a.CT[tw] + (size_t)(j0 + j) * a.ldct + i0 + ch * 8) = v;
        }
    }
    if (a.C[tw]) {                                            // [I][J]: 8 consecutive j of one row i = 8 two-byte LDS gathers
#pragma unroll
        for (int q = 0; q < BM * 16 / NT; ++q) {
            const int id = tid + NT * q;
            const int i = id % BM, ch = id / BM;              // consecutive lanes = consecutive i: same LDS dwords pairwise, no conflicts
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tt[(ch * 8 + e) * TLD + i];
            *reinterpret_cast<bf16x8*>(a.C[tw] + (size_t)(i0 + i) * a.ldc + j0 + ch * 8) = v;
        }
    }
}

template <int WM, int EPI>
__global__ __launch_bounds__(GB_THREADS(WM)) void gemm_nt_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char gb_lds[];
    const int tw = blockIdx.y;
    const int ti = blockIdx.x % a.tiles_i, tj = blockIdx.x / a.tiles_i;
    const int i0 = ti * GB_BM(WM), j0 = tj * GB_N;
    GemmAcc acc;
    gb_mainloop<WM>(acc, a.A[tw], a.lda, i0, a.B[tw], a.ldb, j0, 0, a.K, gb_lds);
    if constexpr (EPI == GEPI_F32) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, g = lane >> 4, c = lane & 15;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int j = j0 + wn + 16 * nb + c;
            const float bj = a.bias[tw] ? a.bias[tw][j] : 0.f;
#pragma unroll
            for (int ma = 0; ma < 4; ++ma)
#pragma unroll
                for (int r = 0; r < 4; ++r) a.F[tw][(size_t)(i0 + wm + 16 * ma + 4 * g + r) * a.ldf + j] = acc.v[ma][nb][r] + bj;
        }
    } else {
        gb_epilogue_bf16<WM, EPI>(acc, a, tw, i0, j0, gb_lds);
    }
}

// ---- weight gradients: every matrix of both towers in one grouped launch, split over the minibatch rows -------------
struct DwTileB { const bf16_t* A; const bf16_t* B; int lda, ldb; int i0, j0; int out_off, ldo; int is_x0; };
struct DwArgsB { const DwTileB* tiles; int nsplit; int rows_per_split; float* slabs; size_t slab_stride;
                 const bf16_t* x0T; int x0_ld; };     // epoch-staged transposed observations of this minibatch (null: the tiles' own x0T)

template <int WM>
__global__ __launch_bounds__(GB_THREADS(WM)) void gemm_dw_bf16_kernel(DwArgsB a) {
    extern __shared__ __attribute__((aligned(16))) char gb_lds[];
    const DwTileB t = a.tiles[blockIdx.x / a.nsplit];
    const int split = blockIdx.x % a.nsplit;                 // the splits of one tile sit on different XCDs; tiles of one split share X^T / dY^T panels
    GemmAcc acc;
    const bool ov = t.is_x0 && a.x0T;
    gb_mainloop<WM>(acc, ov ? a.x0T : t.A, ov ? a.x0_ld : t.lda, t.i0, t.B, t.ldb, t.j0, split * a.rows_per_split, a.rows_per_split, gb_lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, g = lane >> 4, c = lane & 15;
    float* out = a.slabs + (size_t)split * a.slab_stride + t.out_off;
#pragma unroll
    for (int ma = 0; ma < 4; ++ma)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* row = out + (size_t)(t.i0 + wm + 16 * ma + 4 * g + r) * t.ldo + t.j0 + wn + c;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) row[16 * nb] = acc.v[ma][nb][r];
        }
}

// ---- input staging: fp32 observations -> bf16 [rows_pad][Kp0] (+ the [Kp0][rows_pad] copy the first layer's weight
// gradient needs); the act path normalises here (env_normalize.hpp:99-104) and writes the normalised fp32 rows into
// the rollout buffer, as stage_block_inputs does for the fused kernels --------------------------------------------------
struct StageArgsB { const float* obs; int n, O, Kp0, rows_pad; ObsNorm nz; float* obs_out; bf16_t* X; bf16_t* XT; int ldt; };

__global__ __launch_bounds__(256) void bf16_stage_kernel(StageArgsB a) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.rows_pad * a.Kp0) return;
    const int row = (int)(idx / a.Kp0), j = (int)(idx - (size_t)row * a.Kp0);
    float x = 0.f;
    if (row < a.n && j < a.O) {
        x = a.obs[(size_t)row * a.O + j];
        if (a.nz.enabled) {
            x = (x - a.nz.mean[j]) * (1.0f / sqrtf(a.nz.var[j] + a.nz.eps));
            x = tf_min(tf_max(x, -a.nz.clip), a.nz.clip);
        }
        if (a.obs_out) a.obs_out[(size_t)row * a.O + j] = x;
    }
    a.X[idx] = (bf16_t)x;
    if (a.XT) a.XT[(size_t)j * a.ldt + row] = (bf16_t)x;
}

// ---- act epilogue: sampling + neglogp (G:5894-6672) from the head GEMM's fp32 outputs ---------------------------------
struct SampleArgsB {
    const float* head[2]; int ldh;      // [rows_pad][Ap] fp32: tower 0 = mu, tower 1 column 0 = value
    const float* logstd;                // fp32 master
    const float* noise; float* action; float* det_action; float* value; float* neglogp;
    int n, A; uint32_t seed, rng_step, row_base;
};

__global__ __launch_bounds__(256) void bf16_sample_kernel(SampleArgsB a) {
    const int r = threadIdx.x >> 4, part = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + r;
    const bool live = row < a.n;
    float ssq = 0.f, slog = 0.f;
    for (int j = part; j < a.A; j += 16) {
        const float mu = live ? a.head[0][(size_t)row * a.ldh + j] : 0.f;
        const float logstd = mu * 0.0f + a.logstd[j];
        const float sigma = expf(logstd);
        float eps = 0.f;
        if (live) eps = a.noise ? a.noise[(size_t)row * a.A + j] : ctr_normal(a.seed, a.row_base + row, a.rng_step, j);
        const float act = mu + sigma * eps;
        const float z = (act - mu) / sigma;
        ssq += z * z; slog += logstd;
        if (live) {
            if (a.action) a.action[(size_t)row * a.A + j] = act;
            if (a.det_action) a.det_action[(size_t)row * a.A + j] = mu;
        }
    }
    ssq = group16_sum(ssq); slog = group16_sum(slog);
    if (part == 0 && live) {
        if (a.neglogp) a.neglogp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)a.A + slog;
        if (a.value) a.value[row] = a.head[1][(size_t)row * a.ldh];
    }
}

// ---- loss + its gradient w.r.t. the head outputs (G:9428-11290, G:12609-22656): the arithmetic of
// train_fwd_bwd_kernel's middle section, 16 lanes per row, fp32; writes d mu / d v as bf16 in both layouts and the
// per-block partial sums (bias / logstd gradients, loss terms) the gradient assembly adds up in a fixed order ---------
struct LossArgsB {
    const float* head[2]; int ldh;
    const float* logstd;
    const float* actions; const float* advs; const float* returns; const float* old_values; const float* old_neglogp;
    const float* hyper;                 // {lr, cliprange}
    int n, A, Ap, rows_pad; float inv_n, ent_coef, vf_coef;
    bf16_t* dhead[2]; bf16_t* dheadT[2];   // [rows_pad][Ap], [Ap][rows_pad]  (tower 1: only column / row 0 is ever non-zero)
    float* slots[2]; int slot_w, slot_head, slot_aux, slot_loss;
};

#define BL_ROWS 32                 // rows per block of the loss kernel (16 lanes each): one slot row of partial sums per block
__global__ __launch_bounds__(16 * BL_ROWS) void bf16_loss_kernel(LossArgsB a) {
    extern __shared__ __attribute__((aligned(16))) float ls[];      // [R][Ap] dmu | [R][Ap] dlogstd | [R][4] pi terms | [R][2] vf terms
    float* dmu_s = ls; float* dls_s = ls + BL_ROWS * a.Ap; float* pt = dls_s + BL_ROWS * a.Ap; float* vt = pt + 4 * BL_ROWS;
    const int tid = threadIdx.x, r = tid >> 4, part = tid & 15;
    const int row = blockIdx.x * BL_ROWS + r;
    const bool live = row < a.n;
    const float cr = a.hyper[1];
    const float g = a.inv_n;
    // policy tower
    float ssq = 0.f, slog = 0.f, sent = 0.f;
    for (int j = part; j < a.A; j += 16) {
        const float mu = a.head[0][(size_t)row * a.ldh + j];
        const float logstd = mu * 0.0f + a.logstd[j];
        const float act = live ? a.actions[(size_t)row * a.A + j] : mu;
        const float z = (act - mu) / expf(logstd);
        ssq += z * z; slog += logstd; sent += logstd + HALF_LOG_2PIE;
    }
    ssq = group16_sum(ssq); slog = group16_sum(slog); sent = group16_sum(sent);
    const float nlp = 0.5f * ssq + HALF_LOG_2PI * (float)a.A + slog;
    const float adv = live ? a.advs[row] : 0.f;
    const float old_nlp = live ? a.old_neglogp[row] : nlp;
    const float lo = 1.0f - cr, hi = 1.0f + cr;
    const float ratio = expf(old_nlp - nlp);
    const float rmin = tf_min(ratio, hi);
    const float rclip = tf_max(rmin, lo);
    const float m1 = -adv * ratio, m2 = -adv * rclip;
    const float sel = (m1 >= m2) ? 1.0f : 0.0f;                                       // G:12609
    const float pass = ((rmin >= lo) ? 1.0f : 0.0f) * ((ratio <= hi) ? 1.0f : 0.0f);  // G:15357, 16113
    float d_ratio = (-adv) * g * sel;
    d_ratio += (-adv) * g * (1.0f - sel) * pass;
    const float d_nlp = live ? -(d_ratio * ratio) : 0.0f;
    if (part == 0) {
        const float dk = nlp - old_nlp;
        pt[r * 4 + 0] = live ? tf_max(m1, m2) : 0.f;
        pt[r * 4 + 1] = live ? sent : 0.f;
        pt[r * 4 + 2] = live ? dk * dk : 0.f;
        pt[r * 4 + 3] = (live && fabsf(ratio - 1.0f) > cr) ? 1.0f : 0.f;
    }
    for (int j = part; j < a.Ap; j += 16) {
        float dmu = 0.f, dl = 0.f;
        if (j < a.A && live) {
            const float mu = a.head[0][(size_t)row * a.ldh + j];
            const float sigma = expf(mu * 0.0f + a.logstd[j]);
            const float z = (a.actions[(size_t)row * a.A + j] - mu) / sigma;
            dl = d_nlp * (1.0f - z * z) - a.ent_coef * g;                            // AddN_2 G:21299
            dmu = d_nlp * (-(z / sigma)) + dl * 0.0f;                                // AddN_3 G:22656
        }
        dmu_s[r * a.Ap + j] = dmu; dls_s[r * a.Ap + j] = dl;
        a.dhead[0][(size_t)row * a.Ap + j] = (bf16_t)dmu;
    }
    // value tower (G:10213-10837, G:14975-19571)
    if (part == 0) {
        float dv = 0.f, lossv = 0.f;
        if (live) {
            const float v = a.head[1][(size_t)row * a.ldh];
            const float R = a.returns[row], vo = a.old_values[row];
            const float dvo = v - vo;
            const float vmin = tf_min(dvo, cr);
            const float vclip = vo + tf_max(vmin, -cr);
            const float e1 = v - R, e2 = vclip - R;
            const float s1 = e1 * e1, s2 = e2 * e2;
            lossv = tf_max(s1, s2);
            const float gv = a.vf_coef * 0.5f * a.inv_n;
            const float selv = (s1 >= s2) ? 1.0f : 0.0f;
            const float passv = ((vmin >= -cr) ? 1.0f : 0.0f) * ((dvo <= cr) ? 1.0f : 0.0f);
            dv = gv * selv * (2.0f * e1) + gv * (1.0f - selv) * (2.0f * e2) * passv;
        }
        vt[r * 2] = dv; vt[r * 2 + 1] = lossv;
        a.dhead[1][(size_t)row * a.Ap] = (bf16_t)dv;
        a.dheadT[1][row] = (bf16_t)dv;
    }
    __syncthreads();
    // the [features][rows] copy of d mu from the LDS tile: 8 consecutive rows of one feature = one 16-byte store
    for (int i = tid; i < a.Ap * (BL_ROWS / 8); i += 16 * BL_ROWS) {
        const int j = i / (BL_ROWS / 8), pr = i - j * (BL_ROWS / 8);
        bf16x8 v;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (bf16_t)dmu_s[(8 * pr + q) * a.Ap + j];
        *reinterpret_cast<bf16x8*>(a.dheadT[0] + (size_t)j * a.rows_pad + (size_t)blockIdx.x * BL_ROWS + 8 * pr) = v;
    }
    float* s0 = a.slots[0] + (size_t)blockIdx.x * a.slot_w;
    float* s1 = a.slots[1] + (size_t)blockIdx.x * a.slot_w;
    for (int j = tid; j < a.Ap; j += 16 * BL_ROWS) {
        float sb = 0.f, sl = 0.f;
        for (int q = 0; q < BL_ROWS; ++q) { sb += dmu_s[q * a.Ap + j]; sl += dls_s[q * a.Ap + j]; }
        s0[a.slot_head + j] = sb;                              // db_mu  (fp32 sums of the fp32 values, not of their bf16 roundings)
        s0[a.slot_aux + j] = sl;                               // dlogstd
    }
    if (tid >= 256 && tid < 260) { const int k = tid - 256; float s = 0.f; for (int q = 0; q < BL_ROWS; ++q) s += pt[q * 4 + k]; s0[a.slot_loss + k] = s; }
    if (tid == 320) { float sb = 0.f, sl = 0.f; for (int q = 0; q < BL_ROWS; ++q) { sb += vt[q * 2]; sl += vt[q * 2 + 1]; } s1[a.slot_head] = sb; s1[a.slot_loss] = sl; }
}

// ---- bias gradients of the hidden layers: db[j] = sum over rows of dY, one wave per row of the [features][rows] copy ----
struct RowSumArgsB { const bf16_t* src[2 * PPO_MAX_LAYERS]; float* dst[2 * PPO_MAX_LAYERS]; int rows[2 * PPO_MAX_LAYERS]; int first[2 * PPO_MAX_LAYERS + 1]; int n_mats; int ld; int len; };

__global__ __launch_bounds__(256) void bf16_rowsum_kernel(RowSumArgsB a) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int m = 0;
    while (m + 1 < a.n_mats && w >= a.first[m + 1]) ++m;
    const int j = w - a.first[m];
    if (j >= a.rows[m]) return;
    const bf16_t* p = a.src[m] + (size_t)j * a.ld;
    float s = 0.f;
    for (int i = lane * 8; i < a.len; i += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) a.dst[m][j] = s;
}

// ---- bf16 operand mirrors of the fp32 master weights: the straight copy keeps theta's padded layout (a cast of the
// whole vector); the transposed copies come from a table of matrices, 32 x 32 tiles through LDS ------------------------
__global__ __launch_bounds__(256) void bf16_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    bf16x4 o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    reinterpret_cast<bf16x4*>(dst)[i] = o;
}

struct TrMat { int src_off, dst_off, rows, cols, first_tile; };      // src [rows][cols] fp32 -> dst [cols][rows] bf16; tiles of 32 x 32
struct TrArgs { const TrMat* mats; int n_mats; const float* src; bf16_t* dst; };

__global__ __launch_bounds__(256) void bf16_transpose_kernel(TrArgs a) {
    __shared__ float tile[32][33];
    int m = 0;
    while (m + 1 < a.n_mats && (int)blockIdx.x >= a.mats[m + 1].first_tile) ++m;
    const TrMat t = a.mats[m];
    const int tl = blockIdx.x - t.first_tile, tc = t.cols / 32;
    const int r0 = (tl / tc) * 32, c0 = (tl % tc) * 32;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[y + 8 * k][x] = a.src[t.src_off + (size_t)(r0 + y + 8 * k) * t.cols + c0 + x];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) a.dst[t.dst_off + (size_t)(c0 + y + 8 * k) * t.rows + r0 + x] = (bf16_t)tile[x][y + 8 * k];
}

// ---- gradient assembly for the bf16 path: one WAVE per 256-element chunk of the padded parameter vector, four consecutive
// elements per lane (16-byte slab loads); the chunk's sum of squares is a wave reduction.  Same sources as grad_reduce_kernel
// (kind 0 slabs, kind 1 per-block slots, kind 3 per-row-tile bias sums), same fixed summation orders. ---------------------
__global__ __launch_bounds__(256) void bf16_grad_reduce_kernel(ReduceArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    // highest chunks first: the slot-summed vectors (head bias, logstd: hundreds of dependent-free but latency-bound loads per
    // lane) sit at the END of the parameter vector and must not be the launch's tail
    const int chunk = a.n_blocks - (int)(blockIdx.x * 4 + (tid >> 6));
    if (chunk < 0) return;
    if (chunk == a.n_blocks) {                              // loss tail: {pg, vf, ent, kl, cf, rows}
#pragma unroll
        for (int q = 0; q < 5; ++q) {                       // lanes take the slot rows round robin, then meet in a fixed-shape tree
            const int tower = (q == 1) ? 1 : 0, off = a.slot_loss + (q <= 1 ? 0 : q - 1);
            float s = 0.f;
            for (int b = lane; b < a.n_rowblocks; b += 64) s += a.slots[tower][(size_t)b * a.slot_w + off];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) a.grad[(size_t)a.n_blocks * 256 + q] = s;
        }
        if (lane == 5) a.grad[(size_t)a.n_blocks * 256 + 5] = a.n_local;
        if (lane == 6) { a.beta_pow[0] = a.beta_pow[2]; a.beta_pow[1] = a.beta_pow[3]; }
        return;
    }
    const GradSrc s = a.src[chunk];
    const size_t idx = (size_t)chunk * 256 + 4 * lane;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (s.kind == 0) {
        for (int k = 0; k < a.nsplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(a.slabs + (size_t)k * a.slab_stride + idx);
            g[0] += v.x; g[1] += v.y; g[2] += v.z; g[3] += v.w;
        }
    } else if (s.kind == 1 || s.kind == 3) {
        // rows of a small table, 16 bytes per lane and row, all loads independent (offsets are multiples of 4 floats on this path)
        const int e0 = (int)(idx - (size_t)s.base);
        const float* p = (s.kind == 1 ? a.slots[s.tower] : a.direct) + s.slot_off + e0;
        const int rows = s.kind == 1 ? a.n_rowblocks : a.n_direct;
        const size_t stride = s.kind == 1 ? (size_t)a.slot_w : (size_t)a.direct_stride;
        if (e0 < s.count) {
#pragma unroll 8
            for (int b = 0; b < rows; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
                g[0] += v.x; g[1] += v.y; g[2] += v.z; g[3] += v.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (e0 + q >= s.count) g[q] = 0.f;
        }
    }
    *reinterpret_cast<float4*>(a.grad + idx) = make_float4(g[0], g[1], g[2], g[3]);
    float q = (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) a.sumsq[chunk] = q;
}
