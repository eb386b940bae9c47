// ppo_bf16.hpp -- the bf16 matrix-core path of libppo_hip.so for wide networks (BASELINE configs[4]: 256 obs / 64 act,
// MLP [1024,1024,1024], 8192 envs).  gfx950 only.
//
// The fused 16-row-tile kernels of ppo_kernels.hpp stream every weight once per 16 rows: 16 FLOP per weight byte, fine
// against the fp32 matrix rate, a quarter of what v_mfma_f32_16x16x32_bf16 needs (64 FLOP/B per CU against the L1 fill
// rate).  Wide nets at thousands of rows per minibatch are therefore run layer by layer as 128x128-tile GEMMs
// (both towers batched in one launch), activations round-tripping through HBM/L2 as bf16 (8 MB per layer at
// 4096 x 1024, microseconds at HBM rates).  Master weights, gradients, the global-norm clip and Adam stay fp32
// (adam_kernel of ppo_kernels.hpp, which also keeps the bf16 copy of the weights current: bf16_cast_kernel only runs after a weight upload).
//
// Every activation / gradient matrix lives in ONE layout, [rows][features] (round 3; rounds 1-2 wrote a second, [features][rows],
// copy of each from the producing epilogue so that the weight-gradient product could use the same GEMM form: 16 MB more stores per
// launch, and the epilogues were 76 of the 177 us of a train step's forward + backward).  Two GEMM forms:
//     forward   H_{l+1} = tanh(X_l W_l + b)        "NT"  A = X_l   [M][K]      B = W_l^T  [N][K]   (transposed bf16 mirror)
//     backward  dY_{l-1} = (dY_l W_l^T) .* (1-H^2)  "NT"  A = dY_l  [M][N]      B = W_l    [K][N]   (straight bf16 mirror)
//     weights   dW_l = X_l^T dY_l                   "TN"  A = X_l   [M][K]      B = dY_l   [M][N]   (split over M, fp32 slabs)
// NT: both operands have the REDUCTION index contiguous, which is what the 16x16x32 operand map wants (8 consecutive k per lane =
// one 16-byte LDS read) and what LDS-DMA can stage without a transpose.  TN: the reduction index is the ROW of both operands; the
// tiles are staged as they lie (LDS-DMA, [64 rows][128 features] images) and the fragments come out of LDS through
// ds_read_b64_tr_b16, the hardware transpose read (two per fragment).
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
#define GB_HEAD_SPLIT 4             // reduction ranges of the head kernel (HeadArgsB::ksplit)
// WM = waves along i (2 or 4): tile rows BM = 64 WM, 2 WM waves (WM x 2), each a 64 x 64 accumulator block (4 x 4 MFMA tiles)
#define GB_BM(WM) (64 * (WM))
#define GB_THREADS(WM) (128 * (WM))
#define GB_STAGE_BYTES(WM) ((GB_BM(WM) + GB_N) * GB_K * 2)         // A tile then B tile: 48 KB (WM 4) / 32 KB (WM 2)
#define GB_LDS_BYTES(WM) (GB_STAGES * GB_STAGE_BYTES(WM))          // 144 KB / 96 KB: one workgroup per CU
#define GB_PIECES(WM) ((GB_BM(WM) + GB_N) / 8 / (2 * (WM)))        // 1-KB LDS-DMA pieces per wave and stage: 6 / 8

enum { GEPI_TANH = 0, GEPI_TANHGRAD = 1 };

struct GemmArgs {
    const bf16_t* A[2]; const bf16_t* B[2];   // per tower; row-major: A [I][K]; B [J][K] (TANHGRAD) or [K][J] (TANH: the weights as they lie)
    int lda, ldb;
    int K;                                    // reduction length
    int tiles_i;                              // tiles along i (block id -> (ti, tj) = (id % tiles_i, id / tiles_i))
    const float* bias[2];                     // [J] fp32 (TANH) or null
    const bf16_t* H[2]; int ldh;              // TANHGRAD: tanh outputs of this layer, [I][J]
    bf16_t* C[2]; int ldc;                    // out [I][J] bf16
    float* bsum[2]; int bsum_ld;              // TANHGRAD: per row-tile column sums of the fp32 outputs, [tiles_i][bsum_ld] (bias gradients; may be null)
#ifdef PPO_STAMPS
    unsigned long long* stamps;               // diagnostic builds only: [EPI][tower][block][8] cycle stamps of wave 0
#endif
};
#ifdef PPO_STAMPS
#define GSTAMP(i) do { if (gst && threadIdx.x == 0) gst[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

// 16-byte chunk c (0..7) of row r of a [rows][64 bf16] LDS tile lives at chunk c ^ (r & 7): a 16-lane group of a
// ds_read_b128 (16 different rows, same k range) then spreads over 8 slots of the 256-byte bank row instead of 2.
// LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address and again on the read.
__device__ __forceinline__ int gb_swz(int row, int chunk) { return chunk ^ (row & 7); }

// ---- operands whose REDUCTION index is the row ([64 rows][128 features] LDS images, transposed reads) -----------------------------
// 256-byte rows; the 16-byte chunk ch of row m sits at slot ch ^ (((m & 3) << 2) | ((m >> 2) & 3)) (the guide's image (b): the
// transposed reads of a 16x16x32 operand -- per 32-lane half two 4-row blocks 8 rows apart in the same 16 columns -- are conflict-free)
__device__ __forceinline__ int gt_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// operand fragment of k-step ks (rows m = 32 ks .. + 31 of the stage) for the 16 features starting at f0 of a sub-tile: lane
// (r = lane & 15, g = lane >> 4) must hold rows 32 ks + 8 g .. + 7 of feature f0 + r.  Two transposed reads: the 16 lanes of group g
// address the block of rows 32 ks + 8 g + 4 half .. + 3 x columns f0 .. f0 + 15 (lane 4q + p: row q, columns 4p .. 4p + 3) and
// lane r receives column r, row q in element q.
__device__ __forceinline__ bf16x8 gt_frag(const char* sub, int f0, int ks) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lp;
    s16x4 v[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = 32 * ks + 8 * g + 4 * half + q;
        v[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sub + row * 256 + 16 * (((f0 >> 3) + (p >> 1)) ^ gt_swz(row)) + 8 * (p & 1)));
    }
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 w = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
}

// One stage = the [BM][64] A tile followed by the [128][64] B tile, (BM + 128) / 8 pieces of 1 KB (8 rows x 128 B), dealt
// to the waves round robin; lane l of a piece lands on (row 8p + l/8, chunk l%8) and fetches the chunk that belongs there.
// BT: the B operand is stored [K][J] (reduction index = row, e.g. a weight matrix as it lies): its 16 pieces are 4 k-rows x 256 B
// of a [64][128] image read back with transposed reads (gt_frag) instead of [128][64] rows read with ds_read_b128.
template <int WM, bool BT = false>
__device__ __forceinline__ void gb_stage_piece(const bf16_t* __restrict__ A, int lda, int i0, const bf16_t* __restrict__ B, int ldb, int j0, int k0, char* st, const int q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PA = GB_BM(WM) / 8;
    const int piece = wave + 2 * WM * q;                     // wave-uniform
    const bool isA = piece < PA;
    const int r = (isA ? piece : piece - PA) * 8 + (lane >> 3), c = lane & 7;
    const int rt = 4 * (piece - PA) + (lane >> 4), cht = (lane & 15) ^ gt_swz(rt);
    const bf16_t* src = isA ? A + (size_t)(i0 + r) * lda + k0 + gb_swz(r, c) * 8
                      : BT  ? B + (size_t)(k0 + rt) * ldb + j0 + 8 * cht
                            : B + (size_t)(j0 + r) * ldb + k0 + gb_swz(r, c) * 8;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(st + piece * 1024), 16, 0, 0);
}
template <int WM, bool BT = false>
__device__ __forceinline__ void gb_stage(const bf16_t* __restrict__ A, int lda, int i0, const bf16_t* __restrict__ B, int ldb, int j0, int k0, char* st) {
#pragma unroll
    for (int q = 0; q < GB_PIECES(WM); ++q) gb_stage_piece<WM, BT>(A, lda, i0, B, ldb, j0, k0, st, q);
}

// fragment of MFMA k-step ks (0/1) for the 16 rows starting at r0: lane (r = lane & 15, g = lane >> 4) holds
// k = 32 ks + 8 g .. + 7 of row r0 + r
__device__ __forceinline__ bf16x8 gb_frag(const char* lds_tile, int r0, int ks) {
    const int lane = threadIdx.x & 63;
    const int r = r0 + (lane & 15), c = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + gb_swz(r, c) * 16);
}

struct GemmAcc { f32x4 v[4][4]; };           // [mi][ni]: row 16 mi + (lane & 15), columns 16 ni + 4 g + reg of the wave's 64 x 64 block (the B fragment
                                             // goes in as the matrix instruction's FIRST operand: a lane then holds 4 CONSECUTIVE columns of one output row)

template <int N> __device__ __forceinline__ void gb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The epilogue's LDS image of a [BM][128] bf16 tile: 256-byte rows, the 16-byte chunk P of row i at slot P ^ (i & 15).  It is
// where the TanhGrad epilogue finds the layer's tanh outputs (brought in by LDS-DMA during the LAST TWO k stages, into ring buffers
// the loop no longer needs: the loop starts on the buffer that makes buffer 2 its last one, so the image always lies in buffers
// 0 and 1) and where every epilogue parks its result, IN PLACE (a lane overwrites exactly the 8 bytes it read), before the tile
// is streamed out in whole rows.  A lane's 8-byte element (row i, columns 4q .. 4q+3) is half q & 1 of chunk q >> 1: the 16 rows
// of a 32-lane half land on 16 different 16-byte slots (no conflict on the 64-bank reads, 2-way on the 32-bank writes).
#define GB_IMG_ROWS1(WM) ((GB_STAGE_BYTES(WM) / 256) < GB_BM(WM) ? (GB_STAGE_BYTES(WM) / 256) : GB_BM(WM))     // rows of the image inside ring buffer 0
#define GB_IMG_P1(WM) (GB_IMG_ROWS1(WM) / 4 / (2 * (WM)))                                                  // 1-KB pieces per wave: part 1 / part 2
#define GB_IMG_P2(WM) ((GB_BM(WM) - GB_IMG_ROWS1(WM)) / 4 / (2 * (WM)))
__device__ __forceinline__ int gb_img_off(int i, int q) { return i * 256 + 16 * ((q >> 1) ^ (i & 15)) + 8 * (q & 1); }

template <int WM, int NPW>
__device__ __forceinline__ void gb_stage_image(const bf16_t* __restrict__ H, int ldh, int i0, int j0, char* img, int piece0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
        const int piece = piece0 + wave + 2 * WM * q;        // 4 rows of 256 B
        const int row = 4 * piece + (lane >> 4), P = (lane & 15) ^ (row & 15);
        typedef const __attribute__((address_space(1))) void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        __builtin_amdgcn_global_load_lds((gptr)(H + (size_t)(i0 + row) * ldh + j0 + 8 * P), (lptr)(img + piece * 1024), 16, 0, 0);
    }
}

// Main loop.  Per step ONE barrier: [wait until this wave's pieces of stage t have landed: all but the youngest stage's
// loads] -> barrier (every wave's pieces of stage t are in LDS; every wave is done multiplying stage t-1) -> issue stage
// t+2 into the buffer stage t-1 used -> read fragments of stage t, 32 MFMAs per wave.  A __syncthreads() would drain the
// LDS-DMA queue (it waits vmcnt(0)), so the barrier is the raw instruction and the wait is counted.
// The k loop is bound by the bytes a CU can pull from L2 into LDS (measured: 48 KB per stage in ~1975 cycles = 25 B/clk per CU,
// against 1024 cycles of matrix instructions per SIMD), not by the matrix pipe.  (Round 5: a software-pipelined form -- fragment reads of the next
// k-step in flight under the matrix instructions of the current one, the stage's barrier in the middle of its matrix work -- was correct and
// bought 1 %: 23.6 vs 24.6 us per K = 1024 launch on boxes 3 % apart; 254 VGPRs, spills in the chained kernel.  Not kept: profiles/r05_e_*.)
template <int WM, bool IMG, bool BT = false>
__device__ __forceinline__ void gb_mainloop(GemmAcc& acc, const bf16_t* __restrict__ A, int lda, int i0, const bf16_t* __restrict__ B, int ldb, int j0,
                                            int kbeg, int K, char* lds, const bf16_t* __restrict__ H = nullptr, int ldh = 0, unsigned long long* gst = nullptr) {
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    constexpr int SB = GB_STAGE_BYTES(WM), NP = GB_PIECES(WM);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc.v[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = K / GB_K;
    int cur = (3 - nt % 3) % 3;                              // the last stage then sits in buffer 2: buffers 0 and 1 are free for the image
    {
        int nx = cur + 1; if (nx >= GB_STAGES) nx -= GB_STAGES;
        gb_stage<WM, BT>(A, lda, i0, B, ldb, j0, kbeg, lds + cur * SB);
        if (nt > 1) gb_stage<WM, BT>(A, lda, i0, B, ldb, j0, kbeg + GB_K, lds + nx * SB);
    }
    GSTAMP(1);
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gb_wait_vm<NP>();
        else if (IMG && nt > 1) gb_wait_vm<GB_IMG_P1(WM)>();   // the image's first part was requested after the last stage
        else gb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int nb = cur + 2; if (nb >= GB_STAGES) nb -= GB_STAGES;
        const bool more = t + 2 < nt;
#ifndef GB_NOLOAD
        // (round 6: the same requests issued from inside the matrix block, one piece behind every four matrix instructions, measured 1 %: profiles/r06_d_*; not kept)
        if (more) gb_stage<WM, BT>(A, lda, i0, B, ldb, j0, kbeg + (t + 2) * GB_K, lds + nb * SB);
#endif
        if constexpr (IMG) {
            if (t == (nt > 1 ? nt - 2 : 0)) gb_stage_image<WM, GB_IMG_P1(WM)>(H, ldh, i0, j0, lds, 0);                                   // buffer 0 is free
            if constexpr (GB_IMG_P2(WM) > 0) { if (t == nt - 1) gb_stage_image<WM, GB_IMG_P2(WM)>(H, ldh, i0, j0, lds, GB_IMG_ROWS1(WM) / 4); }   // buffer 1 too
        }
        const char* at = lds + cur * SB;
        const char* bt = at + GB_BM(WM) * GB_K * 2;
        // all 16 fragment reads of the stage go out first (two register sets: 64 accumulator + 64 fragment VGPRs per lane)
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a) af[ks][a] = gb_frag(at, wm + 16 * a, ks);
#pragma unroll
            for (int b = 0; b < 4; ++b) bfr[ks][b] = BT ? gt_frag(bt, wn + 16 * b, ks) : gb_frag(bt, wn + 16 * b, ks);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
#ifndef GB_NOMFMA
                for (int b = 0; b < 4; ++b) acc.v[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][b], af[ks][a], acc.v[a][b], 0, 0, 0);
#else
                for (int b = 0; b < 4; ++b) asm volatile("" :: "v"(af[ks][a]), "v"(bfr[ks][b]));
#endif
            }
        __builtin_amdgcn_s_setprio(0);
        if (++cur == GB_STAGES) cur = 0;
    }
    GSTAMP(2);
    __syncthreads();                                         // the image has landed (vmcnt(0)); the epilogue may write buffers 0 and 1
    GSTAMP(3);
}

// tanh for the bf16 path: 1 - 2 / (e^{2x} + 1) on the hardware exp2 / rcp units, no small-|x| branch (the absolute error of the
// cancellation, ~1e-7, is far below the 2^-9 of the bf16 value it is rounded to); e^{2x} = inf gives 1, 0 gives -1.
__device__ __forceinline__ float bf16_tanh_scaled(float x_times_2log2e) {
    const float e = __builtin_amdgcn_exp2f(x_times_2log2e);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// Epilogue: bias + tanh, or TanhGrad against the image, into the image; bias-gradient column sums; then the tile goes out in whole
// 256-byte rows (16-byte stores).  A lane holds 4 consecutive COLUMNS of one output row per accumulator tile.
template <int WM, int EPI>
__device__ __forceinline__ void gb_epilogue_bf16(GemmAcc& acc, const GemmArgs& a, int tw, int i0, int j0, char* lds, const float4 (&bias4)[4], unsigned long long* gst = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int g = lane >> 4, c = lane & 15;
    constexpr int BM = GB_BM(WM), NT = GB_THREADS(WM);
    constexpr float K2 = 2.8853900817779268f;               // 2 log2(e)
    float csum[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[nb][r] = 0.f;
    bf16x4 h4[4][4];
#ifndef GB_NOTANH
    if constexpr (EPI == GEPI_TANHGRAD) {                     // every LDS read in flight before the first use
#pragma unroll
        for (int ma = 0; ma < 4; ++ma)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) h4[ma][nb] = *reinterpret_cast<const bf16x4*>(lds + gb_img_off(wm + 16 * ma + c, (wn >> 2) + 4 * nb + g));
    }
#endif
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const float bb[4] = {bias4[nb].x * K2, bias4[nb].y * K2, bias4[nb].z * K2, bias4[nb].w * K2};
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) {
            float y[4];
#ifdef GB_NOTANH
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc.v[ma][nb][r] + bb[r];
#else
            if constexpr (EPI == GEPI_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = bf16_tanh_scaled(fmaf(acc.v[ma][nb][r], K2, bb[r]));
            } else {                                          // TanhGrad: dY = dX .* (1 - h^2)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float h = (float)h4[ma][nb][r]; y[r] = acc.v[ma][nb][r] * (1.0f - h * h); csum[nb][r] += y[r]; }
            }
#endif
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)y[r];
            *reinterpret_cast<bf16x4*>(lds + gb_img_off(wm + 16 * ma + c, (wn >> 2) + 4 * nb + g)) = o;
        }
    }
    GSTAMP(4);
    // bias gradients ride along: db[j] = sum over rows of dY[:, j].  A lane's 4 row blocks are added in registers, the 16 lanes of
    // a group (16 different rows) in a fixed-shape tree, the WM waves that share the columns meet in LDS behind the image,
    // and the tile's sums go to row `ti` of a small [row tiles][features] table the gradient assembly adds up in tile order.
    float* cs = reinterpret_cast<float*>(lds + BM * 256);                  // [WM][128]
    if constexpr (EPI == GEPI_TANHGRAD) {
        if (a.bsum[tw]) {
            // 16 sums over the 16 lanes of a group in 15 exchanges: each step hands half of the values to the partner lane and adds the
            // half it keeps, so that lane c ends with the complete sum of value c (= column 16 (c >> 2) + 4 g + (c & 3) of the wave's block)
            float v8[8], v4[4], v2[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float lo = csum[k >> 2][k & 3], hi = csum[2 + (k >> 2)][k & 3];
                const float got = __shfl_xor((c & 8) ? lo : hi, 8);
                v8[k] = ((c & 8) ? hi : lo) + got;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float got = __shfl_xor((c & 4) ? v8[k] : v8[k + 4], 4); v4[k] = ((c & 4) ? v8[k + 4] : v8[k]) + got; }
#pragma unroll
            for (int k = 0; k < 2; ++k) { const float got = __shfl_xor((c & 2) ? v4[k] : v4[k + 2], 2); v2[k] = ((c & 2) ? v4[k + 2] : v4[k]) + got; }
            const float got = __shfl_xor((c & 1) ? v2[0] : v2[1], 1);
            cs[(wave >> 1) * GB_N + wn + 16 * (c >> 2) + 4 * g + (c & 3)] = ((c & 1) ? v2[1] : v2[0]) + got;
        }
    }
    __syncthreads();
    GSTAMP(5);
    if constexpr (EPI == GEPI_TANHGRAD) {
        if (a.bsum[tw] && tid < GB_N) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < WM; ++q) v += cs[q * GB_N + tid];
            a.bsum[tw][(size_t)(i0 / BM) * a.bsum_ld + j0 + tid] = v;
        }
    }
    if (a.C[tw]) {
#pragma unroll
        for (int q = 0; q < BM * 16 / NT; ++q) {
            const int id = tid + NT * q;
            const int i = id >> 4, p = id & 15;
            const uint4 o = *reinterpret_cast<const uint4*>(lds + i * 256 + 16 * (p ^ (i & 15)));
#ifdef GB_NOSTORE
            if (o.x == 0x12345678u)
#endif
            *reinterpret_cast<uint4*>(a.C[tw] + (size_t)(i0 + i) * a.ldc + j0 + 8 * p) = o;
        }
    }
    GSTAMP(6);
#ifdef PPO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP(7);
#endif
}

// one output tile of C = epilogue(A B): tower tw, rows i0 .., columns j0 ..
template <int WM, int EPI>
__device__ __forceinline__ void gemm_nt_tile(const GemmArgs& a, int tw, int i0, int j0, char* gb_lds, unsigned long long* gst) {
    GemmAcc acc;
    GSTAMP(0);
    float4 bias4[4];                                         // this lane's 16 bias values: requested before the loop, used after it
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bias4[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI == GEPI_TANH) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bias4[nb] = *reinterpret_cast<const float4*>(a.bias[tw] + j0 + (wave & 1) * 64 + 16 * nb + 4 * (lane >> 4));
    }
    if constexpr (EPI == GEPI_TANHGRAD) gb_mainloop<WM, true>(acc, a.A[tw], a.lda, i0, a.B[tw], a.ldb, j0, 0, a.K, gb_lds, a.H[tw], a.ldh, gst);
    else gb_mainloop<WM, false, true>(acc, a.A[tw], a.lda, i0, a.B[tw], a.ldb, j0, 0, a.K, gb_lds, nullptr, 0, gst);
    {
#ifndef GB_NOEPI
        gb_epilogue_bf16<WM, EPI>(acc, a, tw, i0, j0, gb_lds, bias4, gst);
#else
        if (acc.v[0][0][0] == 123.456f) a.C[tw][0] = (bf16_t)acc.v[1][1][1];
#endif
    }
}

template <int WM, int EPI>
__global__ __launch_bounds__(GB_THREADS(WM)) void gemm_nt_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char gb_lds[];
    const int tw = blockIdx.y;
    const int ti = blockIdx.x % a.tiles_i, tj = blockIdx.x / a.tiles_i;
#ifdef PPO_STAMPS
    unsigned long long* gst = a.stamps ? a.stamps + ((size_t)(EPI * 2 + tw) * 256 + blockIdx.x) * 8 : nullptr;
#else
    unsigned long long* gst = nullptr;
#endif
    gemm_nt_tile<WM, EPI>(a, tw, ti * GB_BM(WM), tj * GB_N, gb_lds, gst);
}

// ---- a CHAIN of layers in one launch ---------------------------------------------------------------------------------------------------
// The hidden layers of a pass depend on each other only inside a ROW tile: tile (ti, tj) of layer l + 1 needs rows ti of ALL column tiles of layer l,
// i.e. the outputs of the tiles_j workgroups that share (tower, ti).  One launch per layer makes that a chip-wide barrier plus a launch: of the
// 24.6 us of a [4096 x 1024 x 1024] x 2 launch only ~18 us are a workgroup's life (tools/step_sequence.py against the in-kernel stamps), and the
// 8 GEMM launches of a train step carry ~50 us of that.  Here the layers of a pass run in ONE launch of tiles_i x tiles_j x 2 workgroups (one per
// CU: 144 KB of LDS each, all resident -- the host only uses this form when they fit the device); after its epilogue a workgroup raises its word of
// a table, and before the next layer it waits for the tiles_j words of its row group only.
// Visibility WITHOUT cache-wide fences: the tiles_j workgroups of a row group are dealt to ONE XCD (workgroup b runs on XCD b % 8), so the layer's
// output tile a workgroup wrote with plain stores (complete in that XCD's L2 once the stores are acknowledged: s_waitcnt vmcnt(0)) is what the
// group's other workgroups read through the same L2; a CU's L1 holds no line of it (it has not been read in this launch).  That DOES rest on the
// dealing, so it is CHECKED, not assumed: every word carries the hardware's XCC id of its writer (s_getreg HW_REG_XCC_ID) and a reader that finds a
// producer on another XCD raises the error word -- the host call that synchronises next returns an error and the handle goes back to one launch per
// layer.  Waits are bounded the same way (a workgroup that never became resident).
#define GB_CHAIN_MAX 4
struct ChainArgs {
    int n;                                    // links: link l + 1 reads link l's C as its A
    int tiles_i, tiles_j;                     // of every link (same output shape)
    unsigned* words;                          // [GB_CHAIN_MAX][2 * tiles_i groups][16]: (epoch << 4) | xcc per workgroup.  ONE table per number of row groups (the host keeps
                                              // GB_CHAIN_SHAPES of them): a workgroup's epoch is its own word + 1, and in a table shared between shapes -- the act path's rows and
                                              // the minibatch's -- a slot means different (link, group) pairs, so a stale word could satisfy a wait
    unsigned* err;                            // raised on a time-out (1) or a row group spread over two XCDs (2)
    GemmArgs link[GB_CHAIN_MAX];
};
#define GB_CHAIN_WORDS (GB_CHAIN_MAX * 64 * 16)           // up to 64 row groups
#define GB_CHAIN_SHAPES 8                                 // row-group counts 8, 16, .. 64

template <int EPI>
__global__ __launch_bounds__(GB_THREADS(4)) void gemm_chain_bf16_kernel(ChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) char gb_lds[];
    const unsigned b = blockIdx.x, x = b & 7u, q = b >> 3;
    const int G = 2 * ca.tiles_i, gpx = G >> 3;                          // row groups; per XCD
    const int gidx = (int)x * gpx + (int)q / ca.tiles_j, tj = (int)q % ca.tiles_j;
    const int tw = gidx / ca.tiles_i, ti = gidx % ca.tiles_i;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    unsigned* mine = ca.words + ((size_t)gidx * 16 + tj);               // + l * G * 16 for link l
    const unsigned epoch = (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 4) + 1u;      // (only this workgroup writes its words)
    for (int l = 0; l < ca.n; ++l) {
        if (l > 0) {
            // (requesting the next layer's weights BEFORE this wait was measured: no gain -- the group's workgroups finish together, there is no
            // idle time to fill; profiles/r05_e_*)
            if (threadIdx.x < (unsigned)ca.tiles_j) {
                const unsigned* w = ca.words + ((size_t)(l - 1) * G + gidx) * 16 + threadIdx.x;
                unsigned polls = 0, v;
                while ((int)(((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) - epoch) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > (1u << 21)) { __hip_atomic_store(ca.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if ((v & 15u) != xcc) __hip_atomic_store(ca.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        gemm_nt_tile<4, EPI>(ca.link[l], tw, ti * GB_BM(4), tj * GB_N, gb_lds, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's stores of the output tile are acknowledged by the L2
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(mine + (size_t)l * G * 16, (epoch << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- weight gradients: every matrix of both towers in one grouped launch, split over the minibatch rows -------------
// dW[i][j] = sum over rows m of X[m][i] * dY[m][j]: the "TN" form.  A stage is 64 rows m of the X tile (BM features, as BM / 128
// sub-tiles) and of the dY tile (128 features): [64][128] sub-tiles with 256-byte rows, brought in by LDS-DMA as they lie.  The
// 16-byte chunk ch of row m sits at slot ch ^ (((m & 3) << 2) | ((m >> 2) & 3)) (the guide's image (b): the transposed reads of
// a 16x16x32 operand -- per 32-lane half two 4-row blocks 8 rows apart in the same 16 columns -- are conflict-free on it).
struct DwTileB { const bf16_t* A; const bf16_t* B; int lda, ldb; int i0, j0; int out_off, ldo; int is_x0; };
// The launch is WORK-balanced, not tile-balanced: the (tile, 64-row stage) pairs of all tiles form one sequence (tile-major) and
// workgroup w takes `per` consecutive stages of it, i.e. the tail of one tile's reduction and/or the head of the next one's
// (per <= nst: at most two segments).  At configs[4] that is 152 tiles x 64 stages over 256 workgroups of 38 stages: ONE round on
// the 256 CUs, where 4 row splits per tile were 608 workgroups = three rounds, the last one 37 % full.  A tile's partial sums go
// to slabs 0 .. (number of workgroups that touch it) - 1; the gradient assembly recomputes that count from (nst, per).
struct DwArgsB { const DwTileB* tiles; int nst, per, total; float* slabs; size_t slab_stride;
                 const bf16_t* x0; int x0_ld          // epoch-staged observations of this minibatch (null: the tiles' own x0)
#ifdef PPO_STAMPS
                 ; unsigned long long* stamps
#endif
                 ; };

template <int WM>
__device__ __forceinline__ void gt_stage(const bf16_t* __restrict__ X, int ldx, int i0, const bf16_t* __restrict__ Y, int ldy, int j0, int m0, char* st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NA = WM / 2;                               // sub-tiles of X
#pragma unroll
    for (int q = 0; q < GB_PIECES(WM); ++q) {
        const int piece = wave + 2 * WM * q;                 // wave-uniform; 16 pieces (4 rows x 256 B each) per sub-tile
        const int sub = piece >> 4;
        const int row = 4 * (piece & 15) + (lane >> 4), ch = (lane & 15) ^ gt_swz(row);
        const bf16_t* src = sub < NA ? X + (size_t)(m0 + row) * ldx + i0 + 128 * sub + 8 * ch : Y + (size_t)(m0 + row) * ldy + j0 + 8 * ch;
        typedef const __attribute__((address_space(1))) void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(st + piece * 1024), 16, 0, 0);
    }
}

template <int WM>
__device__ __forceinline__ void gt_mainloop(GemmAcc& acc, const bf16_t* __restrict__ X, int ldx, int i0, const bf16_t* __restrict__ Y, int ldy, int j0,
                                            int mbeg, int M, char* lds) {
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    constexpr int SB = GB_STAGE_BYTES(WM), NP = GB_PIECES(WM);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc.v[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = M / GB_K;
    gt_stage<WM>(X, ldx, i0, Y, ldy, j0, mbeg, lds);
    if (nt > 1) gt_stage<WM>(X, ldx, i0, Y, ldy, j0, mbeg + GB_K, lds + SB);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gb_wait_vm<NP>(); else gb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifndef GT_NOLOAD
        if (t + 2 < nt) {
            int nb = cur + 2; if (nb >= GB_STAGES) nb -= GB_STAGES;
            gt_stage<WM>(X, ldx, i0, Y, ldy, j0, mbeg + (t + 2) * GB_K, lds + nb * SB);
        }
#endif
        const char* at = lds + cur * SB + (wm >> 7) * 16384;             // the X sub-tile this wave's 64 features lie in
        const char* bt = lds + cur * SB + (WM / 2) * 16384;
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a) af[ks][a] = gt_frag(at, (wm & 127) + 16 * a, ks);
#pragma unroll
            for (int b = 0; b < 4; ++b) bfr[ks][b] = gt_frag(bt, wn + 16 * b, ks);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc.v[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][b], af[ks][a], acc.v[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (++cur == GB_STAGES) cur = 0;
    }
}

template <int WM>
__global__ __launch_bounds__(GB_THREADS(WM)) void gemm_dw_bf16_kernel(DwArgsB a) {
    extern __shared__ __attribute__((aligned(16))) char gb_lds[];
    // consecutive work ranges on ONE XCD (they share the X panel of their tile row): workgroup b runs on XCD b % 8
    const int G = gridDim.x;
#ifndef GT_NOREMAP
    const int w = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
#else
    const int w = blockIdx.x;
#endif
    const int begin = w * a.per, end = min(begin + a.per, a.total);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, g = lane >> 4, c = lane & 15;
    int s0 = begin;
#ifdef PPO_STAMPS
    unsigned long long* gst = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
    int sti = 1;
#endif
    GSTAMP(0);
    while (s0 < end) {
        const int tile = s0 / a.nst;
        const int s1 = min(end, (tile + 1) * a.nst);
#ifdef GT_SAME_PANEL
        DwTileB t = a.tiles[tile];                          // timing diagnostic only: every tile reads tile 0's operand panels (3 MB: resident in every L2); results are wrong
        { const DwTileB t0 = a.tiles[GT_SAME_PANEL]; t.A = t0.A; t.B = t0.B; t.lda = t0.lda; t.ldb = t0.ldb; t.i0 = t0.i0; t.j0 = t0.j0; t.is_x0 = t0.is_x0; }
#else
        const DwTileB t = a.tiles[tile];
#endif
        const int slab = w - (tile * a.nst) / a.per;           // this workgroup's rank among the tile's contributors
        const bool ov = t.is_x0 && a.x0;
        GemmAcc acc;
        __syncthreads();                                     // the ring is free again (second segment)
        gt_mainloop<WM>(acc, ov ? a.x0 : t.A, ov ? a.x0_ld : t.lda, t.i0, t.B, t.ldb, t.j0, (s0 - tile * a.nst) * GB_K, (s1 - s0) * GB_K, gb_lds);
        float* out = a.slabs + (size_t)slab * a.slab_stride + t.out_off;
#ifdef PPO_STAMPS
        GSTAMP(sti); if (gst && threadIdx.x == 0) gst[5 + (sti >> 1)] = (unsigned long long)(s1 - s0);
        ++sti;
#endif
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) {
            float* row = out + (size_t)(t.i0 + wm + 16 * ma + c) * t.ldo + t.j0 + wn + 4 * g;
#pragma unroll
            // (plain stores: the slabs are read back by the assembly launch out of the Infinity Cache; nontemporal stores make this launch 1.4 us faster and that one 4 us slower)
            for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<float4*>(row + 16 * nb) = make_float4(acc.v[ma][nb][0], acc.v[ma][nb][1], acc.v[ma][nb][2], acc.v[ma][nb][3]);
        }
#ifdef PPO_STAMPS
        GSTAMP(sti); ++sti;
#endif
        s0 = s1;
    }
}

// ---- input staging: fp32 observations -> bf16 [rows_pad][Kp0]; the act path normalises here (env_normalize.hpp:99-104) and writes the normalised fp32 rows into
// the rollout buffer, as stage_block_inputs does for the fused kernels --------------------------------------------------
struct StageArgsB { const float* obs; int n, O, Kp0, rows_pad; ObsNorm nz; float* obs_out; bf16_t* X; };

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
}

// The same staging with 16-byte column strips (O and Kp0 multiples of 4): a thread keeps ONE quad of columns -- its four normalisation scales are
// computed once (correctly rounded sqrt and division, the element-wise kernel's expression) -- and walks rows in steps of `rstride`; the grid is
// sized so that (gridDim.x * 256) is a multiple of Kp0 / 4, i.e. every thread's column quad is fixed.  Per element the element-wise form paid a
// 64-bit division, a square root and a division: 30 us for 8192 x 256 observations (profiles/r04_z_kernel_stats_cfg5.csv), 3 % of HBM rate.
__global__ __launch_bounds__(256) void bf16_stage4_kernel(StageArgsB a) {
    const int Q = a.Kp0 >> 2;                                      // column quads per row
    const unsigned q0 = blockIdx.x * 256u + threadIdx.x;
    const int j = 4 * (int)(q0 % (unsigned)Q);
    int row = (int)(q0 / (unsigned)Q);
    const int rstride = (int)((gridDim.x * 256u) / (unsigned)Q);
    const bool live_col = j < a.O;                                 // (O a multiple of 4: a quad is all observation or all padding)
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f};
    if (live_col && a.nz.enabled) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mu[k] = a.nz.mean[j + k]; sc[k] = 1.0f / sqrtf(a.nz.var[j + k] + a.nz.eps); }
    }
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    constexpr int RU = 4;                                          // rows in flight per thread
    for (; row < a.rows_pad; row += RU * rstride) {
        float4 v[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int r = row + u * rstride;
            v[u] = (live_col && r < a.n) ? *reinterpret_cast<const float4*>(a.obs + (size_t)r * a.O + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int r = row + u * rstride;
            if (r >= a.rows_pad) break;
            float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (live_col && r < a.n) {
                if (a.nz.enabled) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { x[k] = (x[k] - mu[k]) * sc[k]; x[k] = tf_min(tf_max(x[k], -a.nz.clip), a.nz.clip); }
                }
                if (a.obs_out) *reinterpret_cast<float4*>(a.obs_out + (size_t)r * a.O + j) = make_float4(x[0], x[1], x[2], x[3]);
            }
            bf16x4_t o; o[0] = (bf16_t)x[0]; o[1] = (bf16_t)x[1]; o[2] = (bf16_t)x[2]; o[3] = (bf16_t)x[3];
            *reinterpret_cast<bf16x4_t*>(a.X + (size_t)r * a.Kp0 + j) = o;
        }
    }
}

// grid of bf16_stage4_kernel: about `want` workgroups, rounded up to a multiple of (Kp0 / 4) / gcd(256, Kp0 / 4)
inline unsigned bf16_stage4_grid(int Kp0, size_t quads, unsigned want = 512) {
    const unsigned Q = (unsigned)Kp0 / 4;
    unsigned g = 256, q = Q;
    while (q) { const unsigned t = g % q; g = q; q = t; }          // g = gcd(256, Q)
    const unsigned unit = Q / g;
    unsigned grid = (unsigned)std::min<size_t>(want, (quads + 255) / 256);
    grid = std::max(1u, (grid + unit - 1) / unit * unit);
    return grid;
}

// ---- the two heads (G:5766-5893 act, G:9188-9427 train): mu = H_L W_mu + b_mu (A columns), v = H_L^v w_v + b_v (one column) -------------
// A launch of their own since round 6.  Until then they were tiles of the big GEMM: 32 tiles of 256 x 128 with the reduction cut in four to
// cover the chip, i.e. 128 workgroups that pulled 48 KB per 64-deep stage for a product of which half the policy columns and 127 of 128 value
// columns are padding, and wrote 16.8 MB of fp32 partial products for the consumers to read back: 12.8 us of a 229 us train step.
// Here a workgroup owns 64 ROWS of one tower and one of `ksplit` reduction ranges (the cut stays: it is what covers the chip, two workgroups
// per CU); a wave owns 16 of the rows.  A row of H is read once, 16-byte loads straight into the matrix instruction's operand map (the
// reduction index is contiguous).  W lies [K][Ap], reduction index = row: the range's rows come in by LDS-DMA as ONE [k][128] image (64 KB at
// K = 1024, gt_swz on the source side, only the 16-byte chunks of columns that exist when TRIM) and out through the transposed reads of
// gt_frag, for the 16-column blocks that exist (A / 16 for the policy, one for the value).  One barrier, no ring.  The fp32 partial products
// of range ks go to F + ks * f_split (bias in range 0) and the consumers add them in range order, as before: a row's result does not depend
// on the workgroup or slot that computed it (act and train model: same bits on the same weights).
// (First form, measured and replaced: 32 rows and the WHOLE reduction per workgroup, wave w on the w-th quarter of K behind a wave-private ring
// of four 32-row images: 10.6 us -- every workgroup pulls all of W, 256 KB, at the ~20 B/clk a CU gets from L2 into LDS; profiles/r06_j_*.)
struct HeadArgsB {
    const bf16_t* H[2]; int ldh;              // last hidden layer's outputs, [rows_pad][K]
    const bf16_t* W[2]; int ldw;              // head weights as they lie, [K][Ap], Ap == 128
    const float* bias[2];                     // [Ap] fp32
    float* F[2]; int ldf; size_t f_split;     // out [ksplit][rows_pad][Ap] fp32: the columns of the tower's 16-column blocks are written
    int K, ksplit;                            // K / ksplit a multiple of 32
#ifdef PPO_STAMPS
    unsigned long long* stamps;               // diagnostic builds only: [tower][block][8] cycle stamps of wave 0
#endif
};
#define BH_ROWS 64
#define BH_KPASS 256                          // rows of W in LDS at a time
#define BH_LDS_BYTES (BH_KPASS * 256)         // 64 KB: two workgroups per CU

// one pass: the image of rows k .. k + 32 nst of W, this wave's H fragments, nst k-steps.  FULL: nst == 8 is known at compile time -- the H loads are
// unconditional, requested BEHIND the image (which comes from L2 and is needed by everybody) and waited for one k-step at a time by the compiler's own counts
template <bool TRIM, bool FULL, int NCB>
__device__ __forceinline__ void bh_pass(f32x4 (&acc)[NCB], const bf16_t* __restrict__ hrow, const bf16_t* __restrict__ W, int ldw, int k, int nst, char* img,
                                        unsigned long long* gst) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
    for (int q = 0; q < BH_KPASS / 16; ++q) {                // pieces of 4 rows x 256 B, dealt to the waves round robin
        const int p = wave + 4 * q;
        if (FULL || p < 8 * nst) {
            const int rt = 4 * p + (lane >> 4), cht = (lane & 15) ^ gt_swz(rt);
            if (!TRIM || cht < 2 * NCB) __builtin_amdgcn_global_load_lds((gptr)(W + (size_t)(k + rt) * ldw + 8 * cht), (lptr)(img + p * 1024), 16, 0, 0);
        }
    }
    bf16x8 hf[BH_KPASS / 32];
#pragma unroll
    for (int s = 0; s < BH_KPASS / 32; ++s) {
        if constexpr (FULL) hf[s] = *reinterpret_cast<const bf16x8*>(hrow + k + 32 * s);
        else hf[s] = s < nst ? *reinterpret_cast<const bf16x8*>(hrow + k + 32 * s) : bf16x8{};
    }
    GSTAMP(1);
    if constexpr (FULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BH_KPASS / 32) : "memory");     // this wave's pieces of the image have landed (the H loads may be out)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP(2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    GSTAMP(3);
    // k-step s + 1's fragments are read under k-step s's matrix instructions (two sets); the scheduling barriers keep the compiler from hoisting all 64 reads
    bf16x8 wf[2][NCB];
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) wf[0][nb] = gt_frag(img, 16 * nb, 0);
#pragma unroll
    for (int s = 0; s < BH_KPASS / 32; ++s) {
        if (FULL || s < nst) {
            if (s + 1 < BH_KPASS / 32 && (FULL || s + 1 < nst)) {
#pragma unroll
                for (int nb = 0; nb < NCB; ++nb) wf[(s + 1) & 1][nb] = gt_frag(img, 16 * nb, s + 1);
            }
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s & 1][nb], hf[s], acc[nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// tower t with NCB 16-column blocks
template <bool TRIM, int NCB>
__device__ __forceinline__ void bh_tower(const HeadArgsB& a, int t, char* bh_lds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int rg = blockIdx.x / a.ksplit, ks = blockIdx.x - rg * a.ksplit;
    const int row = rg * BH_ROWS + wave * 16 + r;
    const int kr = a.K / a.ksplit, k0 = ks * kr;
    const bf16_t* __restrict__ hrow = a.H[t] + (size_t)row * a.ldh + 8 * g;
#ifdef PPO_STAMPS
    unsigned long long* gst = a.stamps ? a.stamps + ((size_t)t * gridDim.x + blockIdx.x) * 8 : nullptr;
#else
    unsigned long long* gst = nullptr;
#endif
    GSTAMP(0);
    f32x4 acc[NCB];
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kp = 0; kp < kr; kp += BH_KPASS) {
        if (kp) __syncthreads();                             // the previous image is spent
        if (kr - kp >= BH_KPASS) bh_pass<TRIM, true, NCB>(acc, hrow, a.W[t], a.ldw, k0 + kp, BH_KPASS / 32, bh_lds, gst);
        else bh_pass<TRIM, false, NCB>(acc, hrow, a.W[t], a.ldw, k0 + kp, (kr - kp) / 32, bh_lds, gst);
    }
    GSTAMP(4);
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) {
        const int j = 16 * nb + 4 * g;
        float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ks == 0) bj = *reinterpret_cast<const float4*>(a.bias[t] + j);
        *reinterpret_cast<float4*>(a.F[t] + (size_t)ks * a.f_split + (size_t)row * a.ldf + j) = make_float4(acc[nb][0] + bj.x, acc[nb][1] + bj.y, acc[nb][2] + bj.z, acc[nb][3] + bj.w);
    }
#ifdef PPO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GSTAMP(5);
}

// NCBP: 16-column blocks of the policy head (A / 16 rounded up to 1, 2, 4 or 8); the value head has one
template <bool TRIM, int NCBP>
__global__ __launch_bounds__(256) void bf16_heads_kernel(HeadArgsB a) {
    extern __shared__ __attribute__((aligned(16))) char bh_lds[];
    if (blockIdx.y == 0) bh_tower<TRIM, NCBP>(a, 0, bh_lds);
    else bh_tower<TRIM, 1>(a, 1, bh_lds);
}

// ---- act epilogue: sampling + neglogp (G:5894-6672) from the head GEMM's fp32 outputs ---------------------------------
// head outputs arrive as `hsplit` partial products hstride floats apart (HeadArgsB::ksplit), added here in range order
__device__ __forceinline__ float head_sum(const float* p, size_t idx, int hsplit, size_t hstride) {
    float s = p[idx];
    for (int k = 1; k < hsplit; ++k) s += p[idx + (size_t)k * hstride];
    return s;
}

struct SampleArgsB {
    const float* head[2]; int ldh; int hsplit; size_t hstride;      // [hsplit][rows_pad][Ap] fp32: tower 0 = mu, tower 1 column 0 = value
    const float* logstd;                // fp32 master
    const float* noise; float* action; float* det_action; float* value; float* neglogp;
    int n, A; uint32_t seed, rng_step, row_base;
};

// one wave per row, one lane per action (two past 64 actions): the row sums take the SAME shape as in bf16_loss_kernel, so the act model's
// neglogp and the train model's are the same bits on the same weights (first-epoch ratio exactly 1)
#define BS_ROWS 4
__global__ __launch_bounds__(64 * BS_ROWS) void bf16_sample_kernel(SampleArgsB a) {
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * BS_ROWS + r;
    const bool live = row < a.n;
    float ssq = 0.f, slog = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        if (j < a.A) {
            const float mu = live ? head_sum(a.head[0], (size_t)row * a.ldh + j, a.hsplit, a.hstride) : 0.f;
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
    }
    for (int o = 32; o > 0; o >>= 1) { ssq += __shfl_xor(ssq, o); slog += __shfl_xor(slog, o); }
    if (lane == 0 && live) {
        if (a.neglogp) a.neglogp[row] = 0.5f * ssq + HALF_LOG_2PI * (float)a.A + slog;
        if (a.value) a.value[row] = head_sum(a.head[1], (size_t)row * a.ldh, a.hsplit, a.hstride);
    }
}

// ---- loss + its gradient w.r.t. the head outputs (G:9428-11290, G:12609-22656): the arithmetic of
// train_fwd_bwd_kernel's middle section, 16 lanes per row, fp32; writes d mu / d v as bf16 and the
// per-block partial sums (bias / logstd gradients, loss terms) the gradient assembly adds up in a fixed order ---------
struct LossArgsB {
    const float* head[2]; int ldh; int hsplit; size_t hstride;
    const float* logstd;
    const float* actions; const float* advs; const float* returns; const float* old_values; const float* old_neglogp;
    const float* hyper;                 // {lr, cliprange}
    int n, A, Ap, rows_pad; float inv_n, ent_coef, vf_coef;
    bf16_t* dhead[2];                   // [rows_pad][Ap]  (tower 1: only column 0 is ever non-zero)
    float* slots[2]; int slot_w, slot_head, slot_aux, slot_loss;
};

#ifndef BL_ROWS
#define BL_ROWS 16                 // rows per block of the loss kernel: one slot row of partial sums per block
#endif
#define BL_EPT 2                   // action elements per lane: a row is ONE wave (64 lanes), A <= 128
// One wave per row, one lane per action: every lane requests its element's partial products at once and the row sums are wave
// reductions.  (Rounds 2-3: 16 lanes per row walking the actions in a loop of dependent loads, 32 rows per block = 128 workgroups of
// latency: 11.7 us for 8.8 MB.)
__global__ __launch_bounds__(64 * BL_ROWS) void bf16_loss_kernel(LossArgsB a) {
    extern __shared__ __attribute__((aligned(16))) float ls[];      // [R][Ap] dmu | [R][Ap] dlogstd | [R][4] pi terms | [R][2] vf terms
    float* dmu_s = ls; float* dls_s = ls + BL_ROWS * a.Ap; float* pt = dls_s + BL_ROWS * a.Ap; float* vt = pt + 4 * BL_ROWS;
    const int tid = threadIdx.x, r = tid >> 6, lane = tid & 63;
    const int row = blockIdx.x * BL_ROWS + r;
    const bool live = row < a.n;
    const float cr = a.hyper[1];
    const float g = a.inv_n;
    // every load of the lane first: head partials, actions, the row scalars
    float hp[BL_EPT][GB_HEAD_SPLIT], act_[BL_EPT], ls_[BL_EPT];
#pragma unroll
    for (int e = 0; e < BL_EPT; ++e) {
        const int j = lane + 64 * e;
#pragma unroll
        for (int k = 0; k < GB_HEAD_SPLIT; ++k) hp[e][k] = (j < a.A && k < a.hsplit) ? a.head[0][(size_t)k * a.hstride + (size_t)row * a.ldh + j] : 0.f;
        act_[e] = (j < a.A && live) ? a.actions[(size_t)row * a.A + j] : 0.f;
        ls_[e] = j < a.A ? a.logstd[j] : 0.f;
    }
    float vp[GB_HEAD_SPLIT];
#pragma unroll
    for (int k = 0; k < GB_HEAD_SPLIT; ++k) vp[k] = (lane == 0 && live && k < a.hsplit) ? a.head[1][(size_t)k * a.hstride + (size_t)row * a.ldh] : 0.f;
    const float adv = live ? a.advs[row] : 0.f;
    const float old_nlp_in = live ? a.old_neglogp[row] : 0.f;
    float Rv = 0.f, vo = 0.f;
    if (lane == 0 && live) { Rv = a.returns[row]; vo = a.old_values[row]; }
    // policy tower
    float mu[BL_EPT], z[BL_EPT], sigma[BL_EPT];
    float ssq = 0.f, slog = 0.f, sent = 0.f;
#pragma unroll
    for (int e = 0; e < BL_EPT; ++e) {
        const int j = lane + 64 * e;
        float m = hp[e][0];
#pragma unroll
        for (int k = 1; k < GB_HEAD_SPLIT; ++k) if (k < a.hsplit) m += hp[e][k];       // the partial products in range order (head_sum)
        mu[e] = m;
        const float logstd = m * 0.0f + ls_[e];
        sigma[e] = expf(logstd);
        const float act = live ? act_[e] : m;
        z[e] = (act - m) / sigma[e];
        if (j < a.A) { ssq += z[e] * z[e]; slog += logstd; sent += logstd + HALF_LOG_2PIE; }
    }
    for (int o = 32; o > 0; o >>= 1) { ssq += __shfl_xor(ssq, o); slog += __shfl_xor(slog, o); sent += __shfl_xor(sent, o); }
    const float nlp = 0.5f * ssq + HALF_LOG_2PI * (float)a.A + slog;
    const float old_nlp = live ? old_nlp_in : nlp;
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
    if (lane == 0) {
        const float dk = nlp - old_nlp;
        pt[r * 4 + 0] = live ? tf_max(m1, m2) : 0.f;
        pt[r * 4 + 1] = live ? sent : 0.f;
        pt[r * 4 + 2] = live ? dk * dk : 0.f;
        pt[r * 4 + 3] = (live && fabsf(ratio - 1.0f) > cr) ? 1.0f : 0.f;
    }
#pragma unroll
    for (int e = 0; e < BL_EPT; ++e) {
        const int j = lane + 64 * e;
        if (j < a.Ap) {
            float dmu = 0.f, dl = 0.f;
            if (j < a.A && live) {
                dl = d_nlp * (1.0f - z[e] * z[e]) - a.ent_coef * g;                      // AddN_2 G:21299
                dmu = d_nlp * (-(z[e] / sigma[e])) + dl * 0.0f;                          // AddN_3 G:22656
            }
            dmu_s[r * a.Ap + j] = dmu; dls_s[r * a.Ap + j] = dl;
            a.dhead[0][(size_t)row * a.Ap + j] = (bf16_t)dmu;
        }
    }
    // value tower (G:10213-10837, G:14975-19571)
    if (lane == 0) {
        float dv = 0.f, lossv = 0.f;
        if (live) {
            float v = vp[0];
#pragma unroll
            for (int k = 1; k < GB_HEAD_SPLIT; ++k) if (k < a.hsplit) v += vp[k];
            const float dvo = v - vo;
            const float vmin = tf_min(dvo, cr);
            const float vclip = vo + tf_max(vmin, -cr);
            const float e1 = v - Rv, e2 = vclip - Rv;
            const float s1 = e1 * e1, s2 = e2 * e2;
            lossv = tf_max(s1, s2);
            const float gv = a.vf_coef * 0.5f * a.inv_n;
            const float selv = (s1 >= s2) ? 1.0f : 0.0f;
            const float passv = ((vmin >= -cr) ? 1.0f : 0.0f) * ((dvo <= cr) ? 1.0f : 0.0f);
            dv = gv * selv * (2.0f * e1) + gv * (1.0f - selv) * (2.0f * e2) * passv;
        }
        vt[r * 2] = dv; vt[r * 2 + 1] = lossv;
        a.dhead[1][(size_t)row * a.Ap] = (bf16_t)dv;
    }
    __syncthreads();
    float* s0 = a.slots[0] + (size_t)blockIdx.x * a.slot_w;
    float* s1 = a.slots[1] + (size_t)blockIdx.x * a.slot_w;
    for (int j = tid; j < 2 * a.Ap; j += 64 * BL_ROWS) {      // threads [0, Ap): db_mu, [Ap, 2 Ap): dlogstd -- fp32 sums of the fp32 values, not of their bf16 roundings
        const float* src = j < a.Ap ? dmu_s + j : dls_s + (j - a.Ap);
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < BL_ROWS; ++q) sb += src[q * a.Ap];
        if (j < a.Ap) s0[a.slot_head + j] = sb; else s0[a.slot_aux + j - a.Ap] = sb;
    }
    if (tid >= 64 * BL_ROWS - 64 && tid < 64 * BL_ROWS - 60) { const int k = tid - (64 * BL_ROWS - 64); float s = 0.f; for (int q = 0; q < BL_ROWS; ++q) s += pt[q * 4 + k]; s0[a.slot_loss + k] = s; }
    if (tid == 64 * BL_ROWS - 32) { float sb = 0.f, sl = 0.f; for (int q = 0; q < BL_ROWS; ++q) { sb += vt[q * 2]; sl += vt[q * 2 + 1]; } s1[a.slot_head] = sb; s1[a.slot_loss] = sl; }
}

// ---- bf16 operand mirror of the fp32 master weights: keeps theta's padded layout (a cast of the whole vector) ----------------
__global__ __launch_bounds__(256) void bf16_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    bf16x4 o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    reinterpret_cast<bf16x4*>(dst)[i] = o;
}

// ---- gradient assembly for the bf16 path: one WAVE per 256-element chunk of the padded parameter vector, four consecutive
// elements per lane (16-byte slab loads); the chunk's sum of squares is a wave reduction.  Same sources as grad_reduce_kernel
// (kind 0 slabs, kind 1 per-block slots, kind 3 per-row-tile bias sums), same fixed summation orders. ---------------------
#define BGR_WAVES 16                // chunks (waves) per workgroup of bf16_grad_reduce_kernel: one sum-of-squares partial per workgroup
// the loss tail {pg, vf, ent, kl, cf, rows} (chunk n_blocks): lanes take the slot rows round robin, then meet in a fixed-shape tree; lane 0 keeps the sums
__device__ __forceinline__ void bgr_tail(const ReduceArgs& a, int lane, float (&tl)[5]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int tower = (k == 1) ? 1 : 0, off = a.slot_loss + (k <= 1 ? 0 : k - 1);
        float s = 0.f;
        for (int b = lane; b < a.n_rowblocks; b += 64) s += a.slots[tower][(size_t)b * a.slot_w + off];
        s = wave_sum_lane0(s);
        if (lane == 0) a.grad[(size_t)a.n_blocks * 256 + k] = s;
        tl[k] = s;
    }
    if (lane == 5) a.grad[(size_t)a.n_blocks * 256 + 5] = a.n_local;
    if (lane == 6) { a.beta_pow[0] = a.beta_pow[2]; a.beta_pow[1] = a.beta_pow[3]; }
}
// one 256-element chunk: this lane's four consecutive elements
__device__ __forceinline__ void bgr_chunk(const ReduceArgs& a, int chunk, int lane, float (&g)[4]) {
    const GradSrc s = a.src[chunk];
    const size_t idx = (size_t)chunk * 256 + 4 * lane;
    g[0] = g[1] = g[2] = g[3] = 0.f;
    if (s.kind == 0) {
        // the lane's 4 elements lie in one tile of the weight-gradient GEMM: as many partial sums as workgroups touched that tile
        const int e = (int)(idx - (size_t)s.base), row = e / s.pcol, col = e - row * s.pcol;
        const int tile = s.tile0 + (row / a.sk_bm) * (s.pcol / GB_N) + col / GB_N - a.sk_tile_base;
        const int cnt = (tile * a.sk_nst + a.sk_nst - 1) / a.sk_per - (tile * a.sk_nst) / a.sk_per + 1;
        for (int k = 0; k < cnt; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(a.slabs + (size_t)k * a.slab_stride + idx);
            g[0] += v.x; g[1] += v.y; g[2] += v.z; g[3] += v.w;
        }
    } else if (s.kind == 1 || s.kind == 3) {
        // rows of a small table, 16 bytes per lane and row, all loads independent (offsets are multiples of 4 floats on this path).
        // A vector of <= 64 (<= 128) elements occupies 16 (32) lanes: the other lanes take every 4th (2nd) row, and the partial sums
        // meet as (S0 + S1) + (S2 + S3) -- a quarter of the dependent-free but latency-bound loads per lane (the head bias / logstd
        // rows, one per 32 minibatch rows, were this launch's critical path)
        const int split = s.count <= 64 ? 4 : (s.count <= 128 ? 2 : 1);
        const int el = lane & (64 / split - 1), sub = lane / (64 / split);
        const int e0 = (int)((size_t)chunk * 256 - (size_t)s.base) + 4 * el;
        const float* p = (s.kind == 1 ? a.slots[s.tower] : a.direct) + s.slot_off + e0;
        const int rows = s.kind == 1 ? a.n_rowblocks : a.n_direct;
        const size_t stride = s.kind == 1 ? (size_t)a.slot_w : (size_t)a.direct_stride;
        if (e0 < s.count) {
#pragma unroll 8
            for (int b = sub; b < rows; b += split) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
                g[0] += v.x; g[1] += v.y; g[2] += v.z; g[3] += v.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (e0 + k >= s.count) g[k] = 0.f;
        }
        if (split > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (split == 4) g[k] += __shfl_xor(g[k], 16);
                g[k] += __shfl_xor(g[k], 32);
                if (sub != 0) g[k] = 0.f;               // (these lanes' own positions are padding)
            }
        }
    }
}
__global__ __launch_bounds__(64 * BGR_WAVES) void bf16_grad_reduce_kernel(ReduceArgs a) {
    __shared__ float wq[BGR_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // highest chunks first: the slot-summed vectors (head bias, logstd: many dependent-free but latency-bound loads per lane) sit at
    // the END of the parameter vector and must not be the launch's tail
    // (a bucket: chunks [chunk_lo, chunk_hi), highest first, the tail block with the bucket that reaches the vector's end)
    const bool whole = a.chunk_hi == 0;
    const int top = whole || a.chunk_hi == a.n_blocks ? a.n_blocks : a.chunk_hi - 1, lo = whole ? 0 : a.chunk_lo;
    const int chunk = top - (int)(blockIdx.x * BGR_WAVES + wave);
    float q = 0.f;
    if (chunk == a.n_blocks) { float tl[5]; bgr_tail(a, lane, tl); }
    else if (chunk >= lo) {
        float g[4];
        bgr_chunk(a, chunk, lane, g);
        *reinterpret_cast<float4*>(a.grad + (size_t)chunk * 256 + 4 * lane) = make_float4(g[0], g[1], g[2], g[3]);
        q = (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
        q = wave_sum_lane0(q);
    }
    if (lane == 0) wq[wave] = q;
    __syncthreads();
    if (tid == 0) {                                          // the workgroup's chunks in index order (wave BGR_WAVES-1 holds the lowest)
        float t = 0.f;
#pragma unroll
        for (int w = BGR_WAVES - 1; w >= 0; --w) t += wq[w];
        a.sumsq[blockIdx.x] = t;
    }
}

// ---- gradient assembly + clip + Adam in ONE launch (single GPU): bf16_grad_reduce_kernel's chunks and adam_kernel's elements are the same 256-element
// chunks, so a persistent launch -- BRA_GRID workgroups of 16 waves, one per CU, each wave up to NR chunks -- keeps the assembled gradient (and the Adam
// slots and weights it requested meanwhile) in REGISTERS across the one thing in between, the global norm.  (store_grad == 0: the assembled gradient is
// not written at all -- 4 of the launch's 41 bytes per parameter; ppo_get_last_grad rebuilds it from the slabs with bf16_grad_reduce_kernel when somebody asks.)  Workgroup b plays the workgroups
// j = r BRA_GRID + b, r = 0 .. NR - 1, of the bf16_grad_reduce_kernel launch it replaces: partial j is formed exactly as there, and the sum over r is
// exactly what adam_kernel's thread b adds up from the partials (it strides them by 256), so the table the workgroups meet on has ONE {epoch, sum} word
// per workgroup and the norm, the clip factor and every element come out with the bits of the two launches (tests/test_bf16_path.py).  What goes away:
// adam_kernel's read of the gradient (4 bytes per parameter), a launch boundary, and the wait for the Adam slots (they arrive while the slabs are
// summed).  Bounded wait; PPO_HIP_NO_REDUCE_ADAM=1 keeps the two launches.
#define BRA_GRID 256
struct ReduceAdamArgs {
    ReduceArgs r;
    float* theta; float* m; float* v; bf16_t* theta_bf;
    const float* hyper; float beta1, beta2, eps, max_norm;
    float* loss_row; float* norm_out;
    unsigned long long* ent;     // [BRA_GRID] {epoch << 32 | sum of this workgroup's partials}, [BRA_GRID] raised when a wait timed out
    int n_old;                   // workgroups of the bf16_grad_reduce_kernel launch this one replaces
    int store_grad;              // 0: only the loss tail of r.grad is written
};
template <int NR>
__global__ __launch_bounds__(64 * BGR_WAVES) void bf16_reduce_adam_kernel(ReduceAdamArgs a) {
    __shared__ float wq[NR][BGR_WAVES];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = (int)blockIdx.x;
    // this launch's epoch, the powers and the learning rate NOW: workgroup 0 advances the powers after the meeting (atomic loads in front of the barriers)
    const unsigned epoch = (unsigned)(__hip_atomic_load(a.ent + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) + 1u;
    const float b1p = __hip_atomic_load(a.r.beta_pow + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float b2p = __hip_atomic_load(a.r.beta_pow + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float lr = __hip_atomic_load(a.hyper, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr bool PREF = NR <= 5;        // Adam slots + weights requested beside the slabs and held across the meeting (more rounds than that do not fit the registers)
    float G[NR][4]; float4 M[PREF ? NR : 1], V[PREF ? NR : 1], T[PREF ? NR : 1];
    float tl[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = r * BRA_GRID + b;
        const int chunk = a.r.n_blocks - (j * BGR_WAVES + wave);
        float q = 0.f;
        G[r][0] = G[r][1] = G[r][2] = G[r][3] = 0.f;
        if constexpr (PREF) M[r] = V[r] = T[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < a.n_old) {
            if (chunk == a.r.n_blocks) bgr_tail(a.r, lane, tl);
            else if (chunk >= 0) {
                const size_t idx = (size_t)chunk * 256 + 4 * lane;
                if constexpr (PREF) { M[r] = *reinterpret_cast<const float4*>(a.m + idx); V[r] = *reinterpret_cast<const float4*>(a.v + idx); T[r] = *reinterpret_cast<const float4*>(a.theta + idx); }
                bgr_chunk(a.r, chunk, lane, G[r]);
                if (a.store_grad) *reinterpret_cast<float4*>(a.r.grad + idx) = make_float4(G[r][0], G[r][1], G[r][2], G[r][3]);
                q = (G[r][0] * G[r][0] + G[r][1] * G[r][1]) + (G[r][2] * G[r][2] + G[r][3] * G[r][3]);
                q = wave_sum_lane0(q);
            }
        }
        if (lane == 0) wq[r][wave] = q;
    }
    __syncthreads();
    if (tid == 0) {
        float S = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r * BRA_GRID + b < a.n_old) {
                float t = 0.f;
#pragma unroll
                for (int w = BGR_WAVES - 1; w >= 0; --w) t += wq[r][w];       // the old workgroup's chunks in index order
                a.r.sumsq[r * BRA_GRID + b] = t;
                S += t;                                                       // adam_kernel's thread b: partials b, b + 256, ... in this order
            }
        __hip_atomic_store(a.ent + b, ((unsigned long long)epoch << 32) | __float_as_uint(S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the meeting: thread t < 256 watches word t; when all show this launch's epoch the values read ARE adam_kernel's per-thread sums ---------------
    float s = 0.f;
    unsigned polls = 0;
    for (;;) {
        bool ok = true;
        if (tid < BRA_GRID) {
            const unsigned long long e = __hip_atomic_load(a.ent + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (int)((unsigned)(e >> 32) - epoch) >= 0; s = __uint_as_float((unsigned)e);
        }
        if (__syncthreads_and(ok ? 1 : 0)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++polls > (1u << 20)) { if (tid == 0) __hip_atomic_store(a.ent + BRA_GRID, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    s = wave_sum_lane0(s);
    if (tid < 256 && lane == 0) red[wave] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    float scale = a.max_norm * tf_min(1.0f / norm, 1.0f / a.max_norm);          // G:24289-24472
    if (!isfinite(norm)) scale = __builtin_nanf("");                            // G:24493-24543
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = r * BRA_GRID + b;
        const int chunk = a.r.n_blocks - (j * BGR_WAVES + wave);
        if (j < a.n_old && chunk >= 0 && chunk < a.r.n_blocks) {
            const size_t idx = (size_t)chunk * 256 + 4 * lane;
            float4 m4, v4, t4;
            if constexpr (PREF) { m4 = M[r]; v4 = V[r]; t4 = T[r]; }
            else { m4 = *reinterpret_cast<const float4*>(a.m + idx); v4 = *reinterpret_cast<const float4*>(a.v + idx); t4 = *reinterpret_cast<const float4*>(a.theta + idx); }
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
            float mo[4], vo[4], to[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_element<false>(G[r][k] * scale, mv[k], vv[k], tv[k], omb1, omb2, alpha, a.eps, mo[k], vo[k], to[k]);
            *reinterpret_cast<float4*>(a.m + idx) = make_float4(mo[0], mo[1], mo[2], mo[3]);
            *reinterpret_cast<float4*>(a.v + idx) = make_float4(vo[0], vo[1], vo[2], vo[3]);
            *reinterpret_cast<float4*>(a.theta + idx) = make_float4(to[0], to[1], to[2], to[3]);
            bf16x4 o4; o4[0] = (bf16_t)to[0]; o4[1] = (bf16_t)to[1]; o4[2] = (bf16_t)to[2]; o4[3] = (bf16_t)to[3];
            *reinterpret_cast<bf16x4*>(a.theta_bf + idx) = o4;
        }
    }
    if (b == 0 && wave == 0) {                                                  // (the wave that summed the loss tail: chunk n_blocks = old workgroup 0, wave 0)
        if (lane == 0) {
            a.r.beta_pow[2] = b1p * a.beta1;                                    // G:31217-31342 (after the applies)
            a.r.beta_pow[3] = b2p * a.beta2;
            if (a.norm_out) *a.norm_out = norm;
            if (a.loss_row) {
#pragma unroll
                for (int k = 0; k < 5; ++k) { float r = tl[k] / a.r.n_local; if (k == 1 || k == 3) r = 0.5f * r; a.loss_row[k] = r; }      // vf_loss, approxkl carry the 0.5
            }
        }
    }
}
