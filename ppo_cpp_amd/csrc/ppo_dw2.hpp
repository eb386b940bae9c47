// ppo_dw2.hpp -- weight gradients AND gradient assembly of one train step in ONE launch, for hidden [256,256] behind an observation /
// action tile of 32 or 64 columns (BASELINE configs[2]: 18 / 18; the reference's 36-observation hexapod, env/hexapod_closed_loop_env.hpp:20)
// (SURVEY 8a rows a13 / a14's inputs: G's .../MatMul_grad/MatMul_1, .../Add_grad/Sum_1 and loss Mean nodes).
//
// What it replaces: weight_grad_kernel (8 row splits, 64x64 tiles, a table of tiles in memory) followed by grad_reduce_kernel
// (a launch whose only job was to add 8 slabs and 128 per-row-block slots: 4.7 us at its launch + round-trip floor).  Here
//   * the tile a workgroup owns follows from blockIdx alone (no table read in front of the first operand load);
//   * 4 row splits x 64 tiles of [64 x 32] (+ one thin strip of the first-layer / policy-head gradient each) = 256 workgroups of
//     8 waves; the rows are walked in 64-row chunks (ANY minibatch that is a whole number of chunks: the host pads the train
//     kernel's grid to 64 rows, its dead rows are zeros), split s owning chunks [s C / 4, (s + 1) C / 4);
//   * the split that finishes a tile LAST adds the 4 slabs in split order (bitwise reproducible whoever is last), writes the
//     finished gradient tile and its sum of squares: the hand-off is the guide's counter form -- slabs stored write-through
//     (sc1), every wave drains its stores, a workgroup barrier, ONE agent-scope arrival; the last arriver reads with sc1 loads
//     (why no release / acquire fences: at the arrival below, with the measurement);
//   * the per-row-block slots of the train kernel (bias / logstd / value-head gradients, loss sums) are spread over the 256
//     workgroups, requested at kernel entry and finished after the matrix work.
// clip + Adam (adam_kernel) then reads `grad` and the 64 + 256 partial sums of squares; nothing else changes.
#pragma once
#include "ppo_kernels.hpp"

#define DW2_THREADS 512
#define DW2_SPLITS 4
#define DW2_TILES 64                        // 2 towers x (4 x 8) tiles of 64 x 32 of the [256 x 256] second-layer gradient
#define DW2_GRID (DW2_TILES * DW2_SPLITS)
#define DW2_CH 64                           // minibatch rows per LDS chunk
#define DW2_NBUF 3
// one chunk in LDS (floats): X [64][64] | Y [64][32] | U [64][UW] | W [64][16], UW = the wider of the observation and the action tile;
// a ring of three chunks + 64 words of flags / scratch
template <int KP0, int AP>
struct Dw2L {
    static constexpr int UW = KP0 > AP ? KP0 : AP;
    static constexpr int OX = 0, OY = DW2_CH * 64, OU = OY + DW2_CH * 32, OW = OU + DW2_CH * UW, BUF = OW + DW2_CH * 16;
    static constexpr int LDS_FLOATS = DW2_NBUF * BUF + 64;       // 108 KB at 32 / 32, 132 KB with a 64-wide tile: one workgroup per CU (the hand-off's measured form)
};
#ifdef PPO_STAMPS
#define DW2_STAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = (i) == 15 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
#else
#define DW2_STAMP(i) do { } while (0)
#endif
#define DW2_SLOTK 8                         // row blocks per lane of a slot job: up to 256 row blocks (4096 minibatch rows)

struct SlotJob { int tower; int slot_off; int dst; int in_norm; };    // in_norm: 0 = a loss sum (no norm, no Adam); else 1 + the element's index in the small-parameter mirror

struct Dw2Args {
    const float* x0g;            // [n][KP0] layer-0 input (written by the policy tower's workgroups)
    const float* h1[2];          // [n][256] layer-1 input per tower
    const float* h2pi;           // [n][256] policy head input
    const float* dy0[2];         // [n][256] dLoss/d(pre-activation of layer 0)
    const float* dy1[2];         // [n][256] ... of layer 1
    const float* dmug;           // [n][AP]
    int n;                       // minibatch rows as the train kernel wrote them: a multiple of 64 (rows past the real count are zeros)
    float* slabs; unsigned long long slab_stride;      // [4][P_pad]
    unsigned* counters;          // [DW2_TILES] arrivals; zero between launches (the last arriver resets its word)
    float* grad;                 // [P_pad + 8]
    float* parts;                // [DW2_TILES + DW2_GRID] partial sums of squares for the global norm
    int w0_off[2], w1_off[2], wmu_off;
    const SlotJob* jobs; int n_jobs, jobs_per_wg;
    const float* slots[2]; int n_rowblocks, slot_w;
    float n_local; float* beta_pow; int tail_off;
    unsigned long long* stamps;  // diagnostic builds only (-DPPO_STAMPS): [grid][16]
};

// write-through (sc1) 16-byte load for bytes another workgroup stored write-through in this launch.  Inline asm: the compiler
// does not count it, so the caller waits with dw2_wait4() (which names every destination) before the first use.
__device__ __forceinline__ f32x4 dw2_ld_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void dw2_wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory");
}

// One 1 KB piece (64 lanes x 16 bytes) global -> LDS without a register stop; dst is wave-uniform, src per lane.  Inline asm on
// purpose: hipcc treats the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of every later ds_read,
// which drains the pieces this kernel keeps in flight across two iterations; its own counted waits (wait_keep_one) order them.
// M0 carries the LDS byte address and is restored in the same statement (the compiler owns M0).
__device__ __forceinline__ void dw2_dma(const float* src_base, unsigned lane_off, float* dst) {
    const char* src = reinterpret_cast<const char*>(src_base) + lane_off;
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)dst));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}

// The thin strip a workgroup carries beside its [64 x 32] tile: a UWK-column operand U (layer-0 input / d mu; row pitch UWK in LDS)
// against 16 columns of a 256-column operand W (layer-0 dY / policy-head input).  Per 64-row chunk wave w owns k-steps 2w, 2w+1.
// (one register array of the wider form for both kinds: two members selected by `kind` end up in scratch)
template <int UWK, int NU>
__device__ __forceinline__ void dw2_read_u(float (&u)[2][NU], const float* ubuf, int wave, int g, int c) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float* p = ubuf + (8 * wave + 4 * ks + g) * UWK + (UWK / 16) * c;
        if constexpr (UWK == 32) {
            const float2 t = *reinterpret_cast<const float2*>(p); u[ks][0] = t.x; u[ks][1] = t.y;
            if constexpr (NU == 4) { u[ks][2] = 0.f; u[ks][3] = 0.f; }       // (every element defined on both paths: keeps the array in registers)
        }
        else { const float4 t = *reinterpret_cast<const float4*>(p); u[ks][0] = t.x; u[ks][1] = t.y; u[ks][2] = t.z; u[ks][3] = t.w; }
    }
}
// U_FIRST: out[U column][W column] (first-layer gradient strip [UWK x 16]); else out[W column][U column] (head strip [16 x UWK])
template <int UWK, bool U_FIRST, int NU>
__device__ __forceinline__ void dw2_strip_mma(const float (&u)[2][NU], const float (&ww)[2], f32x4 (&sacc)[NU]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < UWK / 16; ++t)
            sacc[t] = U_FIRST ? __builtin_amdgcn_mfma_f32_16x16x4f32(u[ks][t], ww[ks], sacc[t], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x4f32(ww[ks], u[ks][t], sacc[t], 0, 0, 0);
}

// The kernel's body; b = the workgroup's linear index 0 .. DW2_GRID - 1 (blockIdx.x).  (Round 5 also called it as the second phase of a fused train + weight-gradient
// launch and, in a third form, let the 64 finishers apply clip + Adam behind a grid-wide meeting: both measured slower than the launches they replaced
// -- profiles/r05_a_*, r05_g_* -- and removed in round 6; branch experiments-r05.)
// PEER (data parallel over peer-mapped regions, ppo_peer.hpp): the workgroup that finishes a tile LAST holds the assembled tile -- it also stores it,
// its strip and the 24-odd slot-job results of the tile's four workgroups into slot [rank] of EVERY rank's gather region (16-byte stores), takes ONE
// system-scope release, and arrives on the local counter; the last of the 64 finishers raises this rank's flag at every peer.  No push launch; the
// sum over the ranks happens inside adam_kernel<.., 2>.  (Round 2 pushed from every workgroup that wrote gradient elements -- 590 of them, each with
// its own fence -- and lost; here at most 64 workgroups fence.)
template <int KP0, int AP, bool PEER = false>
__device__ __forceinline__ void dw2_body(const Dw2Args& a, float* lds, const int b, const PeerDev* pp = nullptr) {
    typedef Dw2L<KP0, AP> LD;
    DW2_STAMP(15); DW2_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    // workgroups are dealt round-robin over the 8 XCDs: XCD x takes row split x >> 1 of tower x & 1, so the rows x 2 operands
    // of that split (1 MB at M = 2048) are fetched from the fabric once per XCD and shared by its 32 tiles through the L2, and
    // they are exactly the rows the train kernel's workgroups on XCD x wrote (its xcd_map 1).  Speed only.
    const int split = (b >> 1) & 3, tower = b & 1, tile = b >> 3;
    const int gtile = tower * 32 + tile;
    const int i0 = (tile >> 3) * 64, j0 = (tile & 7) * 32;
    // strip: kind 0 = [KP0 x 16] of the first-layer gradient (both towers, tiles 0..15), kind 1 = [16 x AP] of the policy-head
    // gradient (policy tower, tiles 16..31), kind 2 = none.  Either way an operand U of KP0 / AP columns (layer-0 input / d mu) and 16
    // columns [16 s, 16 s + 16) of a 256-column operand W (layer-0 dY / policy-head input).
#ifdef DW2_NOSTRIP
    const int kind = 2;          // timing experiment only: results are wrong
#else
    const int kind = uni(tile < 16 ? 0 : (tower == 0 ? 1 : 2));
#endif
    const int sidx = tile & 15;
    const int uwk = kind == 0 ? KP0 : AP;                    // (wave-uniform) columns of this workgroup's U operand
    // the slot-job descriptor of this half-wave: requested first, needed after the first chunk is on its way
    const int jl = tid >> 5, jb = b * a.jobs_per_wg + jl;
    const bool has_job = jl < a.jobs_per_wg && jb < a.n_jobs;
    const int4 jraw = reinterpret_cast<const int4*>(a.jobs)[has_job ? jb : 0];     // unconditional: the wait sits at the first use, not here
    // ---- operand staging: LDS-DMA, 1 KB pieces; the image of a chunk is [row][cols] (16-byte fragment reads of a 64-column row
    // cover all 64 banks per 16 lanes; 32- and 16-column rows alternate bank halves by themselves)
    const int nct = a.n / DW2_CH;                            // chunks of the minibatch; split s owns [s nct / 4, (s + 1) nct / 4)
    const int cb = (split * nct) >> 2, nch = (((split + 1) * nct) >> 2) - cb;
    const size_t r0 = (size_t)cb * DW2_CH;
    const float* Xg = uni(a.h1[tower]) + r0 * 256;
    const float* Yg = uni(a.dy1[tower]) + r0 * 256;
    const float* Ug = uni(kind == 1 ? a.dmug : a.x0g) + r0 * uwk;
    const float* Wg = uni(kind == 1 ? a.h2pi : a.dy0[tower]) + r0 * 256;
    const unsigned lx = (unsigned)(((lane >> 4) * 256 + i0 + 4 * (lane & 15)) * 4);
    const unsigned ly = (unsigned)(((lane >> 3) * 256 + j0 + 4 * (lane & 7)) * 4);
    const unsigned lu = (unsigned)(lane * 16);               // U rows are contiguous in memory ([n][uwk]): a piece is 1 KB as it lies
    const unsigned lw = (unsigned)(((lane >> 2) * 256 + 16 * sidx + 4 * (lane & 3)) * 4);
    // pieces of one 64-row chunk: X 16 (4 rows each), Y 8, U 8 or 16 (2 KB / 4 KB of rows), W 8 half pieces (8 rows x 64 bytes); wave w
    // requests those of ITS rows 8 w .. 8 w + 7: X 2w, 2w+1, Y w, U w (or 2w, 2w+1), W w: `np` requests per chunk and wave, the count the
    // in-loop waits leave in flight
    const int np = uni(3 + (kind == 2 ? 0 : (uwk >> 5) + 1));
    auto stage_x = [&](int ch) __attribute__((always_inline)) {
        float* buf = lds + (ch % DW2_NBUF) * LD::BUF;
        const size_t rb = (size_t)ch * DW2_CH;
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int j = 2 * wave + k; dw2_dma(Xg + (rb + 4 * j) * 256, lx, buf + LD::OX + j * 256); }
    };
    auto stage_rest = [&](int ch) __attribute__((always_inline)) {
        float* buf = lds + (ch % DW2_NBUF) * LD::BUF;
        const size_t rb = (size_t)ch * DW2_CH;
        dw2_dma(Yg + (rb + 8 * wave) * 256, ly, buf + LD::OY + wave * 256);
        if (kind != 2) {
            if (uwk == 32) dw2_dma(Ug + rb * 32 + wave * 256, lu, buf + LD::OU + wave * 256);
            else {
#pragma unroll
                for (int k = 0; k < 2; ++k) dw2_dma(Ug + rb * 64 + (2 * wave + k) * 256, lu, buf + LD::OU + (2 * wave + k) * 256);
            }
            if (lane < 32) dw2_dma(Wg + (rb + 8 * wave) * 256, lw, buf + LD::OW + wave * 128);      // this wave's 8 rows x 16 columns: half a piece
        }
    };
    auto stage = [&](int ch) __attribute__((always_inline)) { stage_x(ch); stage_rest(ch); };
    // wait until at most ONE chunk's pieces of this wave are still in flight (vector-memory operations complete in order)
    auto wait_keep_one = [&]() __attribute__((always_inline)) {
        if (np == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (np == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (np == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    if (nch > 0) stage(0);
    if (nch > 1) stage(1);
    if (nch > 2) stage(2);
    DW2_STAMP(1);
    // ---- matrix work: wave w owns rows 8 w .. 8 w + 7 of every chunk (the K dimension of the gradient is split 8 ways inside the
    // workgroup, 4 x more across the row splits) and computes a partial of the WHOLE [64 x 32] tile from them: per 4-row k-step 4 x 2
    // matrix instructions fed by one 16-byte and one 8-byte LDS read (tiles interleaved: instruction (i, j) covers gradient rows
    // i0 + 4 m + i, columns j0 + 2 n + j); the eight partials meet in LDS at the end.  Strip: the same rows.  A wave reads exactly the
    // pieces it requested itself.
    const int ax = (8 * wave + g) * 64 + 4 * c, by = (8 * wave + g) * 32 + 2 * c;
    const int swo = (8 * wave + g) * 16 + c;
    constexpr int NU = LD::UW / 16;
    struct Frags { float4 xa[2]; float2 yb[2]; float u[2][NU]; float ww[2]; };
    auto read_main = [&](Frags& f, int ch) __attribute__((always_inline)) {
        const float* buf = lds + (ch % DW2_NBUF) * LD::BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f.xa[ks] = *reinterpret_cast<const float4*>(buf + LD::OX + ax + ks * 256);
            f.yb[ks] = *reinterpret_cast<const float2*>(buf + LD::OY + by + ks * 128);
        }
    };
    auto read_strip = [&](Frags& f, int ch) __attribute__((always_inline)) {
        const float* buf = lds + (ch % DW2_NBUF) * LD::BUF;
        if (kind != 2) {
            if (KP0 == AP || kind == 0) dw2_read_u<KP0, NU>(f.u, buf + LD::OU, wave, g, c); else dw2_read_u<AP, NU>(f.u, buf + LD::OU, wave, g, c);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) f.ww[ks] = buf[LD::OW + swo + ks * 64];
        }
    };
    auto read_frags = [&](Frags& f, int ch) __attribute__((always_inline)) { read_main(f, ch); read_strip(f, ch); };
    f32x4 acc[4][2], sacc[NU];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NU; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma_main = [&](const Frags& f, int ks) __attribute__((always_inline)) {          // one k-step of the [64 x 32] tile: 8 matrix instructions
#ifdef DW2_NOMFMA
        if (a.n < 0)
#endif
        {
            const float xv[4] = {f.xa[ks].x, f.xa[ks].y, f.xa[ks].z, f.xa[ks].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], f.yb[ks].x, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], f.yb[ks].y, acc[i][1], 0, 0, 0);
            }
        }
    };
    auto mma_strip = [&](const Frags& f) __attribute__((always_inline)) {
#ifdef DW2_NOMFMA
        if (a.n < 0) {
#endif
        if (kind == 0) dw2_strip_mma<KP0, true, NU>(f.u, f.ww, sacc);
        else if (kind == 1) dw2_strip_mma<AP, false, NU>(f.u, f.ww, sacc);
#ifdef DW2_NOMFMA
        }
#endif
    };
    // chunks 0 and 1 (and the job descriptor) have landed once at most the pieces of chunk 2 are in flight
    if (nch > 2) wait_keep_one(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    if (nch > 0) read_frags(fa, 0);
    // ---- slot jobs: 32 lanes per element, loads issued now, finished after the matrix work ------------------------------------
    float sj[DW2_SLOTK];
    auto load_slots = [&]() __attribute__((always_inline)) {
        const int ln = tid & 31;
        const float* p = (jraw.x ? a.slots[1] : a.slots[0]) + jraw.y;
#pragma unroll
        for (int k = 0; k < DW2_SLOTK; ++k) { const int rb = ln + 32 * k; sj[k] = (has_job && rb < a.n_rowblocks) ? p[(size_t)rb * a.slot_w] : 0.f; }
    };
    load_slots();
    DW2_STAMP(2);
    // One iteration = the matrix instructions of chunk i from registers; chunk i+3 is requested into the buffer of chunk i (this wave
    // read its fragments of it in the last iteration), the fragments of chunk i+1 are read (its pieces have landed: the wait at the
    // bottom of the last iteration).  A wave touches ONLY pieces it requested itself, so there is NO workgroup barrier in the loop: the
    // per-chunk barrier of the round-3 form cost 470 cycles a chunk -- 3.8 k of an 18.9 k-cycle loop whose matrix instructions take
    // 10.2 k -- and kept the two waves of a SIMD in phase, their requests and reads under nobody's matrix instructions
    // (profiles/r04_e_stamps_dw2_loop_experiments.txt).
    auto iteration = [&](int i, Frags& cur, Frags& nxt) __attribute__((always_inline)) {
#ifdef DW2_NOREAD
        const bool st = i + 3 < nch, rd = false;          // timing experiment only: results are wrong
#else
        const bool st = i + 3 < nch, rd = i + 1 < nch;
#endif
        mma_main(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef DW2_NODMA
        if (st) stage(i + 3);
#endif
        if (rd) read_frags(nxt, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_main(cur, 1);
        mma_strip(cur);
        if (i + 3 < nch) wait_keep_one(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk i+2 has landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          // this wave's reads of chunk i+1 are done
    };
    for (int i = 0; i < nch; i += 2) {
        iteration(i, fa, fb);
        if (i + 1 < nch) iteration(i + 1, fb, fa);
    }
    DW2_STAMP(3);
    // ---- park: main partials [4 K quarters][64][32], strip partials [8 waves][16 uwk]; slot jobs finish here too --------------------
    __syncthreads();                                          // the partials overlay the chunk ring: every wave is done with it (and its requests have landed)
    float* park = lds;
    float* spark = lds + 8 * 2048;
    static_assert(8 * 2048 + 8 * 16 * LD::UW <= DW2_NBUF * LD::BUF, "park + strip partials fit the chunk ring");
    float* red2 = lds + DW2_NBUF * LD::BUF + 32;
    const int sn = 16 * uwk;                                 // elements of this workgroup's strip
#pragma unroll
    for (int i = 0; i < 4; ++i)                              // instruction (i, j): gradient rows i0 + 4 m + i, columns j0 + 2 n + j
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float2*>(park + wave * 2048 + (4 * (4 * g + r) + i) * 32 + 2 * c) = make_float2(acc[i][0][r], acc[i][1][r]);
    if (kind == 0) {                                         // [KP0 x 16]: instruction t holds U columns (KP0 / 16) m + t
#pragma unroll
        for (int t = 0; t < KP0 / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) spark[wave * sn + ((KP0 / 16) * (4 * g + r) + t) * 16 + c] = sacc[t][r];
    } else if (kind == 1) {                                  // [16 x AP]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = spark + wave * sn + (4 * g + r) * AP + (AP / 16) * c;
            if constexpr (AP == 32) *reinterpret_cast<float2*>(dst) = make_float2(sacc[0][r], sacc[1][r]);
            else *reinterpret_cast<float4*>(dst) = make_float4(sacc[0][r], sacc[1][r], sacc[2][r], sacc[3][r]);
        }
    }
    {
        float s = ((sj[0] + sj[1]) + (sj[2] + sj[3])) + ((sj[4] + sj[5]) + (sj[6] + sj[7]));
        s = half_sum_lane0(s);                                       // within the 32 lanes of the job
        if ((tid & 31) == 0) {
            if (has_job) st_wt<PEER>(a.grad + jraw.z, s);          // (PEER: the tile's finisher reads it back with a write-through load and pushes it)
            red2[tid >> 5] = (has_job && jraw.w) ? s * s : 0.f;
        }
    }
    __syncthreads();
    float4 m4 = *reinterpret_cast<const float4*>(park + 4 * tid);
#pragma unroll
    for (int q = 1; q < 8; ++q) { const float4 p = *reinterpret_cast<const float4*>(park + q * 2048 + 4 * tid); m4.x += p.x; m4.y += p.y; m4.z += p.z; m4.w += p.w; }
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool strip_thread = kind != 2 && 4 * tid < sn;
    if (strip_thread) {
        sv = *reinterpret_cast<const float4*>(spark + 4 * tid);
#pragma unroll
        for (int w = 1; w < 8; ++w) { const float4 p = *reinterpret_cast<const float4*>(spark + w * sn + 4 * tid); sv.x += p.x; sv.y += p.y; sv.z += p.z; sv.w += p.w; }
    }
    // element offsets inside the padded parameter vector
    const unsigned moff = (unsigned)(a.w1_off[tower] + (i0 + (tid >> 3)) * 256 + j0 + 4 * (tid & 7));
    unsigned soff = 0;
    if (kind == 0) soff = (unsigned)(a.w0_off[tower] + (tid >> 2) * 256 + 16 * sidx + 4 * (tid & 3));                 // W0 [KP0][256]
    else if (kind == 1) soff = (unsigned)(a.wmu_off + (16 * sidx + (4 * tid) / AP) * AP + (4 * tid) % AP);            // W_mu [256][AP]
    // ... and inside a SLAB every 128-byte line belongs to ONE tile: the main tile's row segments (32 floats) and the head's strip rows (AP floats) are whole lines as they
    // lie; a first-layer strip -- 16 columns of a 256-float row, HALF a line whose other half is the neighbour tile's -- is kept contiguous instead ([16 strips][KP0][16]: only
    // the finisher reads a slab, and it writes the natural layout into the gradient).  Rounds 3 - 5 stored the strips in place, which broke the hand-off's own rule (the last
    // arriver must meet no line another workgroup on its XCD touched in this launch: sc1 loads bypass the L1, not the L2); never seen to matter, bit-identical and
    // equally fast either way (profiles/r06_b_dw2_own_lines_ab.txt), so the layout that keeps the rule true is the only one since round 6.
    const unsigned sslab = kind == 0 ? (unsigned)(a.w0_off[tower] + sidx * sn + 4 * tid) : soff;
    float* slab = a.slabs + (size_t)split * a.slab_stride;
    st_wt4<true>(slab + moff, m4);
    if (strip_thread) st_wt4<true>(slab + sslab, sv);
    if (tid == 0) {
        float q = 0.f;
        for (int j = 0; j < 16; ++j) q += red2[j];
        a.parts[DW2_TILES + b] = q;
    }
    if (b == 0 && tid == 64) st_wt<PEER>(a.grad + a.tail_off + 5, a.n_local);
    if (b == 0 && tid == 65) { a.beta_pow[0] = a.beta_pow[2]; a.beta_pow[1] = a.beta_pow[3]; }    // cur <- next (adam writes next)
    DW2_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                                // ... before the one arrival that signals for all of them
    DW2_STAMP(5);
    int* flag = reinterpret_cast<int*>(lds + DW2_NBUF * LD::BUF);
    if (tid == 0) {
        // The hand-off is ordered by the HARDWARE's rules for sc1 accesses, not by release / acquire of the language memory model:
        // every slab store above is `global_store ... sc1` (written through to the memory side, coherent across the XCDs' L2s), every
        // storing wave has waited `s_waitcnt vmcnt(0)` (its stores are complete at that level), the barrier orders all of them before
        // this thread, and the last arriver reads with `global_load ... sc1` (never served by its own XCD's L2).  The stores and loads
        // are volatile inline asm with memory clobbers, so the compiler cannot move them across the atomic either.  The model's form --
        // an agent-scope RELEASE on this arrival and an ACQUIRE fence in the last arriver -- was measured in rounds 4 and 6 (same bits): on
        // gfx950 an agent-scope release is a write-back of the XCD's whole L2 and the acquire an invalidate, 256 + 64 of them per launch:
        // 15.6 -> 30.8 us per launch (profiles/r04_a_dw2_model_fences_measured_not_kept.txt).  ppo_peer.hpp pays for fences because its
        // data crosses DEVICES through plain stores; here both sides are sc1 accesses of one device.
        const unsigned old = __hip_atomic_fetch_add(a.counters + gtile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old == DW2_SPLITS - 1) ? 1 : 0;
    }
    __syncthreads();
    DW2_STAMP(6);
    const bool fin = flag[0] != 0;
    f32x4 t4 = {0.f, 0.f, 0.f, 0.f}, u4 = t4;
    if (fin) {
        // last arriver: the four slabs in split order (its own included: same bits whoever is last), all loads first
        f32x4 p[DW2_SPLITS], q[DW2_SPLITS];
#pragma unroll
        for (int s = 0; s < DW2_SPLITS; ++s) {
            p[s] = dw2_ld_sc1(a.slabs + (size_t)s * a.slab_stride + moff);
            q[s] = dw2_ld_sc1(a.slabs + (size_t)s * a.slab_stride + (strip_thread ? sslab : moff));
        }
        dw2_wait4(p[0], p[1], p[2], p[3]);
        dw2_wait4(q[0], q[1], q[2], q[3]);
        t4 = p[0]; u4 = q[0];
#pragma unroll
        for (int s = 1; s < DW2_SPLITS; ++s) { t4 += p[s]; u4 += q[s]; }
        *reinterpret_cast<float4*>(a.grad + moff) = make_float4(t4[0], t4[1], t4[2], t4[3]);
        float sq = (t4[0] * t4[0] + t4[1] * t4[1]) + (t4[2] * t4[2] + t4[3] * t4[3]);
        if (strip_thread) {
            *reinterpret_cast<float4*>(a.grad + soff) = make_float4(u4[0], u4[1], u4[2], u4[3]);
            sq += (u4[0] * u4[0] + u4[1] * u4[1]) + (u4[2] * u4[2] + u4[3] * u4[3]);
        }
        sq = wave_sum_lane0(sq);
        float* red = lds + DW2_NBUF * LD::BUF + 16;
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w];
            a.parts[gtile] = s;
            __hip_atomic_store(a.counters + gtile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        }
        __syncthreads();
        if constexpr (PEER) {
            const PeerDev& p = *pp;
            const unsigned sq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u, par = sq & 1u;     // (published by the LAST finisher only, below)
            // the slot-job results of this tile's four workgroups (their stores were write-through and drained before their arrival)
            float jv = 0.f; int jdst = -1;
            if (tid < DW2_SPLITS * a.jobs_per_wg) {
                const int sp = tid / a.jobs_per_wg, jl2 = tid - sp * a.jobs_per_wg;
                const int jb2 = ((tile << 3) | (sp << 1) | tower) * a.jobs_per_wg + jl2;
                if (jb2 < a.n_jobs) { jdst = a.jobs[jb2].dst; jv = __hip_atomic_load(a.grad + jdst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            }
            const bool has_n = gtile == 0 && tid == 511;                                     // workgroup 0 (tile 0 of the policy tower) wrote the row count
            const float nloc = has_n ? __hip_atomic_load(a.grad + a.tail_off + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll 1
            for (int k = 0; k < p.world; ++k) {
                const int r = (p.rank + k) % p.world;                                        // every rank starts at a different peer: the links share the load
                float* dst = p.slots[r] + ((unsigned long long)par * p.world + p.rank) * p.cap;
                // system-scope WRITE-THROUGH stores (sc0 sc1): nothing of the slots stays dirty in this XCD's L2, so the finisher needs no cache-wide
                // release -- a system-scope release fence here walks the whole L2 (64 of them per launch: 54.6 -> 72.4 us per train step, measured)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst + moff), "v"(t4) : "memory");
                if (strip_thread) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst + soff), "v"(u4) : "memory");
                if (jdst >= 0) __hip_atomic_store(dst + jdst, jv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (has_n) __hip_atomic_store(dst + a.tail_off + 5, nloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's slot writes are complete at system scope (what a release waits for, too)
            __syncthreads();
            if (tid == 0) flag[0] = (__hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == DW2_TILES - 1u) ? 1 : 0;
            __syncthreads();
            if (flag[0]) {                                              // the last finisher: every tile of this rank is out -> the flags (peer_push_kernel's tail)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
                if (tid < (unsigned)p.world) peer_st_sys(p.flags[tid] + ((size_t)par * PEER_MAX_WORLD + p.rank) * PEER_FLAG_STRIDE, sq);
                if (tid == 0) {
                    __hip_atomic_store(p.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.seq, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read by the NEXT kernel (adam_kernel<.., 2>): the kernel boundary orders it
                }
            }
        }
    }
    DW2_STAMP(7);
}

template <int KP0, int AP>
__global__ __launch_bounds__(DW2_THREADS) void weight_grad_assemble_kernel(Dw2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(Dw2Args)>();
    dw2_body<KP0, AP>(a, lds, (int)blockIdx.x);
}

template <int KP0, int AP>
__global__ __launch_bounds__(DW2_THREADS) void weight_grad_assemble_peer_kernel(Dw2Args a, PeerDev p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(Dw2Args) + sizeof(PeerDev)>();
    dw2_body<KP0, AP, true>(a, lds, (int)blockIdx.x, &p);
}
