// ppo_fused_ab.hpp -- train8_kernel and weight_grad_assemble_kernel of one train step as ONE launch (hidden [256,256], minibatches of up to
// 2048 rows: BASELINE configs[2]).  Reference: one Session::Run = one train step (ppo2/ppo2.hpp:430-468).
//
// Why: the two kernels hand 15 MB of activations / pre-activation gradients across a kernel boundary.  The launch that follows starts with an
// invalidate of its XCD's L2, so weight_grad_assemble_kernel fetches every workspace line from the memory side (2.2 k cycles request -> landing,
// L2 hit rate 70 %: the within-kernel reuse only) although train8_kernel's workgroups on XCD x wrote exactly the rows the tiles on XCD x read
// (TrainArgs::xcd_map 1).  Inside one launch nothing is invalidated: phase A's write-through stores leave their lines in the writing XCD's L2 and
// phase B's plain loads hit them there.
//
// Correctness does NOT depend on that placement: every store of phase A is write-through (sc1: at the memory side once the storing wave's vmcnt
// has drained), the grid-wide meeting below orders all of them before any load of phase B, and a workgroup of phase B that runs on ANOTHER XCD
// than the writer finds no copy of the line in its own L2 (the launch started with an invalidate and phase A reads no workspace) and takes it
// from the memory side.  Placement only decides the hit rate.
//
// Measured (profiles/r05_a_*): +0.4 us per train step against the two launches (+0.9 with a grid-wide meeting) -- inside one group the slowest first-phase
// workgroup is ~7 k cycles behind the median, and a workgroup waits for it just as the kernel boundary would.  Opt-in (PPO_HIP_FUSE_AB=1), not the default.
//
// The meeting: 256 workgroups of 512 threads, one per CU (both bodies need > 80 KB of LDS), all resident -- the host launches this form only when
// the grid fits the device.  No read-modify-write on a shared word (256 same-address agent-scope atomics serialise at ~100 cycles each: the first
// version of this kernel, one arrival counter, spent a median 26 k cycles per workgroup in the meeting -- tools/stamps_fused.py): every workgroup
// owns ONE word of a 256-word table, reads its value e at entry, writes e + 1 (write-through) when its phase A has drained, and 256 of its
// threads watch the 256 words (agent-scope loads) until all read e + 1.  The table is never reset: a replayed hipGraph needs no memset node.
// The wait is BOUNDED (~0.5 s): if the workgroups are not all resident -- another process's kernels hold CUs of this device and wait for CUs
// themselves -- the waiters raise the error word, go on (the step's results are then garbage) and the host call that synchronises next reports
// the error instead of leaving a hung device.  Data-parallel handles whose ranks share a device (the tests' N processes on one GPU) do not use
// this form at all (ppo_dist_init compares the ranks' PCI bus ids).
#pragma once
#include "ppo_dw2.hpp"
#include "ppo_train8.hpp"

#define FAB_THREADS 512
static_assert(T8_THREADS == FAB_THREADS && DW2_THREADS == FAB_THREADS, "both phases run 8 waves per workgroup");

template <int KP0, int AP>
struct FabL { static constexpr int FLOATS = T8L<KP0, AP>::TOTAL > Dw2L<KP0, AP>::LDS_FLOATS ? T8L<KP0, AP>::TOTAL : Dw2L<KP0, AP>::LDS_FLOATS; };

#define FAB_GRID 256
static_assert(FAB_GRID == DW2_GRID, "one table word per workgroup of the weight-gradient phase");
template <int KP0, int AP>
__global__ __launch_bounds__(FAB_THREADS) void train8_dw2_fused_kernel(NetDev net, TrainArgs ta, Dw2Args da, unsigned* meet /* [FAB_GRID] words + [1] error */, int n_rb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    warm_kernargs<sizeof(NetDev) + sizeof(TrainArgs) + sizeof(Dw2Args) + 16>();
    const unsigned lid = blockIdx.x;
    const unsigned epoch = __hip_atomic_load(meet + lid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;       // (only this workgroup writes its word)
    // phase A: forward + loss + backward of row tile lid % n_rb of tower lid / n_rb (the standalone launch's grid (n_rb, 2) in linear order)
    if ((int)lid < 2 * n_rb) train8_body<KP0, AP, true>(net, ta, lds, lid % (unsigned)n_rb, lid / (unsigned)n_rb, (unsigned)n_rb);
    // the meeting
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's write-through stores are complete at the memory side
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(meet + lid, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... of the workgroup's OWN group: the 32 workgroups lid' = (lid & 7) + 8 q wrote the rows (tower lid & 1, row split (lid >> 1) & 3) this workgroup's tile
    // reads (TrainArgs::xcd_map 1) -- by construction of the two index maps, whatever XCD they ran on.  The slot jobs need every workgroup: dw2_body waits for
    // those behind its chunk loop.  (A first version met grid-wide here: the value tower's workgroups, 6 k cycles early, waited for the policy tower's.)
    if (threadIdx.x < 32) {
        unsigned polls = 0;
        for (;;) {
            const unsigned w = __hip_atomic_load(meet + (lid & 7u) + 8u * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(w - epoch) >= 0 || threadIdx.x >= 32)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1u << 20)) { if (threadIdx.x == 0) __hip_atomic_store(meet + FAB_GRID, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
    // phase B: weight gradients + assembly, tile / split from the linear index as in the standalone launch
    const Dw2Meet mt{meet, epoch, meet + FAB_GRID, FAB_GRID};
    dw2_body<KP0, AP, false, true>(da, lds, (int)lid, nullptr, &mt);
}
