// ppo_peer.hpp -- one-shot all-reduce over peer-mapped buffers (xGMI), for the small payloads of the data-parallel PPO
// step (the padded gradient + loss sums of one minibatch, 50 KB at [64,64], 0.6 MB at [256,256];
// the advantage sums of an epoch; the running-statistics table of an env step).
//
// A ring all-reduce of such a payload is latency-bound (2(W-1) hops); xGMI is point-to-point with a link to every peer, so
// the whole exchange can be ONE hop: every rank PUSHES its vector into slot [rank] of every peer's gather region (posted
// writes over all links at once), raises a flag at every peer, and every rank then sums the W slots of ITS OWN region in
// rank order -- the same fixed summation order on every rank, so the replicas stay bit-identical (no reduction tree whose
// shape depends on the rank).  Two plain kernels per collective, capturable into the update's hipGraph like any other launch.
//
// Protocol (per rank: `seq` = number of collectives issued so far, local; all ranks issue the same sequence):
//   push: s = seq + 1, parity p = s & 1.  Every workgroup copies its chunk of the source into slots[r][p][rank] for every r,
//         then a SYSTEM-scope release fence, then one arrival on a local counter; the last workgroup to arrive stores s into
//         flags[r][p][rank] at every r (system scope) and publishes seq = s.
//   sum:  s = seq.  Every workgroup waits until flags[mine][p][r] == s for all r (bounded spin: a dead peer sets `err`
//         instead of hanging the GPU), takes a SYSTEM-scope acquire fence, and adds the W slots of its chunk in rank order.
// Why two parities are enough: a peer can only push collective s+2 after its own sum of s+1 has finished, which needs MY
// push of s+1, which is stream-ordered after MY sum of s -- so nobody overwrites a slot that is still being read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ unsigned peer_ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void peer_st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

#define PEER_MAX_WORLD 8
#define PEER_FLAG_STRIDE 16            // unsigned per (parity, source) flag: one 64-byte line each

struct PeerDev {
    float* slots[PEER_MAX_WORLD];      // base of every rank's gather region [2][world][cap] (mine: local; others: IPC-mapped)
    unsigned* flags[PEER_MAX_WORLD];   // base of every rank's flag block [2][PEER_MAX_WORLD][PEER_FLAG_STRIDE]
    unsigned* seq;                     // local: collectives issued
    unsigned* arrive;                  // local: workgroup arrival counter of the push kernel
    unsigned* err;                     // local: set when a wait timed out
    unsigned long long cap;            // floats per slot
    int world, rank;
    unsigned spin_limit;               // wait iterations before giving up
    // running-statistics table of an env step (norm_batch_kernel / norm_finalize_kernel exchange it themselves, no launch of its own):
    // every region carries a second, small gather area [2 parities][world][scap] behind the slots and flags [2][2 jobs][PEER_MAX_WORLD]
    // (one 64-byte line each) behind the collective's flags; the sequence numbers of the two jobs (observations, rewards) are local
    float* sslots[PEER_MAX_WORLD];
    unsigned* sseq;                    // local: [2] env steps exchanged so far per job
    int scap;                          // floats per statistics slot (0: the exchange is not available)
};
#define PEER_SFLAG_OFF (2 * PEER_MAX_WORLD * PEER_FLAG_STRIDE)       // first statistics flag (unsigned index into a region's flag block)
__device__ __forceinline__ unsigned* peer_sflag(const PeerDev& p, int at_rank, unsigned par, int job, int from_rank) {
    return p.flags[at_rank] + PEER_SFLAG_OFF + (((size_t)par * 2 + job) * PEER_MAX_WORLD + from_rank) * PEER_FLAG_STRIDE;
}

// the last workgroup of a statistics job: `n` floats at `src` (this rank's batch moments) into slot [rank] of EVERY rank's statistics
// area, then the flags.  Whole workgroup.
__device__ __forceinline__ void peer_stats_publish(const PeerDev& p, int job, const float* src, int off, int n) {
    __shared__ unsigned s_seq;
    if (threadIdx.x == 0) s_seq = __hip_atomic_load(p.sseq + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __syncthreads();
    const unsigned s = s_seq, par = s & 1u;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (the strip form: other workgroups of this launch wrote parts of it, write-through)
        for (int k = 0; k < p.world; ++k) {
            const int r = (p.rank + k) % p.world;
            p.sslots[r][((size_t)par * p.world + p.rank) * p.scap + off + i] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                // system scope: the slot writes are out before any flag
    __syncthreads();
    if (threadIdx.x < (unsigned)p.world) peer_st_sys(peer_sflag(p, threadIdx.x, par, job, p.rank), s);
    if (threadIdx.x == 0) __hip_atomic_store(p.sseq + job, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read by the NEXT kernel (norm_finalize_kernel)
}

// norm_finalize_kernel's side: wait for every rank's flag of this job, then copy the `n` floats at `off` of every rank's slot into the
// local table `dst` ([world][stride], the layout the all-reduced table had).  Whole workgroup; returns false when a peer never answered.
__device__ __forceinline__ bool peer_stats_collect(const PeerDev& p, int job, float* dst, int stride, int off, int n) {
    __shared__ unsigned s_bad;
    const unsigned s = __hip_atomic_load(p.sseq + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), par = s & 1u;
    if (threadIdx.x == 0) s_bad = 0u;
    __syncthreads();
    if (threadIdx.x < (unsigned)p.world) {
        const unsigned* f = peer_sflag(p, p.rank, par, job, threadIdx.x);
        unsigned it = 0;
        while (peer_ld_sys(f) != s) {
            __builtin_amdgcn_s_sleep(2);
            if (++it > p.spin_limit) { __hip_atomic_store(p.err, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_bad = 1u; break; }
        }
    }
    if (threadIdx.x < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    if (s_bad) return false;
    const float* mine = p.sslots[p.rank] + (size_t)par * p.world * p.scap;
    for (int i = threadIdx.x; i < n * p.world; i += blockDim.x) {
        const int r = i / n, e = i - r * n;
        dst[(size_t)r * stride + off + e] = __hip_atomic_load(mine + (size_t)r * p.scap + off + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    return true;
}


#define PEER_THREADS 256
// V = float4 per thread of the push kernel = 1024-float pieces per workgroup: fewer, larger workgroups pay fewer system-scope
// fences per exchange (0.6 MB: 62 us per train step at V = 4 against 69 at V = 1, one rank), small payloads want the
// parallelism (52 KB: 30 against 35 us).  The host picks per call; `cap` is a multiple of the largest chunk.
#define PEER_CHUNK_MAX 4096

// grid = ceil(count / (1024 V)); src must be 16-byte aligned, slots are
template <int PEER_V>
__global__ __launch_bounds__(PEER_THREADS) void peer_push_kernel(PeerDev p, const float* __restrict__ src, unsigned long long count) {
    __shared__ unsigned s_seq, s_last;
    if (threadIdx.x == 0) s_seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __syncthreads();
    constexpr int PEER_CHUNK = 1024 * PEER_V;
    const unsigned s = s_seq, par = s & 1u;
    float4 v[PEER_V];
#pragma unroll
    for (int q = 0; q < PEER_V; ++q) {
        const unsigned long long i0 = (unsigned long long)blockIdx.x * PEER_CHUNK + 1024ull * q + 4ull * threadIdx.x;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + 3 < count) v[q] = *reinterpret_cast<const float4*>(src + i0);
        else {
            if (i0 < count) v[q].x = src[i0];
            if (i0 + 1 < count) v[q].y = src[i0 + 1];
            if (i0 + 2 < count) v[q].z = src[i0 + 2];
        }
    }
#pragma unroll 1
    for (int k = 0; k < p.world; ++k) {
        const int r = (p.rank + k) % p.world;                   // every rank starts at a different peer: the links share the load
        float* dst = p.slots[r] + ((unsigned long long)par * p.world + p.rank) * p.cap;      // cap is a multiple of PEER_CHUNK
#pragma unroll
        for (int q = 0; q < PEER_V; ++q) {
            const unsigned long long i0 = (unsigned long long)blockIdx.x * PEER_CHUNK + 1024ull * q + 4ull * threadIdx.x;
            if (i0 < count) *reinterpret_cast<float4*>(dst + i0) = v[q];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                // system-scope RELEASE only (write-back + wait): __threadfence_system() would also invalidate the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        // relaxed: every workgroup's slot writes were written back and waited for by the release fence above, BEFORE its
        // arrival; the last arriver reads nothing of theirs, it only must not raise the flags earlier (an acq_rel atomic here
        // would be one more L2 write-back + invalidate per workgroup)
        const unsigned old = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
        // The other workgroups' slot writes reach this point through their release fences and the arrival counter: one acquire on
        // the counter's side and one system-scope release in front of the flags make the chain a synchronises-with edge by the
        // memory model, not by what the caches happen to do (once per collective, in one workgroup: the L2 is clean already).
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
        if (threadIdx.x < (unsigned)p.world)
            peer_st_sys(p.flags[threadIdx.x] + ((size_t)par * PEER_MAX_WORLD + p.rank) * PEER_FLAG_STRIDE, s);
        if (threadIdx.x == 0) {
            __hip_atomic_store(p.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.seq, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read by the NEXT kernel (sum): the kernel boundary orders it
        }
    }
}

// grid = ceil(count / (1024 V)); dst[i] = sum_r slot[r][i] in rank order.  Thread t owns elements t, t+256, t+512, ... of
// the workgroup's chunk, so every 256-element chunk is laid over the 256 threads exactly as in grad_reduce_kernel /
// grad_sumsq_kernel and its sum of squares is formed by the same tree (bit-identical to what those kernels would write).
// sumsq (may be null): one partial per 256-element chunk; chunks at or beyond `sumsq_chunks` are not written (the loss sums
// ride behind the gradient in the same payload and must not enter the norm).
template <int PEER_V>
__global__ __launch_bounds__(PEER_THREADS) void peer_sum_kernel(PeerDev p, float* __restrict__ dst, unsigned long long count, float* __restrict__ sumsq,
                                                               unsigned sumsq_chunks) {
    constexpr int PEER_CHUNK = 1024 * PEER_V;
    __shared__ unsigned s_seq;
    __shared__ float s_part[PEER_CHUNK / 256][PEER_THREADS / 64];
    if (threadIdx.x == 0) s_seq = __hip_atomic_load(p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned s = s_seq, par = s & 1u;
    if (threadIdx.x < (unsigned)p.world) {
        const unsigned* f = p.flags[p.rank] + ((size_t)par * PEER_MAX_WORLD + threadIdx.x) * PEER_FLAG_STRIDE;
        unsigned it = 0;
        while (peer_ld_sys(f) != s) {
            __builtin_amdgcn_s_sleep(2);
            if (++it > p.spin_limit) { __hip_atomic_store(p.err, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    // ONE system-scope acquire per workgroup by the polling wave, after its flags matched and in front of the barrier the other
    // waves load behind (guide: one relaxed poll, one acquire, then the loads).  The slots are read with SYSTEM-scope loads
    // (sc0 sc1: served by memory, never by a stale cache line) on top of that.
    if (threadIdx.x < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    const float* mine = p.slots[p.rank] + (unsigned long long)par * p.world * p.cap;
    // every load of the workgroup is issued before the first addition (one round trip, not one per rank); the additions run
    // in rank order, skipped -- not fed with zeros -- for ranks that do not exist (-0 + 0 would flip a sign bit)
    float v[PEER_CHUNK / 256][PEER_MAX_WORLD];
#pragma unroll
    for (int k = 0; k < PEER_CHUNK / 256; ++k) {
        const unsigned long long i = (unsigned long long)blockIdx.x * PEER_CHUNK + 256 * k + threadIdx.x;
#pragma unroll
        for (int r = 0; r < PEER_MAX_WORLD; ++r)
            v[k][r] = (r < p.world && i < count) ? __hip_atomic_load(mine + (unsigned long long)r * p.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
    }
    float acc[PEER_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < PEER_CHUNK / 256; ++k) {
        const unsigned long long i = (unsigned long long)blockIdx.x * PEER_CHUNK + 256 * k + threadIdx.x;
        float a = v[k][0];
#pragma unroll
        for (int r = 1; r < PEER_MAX_WORLD; ++r) if (r < p.world) a += v[k][r];
        if (i < count) dst[i] = a;
        acc[k] = a;
    }
    if (sumsq) {
#pragma unroll
        for (int k = 0; k < PEER_CHUNK / 256; ++k) {
            float q = acc[k] * acc[k];
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if ((threadIdx.x & 63) == 0) s_part[k][threadIdx.x >> 6] = q;
        }
        __syncthreads();
        const unsigned c = blockIdx.x * (PEER_CHUNK / 256) + threadIdx.x;
        if (threadIdx.x < PEER_CHUNK / 256 && c < sumsq_chunks)
            sumsq[c] = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
    }
}
