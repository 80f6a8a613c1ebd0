// One-shot all-reduce for small fp32 vectors between the GPUs of one node (xGMI peer access).
//
// SyncBatchNorm puts two tiny collectives (2C floats, at most 16 KB) on the critical path of
// every layer: what they cost is latency, not bandwidth.  A ring/tree collective launched
// through a communication library pays a kernel launch on another stream plus several
// synchronisation hops; here the exchange is ONE ordinary kernel on the compute stream:
//
//   1. every rank stores its vector into slot [parity][rank] of EVERY peer's receive buffer
//      (cache-bypassing stores over xGMI into IPC-mapped fine-grained memory);
//   2. once those stores are acknowledged, one flag store per peer (the call's sequence number);
//   3. every rank polls its own flags (local memory) until all peers' numbers have arrived;
//   4. it sums the world slots in rank order -- the same order on every rank, so the result is
//      bit-identical everywhere.
//
// Two parities alternate: a rank can only be one call ahead of the slowest peer (it needs that
// peer's flag of call k to finish call k), so slot k%2 is never rewritten while still being read.
// The sequence number lives in device memory and is advanced by the kernel itself, which makes
// the launch replayable from a captured HIP graph.
//
// Every wait is bounded (wall-clock timeout): a missing peer never leaves a kernel spinning.  A
// timed-out exchange sets a status word in device memory AND in a pinned host word (so the host
// can poll it every step without synchronising) and returns NaN for every element: whatever is
// computed from it is poisoned visibly instead of silently wrong, and the trainer aborts the
// job when it sees the status.  The host side verifies the path against the library collective
// before using it.
#pragma once
#include "common.h"

#define CAR_MAX_WORLD 8
#define CAR_MAX_FLOATS 4096        // vector length one block exchanges through LDS (16 KB)

struct CarDev {
    int world, rank;
    int max4;                      // capacity of one slot in float4 units
    unsigned* seq;                 // device counter (local, ordinary memory)
    unsigned* status;              // non-zero after a timeout (local)
    unsigned* status_host;         // the same, in pinned host memory (device-visible address)
    unsigned* flags;               // local flags [2][CAR_MAX_WORLD]
    float4* recv;                  // local receive buffer [2][world][max4]
    unsigned* peer_flags[CAR_MAX_WORLD];
    float4* peer_recv[CAR_MAX_WORLD];
    long long timeout_ticks;
};

struct CarHost {
    CarDev dev;
    void* local_base;              // fine-grained allocation: flags then receive buffer
    size_t local_bytes;
    void* opened[CAR_MAX_WORLD];   // IPC mappings to close
    unsigned* seq_status;          // ordinary device memory: [seq, status]
    unsigned* status_pinned;       // host address of the pinned status word
};

// Block-wide exchange (any block size, every thread must call it): vals[0..n) in LDS is replaced
// by its sum over the ranks.  ``ctl`` is two LDS words of scratch.  Returns false (uniformly)
// after a timeout; vals is then all NaN and both status words are set.
__device__ __forceinline__ bool car_exchange_block(const CarDev& d, float* vals, int n, unsigned* ctl) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) {
        // device-scope accesses: the next call may run on another XCD, and its view of the counter
        // must not depend on the cache maintenance of the kernel boundary in between
        unsigned q = __hip_atomic_load(d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        __hip_atomic_store(d.seq, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = q;
        // After one timeout the exchange is dead for good (the job is about to be ended by the
        // host): later calls must not each wait out the full timeout again.
        ctl[1] = __hip_atomic_load(d.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned seq = ctl[0];
    const int slot = (int)(seq & 1u);
    if (ctl[1]) {
        for (int i = tid; i < n; i += nt) vals[i] = __builtin_nanf("");
        __syncthreads();
        return false;
    }

    // 1. push my vector into every rank's slot [slot][my rank].  The exchange buffers are only
    //    ever touched with system-scope (cache-bypassing, write-through) accesses, so no cache
    //    write-back or invalidate is needed around them: a release/acquire FENCE at system scope
    //    would flush and drop this XCD's whole L2, which the neighbouring conv kernels pay for.
    for (int i = tid; i < n; i += nt) {
        const float v = vals[i];
        for (int r = 0; r < d.world; ++r) {
            float* dst = reinterpret_cast<float*>(d.peer_recv[r] + ((size_t)slot * d.world + d.rank) * d.max4) + i;
            __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have been acknowledged ...
    __syncthreads();                                   // ... and so have every other thread's

    // 2./3. announce, then wait for everybody (thread r handles peer r)
    if (tid < d.world) {
        __hip_atomic_store(d.peer_flags[tid] + slot * CAR_MAX_WORLD + d.rank, seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(d.flags + slot * CAR_MAX_WORLD + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) !=
               seq) {
            if (wall_clock64() - t0 > d.timeout_ticks) {
                ctl[1] = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (ctl[1]) {                                  // block-uniform: every thread reads the same LDS word
        if (tid == 0) {
            atomicOr(d.status, 1u);
            __hip_atomic_store(d.status_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int i = tid; i < n; i += nt) vals[i] = __builtin_nanf("");
        __syncthreads();
        return false;
    }

    // 4. sum in rank order (loads issued only after the flags were seen; they bypass the caches)
    for (int i = tid; i < n; i += nt) {
        const float* src = reinterpret_cast<const float*>(d.recv + (size_t)slot * d.world * d.max4) + i;
        float acc = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < d.world; ++r)
            acc += __hip_atomic_load(src + (size_t)r * d.max4 * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        vals[i] = acc;
    }
    __syncthreads();
    return true;
}
