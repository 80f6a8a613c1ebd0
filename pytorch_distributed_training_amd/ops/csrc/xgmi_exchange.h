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
// The payload carries a checksum: next to its vector every rank stores one word that mixes the call's sequence
// number with a position-dependent hash of the values, and the receiver recomputes it over what it read.  The
// protocol orders payload and flag by "stores acknowledged, then flag" without any fence; should the fabric ever
// deliver a flag ahead of a payload word (or a slot be overwritten by a rank that is two calls ahead because a
// flag was lost), the mismatch is SEEN: the read is repeated a bounded number of times and then fails like a
// timeout (status 2) instead of feeding stale statistics into the step.
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
#define CAR_CTL_WORDS (3 + CAR_MAX_WORLD)   // LDS scratch of car_exchange_block: sequence, status, own checksum, per-rank checksums
#define CAR_CHK_OFFSET 64          // checksum words [2][CAR_MAX_WORLD] behind the flags [2][CAR_MAX_WORLD] (in 4-byte words)
#define CAR_READ_RETRIES 64

struct CarDev {
    int world, rank;
    int max4;                      // capacity of one slot in float4 units
    unsigned* seq;                 // device counter (local, ordinary memory)
    unsigned* status;              // non-zero after a timeout (local)
    unsigned* rereads;             // payload reads repeated because a checksum did not match yet (local, diagnostic)
    unsigned* status_host;         // the same, in pinned host memory (device-visible address)
    unsigned* flags;               // local flags [2][CAR_MAX_WORLD], checksums [2][CAR_MAX_WORLD] at word CAR_CHK_OFFSET
    float4* recv;                  // local receive buffer [2][world][max4]
    unsigned* peer_flags[CAR_MAX_WORLD];
    float4* peer_recv[CAR_MAX_WORLD];
    long long timeout_ticks;
    unsigned corrupt;              // test hook: XORed into the checksum this rank sends (0 in production)
};

struct CarHost {
    CarDev dev;
    void* local_base;              // fine-grained allocation: flags then receive buffer
    size_t local_bytes;
    void* opened[CAR_MAX_WORLD];   // IPC mappings to close
    unsigned* seq_status;          // ordinary device memory: [seq, status, rereads]
    unsigned* status_pinned;       // host address of the pinned status word
};

__device__ __forceinline__ unsigned car_mix(float v, int i) {
    return (__float_as_uint(v) ^ ((unsigned)i * 0x9E3779B1u)) * 0x85EBCA6Bu;
}

// XOR over the block of per-thread words into LDS word *dst (which the caller has zeroed before a barrier)
__device__ __forceinline__ void car_xor_into(unsigned h, unsigned* dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h ^= (unsigned)__shfl_xor((int)h, off, 64);
    if ((threadIdx.x & 63) == 0) atomicXor(dst, h);
}

// Block-wide exchange (any block size, every thread must call it): vals[0..n) in LDS is replaced
// by its sum over the ranks.  ``ctl`` is CAR_CTL_WORDS LDS words of scratch.  Returns false (uniformly)
// after a timeout or a checksum failure; vals is then all NaN and both status words are set.
__device__ __forceinline__ bool car_exchange_block(const CarDev& d, float* vals, int n, unsigned* ctl) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) {
        // device-scope accesses: the next call may run on another XCD, and its view of the counter
        // must not depend on the cache maintenance of the kernel boundary in between
        unsigned q = __hip_atomic_load(d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        __hip_atomic_store(d.seq, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = q;
        // After one failure the exchange is dead for good (the job is about to be ended by the
        // host): later calls must not each wait out the full timeout again.
        ctl[1] = __hip_atomic_load(d.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[2] = 0u;
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
    unsigned mine = 0u;
    for (int i = tid; i < n; i += nt) {
        const float v = vals[i];
        mine ^= car_mix(v, i);
        for (int r = 0; r < d.world; ++r) {
            float* dst = reinterpret_cast<float*>(d.peer_recv[r] + ((size_t)slot * d.world + d.rank) * d.max4) + i;
            __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    car_xor_into(mine, &ctl[2]);
    __syncthreads();
    if (tid < d.world)                                 // the checksum travels with the payload, ahead of the flag
        __hip_atomic_store(d.peer_flags[tid] + CAR_CHK_OFFSET + slot * CAR_MAX_WORLD + d.rank, ctl[2] ^ (seq * 0xC2B2AE35u) ^ d.corrupt,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

    // 4. sum in rank order (loads issued only after the flags were seen; they bypass the caches), checking every
    //    rank's payload against the checksum it sent; a mismatch is read again (bounded), then fails the exchange
    if (!ctl[1]) {
        for (int attempt = 0;; ++attempt) {
            if (tid < d.world) ctl[3 + tid] = 0u;
            __syncthreads();
            unsigned h[CAR_MAX_WORLD];
#pragma unroll
            for (int r = 0; r < CAR_MAX_WORLD; ++r) h[r] = 0u;
            for (int i = tid; i < n; i += nt) {
                const float* src = reinterpret_cast<const float*>(d.recv + (size_t)slot * d.world * d.max4) + i;
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < CAR_MAX_WORLD; ++r) {
                    if (r < d.world) {
                        const float v = __hip_atomic_load(src + (size_t)r * d.max4 * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        h[r] ^= car_mix(v, i);
                        acc = r == 0 ? v : acc + v;
                    }
                }
                vals[i] = acc;                         // (overwritten by the next attempt if this one fails: the inputs
            }                                          //  are not needed any more, every rank has pushed them)
#pragma unroll
            for (int r = 0; r < CAR_MAX_WORLD; ++r)
                if (r < d.world) car_xor_into(h[r], &ctl[3 + r]);
            __syncthreads();
            bool ok = true;
            for (int r = 0; r < d.world; ++r) {
                const unsigned want = __hip_atomic_load(d.flags + CAR_CHK_OFFSET + slot * CAR_MAX_WORLD + r, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_SYSTEM);
                ok = ok && ((ctl[3 + r] ^ (seq * 0xC2B2AE35u)) == want);
            }
            // (every thread read the same LDS words and the same flags page: the verdict is block-uniform up to a
            // checksum word still in flight, so it is agreed through LDS)
            __syncthreads();
            if (tid == 0) ctl[2] = ok ? 1u : 0u;
            __syncthreads();
            if (ctl[2]) break;
            if (tid == 0) atomicAdd(d.rereads, 1u);      // seen on real links = the flag does overtake payload words there
            if (attempt >= CAR_READ_RETRIES) {
                if (tid == 0) ctl[1] = 2u;
                __syncthreads();
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (ctl[1]) {                                  // block-uniform: every thread reads the same LDS word
        if (tid == 0) {
            atomicOr(d.status, ctl[1]);
            __hip_atomic_store(d.status_host, ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        for (int i = tid; i < n; i += nt) vals[i] = __builtin_nanf("");
        __syncthreads();
        return false;
    }
    __syncthreads();
    return true;
}
