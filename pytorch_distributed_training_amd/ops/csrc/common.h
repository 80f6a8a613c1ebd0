// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, blocks are multiples of 64 threads (256 unless noted);
//   * activations are NHWC bf16, accessed 16 bytes (8 channels) per lane;
//   * per-channel vectors and all accumulations are fp32;
//   * every launcher takes raw device pointers plus a hipStream_t and returns a hipError_t
//     cast to int (0 = success), so the library has a plain C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define PDT_API extern "C" __attribute__((visibility("default")))

enum { MASK_NONE = 0, MASK_TENSOR = 1, MASK_RECOMPUTE = 2 };

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st_bf16x8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
// Streaming load: the line is not wanted again soon (last use of a tensor in this pass), so it
// should not displace what the next kernel is about to read from L2 / the Infinity Cache.
__device__ __forceinline__ bf16x8 ld_stream_bf16x8(const bf16* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
}

// Gradient of a 3x3 / stride 2 / pad 1 max pooling at input element (b, h, w), channel vector cv:
// the sum of the gradients of the (at most 2 x 2) windows whose stored winner position (0..8,
// one byte per channel, packed 8 to a 64-bit word) is this element.  A gather, so the dense
// H x W gradient never has to exist in memory.
__device__ __forceinline__ void pool3x3s2_grad8(const bf16* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                int b, int h, int w, int cv, int cvec, int Ho, int Wo,
                                                float (&acc)[8]) {
    // windows (ho, wo) covering (h, w): ho*2-1 <= h <= ho*2+1, i.e. ho in {h>>1, (h+1)>>1}.  All four
    // candidates are loaded unconditionally (a candidate that does not exist re-reads the first one
    // and is given a winner code no window can hold), so the loads go out back to back.
    const int ho0 = h >> 1, wo0 = w >> 1;
    const int ho1 = (h + 1) >> 1, wo1 = (w + 1) >> 1;
    const bool vh = ho1 != ho0 && ho1 < Ho, vw = wo1 != wo0 && wo1 < Wo;
    const int hs[2] = {ho0, vh ? ho1 : ho0}, ws[2] = {wo0, vw ? wo1 : wo0};
    unsigned long long packed[4];
    bf16x8 gv[4];
    unsigned want[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ho = hs[q >> 1], wo = ws[q & 1];
        const bool valid = ((q >> 1) == 0 || vh) && ((q & 1) == 0 || vw);
        want[q] = valid ? (unsigned)((h - (ho * 2 - 1)) * 3 + (w - (wo * 2 - 1))) : 0xffu;
        const long long o = (((long long)b * Ho + ho) * Wo + wo) * cvec + cv;
        packed[q] = reinterpret_cast<const unsigned long long*>(idx)[o];
        gv[q] = *reinterpret_cast<const bf16x8*>(dout + o * 8);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            a += ((unsigned)((packed[q] >> (8 * k)) & 0xff) == want[q]) ? (float)gv[q][k] : 0.0f;
        acc[k] = a;
    }
}

// Streaming store: a large result that the next kernel cannot find in cache anyway goes out
// without claiming cache lines.
__device__ __forceinline__ void st_stream_bf16x8(bf16* p, bf16x8 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
}

__device__ __forceinline__ bf16x8 zero_bf16x8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

// Buffer (SRD-based) 16-byte load: the descriptor carries the tensor size, and a byte offset at
// or beyond it makes the hardware return zeros -- a branch-free, select-free way to zero-fill
// padding / out-of-tile lanes.  Branch-free loads also let the compiler count outstanding
// loads (s_waitcnt vmcnt(N)) instead of draining the queue at every conditional.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define PDT_OOB_OFFSET 0x7ffffff0u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_ld_bf16x8(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    union { u32x4 u; bf16x8 v; } cvt;
    cvt.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return cvt.v;
}

// Sum over the 64 lanes of a wave (all lanes get the result).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

static inline int pdt_launch_status() { return (int)hipGetLastError(); }

// Per-channel partial sums (BatchNorm statistics and their backward counterparts) are added with float atomics into
// one of ``nrep`` replica rows of an accumulator [nrep][2][C], picked by blockIdx % nrep.  With the default 16 rows
// several workgroups share a row and their adds arrive in any order: fast, and the low bits differ from run to run.
// Deterministic mode (pdt_set_deterministic) uses PDT_DET_NREP rows and caps every accumulating grid at that many
// workgroups: a row then has ONE writer, and the fold adds the rows in index order -- bit-identical results.
#define PDT_NREP 16
#define PDT_DET_NREP 1024
extern int g_pdt_deterministic;
static inline int pdt_nrep() { return g_pdt_deterministic ? PDT_DET_NREP : PDT_NREP; }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
