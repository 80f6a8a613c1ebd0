// Convolution forward / data gradient for gfx950 (MI355X) as a streaming implicit GEMM fed by LDS-DMA:
//
//   out[row(m)][n] = sum_{t < T} sum_{c < Ci} in[pix(m, t)][c] * wgt[n][widx(t)*Ci + c]     (+ epilogue)
//
// bf16 in, fp32 accumulate on the matrix cores (v_mfma_f32_16x16x32_bf16), bf16 out.  Same geometry
// contract as conv_igemm.hip (GEMM row = a point of a per-image Hg x Wg grid, a tap table with input
// offsets, output scatter with stride OS for the parity classes of a strided data gradient); it needs
// Ci and N to be multiples of 64, which covers every ResNet convolution but the stem and the classifier.
//
// Most of these layers move far more bytes than they multiply (1x1 convolutions with K = 64..512
// against 4-13 activation-sized tensors read and written by the epilogue), so the kernel is built
// around keeping bytes in flight rather than around the MFMA loop:
//
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no
//     ds_write pass.  The DMA writes 64 lanes x 16 B linearly (8 rows of a 128-byte K slice), so the
//     bank-conflict swizzle of the fragment reads is applied to the per-lane SOURCE address; taps
//     that fall into the padding read zeros through the buffer descriptor's range check;
//   * an R-stage LDS ring per workgroup (two stages of K = 64 for a 128x128 tile: 64 KB, two workgroups
//     per CU).  Workgroups are persistent and the ring runs ACROSS tiles: the first K slices of a
//     workgroup's next tile are requested before the epilogue of the current one, so with one or two
//     K steps per tile (K = 64, 128) the loads of tile t+1 fly under the epilogue of tile t; waits are
//     counted (s_waitcnt vmcnt(N)) and the barrier is a bare s_barrier, so younger DMA groups and
//     epilogue stores stay in flight across it;
//   * the epilogue never touches LDS and needs no barrier: the weight rows are assigned to the MFMA operand rows so
//     that a lane ends up with 16 CONSECUTIVE channels of one pixel = 32 contiguous bytes (see g1_swz_w), so every
//     epilogue operand (residual-gradient addend, packed ReLU mask, the BatchNorm input whose backward sums are
//     being fused) is read with 16-byte accesses straight into registers; for the result, lanes l and l ^ 8 trade
//     one 16-byte half (one DPP rotation) so that a store instruction writes whole 128-byte lines;
//   * all epilogue loads of a wave's 64 x 64 sub-tile are issued before the first one is used.
//
// Epilogue options (the same contract as conv_igemm.hip): per-channel BatchNorm statistics of the
// stored values (forward), "out = (out + addend) * mask" with the addend at full or half resolution
// and the mask as one bit per element, BatchNorm-backward sums of the layer that consumes the result.
//
// Second operand (1x1 / stride 1 geometry only): after the taps the reduction continues over Ci2 channels of
// a second tensor whose row m is GEMM row m, against weight columns [T*Ci, T*Ci + Ci2); and the accumulators
// may start from a per-output-channel constant instead of zero.  Together they evaluate
//   out = in . W1^T + in2 . W2^T + bias
// in one pass, which is how a BatchNorm backward is folded into the data gradient of the 1x1 convolution
// in front of it (bn_fold.hip): no normalised-gradient tensor is ever written.  The same instantiations can clamp
// the result at zero and write its packed sign mask: with BatchNorm's scale folded into the weights and its shift
// as start value, a block tail  relu(bn(conv(a)) [+ bn(conv_shortcut(x))] [+ x])  is this kernel's epilogue.
#include "common.h"
#include <type_traits>

#define DMA_MAX_TAPS 52
#define DMA_MAX_BIAS 512

struct ConvDmaParams {
    const bf16* in;
    const bf16* wgt;
    bf16* out;
    const bf16* addend;
    const unsigned char* mask_bits;
    float* stat_acc;
    const float* stat_shift;           // forward statistics: per-channel centre (may be null -> 0)
    const bf16* bstat_y;               // backward sums of the consumer (see conv_igemm.hip)
    const float* bstat_mean;
    const float* bstat_scale;
    const float* bstat_shift;
    int M, N;
    unsigned long long in_bytes;       // the gathered operand may exceed a descriptor's 2 GiB reach: rebased per tile
    unsigned wgt_bytes;
    int Hi, Wi, pix_stride, Ci;        // input image, elements between two pixels, channels per tap
    int Hg, Wg, IS;                    // GEMM row (b, y, x) on the Hg x Wg grid; tap t reads pixel (y*IS + offy, x*IS + offx)
    int Ho, Wo, OS, oy0, ox0;          // ... and is written to output pixel (y*OS + oy0, x*OS + ox0)
    int wrow, T;                       // weight row length (elements), taps
    int simple_out;                    // output row == GEMM row
    unsigned magic_w, magic_h;         // 2^20 / Wg + 1, 2^20 / Hg + 1: exact multiply-shift division of small offsets
    unsigned long long addend_half_bytes;   // size of the half-resolution addend tensor (addend_half)
    int addend_half;                   // addend is [B][Ho/2][Wo/2][N], present at even output (y, x) only
    int nrep;                          // replica rows of the statistics accumulator (see common.h)
    const bf16* in2;                   // second operand [M][pix_stride2] (may be null), reduced after the taps
    const float* bias;                 // [N] start value of the accumulators (may be null)
    unsigned char* bits_out;           // packed [out > 0] of the stored result, one bit per element (may be null)
    int relu_out;                      // clamp the result at zero before it is stored
    int sums_only;                     // stat_acc receives plain per-channel sums of the result (row 1 is not defined)
    unsigned long long in2_bytes;
    int Ci2, pix_stride2;
    int taps[DMA_MAX_TAPS];            // offy (int8) | offx (int8) << 8 | widx (uint16) << 16
#ifdef PDT_DMA_DIAG
    int diag;                          // ablation build only (bench/diag): 1 no result stores, 2 weight DMA for a block's
                                       // first tile only, 4 no MFMA, 8 operand DMA for the first tile only, 16 no epilogue loads, 32 no epilogue
#endif
};
#ifdef PDT_DMA_DIAG
#define DIAG(bit) (p.diag & (bit))
#else
#define DIAG(bit) 0
#endif

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    union { bf16x2 v; unsigned u; } c;
    c.v[0] = f2bf(lo);
    c.v[1] = f2bf(hi);
    return c.u;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Swizzle of the 16-byte chunk index inside an LDS row, chosen so that the ds_read_b128 fragment reads of the
// 16x16x32 operand layout (lane l: row l & 15, chunk l >> 4) are bank-conflict free for the instruction's
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: 128-byte rows (BK = 64) and 64-byte rows (BK = 32).
template <int BK>
__device__ __forceinline__ int g1_swz(int row) {
    return BK == 64 ? ((row >> 1) & 7) : ((row >> 1) & 2);
}

// Which of a wave's 64 output channels an MFMA produces where is free: the lane that supplies operand row a of the
// 16-row block j may read ANY weight row.  Row 16*(a >> 2) + 4*j + (a & 3) is the one that leaves lane group fq with
// the 16 CONSECUTIVE channels 16*fq .. 16*fq + 15 of its pixel (register r of block j: channel 16*fq + 4*j + r), so
// the epilogue needs no exchange between lanes to move 32 contiguous bytes per lane.  The weight tile's chunk swizzle
// follows the fragment position (4*(row >> 4) + (row & 3) inside the 64-row block), not the row, to stay conflict-free.
template <int BK>
__device__ __forceinline__ int g1_swz_w(int row) {
    return g1_swz<BK>(((row >> 4) & 3) * 4 + (row & 3));
}

// row_ror:8 inside each row of 16 lanes: lane l receives the value of lane l ^ 8
__device__ __forceinline__ unsigned dpp_xor8(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false);   // (every lane has a source)
}

// max(x, 0) of both bf16 halves: as 16-bit integers, the negative values are exactly the patterns with the sign bit set
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_relu(unsigned v) {
    union { unsigned u; s16x2_t h; } c;
    c.u = v;
    c.h = __builtin_elementwise_max(c.h, s16x2_t{0, 0});
    return c.u;
}

// FOLD: the launch has a second operand and/or accumulator start values (its own instantiations, so the plain
// kernels carry neither the extra scalar registers nor the extra LDS)
// GEN: the generic epilogue (output rows scattered over a strided grid, or the addend at half resolution); the launches
// whose output row is the GEMM row take the lean one (two instantiations: both epilogues in one kernel cost 60 registers)
// PF: the epilogue's operands (addend, packed mask) of tile t+1 are requested while tile t is multiplied and stay in flight
// under its epilogue (a second register set; requested by inline assembly so that the compiler's own waits do not drain
// them, complete at the next tile's first wait).  The one- and two-K-step launches otherwise wait for these loads with
// nothing else to do: per tile their latency was exposed in full.  Only for the lean epilogue without the backward-sums
// operands, and the statistics of such a launch are plain per-channel sums (no centre, no squares): that is what the
// register budget allows next to 64 accumulators.
template <int BN, int WM, int WN, int BK, int R, bool BSTAT, bool FOLD = false, bool GEN = false, bool PF = false>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(const ConvDmaParams p) {
    static_assert(!PF || (!BSTAT && !GEN && R == 2), "the prefetching variant: lean epilogue, no backward-sums operands, ring of two");
    constexpr int BM = 128, RB = BK * 2;
    constexpr int CH = BK / 8;                              // 16-byte chunks per LDS row
    constexpr int RPI = 64 / CH;                            // LDS rows one DMA instruction fills (1 KiB)
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    static_assert(NI == 4, "a wave owns 64 output channels (4 x 4 lane/register transpose)");
    static_assert(WM * WN == 4, "four waves");
    static_assert(R >= 2, "at least two stages");
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_DMA = BM / RPI / 4, B_DMA = BN / RPI / 4;   // DMA instructions per wave and stage
    constexpr int G = A_DMA + B_DMA;
    constexpr int KS = BK / 32;
    static_assert(A_DMA >= 1 && B_DMA >= 1, "every wave takes part in filling both operands");

    // One array for everything (a second __shared__ object next to the DMA ring makes the compiler drain the ring
    // before every fragment read), and not a byte more than this: two workgroups of 66 KB share a CU, two of 68 KB
    // do not (measured: the kernel then runs at half residency, whatever the occupancy query says).  The
    // accumulator start values therefore live in the statistics hand-over area and are staged again after each
    // (rare) statistics flush.
    static_assert(DMA_MAX_BIAS * 4 <= 4 * 2 * 64 * 4, "accumulator start values share the statistics hand-over area");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[R * STAGE + 4 * 2 * 64 * 4];
    float* red = reinterpret_cast<float*>(smem + R * STAGE);       // [wave][2][64] statistics hand-over
    float* bias_lds = red;                                         // [N] accumulator start values between flushes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    // fast epilogue: byte offset of this lane's 32 bytes inside the wave tile (loads), and of its 16-byte piece of the
    // pixel of lane frow & 7 after the exchange with lane frow ^ 8 (stores: (frow & 8) is the piece's 8 elements)
    const unsigned ep_ld = (unsigned)(frow * p.N + fq * 16) * 2u;
    const unsigned ep_st = (unsigned)((frow & 7) * p.N + fq * 16 + (frow & 8)) * 2u;

    const int n_tiles_n = p.N / BN;
    const int n_tiles_m = (p.M + BM - 1) / BM;
    const int nwg = n_tiles_m * n_tiles_n;
    const int kc = p.Ci / BK;                 // K steps per tap
    const int kc2 = FOLD ? p.Ci2 / BK : 0;    // K steps of the second operand (0: none)
    const int kdiff = kc2 - kc;
    const int nphase = p.T + (kc2 > 0 ? 1 : 0);
    const int nk = p.T * kc + kc2;
    const int HgWg = p.Hg * p.Wg;

    // The gathered operand is addressed with 32-bit offsets from a base that moves with the tile: the first image the
    // tile's rows touch (second operand / linear launches: the tile's first pixel).  A tile reaches a few images at
    // most, so the offsets stay small whatever the batch, and one launch covers any tensor size.  Base and size are
    // workgroup-uniform state of the DMA cursor (scalar registers); the descriptor is assembled where it is used.
    unsigned long long a_ptr = (unsigned long long)p.in, a2_ptr = FOLD ? (unsigned long long)p.in2 : 0ull;
    unsigned a_nrec = 0u, a2_nrec = 0u;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wgt, p.wgt_bytes);

    auto tile_origin = [&](int v, int& m0, int& n0) {
        // XCD-aware bijective remap: ids v, v+8, ... run on one XCD; give each XCD a contiguous range of tiles
        int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        m0 = (wgid / n_tiles_n) * BM;
        n0 = (wgid % n_tiles_n) * BN;
    };

    // ---- DMA side: a cursor (tile, K step) that runs R-1 steps ahead of the compute cursor, across tiles.
    // Instruction j of this wave fills LDS rows (wave*DMA + j)*RPI .. of the operand image; lane l lands at
    // row +l/CH, physical chunk l%CH, which must hold logical chunk (l%CH) ^ swz(row).
    const int d_row = lane / CH, d_chunk = lane % CH;
    int a_base[A_DMA], a_y[A_DMA], a_x[A_DMA];                 // image base pixel and grid coordinates * IS of this lane's rows
    unsigned a_sw[A_DMA], voff_b[B_DMA];
    int iv = blockIdx.x, it = 0, ic = 0, ibuf = 0;            // gridDim.x <= nwg (launcher); cursor = (tile, tap, channel slice)
    int i_m0 = 0, i_n0 = 0;                                   // origin of the cursor's tile
    auto setup_issue_tile = [&]() {
        int m0, n0;
        tile_origin(iv, m0, n0);
        i_m0 = m0;
        i_n0 = n0;
        const unsigned b0 = FOLD ? 0u : (unsigned)m0 / (unsigned)HgWg;            // first image of the tile
        {
            const unsigned long long pix0 = FOLD ? (unsigned long long)m0 : (unsigned long long)b0 * (unsigned)(p.Hi * p.Wi);
            const unsigned long long off0 = pix0 * (unsigned)p.pix_stride * 2ull, left = p.in_bytes - off0;
            a_ptr = (unsigned long long)p.in + off0;
            a_nrec = left < 0x7fff0000ull ? (unsigned)left : 0x7fff0000u;           // (PDT_OOB_OFFSET stays beyond it)
            if (FOLD) {
                const unsigned long long off2 = pix0 * (unsigned)p.pix_stride2 * 2ull, left2 = p.in2_bytes - off2;
                a2_ptr = (unsigned long long)p.in2 + off2;
                a2_nrec = left2 < 0x7fff0000ull ? (unsigned)left2 : 0x7fff0000u;
            }
        }
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const int row = (wave * A_DMA + j) * RPI + d_row;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;                       // rows past the end are computed and thrown away
            if (FOLD) {
                // (launcher: one tap at offset 0, IS == 1 and the grid is the input image: GEMM row m is input pixel m)
                a_base[j] = m - m0;
                a_y[j] = 0;
                a_x[j] = 0;
            } else {
                unsigned b = (unsigned)m / (unsigned)HgWg, rem = (unsigned)m - b * (unsigned)HgWg;
                unsigned y = rem / (unsigned)p.Wg, x = rem - y * (unsigned)p.Wg;
                a_base[j] = (int)(b - b0) * p.Hi * p.Wi;
                a_y[j] = (int)y * p.IS;
                a_x[j] = (int)x * p.IS;
            }
            a_sw[j] = (unsigned)((d_chunk ^ g1_swz<BK>(row)) * 16);
        }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j) {
            const int row = (wave * B_DMA + j) * RPI + d_row;
            voff_b[j] = ((unsigned)(n0 + row) * (unsigned)p.wrow + (unsigned)((d_chunk ^ g1_swz_w<BK>(row)) * 8)) * 2u;
        }
    };
    // issue the DMA of the step under the cursor into stage ibuf and advance; false once every step was issued
    auto issue_next = [&]() -> bool {
        if (iv >= nwg) return false;
        unsigned char* As = smem + ibuf * STAGE;
        unsigned char* Bs = As + A_BYTES;
        const int c0 = ic * BK;
        const bool second = FOLD && it == p.T;                // workgroup-uniform: the phase of the second operand
        unsigned wofs;
        if (!second) {
            const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(reinterpret_cast<const void*>(a_ptr), a_nrec);
            const int tap = p.taps[__builtin_amdgcn_readfirstlane(it)];
            const int oy = (int)(signed char)(tap & 0xff), ox = (int)(signed char)((tap >> 8) & 0xff), wi = (tap >> 16) & 0xffff;
            wofs = (unsigned)(wi * p.Ci + c0) * 2u;
#pragma unroll
            for (int j = 0; j < A_DMA; ++j) {
                if (DIAG(8) && iv != (int)blockIdx.x) continue;
                if (FOLD) {
                    dma16(a_rsrc, As + (wave * A_DMA + j) * 1024, (unsigned)(a_base[j] * p.pix_stride + c0) * 2u + a_sw[j], 0u);
                    continue;
                }
                const int iy = a_y[j] + oy, ix = a_x[j] + ox;
                const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                const unsigned off = (unsigned)((a_base[j] + iy * p.Wi + ix) * p.pix_stride + c0) * 2u + a_sw[j];
                // a tap in the padding: an offset past the descriptor's size makes the DMA write zeros
                dma16(a_rsrc, As + (wave * A_DMA + j) * 1024, ok ? off : PDT_OOB_OFFSET, 0u);
            }
        } else {
            wofs = (unsigned)(p.T * p.Ci + c0) * 2u;
            const __amdgpu_buffer_rsrc_t a2_rsrc = make_rsrc(reinterpret_cast<const void*>(a2_ptr), a2_nrec);
#pragma unroll
            for (int j = 0; j < A_DMA; ++j)
                dma16(a2_rsrc, As + (wave * A_DMA + j) * 1024, (unsigned)(a_base[j] * p.pix_stride2 + c0) * 2u + a_sw[j], 0u);
        }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j)
            if (!(DIAG(2) && iv != (int)blockIdx.x)) dma16(w_rsrc, Bs + (wave * B_DMA + j) * 1024, voff_b[j], wofs);
        ibuf = ibuf + 1 == R ? 0 : ibuf + 1;
        // (kc + (second ? kdiff : 0), not second ? kc2 : kc: a conditional between two variables captured by reference
        // selects an ADDRESS, which keeps the closure in scratch memory and makes the cursor look divergent)
        if (++ic == kc + (second ? kdiff : 0)) {
            ic = 0;
            if (++it == nphase) {
                it = 0;
                iv += gridDim.x;
                if (iv < nwg) setup_issue_tile();
            }
        }
        return true;
    };

    f32x4 acc[MI][NI];
    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * WTN + (frow >> 2) * 16 + j * 4 + (frow & 3);     // see g1_swz_w
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * RB + (((kk * 4 + fq) ^ g1_swz_w<BK>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WTM + i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8*>(As + row * RB + (((kk * 4 + fq) ^ g1_swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // per-lane statistics of the 16 channels this lane owns after the transpose (kept across the tiles of a
    // workgroup while they cover the same channels), and the per-channel constants they need
    // (a launch with a second operand / start values and no backward-sums epilogue never carries statistics: launcher)
    constexpr bool NOSTATS = FOLD && !BSTAT;
    const bool want_stats = !NOSTATS && p.stat_acc != nullptr;
    const bool bwd_mask = BSTAT && p.bstat_scale != nullptr;
    constexpr int NS1 = (PF || NOSTATS) ? 1 : 16;             // (PF: plain sums only)
    constexpr int NS0 = NOSTATS ? 1 : 16;
    float s0[NS0], s1[NS1], ctr[NS1], msc[BSTAT ? 16 : 1], msh[BSTAT ? 16 : 1];
#pragma unroll
    for (int k = 0; k < NS0; ++k) s0[k] = 0.f;
#pragma unroll
    for (int k = 0; k < NS1; ++k) { s1[k] = 0.f; ctr[k] = 0.f; }
    int stat_n0 = -1;

    // Fold the per-lane partial sums of a channel tile into the replicated accumulator: across the 16 lanes
    // of a row (same channels, different pixels) by shuffles, across the waves that share the channels
    // through a small LDS area of its own (the DMA ring stays untouched), then ONE atomic per channel and
    // workgroup -- hundreds of workgroups finish together and all add into the same few hundred addresses.
    // Called by every thread of the workgroup at the same point (the condition is workgroup-uniform).
    auto flush_stats = [&](int n0) {
        if constexpr (NOSTATS) return;
        float mine0 = 0.f, mine1 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a = s0[NOSTATS ? 0 : k], b = PF ? 0.f : s1[NS1 == 1 ? 0 : k];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                a += __shfl_xor(a, off, 64);
                if (!PF) b += __shfl_xor(b, off, 64);
            }
            mine0 = frow == k ? a : mine0;          // lane (fq, frow) keeps channel fq*16 + frow
            mine1 = frow == k ? b : mine1;
            s0[NOSTATS ? 0 : k] = 0.f;
            if (!PF) s1[NS1 == 1 ? 0 : k] = 0.f;
        }
        red[(wave * 2 + 0) * 64 + lane] = mine0;
        red[(wave * 2 + 1) * 64 + lane] = mine1;
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, c = i % BN;
            const int wn_c = c / 64;
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) sum += red[((m * WN + wn_c) * 2 + which) * 64 + (c & 63)];
            atomicAdd(p.stat_acc + ((size_t)(blockIdx.x % p.nrep) * 2 + which) * p.N + n0 + c, sum);
        }
        __syncthreads();
    };
    auto load_consts = [&](int nb) {
        if constexpr (PF || NOSTATS) return;
        const float* center = BSTAT ? p.bstat_mean : p.stat_shift;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            f32x4 c = center ? *reinterpret_cast<const f32x4*>(center + nb + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (NS1 == 16) { ctr[k] = c[0]; ctr[k + 1] = c[1]; ctr[k + 2] = c[2]; ctr[k + 3] = c[3]; }
            if (BSTAT) {
                f32x4 a = bwd_mask ? *reinterpret_cast<const f32x4*>(p.bstat_scale + nb + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 b = bwd_mask ? *reinterpret_cast<const f32x4*>(p.bstat_shift + nb + k) : f32x4{1.f, 1.f, 1.f, 1.f};
                msc[k] = a[0]; msc[k + 1] = a[1]; msc[k + 2] = a[2]; msc[k + 3] = a[3];
                msh[k] = b[0]; msh[k + 1] = b[1]; msh[k + 2] = b[2]; msh[k + 3] = b[3];
            }
        }
    };

    // ---- PF: the two register sets of the epilogue's operands and the requests that fill the second one
    constexpr int PMI = PF ? MI : 1;
    u32x4 cav[PMI][2], nav[PMI][2];                            // addend: this tile's / the next tile's
    unsigned cmb[PMI], nmb[PMI];                               // packed mask words
    const bool pf_addend = PF && p.addend != nullptr && !DIAG(16), pf_mask = PF && p.mask_bits != nullptr && !DIAG(16);
    auto issue_epilogue_loads = [&](int m0, int n0, bool live) {
        // Descriptors based at each row group of the wave tile and ending with the tensor (an absent operand: zero records, the
        // loads return zeros without touching memory, so the number of requests per tile is a constant; the same after
        // the workgroup's last tile: the requests are issued unconditionally, see the tile loop).  They are
        // assembled by hand because the assembly statement wants them as four scalar registers.
        const int row0 = m0 + wm * WTM;
        const unsigned long long e0 = (unsigned long long)row0 * (unsigned)p.N + (unsigned)(n0 + wn * WTN);
        const int rows_left = live ? p.M - row0 : 0;
        const unsigned long long a_addr = (unsigned long long)p.addend + e0 * 2ull;
        const unsigned long long m_addr = (unsigned long long)p.mask_bits + (e0 >> 3);
        const int row16 = 16 * p.N * 2;
        const unsigned bit_off = ep_ld >> 4;
        // Addend at HALF resolution (the gradient of a stride-2 1x1 shortcut, present at the even pixels only): one
        // descriptor per wave tile, based at the half-resolution image of the tile's first row, and a per-lane offset --
        // the row's position carried from one per-tile decomposition by exact multiply-shift divisions -- that points
        // past the descriptor for odd pixels and rows past the end, so those lanes receive the zeros they need.
        const bool half = p.addend_half != 0;                              // workgroup-uniform
        unsigned hy0 = 0, hx0 = 0, h_nrec = 0;
        unsigned long long h_addr = 0;
        const unsigned Hh = (unsigned)p.Ho >> 1, Wh = (unsigned)p.Wo >> 1;
        if (half) {
            const unsigned mw = (unsigned)(row0 < p.M ? row0 : p.M - 1);
            const unsigned hb0 = mw / (unsigned)HgWg, rem = mw - hb0 * (unsigned)HgWg;
            hy0 = rem / (unsigned)p.Wg;
            hx0 = rem - hy0 * (unsigned)p.Wg;
            const unsigned long long base = ((unsigned long long)hb0 * Hh * Wh * (unsigned)p.N + (unsigned)(n0 + wn * WTN)) * 2ull;
            const unsigned long long left = p.addend_half_bytes > base ? p.addend_half_bytes - base : 0ull;
            h_addr = (unsigned long long)p.addend + base;
            h_nrec = (live && pf_addend) ? (left < 0x7fff0000ull ? (unsigned)left : 0x7fff0000u) : 0u;
        }
#pragma unroll
        for (int e = 0; e < PMI; ++e) {
            // (the row group's offset is part of the descriptor's base: the range check never sees a scalar offset)
            int have = rows_left - e * 16;
            have = have < 0 ? 0 : (have > 16 ? 16 : have);
            const unsigned nrec = (unsigned)(have * p.N * 2);
            unsigned a_off = ep_ld;
            if (half) {
                const unsigned t = (unsigned)(e * 16 + frow);
                const unsigned tx = hx0 + t, qx = (tx * p.magic_w) >> 20, x = tx - qx * (unsigned)p.Wg;
                const unsigned ty = hy0 + qx, qy = (ty * p.magic_h) >> 20, y = ty - qy * (unsigned)p.Hg;
                const bool ok = !((x | y) & 1u) && (int)t < rows_left;
                a_off = ok ? ((qy * Hh + (y >> 1)) * Wh + (x >> 1)) * (unsigned)(p.N * 2) + (unsigned)(fq * 32) : PDT_OOB_OFFSET;
            }
            const unsigned long long a_e = half ? h_addr : a_addr + (unsigned)(e * row16), m_e = m_addr + (unsigned)((e * row16) >> 4);
            const u32x4 a_desc = {(unsigned)a_e, (unsigned)(a_e >> 32) & 0xffffu, half ? h_nrec : (pf_addend ? nrec : 0u), 0x00020000u};
            const u32x4 m_desc = {(unsigned)m_e, (unsigned)(m_e >> 32) & 0xffffu, pf_mask ? nrec >> 4 : 0u, 0x00020000u};
            // (s_nop 4: the scalar registers may have been written by the instructions just before)
            asm volatile(
                "s_nop 4\n\t"
                "buffer_load_dwordx4 %0, %3, %4, 0 offen nt\n\t"
                "buffer_load_dwordx4 %1, %3, %4, 0 offen offset:16 nt\n\t"
                "buffer_load_ushort %2, %5, %6, 0 offen"
                : "=&v"(nav[e][0]), "=&v"(nav[e][1]), "=&v"(nmb[e])
                : "v"(a_off), "s"(a_desc), "v"(bit_off), "s"(m_desc)
                : "memory");
        }
    };
    // The second set has landed once this returns: the wait leaves the YOUNGER operations in flight (the result stores
    // of the tile that just ended, ``younger`` of them; the counter retires in issue order), and it names the set's
    // registers, so no instruction that reads them can be scheduled above it.  Then the sets change roles.
    auto take_epilogue_loads = [&](auto younger_tag) {
        static_assert(PMI == 1 || PMI == 2 || PMI == 4, "row groups per wave tile");
        constexpr int younger = decltype(younger_tag)::value;
        if constexpr (PMI == 4)
            asm volatile("s_waitcnt vmcnt(%12)"
                         : "+v"(nav[0][0]), "+v"(nav[0][1]), "+v"(nav[1][0]), "+v"(nav[1][1]), "+v"(nav[2][0]), "+v"(nav[2][1]),
                           "+v"(nav[3][0]), "+v"(nav[3][1]), "+v"(nmb[0]), "+v"(nmb[1]), "+v"(nmb[2]), "+v"(nmb[3])
                         : "n"(younger)
                         : "memory");
        else if constexpr (PMI == 2)
            asm volatile("s_waitcnt vmcnt(%6)"
                         : "+v"(nav[0][0]), "+v"(nav[0][1]), "+v"(nav[1][0]), "+v"(nav[1][1]), "+v"(nmb[0]), "+v"(nmb[1])
                         : "n"(younger)
                         : "memory");
#pragma unroll
        for (int e = 0; e < PMI; ++e) {
            cav[e][0] = nav[e][0];
            cav[e][1] = nav[e][1];
            cmb[e] = nmb[e];
        }
    };

    // ---- pipeline: R stages; at step s the DMA of step s+R-1 is issued (after the barrier that proves every
    // wave is done with the stage it overwrites) and the wait leaves the R-2 younger groups in flight
    const bool has_bias = FOLD && p.bias != nullptr;
    auto stage_bias = [&]() {                  // workgroup-uniform call sites; the barrier drains what is in flight
        for (int i = tid; i < p.N; i += 256) bias_lds[i] = p.bias[i];
        __syncthreads();
    };
    const bool bias_in_lds = has_bias && p.N <= DMA_MAX_BIAS;      // wider layers read their 16 values per tile from memory
    if (bias_in_lds) stage_bias();
    setup_issue_tile();
    const int first_m0 = i_m0, first_n0 = i_n0;
    bool issuing = true;
#pragma unroll
    for (int i = 0; i < R - 1; ++i) issuing = issue_next() && issuing;
    if constexpr (PF) {               // (first tile: its epilogue operands behind its first stage, waited for at once)
        issue_epilogue_loads(first_m0, first_n0, true);
        take_epilogue_loads(std::integral_constant<int, 0>{});
    }
    int buf = 0;
    int v = blockIdx.x;
    while (true) {
        int m0, n0;
        tile_origin(v, m0, n0);               // the tile computed now
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            // accumulator register r of lane group fq: channel n0 + wn*WTN + fq*16 + j*4 + r
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias_in_lds) b0 = *reinterpret_cast<const f32x4*>(bias_lds + n0 + wn * WTN + fq * 16 + j * 4);
            else if (has_bias) b0 = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WTN + fq * 16 + j * 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = b0;
        }
        const int vn = v + gridDim.x;
        const bool more = vn < nwg;
        // one K step: wait for its stage, refill the stage after it, multiply
        auto k_step = [&](bool last, bool first) {
            // This wave's share of the stage has landed: all but the (R-2)*G youngest operations are done, and
            // the operations issued after this step's group are at least the R-2 groups behind it (epilogue loads
            // and stores in between only make the wait more conservative).  Once the cursor has run out, fewer
            // groups follow: drain.
            // (PF, a tile's first step: its stage was requested BEFORE the epilogue operands that the wait at the end of
            // the previous tile -- or of the prologue -- has seen land, so it has landed too; waiting here would only
            // wait for the previous tile's result stores, and exposing their round trip once per tile is what held the
            // store-heavy launches at 4.2 TB/s.)
            if (PF && first) {
            } else if (issuing) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ... and so has every other wave's; all waves are also done reading the stage refilled next
            __builtin_amdgcn_s_barrier();
            // (PF: the cursor stands on the next tile's first step during this tile's last one (R == 2): that tile's
            // epilogue operands go out behind its first operand slices)
            const int pm0 = i_m0, pn0 = i_n0;
            if (issuing) issuing = issue_next();
            if (PF && last) issue_epilogue_loads(pm0, pn0, more);
            if (!DIAG(4)) compute(buf);
            buf = buf + 1 == R ? 0 : buf + 1;
        };
        if constexpr (PF) {
            // the last step stands outside the loop: the requests it issues define the second register set on a straight
            // path to the wait that ends the tile (a definition under a condition or inside the loop would let the compiler
            // place register copies of that set between the request and the wait)
            for (int ks = 0; ks < nk - 1; ++ks) k_step(false, ks == 0);
            k_step(true, nk == 1);
        } else {
            for (int ks = 0; ks < nk; ++ks) k_step(false, false);
        }

        // ---- epilogue, straight from the accumulators ------------------------------------------------------
        const int nb = n0 + wn * WTN + fq * 16;                       // first of this lane's 16 channels
        if (want_stats && stat_n0 != n0) {
            if (stat_n0 >= 0) {
                flush_stats(stat_n0);
                if (bias_in_lds) stage_bias();
            }
            stat_n0 = n0;
            load_consts(nb);
        }
        // LEAN EPILOGUE (output row == GEMM row, addend at full resolution: every 1x1 / stride-1 launch and every
        // stride-1 3x3 one): the wave tile's operands are addressed through buffer descriptors based at the
        // wave tile's origin, with a loop-invariant per-lane offset and the row group as SCALAR offset -- no per-row
        // index arithmetic, no 64-bit address math, no predicates; ReLU and the sign mask work on the packed bf16
        // pairs.  The epilogue's vector instructions are what bounds the one- and two-K-step launches (with every load
        // and store switched off the 64 -> 256 channel block tail at 56 x 56 still took 45 % of its time), so their
        // count is what this path is about.
        constexpr int EG = (BSTAT && MI > 2) ? 2 : MI;
        const bool has_addend = p.addend != nullptr && !DIAG(16), has_mask = p.mask_bits != nullptr && !DIAG(16);
        if (DIAG(32)) {
            // (ablation: no epilogue at all; the accumulators are kept alive)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if constexpr (!GEN) {
            const int row0 = m0 + wm * WTM;                                  // wave-uniform
            const size_t e0 = (size_t)row0 * (size_t)p.N + (size_t)(n0 + wn * WTN);
            // One descriptor per tensor and row group, based at the group's first row and ending with the tensor (at most
            // ``rows`` rows on): loads of rows past M return zeros, their stores are dropped.  The group's offset must be
            // part of the BASE: the range check sees the vector offset and the instruction offset only, never the
            // scalar one.
            const int rows_left = p.M - row0;
            const bool partial = rows_left < WTM;
            const int row16 = 16 * p.N * 2;
            auto rows_bytes = [&](int first, int rows) -> unsigned {
                int have = rows_left - first;
                have = have < 0 ? 0 : (have > rows ? rows : have);
                return (unsigned)(have * p.N * 2);
            };
            const bool relu = FOLD && !BSTAT && p.relu_out, want_bits = FOLD && !BSTAT && p.bits_out != nullptr;
            const unsigned char* out_b = reinterpret_cast<const unsigned char*>(p.out + e0);
            const unsigned char* add_b = reinterpret_cast<const unsigned char*>(p.addend + e0);
            const unsigned char* y_b = reinterpret_cast<const unsigned char*>(p.bstat_y + e0);
            const unsigned char* msk_b = p.mask_bits + (e0 >> 3);
            const unsigned char* bit_b = p.bits_out + (e0 >> 3);
#pragma unroll
            for (int i0 = 0; i0 < MI; i0 += EG) {
                u32x4 yv[BSTAT ? EG : 1][2], av[EG][2];
                unsigned mb[EG];
#pragma unroll
                for (int e = 0; e < EG; ++e) {
                    const int so = (i0 + e) * row16;
                    if constexpr (PF) {
                        av[e][0] = cav[PF ? i0 + e : 0][0];
                        av[e][1] = cav[PF ? i0 + e : 0][1];
                        mb[e] = cmb[PF ? i0 + e : 0];
                        continue;
                    }
                    const unsigned nrec = rows_bytes((i0 + e) * 16, 16);
                    if (BSTAT) {
                        const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(y_b + so, nrec);
                        yv[e][0] = __builtin_amdgcn_raw_buffer_load_b128(y_rs, (int)ep_ld, 0, 0);
                        yv[e][1] = __builtin_amdgcn_raw_buffer_load_b128(y_rs, (int)ep_ld + 16, 0, 0);
                    }
                    if (has_addend) {                                       // (non-temporal: last use of the tensor)
                        const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(add_b + so, nrec);
                        av[e][0] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)ep_ld, 0, 2);
                        av[e][1] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)ep_ld + 16, 0, 2);
                    }
                    if (has_mask) {
                        const __amdgpu_buffer_rsrc_t m_rs = make_rsrc(msk_b + (so >> 4), nrec >> 4);
                        mb[e] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(m_rs, (int)(ep_ld >> 4), 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < EG; ++e) {
                    const int i = i0 + e;
                    const int so = i * row16;
                    float val[16];
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) val[j * 4 + r] = acc[i][j][r];
                    if (has_addend) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const unsigned a = av[e][h][d];
                                const int k = h * 8 + d * 2;
                                val[k] += bf_lo(a);
                                val[k + 1] += bf_hi(a);
                            }
                    }
                    unsigned pk[8];                                         // the bf16 pairs that will be stored
#pragma unroll
                    for (int d = 0; d < 8; ++d) pk[d] = pack2(val[d * 2], val[d * 2 + 1]);
                    if (has_mask) {
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)mb[e], 2 * d, 1);
                            const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)mb[e], 2 * d + 1, 1);
                            pk[d] &= (lo & 0xffffu) | (hi & ~0xffffu);           // (one bit-field insert)
                        }
                    }
                    if (relu) {
#pragma unroll
                        for (int d = 0; d < 8; ++d) pk[d] = pk_relu(pk[d]);
                    }
                    // (PF: issued whether or not the mask is wanted -- an absent tensor has a descriptor of zero records --
                    // so that the number of stores behind the next tile's requests is a constant)
                    if (want_bits || (PF && FOLD)) {
                        // after the clamp a half is 0x0000 .. 0x7fff (or a NaN pattern >= 0x7f81): adding 0x7fff sets bit 15
                        // of exactly the positive ones, with no carry into the upper half
                        unsigned m16 = 0;
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            const unsigned t = (relu ? pk[d] : (pk[d] & 0x7fff7fffu)) + 0x7fff7fffu;
                            const unsigned two = __builtin_amdgcn_ubfe(t, 15, 1) | (__builtin_amdgcn_ubfe(t, 31, 1) << 1);
                            m16 |= two << (2 * d);
                        }
                        if (!relu) {                                         // (not used today: signs would have to count)
#pragma unroll
                            for (int d = 0; d < 8; ++d) {
                                const unsigned neg = __builtin_amdgcn_ubfe(pk[d], 15, 1) | (__builtin_amdgcn_ubfe(pk[d], 31, 1) << 1);
                                m16 &= ~(neg << (2 * d));
                            }
                        }
                        const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(bit_b + (so >> 4), want_bits ? rows_bytes(i * 16, 16) >> 4 : 0u);
                        __builtin_amdgcn_raw_buffer_store_b16((short)m16, b_rs, (int)(ep_ld >> 4), 0, 0);
                    }
                    auto add_stats = [&]() {
                        // rows past M (last tile only) were computed from a repeated input row: keep them out of the sums
                        if (partial && (BSTAT || want_stats) && i * 16 + frow >= rows_left) return;
                        if constexpr (BSTAT) {
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int d = 0; d < 4; ++d) {
                                    const unsigned yy2 = yv[e][h][d], pp = pk[h * 4 + d];
                                    const int k = h * 8 + d * 2;
                                    const float y0 = bf_lo(yy2), y1 = bf_hi(yy2);
                                    float d0 = bf_lo(pp), d1 = bf_hi(pp);
                                    if (bwd_mask) {
                                        d0 = (y0 * msc[k] + msh[k]) > 0.0f ? d0 : 0.0f;
                                        d1 = (y1 * msc[k + 1] + msh[k + 1]) > 0.0f ? d1 : 0.0f;
                                    }
                                    s0[k] += d0;
                                    s1[k] += d0 * (y0 - ctr[k]);
                                    s0[k + 1] += d1;
                                    s1[k + 1] += d1 * (y1 - ctr[k + 1]);
                                }
                        } else if constexpr (NOSTATS) {
                        } else if (want_stats) {
                            if constexpr (PF) {
#pragma unroll
                                for (int d = 0; d < 8; ++d) {
                                    s0[2 * d] += bf_lo(pk[d]);
                                    s0[2 * d + 1] += bf_hi(pk[d]);
                                }
                            } else {
#pragma unroll
                                for (int d = 0; d < 8; ++d) {
                                    const float a = bf_lo(pk[d]) - ctr[PF ? 0 : 2 * d], b = bf_hi(pk[d]) - ctr[PF ? 0 : 2 * d + 1];
                                    s0[2 * d] += a;
                                    s1[PF ? 0 : 2 * d] += a * a;
                                    s0[2 * d + 1] += b;
                                    s1[PF ? 0 : 2 * d + 1] += b * b;
                                }
                            }
                        }
                    };
                    auto store_result = [&]() {
                        // whole 128-byte lines per store instruction: see the generic path below
                        const bool up = (frow & 8) != 0;
                        u32x4 first, second;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned got = dpp_xor8(up ? pk[d] : pk[4 + d]);
                            first[d] = up ? got : pk[d];
                            second[d] = up ? pk[4 + d] : got;
                        }
                        if (!DIAG(1)) {
                            const __amdgpu_buffer_rsrc_t o1 = make_rsrc(out_b + so, rows_bytes(i * 16, 8));
                            const __amdgpu_buffer_rsrc_t o2 = make_rsrc(out_b + so + (row16 >> 1), rows_bytes(i * 16 + 8, 8));
                            __builtin_amdgcn_raw_buffer_store_b128(first, o1, (int)ep_st, 0, 2);
                            __builtin_amdgcn_raw_buffer_store_b128(second, o2, (int)ep_st, 0, 2);
                        }
                    };
                    if (BSTAT) {
                        store_result();
                        add_stats();
                    } else {
                        add_stats();
                        store_result();
                    }
                }
            }
        } else {
        // GENERIC EPILOGUE (output rows scattered over a strided grid -- the output-parity classes of a stride-2 data
        // gradient -- or the addend at half resolution): every row has an address of its own, so operands go through flat
        // 64-bit addresses; what is shared with the lean path is the arithmetic on packed bf16 pairs.  The row's position
        // on the GEMM grid is NOT divided out per row (two 32-bit divisions are ~60 vector instructions, per row group):
        // the wave tile's first row is decomposed once per tile on wave-uniform values, and a row is that position plus
        // i * 16 + frow, carried into (x, y, image) with multiply-shift divisions that are exact for the few hundred
        // values that can occur (launcher: magic = 2^20 / d + 1, (d + 256) d < 2^20).
        // Epilogue operands of EG row groups first, so they are all in flight together: all MI of them, except with
        // the backward-sums epilogue on a 64-row wave tile, where four groups of loads on top of the 80 registers of
        // sums and per-channel constants would not fit 256 VGPRs -- two passes of two groups there.
        static_assert(!(GEN && FOLD), "the second-operand launches are 1x1 / dense: lean epilogue only");
        const bool grid_pos = !p.simple_out || p.addend_half;
        const int roww = m0 + wm * WTM;                                   // first row of the wave tile (wave-uniform)
        const bool any_live = roww < p.M;
        unsigned gb0 = 0, gy0 = 0, gx0 = 0;
        if (grid_pos) {
            const unsigned mw = (unsigned)__builtin_amdgcn_readfirstlane(any_live ? roww : p.M - 1);
            gb0 = mw / (unsigned)HgWg;
            const unsigned rem = mw - gb0 * (unsigned)HgWg;
            gy0 = rem / (unsigned)p.Wg;
            gx0 = rem - gy0 * (unsigned)p.Wg;
        }
        const size_t orow0 = (size_t)(any_live ? roww : p.M - 1);
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += EG) {
        size_t oidx[EG];
        bool live[EG];
        u32x4 yv[BSTAT ? EG : 1][2], av[EG][2];
        unsigned mb[EG];
#pragma unroll
        for (int e = 0; e < EG; ++e) {
            const int i = i0 + e;
            live[e] = roww + i * 16 + frow < p.M;
            const unsigned t = live[e] ? (unsigned)(i * 16 + frow) : 0u;   // rows past the end: the tile's first row (loads stay inside)
            unsigned yo = 0, xo = 0, bimg = 0;
            size_t orow = orow0 + t;
            if (grid_pos) {
                const unsigned tx = gx0 + t, qx = (tx * p.magic_w) >> 20, x = tx - qx * (unsigned)p.Wg;
                const unsigned ty = gy0 + qx, qy = (ty * p.magic_h) >> 20, y = ty - qy * (unsigned)p.Hg;
                bimg = gb0 + qy;
                yo = y * (unsigned)p.OS + (unsigned)p.oy0;
                xo = x * (unsigned)p.OS + (unsigned)p.ox0;
                if (!p.simple_out) orow = ((size_t)bimg * p.Ho + yo) * p.Wo + xo;
            }
            const size_t o = orow * p.N + nb;
            oidx[e] = o;
            if (BSTAT) {
                const u32x4* yp = reinterpret_cast<const u32x4*>(p.bstat_y + o);
                yv[e][0] = yp[0];
                yv[e][1] = yp[1];
            }
            if (has_addend) {
                long long ai = (long long)o;
                if (p.addend_half)
                    ai = ((yo | xo) & 1u) ? -1
                                          : (long long)((((size_t)bimg * (p.Ho >> 1) + (yo >> 1)) * (p.Wo >> 1) + (xo >> 1)) * p.N + nb);
                if (ai >= 0) {
                    const u32x4* ap = reinterpret_cast<const u32x4*>(p.addend + ai);
                    av[e][0] = __builtin_nontemporal_load(ap);        // last use of the incoming gradient
                    av[e][1] = __builtin_nontemporal_load(ap + 1);
                } else {
                    av[e][0] = u32x4{0u, 0u, 0u, 0u};
                    av[e][1] = u32x4{0u, 0u, 0u, 0u};
                }
            }
            if (has_mask) mb[e] = *reinterpret_cast<const unsigned short*>(p.mask_bits + (o >> 3));
        }
#pragma unroll
        for (int e = 0; e < EG; ++e) {
            const int i = i0 + e;
            // register r of block j holds channel 16*fq + 4*j + r of this lane's pixel (see g1_swz_w); the value (plus the
            // addend) is rounded to bf16 ONCE, and everything after that works on the packed pairs that will be stored
            float val[16];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) val[j * 4 + r] = acc[i][j][r];
            if (has_addend) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned a = av[e][h][d];
                        const int k = h * 8 + d * 2;
                        val[k] += bf_lo(a);
                        val[k + 1] += bf_hi(a);
                    }
            }
            unsigned pk[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) pk[d] = pack2(val[d * 2], val[d * 2 + 1]);
            if (has_mask) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)mb[e], 2 * d, 1);
                    const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)mb[e], 2 * d + 1, 1);
                    pk[d] &= (lo & 0xffffu) | (hi & ~0xffffu);
                }
            }
            // (order chosen per variant by the register allocation it leads to: the backward-sums variants spill otherwise)
            auto add_stats = [&]() {
                if (!live[e]) return;
                if constexpr (BSTAT) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned yy2 = yv[e][h][d], pp = pk[h * 4 + d];
                            const int k = h * 8 + d * 2;
                            const float y0 = bf_lo(yy2), y1 = bf_hi(yy2);
                            float d0 = bf_lo(pp), d1 = bf_hi(pp);
                            if (bwd_mask) {
                                d0 = (y0 * msc[k] + msh[k]) > 0.0f ? d0 : 0.0f;
                                d1 = (y1 * msc[k + 1] + msh[k + 1]) > 0.0f ? d1 : 0.0f;
                            }
                            s0[k] += d0;
                            s1[k] += d0 * (y0 - ctr[k]);
                            s0[k + 1] += d1;
                            s1[k + 1] += d1 * (y1 - ctr[k + 1]);
                        }
                } else if constexpr (NOSTATS || PF) {
                } else if (want_stats) {
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const float a = bf_lo(pk[d]) - ctr[2 * d], b = bf_hi(pk[d]) - ctr[2 * d + 1];
                        s0[2 * d] += a;
                        s1[2 * d] += a * a;
                        s0[2 * d + 1] += b;
                        s1[2 * d + 1] += b * b;
                    }
                }
            };
            auto store_result = [&]() {
                // Result stores in whole 128-byte lines: lane (fq, frow) holds bytes [32 fq, 32 fq + 32) of its pixel's
                // 128-byte segment; lanes frow and frow ^ 8 trade one 16-byte half (a rotation by 8 inside the row of 16
                // lanes), after which the eight lanes (fq, frow & 8) hold the eight 16-byte pieces of ONE pixel: a store
                // instruction writes 8 full lines instead of 16 rows x 4 pieces of 16 bytes at a stride of 32 (streaming
                // stores in that split pattern reach 2.2 TB/s, in whole lines 5.2: bench/diag/store_pattern.hip).
                const u32x4 o0 = {pk[0], pk[1], pk[2], pk[3]}, o1 = {pk[4], pk[5], pk[6], pk[7]};
                const bool up = (frow & 8) != 0;
                u32x4 got;
#pragma unroll
                for (int d = 0; d < 4; ++d) got[d] = dpp_xor8(up ? o0[d] : o1[d]);
                // the partner's output element (a multiple of 16, so bit 0 carries "row past the end") as a signed
                // 32-bit distance from this lane's: the two pixels are at most a few images apart
                const unsigned mine = (unsigned)oidx[e] | (live[e] ? 0u : 1u);
                const unsigned other = dpp_xor8(mine);
                const int dist = (int)((other & ~1u) - (unsigned)oidx[e]);
                // piece 2 fq + (frow >> 3) of the pixel of lane frow & 7 (first) and of lane 8 + (frow & 7) (second)
                const bool ok_first = !((up ? other : mine) & 1u), ok_second = !((up ? mine : other) & 1u);
                bf16* o_first = p.out + oidx[e] + (up ? dist + 8 : 0);
                bf16* o_second = p.out + oidx[e] + (up ? 8 : dist);
                const u32x4 v_first = up ? got : o0, v_second = up ? o1 : got;
                if (ok_first && !DIAG(1)) __builtin_nontemporal_store(v_first, reinterpret_cast<u32x4*>(o_first));
                if (ok_second && !DIAG(1)) __builtin_nontemporal_store(v_second, reinterpret_cast<u32x4*>(o_second));
            };
            if (BSTAT) {
                store_result();
                add_stats();
            } else {
                add_stats();
                store_result();
            }
        }
        }
        }
        // (PF: the lean epilogue issued two result stores per row group, three with the sign mask, behind the requests)
#ifdef PDT_DMA_DIAG
        if constexpr (PF) take_epilogue_loads(std::integral_constant<int, 0>{});      // (the ablation build may skip stores)
#else
        if constexpr (PF) take_epilogue_loads(std::integral_constant<int, (FOLD ? 3 : 2) * MI>{});
#endif
        if (!more) break;
        v = vn;
    }
    if (want_stats && stat_n0 >= 0) flush_stats(stat_n0);
}

template <int BN, int WM, int WN, int BK, int R, bool BSTAT, bool FOLD, bool GEN, bool PF>
static int g1_resident_blocks_of() {
    static int cached = 0;
    if (cached > 0) return cached;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 512;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_dma_kernel<BN, WM, WN, BK, R, BSTAT, FOLD, GEN, PF>, 256, 0) != hipSuccess || per_cu <= 0)
        per_cu = 2;
    cached = per_cu * cus;
    return cached;
}

template <int BN, int WM, int WN, int BK, int R, bool BSTAT, bool FOLD = false>
static int g1_resident_blocks() {
    return g1_resident_blocks_of<BN, WM, WN, BK, R, BSTAT, FOLD, false, false>();
}

template <int BN, int WM, int WN, int BK, int R, bool BSTAT, bool FOLD, bool GEN, bool PF>
static int g1_launch_of(const ConvDmaParams& p, hipStream_t s) {
    long long nwg = (long long)((p.M + 127) / 128) * (p.N / BN);
    if (nwg <= 0 || nwg > 0x7fffffffLL) return -2;
    long long grid = g1_resident_blocks_of<BN, WM, WN, BK, R, BSTAT, FOLD, GEN, PF>();
    if (grid > nwg) grid = nwg;
    if (g_pdt_deterministic && grid > p.nrep) grid = p.nrep;      // one statistics row per workgroup
    hipLaunchKernelGGL((conv_dma_kernel<BN, WM, WN, BK, R, BSTAT, FOLD, GEN, PF>), dim3((unsigned)grid), dim3(256), 0, s, p);
    return pdt_launch_status();
}

// PDT_DMA_PREFETCH=0: never pick the prefetching instantiations (A/B measurements)
static const bool g_dma_prefetch = !(getenv("PDT_DMA_PREFETCH") && atoi(getenv("PDT_DMA_PREFETCH")) == 0);

template <int BN, int WM, int WN, int BK, int R, bool BSTAT, bool FOLD = false>
static int g1_launch(const ConvDmaParams& p, hipStream_t s) {
    // the lean epilogue wants output row == GEMM row and a full-resolution addend (always true with a second operand /
    // start values: the launcher admits only the plain 1x1 geometry there)
    // the prefetching variant: epilogue operands to hide, and statistics (if any) that are plain sums
    const bool pf = !BSTAT && R == 2 && g_dma_prefetch && (p.addend || p.mask_bits) && (!p.stat_acc || p.sums_only);
    if constexpr (!FOLD) {
        // (an addend at half resolution is understood by the prefetching variant and the generic one only)
        if (!p.simple_out || (p.addend_half && !pf)) return g1_launch_of<BN, WM, WN, BK, R, BSTAT, false, true, false>(p, s);
    }
    if constexpr (!BSTAT && R == 2) {
        if (pf) return g1_launch_of<BN, WM, WN, BK, R, false, FOLD, false, true>(p, s);
    }
    return g1_launch_of<BN, WM, WN, BK, R, BSTAT, FOLD, false, false>(p, s);
}

// tile: 0 = auto (3 when N is a multiple of 128, else 4); 3 = 128 x 128 with K slices of 64 and two stages, 4 = 128 x 64
// with K slices of 64 and two stages (1 / 2 are old names of 3 / 4); 5 = 128 x 64 with K slices of 32 and four stages,
// for operands with 32 channels per tap (the stem), never chosen automatically
// Persistent grid size (resident workgroups) of a kernel variant: tile 3 / 4, with or without the backward-sums epilogue
// and the second-operand path (diagnostics).
PDT_API int pdt_conv_dma_resident(int tile, int bstat, int fold) {
    if (tile == 3) {
        if (fold) return bstat ? g1_resident_blocks<128, 2, 2, 64, 2, true, true>() : g1_resident_blocks<128, 2, 2, 64, 2, false, true>();
        return bstat ? g1_resident_blocks<128, 2, 2, 64, 2, true>() : g1_resident_blocks<128, 2, 2, 64, 2, false>();
    }
    if (fold) return bstat ? g1_resident_blocks<64, 4, 1, 64, 2, true, true>() : g1_resident_blocks<64, 4, 1, 64, 2, false, true>();
    return bstat ? g1_resident_blocks<64, 4, 1, 64, 2, true>() : g1_resident_blocks<64, 4, 1, 64, 2, false>();
}

// in2 / Ci2 / pix_stride2 / bias: the optional second operand and accumulator start values (see the file header);
// the weight rows then hold T*Ci columns for the taps followed by Ci2 columns for the second operand.
PDT_API int pdt_conv_dma_ex(const void* in, const void* wgt, void* out, const void* addend, int addend_half,
                            const void* mask_bits, float* stat_acc, const float* stat_shift, const void* bstat_y,
                            const float* bstat_mean, const float* bstat_scale, const float* bstat_shift, long long M, int Hi,
                            int Wi, int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS, int oy0, int ox0,
                            int N, int wrow, int T, const signed char* offy, const signed char* offx,
                            const unsigned short* widx, int tile, const void* in2, int Ci2, int pix_stride2,
                            const float* bias, int relu_out, void* bits_out, int sums_only, hipStream_t s) {
    if (T <= 0 || T > DMA_MAX_TAPS) return -3;
    if (M <= 0) return 0;
    if (!in2) Ci2 = 0;
    if (M >= 0x7fffffffLL || N < 64 || N % 64 != 0 || wrow < T * Ci + Ci2) return -4;
    if (in2 || bias) {
        // these instantiations address their operands by GEMM row: the plain 1x1 / stride 1 geometry only
        if (tile == 5 || T != 1 || offy[0] != 0 || offx[0] != 0 || IS != 1 || Hg != Hi || Wg != Wi) return -11;
    }
    if (in2) {
        if (Ci2 < 64 || Ci2 % 64 != 0 || pix_stride2 < Ci2 || pix_stride2 % 8 != 0) return -11;
    }
    if ((relu_out || bits_out) && (!(in2 || bias) || bstat_y)) return -12;   // forward instantiations with a second operand / start values only
    if ((in2 || bias) && stat_acc && !bstat_y) return -12;                   // ... which carry no statistics epilogue
    if (tile == 5) {
        // K slices of 32 over a gathered operand whose "pixels" may overlap (the stem reads 32 consecutive values of its
        // packed NHWC4 image per tap: pix_stride 4, Ci 32): every DMA source address must still be 16-byte aligned
        if (Ci < 32 || Ci % 32 != 0 || bstat_y || addend || mask_bits) return -4;
        if ((IS * pix_stride) % 8 != 0 || (Wi * pix_stride) % 8 != 0) return -4;
        for (int i = 0; i < T; ++i)
            if (((int)offx[i] * pix_stride) % 8 != 0) return -4;
    } else if (Ci < 64 || Ci % 64 != 0 || pix_stride < Ci || pix_stride % 8 != 0) {
        return -4;
    }
    if (bstat_y && (!stat_acc || !bstat_mean || (bstat_scale && !bstat_shift))) return -8;
    if (Hg <= 0 || Wg <= 0 || IS < 1 || OS < 1 || M % ((long long)Hg * Wg) != 0) return -5;
    if (addend && addend_half && ((Ho | Wo) & 1)) return -10;
    ConvDmaParams p;
    p.in = (const bf16*)in; p.wgt = (const bf16*)wgt; p.out = (bf16*)out;
    p.addend = (const bf16*)addend; p.mask_bits = (const unsigned char*)mask_bits;
    p.stat_acc = stat_acc; p.stat_shift = stat_shift;
    p.bstat_y = (const bf16*)bstat_y; p.bstat_mean = bstat_mean; p.bstat_scale = bstat_scale; p.bstat_shift = bstat_shift;
    p.M = (int)M; p.N = N;
    p.Hi = Hi; p.Wi = Wi; p.pix_stride = pix_stride; p.Ci = Ci;
    p.Hg = Hg; p.Wg = Wg; p.IS = IS; p.Ho = Ho; p.Wo = Wo; p.OS = OS; p.oy0 = oy0; p.ox0 = ox0;
    p.wrow = wrow; p.T = T;
    p.simple_out = (OS == 1 && oy0 == 0 && ox0 == 0 && Hg == Ho && Wg == Wo) ? 1 : 0;
    if ((long long)(Wg + 256) * Wg >= (1 << 20) || (long long)(Hg + 256) * Hg >= (1 << 20)) return -13;
    p.magic_w = (1u << 20) / (unsigned)Wg + 1u;
    p.magic_h = (1u << 20) / (unsigned)Hg + 1u;
    p.addend_half_bytes = (addend && addend_half) ? (unsigned long long)(M / ((long long)Hg * Wg)) * (Ho >> 1) * (Wo >> 1) * N * 2ull : 0ull;
    p.addend_half = (addend && addend_half) ? 1 : 0;
    p.nrep = pdt_nrep();
    p.in2 = (const bf16*)in2; p.bias = bias; p.Ci2 = Ci2; p.pix_stride2 = in2 ? pix_stride2 : 0;
    p.relu_out = relu_out ? 1 : 0; p.bits_out = (unsigned char*)bits_out;
    p.sums_only = (sums_only && stat_acc && !stat_shift && !bstat_y) ? 1 : 0;
#ifdef PDT_DMA_DIAG
    { const char* e = getenv("PDT_DMA_DIAG"); p.diag = e ? atoi(e) : 0; }
#endif
    p.in2_bytes = in2 ? (unsigned long long)(M * (long long)pix_stride2 * 2) : 0ull;
    {
        long long images = M / ((long long)Hg * Wg);
        long long in_b = images * Hi * Wi * pix_stride * 2, w_b = (long long)N * wrow * 2;
        // 32-bit byte offsets from a per-tile base: what ONE tile reaches (its rows' images, at most 256 rows) and the
        // weights must stay within a descriptor, and PDT_OOB_OFFSET beyond both
        long long reach = (256 / ((long long)Hg * Wg) + 2) * Hi * Wi * pix_stride * 2;
        long long reach2 = 256LL * (pix_stride2 > pix_stride ? pix_stride2 : pix_stride) * 2;
        if (reach >= 0x7fff0000LL || reach2 >= 0x7fff0000LL || w_b >= 0x7fff0000LL) return -7;
        p.in_bytes = (unsigned long long)in_b;
        p.wgt_bytes = (unsigned)w_b;
    }
    for (int i = 0; i < T; ++i)
        p.taps[i] = ((int)offy[i] & 0xff) | (((int)offx[i] & 0xff) << 8) | ((int)widx[i] << 16);
    for (int i = T; i < DMA_MAX_TAPS; ++i) p.taps[i] = 0;
    const bool bs = bstat_y != nullptr;
    // Tiles: 3 = 128 x 128 (2 x 2 waves, K slices of 64, ring of 2), 4 = 128 x 64 (4 x 1 waves, ring of 2: 50 KB of LDS, so the plain
    // variant -- 164 VGPRs -- runs three workgroups per CU; a ring of 3 measured 5-8 % slower on the 64-channel 3x3 layers).  Tiles 1 / 2
    // named ring-of-4 pipelines on K slices of 32, which measured slower everywhere; they now select 3 / 4.
    if (tile == 1 || tile == 2) tile += 2;
    if (tile == 0) tile = (N % 128 == 0) ? 3 : 4;
    if (tile == 3 && N % 128 != 0) tile = 4;
    if (in2 || bias) {
        if (tile == 5) return -11;
        if (tile == 3) return bs ? g1_launch<128, 2, 2, 64, 2, true, true>(p, s) : g1_launch<128, 2, 2, 64, 2, false, true>(p, s);
        return bs ? g1_launch<64, 4, 1, 64, 2, true, true>(p, s) : g1_launch<64, 4, 1, 64, 2, false, true>(p, s);
    }
    switch (tile) {
        case 3: return bs ? g1_launch<128, 2, 2, 64, 2, true>(p, s) : g1_launch<128, 2, 2, 64, 2, false>(p, s);
        case 4: return bs ? g1_launch<64, 4, 1, 64, 2, true>(p, s) : g1_launch<64, 4, 1, 64, 2, false>(p, s);
        case 5: return g1_launch<64, 4, 1, 32, 4, false>(p, s);       // 128 x 64, K slices of 32, ring of 4: the stem
        default: return -6;
    }
}

PDT_API int pdt_conv_dma(const void* in, const void* wgt, void* out, const void* addend, int addend_half,
                         const void* mask_bits, float* stat_acc, const float* stat_shift, const void* bstat_y,
                         const float* bstat_mean, const float* bstat_scale, const float* bstat_shift, long long M, int Hi,
                         int Wi, int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS, int oy0, int ox0,
                         int N, int wrow, int T, const signed char* offy, const signed char* offx,
                         const unsigned short* widx, int tile, int sums_only, hipStream_t s) {
    return pdt_conv_dma_ex(in, wgt, out, addend, addend_half, mask_bits, stat_acc, stat_shift, bstat_y, bstat_mean,
                           bstat_scale, bstat_shift, M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0, ox0, N, wrow,
                           T, offy, offx, widx, tile, nullptr, 0, 0, nullptr, 0, nullptr, sums_only, s);
}
