// Implicit-GEMM convolution for gfx950 (MI355X): forward and data-gradient, NHWC bf16 in,
// fp32 accumulate on the matrix cores (v_mfma_f32_16x16x32_bf16), bf16 out.
//
//   out[m][n] = sum_{t < T} sum_{c < Ci} in[pix(m, t)][c] * wgt[n][widx(t)*Ci + c]   (+ epilogue)
//
// A GEMM row m is a point (b, y, x) of a per-image grid Hg x Wg.  Tap t reads the input pixel
// (y*IS + offy[t], x*IS + offx[t]) (zero outside the image) and the row is written to output
// pixel (y*OS + oy0, x*OS + ox0).  With the tap table this one kernel covers
//   * forward convolution, any R x S / stride / pad            (IS = stride, OS = 1),
//   * the 7x7/s2 stem on a zero-haloed NHWC4 image: 7 taps of 32 contiguous values each,
//   * data gradient, stride 1                                   (flipped taps, IS = OS = 1),
//   * data gradient, stride 2, one launch per output-parity class (IS = 1, OS = 2): each class
//     only multiplies the taps that really contribute, so no MFMA work is wasted on zeros,
//   * the classifier GEMM and its data gradient                 (1x1 "images").
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step BK (one tap, BK channels).  Global
// loads are 16 B per lane, staged through two register sets into double-buffered LDS: the
// loads for step k+2 are issued before the MFMAs of step k and written to LDS after the MFMAs
// of step k+1 (two compute phases of latency cover, one barrier per step).  LDS rows are BK bf16 long; the 16-byte chunk index is XOR-swizzled with the row so
// the ds_read_b128 fragment reads of the 16x16x32 operand layout (lane l: row l&15, k-chunk
// l>>4) are bank-conflict free.  Workgroups are persistent (one grid-full of resident blocks,
// each looping over tiles): the first loads of a block's next tile are issued before the
// epilogue of the current one, which matters most for the 1x1 layers with one or two K steps.
// Tile ids are remapped so that each XCD's L2 sees a contiguous range of tiles that share
// activation rows.
//
// The MFMA operands are swapped (weights on the "A" side), so a lane's four accumulator
// registers are four CONSECUTIVE CHANNELS of one pixel.  Epilogue:
//   * optional per-channel bias (in registers),
//   * the tile goes to LDS as bf16 with 8-byte writes and leaves as full 16-byte vectors,
//   * optional "out = (out + addend) * (mask > 0)": residual-gradient accumulation of a block
//     and the ReLU backward of the tensor the gradient belongs to, so downstream BatchNorm
//     backward passes need no mask tensor,
//   * optional BatchNorm statistics: in the output loop every thread keeps meeting the same 8
//     channels, so per-channel sum(v - s) and sum((v - s)^2) over the tile's valid rows are
//     accumulated for free on the values being stored, reduced across threads through LDS and
//     added with float atomics to a replicated accumulator (see reduce.hip).
#include "common.h"

#define MAX_TAPS 52

struct IgemmParams {
    const bf16* in;
    const bf16* wgt;
    bf16* out;
    const bf16* addend;
    const bf16* mask;
    const float* bias;
    float* stat_acc;
    const float* stat_shift;
    // BatchNorm-BACKWARD sums of the layer that consumes this (data-gradient) output g: with
    // bstat_y (that layer's raw conv output, same shape as the output here) the statistics slot
    // accumulates [sum(dy), sum(dy * (y - mean))] instead of the forward moments, dy = g, or
    // g * [y*scale + shift > 0] when bstat_scale is given (the layer's own ReLU, recomputed).
    const bf16* bstat_y;
    const float* bstat_mean;
    const float* bstat_scale;
    const float* bstat_shift;
    long long M;
    unsigned in_bytes, wgt_bytes;      // sizes of the input / weight tensors (buffer descriptors)
    int Hi, Wi, pix_stride, Ci;
    int Hg, Wg, IS;
    int Ho, Wo, OS, oy0, ox0;
    int N, ldo, wrow, T;
    int simple_out;
    int mask_bits;                     // mask points to packed ReLU bits (1 byte per 8 channels), not to a tensor
    // addend is a half-resolution tensor [B][Ho/2][Wo/2][ldo] that exists at the output pixels with two
    // even coordinates only (the data gradient of a stride-2 1x1 projection shortcut: three quarters of
    // its full-resolution form are zeros that would be written and read back for nothing)
    int addend_half;
    int nrep;                          // replica rows of the statistics accumulator (see common.h)
    // one 32-bit word per tap: offy (int8) | offx (int8) << 8 | widx (uint16) << 16.  Sub-dword
    // entries would be fetched with VECTOR loads (no scalar byte loads on gfx9), whose waits
    // drain the tile prefetches; a dword indexed by a wave-uniform tap is one s_load_dword.
    int taps[MAX_TAPS];
};

__device__ __forceinline__ void unpack_tap(int w, int& oy, int& ox, int& widx) {
    oy = (int)(signed char)(w & 0xff);
    ox = (int)(signed char)((w >> 8) & 0xff);
    widx = (w >> 16) & 0xffff;
}

template <int BK>
__device__ __forceinline__ int swz_chunk(int row, int chunk) {
    if (BK == 64) return chunk ^ ((row >> 1) & 7);
    return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);   // BK == 32
}

template <int BM, int BN, int BK, int WM, int WN, bool BSTAT>
__global__ __launch_bounds__(256, ((BM * BN == 128 * 64 && !BSTAT) ? 3 : 2)) void igemm_kernel(const IgemmParams p) {
    constexpr int CH = BK / 8;            // 16-byte chunks per LDS row
    constexpr int RPP = 256 / CH;         // rows covered by one pass of the 256 threads
    constexpr int A_IT = BM / RPP;
    constexpr int B_IT = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int KS = BK / 32;
    constexpr int RB = BK * 2;            // LDS row bytes
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
    constexpr int CLD = BN + 8;           // bf16 staging row stride (elements); rows stay 16-B aligned
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);
    constexpr int C_BYTES = BM * CLD * 2;
    constexpr int NV = BN / 8;            // 8-channel vectors per tile row
    constexpr int TR = 256 / NV;          // threads that share one channel vector in the output loop
    constexpr int RED_BYTES = TR * 2 * BN * 4;
    constexpr int EPI_BYTES = C_BYTES + RED_BYTES;
    constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/thread mismatch");

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent workgroups: block b processes virtual tile ids b, b + gridDim, ...
    // XCD-aware remap of the virtual id: ids v, v+8, ... run on one XCD (round-robin dispatch);
    // give each XCD a contiguous range of tiles so tiles that share activation rows share an L2.
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int n_tiles_m = (int)((p.M + BM - 1) / BM);
    const int nwg = n_tiles_m * n_tiles_n;
    const int HgWg = p.Hg * p.Wg;

    // ---- per-thread staging geometry of the tile whose loads are being issued
    const int chunk = tid % CH;
    const int row_in_pass = tid / CH;
    int a_base[A_IT], a_y[A_IT], a_x[A_IT];
    bool a_ok[A_IT];
    unsigned b_off[B_IT];               // byte offset of the weight row, or PDT_OOB_OFFSET
    long long ld_m0 = 0;
    int ld_n0 = 0;
    auto setup_tile = [&](int v) {
        int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        const int tile_n = wgid % n_tiles_n;
        const int tile_m = wgid / n_tiles_n;
        ld_m0 = (long long)tile_m * BM;
        ld_n0 = tile_n * BN;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            long long m = ld_m0 + row_in_pass + j * RPP;
            a_ok[j] = m < p.M;
            unsigned mm = a_ok[j] ? (unsigned)m : 0u;          // M < 2^31 (checked by the launcher)
            int b = (int)(mm / (unsigned)HgWg);
            int rem = (int)(mm - (unsigned)b * (unsigned)HgWg);
            int y = (int)((unsigned)rem / (unsigned)p.Wg), x = rem - y * p.Wg;
            a_base[j] = b * p.Hi * p.Wi;
            a_y[j] = y * p.IS;
            a_x[j] = x * p.IS;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int n = ld_n0 + row_in_pass + j * RPP;
            b_off[j] = n < p.N ? (unsigned)(n * p.wrow + chunk * 8) * 2u : PDT_OOB_OFFSET;
        }
    };
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t wgt_rsrc = make_rsrc(p.wgt, p.wgt_bytes);

    const int kc = p.Ci / BK;             // K steps per tap
    const int nk = p.T * kc;

    // Two register sets: the loads of K step k+2 are issued before the MFMAs of step k and are
    // only written to LDS at the end of step k+1, so they have two compute phases to land.
    bf16x8 a_r0[A_IT], b_r0[B_IT], a_r1[A_IT], b_r1[B_IT];

    auto load_tiles = [&](bf16x8 (&a_reg)[A_IT], bf16x8 (&b_reg)[B_IT], int ks) {
        const int t = __builtin_amdgcn_readfirstlane(ks / kc);
        const int c0 = (ks - t * kc) * BK;
        int oy, ox, wi;
        unpack_tap(p.taps[t], oy, ox, wi);
        const unsigned wofs = (unsigned)(wi * p.Ci + c0) * 2u;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int iy = a_y[j] + oy, ix = a_x[j] + ox;
            bool ok = a_ok[j] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            unsigned off = (unsigned)((a_base[j] + iy * p.Wi + ix) * p.pix_stride + c0 + chunk * 8) * 2u;
            a_reg[j] = buf_ld_bf16x8(in_rsrc, ok ? off : PDT_OOB_OFFSET);
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            b_reg[j] = buf_ld_bf16x8(wgt_rsrc, b_off[j] == PDT_OOB_OFFSET ? PDT_OOB_OFFSET : b_off[j] + wofs);
    };
    auto store_tiles = [&](const bf16x8 (&a_reg)[A_IT], const bf16x8 (&b_reg)[B_IT], int buf) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int row = row_in_pass + j * RPP;
            *reinterpret_cast<bf16x8*>(As + row * RB + (swz_chunk<BK>(row, chunk) << 4)) = a_reg[j];
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int row = row_in_pass + j * RPP;
            *reinterpret_cast<bf16x8*>(Bs + row * RB + (swz_chunk<BK>(row, chunk) << 4)) = b_reg[j];
        }
    };

    // acc[i][j]: operands are swapped (weights feed the MFMA "A" side), so the 16x16 result has
    // channels on its rows: lane (fq, frow) holds pixel i*16+frow, channels j*16 + fq*4 + {0..3}.
    f32x4 acc[MI][NI];
    const int frow = lane & 15, fq = lane >> 4;

    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* Bs = As + A_BYTES;
        if constexpr (MI < 4) {
            // few pixel fragments per wave: nothing to pipeline, let the compiler schedule freely
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                bf16x8 af[MI], bfr[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int row = wm * WTM + i * 16 + frow;
                    af[i] = *reinterpret_cast<const bf16x8*>(As + row * RB + (swz_chunk<BK>(row, kk * 4 + fq) << 4));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int row = wn * WTN + j * 16 + frow;
                    bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * RB + (swz_chunk<BK>(row, kk * 4 + fq) << 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            // The weight fragments of this k-slice and the first pixel fragment are read up front;
            // pixel fragment i+1 is requested BEFORE the MFMAs of fragment i are issued (two
            // fragment registers alternate), so an LDS read has NI MFMAs to land behind instead of
            // stalling the wave once per fragment.
            bf16x8 af[2], bfr[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int row = wn * WTN + j * 16 + frow;
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * RB + (swz_chunk<BK>(row, kk * 4 + fq) << 4));
            }
            {
                int row = wm * WTM + frow;
                af[0] = *reinterpret_cast<const bf16x8*>(As + row * RB + (swz_chunk<BK>(row, kk * 4 + fq) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (i + 1 < MI) {
                    int row = wm * WTM + (i + 1) * 16 + frow;
                    af[(i + 1) & 1] =
                        *reinterpret_cast<const bf16x8*>(As + row * RB + (swz_chunk<BK>(row, kk * 4 + fq) << 4));
                }
                if (kk > 0 && i == 0) {
                    // the region since the previous pin holds the last MFMAs of the previous k-slice
                    // and this slice's first reads: alternate them instead of reading after the MFMAs
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep that read ahead of the MFMAs below
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i & 1], acc[i][j], 0, 0, 0);
            }
        }
    };

    bf16* Cs = reinterpret_cast<bf16*>(smem);
    float* red = reinterpret_cast<float*>(smem + C_BYTES);     // [TR][2][BN]
    const bool want_stats = p.stat_acc != nullptr;

    // BatchNorm partial sums of the output loop's 8 channels (thread t: channel vector t % NV)
    const int cv = tid % NV;
    // BSTAT is a template parameter so the plain instantiations do not carry its registers
    constexpr bool bwd_stats = BSTAT;
    const bool bwd_mask = BSTAT && p.bstat_scale != nullptr;
    float sh[8], s0[8], s1[8], msc[BSTAT ? 8 : 1], msh[BSTAT ? 8 : 1];
    int stat_n0 = -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sh[k] = 0.f;
        s0[k] = 0.f;
        s1[k] = 0.f;
        if (BSTAT) {
            msc[k] = 0.f;
            msh[k] = 1.f;
        }
    }

    int v = blockIdx.x;                       // the launcher guarantees gridDim.x <= nwg
    setup_tile(v);
    if (nk > 0) {
        load_tiles(a_r0, b_r0, 0);
        if (nk > 1) load_tiles(a_r1, b_r1, 1);
    }
    while (true) {
        const long long m0 = ld_m0;           // the tile being computed now
        const int n0 = ld_n0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (nk > 0) {
            store_tiles(a_r0, b_r0, 0);
            __syncthreads();
            for (int ks = 0; ks < nk; ks += 2) {
                // even step: LDS[0] holds step ks, set 1 holds step ks+1, set 0 is free
                if (ks + 2 < nk) load_tiles(a_r0, b_r0, ks + 2);
                compute(0);
                if (ks + 1 < nk) store_tiles(a_r1, b_r1, 1);
                __syncthreads();
                if (ks + 1 >= nk) break;
                // odd step: LDS[1] holds step ks+1, set 0 holds step ks+2, set 1 is free
                if (ks + 3 < nk) load_tiles(a_r1, b_r1, ks + 3);
                compute(1);
                if (ks + 2 < nk) store_tiles(a_r0, b_r0, 0);
                __syncthreads();
            }
        }

        // Cross-tile pipelining: issue the first loads of this block's NEXT tile now, so their
        // latency is covered by the epilogue below.
        const int vn = v + gridDim.x;
        const bool more = vn < nwg;
        if (more) {
            setup_tile(vn);
            if (nk > 0) {
                load_tiles(a_r0, b_r0, 0);
                if (nk > 1) load_tiles(a_r1, b_r1, 1);
            }
        }

        // ---- epilogue -----------------------------------------------------------------------
        long long rows_left = p.M - m0;
        const int rows_valid = rows_left < BM ? (int)rows_left : BM;

        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int n = n0 + wn * WTN + j * 16 + fq * 4 + r;
                    float bv = n < p.N ? p.bias[n] : 0.0f;
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j][r] += bv;
                }
        }

        // accumulators -> bf16 tile image in LDS (8-byte writes: 4 consecutive channels of a pixel)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int row = wm * WTM + i * 16 + frow;
                int col = wn * WTN + j * 16 + fq * 4;
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = f2bf(acc[i][j][r]);
                *reinterpret_cast<bf16x4*>(Cs + row * CLD + col) = pk;
            }
        __syncthreads();

        // Output loop: thread t always handles channel vector t % NV, rows t / NV + k*TR.  The BN
        // statistics of those 8 channels are accumulated on the way (from the bf16-rounded values
        // that are stored: what a separate statistics pass over the output would see).
        const int n = n0 + cv * 8;
        if (want_stats && stat_n0 != n0) {       // accumulators are empty whenever n0 changes
            stat_n0 = n0;
            const float* center = bwd_stats ? p.bstat_mean : p.stat_shift;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = n + k < p.N;
                sh[k] = (center && in) ? center[n + k] : 0.0f;
                if (BSTAT) {
                    if (bwd_mask) {
                        msc[k] = in ? p.bstat_scale[n + k] : 0.0f;
                        msh[k] = in ? p.bstat_shift[n + k] : 1.0f;
                    }
                }
            }
        }
        // element index of the addend for GEMM row m whose output index is o; -1: no addend at that pixel
        auto addend_index = [&](long long m, long long o) -> long long {
            if (!p.addend_half) return o;
            unsigned mu = (unsigned)m;
            int b = (int)(mu / (unsigned)HgWg);
            int rem = (int)(mu - (unsigned)b * (unsigned)HgWg);
            int y = (int)((unsigned)rem / (unsigned)p.Wg), x = rem - y * p.Wg;
            int yo = y * p.OS + p.oy0, xo = x * p.OS + p.ox0;
            if ((yo | xo) & 1) return -1;
            return (((long long)b * (p.Ho >> 1) + (yo >> 1)) * (p.Wo >> 1) + (xo >> 1)) * p.ldo + n;
        };
        if (n < p.ldo) {                         // ldo is a multiple of 8 (checked by the launcher)
            if constexpr (!BSTAT) {
                // plain instantiations (forward, data gradients without fused statistics of the
                // consumer): at most two global loads per row, one row per trip
                for (int row = tid / NV; row < rows_valid; row += TR) {
                    long long m = m0 + row;
                    long long orow;
                    if (p.simple_out) {
                        orow = m;
                    } else {
                        unsigned mu = (unsigned)m;
                        int b = (int)(mu / (unsigned)HgWg);
                        int rem = (int)(mu - (unsigned)b * (unsigned)HgWg);
                        int y = (int)((unsigned)rem / (unsigned)p.Wg), x = rem - y * p.Wg;
                        orow = ((long long)b * p.Ho + (y * p.OS + p.oy0)) * p.Wo + (x * p.OS + p.ox0);
                    }
                    bf16x8 ov = *reinterpret_cast<const bf16x8*>(Cs + row * CLD + cv * 8);
                    long long o = orow * p.ldo + n;
                    if (p.addend) {
                        const long long ai = addend_index(m, o);
                        if (ai >= 0) {
                            bf16x8 a = ld_bf16x8(p.addend + ai);
#pragma unroll
                            for (int k = 0; k < 8; ++k) ov[k] = f2bf(bf2f(ov[k]) + bf2f(a[k]));
                        }
                    }
                    if (p.mask) {
                        if (p.mask_bits) {
                            unsigned mb = reinterpret_cast<const unsigned char*>(p.mask)[o >> 3];
#pragma unroll
                            for (int k = 0; k < 8; ++k) ov[k] = ((mb >> k) & 1u) ? ov[k] : f2bf(0.0f);
                        } else {
                            bf16x8 mk = ld_bf16x8(p.mask + o);
#pragma unroll
                            for (int k = 0; k < 8; ++k) ov[k] = bf2f(mk[k]) > 0.0f ? ov[k] : f2bf(0.0f);
                        }
                    }
                    st_stream_bf16x8(p.out + o, ov);
                    if (want_stats) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float d = bf2f(ov[k]) - sh[k];
                            s0[k] += d;
                            s1[k] += d * d;
                        }
                    }
                }
            } else {
            // Backward-statistics instantiations: EU rows per trip, all their global loads (addend,
            // mask, BatchNorm input) issued before the first result is used, so a thread keeps EU
            // times the bytes in flight -- the epilogue of the 1x1 data gradients moves 3-4x the
            // bytes of their main loop (measured: -0.95 ms per ResNet-50 step at batch 512 with 2
            // rows; 3 rows spill registers and lose 4 ms).
            // (These variants read the ReLU mask in packed form only: the tensor form costs 8 more
            // registers, which the 128x128 tile does not have.)
            constexpr int EU = 2;
            for (int row0 = tid / NV; row0 < rows_valid; row0 += EU * TR) {
                long long oo[EU];
                bool live[EU];
                bf16x8 ov[EU], av[EU], yv[EU];
                unsigned mb[EU];
#pragma unroll
                for (int e = 0; e < EU; ++e) {
                    int row = row0 + e * TR;
                    live[e] = row < rows_valid;
                    if (!live[e]) row = row0;               // harmless duplicate of the first row
                    long long m = m0 + row;
                    long long orow;
                    if (p.simple_out) {
                        orow = m;
                    } else {
                        unsigned mu = (unsigned)m;
                        int b = (int)(mu / (unsigned)HgWg);
                        int rem = (int)(mu - (unsigned)b * (unsigned)HgWg);
                        int y = (int)((unsigned)rem / (unsigned)p.Wg), x = rem - y * p.Wg;
                        orow = ((long long)b * p.Ho + (y * p.OS + p.oy0)) * p.Wo + (x * p.OS + p.ox0);
                    }
                    oo[e] = orow * p.ldo + n;
                    if (p.addend) {                         // last use of the incoming gradient: streaming load
                        const long long ai = addend_index(m, oo[e]);
                        av[e] = ai >= 0 ? ld_stream_bf16x8(p.addend + ai) : zero_bf16x8();
                    }
                    if (bwd_stats) yv[e] = ld_bf16x8(p.bstat_y + oo[e]);
                    if (p.mask) mb[e] = reinterpret_cast<const unsigned char*>(p.mask)[oo[e] >> 3];
                }
#pragma unroll
                for (int e = 0; e < EU; ++e) {
                    // the tile row comes out of LDS only now: it would occupy registers for nothing
                    // while the global loads above are in flight
                    const int row = live[e] ? row0 + e * TR : row0;
                    ov[e] = *reinterpret_cast<const bf16x8*>(Cs + row * CLD + cv * 8);
                    if (p.addend) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) ov[e][k] = f2bf(bf2f(ov[e][k]) + bf2f(av[e][k]));
                    }
                    if (p.mask) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) ov[e][k] = ((mb[e] >> k) & 1u) ? ov[e][k] : f2bf(0.0f);
                    }
                    if (!live[e]) continue;
                    st_stream_bf16x8(p.out + oo[e], ov[e]);
                    if (bwd_stats) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float yy = bf2f(yv[e][k]);
                            float dy = bf2f(ov[e][k]);
                            if (BSTAT) {
                                if (bwd_mask) dy = (yy * msc[k] + msh[k]) > 0.0f ? dy : 0.0f;
                            }
                            s0[k] += dy;
                            s1[k] += dy * (yy - sh[k]);
                        }
                    } else if (want_stats) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float d = bf2f(ov[e][k]) - sh[k];
                            s0[k] += d;
                            s1[k] += d * d;
                        }
                    }
                }
            }
            }
        }
        // The partial sums stay in registers while this block's next tile covers the same
        // channels (the usual case: the tile stride of a block is a multiple of the channel-tile
        // count), so the LDS reduction and the atomics run once per block, not once per tile.
        if (want_stats && (!more || ld_n0 != n0)) {
            const int tr = tid / NV;
            float* r0 = red + (tr * 2 + 0) * BN + cv * 8;
            float* r1 = red + (tr * 2 + 1) * BN + cv * 8;
            *reinterpret_cast<f32x4*>(r0) = f32x4{s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<f32x4*>(r0 + 4) = f32x4{s0[4], s0[5], s0[6], s0[7]};
            *reinterpret_cast<f32x4*>(r1) = f32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4*>(r1 + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
            __syncthreads();
            for (int i = tid; i < 2 * BN; i += 256) {
                int which = i / BN, col = i % BN;
                if (n0 + col < p.N) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < TR; ++q) sum += red[(q * 2 + which) * BN + col];
                    atomicAdd(p.stat_acc + ((size_t)(blockIdx.x % p.nrep) * 2 + which) * p.N + n0 + col, sum);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s0[k] = 0.f;
                s1[k] = 0.f;
            }
        }
        if (!more) break;
        v = vn;
        __syncthreads();          // the epilogue's LDS reads are done before the next tile is staged
    }
}

// Resident workgroups of one instantiation on the whole chip (queried once, then cached).
template <int BM, int BN, int BK, int WM, int WN, bool BSTAT>
static int resident_blocks() {
    static int cached = 0;
    if (cached > 0) return cached;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 512;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, igemm_kernel<BM, BN, BK, WM, WN, BSTAT>, 256, 0) != hipSuccess ||
        per_cu <= 0)
        per_cu = 2;
    cached = per_cu * cus;
    return cached;
}

template <int BM, int BN, int BK, int WM, int WN, bool BSTAT>
static int launch_igemm_t(const IgemmParams& p, hipStream_t s) {
    int n_tiles_n = (p.N + BN - 1) / BN;
    long long n_tiles_m = (p.M + BM - 1) / BM;
    long long nwg = n_tiles_m * n_tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return -2;
    long long grid = resident_blocks<BM, BN, BK, WM, WN, BSTAT>();
    if (grid > nwg) grid = nwg;
    if (g_pdt_deterministic && grid > p.nrep) grid = p.nrep;      // one statistics row per workgroup
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WM, WN, BSTAT>), dim3((unsigned)grid), dim3(256), 0, s, p);
    return pdt_launch_status();
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_igemm(const IgemmParams& p, hipStream_t s) {
    return p.bstat_y ? launch_igemm_t<BM, BN, BK, WM, WN, true>(p, s) : launch_igemm_t<BM, BN, BK, WM, WN, false>(p, s);
}

// tile: 0 = auto, 1 = 128x128x64, 2 = 128x64x64, 3 = 64x64x64, 4 = 128x64x32 (stem), 5 = 64x128x64
static int conv_igemm_impl(const void* in, const void* wgt, void* out, const void* addend, int addend_half,
                           const void* mask, int mask_bits, const float* bias, float* stat_acc, const float* stat_shift, const void* bstat_y,
                           const float* bstat_mean, const float* bstat_scale, const float* bstat_shift, long long M,
                           int Hi, int Wi, int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS,
                           int oy0, int ox0, int N, int ldo, int wrow, int T, const signed char* offy,
                           const signed char* offx, const unsigned short* widx, int tile, hipStream_t s) {
    if (T < 0 || T > MAX_TAPS) return -3;
    if (bstat_y && (!stat_acc || !bstat_mean || (bstat_scale && !bstat_shift) || ldo != N)) return -8;
    if (M <= 0) return 0;
    if (M >= 0x7fffffffLL) return -2;
    if (ldo % 8 != 0 || pix_stride % 4 != 0) return -4;
    IgemmParams p;
    p.in = (const bf16*)in; p.wgt = (const bf16*)wgt; p.out = (bf16*)out;
    p.addend = (const bf16*)addend; p.mask = (const bf16*)mask; p.bias = bias;
    p.stat_acc = stat_acc; p.stat_shift = stat_shift;
    p.bstat_y = (const bf16*)bstat_y; p.bstat_mean = bstat_mean; p.bstat_scale = bstat_scale; p.bstat_shift = bstat_shift;
    p.M = M; p.Hi = Hi; p.Wi = Wi; p.pix_stride = pix_stride; p.Ci = Ci;
    {
        long long images = M / ((long long)Hg * Wg);
        long long in_b = images * Hi * Wi * pix_stride * 2, w_b = (long long)N * wrow * 2;
        if (in_b >= (long long)PDT_OOB_OFFSET || w_b >= (long long)PDT_OOB_OFFSET) return -7;
        p.in_bytes = (unsigned)in_b;
        p.wgt_bytes = (unsigned)w_b;
    }
    p.Hg = Hg; p.Wg = Wg; p.IS = IS; p.Ho = Ho; p.Wo = Wo; p.OS = OS; p.oy0 = oy0; p.ox0 = ox0;
    p.N = N; p.ldo = ldo; p.wrow = wrow; p.T = T;
    p.addend_half = (addend && addend_half) ? 1 : 0;
    p.nrep = pdt_nrep();
    if (p.addend_half && ((Ho | Wo) & 1)) return -10;
    p.mask_bits = (mask && mask_bits) ? 1 : 0;
    if (p.mask_bits && ldo != N) return -8;        // the bit image is indexed like a dense [rows][N] tensor
    if (bstat_y && mask && !mask_bits) return -9;  // the backward-statistics variants take the mask as bits only
    p.simple_out = (OS == 1 && oy0 == 0 && ox0 == 0 && Hg == Ho && Wg == Wo) ? 1 : 0;
    for (int i = 0; i < T; ++i)
        p.taps[i] = ((int)offy[i] & 0xff) | (((int)offx[i] & 0xff) << 8) | ((int)widx[i] << 16);
    for (int i = T; i < MAX_TAPS; ++i) p.taps[i] = 0;
    if (T > 0 && Ci % 32 != 0) return -5;
    if (tile == 0) {
        if (T > 0 && Ci % 64 != 0) tile = 4;
        else if (N <= 64) tile = 2;
        else {
            // 128x128 tiles as soon as they give every CU one workgroup; below that the 64x64
            // tile spreads the work over more of the chip (measured: 392 tiles of 128x128 beat
            // 1568 of 64x64 by 15-20 % on the 7x7 layers at batch 256)
            long long wg128 = ((M + 127) / 128) * ((N + 127) / 128);
            tile = (wg128 >= 256) ? 1 : 3;
        }
    }
    if (tile != 4 && T > 0 && Ci % 64 != 0) return -5;
    switch (tile) {
        case 1: return launch_igemm<128, 128, 64, 2, 2>(p, s);
        case 2: return launch_igemm<128, 64, 64, 4, 1>(p, s);
        case 3: return launch_igemm<64, 64, 64, 2, 2>(p, s);
        case 4: return launch_igemm<128, 64, 32, 4, 1>(p, s);
        case 5: return launch_igemm<64, 128, 64, 1, 4>(p, s);
        default: return -6;
    }
}

PDT_API int pdt_conv_igemm(const void* in, const void* wgt, void* out, const void* addend, int addend_half,
                           const void* mask, int mask_bits, const float* bias, float* stat_acc, const float* stat_shift, long long M, int Hi, int Wi,
                           int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS, int oy0, int ox0,
                           int N, int ldo, int wrow, int T, const signed char* offy, const signed char* offx,
                           const unsigned short* widx, int tile, hipStream_t s) {
    return conv_igemm_impl(in, wgt, out, addend, addend_half, mask, mask_bits, bias, stat_acc, stat_shift, nullptr, nullptr, nullptr,
                           nullptr, M,
                           Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0, ox0, N, ldo, wrow, T, offy, offx, widx,
                           tile, s);
}

// Data gradient whose epilogue also accumulates the BatchNorm-backward sums of the consuming layer
// (see IgemmParams::bstat_y).
PDT_API int pdt_conv_igemm_bstats(const void* in, const void* wgt, void* out, const void* addend, int addend_half,
                                  const void* mask, int mask_bits, float* stat_acc, const void* bstat_y, const float* bstat_mean,
                                  const float* bstat_scale, const float* bstat_shift, long long M, int Hi, int Wi,
                                  int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS, int oy0,
                                  int ox0, int N, int ldo, int wrow, int T, const signed char* offy,
                                  const signed char* offx, const unsigned short* widx, int tile, hipStream_t s) {
    return conv_igemm_impl(in, wgt, out, addend, addend_half, mask, mask_bits, nullptr, stat_acc, nullptr, bstat_y, bstat_mean, bstat_scale,
                           bstat_shift, M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0, ox0, N, ldo, wrow, T, offy,
                           offx, widx, tile, s);
}
