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
// loads are 16 B per lane, staged through registers into double-buffered LDS (loads for step
// k+1 are issued before the MFMAs of step k and written to LDS after them: one barrier per
// step).  LDS rows are BK bf16 long; the 16-byte chunk index is XOR-swizzled with the row so
// the ds_read_b128 fragment reads of the 16x16x32 operand layout (lane l: row l&15, k-chunk
// l>>4) are bank-conflict free.  Workgroup ids are remapped so that each XCD's L2 sees a
// contiguous range of tiles that share activation rows.
//
// Epilogue (through an fp32 LDS image of the tile, so global stores are full 16-B vectors):
//   * optional per-channel bias,
//   * optional "out += addend [* (mask > 0)]"  (residual-gradient accumulation of a block),
//   * optional BatchNorm statistics: per-channel sum(v - s) and sum((v - s)^2) over the tile's
//     valid rows, added with float atomics to a replicated accumulator (see reduce.hip).
#include "common.h"

#define NREP 16
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
    long long M;
    int Hi, Wi, pix_stride, Ci;
    int Hg, Wg, IS;
    int Ho, Wo, OS, oy0, ox0;
    int N, ldo, wrow, T;
    int simple_out;
    signed char offy[MAX_TAPS];
    signed char offx[MAX_TAPS];
    unsigned short widx[MAX_TAPS];
};

template <int BK>
__device__ __forceinline__ int swz_chunk(int row, int chunk) {
    if (BK == 64) return chunk ^ ((row >> 1) & 7);
    return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);   // BK == 32
}

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p) {
    constexpr int CH = BK / 8;            // 16-byte chunks per LDS row
    constexpr int RPP = 256 / CH;         // rows covered by one pass of the 256 threads
    constexpr int A_IT = BM / RPP;
    constexpr int B_IT = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int KS = BK / 32;
    constexpr int RB = BK * 2;            // LDS row bytes
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
    constexpr int CLD = BN + 4;           // fp32 staging row stride
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);
    constexpr int C_BYTES = BM * CLD * 4;
    constexpr int LDS_BYTES = STAGE_BYTES > C_BYTES ? STAGE_BYTES : C_BYTES;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/thread mismatch");

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + 2 * 4 * BN * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile id: blocks b, b+8, ... share an XCD; give each XCD a contiguous range.
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int n_tiles_m = (int)((p.M + BM - 1) / BM);
    const int nwg = n_tiles_m * n_tiles_n;
    int wgid;
    {
        int orig = blockIdx.x;
        int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_n = wgid % n_tiles_n;
    const int tile_m = wgid / n_tiles_n;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread staging geometry
    const int chunk = tid % CH;
    const int row_in_pass = tid / CH;
    int a_base[A_IT], a_y[A_IT], a_x[A_IT];
    bool a_ok[A_IT];
    const int HgWg = p.Hg * p.Wg;
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        long long m = m0 + row_in_pass + j * RPP;
        a_ok[j] = m < p.M;
        unsigned mm = a_ok[j] ? (unsigned)m : 0u;          // M < 2^31 (checked by the launcher)
        int b = (int)(mm / (unsigned)HgWg);
        int rem = (int)(mm - (unsigned)b * (unsigned)HgWg);
        int y = (int)((unsigned)rem / (unsigned)p.Wg), x = rem - y * p.Wg;
        a_base[j] = b * p.Hi * p.Wi;
        a_y[j] = y * p.IS;
        a_x[j] = x * p.IS;
    }
    long long b_off[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        int n = n0 + row_in_pass + j * RPP;
        b_ok[j] = n < p.N;
        b_off[j] = (long long)(b_ok[j] ? n : 0) * p.wrow + chunk * 8;
    }

    bf16x8 a_reg[A_IT], b_reg[B_IT];
    const int kc = p.Ci / BK;             // K steps per tap
    const int nk = p.T * kc;

    auto load_tiles = [&](int t, int c0) {
        const int oy = p.offy[t], ox = p.offx[t];
        const long long wofs = (long long)p.widx[t] * p.Ci + c0;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int iy = a_y[j] + oy, ix = a_x[j] + ox;
            bool ok = a_ok[j] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            if (ok) {
                long long off = (long long)(a_base[j] + iy * p.Wi + ix) * p.pix_stride + c0 + chunk * 8;
                a_reg[j] = ld_bf16x8(p.in + off);
            } else {
                a_reg[j] = zero_bf16x8();
            }
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (b_ok[j]) b_reg[j] = ld_bf16x8(p.wgt + b_off[j] + wofs);
            else b_reg[j] = zero_bf16x8();
        }
    };
    auto store_tiles = [&](int buf) {
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

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;

    if (nk > 0) {
        int t = 0, cstep = 0;
        load_tiles(0, 0);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int ks = 0; ks < nk; ++ks) {
            int nt = t, ncs = cstep + 1;
            if (ncs == kc) { ncs = 0; nt = t + 1; }
            const bool has_next = ks + 1 < nk;
            if (has_next) load_tiles(nt, ncs * BK);

            const unsigned char* As = smem + cur * (A_BYTES + B_BYTES);
            const unsigned char* Bs = As + A_BYTES;
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            if (has_next) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
            t = nt;
            cstep = ncs;
        }
    }

    // ---- epilogue: accumulators -> fp32 LDS image -> (stats) -> fused output ------------------
    float* Cs = reinterpret_cast<float*>(smem);
    float* red = reinterpret_cast<float*>(smem + LDS_BYTES);   // [2][4][BN] partials for stats
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = wm * WTM + i * 16 + fq * 4 + r;
                int col = wn * WTN + j * 16 + frow;
                Cs[row * CLD + col] = acc[i][j][r];
            }
    __syncthreads();

    long long rows_left = p.M - m0;
    const int rows_valid = rows_left < BM ? (int)rows_left : BM;

    if (p.stat_acc) {
        constexpr int PARTS = 256 / BN;          // threads per column
        constexpr int PARTS_C = PARTS > 0 ? PARTS : 1;
        if (BN <= 256) {
            int col = tid % BN, part = tid / BN;
            int n = n0 + col;
            float sh = (p.stat_shift && n < p.N) ? p.stat_shift[n] : 0.0f;
            float s0 = 0.f, s1 = 0.f;
            for (int r = part; r < rows_valid; r += PARTS_C) {
                float d = Cs[r * CLD + col] - sh;
                s0 += d;
                s1 += d * d;
            }
            red[(0 * 4 + part) * BN + col] = s0;
            red[(1 * 4 + part) * BN + col] = s1;
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            int which = i / BN, col = i % BN;
            int n = n0 + col;
            if (n < p.N) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < PARTS_C; ++q) s += red[(which * 4 + q) * BN + col];
                atomicAdd(p.stat_acc + ((size_t)(blockIdx.x % NREP) * 2 + which) * p.N + n, s);
            }
        }
    }

    constexpr int NV = BN / 8;                   // 8-channel vectors per tile row
    for (int v = tid; v < BM * NV; v += 256) {
        int row = v / NV, cv = v % NV;
        if (row >= rows_valid) continue;
        int n = n0 + cv * 8;
        if (n >= p.ldo) continue;                // ldo is padded to a multiple of 8 by the host
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
        const float* src = Cs + row * CLD + cv * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(src);
        f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
        float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (p.bias) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] += (n + k < p.N) ? p.bias[n + k] : 0.0f;
        }
        long long o = orow * p.ldo + n;
        if (p.addend) {
            bf16x8 a = ld_bf16x8(p.addend + o);
            if (p.mask) {
                bf16x8 mk = ld_bf16x8(p.mask + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] += bf2f(mk[k]) > 0.0f ? bf2f(a[k]) : 0.0f;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] += bf2f(a[k]);
            }
        }
        bf16x8 ov;
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[k] = f2bf(f[k]);
        st_bf16x8(p.out + o, ov);
    }
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_igemm(const IgemmParams& p, hipStream_t s) {
    int n_tiles_n = (p.N + BN - 1) / BN;
    long long n_tiles_m = (p.M + BM - 1) / BM;
    long long nwg = n_tiles_m * n_tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffLL) return -2;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WM, WN>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return pdt_launch_status();
}

// tile: 0 = auto, 1 = 128x128x64, 2 = 128x64x64, 3 = 64x64x64, 4 = 128x64x32 (stem), 5 = 64x128x64
PDT_API int pdt_conv_igemm(const void* in, const void* wgt, void* out, const void* addend, const void* mask,
                           const float* bias, float* stat_acc, const float* stat_shift, long long M, int Hi, int Wi,
                           int pix_stride, int Ci, int Hg, int Wg, int IS, int Ho, int Wo, int OS, int oy0, int ox0,
                           int N, int ldo, int wrow, int T, const signed char* offy, const signed char* offx,
                           const unsigned short* widx, int tile, hipStream_t s) {
    if (T < 0 || T > MAX_TAPS) return -3;
    if (M <= 0) return 0;
    if (M >= 0x7fffffffLL) return -2;
    if (ldo % 8 != 0 || pix_stride % 4 != 0) return -4;
    IgemmParams p;
    p.in = (const bf16*)in; p.wgt = (const bf16*)wgt; p.out = (bf16*)out;
    p.addend = (const bf16*)addend; p.mask = (const bf16*)mask; p.bias = bias;
    p.stat_acc = stat_acc; p.stat_shift = stat_shift;
    p.M = M; p.Hi = Hi; p.Wi = Wi; p.pix_stride = pix_stride; p.Ci = Ci;
    p.Hg = Hg; p.Wg = Wg; p.IS = IS; p.Ho = Ho; p.Wo = Wo; p.OS = OS; p.oy0 = oy0; p.ox0 = ox0;
    p.N = N; p.ldo = ldo; p.wrow = wrow; p.T = T;
    p.simple_out = (OS == 1 && oy0 == 0 && ox0 == 0 && Hg == Ho && Wg == Wo) ? 1 : 0;
    for (int i = 0; i < T; ++i) { p.offy[i] = offy[i]; p.offx[i] = offx[i]; p.widx[i] = widx[i]; }
    for (int i = T; i < MAX_TAPS; ++i) { p.offy[i] = 0; p.offx[i] = 0; p.widx[i] = 0; }
    if (T > 0 && Ci % 32 != 0) return -5;
    if (tile == 0) {
        if (T > 0 && Ci % 64 != 0) tile = 4;
        else if (N <= 64) tile = 2;
        else {
            long long wg128 = ((M + 127) / 128) * ((N + 127) / 128);
            tile = (wg128 >= 512) ? 1 : 3;
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
