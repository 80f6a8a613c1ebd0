// BatchNorm backward folded into the 1x1 convolution in front of it (gfx950 / MI355X).
//
// A bottleneck block ends in  y = a . W^T  (1x1 convolution, a: [M][C], W: [N][C], N = 4C)  followed by a BatchNorm
// whose backward is, per channel n,
//
//   d[m][n] = A_n g[m][n] + B_n y[m][n] + K_n        A = scale,  B = -scale * invstd * s1/count,
//                                                    K = -scale * s0/count - B * mean
//
// (s0 = sum g, s1 = sum g * xhat over the batch).  d is the widest tensor of the block (N channels); writing it and
// reading it back twice (data gradient, weight gradient) is a quarter of the block's backward traffic.  Because
// y is itself a product with W, both consumers can be evaluated WITHOUT d and without re-reading y:
//
//   data gradient    d . W        = g . (A o W)  +  a . S  +  bias        S = W^T diag(B) W  [C][C],  bias = K . W
//   weight gradient  d^T . a      = A o (g^T a)  +  B o (W . G)  +  K (x) colsum(a)        G = a^T a  [C][C]
//
// so the C-wide tensor a (a quarter of y's bytes) stands in for y, at the price of two C x C matrices.  The big
// products run in the convolution kernels (conv_dma.hip: second operand + accumulator start values; conv_wgrad.hip:
// per-row output scale, and G as an ordinary weight-gradient launch with a on both sides); this file holds the two
// small kernels around them:
//
//   * bnfold_prepare:  coefficients, the folded weight rows [A o Wt | S] in the data-gradient layout, and bias;
//   * bnfold_wgrad:    dW += B o (W . G) + K (x) colsum(a)   (+ A o T when the product T = g^T a was kept unscaled);
//   * bnfold_sums:     the backward sum s1 = sum g * xhat from T as well: sum_m g y = sum_c W[n][c] T[n][c], so the
//                      kernel that produces g only has to add up g itself (s0) and does not read y either.
//
// Both multiply on the matrix cores (v_mfma_f32_32x32x16_bf16, operands straight from global memory: the matrices
// are a few hundred rows, launch latency is what counts).  G holds sums over ~10^6 pixels, so its bf16 operand is
// split into a high and a low part (two MFMAs, 16 mantissa bits).
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct BnFoldCoef {
    float A, B, K;
};

__device__ __forceinline__ BnFoldCoef bnfold_coef(const float* __restrict__ sums, const float* __restrict__ scale,
                                                  const float* __restrict__ invstd, const float* __restrict__ mean,
                                                  float inv_count, int N, int n) {
    BnFoldCoef c;
    const float k0 = sums[n] * inv_count, k1 = sums[N + n] * inv_count;
    c.A = scale[n];
    c.B = -c.A * k1 * invstd[n];
    c.K = -c.A * k0 - c.B * mean[n];
    return c;
}

// Sum the four waves' 32 x 32 partial tiles through LDS; thread t ends up with elements t, t+256, t+512, t+768 of the
// [reg][lane] image (reg = e / 64, lane = e % 64).
__device__ __forceinline__ void reduce_tiles(const f32x16& acc, float* red, int wave, int lane, float (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + q * 256;
        out[q] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
    }
}

// Accumulator element (reg, lane) of a 32x32 MFMA tile: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31.
__device__ __forceinline__ void tile_rc(int e, int& row, int& col) {
    const int reg = e >> 6, lane = e & 63;
    row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    col = lane & 31;
}

// grid: (C/32)^2 blocks computing 32x32 tiles of S, then C blocks (one per row of Wt) for the scaled copy and bias.
__global__ __launch_bounds__(256) void bnfold_prepare_kernel(const float* __restrict__ sums, const float* __restrict__ scale,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ mean, float inv_count,
                                                              const bf16* __restrict__ wt, bf16* __restrict__ wfold,
                                                              float* __restrict__ bias, float* __restrict__ coef, int C,
                                                              int N) {
    __shared__ float red[4 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = C / 32, nS = tiles * tiles;
    const int ldf = N + C;
    if ((int)blockIdx.x < nS) {
        // S[c'][c] = sum_n (B_n Wt[c'][n]) Wt[c][n]; wave w takes n in [w N/4, (w+1) N/4)
        const int bi = blockIdx.x / tiles, bj = blockIdx.x % tiles;
        const int r = lane & 31, h = lane >> 5;
        const int kq = N / 4;
        const bf16* pa = wt + (size_t)(32 * bi + r) * N + wave * kq + 8 * h;
        const bf16* pb = wt + (size_t)(32 * bj + r) * N + wave * kq + 8 * h;
        const int nb = wave * kq + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int s = 0; s < kq / 16; ++s) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(pb + 16 * s);
            const int n0 = nb + 16 * s;
            bf16x8 as;
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + N + n0 + j);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + n0 + j);
                const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + n0 + j);
#pragma unroll
                for (int k = 0; k < 4; ++k) as[j + k] = f2bf(bf2f(av[j + k]) * (-sc[k] * s1[k] * inv_count * is[k]));
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, bv, acc, 0, 0, 0);
        }
        float out[4];
        reduce_tiles(acc, red, wave, lane, out);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int row, col;
            tile_rc(tid + q * 256, row, col);
            wfold[(size_t)(32 * bi + row) * ldf + N + 32 * bj + col] = f2bf(out[q]);
        }
        return;
    }
    // row c of the data-gradient weights: wfold[c][n] = A_n Wt[c][n]; bias[c] = sum_n K_n Wt[c][n]
    const int c = blockIdx.x - nS;
    float part = 0.f;
    for (int n0 = tid * 8; n0 < N; n0 += 256 * 8) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wt + (size_t)c * N + n0);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const BnFoldCoef f = bnfold_coef(sums, scale, invstd, mean, inv_count, N, n0 + k);
            const float w = bf2f(wv[k]);
            o[k] = f2bf(w * f.A);
            part += w * f.K;
            if (c == 0) {
                coef[n0 + k] = f.A;
                coef[N + n0 + k] = f.B;
                coef[2 * N + n0 + k] = f.K;
            }
        }
        *reinterpret_cast<bf16x8*>(wfold + (size_t)c * ldf + n0) = o;
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (tid == 0) bias[c] = red[0] + red[1] + red[2] + red[3];
}

// wt: data-gradient copy of the weights [C][N] (bf16); outputs: wfold [C][N + C] bf16, bias [C], coef [3][N] (A, B, K)
PDT_API int pdt_bnfold_prepare(const float* sums, const float* scale, const float* invstd, const float* mean,
                               float inv_count, const void* wt, void* wfold, float* bias, float* coef, int C, int N,
                               hipStream_t s) {
    if (C < 64 || C % 64 != 0 || N < 64 || N % 64 != 0) return -4;
    const int tiles = C / 32;
    hipLaunchKernelGGL(bnfold_prepare_kernel, dim3(tiles * tiles + C), dim3(256), 0, s, sums, scale, invstd, mean,
                       inv_count, (const bf16*)wt, (bf16*)wfold, bias, coef, C, N);
    return pdt_launch_status();
}

// dw[n][c] += B_n (W . G)[n][c] + K_n * rows * amean[c]       (W: [N][C] bf16, G: [C][C] fp32 symmetric up to rounding)
// grid: (N/32) (C/32) blocks; the four waves of a block split the reduction over c'.
__global__ __launch_bounds__(256) void bnfold_wgrad_kernel(float* __restrict__ dw, const bf16* __restrict__ w,
                                                            const float* __restrict__ gram,
                                                            const float* __restrict__ amean, float rows,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ prod, int C, int N) {
    __shared__ float red[4 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_c = C / 32;
    const int bi = blockIdx.x / tiles_c, bj = blockIdx.x % tiles_c;
    const int r = lane & 31, h = lane >> 5;
    const int kq = C / 4;                                  // C is a multiple of 64: kq a multiple of 16
    const bf16* pa = w + (size_t)(32 * bi + r) * C + wave * kq + 8 * h;
    const float* pg = gram + (size_t)(32 * bj + r) * C + wave * kq + 8 * h;      // G[c][c'] stands in for G[c'][c]
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
    for (int s = 0; s < kq / 16; ++s) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(pg + 16 * s);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(pg + 16 * s + 4);
        bf16x8 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hi[k] = f2bf(g0[k]);
            lo[k] = f2bf(g0[k] - bf2f(hi[k]));
            hi[4 + k] = f2bf(g1[k]);
            lo[4 + k] = f2bf(g1[k] - bf2f(hi[4 + k]));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, lo, acc, 0, 0, 0);
    }
    float out[4];
    reduce_tiles(acc, red, wave, lane, out);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, col;
        tile_rc(tid + q * 256, row, col);
        const int n = 32 * bi + row, c = 32 * bj + col;
        float v = coef[N + n] * out[q] + coef[2 * N + n] * (rows * amean[c]);
        if (prod) v += coef[n] * prod[(size_t)n * C + c];
        dw[(size_t)n * C + c] += v;
    }
}

// prod: the unscaled product T = g^T a [N][C] (then dw also receives A o T), or null when a scaled weight-gradient
// launch has already added it
PDT_API int pdt_bnfold_wgrad(float* dw, const void* w, const float* gram, const float* amean, float rows,
                             const float* coef, const float* prod, int C, int N, hipStream_t s) {
    if (C < 64 || C % 64 != 0 || N < 32 || N % 32 != 0) return -4;
    hipLaunchKernelGGL(bnfold_wgrad_kernel, dim3((N / 32) * (C / 32)), dim3(256), 0, s, dw, (const bf16*)w, gram, amean,
                       rows, coef, prod, C, N);
    return pdt_launch_status();
}

// sums[0][n] = s0[n];  sums[1][n] = invstd[n] (sum_c W[n][c] T[n][c] - mean[n] s0[n]);  dbeta += s0, dgamma += s1.
// One wave per channel n.
__global__ __launch_bounds__(256) void bnfold_sums_kernel(const float* __restrict__ prod, const bf16* __restrict__ w,
                                                           const float* __restrict__ s0, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ sums,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                                           int N) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float part = 0.f;
    for (int c = lane * 8; c < C; c += 64 * 8) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + (size_t)n * C + c);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(prod + (size_t)n * C + c);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(prod + (size_t)n * C + c + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) part += bf2f(wv[k]) * t0[k] + bf2f(wv[4 + k]) * t1[k];
    }
    part = wave_sum(part);
    if (lane == 0) {
        const float a = s0[n];
        const float b = invstd[n] * (part - mean[n] * a);
        sums[n] = a;
        sums[N + n] = b;
        if (dbeta) dbeta[n] += a;
        if (dgamma) dgamma[n] += b;
    }
}

PDT_API int pdt_bnfold_sums(const float* prod, const void* w, const float* s0, const float* mean, const float* invstd,
                            float* sums, float* dgamma, float* dbeta, int C, int N, hipStream_t s) {
    if (C < 8 || C % 8 != 0 || N < 1) return -4;
    hipLaunchKernelGGL(bnfold_sums_kernel, dim3((N + 3) / 4), dim3(256), 0, s, prod, (const bf16*)w, s0, mean, invstd, sums,
                       dgamma, dbeta, C, N);
    return pdt_launch_status();
}
