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
// The forward pass has the mirror image.  The batch statistics of y follow from the column sums and the Gram matrix
// of a (the same G the backward wants):  sum_m y[m][n] = W[n] . colsum(a),  sum_m y[m][n]^2 = W[n] G W[n]^T  -- so
// they are known BEFORE the convolution runs, BatchNorm's scale can be folded into the weights, and the block tail
// relu(bn(y) + shortcut) becomes the convolution's epilogue: y is not written in forward either.
//
//   * bnfold_fwd_stats:      the shifted sums [sum (y - s), sum (y - s)^2] in the accumulator format of the other
//                            statistics kernels (reduce.hip), from W, G and colsum(a);
//   * bnfold_scale_weights:  rows of [scale o W | scale2 o W2] and the summed shifts.
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

// acc row 0: sum_m (y - s) = rows (W[n].amean - s_n) ;  sum_m (y - s)^2 = W[n] Gc W[n]^T + rows (W[n].amean - s_n)^2
// (y = a W^T, G = a^T a, Gc = G - rows amean amean^T: the CENTRED Gram matrix, so that the variance is a quadratic form
// of its own and not the difference of two numbers ~ mean^2).  One block per 32 channels n; its four waves split the
// 32-column tiles of W.G, each element of which is multiplied by W[n][c] and added up over c.
__global__ __launch_bounds__(256) void bnfold_fwd_stats_kernel(const bf16* __restrict__ w, const float* __restrict__ gram,
                                                                const float* __restrict__ amean, float rows,
                                                                const float* __restrict__ shift,
                                                                float* __restrict__ acc, int C, int N) {
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    float q[16];                                   // partial sum_c (W.G)[n][c] W[n][c] for this lane's 16 rows
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = 0.f;
    const bf16* pa = w + (size_t)(n0 + r) * C + 8 * h;
    for (int ct = wave; ct < C / 32; ct += 4) {
        const float* pg = gram + (size_t)(32 * ct + r) * C + 8 * h;          // G[c][c'] stands in for G[c'][c]
        const float* pm = amean + 8 * h;
        const float mr = rows * amean[32 * ct + r];
        f32x16 t;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll 4
        for (int s = 0; s < C / 16; ++s) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
            f32x4 g0 = *reinterpret_cast<const f32x4*>(pg + 16 * s);
            f32x4 g1 = *reinterpret_cast<const f32x4*>(pg + 16 * s + 4);
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(pm + 16 * s);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(pm + 16 * s + 4);
            // centred BEFORE the split into bf16 pairs: with |mean| >> sigma the raw entries are all ~ rows mean^2 and
            // the 16 bits a pair keeps would be spent on that offset (0.5 % variance error at 30 sigma)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g0[k] = fmaf(-mr, m0[k], g0[k]);
                g1[k] = fmaf(-mr, m1[k], g1[k]);
            }
            bf16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = f2bf(g0[k]);
                lo[k] = f2bf(g0[k] - bf2f(hi[k]));
                hi[4 + k] = f2bf(g1[k]);
                lo[4 + k] = f2bf(g1[k] - bf2f(hi[4 + k]));
            }
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, hi, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, lo, t, 0, 0, 0);
        }
        // t[reg]: row (reg & 3) + 8 (reg >> 2) + 4 h, column c = 32 ct + r
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            q[i] += t[i] * bf2f(w[(size_t)(n0 + row) * C + 32 * ct + r]);
        }
    }
    // sum over the 32 columns held by the lanes of each half, then over the waves
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = q[i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        q[i] = v;
    }
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[0][wave][(i & 3) + 8 * (i >> 2) + 4 * h] = q[i];
    }
    // W[n] . amean: thread (row = tid / 8, part = tid % 8)
    {
        const int row = tid >> 3, part = tid & 7;
        float d = 0.f;
        for (int c = part; c < C; c += 8) d += bf2f(w[(size_t)(n0 + row) * C + c]) * amean[c];
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (part == 0) red[1][0][row] = d;
    }
    __syncthreads();
    if (tid < 32) {
        const int n = n0 + tid;
        const float quad = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];   // sum_m (y - mean y)^2
        const float off = red[1][0][tid] - (shift ? shift[n] : 0.f);                                // mean y - s
        acc[n] = rows * off;
        acc[N + n] = fmaf(rows * off, off, quad);
    }
}

// acc: the [nrep][2][N] accumulator of the statistics kernels (row 0 is written, the others stay zero)
PDT_API int pdt_bnfold_fwd_stats(const void* w, const float* gram, const float* amean, float rows, const float* shift,
                                 float* acc, int C, int N, hipStream_t s) {
    if (C < 32 || C % 32 != 0 || N < 32 || N % 32 != 0) return -4;
    hipLaunchKernelGGL(bnfold_fwd_stats_kernel, dim3(N / 32), dim3(256), 0, s, (const bf16*)w, gram, amean, rows, shift, acc,
                       C, N);
    return pdt_launch_status();
}

// wf[n][0..C) = scale[n] w[n][:],  wf[n][C..C+C2) = scale2[n] w2[n][:]  (second part optional);  shift_out = shift (+ shift2)
__global__ __launch_bounds__(256) void bnfold_scale_weights_kernel(const bf16* __restrict__ w, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    const bf16* __restrict__ w2,
                                                                    const float* __restrict__ scale2,
                                                                    const float* __restrict__ shift2,
                                                                    bf16* __restrict__ wf, float* __restrict__ shift_out,
                                                                    int C, int C2, int N) {
    const int n = blockIdx.x;
    const int ld = C + C2;
    const float sc = scale[n], sc2 = w2 ? scale2[n] : 0.f;
    for (int c = threadIdx.x * 8; c < ld; c += 256 * 8) {
        const bool second = c >= C;
        const bf16x8 v = second ? *reinterpret_cast<const bf16x8*>(w2 + (size_t)n * C2 + (c - C))
                                : *reinterpret_cast<const bf16x8*>(w + (size_t)n * C + c);
        const float f = second ? sc2 : sc;
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(v[k]) * f);
        *reinterpret_cast<bf16x8*>(wf + (size_t)n * ld + c) = o;
    }
    if (threadIdx.x == 0) shift_out[n] = shift[n] + (w2 ? shift2[n] : 0.f);
}

PDT_API int pdt_bnfold_scale_weights(const void* w, const float* scale, const float* shift, const void* w2,
                                     const float* scale2, const float* shift2, void* wf, float* shift_out, int C, int C2,
                                     int N, hipStream_t s) {
    if (C % 8 != 0 || C2 % 8 != 0 || N < 1 || (w2 == nullptr) != (C2 == 0)) return -4;
    hipLaunchKernelGGL(bnfold_scale_weights_kernel, dim3(N), dim3(256), 0, s, (const bf16*)w, scale, shift, (const bf16*)w2,
                       scale2, shift2, (bf16*)wf, shift_out, C, C2, N);
    return pdt_launch_status();
}
