// Reduction kernels for gfx950: BatchNorm statistics, BatchNorm backward reduce, BN finalize,
// fused softmax cross-entropy (+ validation metrics) and bias-gradient column sums.
//
// Per-channel reductions over an NHWC tensor are column sums of a [rows][C] matrix.  A block
// owns a slab of at most 256 channels (32 lanes x 8 channels, 16 B per lane) and a chunk of
// rows; its 2 x slab partial sums are added with float atomics into one of NREP replica rows
// of a small persistent accumulator [NREP][2][C] (replicas spread the contention that a single
// row of addresses would see from ~1000 workgroups).  A tiny second kernel folds the replicas,
// ZEROES them for the next use and produces the per-channel result, so no memset launch is
// needed.  The same accumulator format is fed by the convolution epilogue (conv_igemm.hip).
//
// Statistics are accumulated SHIFTED by a per-channel value s (the BatchNorm running mean):
// sum(y - s) and sum((y - s)^2).  var = E[(y-s)^2] - E[y-s]^2 then only cancels when the batch
// mean is many standard deviations away from the running mean, and -- because SyncBatchNorm
// keeps running means identical on all ranks -- the shifted moments of different ranks can be
// averaged by ONE all-reduce.
#include "common.h"
#include "xgmi_exchange.h"

#define NREP PDT_NREP

int g_pdt_deterministic = 0;

// Deterministic mode: one accumulator row per workgroup, ordered folds, weight gradients through per-slice slabs
// (conv_wgrad.hip) -- two runs on the same inputs give bit-identical results, at a cost in speed.
PDT_API int pdt_set_deterministic(int on) {
    g_pdt_deterministic = on ? 1 : 0;
    return 0;
}
PDT_API int pdt_accumulator_rows() { return pdt_nrep(); }

// Fold the nrep replicas of accumulator element i (row stride `stride`) in index order and re-zero them.  All
// loads of a group of 16 are issued before the first store so they overlap (a load/store pair per replica
// serialises on the load latency).
__device__ __forceinline__ float acc_take(float* p, size_t stride, int nrep) {
    float s = 0.0f;
    for (int r0 = 0; r0 < nrep; r0 += NREP) {
        float v[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r) v[r] = p[(size_t)(r0 + r) * stride];
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            s += v[r];
            p[(size_t)(r0 + r) * stride] = 0.0f;
        }
    }
    return s;
}

__global__ void zero_floats_kernel(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.0f;
}


// N accumulator elements at once: every load of all N elements is in flight before the first store, so a thread pays
// one memory round trip for them instead of N (these folds are single-workgroup kernels on the critical path of every
// layer: their time IS round trips).  Elements with ok[n] false are skipped.
template <int N>
__device__ __forceinline__ void acc_take_n(float* const (&p)[N], const bool (&ok)[N], size_t stride, int nrep,
                                           float (&out)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) out[n] = 0.0f;
    for (int r0 = 0; r0 < nrep; r0 += NREP) {
        float v[N][NREP];
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int r = 0; r < NREP; ++r) v[n][r] = ok[n] ? p[n][(size_t)(r0 + r) * stride] : 0.0f;
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                out[n] += v[n][r];
                if (ok[n]) p[n][(size_t)(r0 + r) * stride] = 0.0f;
            }
    }
}

// ------------------------------------------------------------------------------------------
// Column reduce skeleton
// ------------------------------------------------------------------------------------------
struct ColGeom {
    int cv;        // column vectors (8 channels each) per block: min(C/8, 32)
    int rl;        // row lanes per block = 256 / cv
    int slabs;     // column slabs = (C/8) / cv
    int rows_per_block;
    int row_chunks;
};

static inline ColGeom col_geom(long long rows, int C) {
    ColGeom g;
    int cvec = C / 8;
    g.cv = cvec < 32 ? cvec : 32;
    g.rl = 256 / g.cv;
    g.slabs = cvec / g.cv;
    // (deterministic mode: at most one workgroup per accumulator row)
    long long target_blocks = (g_pdt_deterministic ? PDT_DET_NREP : 2048) / g.slabs;
    long long rpb = (rows + target_blocks - 1) / target_blocks;
    long long min_rpb = (long long)g.rl * 8;
    if (rpb < min_rpb) rpb = min_rpb;
    rpb = (rpb + g.rl - 1) / g.rl * g.rl;
    g.rows_per_block = (int)rpb;
    g.row_chunks = (int)((rows + rpb - 1) / rpb);
    return g;
}

// Reduce the per-thread 2x8 partials over the row lanes of the block and add them to the
// replica accumulator.  acc0/acc1: per-thread partial sums for its 8 channels.
__device__ __forceinline__ void block_col_reduce_atomic(float (&a0)[8], float (&a1)[8], int cv, int rl, int col_lane,
                                                        int row_lane, int c_base, int C, float* __restrict__ acc,
                                                        int nrep) {
    __shared__ float red[2][256][8 + 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = a0[k]; red[1][threadIdx.x][k] = a1[k]; }
    __syncthreads();
    // thread t < cv*8*2 finalises one (which, column) pair
    int ncol = cv * 8;
    for (int t = threadIdx.x; t < 2 * ncol; t += 256) {
        int which = t / ncol;
        int col = t % ncol;
        int cl = col / 8, k = col % 8;
        float s = 0.0f;
        for (int r = 0; r < rl; ++r) s += red[which][r * cv + cl][k];
        float* dst = acc + ((size_t)(blockIdx.x % nrep) * 2 + which) * C + c_base + col;
        atomicAdd(dst, s);
    }
}

// ---- BN statistics: acc += [sum(y - s), sum((y - s)^2)] -----------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16* __restrict__ y, const float* __restrict__ shift,
                                                        float* __restrict__ acc, long long rows, int C, int cv,
                                                        int rl, int slabs, int rows_per_block, int nrep) {
    int slab = blockIdx.x % slabs;
    int chunk = blockIdx.x / slabs;
    int col_lane = threadIdx.x % cv;
    int row_lane = threadIdx.x / cv;
    int c_base = slab * cv * 8;
    int c0 = c_base + col_lane * 8;
    float s[8], a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = shift ? shift[c0 + k] : 0.0f; a0[k] = 0.0f; a1[k] = 0.0f; }
    long long r0 = (long long)chunk * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long long r = r0 + row_lane; r < r1; r += rl) {
        bf16x8 v = ld_bf16x8(y + r * C + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float d = bf2f(v[k]) - s[k];
            a0[k] += d;
            a1[k] += d * d;
        }
    }
    block_col_reduce_atomic(a0, a1, cv, rl, col_lane, row_lane, c_base, C, acc, nrep);
}

PDT_API int pdt_bn_stats_accum(const void* y, const float* shift, float* acc, long long rows, int C, hipStream_t s) {
    if (C % 8 != 0 || (C / 8 < 32 ? 256 % (C / 8) != 0 : (C / 8) % 32 != 0)) return -1;
    ColGeom g = col_geom(rows, C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(g.slabs * g.row_chunks), dim3(256), 0, s, (const bf16*)y, shift, acc, rows,
                       C, g.cv, g.rl, g.slabs, g.rows_per_block, pdt_nrep());
    return pdt_launch_status();
}

// ---- replicas -> shifted moments [E[y-s], E[(y-s)^2]] ; zero the replicas -------------------
__global__ __launch_bounds__(256) void bn_moments_kernel(float* __restrict__ acc, float* __restrict__ moments, int C,
                                                          float inv_count, int nrep) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 2*C
    if (i >= 2 * C) return;
    moments[i] = acc_take(acc + i, (size_t)2 * C, nrep) * inv_count;
}

PDT_API int pdt_bn_moments(float* acc, float* moments, int C, float inv_count, hipStream_t s) {
    hipLaunchKernelGGL(bn_moments_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, acc, moments, C, inv_count, pdt_nrep());
    return pdt_launch_status();
}

// ---- finalize: (summed shifted moments) -> scale, shift, mean, invstd ; update running stats --
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ moments, float inv_world,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float momentum, float eps,
                                                           float unbias, float* __restrict__ scale,
                                                           float* __restrict__ shift, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = rmean[c];
    float m0 = moments[c] * inv_world;
    float m1 = moments[C + c] * inv_world;
    float mean = s + m0;
    float var = fmaxf(m1 - m0 * m0, 0.0f);
    float invstd = rsqrtf(var + eps);
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    rmean[c] = (1.0f - momentum) * s + momentum * mean;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var * unbias;
}

PDT_API int pdt_bn_finalize(const float* moments, float inv_world, const float* gamma, const float* beta,
                            float* rmean, float* rvar, float momentum, float eps, float unbias, float* scale,
                            float* shift, float* mean_out, float* invstd_out, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, moments, inv_world, gamma, beta, rmean,
                       rvar, momentum, eps, unbias, scale, shift, mean_out, invstd_out, C);
    return pdt_launch_status();
}

// ---- fused: replicas -> local moments -> [exchange over the ranks] -> finalize ----------------
// One block; replaces bn_moments + (all-reduce) + bn_finalize, i.e. two or three dependent tiny
// launches on the critical path of every layer.  ``use_car`` selects the xGMI exchange.
__global__ __launch_bounds__(1024) void bn_finalize_fused_kernel(
    const CarDev car, int use_car, float* __restrict__ acc, float inv_count, float inv_world,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, float momentum, float eps, float unbias, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out, int C, int nrep) {
    __shared__ float vals[CAR_MAX_FLOATS];
    __shared__ unsigned ctl[CAR_CTL_WORDS];
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        vals[i] = acc_take(acc + i, (size_t)2 * C, nrep) * inv_count;
    }
    __syncthreads();
    // (a timed-out exchange leaves NaN in vals: every output below is then NaN, never stale memory)
    if (use_car) car_exchange_block(car, vals, 2 * C, ctl);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = rmean[c];
        float m0 = vals[c] * inv_world;
        float m1 = vals[C + c] * inv_world;
        float mean = s + m0;
        float var = fmaxf(m1 - m0 * m0, 0.0f);
        float invstd = rsqrtf(var + eps);
        float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
        mean_out[c] = mean;
        invstd_out[c] = invstd;
        rmean[c] = (1.0f - momentum) * s + momentum * mean;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var * unbias;
    }
}

static bool car_ready(const CarHost* h, int n) {
    if (!h || n > CAR_MAX_FLOATS || n / 4 > h->dev.max4 || (n & 3)) return false;
    for (int r = 0; r < h->dev.world; ++r)
        if (!h->dev.peer_recv[r]) return false;
    return true;
}

PDT_API int pdt_bn_finalize_fused(void* car, float* acc, float inv_count, float inv_world, const float* gamma,
                                  const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                  float unbias, float* scale, float* shift, float* mean_out, float* invstd_out, int C,
                                  hipStream_t s) {
    if (C < 1 || 2 * C > CAR_MAX_FLOATS) return -1;
    const CarHost* h = static_cast<const CarHost*>(car);
    if (h && !car_ready(h, 2 * C)) return -3;
    CarDev dev = h ? h->dev : CarDev{};
    int threads = 2 * C >= 1024 ? 1024 : (2 * C >= 256 ? 2 * C : 256);     // one accumulator element per thread
    hipLaunchKernelGGL(bn_finalize_fused_kernel, dim3(1), dim3(threads), 0, s, dev, h ? 1 : 0, acc, inv_count, inv_world,
                       gamma, beta, rmean, rvar, momentum, eps, unbias, scale, shift, mean_out, invstd_out, C, pdt_nrep());
    return pdt_launch_status();
}

__global__ __launch_bounds__(256) void bn_eval_affine_kernel(const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ rmean,
                                                              const float* __restrict__ rvar, float eps,
                                                              float* __restrict__ scale, float* __restrict__ shift,
                                                              int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] * rsqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

PDT_API int pdt_bn_eval_affine(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, rmean, rvar, eps, scale,
                       shift, C);
    return pdt_launch_status();
}

// ---- BN backward reduce: acc += [sum(dy), sum(dy * (y - mean))], dy = g * relu-mask ---------
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16* __restrict__ g, const bf16* __restrict__ y,
                                                             const bf16* __restrict__ msrc,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ acc,
                                                             long long rows, int C, int cv, int rl, int slabs,
                                                             int rows_per_block, int nrep) {
    int slab = blockIdx.x % slabs;
    int chunk = blockIdx.x / slabs;
    int col_lane = threadIdx.x % cv;
    int row_lane = threadIdx.x / cv;
    int c_base = slab * cv * 8;
    int c0 = c_base + col_lane * 8;
    float mu[8], sc[8], sh[8], a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k];
        sc[k] = scale[c0 + k];
        sh[k] = shift[c0 + k];
        a0[k] = 0.0f;
        a1[k] = 0.0f;
    }
    long long r0 = (long long)chunk * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long long r = r0 + row_lane; r < r1; r += rl) {
        bf16x8 gv = ld_bf16x8(g + r * C + c0);
        bf16x8 yv = ld_bf16x8(y + r * C + c0);
        bf16x8 mv;
        if (MODE == MASK_TENSOR) mv = ld_bf16x8(msrc + r * C + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float yy = bf2f(yv[k]);
            float dy = bf2f(gv[k]);
            if (MODE == MASK_TENSOR) dy = bf2f(mv[k]) > 0.0f ? dy : 0.0f;
            if (MODE == MASK_RECOMPUTE) dy = (yy * sc[k] + sh[k]) > 0.0f ? dy : 0.0f;
            a0[k] += dy;
            a1[k] += dy * (yy - mu[k]);
        }
    }
    block_col_reduce_atomic(a0, a1, cv, rl, col_lane, row_lane, c_base, C, acc, nrep);
}

PDT_API int pdt_bn_bwd_reduce_accum(const void* g, const void* y, const void* msrc, const float* mean,
                                    const float* scale, const float* shift, float* acc, long long rows, int C,
                                    int mode, hipStream_t s) {
    if (C % 8 != 0 || (C / 8 < 32 ? 256 % (C / 8) != 0 : (C / 8) % 32 != 0)) return -1;
    ColGeom gm = col_geom(rows, C);
    dim3 grid(gm.slabs * gm.row_chunks), block(256);
#define LAUNCH(M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<M>), grid, block, 0, s, (const bf16*)g, (const bf16*)y, \
                                     (const bf16*)msrc, mean, scale, shift, acc, rows, C, gm.cv, gm.rl, gm.slabs, gm.rows_per_block, pdt_nrep())
    if (mode == MASK_NONE) LAUNCH(MASK_NONE);
    else if (mode == MASK_TENSOR) LAUNCH(MASK_TENSOR);
    else LAUNCH(MASK_RECOMPUTE);
#undef LAUNCH
    return pdt_launch_status();
}

// The same sums for the BatchNorm under a fused ReLU + 3x3/s2 max pooling (the ResNet stem), taken
// straight from the POOLED gradient and the pooling's winner indices: dy = poolgrad * [y*scale +
// shift > 0].  The dense H x W gradient is neither written nor read.
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(
    const bf16* __restrict__ dout, const unsigned char* __restrict__ idx, const bf16* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ acc, long long rows, int C, int H, int W, int Ho, int Wo, int cv, int rl, int slabs,
    int rows_per_block, int nrep) {
    int slab = blockIdx.x % slabs;
    int chunk = blockIdx.x / slabs;
    int col_lane = threadIdx.x % cv;
    int row_lane = threadIdx.x / cv;
    int c_base = slab * cv * 8;
    int c0 = c_base + col_lane * 8;
    const int cvec = C / 8;
    float mu[8], sc[8], sh[8], a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c0 + k];
        sc[k] = scale[c0 + k];
        sh[k] = shift[c0 + k];
        a0[k] = 0.0f;
        a1[k] = 0.0f;
    }
    long long r0 = (long long)chunk * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long long r = r0 + row_lane; r < r1; r += rl) {
        // 32-bit index arithmetic (the launcher checks rows < 2^31): 64-bit divisions cost more than the loads
        const unsigned ru = (unsigned)r;
        const unsigned tu = ru / (unsigned)W;
        int w = (int)(ru - tu * (unsigned)W);
        int b = (int)(tu / (unsigned)H);
        int h = (int)(tu - (unsigned)b * (unsigned)H);
        bf16x8 yv = ld_bf16x8(y + r * C + c0);
        float g[8];
        pool3x3s2_grad8(dout, idx, b, h, w, c0 / 8, cvec, Ho, Wo, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float yy = bf2f(yv[k]);
            float dy = (yy * sc[k] + sh[k]) > 0.0f ? g[k] : 0.0f;
            a0[k] += dy;
            a1[k] += dy * (yy - mu[k]);
        }
    }
    block_col_reduce_atomic(a0, a1, cv, rl, col_lane, row_lane, c_base, C, acc, nrep);
}

PDT_API int pdt_pool_bn_bwd_reduce_accum(const void* dout, const void* idx, const void* y, const float* mean,
                                         const float* scale, const float* shift, float* acc, int B, int H, int W,
                                         int C, int Ho, int Wo, hipStream_t s) {
    if (C % 8 != 0 || (C / 8 < 32 ? 256 % (C / 8) != 0 : (C / 8) % 32 != 0)) return -1;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1 || B < 1) return -2;
    long long rows = (long long)B * H * W;
    if (rows >= 0x7fffffffLL) return -3;
    ColGeom gm = col_geom(rows, C);
    hipLaunchKernelGGL(pool_bn_bwd_reduce_kernel, dim3(gm.slabs * gm.row_chunks), dim3(256), 0, s, (const bf16*)dout,
                       (const unsigned char*)idx, (const bf16*)y, mean, scale, shift, acc, rows, C, H, W, Ho, Wo, gm.cv,
                       gm.rl, gm.slabs, gm.rows_per_block, pdt_nrep());
    return pdt_launch_status();
}

// replicas -> sums [sum(dy), sum(dy*xhat)] ; dbeta += sums[0], dgamma += sums[1] ; zero replicas
// Sixteen lanes per channel, one replica row each per trip, folded by shuffles: a dozen registers per thread, so a
// wave of this kernel finds room on a SIMD whose register file the co-running weight-gradient waves have almost
// filled (one thread per channel holding all replicas needed 108 and queued behind them for ~0.1 ms in stage 1).
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(float* __restrict__ acc, const float* __restrict__ invstd,
                                                           float* __restrict__ sums, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int C, int nrep) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, rl = threadIdx.x & 15;
    const bool live = c < C;                    // (whole 16-lane groups: the shuffles below stay inside a group)
    float a0 = 0.0f, a1 = 0.0f;
    if (live)
        for (int r = rl; r < nrep; r += 16) {
            float* q = acc + (size_t)r * 2 * C + c;
            a0 += q[0];
            a1 += q[C];
            q[0] = 0.0f;
            q[C] = 0.0f;
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64);
        a1 += __shfl_xor(a1, off, 64);
    }
    if (!live || rl != 0) return;
    const float s0 = a0, s1 = a1 * invstd[c];
    sums[c] = s0;
    sums[C + c] = s1;
    if (dbeta) dbeta[c] += s0;
    if (dgamma) dgamma[c] += s1;
}

// fused: replicas -> local sums (+ local dgamma / dbeta) -> exchange over the ranks -> sums
__global__ __launch_bounds__(1024) void bn_bwd_sums_fused_kernel(const CarDev car, float* __restrict__ acc,
                                                                  const float* __restrict__ invstd,
                                                                  float* __restrict__ sums, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int C, int nrep) {
    __shared__ float vals[CAR_MAX_FLOATS];
    __shared__ unsigned ctl[CAR_CTL_WORDS];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float is = invstd[c];
        const float db = dbeta ? dbeta[c] : 0.0f, dg = dgamma ? dgamma[c] : 0.0f;
        float* ptr[2] = {acc + c, acc + C + c};
        const bool ok[2] = {true, true};
        float t[2];
        acc_take_n<2>(ptr, ok, (size_t)2 * C, nrep, t);
        const float s0 = t[0], s1 = t[1] * is;
        vals[c] = s0;
        vals[C + c] = s1;
        if (dbeta) dbeta[c] = db + s0;      // local: the gradient buckets are summed over the ranks later
        if (dgamma) dgamma[c] = dg + s1;
    }
    __syncthreads();
    car_exchange_block(car, vals, 2 * C, ctl);      // after a timeout: NaN sums, status set
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sums[i] = vals[i];
}

PDT_API int pdt_bn_bwd_sums_fused(void* car, float* acc, const float* invstd, float* sums, float* dgamma,
                                  float* dbeta, int C, hipStream_t s) {
    const CarHost* h = static_cast<const CarHost*>(car);
    if (C < 1 || !car_ready(h, 2 * C)) return -3;
    int threads = C >= 1024 ? 1024 : (C >= 256 ? C : 256);
    hipLaunchKernelGGL(bn_bwd_sums_fused_kernel, dim3(1), dim3(threads), 0, s, h->dev, acc, invstd, sums, dgamma, dbeta,
                       C, pdt_nrep());
    return pdt_launch_status();
}

PDT_API int pdt_bn_bwd_sums(float* acc, const float* invstd, float* sums, float* dgamma, float* dbeta, int C,
                            hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(C * 16, 256)), dim3(256), 0, s, acc, invstd, sums, dgamma, dbeta, C,
                       pdt_nrep());
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// Softmax cross-entropy, one wave per row, fp32 math, single pass over the logits held in
// registers (N <= 64*REG columns).  Writes dlogits = (softmax - onehot) * gscale / B into a
// row of stride ld_out (columns >= N zeroed up to ld_out: the padded reduction dimension of
// the classifier's data-gradient GEMM) and adds loss_i / B into *loss.
// ------------------------------------------------------------------------------------------
#define CE_REG 32   // up to 2048 classes
__global__ __launch_bounds__(256) void softmax_ce_kernel(const bf16* __restrict__ logits,
                                                          const long long* __restrict__ labels,
                                                          bf16* __restrict__ dlogits, float* __restrict__ loss,
                                                          float* __restrict__ row_loss, int B, int N, int ld_in,
                                                          int ld_out, float gscale) {
    int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    int row = blockIdx.x * 4 + wave;
    if (row >= B) return;
    const bf16* src = logits + (size_t)row * ld_in;
    float v[CE_REG];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < CE_REG; ++j) {
        int c = lane + j * 64;
        v[j] = (c < N) ? bf2f(src[c]) : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float se = 0.0f;
#pragma unroll
    for (int j = 0; j < CE_REG; ++j) {
        int c = lane + j * 64;
        float e = (c < N) ? __expf(v[j] - mx) : 0.0f;
        v[j] = e;
        se += e;
    }
    se = wave_sum(se);
    int lab = (int)labels[row];
    float inv = 1.0f / se;
    float gs = gscale / (float)B;
    bf16* dst = dlogits + (size_t)row * ld_out;
#pragma unroll
    for (int j = 0; j < CE_REG; ++j) {
        int c = lane + j * 64;
        if (c < ld_out) {
            float p = (c < N) ? v[j] * inv : 0.0f;
            float d = (c < N) ? (p - (c == lab ? 1.0f : 0.0f)) * gs : 0.0f;
            dst[c] = f2bf(d);
        }
    }
    if (lane == 0) {
        float lab_logit = (lab >= 0 && lab < N) ? bf2f(src[lab]) : mx;
        float li = logf(se) + mx - lab_logit;
        if (row_loss) row_loss[row] = li;
        else atomicAdd(loss, li / (float)B);
    }
}

// Deterministic loss: sum of the per-row losses in a fixed tree order, one workgroup.
__global__ __launch_bounds__(256) void ordered_mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] / (float)n;
}

// row_loss: [B] scratch of the deterministic variant (the loss is then an ordered sum instead of B float atomics), or NULL
PDT_API int pdt_softmax_ce(const void* logits, const void* labels, void* dlogits, float* loss, float* row_loss, int B,
                           int N, int ld_in, int ld_out, float gscale, hipStream_t s) {
    if (N > 64 * CE_REG || ld_out > 64 * CE_REG) return -1;
    if (row_loss) {
        hipLaunchKernelGGL(softmax_ce_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, (const bf16*)logits,
                           (const long long*)labels, (bf16*)dlogits, loss, row_loss, B, N, ld_in, ld_out, gscale);
        hipLaunchKernelGGL(ordered_mean_kernel, dim3(1), dim3(256), 0, s, row_loss, loss, B);
        return pdt_launch_status();
    }
    // zeroed by a kernel, not a memset: inside a captured graph a memset node was seen to run
    // out of order with the kernel that accumulates into the same word
    hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, s, loss, 1);
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, (const bf16*)logits,
                       (const long long*)labels, (bf16*)dlogits, loss, nullptr, B, N, ld_in, ld_out, gscale);
    return pdt_launch_status();
}

// Validation metrics: out[4] += [sum of per-row loss, #(rank<1), #(rank<5), rows]
__global__ __launch_bounds__(256) void ce_metrics_kernel(const bf16* __restrict__ logits,
                                                          const long long* __restrict__ labels,
                                                          float* __restrict__ out, int B, int N, int ld_in) {
    int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    int row = blockIdx.x * 4 + wave;
    if (row >= B) return;
    const bf16* src = logits + (size_t)row * ld_in;
    int lab = (int)labels[row];
    float lab_logit = (lab >= 0 && lab < N) ? bf2f(src[lab]) : INFINITY;
    float mx = -INFINITY;
    float v[CE_REG];
    int greater = 0;
#pragma unroll
    for (int j = 0; j < CE_REG; ++j) {
        int c = lane + j * 64;
        v[j] = (c < N) ? bf2f(src[c]) : -INFINITY;
        mx = fmaxf(mx, v[j]);
        greater += (c < N && v[j] > lab_logit) ? 1 : 0;
    }
    mx = wave_max(mx);
    float se = 0.0f;
#pragma unroll
    for (int j = 0; j < CE_REG; ++j) {
        int c = lane + j * 64;
        se += (c < N) ? __expf(v[j] - mx) : 0.0f;
    }
    se = wave_sum(se);
    float gsum = wave_sum((float)greater);
    if (lane == 0) {
        atomicAdd(out + 0, logf(se) + mx - lab_logit);
        atomicAdd(out + 1, gsum < 1.0f ? 1.0f : 0.0f);
        atomicAdd(out + 2, gsum < 5.0f ? 1.0f : 0.0f);
        atomicAdd(out + 3, 1.0f);
    }
}

PDT_API int pdt_ce_metrics(const void* logits, const void* labels, float* out, int B, int N, int ld_in,
                           hipStream_t s) {
    if (N > 64 * CE_REG) return -1;
    hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, s, out, 4);
    hipLaunchKernelGGL(ce_metrics_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, (const bf16*)logits,
                       (const long long*)labels, out, B, N, ld_in);
    return pdt_launch_status();
}

// Column sums of a small [rows][ld] bf16 matrix: out[c] += sum_r x[r][c]  (classifier bias gradient)
// Rows are split over blockIdx.y (32 per block, independent loads) and the partial sums are added with
// float atomics, like every other gradient: one thread walking all rows is a chain of dependent loads
// (120 us for 512 rows).
#define COLSUM_ROWS 32
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, float* __restrict__ out, int rows,
                                                      int N, int ld, int rows_per_block) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.0f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) s += bf2f(x[(size_t)r * ld + c]);
    if (gridDim.y == 1) out[c] += s;          // single writer: plain, ordered
    else atomicAdd(out + c, s);
}

PDT_API int pdt_colsum_add(const void* x, float* out, int rows, int N, int ld, hipStream_t s) {
    if (rows <= 0 || N <= 0) return 0;
    // deterministic mode: one row chunk, i.e. one thread walks a whole column (slow, ordered)
    int rpb = g_pdt_deterministic ? rows : COLSUM_ROWS;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), cdiv(rows, rpb)), dim3(256), 0, s, (const bf16*)x, out,
                       rows, N, ld, rpb);
    return pdt_launch_status();
}
