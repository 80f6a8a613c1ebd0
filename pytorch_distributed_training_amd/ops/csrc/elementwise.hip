// Memory-bound elementwise kernels for gfx950: fused SGD, BatchNorm apply / backward
// elementwise, input layout transform, pooling, weight repacking.
//
// All of them stream NHWC bf16 data 16 bytes per lane (8 channels).  The grid is grid-strided
// (and capped far above what a ResNet step needs, see ew_grid); since the grid stride in vectors is a
// multiple of 256 and C/8 divides 256 for every channel count used by ResNets (C a power of
// two between 64 and 2048), each thread always lands on the SAME 8 channels, so per-channel
// constants are loaded into registers once.
#include "common.h"
#include <cstdlib>

static inline int ew_grid(long long nvec, int per_thread = 1) {
    // Cap on the grid (PDT_EW_BLOCKS overrides): 65536 blocks, i.e. none at the tensor sizes of a ResNet step.  Inside
    // the step many short-lived blocks beat 2048 looping ones by 0.3 ms of 60 (batch 1024): they interleave better
    // with the co-running weight-gradient workgroups, which outweighs reloading the per-channel constants per block.
    static const long long cap = [] { const char* e = getenv("PDT_EW_BLOCKS"); return e ? atoll(e) : 65536LL; }();
    long long blocks = (nvec + 256LL * per_thread - 1) / (256LL * per_thread);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ------------------------------------------------------------------------------------------
// Fused SGD with momentum and weight decay over flat fp32 buffers (one pass):
//   g' = g*grad_scale + wd*p ; m = first ? g' : mu*m + (1-damp)*g' ; p -= lr*(nesterov ? g'+mu*m : m)
// also zeroes g (weight-gradient kernels accumulate into it) and rewrites the bf16 mirror.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, bf16* __restrict__ pc, long long n4,
                                                   float lr, float mu, float wd, float gscale, float damp,
                                                   int nesterov, int first, int use_mom) {
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        bf16x4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float grad = gv[k] * gscale + wd * pv[k];
            float step = grad;
            if (use_mom) {
                float mm = first ? grad : mu * mv[k] + (1.0f - damp) * grad;
                mv[k] = mm;
                step = nesterov ? grad + mu * mm : mm;
            }
            pv[k] -= lr * step;
            out[k] = f2bf(pv[k]);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        reinterpret_cast<f32x4*>(g)[i] = z;
        if (pc) reinterpret_cast<bf16x4*>(pc)[i] = out;
    }
}

PDT_API int pdt_sgd_step(float* p, float* g, float* m, void* pc, long long n, float lr, float mu, float wd,
                         float gscale, float damp, int nesterov, int first, hipStream_t s) {
    if (n % 4 != 0) return -1;
    long long n4 = n / 4;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n4, 2)), dim3(256), 0, s, p, g, m, (bf16*)pc, n4, lr, mu, wd,
                       gscale, damp, nesterov, first, mu != 0.0f ? 1 : 0);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// BatchNorm apply: out = relu?(y*scale[c] + shift[c] (+ residual))
// ------------------------------------------------------------------------------------------
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const bf16* __restrict__ res, bf16* __restrict__ out,
                                                        unsigned char* __restrict__ bits, long long nvec, int cvec,
                                                        float* __restrict__ colsum, int nrep) {
    // bits (optional): one byte per 8-channel vector, bit k = [out channel k > 0] -- the ReLU mask
    // backward needs, 1/16 of the tensor it would otherwise re-read for the sign.
    // colsum (optional): replica accumulator [nrep][2][C]; row 0 receives the per-channel sums of the values
    // written (what the folded BatchNorm backward of the NEXT layer wants from this tensor, see bn_fold.hip).
    __shared__ float red[256][8 + 1];
    long long stride = (long long)gridDim.x * blockDim.x;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec && !colsum) return;
    // Walk the tensor from its END: the conv kernel that produced y wrote it front to back, so the
    // tail is what the memory-side cache still holds; and the front of `out`, written last here,
    // is what the next conv reads first.  nvec and the grid stride are multiples of cvec, so the
    // channel group of a thread is fixed (and defined for threads past the end, too).
    const int cg = cvec - 1 - (int)(t % cvec);
    const int c0 = cg * 8;
    float sc[8], sh[8], cs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; cs[k] = 0.0f; }
    for (; t < nvec; t += stride) {
        const long long i = nvec - 1 - t;
        bf16x8 v = ld_stream_bf16x8(y + i * 8);      // not read again before backward
        bf16x8 r;
        if (RES) r = ld_stream_bf16x8(res + i * 8);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = bf2f(v[k]) * sc[k] + sh[k];
            if (RES) f += bf2f(r[k]);
            if (RELU) f = fmaxf(f, 0.0f);
            o[k] = f2bf(f);
            cs[k] += bf2f(o[k]);
        }
        st_bf16x8(out + i * 8, o);
        if (bits) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) m |= (bf2f(o[k]) > 0.0f ? 1u : 0u) << k;
            bits[i] = (unsigned char)m;
        }
    }
    if (!colsum) return;
    // threads tid, tid + cvec, ... of the block share a channel group (256 % cvec == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = cs[k];
    __syncthreads();
    const int C = cvec * 8;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = cvec - 1 - c / 8, k = c % 8;
        float sum = 0.0f;
        for (int r = g; r < 256; r += cvec) sum += red[r][k];
        atomicAdd(colsum + (size_t)(blockIdx.x % nrep) * 2 * C + c, sum);
    }
}

// colsum: optional replica accumulator [nrep][2][C] (see the kernel); not in deterministic mode (rows are shared)
PDT_API int pdt_bn_apply_ex(const void* y, const float* scale, const float* shift, const void* res, void* out, void* bits,
                            long long rows, int C, int relu, float* colsum, hipStream_t s) {
    // the grid stride (grid*256 vectors) must be a multiple of C/8 so channels stay fixed per thread
    if (C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return -1;
    if (colsum && g_pdt_deterministic) return -9;
    long long nvec = rows * (C / 8);
    int cvec = C / 8;
    dim3 grid(ew_grid(nvec, 4)), block(256);
    const bf16* yy = (const bf16*)y; const bf16* rr = (const bf16*)res; bf16* oo = (bf16*)out;
    unsigned char* bb = (unsigned char*)bits;
    const int nrep = pdt_nrep();
    if (res && relu) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, block, 0, s, yy, scale, shift, rr, oo, bb, nvec, cvec, colsum, nrep);
    else if (res) hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, block, 0, s, yy, scale, shift, rr, oo, bb, nvec, cvec, colsum, nrep);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, block, 0, s, yy, scale, shift, rr, oo, bb, nvec, cvec, colsum, nrep);
    else hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, block, 0, s, yy, scale, shift, rr, oo, bb, nvec, cvec, colsum, nrep);
    return pdt_launch_status();
}

PDT_API int pdt_bn_apply(const void* y, const float* scale, const float* shift, const void* res, void* out, void* bits,
                         long long rows, int C, int relu, hipStream_t s) {
    return pdt_bn_apply_ex(y, scale, shift, res, out, bits, rows, C, relu, nullptr, s);
}

// Block tail with a projection shortcut: out = relu(y*scale + shift + y2*scale2 + shift2).  The
// shortcut's normalised tensor is never written: its BatchNorm is applied here, on the fly.
__global__ __launch_bounds__(256) void bn_apply2_kernel(const bf16* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const bf16* __restrict__ y2, const float* __restrict__ scale2,
                                                         const float* __restrict__ shift2, bf16* __restrict__ out,
                                                         unsigned char* __restrict__ bits, long long nvec, int cvec,
                                                         int relu) {
    long long stride = (long long)gridDim.x * blockDim.x;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec) return;
    int c0 = (int)((nvec - 1 - t) % cvec) * 8;      // back to front, as in bn_apply
    float sc[8], sh[8], sc2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = scale[c0 + k];
        sc2[k] = scale2[c0 + k];
        sh[k] = shift[c0 + k] + shift2[c0 + k];
    }
    for (; t < nvec; t += stride) {
        const long long i = nvec - 1 - t;
        bf16x8 v = ld_stream_bf16x8(y + i * 8);
        bf16x8 r = ld_stream_bf16x8(y2 + i * 8);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = bf2f(v[k]) * sc[k] + bf2f(r[k]) * sc2[k] + sh[k];
            if (relu) f = fmaxf(f, 0.0f);
            o[k] = f2bf(f);
        }
        st_bf16x8(out + i * 8, o);
        if (bits) {                                  // ReLU mask byte, as in bn_apply
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) m |= (bf2f(o[k]) > 0.0f ? 1u : 0u) << k;
            bits[i] = (unsigned char)m;
        }
    }
}

PDT_API int pdt_bn_apply2(const void* y, const float* scale, const float* shift, const void* y2, const float* scale2,
                          const float* shift2, void* out, void* bits, long long rows, int C, int relu, hipStream_t s) {
    if (C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return -1;
    long long nvec = rows * (C / 8);
    hipLaunchKernelGGL(bn_apply2_kernel, dim3(ew_grid(nvec, 4)), dim3(256), 0, s, (const bf16*)y, scale, shift,
                       (const bf16*)y2, scale2, shift2, (bf16*)out, (unsigned char*)bits, nvec, C / 8, relu);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// BatchNorm backward elementwise (fused ReLU mask):
//   dy = g * mask ; dx = (dy - s0 - xhat*s1) * scale,  xhat = (y-mean)*invstd, [s0,s1] = sums*inv_count
// folded to dx = dy*A + y*Bc + Cc with A = scale, Bc = -invstd*s1*scale, Cc = (mean*invstd*s1 - s0)*scale.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_elemt_kernel(const bf16* __restrict__ g, const bf16* __restrict__ y,
                                                            const bf16* __restrict__ msrc,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ sums, float inv_count,
                                                            bf16* __restrict__ dx, long long nvec, int cvec, int C) {
    long long stride = (long long)gridDim.x * blockDim.x;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec) return;
    // back to front: g was just written front to back by a data-gradient kernel, and the front of
    // dx, written last, is what the next data-/weight-gradient kernels read first
    int c0 = (int)((nvec - 1 - t) % cvec) * 8;
    float A[8], Bc[8], Cc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s0 = sums[c0 + k] * inv_count, s1 = sums[C + c0 + k] * inv_count;
        float sc = scale[c0 + k], is = invstd[c0 + k], mu = mean[c0 + k];
        A[k] = sc;
        Bc[k] = -is * s1 * sc;
        Cc[k] = (mu * is * s1 - s0) * sc;
        sh[k] = shift[c0 + k];
    }
    for (; t < nvec; t += stride) {
        const long long i = nvec - 1 - t;
        bf16x8 gv = ld_stream_bf16x8(g + i * 8);     // last use of g and y in this step
        bf16x8 yv = ld_stream_bf16x8(y + i * 8);
        bf16x8 mv;
        if (MODE == MASK_TENSOR) mv = ld_bf16x8(msrc + i * 8);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float yy = bf2f(yv[k]);
            float dy = bf2f(gv[k]);
            if (MODE == MASK_TENSOR) dy = bf2f(mv[k]) > 0.0f ? dy : 0.0f;
            if (MODE == MASK_RECOMPUTE) dy = (yy * A[k] + sh[k]) > 0.0f ? dy : 0.0f;
            o[k] = f2bf(dy * A[k] + yy * Bc[k] + Cc[k]);
        }
        st_stream_bf16x8(dx + i * 8, o);
    }
}

PDT_API int pdt_bn_bwd_elemt(const void* g, const void* y, const void* msrc, const float* mean, const float* invstd,
                             const float* scale, const float* shift, const float* sums, float inv_count, void* dx,
                             long long rows, int C, int mode, hipStream_t s) {
    if (C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return -1;
    int cvec = C / 8;
    long long nvec = rows * cvec;
    dim3 grid(ew_grid(nvec, 4)), block(256);
#define LAUNCH(M) hipLaunchKernelGGL((bn_bwd_elemt_kernel<M>), grid, block, 0, s, (const bf16*)g, (const bf16*)y, \
                                     (const bf16*)msrc, mean, invstd, scale, shift, sums, inv_count, (bf16*)dx, nvec, cvec, C)
    if (mode == MASK_NONE) LAUNCH(MASK_NONE);
    else if (mode == MASK_TENSOR) LAUNCH(MASK_TENSOR);
    else LAUNCH(MASK_RECOMPUTE);
#undef LAUNCH
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// GPU-side training augmentation: RandomResizedCrop + horizontal flip of a batch of stored uint8 images.
//   src [B][S][S][3] uint8, box [B][5] int32 = (x, y, w, h, flip) chosen on the host, dst [B][O][O][3] uint8.
// Bilinear sampling with half-pixel centres (the arithmetic of F.interpolate(mode="bilinear",
// align_corners=False) on the cropped box), result rounded to nearest.  One thread per output pixel.
// The host then never decodes or resizes: it only gathers fixed-size records (memcpy) -- at >13 k images/s
// per GPU a CPU resize per image would need ~100 cores.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_resize_flip_kernel(const unsigned char* __restrict__ src,
                                                                const int* __restrict__ box,
                                                                unsigned char* __restrict__ dst, int B, int S, int O) {
    const long long total = (long long)B * O * O;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const unsigned iu = (unsigned)idx;                       // total < 2^31 (checked by the launcher)
    const unsigned t = iu / (unsigned)O;
    const int ox = (int)(iu - t * (unsigned)O);
    const int b = (int)(t / (unsigned)O);
    const int oy = (int)(t - (unsigned)b * (unsigned)O);
    const int* bx = box + b * 5;
    const int x0 = bx[0], y0 = bx[1], cw = bx[2], ch = bx[3], flip = bx[4];
    const int oxs = flip ? O - 1 - ox : ox;
    float fx = ((float)oxs + 0.5f) * ((float)cw / (float)O) - 0.5f;
    float fy = ((float)oy + 0.5f) * ((float)ch / (float)O) - 0.5f;
    fx = fx < 0.0f ? 0.0f : fx;
    fy = fy < 0.0f ? 0.0f : fy;
    int ix = (int)fx, iy = (int)fy;
    const float ax = fx - (float)ix, ay = fy - (float)iy;
    const int ix1 = ix + 1 < cw ? ix + 1 : cw - 1, iy1 = iy + 1 < ch ? iy + 1 : ch - 1;
    ix = ix < cw ? ix : cw - 1;
    iy = iy < ch ? iy : ch - 1;
    const unsigned char* img = src + (long long)b * S * S * 3;
    const unsigned char* p00 = img + ((long long)(y0 + iy) * S + x0 + ix) * 3;
    const unsigned char* p01 = img + ((long long)(y0 + iy) * S + x0 + ix1) * 3;
    const unsigned char* p10 = img + ((long long)(y0 + iy1) * S + x0 + ix) * 3;
    const unsigned char* p11 = img + ((long long)(y0 + iy1) * S + x0 + ix1) * 3;
    unsigned char* o = dst + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = (float)p00[c] + ax * ((float)p01[c] - (float)p00[c]);
        const float bot = (float)p10[c] + ax * ((float)p11[c] - (float)p10[c]);
        const float v = top + ay * (bot - top);
        o[c] = (unsigned char)(int)(v + 0.5f);
    }
}

PDT_API int pdt_crop_resize_flip(const void* src, const int* box, void* dst, int B, int S, int O, hipStream_t s) {
    const long long total = (long long)B * O * O;
    if (B <= 0 || S <= 0 || O <= 0) return 0;
    if (total >= 0x7fffffffLL) return -2;
    hipLaunchKernelGGL(crop_resize_flip_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)src, box, (unsigned char*)dst, B, S, O);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// Input transform: loader batch -> zero-haloed NHWC4 bf16 stem input [B][H+6][Wp][4].
// The stem 7x7/s2/p3 convolution then needs no bounds checks: pixel (h, w) of the image sits
// at (h+3, w+3); channel 3 is zero.  Source is fp32 NCHW (already normalised) or uint8 NHWC
// (normalised here with (x/255 - mean)/std).
// ------------------------------------------------------------------------------------------
template <bool U8>
__global__ __launch_bounds__(256) void input_pad_kernel(const void* __restrict__ src, bf16* __restrict__ dst, int B,
                                                         int H, int W, int Hp, int Wp, float m0, float m1, float m2,
                                                         float i0, float i1, float i2) {
    long long total = (long long)B * Hp * Wp;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const unsigned iu = (unsigned)idx;                    // total < 2^31 (checked by the launcher)
        const unsigned t = iu / (unsigned)Wp;
        int wp = (int)(iu - t * (unsigned)Wp);
        int b = (int)(t / (unsigned)Hp);
        int hp = (int)(t - (unsigned)b * (unsigned)Hp);
        int h = hp - 3, w = wp - 3;
        bf16x4 o;
        o[0] = o[1] = o[2] = o[3] = f2bf(0.0f);
        if (h >= 0 && h < H && w >= 0 && w < W) {
            if (U8) {
                const unsigned char* p = (const unsigned char*)src + (((long long)b * H + h) * W + w) * 3;
                o[0] = f2bf(((float)p[0] * (1.0f / 255.0f) - m0) * i0);
                o[1] = f2bf(((float)p[1] * (1.0f / 255.0f) - m1) * i1);
                o[2] = f2bf(((float)p[2] * (1.0f / 255.0f) - m2) * i2);
            } else {
                const float* p = (const float*)src + ((long long)b * 3 * H + h) * W + w;
                long long plane = (long long)H * W;
                o[0] = f2bf(p[0]);
                o[1] = f2bf(p[plane]);
                o[2] = f2bf(p[2 * plane]);
            }
        }
        reinterpret_cast<bf16x4*>(dst)[idx] = o;
    }
}

PDT_API int pdt_input_pad(const void* src, void* dst, int B, int H, int W, int Hp, int Wp, int is_u8, float m0,
                          float m1, float m2, float s0, float s1, float s2, hipStream_t s) {
    long long total = (long long)B * Hp * Wp;
    if (total >= 0x7fffffffLL) return -3;
    dim3 grid(ew_grid(total, 2)), block(256);
    if (is_u8) hipLaunchKernelGGL((input_pad_kernel<true>), grid, block, 0, s, src, (bf16*)dst, B, H, W, Hp, Wp, m0, m1, m2, 1.0f / s0, 1.0f / s1, 1.0f / s2);
    else hipLaunchKernelGGL((input_pad_kernel<false>), grid, block, 0, s, src, (bf16*)dst, B, H, W, Hp, Wp, m0, m1, m2, 1.0f, 1.0f, 1.0f);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// MaxPool 3x3 / stride 2 / pad 1, NHWC.  Forward stores the winning window position (0..8)
// per output element; backward is a gather: every input element sums the gradients of the
// (at most 2x2) windows that selected it, so no atomics are needed.  Ties: first maximum in
// (r, s) scan order wins, like ATen.
// ------------------------------------------------------------------------------------------
// AFFINE: the input is a raw conv output and relu(x*scale + shift), rounded to bf16 exactly as the
// stand-alone BatchNorm-apply pass would store it, is pooled instead -- the stem's normalised
// 112x112 activation is never written.
// WINNERS (with AFFINE): also stores the raw input value of every window's winner and the packed sign mask of the
// pooled result (bit k of byte i = channel 8i + k), so that the kernel which later produces the gradient of the pooled
// tensor can apply this ReLU's backward from the mask and add up the BatchNorm-backward sums of the layer in front
// of the pooling in its epilogue: sum(dy) and sum(dy * (y - mean)) over the dense tensor are sums over the windows of
// the pooled gradient times the winner's value, because every window hands its gradient to exactly one element.
template <bool AFFINE, bool WINNERS = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                           unsigned char* __restrict__ idx, int B, int H, int W,
                                                           int C, int Ho, int Wo, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           bf16* __restrict__ ywin = nullptr,
                                                           unsigned char* __restrict__ bits = nullptr) {
    int cvec = C / 8;
    long long total = (long long)B * Ho * Wo * cvec;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // 32-bit index arithmetic (total < 2^31, checked by the launchers)
        const unsigned iu = (unsigned)i;
        unsigned t = iu / (unsigned)cvec;
        int cv = (int)(iu - t * (unsigned)cvec);
        unsigned t2 = t / (unsigned)Wo;
        int wo = (int)(t - t2 * (unsigned)Wo);
        int b = (int)(t2 / (unsigned)Ho);
        int ho = (int)(t2 - (unsigned)b * (unsigned)Ho);
        float best[8], sc[8], sh[8], raw[WINNERS ? 8 : 1];
        unsigned bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            best[k] = -INFINITY;
            bi[k] = 0;
            if (WINNERS) raw[k] = 0.0f;
            if (AFFINE) { sc[k] = scale[cv * 8 + k]; sh[k] = shift[cv * 8 + k]; }
        }
        // all nine window positions are loaded unconditionally and back to back (a position outside the image re-reads
        // the window's centre, which always exists, and is given -inf): with a branch per position a wave has one load
        // in flight at a time
        bf16x8 v[9];
        bool ok[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int h = ho * 2 - 1 + q / 3, w = wo * 2 - 1 + q % 3;
            ok[q] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            const int hc = ok[q] ? h : ho * 2, wc = ok[q] ? w : wo * 2;
            v[q] = ld_bf16x8(x + (((long long)b * H + hc) * W + wc) * C + cv * 8);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f0 = bf2f(v[q][k]);
                float f = f0;
                if (AFFINE) f = bf2f(f2bf(fmaxf(f * sc[k] + sh[k], 0.0f)));
                f = ok[q] ? f : -INFINITY;
                if (f > best[k]) {
                    best[k] = f;
                    bi[k] = (unsigned)q;
                    if (WINNERS) raw[k] = f0;
                }
            }
        }
        bf16x8 o;
        unsigned long long packed = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[k] = f2bf(best[k]); packed |= (unsigned long long)bi[k] << (8 * k); }
        st_bf16x8(out + i * 8, o);
        reinterpret_cast<unsigned long long*>(idx)[i] = packed;
        if (WINNERS) {
            bf16x8 r;
            unsigned m8 = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                r[k] = f2bf(raw[k]);
                m8 |= (best[k] > 0.0f ? 1u : 0u) << k;
            }
            st_bf16x8(ywin + i * 8, r);
            bits[i] = (unsigned char)m8;
        }
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16* __restrict__ dout,
                                                           const unsigned char* __restrict__ idx,
                                                           bf16* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                           int Wo) {
    int cvec = C / 8;
    long long total = (long long)B * H * W * cvec;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int cv = (int)(i % cvec);
        long long t = i / cvec;
        int w = (int)(t % W); t /= W;
        int h = (int)(t % H);
        int b = (int)(t / H);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
        // windows (ho, wo) covering (h, w): ho*2-1 <= h <= ho*2+1
        int ho_lo = (h >= 1) ? (h - 1 + 1) / 2 : 0;   // ceil((h-1)/2)
        int ho_hi = (h + 1) / 2;
        int wo_lo = (w >= 1) ? (w - 1 + 1) / 2 : 0;
        int wo_hi = (w + 1) / 2;
        for (int ho = ho_lo; ho <= ho_hi && ho < Ho; ++ho) {
            int r = h - (ho * 2 - 1);
            for (int wo = wo_lo; wo <= wo_hi && wo < Wo; ++wo) {
                int s2 = w - (wo * 2 - 1);
                unsigned char want = (unsigned char)(r * 3 + s2);
                long long o = (((long long)b * Ho + ho) * Wo + wo) * cvec + cv;
                unsigned long long packed = reinterpret_cast<const unsigned long long*>(idx)[o];
                bf16x8 gv = ld_bf16x8(dout + o * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((unsigned char)(packed >> (8 * k)) == want) acc[k] += bf2f(gv[k]);
            }
        }
        bf16x8 o8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = f2bf(acc[k]);
        st_bf16x8(dx + i * 8, o8);
    }
}

// BatchNorm backward elementwise pass of the layer under a fused ReLU + max pooling, from the POOLED
// gradient: dx = poolgrad*[y*scale+shift > 0]*A + y*Bc + Cc (coefficients as in bn_bwd_elemt).
__global__ __launch_bounds__(256) void pool_bn_bwd_elemt_kernel(
    const bf16* __restrict__ dout, const unsigned char* __restrict__ idx, const bf16* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ sums, float inv_count, bf16* __restrict__ dx,
    long long nvec, int cvec, int C, int H, int W, int Ho, int Wo) {
    long long stride = (long long)gridDim.x * blockDim.x;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvec) return;
    const int cv = (int)((nvec - 1 - t) % cvec);        // the grid stride is a multiple of cvec
    const int c0 = cv * 8;
    float A[8], Bc[8], Cc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s0 = sums[c0 + k] * inv_count, s1 = sums[C + c0 + k] * inv_count;
        float sc = scale[c0 + k], is = invstd[c0 + k], mu = mean[c0 + k];
        A[k] = sc;
        Bc[k] = -is * s1 * sc;
        Cc[k] = (mu * is * s1 - s0) * sc;
        sh[k] = shift[c0 + k];
    }
    for (; t < nvec; t += stride) {
        const long long i = nvec - 1 - t;               // back to front, like bn_bwd_elemt
        // 32-bit index arithmetic (nvec < 2^31, checked by the launcher)
        const unsigned ru = (unsigned)i / (unsigned)cvec;
        const unsigned tu = ru / (unsigned)W;
        int w = (int)(ru - tu * (unsigned)W);
        int b = (int)(tu / (unsigned)H);
        int h = (int)(tu - (unsigned)b * (unsigned)H);
        bf16x8 yv = ld_stream_bf16x8(y + i * 8);
        float g[8];
        pool3x3s2_grad8(dout, idx, b, h, w, cv, cvec, Ho, Wo, g);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float yy = bf2f(yv[k]);
            float dy = (yy * A[k] + sh[k]) > 0.0f ? g[k] : 0.0f;
            o[k] = f2bf(dy * A[k] + yy * Bc[k] + Cc[k]);
        }
        st_stream_bf16x8(dx + i * 8, o);
    }
}

PDT_API int pdt_pool_bn_bwd_elemt(const void* dout, const void* idx, const void* y, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, const float* sums,
                                  float inv_count, void* dx, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    if (C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return -1;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1 || B < 1) return -2;
    int cvec = C / 8;
    long long nvec = (long long)B * H * W * cvec;
    if (nvec >= 0x7fffffffLL) return -3;
    hipLaunchKernelGGL(pool_bn_bwd_elemt_kernel, dim3(ew_grid(nvec, 4)), dim3(256), 0, s, (const bf16*)dout,
                       (const unsigned char*)idx, (const bf16*)y, mean, invstd, scale, shift, sums, inv_count,
                       (bf16*)dx, nvec, cvec, C, H, W, Ho, Wo);
    return pdt_launch_status();
}

PDT_API int pdt_maxpool_fwd(const void* x, void* out, void* idx, int B, int H, int W, int C, int Ho, int Wo,
                            hipStream_t s) {
    if (C % 8) return -1;
    long long total = (long long)B * Ho * Wo * (C / 8);
    if (total >= 0x7fffffffLL) return -3;
    hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)x, (bf16*)out,
                       (unsigned char*)idx, B, H, W, C, Ho, Wo, nullptr, nullptr);
    return pdt_launch_status();
}

PDT_API int pdt_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, void* out, void* idx, int B,
                                    int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    if (C % 8) return -1;
    long long total = (long long)B * Ho * Wo * (C / 8);
    if (total >= 0x7fffffffLL) return -3;
    hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)y, (bf16*)out,
                       (unsigned char*)idx, B, H, W, C, Ho, Wo, scale, shift);
    return pdt_launch_status();
}

// ... and the winners' raw values plus the packed sign mask of the result (see maxpool_fwd_kernel)
PDT_API int pdt_bn_relu_maxpool_fwd_winners(const void* y, const float* scale, const float* shift, void* out, void* idx,
                                            void* ywin, void* bits, int B, int H, int W, int C, int Ho, int Wo,
                                            hipStream_t s) {
    if (C % 8 || !ywin || !bits) return -1;
    long long total = (long long)B * Ho * Wo * (C / 8);
    if (total >= 0x7fffffffLL) return -3;
    hipLaunchKernelGGL((maxpool_fwd_kernel<true, true>), dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)y,
                       (bf16*)out, (unsigned char*)idx, B, H, W, C, Ho, Wo, scale, shift, (bf16*)ywin,
                       (unsigned char*)bits);
    return pdt_launch_status();
}

PDT_API int pdt_maxpool_bwd(const void* dout, const void* idx, void* dx, int B, int H, int W, int C, int Ho, int Wo,
                            hipStream_t s) {
    if (C % 8) return -1;
    long long total = (long long)B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)dout,
                       (const unsigned char*)idx, (bf16*)dx, B, H, W, C, Ho, Wo);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// Global average pool [B, HW, C] -> [B, C] and its backward (broadcast / HW).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int B,
                                                           int HW, int C) {
    int cvec = C / 8;
    long long total = (long long)B * cvec;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int cv = (int)(i % cvec);
    int b = (int)(i / cvec);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    const bf16* p = x + (long long)b * HW * C + cv * 8;
    for (int j = 0; j < HW; ++j) {
        bf16x8 v = ld_bf16x8(p + (long long)j * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
    }
    float inv = 1.0f / (float)HW;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = f2bf(acc[k] * inv);
    st_bf16x8(out + i * 8, o);
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16* __restrict__ dp,
                                                           const bf16* __restrict__ relu_mask,
                                                           bf16* __restrict__ dx, int B, int HW, int C) {
    int cvec = C / 8;
    long long total = (long long)B * HW * cvec;
    long long stride = (long long)gridDim.x * blockDim.x;
    float inv = 1.0f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int cv = (int)(i % cvec);
        int b = (int)(i / ((long long)HW * cvec));
        bf16x8 v = ld_bf16x8(dp + ((long long)b * cvec + cv) * 8);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(v[k]) * inv);
        if (relu_mask) {
            bf16x8 mk = ld_bf16x8(relu_mask + i * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = bf2f(mk[k]) > 0.0f ? o[k] : f2bf(0.0f);
        }
        st_bf16x8(dx + i * 8, o);
    }
}

PDT_API int pdt_avgpool_fwd(const void* x, void* out, int B, int HW, int C, hipStream_t s) {
    if (C % 8) return -1;
    long long total = (long long)B * (C / 8);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, (const bf16*)x, (bf16*)out, B, HW, C);
    return pdt_launch_status();
}

PDT_API int pdt_avgpool_bwd(const void* dp, const void* relu_mask, void* dx, int B, int HW, int C, hipStream_t s) {
    if (C % 8) return -1;
    long long total = (long long)B * HW * (C / 8);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)dp,
                       (const bf16*)relu_mask, (bf16*)dx, B, HW, C);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------
// Weight repacking (tiny, once per optimizer step).
//   transpose:  w[Co][T][Ci] -> wt[Ci][T][Co_pad]  (data-gradient operand; Co_pad >= Co, pad = 0)
//   stem pack:  w[64][7][7][3] -> wp[64][7][32]     (k = s*4 + c, zero where c == 3 or s == 7)
//   stem unpack-add: dwp[64][7][32] fp32 -> dw[64][7][7][3] += ...
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_w_kernel(const bf16* __restrict__ w, bf16* __restrict__ wt, int Co,
                                                           int T, int Ci, int Co_pad) {
    long long total = (long long)Ci * T * Co_pad;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int co = (int)(i % Co_pad);
        long long t2 = i / Co_pad;
        int t = (int)(t2 % T);
        int ci = (int)(t2 / T);
        wt[i] = (co < Co) ? w[((long long)co * T + t) * Ci + ci] : f2bf(0.0f);
    }
}

PDT_API int pdt_transpose_w(const void* w, void* wt, int Co, int T, int Ci, int Co_pad, hipStream_t s) {
    long long total = (long long)Ci * T * Co_pad;
    hipLaunchKernelGGL(transpose_w_kernel, dim3(ew_grid(total, 1)), dim3(256), 0, s, (const bf16*)w, (bf16*)wt, Co, T, Ci, Co_pad);
    return pdt_launch_status();
}

// All data-gradient weight copies in ONE launch.  Every conv weight sits at the same element
// offset in the flat compute mirror (w_base, [Co][T][Ci]) and in the flat transposed buffer
// (wt_base, [Ci][T][Co]); table row e = {offset, Co, T, Ci, first_block, n_blocks, tiled}.
// tiled = 1 (Co and Ci multiples of 64, every ResNet body layer): a block transposes one 64 x 64
// (co, ci) tile of one tap through LDS -- 128-byte rows in, 128-byte rows out; a plain gather reads
// 2 bytes per cache line.  tiled = 0: a block covers TW_PER_BLOCK consecutive output elements.
#define TW_PER_BLOCK 2048
#define TW_COLS 7
__global__ __launch_bounds__(256) void transpose_w_batched_kernel(const bf16* __restrict__ w_base,
                                                                   bf16* __restrict__ wt_base,
                                                                   const int* __restrict__ table, int n_entries) {
    __shared__ bf16 tile[64][64 + 2];
    const int b = blockIdx.x;
    int lo = 0, hi = n_entries - 1;           // last entry whose first_block <= b
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (table[mid * TW_COLS + 4] <= b) lo = mid; else hi = mid - 1;
    }
    const int* e = table + lo * TW_COLS;
    const int off = e[0], Co = e[1], T = e[2], Ci = e[3], first = e[4], tiled = e[6];
    const bf16* w = w_base + off;
    bf16* wt = wt_base + off;
    if (tiled) {
        const int nci = Ci / 64, nco = Co / 64;
        int q = b - first;
        const int ci0 = (q % nci) * 64; q /= nci;
        const int co0 = (q % nco) * 64;
        const int t = q / nco;
        // in: rows co0..co0+63, 64 consecutive ci each (8 vectors of 16 B per row, 2 vectors per thread)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int v = threadIdx.x + k * 256;
            int r = v >> 3, c8 = (v & 7) * 8;
            bf16x8 x = ld_bf16x8(w + ((size_t)(co0 + r) * T + t) * Ci + ci0 + c8);
#pragma unroll
            for (int u = 0; u < 8; ++u) tile[r][c8 + u] = x[u];
        }
        __syncthreads();
        // out: rows ci0..ci0+63, 64 consecutive co each
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int v = threadIdx.x + k * 256;
            int r = v >> 3, c8 = (v & 7) * 8;
            bf16x8 y;
#pragma unroll
            for (int u = 0; u < 8; ++u) y[u] = tile[c8 + u][r];
            st_bf16x8(wt + ((size_t)(ci0 + r) * T + t) * Co + co0 + c8, y);
        }
        return;
    }
    const int total = Ci * T * Co;
    const int start = (b - first) * TW_PER_BLOCK;
#pragma unroll
    for (int k = 0; k < TW_PER_BLOCK / 256; ++k) {
        int i = start + k * 256 + threadIdx.x;
        if (i < total) {
            int co = i % Co;
            int t2 = i / Co;
            int t = t2 % T;
            int ci = t2 / T;
            wt[i] = w[(co * T + t) * Ci + ci];
        }
    }
}

PDT_API int pdt_transpose_w_batched(const void* w_base, void* wt_base, const int* table, int n_entries,
                                    int total_blocks, hipStream_t s) {
    if (n_entries <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(transpose_w_batched_kernel, dim3(total_blocks), dim3(256), 0, s, (const bf16*)w_base,
                       (bf16*)wt_base, table, n_entries);
    return pdt_launch_status();
}

__global__ __launch_bounds__(256) void stem_pack_kernel(const bf16* __restrict__ w, bf16* __restrict__ wp, int Co) {
    int total = Co * 7 * 32;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int k = i % 32;
    int r = (i / 32) % 7;
    int co = i / (32 * 7);
    int s2 = k / 4, c = k % 4;
    wp[i] = (s2 < 7 && c < 3) ? w[((co * 7 + r) * 7 + s2) * 3 + c] : f2bf(0.0f);
}

__global__ __launch_bounds__(256) void stem_unpack_add_kernel(const float* __restrict__ dwp, float* __restrict__ dw,
                                                               int Co) {
    int total = Co * 7 * 7 * 3;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = i % 3;
    int s2 = (i / 3) % 7;
    int r = (i / 21) % 7;
    int co = i / 147;
    dw[i] += dwp[(co * 7 + r) * 32 + s2 * 4 + c];
}

PDT_API int pdt_stem_pack(const void* w, void* wp, int Co, hipStream_t s) {
    hipLaunchKernelGGL(stem_pack_kernel, dim3(cdiv(Co * 7 * 32, 256)), dim3(256), 0, s, (const bf16*)w, (bf16*)wp, Co);
    return pdt_launch_status();
}

PDT_API int pdt_stem_unpack_add(const float* dwp, float* dw, int Co, hipStream_t s) {
    hipLaunchKernelGGL(stem_unpack_add_kernel, dim3(cdiv(Co * 147, 256)), dim3(256), 0, s, dwp, dw, Co);
    return pdt_launch_status();
}


// ------------------------------------------------------------------------------------------
// out[b][y][x][:] = in[b][2y][2x][:]  -- the pixels a stride-2 1x1 convolution reads, made dense (so that the products
// of a folded projection shortcut, bn_fold.hip, see an ordinary row-aligned operand)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subsample2_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int Ho,
                                                          int Wo, int cvec, long long nvec) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const int cv = (int)(i % cvec);
        long long pix = i / cvec;
        const int x = (int)(pix % Wo);
        pix /= Wo;
        const int y = (int)(pix % Ho);
        const long long b = pix / Ho;
        const long long src = ((b * (2 * Ho) + 2 * y) * (2 * Wo) + 2 * x) * cvec + cv;
        st_bf16x8(out + i * 8, ld_bf16x8(in + src * 8));
    }
}

PDT_API int pdt_subsample2(const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
    if (C % 8 != 0 || (H & 1) || (W & 1) || B <= 0) return -1;
    const long long nvec = (long long)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(subsample2_kernel, dim3(ew_grid(nvec, 2)), dim3(256), 0, s, (const bf16*)in, (bf16*)out, H / 2, W / 2,
                       C / 8, nvec);
    return pdt_launch_status();
}


// ---- small runtime helpers (kept behind the C ABI so a recorded step can replay them) ---------
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0.0f;
}

PDT_API int pdt_zero_f32(float* p, long long n, hipStream_t s) {
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
    return pdt_launch_status();
}

PDT_API int pdt_event_create(void** out) {
    // The events only order streams of one device against each other: no timestamps, and no system-scope fence (a
    // device-scope release is all the waiting stream needs; the system-scope one writes the L2 back at every record).
    // PDT_EVENT_SYSTEM_FENCE=1 restores the default fence (A/B measurements).
    static const bool fence = [] { const char* v = getenv("PDT_EVENT_SYSTEM_FENCE"); return v && v[0] == '1'; }();
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | (fence ? 0u : hipEventDisableSystemFence));
    if (e != hipSuccess) return (int)e;
    *out = (void*)ev;
    return 0;
}

PDT_API int pdt_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : 0; }

// Work enqueued on `waiter` after this call starts only when everything enqueued on `waited`
// before it has finished.
PDT_API int pdt_stream_wait_stream(hipStream_t waiter, hipStream_t waited, void* ev) {
    hipError_t e = hipEventRecord((hipEvent_t)ev, waited);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamWaitEvent(waiter, (hipEvent_t)ev, 0);
}
