// Weight-gradient convolution for gfx950 (MI355X), NHWC bf16 activations, fp32 output:
//
//   dw[n][widx(t)*Ci + c] += sum_{m in split} dy[m][n] * in[pix(m, t)][c]
//
// i.e. for every filter tap t a GEMM  dW_t[Cout x Cin] = dY^T[Cout x P] . X_t[P x Cin]  whose
// reduction runs over the P = B*Ho*Wo output pixels.  Both operands are stored pixel-major
// (the reduction index is the slow one), which is the wrong way round for the MFMA operand
// layout (each lane wants 8 consecutive k for one row/column).  The tiles are therefore staged
// row-major [pixel][channel] in LDS and read with the gfx950 transposing LDS load
// ds_read_b64_tr_b16: one 16-lane group fetches 4 pixels x 16 channels and gets them delivered
// channel-major.  The pixel <-> MFMA-k assignment is free as long as A and B agree, so lane
// group g takes pixels {4g..4g+3} and {16+4g..16+4g+3} of each 32-pixel step; with the row
// stride padded to (channels*2 + 32) bytes the 8 rows a 32-lane half touches fall into 8
// distinct 32-byte bank ranges: conflict-free.
//
// The reduction is huge for the early layers (B*56*56) and the output tiny, so parallelism
// comes from split-K: each block reduces one slice of pixels for one (Cout tile, Cin tile,
// tap) and adds its fp32 tile into the flat gradient buffer with float atomics (the buffer
// is zeroed by the fused optimizer kernel).  Workgroups that share the same dY slice are
// adjacent after the XCD remap, so re-reads of dY across taps / Cin tiles hit in L2.
//
// Deterministic variant (a workspace is passed): slice s STORES its tile into slab s of the workspace
// (every slab element has exactly one writer) and a second kernel adds the slabs to the gradient in
// slice order -- no float atomics, bit-identical results from run to run, one more pass over
// split x output bytes.
#include "common.h"

// Measurement switches (read once): PDT_WGRAD_DMA=1 lets the automatic choice use the LDS-DMA 128 x 128 kernel
// (off by default: inside the training step it measured 0.9 ms slower than the register-staged kernel, because the
// weight gradients run in the shadow of the critical stream and the DMA version takes more bandwidth from it);
// PDT_WGRAD_BLOCKS / PDT_WGRAD_ATOMIC_MB override the split-K sizing defaults (768 workgroups = three per CU / 24-32
// MB by batch: the step is fastest there at per-GPU batch 512 and 1024 -- 1024 workgroups are 0.15-0.2 ms slower,
// 512 and 640 0.5-0.7 ms, 2048 0.6 ms; the budget's measurements are next to its rule in pdt_conv_wgrad_ex).
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
static const int g_pdt_wgrad_dma = env_int("PDT_WGRAD_DMA", 0);
static const int g_pdt_wgrad_blocks = env_int("PDT_WGRAD_BLOCKS", 768);    // tuned inside the step, not stand-alone (three per CU)
static const int g_pdt_wgrad_atomic_mb = env_int("PDT_WGRAD_ATOMIC_MB", 0);     // 0: by batch (see pdt_conv_wgrad_ex)
static const int g_pdt_wgrad_big_tile = env_int("PDT_WGRAD_BIG_TILE", 1);      // tile of the layers with N, Ci > 64
// layers with 64 channels on ONE side: 1 = a tile that is 128 wide on the other side (the 64-wide operand is read twice
// per 128 instead of twice per 64 channels of the other one), 0 = 64 x 64
static const int g_pdt_wgrad_wide_small = env_int("PDT_WGRAD_WIDE_SMALL", 1);
// 3x3 / stride 1 / padding 1: one kernel row per workgroup with the gathered operand staged once (0: one tap per workgroup)
static const int g_pdt_wgrad_row3 = env_int("PDT_WGRAD_ROW3", 1);
static const int g_pdt_wgrad_row3_min_w = env_int("PDT_WGRAD_ROW3_MIN_W", 0);   // ... on images at least this wide

#define MAX_TAPS 52
#define BKP 64   // pixels per K step

struct WgradParams {
    const bf16* dy;
    const bf16* in;
    float* dw;
    float* ws;                         // deterministic mode: [splitk][slab] partial outputs, else NULL
    long long slab;                    // floats per slab (= N * wrow)
    long long M;
    unsigned long long in_bytes, dy_bytes;   // tensor sizes; descriptors are based at the workgroup's slice (see slice_rsrc)
    int Hi, Wi, pix_stride, Ci;
    int Hg, Wg, IS;
    int N, ldy, wrow, T;
    int splitk;
    long long rows_per_split;
    int step_b, step_y, step_x;        // BKP GEMM rows as (images, grid rows, grid columns): see row_advance
    int use_dma;                       // 128 x 128 tile: LDS-DMA staging (needs 16-byte aligned pixel rows)
    int lp;                            // wgrad_row3_kernel: log2 of the positions per image row
    const float* row_scale;            // [N] factor on output row n (null: 1): dW of a per-channel scaled dY without
                                       // ever forming that tensor (folded BatchNorm backward, bn_fold.hip)
    // one 32-bit word per tap: offy (int8) | offx (int8) << 8 | widx (uint16) << 16.  Sub-dword
    // entries would be fetched with VECTOR loads (no scalar byte loads on gfx9), whose waits
    // drain the tile prefetches; a dword indexed by a wave-uniform tap is one s_load_dword.
    int taps[MAX_TAPS];
};

__host__ __device__ __forceinline__ void unpack_tap(int w, int& oy, int& ox, int& widx) {
    oy = (int)(signed char)(w & 0xff);
    ox = (int)(signed char)((w >> 8) & 0xff);
    widx = (w >> 16) & 0xffff;
}

// Descriptor for ``bytes`` of a tensor from byte ``off`` on: offsets inside a slice are 32-bit, the slice's start is
// not -- one launch covers tensors of any size (the launcher keeps what one slice reaches below 2 GiB).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const void* base, unsigned long long off,
                                                             unsigned long long total) {
    const unsigned long long left = total > off ? total - off : 0ull;
    return make_rsrc(static_cast<const unsigned char*>(base) + off, left < 0x7fff0000ull ? (unsigned)left : 0x7fff0000u);
}

__device__ __forceinline__ bf16x8 tr_read_pair(const unsigned char* base0, const unsigned char* base1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// Position of a GEMM row (an output pixel) as (image, y, x) on the Hg x Wg grid, advanced by one K step (BKP rows) at a
// time with the launch-constant increment (step_b, step_y, step_x): two compares and a handful of adds per step instead
// of two integer divisions per staged row (which made the main loop VALU-bound: ~200 vector instructions against 32 MFMAs).
struct RowPos {
    int b, y, x;
};

__device__ __forceinline__ RowPos row_pos(unsigned m, int HgWg, int Wg) {
    RowPos r;
    r.b = (int)(m / (unsigned)HgWg);
    int rem = (int)(m - (unsigned)r.b * (unsigned)HgWg);
    r.y = (int)((unsigned)rem / (unsigned)Wg);
    r.x = rem - r.y * Wg;
    return r;
}

__device__ __forceinline__ void row_advance(RowPos& r, const WgradParams& p) {
    r.x += p.step_x;
    int cx = r.x >= p.Wg;
    r.x -= cx ? p.Wg : 0;
    r.y += p.step_y + cx;
    int cy = r.y >= p.Hg;
    r.y -= cy ? p.Hg : 0;
    r.b += p.step_b + cy;
}

// LINEAR: the gathered operand is the input tensor itself (1x1 filter, stride 1, no padding): GEMM row m reads input
// pixel m, so both operands advance by a constant byte stride per K step and the slice end is enforced by the buffer
// descriptors' sizes -- no per-row index arithmetic at all.
template <int BMO, int BNC, int WM, int WN, bool LINEAR>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    constexpr int RSA = (BMO + 16) * 2;       // LDS row stride (bytes), dY tile
    constexpr int RSB = (BNC + 16) * 2;       // LDS row stride (bytes), X tile
    constexpr int A_BYTES = BKP * RSA, B_BYTES = BKP * RSB;
    constexpr int AV = BMO / 8, BV = BNC / 8; // 16-byte vectors per tile row
    constexpr int A_RPP = 256 / AV, B_RPP = 256 / BV;
    constexpr int A_IT = BKP / A_RPP, B_IT = BKP / B_RPP;
    constexpr int WTM = BMO / WM, WTN = BNC / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    static_assert(A_RPP <= BKP && BKP % A_RPP == 0, "every thread stages whole rows of the dY tile");
    static_assert(B_RPP <= BKP && BKP % B_RPP == 0, "every thread stages whole rows of the X tile");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int n_ct = (p.N + BMO - 1) / BMO;
    const int n_cin = (p.Ci + BNC - 1) / BNC;
    const int inner = p.T * n_cin;
    const int nwg = p.splitk * n_ct * inner;
    int wgid;
    {
        int orig = blockIdx.x;
        int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int in_idx = wgid % inner;
    const int rest = wgid / inner;
    const int ct = rest % n_ct;
    const int split = rest / n_ct;
    const int tap = in_idx / n_cin;
    const int cin_tile = in_idx % n_cin;
    const int co0 = ct * BMO, ci0 = cin_tile * BNC;
    int oy, ox, wi_tap;
    unpack_tap(p.taps[__builtin_amdgcn_readfirstlane(tap)], oy, ox, wi_tap);

    // M < 2^31 (checked by the launcher): 32-bit row arithmetic
    const unsigned r_begin = (unsigned)((long long)split * p.rows_per_split);
    unsigned r_end = r_begin + (unsigned)p.rows_per_split;
    if (r_end > (unsigned)p.M) r_end = (unsigned)p.M;
    const int nk = r_begin < r_end ? (int)((r_end - r_begin + BKP - 1) / BKP) : 0;
    if (nk == 0) return;

    const int a_cv = tid % AV, a_row = tid / AV;
    const int b_cv = tid % BV, b_row = tid / BV;
    const bool a_col_ok = co0 + a_cv * 8 < p.N;          // N is checked at 8-granularity (ldy padded)
    const bool b_col_ok = ci0 + b_cv * 8 < p.Ci;

    // The descriptors start at this slice's first row (gathered operand: first image) and the dY one ends at its last
    // row: rows past the slice read zeros through the range check.
    const unsigned b_first = LINEAR ? 0u : r_begin / (unsigned)(p.Hg * p.Wg);
    const __amdgpu_buffer_rsrc_t dy_rsrc = slice_rsrc(p.dy, (unsigned long long)r_begin * (unsigned)p.ldy * 2ull,
                                                      (unsigned long long)r_end * (unsigned)p.ldy * 2ull);
    const __amdgpu_buffer_rsrc_t in_rsrc =
        LINEAR ? slice_rsrc(p.in, (unsigned long long)r_begin * (unsigned)p.pix_stride * 2ull,
                            (unsigned long long)r_end * (unsigned)p.pix_stride * 2ull)
               : slice_rsrc(p.in, (unsigned long long)b_first * (unsigned)(p.Hi * p.Wi) * (unsigned)p.pix_stride * 2ull, p.in_bytes);
    // two register sets: the loads of K step s+2 are issued before step s is multiplied, so a load has two steps of
    // MFMA work (not one) to come back before its ds_write needs it
    bf16x8 a_reg0[A_IT], b_reg0[B_IT], a_reg1[A_IT], b_reg1[B_IT];
    unsigned a_off[A_IT], b_off[B_IT];
    RowPos pos[B_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        unsigned m = a_row + j * A_RPP;                    // relative to the slice
        a_off[j] = a_col_ok ? (m * (unsigned)p.ldy + co0 + a_cv * 8) * 2u : PDT_OOB_OFFSET;
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        unsigned m = b_row + j * B_RPP;
        if (LINEAR) b_off[j] = b_col_ok ? (m * (unsigned)p.pix_stride + ci0 + b_cv * 8) * 2u : PDT_OOB_OFFSET;
        else {
            pos[j] = row_pos(r_begin + m, p.Hg * p.Wg, p.Wg);
            pos[j].b -= (int)b_first;
        }
    }
    const unsigned a_step = a_col_ok ? (unsigned)(BKP * p.ldy * 2) : 0u;
    const unsigned b_step = b_col_ok ? (unsigned)(BKP * p.pix_stride * 2) : 0u;
    const unsigned b_col = (unsigned)(ci0 + b_cv * 8);
    unsigned m_b = r_begin + b_row;                        // GEMM row of this thread's first X row in the coming K step

    // branch-free staging loads: rows outside the slice / tile / image read zeros through the range check
    auto load_tiles = [&](bf16x8 (&a_reg)[A_IT], bf16x8 (&b_reg)[B_IT]) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            a_reg[j] = buf_ld_bf16x8(dy_rsrc, a_off[j]);
            a_off[j] += a_step;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (LINEAR) {
                b_reg[j] = buf_ld_bf16x8(in_rsrc, b_off[j]);
                b_off[j] += b_step;
            } else {
                int iy = pos[j].y * p.IS + oy, ix = pos[j].x * p.IS + ox;
                bool ok = m_b + j * B_RPP < r_end && b_col_ok && (unsigned)iy < (unsigned)p.Hi &&
                          (unsigned)ix < (unsigned)p.Wi;
                unsigned off = (unsigned)(((pos[j].b * p.Hi + iy) * p.Wi + ix) * p.pix_stride + b_col) * 2u;
                b_reg[j] = buf_ld_bf16x8(in_rsrc, ok ? off : PDT_OOB_OFFSET);
                row_advance(pos[j], p);
            }
        }
        m_b += BKP;
    };
    auto store_tiles = [&](int buf, const bf16x8 (&a_reg)[A_IT], const bf16x8 (&b_reg)[B_IT]) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j)
            *reinterpret_cast<bf16x8*>(As + (a_row + j * A_RPP) * RSA + a_cv * 16) = a_reg[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            *reinterpret_cast<bf16x8*>(Bs + (b_row + j * B_RPP) * RSB + b_cv * 16) = b_reg[j];
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry: group g = lane>>4 ; inside the group lane 4q+pp supplies
    // the address of pixel row q, channels 4pp..4pp+3.
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int trow0 = 4 * g + q4;             // pixel row of read #0 inside a 32-pixel step
    const int tcol = 4 * pp;                  // channel offset inside the 16-channel block

    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const int r0 = kk * 32 + trow0, r1 = r0 + 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int col = wm * WTM + i * 16 + tcol;
                af[i] = tr_read_pair(As + r0 * RSA + col * 2, As + r1 * RSA + col * 2);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int col = wn * WTN + j * 16 + tcol;
                bfr[j] = tr_read_pair(Bs + r0 * RSB + col * 2, Bs + r1 * RSB + col * 2);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    load_tiles(a_reg0, b_reg0);                            // step 0
    store_tiles(0, a_reg0, b_reg0);
    if (nk > 1) load_tiles(a_reg0, b_reg0);                // step 1, in flight
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        // LDS stage 0 holds step ks, set 0 holds step ks+1 (in flight)
        if (ks + 2 < nk) load_tiles(a_reg1, b_reg1);
        compute(0);
        if (ks + 1 >= nk) break;
        store_tiles(1, a_reg0, b_reg0);                    // waits for set 0 only: set 1 was requested later
        __syncthreads();
        // LDS stage 1 holds step ks+1, set 1 holds step ks+2 (in flight)
        if (ks + 3 < nk) load_tiles(a_reg0, b_reg0);
        compute(1);
        if (ks + 2 < nk) store_tiles(0, a_reg1, b_reg1);
        __syncthreads();
    }

    const int frow = lane & 15, fq = lane >> 4;
    const long long wofs = (long long)wi_tap * p.Ci;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int c = ci0 + wn * WTN + j * 16 + frow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = co0 + wm * WTM + i * 16 + fq * 4 + r;
                if (n < p.N && c < p.Ci) {
                    const float v = p.row_scale ? acc[i][j][r] * p.row_scale[n] : acc[i][j][r];
                    if (p.ws) p.ws[(long long)split * p.slab + (long long)n * p.wrow + wofs + c] = v;
                    else atomicAdd(p.dw + (long long)n * p.wrow + wofs + c, v);
                }
            }
        }
}

// LDS-DMA variant of the 128 x 128 tile: both operand tiles go global -> LDS with buffer_load ... lds (no staging
// registers, no ds_write pass), two stages of 32 KB.  One DMA instruction fills 1 KB = four 256-byte pixel rows; the
// LDS image is linear, so the bank swizzle of the transposing fragment reads is applied to the per-lane SOURCE address:
// LDS chunk c (16 bytes = 8 channels) of row r holds source chunk c ^ 2(r & 7).  A ds_read_b64_tr_b16 serves 32 lanes
// per cycle = 8 consecutive rows x 32 bytes; with that swizzle the 8 rows land in 8 different 32-byte bank groups.
// Rows past the slice / image / channel range read zeros through the descriptor's range check, which the DMA writes
// to LDS like any other data.
typedef __attribute__((address_space(3))) void wg_lds_void_t;

template <bool LINEAR>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams p) {
    constexpr int BMO = 128, BNC = 128, WM = 2, WN = 2;
    constexpr int ROWB = 256;                              // bytes per LDS row (128 channels)
    constexpr int A_BYTES = BKP * ROWB, B_BYTES = BKP * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int IT = BKP / 4 / 4;                        // DMA instructions per wave, operand and K step (4 rows each)
    constexpr int WTM = BMO / WM, WTN = BNC / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int n_ct = (p.N + BMO - 1) / BMO;
    const int n_cin = (p.Ci + BNC - 1) / BNC;
    const int inner = p.T * n_cin;
    const int nwg = p.splitk * n_ct * inner;
    int wgid;
    {
        int orig = blockIdx.x;
        int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int in_idx = wgid % inner;
    const int rest = wgid / inner;
    const int ct = rest % n_ct;
    const int split = rest / n_ct;
    const int tap = in_idx / n_cin;
    const int cin_tile = in_idx % n_cin;
    const int co0 = ct * BMO, ci0 = cin_tile * BNC;
    int oy, ox, wi_tap;
    unpack_tap(p.taps[__builtin_amdgcn_readfirstlane(tap)], oy, ox, wi_tap);

    const unsigned r_begin = (unsigned)((long long)split * p.rows_per_split);
    unsigned r_end = r_begin + (unsigned)p.rows_per_split;
    if (r_end > (unsigned)p.M) r_end = (unsigned)p.M;
    const int nk = r_begin < r_end ? (int)((r_end - r_begin + BKP - 1) / BKP) : 0;
    if (nk == 0) return;

    const unsigned b_first = LINEAR ? 0u : r_begin / (unsigned)(p.Hg * p.Wg);
    const __amdgpu_buffer_rsrc_t dy_rsrc = slice_rsrc(p.dy, (unsigned long long)r_begin * (unsigned)p.ldy * 2ull,
                                                      (unsigned long long)r_end * (unsigned)p.ldy * 2ull);
    const __amdgpu_buffer_rsrc_t in_rsrc =
        LINEAR ? slice_rsrc(p.in, (unsigned long long)r_begin * (unsigned)p.pix_stride * 2ull,
                            (unsigned long long)r_end * (unsigned)p.pix_stride * 2ull)
               : slice_rsrc(p.in, (unsigned long long)b_first * (unsigned)(p.Hi * p.Wi) * (unsigned)p.pix_stride * 2ull, p.in_bytes);

    // DMA instruction j of this wave fills rows (j * 4 + wave) * 4 + (lane >> 4) of both tiles
    const int lrow = lane >> 4, lchunk = lane & 15;
    unsigned a_off[IT], b_off[IT];
    RowPos pos[IT];
    unsigned b_colb[IT];                                   // general path: byte offset of this lane's source chunk
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int row = (j * 4 + wave) * 4 + lrow;
        const int src_chunk = lchunk ^ ((row & 7) * 2);
        const unsigned m = (unsigned)row;                  // relative to the slice
        const bool a_ok = co0 + src_chunk * 8 < p.N;
        const bool b_ok = ci0 + src_chunk * 8 < p.Ci;
        a_off[j] = a_ok ? (m * (unsigned)p.ldy + co0 + src_chunk * 8) * 2u : PDT_OOB_OFFSET;
        if (LINEAR) {
            b_off[j] = b_ok ? (m * (unsigned)p.pix_stride + ci0 + src_chunk * 8) * 2u : PDT_OOB_OFFSET;
        } else {
            pos[j] = row_pos(r_begin + m, p.Hg * p.Wg, p.Wg);
            pos[j].b -= (int)b_first;
            b_colb[j] = b_ok ? (unsigned)(ci0 + src_chunk * 8) * 2u : PDT_OOB_OFFSET;
        }
    }
    // steps are zero for lanes parked on the out-of-range offset (their chunk lies beyond the channel count)
    unsigned a_step[IT], b_step[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        a_step[j] = a_off[j] == PDT_OOB_OFFSET ? 0u : (unsigned)(BKP * p.ldy * 2);
        b_step[j] = (!LINEAR || b_off[j] == PDT_OOB_OFFSET) ? 0u : (unsigned)(BKP * p.pix_stride * 2);
    }
    unsigned m_b = r_begin + wave * 4 + lrow;              // GEMM row of this lane's first X row in the coming K step

    auto issue = [&](int buf) {
        unsigned char* As = smem + buf * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rsrc, (wg_lds_void_t*)(As + (j * 4 + wave) * 1024), 16,
                                                     (int)a_off[j], 0, 0, 0);
            a_off[j] += a_step[j];
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            unsigned off;
            if (LINEAR) {
                off = b_off[j];
                b_off[j] += b_step[j];
            } else {
                const int iy = pos[j].y * p.IS + oy, ix = pos[j].x * p.IS + ox;
                const bool ok = m_b + j * 16 < r_end && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                                b_colb[j] != PDT_OOB_OFFSET;
                off = ok ? (unsigned)(((pos[j].b * p.Hi + iy) * p.Wi + ix) * p.pix_stride) * 2u + b_colb[j] : PDT_OOB_OFFSET;
                row_advance(pos[j], p);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (wg_lds_void_t*)(Bs + (j * 4 + wave) * 1024), 16, (int)off,
                                                     0, 0, 0);
        }
        m_b += BKP;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane geometry (see wgrad_kernel) with the swizzled chunk position; (row & 7) is the same for
    // both reads of a pair and for every 32-pixel sub-step, so the column offsets are loop constants
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int trow0 = 4 * g + q4;
    const int sw = (trow0 & 7) * 2;
    int acol[MI], bcol[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) acol[i] = (((wm * WTM + i * 16) / 8 + (pp >> 1)) ^ sw) * 16 + (pp & 1) * 8;
#pragma unroll
    for (int j = 0; j < NI; ++j) bcol[j] = (((wn * WTN + j * 16) / 8 + (pp >> 1)) ^ sw) * 16 + (pp & 1) * 8;

    issue(0);
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        // this wave's share of stage ks has landed ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and every other wave's; all waves are also done reading the other stage (step ks - 1)
        __builtin_amdgcn_s_barrier();
        if (ks + 1 < nk) issue(buf ^ 1);
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const int r0 = kk * 32 + trow0, r1 = r0 + 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = tr_read_pair(As + r0 * ROWB + acol[i], As + r1 * ROWB + acol[i]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = tr_read_pair(Bs + r0 * ROWB + bcol[j], Bs + r1 * ROWB + bcol[j]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    const int frow = lane & 15, fq = lane >> 4;
    const long long wofs = (long long)wi_tap * p.Ci;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int c = ci0 + wn * WTN + j * 16 + frow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = co0 + wm * WTM + i * 16 + fq * 4 + r;
                if (n < p.N && c < p.Ci) {
                    const float v = p.row_scale ? acc[i][j][r] * p.row_scale[n] : acc[i][j][r];
                    if (p.ws) p.ws[(long long)split * p.slab + (long long)n * p.wrow + wofs + c] = v;
                    else atomicAdd(p.dw + (long long)n * p.wrow + wofs + c, v);
                }
            }
        }
}

// Variant for layers with a narrow gathered operand and several taps (the ResNet stem on its packed
// NHWC4 image: 7 taps of 32 values): one workgroup owns ALL taps of its (Cout tile, pixel slice), so a
// dY tile is loaded once per K step and multiplied against TAPS gathered X tiles -- with one tap per
// workgroup the same dY tile crosses L2 -> LDS seven times and that traffic is all the kernel does.
// Single LDS stage (A + TAPS x B stays under 64 KB, three workgroups per CU cover each other's barriers),
// the next step's tiles wait in registers.
template <int BMO, int BNC, int WM, int WN, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_taps_kernel(const WgradParams p) {
    constexpr int RSA = (BMO + 16) * 2;
    constexpr int RSB = (BNC + 16) * 2;
    constexpr int A_BYTES = BKP * RSA, B_BYTES = BKP * RSB;
    constexpr int AV = BMO / 8, BV = BNC / 8;
    constexpr int A_RPP = 256 / AV, B_RPP = 256 / BV;
    constexpr int A_IT = (BKP + A_RPP - 1) / A_RPP, B_IT = (BKP + B_RPP - 1) / B_RPP;
    constexpr int WTM = BMO / WM, WTN = BNC / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    static_assert(A_BYTES + TAPS * B_BYTES <= 65536, "one LDS stage must fit the static limit");
    static_assert(BKP % A_RPP == 0 && BKP % B_RPP == 0, "tile/thread mismatch");

    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + TAPS * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int n_ct = (p.N + BMO - 1) / BMO;
    const int groups = p.T / TAPS;            // tap groups (a 3x3 filter: one kernel row per workgroup)
    const int nwg = p.splitk * n_ct * groups;
    int wgid;
    {
        int orig = blockIdx.x;
        int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int grp = __builtin_amdgcn_readfirstlane(wgid % groups);
    const int rest = wgid / groups;
    const int ct = rest % n_ct;
    const int split = rest / n_ct;
    const int co0 = ct * BMO;
    int oy[TAPS], ox[TAPS], wi_tap[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) unpack_tap(p.taps[grp * TAPS + t], oy[t], ox[t], wi_tap[t]);

    const unsigned r_begin = (unsigned)((long long)split * p.rows_per_split);
    unsigned r_end = r_begin + (unsigned)p.rows_per_split;
    if (r_end > (unsigned)p.M) r_end = (unsigned)p.M;
    const int nk = r_begin < r_end ? (int)((r_end - r_begin + BKP - 1) / BKP) : 0;
    if (nk == 0) return;

    const int a_cv = tid % AV, a_row = tid / AV;
    const int b_cv = tid % BV, b_row = tid / BV;
    const bool a_col_ok = co0 + a_cv * 8 < p.N;
    const bool b_col_ok = b_cv * 8 < p.Ci;

    const unsigned b_first = r_begin / (unsigned)(p.Hg * p.Wg);
    const __amdgpu_buffer_rsrc_t dy_rsrc = slice_rsrc(p.dy, (unsigned long long)r_begin * (unsigned)p.ldy * 2ull,
                                                      (unsigned long long)r_end * (unsigned)p.ldy * 2ull);
    const __amdgpu_buffer_rsrc_t in_rsrc =
        slice_rsrc(p.in, (unsigned long long)b_first * (unsigned)(p.Hi * p.Wi) * (unsigned)p.pix_stride * 2ull, p.in_bytes);
    bf16x8 a_reg[A_IT], b_reg[TAPS][B_IT];
    unsigned a_off[A_IT];
    RowPos pos[B_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        unsigned m = a_row + j * A_RPP;                    // relative to the slice
        a_off[j] = a_col_ok ? (m * (unsigned)p.ldy + co0 + a_cv * 8) * 2u : PDT_OOB_OFFSET;
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        pos[j] = row_pos(r_begin + b_row + j * B_RPP, p.Hg * p.Wg, p.Wg);
        pos[j].b -= (int)b_first;
    }
    const unsigned a_step = a_col_ok ? (unsigned)(BKP * p.ldy * 2) : 0u;
    unsigned m_b = r_begin + b_row;
    auto load_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            a_reg[j] = buf_ld_bf16x8(dy_rsrc, a_off[j]);
            a_off[j] += a_step;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const bool row_ok = m_b + j * B_RPP < r_end && b_col_ok;
            const int y0 = pos[j].y * p.IS, x0 = pos[j].x * p.IS;
            const int base = pos[j].b * p.Hi;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                int iy = y0 + oy[t], ix = x0 + ox[t];
                bool ok = row_ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                unsigned off = (unsigned)(((base + iy) * p.Wi + ix) * p.pix_stride + b_cv * 8) * 2u;
                b_reg[t][j] = buf_ld_bf16x8(in_rsrc, ok ? off : PDT_OOB_OFFSET);
            }
            row_advance(pos[j], p);
        }
        m_b += BKP;
    };
    unsigned char* As = smem;
    unsigned char* Bs = smem + A_BYTES;
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int row = a_row + j * A_RPP;
            *reinterpret_cast<bf16x8*>(As + row * RSA + a_cv * 16) = a_reg[j];
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int j = 0; j < B_IT; ++j) {
                int row = b_row + j * B_RPP;
                *reinterpret_cast<bf16x8*>(Bs + t * B_BYTES + row * RSB + b_cv * 16) = b_reg[t][j];
            }
    };

    f32x4 acc[TAPS][MI][NI];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int trow0 = 4 * g + q4;
    const int tcol = 4 * pp;

    load_tiles();
    for (int ks = 0; ks < nk; ++ks) {
        store_tiles();
        __syncthreads();
        if (ks + 1 < nk) load_tiles();
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8 af[MI];
            const int r0 = kk * 32 + trow0, r1 = r0 + 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int col = wm * WTM + i * 16 + tcol;
                af[i] = tr_read_pair(As + r0 * RSA + col * 2, As + r1 * RSA + col * 2);
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                bf16x8 bfr[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn * WTN + j * 16 + tcol;
                    const unsigned char* Bt = Bs + t * B_BYTES;
                    bfr[j] = tr_read_pair(Bt + r0 * RSB + col * 2, Bt + r1 * RSB + col * 2);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[t][i][j], 0, 0, 0);
            }
        }
        __syncthreads();          // every wave is done with this stage before it is overwritten
    }

    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const long long wofs = (long long)wi_tap[t] * p.Ci;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int c = wn * WTN + j * 16 + frow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int n = co0 + wm * WTM + i * 16 + fq * 4 + r;
                    if (n < p.N && c < p.Ci) {
                        const float v = p.row_scale ? acc[t][i][j][r] * p.row_scale[n] : acc[t][i][j][r];
                        if (p.ws) p.ws[(long long)split * p.slab + (long long)n * p.wrow + wofs + c] = v;
                        else atomicAdd(p.dw + (long long)n * p.wrow + wofs + c, v);
                    }
                }
            }
    }
}

// 3x3 / stride 1 / padding 1: one workgroup owns one KERNEL ROW (three taps) of its (Cout tile, Cin tile, pixel slice)
// and stages the gathered operand ONCE for the three taps.  With one tap per workgroup every dY tile and every X tile
// crosses L2 -> LDS nine times, 64 FLOP per byte, and the kernel is bound by that traffic (and takes it from the critical
// stream's kernels it runs beside); here a K step moves one dY tile and one X tile for three taps' worth of multiplies
// (127 FLOP per byte at 128 x 64).
//
// The reduction index is not the output pixel but a POSITION on rows padded to a power of two: an image row of W pixels
// occupies P >= W + 1 positions, q = 0 and q > W being padding (dY reads zero there, X reads zero there).  A K step of
// 64 positions is 64 / P whole image rows, so tap ox = -1 / 0 / +1 multiplies dY[position] with X[position + ox]: the
// SAME LDS tile read one row up or down (the transposing fragment read takes a per-lane row address, so the shift is an
// address offset).  A pixel's left / right neighbour outside the image is the padding position next to it -- zero, as
// the convolution's padding demands -- and the two rows beyond the tile's ends are only ever multiplied with dY's
// padding positions (they are kept zero so that no stale NaN can meet a zero).  Rows above / below the image (oy = +-1)
// are refused at load time, like in the gathering kernels.  With W = 7, 14, 28, 56 the padding costs 1/8 of the MFMA work.
template <int BMO, int BNC, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_row3_kernel(const WgradParams p) {
    constexpr int RSA = (BMO + 16) * 2;
    constexpr int RSB = (BNC + 16) * 2;
    constexpr int A_BYTES = BKP * RSA, B_BYTES = (BKP + 2) * RSB;
    constexpr int AV = BMO / 8, BV = BNC / 8;
    constexpr int A_RPP = 256 / AV, B_RPP = 256 / BV;
    constexpr int A_IT = BKP / A_RPP, B_IT = BKP / B_RPP;
    constexpr int WTM = BMO / WM, WTN = BNC / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    static_assert(2 * (A_BYTES + B_BYTES) <= 65536, "two LDS stages must fit the static limit");
    static_assert(A_RPP <= BKP && BKP % A_RPP == 0 && B_RPP <= BKP && BKP % B_RPP == 0, "tile/thread mismatch");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int n_ct = (p.N + BMO - 1) / BMO;
    const int n_cin = (p.Ci + BNC - 1) / BNC;
    const int inner = 3 * n_cin;
    const int nwg = p.splitk * n_ct * inner;
    int wgid;
    {
        int orig = blockIdx.x;
        int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int in_idx = wgid % inner;
    const int rest = wgid / inner;
    const int ct = rest % n_ct;
    const int split = rest / n_ct;
    const int krow = __builtin_amdgcn_readfirstlane(in_idx / n_cin);       // kernel row: oy = krow - 1
    const int cin_tile = in_idx % n_cin;
    const int co0 = ct * BMO, ci0 = cin_tile * BNC;
    const int oy = krow - 1;
    int wi_tap[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wi_tap[t] = (p.taps[krow * 3 + t] >> 16) & 0xffff;

    // positions: p.M of them, P = 1 << p.lp per image row
    const int lp = p.lp, P = 1 << lp, W = p.Wi, H = p.Hi;
    const unsigned r_begin = (unsigned)((long long)split * p.rows_per_split);
    unsigned r_end = r_begin + (unsigned)p.rows_per_split;
    if (r_end > (unsigned)p.M) r_end = (unsigned)p.M;
    const int nk = r_begin < r_end ? (int)((r_end - r_begin + BKP - 1) / BKP) : 0;
    if (nk == 0) return;

    const int a_cv = tid % AV, a_row = tid / AV;
    const int b_cv = tid % BV, b_row = tid / BV;
    const bool a_col_ok = co0 + a_cv * 8 < p.N;
    const bool b_col_ok = ci0 + b_cv * 8 < p.Ci;

    // Whole tensors behind the descriptors: a slice is a whole number of K steps, positions past the last image are
    // past the tensors' ends and read zeros.
    // ... and the descriptors start at the slice's first image row (the gathered operand's: one row earlier, for oy = -1),
    // so the 32-bit offsets below are relative to the slice whatever the tensors' sizes
    const unsigned row0 = r_begin >> lp;                   // first image row (b * H + y) of the slice
    const unsigned xrow0 = row0 > 0u ? row0 - 1u : 0u;
    const __amdgpu_buffer_rsrc_t dy_rsrc = slice_rsrc(p.dy, (unsigned long long)row0 * (unsigned)W * (unsigned)p.ldy * 2ull, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t in_rsrc =
        slice_rsrc(p.in, (unsigned long long)xrow0 * (unsigned)W * (unsigned)p.pix_stride * 2ull, p.in_bytes);
    bf16x8 a_reg0[A_IT], b_reg0[B_IT], a_reg1[A_IT], b_reg1[B_IT];
    unsigned a_off[A_IT], b_off[B_IT];
    int b_y[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int pos = a_row + j * A_RPP, q = pos & (P - 1);
        const unsigned rr = (unsigned)(pos >> lp);         // relative to row0
        const bool ok = a_col_ok && q >= 1 && q <= W;
        a_off[j] = ok ? ((rr * (unsigned)W + (unsigned)(q - 1)) * (unsigned)p.ldy + co0 + a_cv * 8) * 2u : PDT_OOB_OFFSET;
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int pos = b_row + j * B_RPP, q = pos & (P - 1);
        const unsigned rr = row0 + (unsigned)(pos >> lp);
        b_ok[j] = b_col_ok && q >= 1 && q <= W;
        b_y[j] = (int)(rr % (unsigned)H);
        // (rr - xrow0 + oy) * W + q - 1 may be "negative" for the tensor's very first row: refused by the row test below
        b_off[j] = (((rr - xrow0 + (unsigned)oy) * (unsigned)W + (unsigned)(q - 1)) * (unsigned)p.pix_stride + ci0 + b_cv * 8) * 2u;
    }
    const int rows_step = BKP >> lp;                       // image rows per K step
    const unsigned a_step = (unsigned)(rows_step * W * p.ldy * 2);
    const unsigned b_step = (unsigned)(rows_step * W * p.pix_stride * 2);
    const int y_step = rows_step % H;

    auto load_tiles = [&](bf16x8 (&a_reg)[A_IT], bf16x8 (&b_reg)[B_IT]) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            a_reg[j] = buf_ld_bf16x8(dy_rsrc, a_off[j]);
            a_off[j] += a_off[j] == PDT_OOB_OFFSET ? 0u : a_step;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const bool ok = b_ok[j] && (unsigned)(b_y[j] + oy) < (unsigned)H;
            b_reg[j] = buf_ld_bf16x8(in_rsrc, ok ? b_off[j] : PDT_OOB_OFFSET);
            b_off[j] += b_step;
            b_y[j] += y_step;
            b_y[j] -= b_y[j] >= H ? H : 0;
        }
    };
    auto store_tiles = [&](int buf, const bf16x8 (&a_reg)[A_IT], const bf16x8 (&b_reg)[B_IT]) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* Bs = As + A_BYTES + RSB;            // row 0 of the X stage is the halo above the tile
#pragma unroll
        for (int j = 0; j < A_IT; ++j)
            *reinterpret_cast<bf16x8*>(As + (a_row + j * A_RPP) * RSA + a_cv * 16) = a_reg[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            *reinterpret_cast<bf16x8*>(Bs + (b_row + j * B_RPP) * RSB + b_cv * 16) = b_reg[j];
    };

    // the halo rows of both stages stay zero for the whole kernel
    for (int i = tid; i < 4 * (RSB / 4); i += 256) {
        const int which = i / (RSB / 4), w = i % (RSB / 4);
        unsigned char* Bs = smem + (which >> 1) * (A_BYTES + B_BYTES) + A_BYTES + ((which & 1) ? (BKP + 1) * RSB : 0);
        reinterpret_cast<unsigned*>(Bs)[w] = 0u;
    }

    f32x4 acc[3][MI][NI];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int trow0 = 4 * g + q4;
    const int tcol = 4 * pp;

    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* Bs = As + A_BYTES + RSB;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8 af[MI];
            const int r0 = kk * 32 + trow0, r1 = r0 + 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int col = wm * WTM + i * 16 + tcol;
                af[i] = tr_read_pair(As + r0 * RSA + col * 2, As + r1 * RSA + col * 2);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bf16x8 bfr[NI];
                const unsigned char* Bt = Bs + (t - 1) * RSB;          // ox = t - 1: the tile one row up / down
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn * WTN + j * 16 + tcol;
                    bfr[j] = tr_read_pair(Bt + r0 * RSB + col * 2, Bt + r1 * RSB + col * 2);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[t][i][j], 0, 0, 0);
            }
        }
    };

    load_tiles(a_reg0, b_reg0);                            // step 0
    store_tiles(0, a_reg0, b_reg0);
    if (nk > 1) load_tiles(a_reg0, b_reg0);                // step 1, in flight
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        if (ks + 2 < nk) load_tiles(a_reg1, b_reg1);
        compute(0);
        if (ks + 1 >= nk) break;
        store_tiles(1, a_reg0, b_reg0);
        __syncthreads();
        if (ks + 3 < nk) load_tiles(a_reg0, b_reg0);
        compute(1);
        if (ks + 2 < nk) store_tiles(0, a_reg1, b_reg1);
        __syncthreads();
    }

    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const long long wofs = (long long)wi_tap[t] * p.Ci;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int c = ci0 + wn * WTN + j * 16 + frow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int n = co0 + wm * WTM + i * 16 + fq * 4 + r;
                    if (n < p.N && c < p.Ci) {
                        const float v = p.row_scale ? acc[t][i][j][r] * p.row_scale[n] : acc[t][i][j][r];
                        if (p.ws) p.ws[(long long)split * p.slab + (long long)n * p.wrow + wofs + c] = v;
                        else atomicAdd(p.dw + (long long)n * p.wrow + wofs + c, v);
                    }
                }
            }
    }
}

// dw[n][widx(t)*Ci + c] += sum over slices (in slice order) of the slab entries
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const WgradParams p) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    long long per_n = (long long)p.T * p.Ci;
    if (i >= (long long)p.N * per_n) return;
    int n = (int)(i / per_n);
    int rem = (int)(i - (long long)n * per_n);
    int t = rem / p.Ci, c = rem - t * p.Ci;
    int oy, ox, widx;
    unpack_tap(p.taps[t], oy, ox, widx);
    long long off = (long long)n * p.wrow + (long long)widx * p.Ci + c;
    float s = p.dw[off];
    for (int k = 0; k < p.splitk; ++k) s += p.ws[(long long)k * p.slab + off];
    p.dw[off] = s;
}

static int launch_fold(const WgradParams& p, hipStream_t s) {
    if (!p.ws) return pdt_launch_status();
    long long total = (long long)p.N * p.T * p.Ci;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    return pdt_launch_status();
}

// Rows a slice may have so that what it reaches of either operand stays inside one descriptor (the kernels address a
// slice with 32-bit offsets from its first row / image): 1.5 GiB over the larger per-row footprint, less two images of
// slack for the gathered operand's image granularity.  < 2 K steps: the geometry cannot be served (-7).
static long long slice_rows_limit(const WgradParams& p, long long rows_per_image) {
    const long long dy_row = (long long)p.ldy * 2;
    const long long in_row = ((long long)p.Hi * p.Wi * p.pix_stride * 2 + rows_per_image - 1) / rows_per_image;
    return 0x60000000LL / (dy_row > in_row ? dy_row : in_row) - 2 * rows_per_image;
}

// split / rows_per_split from a wanted split: whole K steps per slice, every slice within reach
static int size_slices(WgradParams& p, long long split, long long rows_per_image, long long ws_slabs) {
    const long long lim = slice_rows_limit(p, rows_per_image);
    if (lim < 2 * BKP) return -7;
    if (split < 1) split = 1;
    long long rps = (p.M + split - 1) / split;
    if (rps > lim) rps = lim;
    rps = (rps + BKP - 1) / BKP * BKP;
    split = (p.M + rps - 1) / rps;
    if (p.ws && split > ws_slabs) return -8;
    p.splitk = (int)split;
    p.rows_per_split = rps;
    return 0;
}

template <int BMO, int BNC, int WM, int WN, int TAPS>
static int launch_wgrad_taps(WgradParams& p, hipStream_t s, int target_blocks, long long ws_slabs) {
    if (p.T % TAPS != 0 || p.Ci > BNC) return -6;
    int n_ct = (p.N + BMO - 1) / BMO * (p.T / TAPS);      // output tiles x tap groups
    long long max_split = (p.M + 2 * BKP - 1) / (2 * BKP);
    long long split = (target_blocks + n_ct - 1) / n_ct;
    if (split > max_split) split = max_split;
    if (p.ws && split > ws_slabs) split = ws_slabs;
    if (int rc = size_slices(p, split, (long long)p.Hg * p.Wg, ws_slabs)) return rc;
    long long nwg = (long long)n_ct * p.splitk;
    if (nwg > 0x7fffffffLL) return -2;
    hipLaunchKernelGGL((wgrad_taps_kernel<BMO, BNC, WM, WN, TAPS>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return launch_fold(p, s);
}

template <int BMO, int BNC, int WM, int WN>
static int launch_wgrad(WgradParams& p, hipStream_t s, int target_blocks, long long atomic_budget_bytes,
                        long long ws_slabs) {
    int n_ct = (p.N + BMO - 1) / BMO;
    int n_cin = (p.Ci + BNC - 1) / BNC;
    long long tiles = (long long)n_ct * n_cin * p.T;
    long long max_split = (p.M + 2 * BKP - 1) / (2 * BKP);     // at least two K steps per slice
    long long split = (target_blocks + tiles - 1) / tiles;
    if (split > max_split) split = max_split;
    // Every slice adds the whole fp32 output once with float atomics, which run at ~1.3 TB/s
    // chip-wide: keep split * output bytes within a budget so the atomics stay a small part of
    // the kernel (large-output layers have a short reduction anyway).
    long long out_bytes = (long long)p.N * p.T * p.Ci * 4;
    long long budget_split = atomic_budget_bytes / (out_bytes > 0 ? out_bytes : 1);
    if (budget_split < 1) budget_split = 1;
    if (split > budget_split) split = budget_split;
    if (p.ws && split > ws_slabs) split = ws_slabs;
    if (int rc = size_slices(p, split, (long long)p.Hg * p.Wg, ws_slabs)) return rc;
    long long nwg = tiles * p.splitk;
    if (nwg > 0x7fffffffLL) return -2;
    // 1x1 / stride 1 / no padding over a dense input: GEMM row m is input pixel m
    const bool linear = p.T == 1 && (p.taps[0] & 0xffff) == 0 && p.IS == 1 && p.Hg == p.Hi && p.Wg == p.Wi;
    if (BMO == 128 && BNC == 128 && p.use_dma) {
        if (linear) hipLaunchKernelGGL((wgrad_dma_kernel<true>), dim3((unsigned)nwg), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wgrad_dma_kernel<false>), dim3((unsigned)nwg), dim3(256), 0, s, p);
        return launch_fold(p, s);
    }
    if (linear) hipLaunchKernelGGL((wgrad_kernel<BMO, BNC, WM, WN, true>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<BMO, BNC, WM, WN, false>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return launch_fold(p, s);
}

// 3x3, stride 1, padding 1 on a dense grid: true iff the tap list is that filter; on success the taps are reordered to
// [kernel row][ox = -1, 0, +1] and lp / M / in_bytes describe the padded position space of wgrad_row3_kernel.
static bool row3_applies(const WgradParams& p) {
    if (p.T != 9 || p.IS != 1 || p.Hg != p.Hi || p.Wg != p.Wi || p.Wi + 1 > BKP) return false;
    int seen = 0;
    for (int i = 0; i < 9; ++i) {
        int oy, ox, w;
        unpack_tap(p.taps[i], oy, ox, w);
        if (oy < -1 || oy > 1 || ox < -1 || ox > 1) return false;
        seen |= 1 << ((oy + 1) * 3 + ox + 1);
    }
    return seen == 0x1ff;
}

template <int BMO, int BNC, int WM, int WN>
static int launch_wgrad_row3(WgradParams& p, hipStream_t s, int target_blocks, long long atomic_budget_bytes,
                             long long ws_slabs) {
    if (!row3_applies(p)) return -6;
    int ordered[9];
    for (int i = 0; i < 9; ++i) {
        int oy, ox, w;
        unpack_tap(p.taps[i], oy, ox, w);
        ordered[(oy + 1) * 3 + ox + 1] = p.taps[i];
    }
    for (int i = 0; i < 9; ++i) p.taps[i] = ordered[i];
    int lp = 3;
    while ((1 << lp) < p.Wi + 1) ++lp;
    p.lp = lp;
    const long long image_rows = p.M / p.Wg;               // B * H
    const long long positions = image_rows << lp;
    if (positions >= 0x7fffffffLL) return -2;
    p.M = positions;
    int n_ct = (p.N + BMO - 1) / BMO;
    int n_cin = (p.Ci + BNC - 1) / BNC;
    long long tiles = (long long)n_ct * n_cin * 3;
    long long max_split = (p.M + 2 * BKP - 1) / (2 * BKP);
    long long split = (target_blocks + tiles - 1) / tiles;
    if (split > max_split) split = max_split;
    long long out_bytes = (long long)p.N * p.T * p.Ci * 4;
    long long budget_split = atomic_budget_bytes / (out_bytes > 0 ? out_bytes : 1);
    if (budget_split < 1) budget_split = 1;
    if (split > budget_split) split = budget_split;
    if (p.ws && split > ws_slabs) split = ws_slabs;
    // (a "row" of this kernel is a position: P per image row of W pixels)
    if (int rc = size_slices(p, split, (long long)p.Hg << lp, ws_slabs)) return rc;
    long long nwg = tiles * p.splitk;
    if (nwg > 0x7fffffffLL) return -2;
    hipLaunchKernelGGL((wgrad_row3_kernel<BMO, BNC, WM, WN>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    return launch_fold(p, s);
}

// tile: 0 = auto, 1 = 128x128, 2 = 64x64, 3 = 64x32 (one tap per workgroup), 4 = 128x64,
//       5 = 64x32 with all 7 taps per workgroup (the stem; auto picks it for T == 7, Ci <= 32),
//       6 / 7 = 64x64 / 128x64 with three taps (one kernel row) per workgroup, Ci <= 64; 9 = 64x128;
//       10 / 11 = 128x64 / 64x64 with one kernel row per workgroup and the gathered operand staged once (3x3, stride 1,
//       padding 1 only; auto picks them there)
// ws / ws_bytes: workspace of the deterministic variant (NULL: float atomics).  It must stay untouched by other
// streams until this launch has finished: one workspace per stream that issues weight gradients.
// row_scale: optional [N] factor applied to output row n before it is added to dw
PDT_API int pdt_conv_wgrad_ex(const void* dy, const void* in, float* dw, long long M, int Hi, int Wi, int pix_stride,
                              int Ci, int Hg, int Wg, int IS, int N, int ldy, int wrow, int T, const signed char* offy,
                              const signed char* offx, const unsigned short* widx, int tile, int target_blocks,
                              int atomic_budget_mb, float* ws, long long ws_bytes, const float* row_scale,
                              hipStream_t s) {
    if (T <= 0 || T > MAX_TAPS) return -3;
    if (M <= 0) return 0;
    if (M >= 0x7fffffffLL) return -2;
    if (ldy % 8 != 0 || Ci % 8 != 0 || pix_stride % 4 != 0) return -4;
    WgradParams p;
    p.dy = (const bf16*)dy; p.in = (const bf16*)in; p.dw = dw;
    p.ws = ws;
    p.row_scale = row_scale;
    p.slab = (long long)N * wrow;
    long long ws_slabs = ws ? ws_bytes / (p.slab * 4) : 0;
    if (ws && ws_slabs < 1) return -8;
    p.M = M; p.Hi = Hi; p.Wi = Wi; p.pix_stride = pix_stride; p.Ci = Ci;
    {
        long long images = M / ((long long)Hg * Wg);
        long long in_b = images * Hi * Wi * pix_stride * 2, dy_b = M * (long long)ldy * 2;
        p.in_bytes = (unsigned long long)in_b;
        p.dy_bytes = (unsigned long long)dy_b;
    }
    p.Hg = Hg; p.Wg = Wg; p.IS = IS; p.N = N; p.ldy = ldy; p.wrow = wrow; p.T = T;
    {
        int hw = Hg * Wg;
        p.step_b = BKP / hw;
        p.step_y = (BKP % hw) / Wg;
        p.step_x = (BKP % hw) % Wg;
    }
    for (int i = 0; i < T; ++i)
        p.taps[i] = ((int)offy[i] & 0xff) | (((int)offx[i] & 0xff) << 8) | ((int)widx[i] << 16);
    for (int i = T; i < MAX_TAPS; ++i) p.taps[i] = 0;
    // tile 8 = tile 1 with LDS-DMA operand staging (the automatic choice where tile 1 would be picked and rows are
    // 16-byte aligned); a forced tile 1 keeps the register-staged kernel (A/B measurements)
    p.use_dma = 0;
    p.lp = 0;
    const bool dma_ok = ldy % 8 == 0 && pix_stride % 8 == 0 && Ci % 8 == 0;
    if (tile == 8) {
        if (!dma_ok) return -4;
        p.use_dma = 1;
        tile = 1;
    }
    const bool user_blocks = target_blocks > 0;
    if (!user_blocks) target_blocks = g_pdt_wgrad_blocks;
    // Atomic-traffic budget of the split: 24 MB up to 512 images per launch, 32 MB from 1024 on (the operands grow with
    // the batch, the output does not; measured inside the step: 18 / 24 / 32 MB give 17.27 / 17.12 / 17.30 ms at batch
    // 256, 24 / 32 / 48 MB 31.35 / 31.46 / 31.61 at 512 and 59.67 / 59.35 / 59.73 at 1024).
    const long long images = M / ((long long)Hg * Wg);
    const int auto_mb = images >= 1024 ? 32 : 24;
    long long budget = (long long)(atomic_budget_mb > 0 ? atomic_budget_mb
                                   : (g_pdt_wgrad_atomic_mb > 0 ? g_pdt_wgrad_atomic_mb : auto_mb)) << 20;
    if (tile == 0 && g_pdt_wgrad_row3 && Ci >= 64 && Wi >= g_pdt_wgrad_row3_min_w && row3_applies(p)) tile = N <= 64 ? 11 : 10;
    if (tile == 0) {
        if (Ci <= 32) tile = (T == 7) ? 5 : 3;
        else if (T == 9 && Ci <= 64) tile = (N <= 64) ? 6 : 7;    // 3x3 on 64 channels: one kernel row per workgroup
        else if (N <= 64 || Ci <= 64) {
            tile = 2;
            if (g_pdt_wgrad_wide_small && N <= 64 && Ci >= 128) tile = 9;
            if (g_pdt_wgrad_wide_small && Ci <= 64 && N >= 128) tile = 4;
        }
        else {
            tile = g_pdt_wgrad_big_tile;
            p.use_dma = tile == 1 && dma_ok && g_pdt_wgrad_dma;
        }
    }
    switch (tile) {
        case 1: return launch_wgrad<128, 128, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 2: return launch_wgrad<64, 64, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 3: return launch_wgrad<64, 32, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 4: return launch_wgrad<128, 64, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 5: return launch_wgrad_taps<64, 32, 2, 2, 7>(p, s, user_blocks ? target_blocks : 768, ws_slabs);
        case 6: return launch_wgrad_taps<64, 64, 2, 2, 3>(p, s, target_blocks, ws_slabs);
        case 7: return launch_wgrad_taps<128, 64, 2, 2, 3>(p, s, target_blocks, ws_slabs);
        case 9: return launch_wgrad<64, 128, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 10: return launch_wgrad_row3<128, 64, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        case 11: return launch_wgrad_row3<64, 64, 2, 2>(p, s, target_blocks, budget, ws_slabs);
        default: return -6;
    }
}

PDT_API int pdt_conv_wgrad(const void* dy, const void* in, float* dw, long long M, int Hi, int Wi, int pix_stride,
                           int Ci, int Hg, int Wg, int IS, int N, int ldy, int wrow, int T, const signed char* offy,
                           const signed char* offx, const unsigned short* widx, int tile, int target_blocks,
                           int atomic_budget_mb, float* ws, long long ws_bytes, hipStream_t s) {
    return pdt_conv_wgrad_ex(dy, in, dw, M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, N, ldy, wrow, T, offy, offx, widx, tile,
                             target_blocks, atomic_budget_mb, ws, ws_bytes, nullptr, s);
}
