// Does the epilogue's access pattern cost bandwidth?  A wave of the convolution epilogue owns 16 rows x 64 channels of a
// bf16 tile; lane (fq = lane >> 4, frow = lane & 15) holds 16 consecutive channels (32 bytes) of row frow and moves
// them with two 16-byte accesses, so one instruction touches 4 x 16 bytes at a stride of 32 per row.  "split" is that
// pattern, "dense" the alternative in which one instruction covers 64 contiguous bytes of a row.
//   hipcc --offload-arch=gfx950 -O3 bench/diag/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: split (lane's 32 bytes by two instructions), 1: dense (64 contiguous bytes per row and instruction),
// 2 / 3: the same two with ordinary (cached) stores instead of non-temporal ones, 4: full 128-byte lines per instruction
// (lane l: row l / 8, chunk l % 8 -- needs the tile transposed through LDS), non-temporal, 5: the same, cached
template <int MODE, bool LOAD>
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ p, const u32x4* __restrict__ q, long long rows, int row_vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    long long stride = (long long)gridDim.x * 4 * 16;
    for (long long r0 = ((long long)blockIdx.x * 4 + wave) * 16; r0 < rows; r0 += stride) {
        // the wave covers rows r0..r0+15, and walks the row's 128-byte segments (64 channels each)
        for (int seg = 0; seg < row_vec / 8; ++seg) {
            long long base = (r0 + frow) * row_vec + seg * 8;          // in 16-byte vectors
            int o0 = (MODE == 0 || MODE == 2) ? fq * 2 : fq, o1 = (MODE == 0 || MODE == 2) ? fq * 2 + 1 : 4 + fq;
            u32x4 a = {1u, 2u, 3u, 4u}, b = {5u, 6u, 7u, 8u};
            if (MODE >= 4) {                   // rows r0 + lane/8 and r0 + 8 + lane/8, chunk lane % 8
                base = (r0 + (lane >> 3)) * row_vec + seg * 8;
                o0 = lane & 7;
                o1 = 8 * row_vec + (lane & 7);
            }
            if (LOAD) {
                a = __builtin_nontemporal_load(q + base + o0);
                b = __builtin_nontemporal_load(q + base + o1);
                a += b;
                if (a[0] == 0x12345678u) p[base + o0] = a;            // never true: keeps the loads alive
            } else {
                if (MODE == 2 || MODE == 3 || MODE == 5) {
                    p[base + o0] = a;
                    p[base + o1] = b;
                } else {
                    __builtin_nontemporal_store(a, p + base + o0);
                    __builtin_nontemporal_store(b, p + base + o1);
                }
            }
        }
    }
}

template <int MODE, bool LOAD>
static float run(u32x4* p, const u32x4* q, long long rows, int row_vec) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, LOAD>), dim3(2048), dim3(256), 0, 0, p, q, rows, row_vec);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, LOAD>), dim3(2048), dim3(256), 0, 0, p, q, rows, row_vec);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main() {
    const long long rows = 3211264;            // 1024 x 56 x 56
    for (int C : {64, 256}) {
        const int row_vec = C * 2 / 16;
        size_t bytes = (size_t)rows * C * 2;
        u32x4 *p, *q;
        hipMalloc(&p, bytes); hipMalloc(&q, bytes);
        hipMemset(q, 1, bytes);
        float t;
        t = run<0, false>(p, q, rows, row_vec); printf("C %3d store split  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<1, false>(p, q, rows, row_vec); printf("C %3d store dense  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<2, false>(p, q, rows, row_vec); printf("C %3d store split, cached  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<3, false>(p, q, rows, row_vec); printf("C %3d store dense, cached  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<4, false>(p, q, rows, row_vec); printf("C %3d store lines          %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<5, false>(p, q, rows, row_vec); printf("C %3d store lines, cached  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<0, true>(p, q, rows, row_vec);  printf("C %3d load  split  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        t = run<1, true>(p, q, rows, row_vec);  printf("C %3d load  dense  %7.1f us %5.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        hipFree(p); hipFree(q);
    }
    return 0;
}
