// Practical HBM ceiling for the STREAM MIX of the fused backward GEMM pass of the north-star layer (dsw_wgrad_x3.hip,
// FUSE): per row of N = B*V it reads three 128-byte basis rows (T0..T2) and one 256-byte dY row and writes three
// 128-byte dgrad-plane rows.  This program moves exactly those bytes with no arithmetic to speak of, in the two
// traversal orders a kernel can choose (each workgroup a contiguous slab of rows / tiles interleaved across the grid),
// so that the kernel's achieved bandwidth can be set against what the memory system gives THIS access pattern rather
// than against a two-stream copy.   hipcc --offload-arch=gfx950 -O3 tools/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool SLAB, bool NT, bool ST4 = false>
__global__ __launch_bounds__(256) void mix_kernel(const f4* __restrict__ T0, const f4* __restrict__ T1,
                                                  const f4* __restrict__ T2, const f4* __restrict__ dY,
                                                  f4* G0, f4* G1, f4* G2, long n_rows, int tile_rows) {
    // a tile = tile_rows rows; 8 lanes per 128-byte row, 16 lanes per dY row
    const long tiles = n_rows / tile_rows;
    const long per_wg = (tiles + gridDim.x - 1) / gridDim.x;
    const long t_begin = SLAB ? (long)blockIdx.x * per_wg : blockIdx.x;
    const long t_end = SLAB ? (t_begin + per_wg < tiles ? t_begin + per_wg : tiles) : tiles;
    const long t_step = SLAB ? 1 : gridDim.x;
    for (long t = t_begin; t < t_end; t += t_step) {
        const long r0 = t * tile_rows;
        for (int e = threadIdx.x; e < tile_rows * 8; e += 256) {
            const long i = r0 * 8 + e;
            f4 a, b, c, d0, d1;
            if (NT) {
                a = __builtin_nontemporal_load(&T0[i]); b = __builtin_nontemporal_load(&T1[i]);
                c = __builtin_nontemporal_load(&T2[i]);
                d0 = __builtin_nontemporal_load(&dY[2 * i]); d1 = __builtin_nontemporal_load(&dY[2 * i + 1]);
            } else {
                a = T0[i]; b = T1[i]; c = T2[i]; d0 = dY[2 * i]; d1 = dY[2 * i + 1];
            }
            f4 g0 = a + d0;
            f4 g1 = b + d1;
            f4 g2 = c + d0 - d1;
            if (ST4) {
                // the store shape of an MFMA 32x32 accumulator written straight to memory: 4 bytes per lane, one
                // instruction = two 128-byte row segments (rows 4 apart), four instructions per 8 rows of a plane
                const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
                const long rbase = r0 + (e - threadIdx.x) / 8 + 8 * wave + 4 * half;
                float* p0 = reinterpret_cast<float*>(G0); float* p1 = reinterpret_cast<float*>(G1);
                float* p2 = reinterpret_cast<float*>(G2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long o = (rbase + q) * 32 + l31;
                    if (NT) {
                        __builtin_nontemporal_store(g0[q], &p0[o]); __builtin_nontemporal_store(g1[q], &p1[o]);
                        __builtin_nontemporal_store(g2[q], &p2[o]);
                    } else {
                        p0[o] = g0[q]; p1[o] = g1[q]; p2[o] = g2[q];
                    }
                }
            } else if (NT) {
                __builtin_nontemporal_store(g0, &G0[i]); __builtin_nontemporal_store(g1, &G1[i]);
                __builtin_nontemporal_store(g2, &G2[i]);
            } else {
                G0[i] = g0; G1[i] = g1; G2[i] = g2;
            }
        }
    }
}

template <bool SLAB, bool NT, bool ST4 = false>
static float run(const f4* T0, const f4* T1, const f4* T2, const f4* dY, f4* G0, f4* G1, f4* G2,
                 long n_rows, int tile_rows, int grid, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_kernel<SLAB, NT, ST4>), dim3(grid), dim3(256), 0, 0, T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((mix_kernel<SLAB, NT, ST4>), dim3(grid), dim3(256), 0, 0, T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main() {
    const long n_rows = 16L * 49152;            // the north-star batch: B = 16 spheres of nside 64
    const size_t row32 = 32 * sizeof(float);
    f4 *T0, *T1, *T2, *dY, *G0, *G1, *G2;
    CK(hipMalloc(&T0, n_rows * row32)); CK(hipMalloc(&T1, n_rows * row32)); CK(hipMalloc(&T2, n_rows * row32));
    CK(hipMalloc(&dY, n_rows * row32 * 2));
    CK(hipMalloc(&G0, n_rows * row32)); CK(hipMalloc(&G1, n_rows * row32)); CK(hipMalloc(&G2, n_rows * row32));
    CK(hipMemset(T0, 0, n_rows * row32)); CK(hipMemset(T1, 0, n_rows * row32)); CK(hipMemset(T2, 0, n_rows * row32));
    CK(hipMemset(dY, 0, n_rows * row32 * 2));
    const double bytes = (double)n_rows * row32 * 8;   // 3 + 2 read, 3 written
    printf("stream mix of the fused backward GEMM pass: %.1f MB per launch (5 parts read, 3 written)\n", bytes / 1e6);
    printf("%-10s %-4s %6s %6s %9s %9s\n", "order", "nt", "tile", "grid", "us", "TB/s");
    for (int tile_rows : {128}) {
        for (int grid : {512, 1024, 6144}) {
            const float s0 = run<true, false>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float s1 = run<true, true>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float i0 = run<false, false>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float i1 = run<false, true>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float q0 = run<true, false, true>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float q1 = run<false, false, true>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            const float q2 = run<false, true, true>(T0, T1, T2, dY, G0, G1, G2, n_rows, tile_rows, grid, 20);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f   (4-byte stores)\n", "slab", "no", tile_rows, grid, q0, bytes / q0 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f   (4-byte stores)\n", "interleave", "no", tile_rows, grid, q1, bytes / q1 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f   (4-byte stores)\n", "interleave", "yes", tile_rows, grid, q2, bytes / q2 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f\n", "slab", "no", tile_rows, grid, s0, bytes / s0 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f\n", "slab", "yes", tile_rows, grid, s1, bytes / s1 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f\n", "interleave", "no", tile_rows, grid, i0, bytes / i0 / 1e6);
            printf("%-10s %-4s %6d %6d %9.1f %9.2f\n", "interleave", "yes", tile_rows, grid, i1, bytes / i1 / 1e6);
        }
    }
    return 0;
}
