// Channel-mix contractions of the Chebyshev convolution on the matrix cores (MFMA), gfx950.
//
//   forward   Y[n, o]      = bias[o] + sum_{k,f} T_k[n, f] * W[f, k, o]          (layers.py:171-178)
//   dgrad     G_k[n, f]    = sum_o dY[n, o] * W[f, k, o]                         (autograd of :177)
//   wgrad     dW[f, k, o]  = sum_n T_k[n, f] * dY[n, o],  db[o] = sum_n dY[n, o]
//
// n runs over the N = B*V node rows (hundreds of thousands), the other two extents are channel
// counts (2..1536), i.e. every contraction is "tall and skinny".  The T_k are K separate
// [N, Fin] planes (T_0 is the caller's x itself) so the reference's [B*V, Fin*K] operand
// (layers.py:171-173: view/permute/contiguous = one full extra copy) is never materialised.
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TFLOP/s peak on MI355X).
// The summation order over the reduction index differs from a k-ascending loop (operands are
// consumed in 8-wide groups, lane-half h taking elements 4h..4h+3) - irrelevant at fp32 tolerance.
// bf16 storage is supported by widening to fp32 while staging through LDS (fp32 accumulate).
#include "dsw_common.h"

namespace {

constexpr int BM = 128;       // rows of the tall operand per workgroup (4 waves x 32)
constexpr int BN = 64;        // output columns per workgroup (2 MFMA tiles per wave)
constexpr int BK = 32;        // reduction chunk staged in LDS
constexpr int LDA = BK + 4;   // +4 floats: conflict-free ds_read_b128 of 16 rows (stride 36 words)
constexpr int LDB = BN;

template <bool BF16>
static __device__ __forceinline__ float ld1(const void* p, size_t i) {
    if constexpr (BF16) return bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
    else return static_cast<const float*>(p)[i];
}
template <bool BF16>
static __device__ __forceinline__ void st1(void* p, size_t i, float v) {
    if constexpr (BF16) static_cast<uint16_t*>(p)[i] = f32_to_bf16(v);
    else static_cast<float*>(p)[i] = v;
}
// 4 consecutive elements; `vec` promises 4-element alignment and in-bounds
template <bool BF16>
static __device__ __forceinline__ float4 ld4(const void* p, size_t i) {
    if constexpr (BF16) {
        const uint2 t = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p) + i);
        return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
                           __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
    }
}

struct TsGemmParams {
    // tall operand A: n_planes_a planes of [M, lda]; plane 0 = A0, plane p>0 = A1 + (p-1)*a_plane_stride
    const void* A0;
    const void* A1;
    size_t a_plane_stride;  // elements
    int lda;
    int n_planes_a;
    int kd_per_plane;       // reduction extent inside one plane
    // small operand: element (a-plane p, c-plane q, kd, n) at Bsrc[p*b_sp + q*b_sq + kd*b_skd + n*b_sn]
    const void* Bsrc;
    long b_sp, b_sq, b_skd, b_sn;
    // output: n_planes_c planes of [M, ldc]; plane 0 = C0, plane q>0 = C1 + (q-1)*c_plane_stride
    void* C0;
    void* C1;
    size_t c_plane_stride;
    int ldc;
    int n_planes_c;
    int n_per_plane;        // valid columns per output plane
    const void* bias;       // [n_per_plane] or null (same dtype as the data)
    long M;
    int a_vec;              // 1 if float4/bf16x4 loads of A are legal
    int b_vec;              // 1 if b_sn == 1 and 4-wide loads of B are legal
};

// C[q] (M x n_per_plane) = sum_p A[p] (M x kd) * B[p,q] (kd x n_per_plane) (+ bias)
template <bool BF16>
__global__ __launch_bounds__(256) void ts_gemm_kernel(const TsGemmParams P) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const long row0 = (long)blockIdx.x * BM;
    const int ntile_per_plane = (P.n_per_plane + BN - 1) / BN;
    const int q = blockIdx.y / ntile_per_plane;              // output plane
    const int col0 = (blockIdx.y - q * ntile_per_plane) * BN;  // first column in that plane

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    // A staging: thread -> rows ar + 32*i (i<4), 4 columns at ac4
    const int ar = tid >> 3, ac4 = (tid & 7) * 4;
    // B staging: thread -> reduction rows br + 16*i (i<2), 4 columns at bc4
    const int br = tid >> 4, bc4 = (tid & 15) * 4;

    const int chunks = (P.kd_per_plane + BK - 1) / BK;
    const int total = P.n_planes_a * chunks;

    float4 ra[4], rb[2];
    auto fetch = [&](int it) {
        const int p = it / chunks;
        const int k0 = (it - p * chunks) * BK;
        const void* A = (p == 0) ? P.A0 : P.A1;
        const size_t abase = (p == 0) ? 0 : (size_t)(p - 1) * P.a_plane_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = row0 + ar + 32 * i;
            const int kc = k0 + ac4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < P.M) {
                const size_t off = abase + (size_t)r * P.lda + kc;
                if (P.a_vec && kc + 3 < P.kd_per_plane) {
                    v = ld4<BF16>(A, off);
                } else {
                    if (kc + 0 < P.kd_per_plane) v.x = ld1<BF16>(A, off + 0);
                    if (kc + 1 < P.kd_per_plane) v.y = ld1<BF16>(A, off + 1);
                    if (kc + 2 < P.kd_per_plane) v.z = ld1<BF16>(A, off + 2);
                    if (kc + 3 < P.kd_per_plane) v.w = ld1<BF16>(A, off + 3);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kd = k0 + br + 16 * i;
            const int n = col0 + bc4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kd < P.kd_per_plane) {
                const long off = (long)p * P.b_sp + (long)q * P.b_sq + (long)kd * P.b_skd + (long)n * P.b_sn;
                if (P.b_vec && n + 3 < P.n_per_plane) {
                    v = ld4<BF16>(P.Bsrc, (size_t)off);
                } else {
                    if (n + 0 < P.n_per_plane) v.x = ld1<BF16>(P.Bsrc, (size_t)(off + 0 * P.b_sn));
                    if (n + 1 < P.n_per_plane) v.y = ld1<BF16>(P.Bsrc, (size_t)(off + 1 * P.b_sn));
                    if (n + 2 < P.n_per_plane) v.z = ld1<BF16>(P.Bsrc, (size_t)(off + 2 * P.b_sn));
                    if (n + 3 < P.n_per_plane) v.w = ld1<BF16>(P.Bsrc, (size_t)(off + 3 * P.b_sn));
                }
            }
            rb[i] = v;
        }
    };

    fetch(0);
    for (int it = 0; it < total; ++it) {
        __syncthreads();  // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[(ar + 32 * i) * LDA + ac4]) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float4*>(&Bs[(br + 16 * i) * LDB + bc4]) = rb[i];
        __syncthreads();
        if (it + 1 < total) fetch(it + 1);  // global loads in flight under the MFMAs

        const float* arow = &As[(wave * 32 + l31) * LDA + 4 * half];
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 8 * c);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kk = 8 * c + 4 * half + t;
                const float b0 = Bs[kk * LDB + l31];
                const float b1 = Bs[kk * LDB + 32 + l31];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b1, acc1, 0, 0, 0);
            }
        }
    }

    // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
    void* C = (q == 0) ? P.C0 : P.C1;
    const size_t cbase = (q == 0) ? 0 : (size_t)(q - 1) * P.c_plane_stride;
    const int cA = col0 + l31, cB = col0 + 32 + l31;
    const float biasA = (P.bias != nullptr && cA < P.n_per_plane) ? ld1<BF16>(P.bias, cA) : 0.f;
    const float biasB = (P.bias != nullptr && cB < P.n_per_plane) ? ld1<BF16>(P.bias, cB) : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long r = row0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
        if (r < P.M) {
            const size_t off = cbase + (size_t)r * P.ldc;
            if (cA < P.n_per_plane) st1<BF16>(C, off + cA, acc0[i] + biasA);
            if (cB < P.n_per_plane) st1<BF16>(C, off + cB, acc1[i] + biasB);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad: partial[s][(k*Fin + f)][o] = sum_{n in slab s} T_k[n, f] * dY[n, o]; row Kd = column sums
// ---------------------------------------------------------------------------------------------
constexpr int WR = 32;  // node rows per staged chunk

struct WgradParams {
    const void* X;          // T_0
    const void* T;          // T_1.. planes
    size_t plane_stride;    // elements
    const void* dY;
    float* partial;         // [S][Kd + 1][Fout]
    long N;
    int Fin, Fout, K;
    long rows_per_slab;
    int tiles_per_plane;    // ceil(Fin / 32)
    int t_vec, dy_vec;
};

template <bool BF16>
__global__ __launch_bounds__(256) void cheb_wgrad_kernel(const WgradParams P) {
    __shared__ __attribute__((aligned(16))) float Ts[4][WR * 32];
    __shared__ __attribute__((aligned(16))) float Ds[WR * BN];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.y * 4 + wave;
    const int ntiles = P.K * P.tiles_per_plane;
    const bool active = tile < ntiles;
    const int k = active ? tile / P.tiles_per_plane : 0;
    const int f0 = active ? (tile - k * P.tiles_per_plane) * 32 : 0;
    const int o0 = blockIdx.z * BN;
    const int Kd = P.K * P.Fin;
    const long n_begin = (long)blockIdx.x * P.rows_per_slab;
    const long n_end = (n_begin + P.rows_per_slab < P.N) ? n_begin + P.rows_per_slab : P.N;

    const void* A = (k == 0) ? P.X : P.T;
    const size_t abase = (k == 0) ? 0 : (size_t)(k - 1) * P.plane_stride;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float colsum = 0.f;

    const int tr = lane >> 3, tc4 = (lane & 7) * 4;   // T tile: rows tr + 8*i (i<4)
    const int dr = tid >> 4, dc4 = (tid & 15) * 4;    // dY tile: rows dr + 16*i (i<2)

    float4 rt[4], rd[2];
    auto fetch = [&](long n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = n0 + tr + 8 * i;
            const int f = f0 + tc4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active && r < n_end) {
                const size_t off = abase + (size_t)r * P.Fin + f;
                if (P.t_vec && f + 3 < P.Fin) {
                    v = ld4<BF16>(A, off);
                } else {
                    if (f + 0 < P.Fin) v.x = ld1<BF16>(A, off + 0);
                    if (f + 1 < P.Fin) v.y = ld1<BF16>(A, off + 1);
                    if (f + 2 < P.Fin) v.z = ld1<BF16>(A, off + 2);
                    if (f + 3 < P.Fin) v.w = ld1<BF16>(A, off + 3);
                }
            }
            rt[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long r = n0 + dr + 16 * i;
            const int o = o0 + dc4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < n_end) {
                const size_t off = (size_t)r * P.Fout + o;
                if (P.dy_vec && o + 3 < P.Fout) {
                    v = ld4<BF16>(P.dY, off);
                } else {
                    if (o + 0 < P.Fout) v.x = ld1<BF16>(P.dY, off + 0);
                    if (o + 1 < P.Fout) v.y = ld1<BF16>(P.dY, off + 1);
                    if (o + 2 < P.Fout) v.z = ld1<BF16>(P.dY, off + 2);
                    if (o + 3 < P.Fout) v.w = ld1<BF16>(P.dY, off + 3);
                }
            }
            rd[i] = v;
        }
    };

    if (n_begin < n_end) fetch(n_begin);
    for (long n0 = n_begin; n0 < n_end; n0 += WR) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&Ts[wave][(tr + 8 * i) * 32 + tc4]) = rt[i];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float4*>(&Ds[(dr + 16 * i) * BN + dc4]) = rd[i];
        __syncthreads();
        if (n0 + WR < n_end) fetch(n0 + WR);

        if (active) {
#pragma unroll
            for (int s = 0; s < WR / 2; ++s) {
                const int n = 2 * s + half;
                const float a = Ts[wave][n * 32 + l31];       // A^T[f][n]
                const float b0 = Ds[n * BN + l31];             // dY[n][o]
                const float b1 = Ds[n * BN + 32 + l31];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        }
        if (blockIdx.y == 0 && tid < BN) {
#pragma unroll
            for (int r = 0; r < WR; ++r) colsum += Ds[r * BN + tid];
        }
    }

    float* out = P.partial + (size_t)blockIdx.x * (size_t)(Kd + 1) * P.Fout;
    if (active) {
        const int oA = o0 + l31, oB = o0 + 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = f0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (f < P.Fin) {
                const size_t off = (size_t)(k * P.Fin + f) * P.Fout;
                if (oA < P.Fout) out[off + oA] = acc0[i];
                if (oB < P.Fout) out[off + oB] = acc1[i];
            }
        }
    }
    if (blockIdx.y == 0 && tid < BN && o0 + tid < P.Fout) out[(size_t)Kd * P.Fout + o0 + tid] = colsum;
}

// dW[f, k, o] = sum_s partial[s][k*Fin + f][o];  db[o] = sum_s partial[s][Kd][o]  (deterministic order)
template <bool BF16>
__global__ __launch_bounds__(256) void cheb_wgrad_reduce_kernel(const float* __restrict__ partial, int S,
                                                                int Fin, int Fout, int K, void* dW,
                                                                void* db) {
    const int Kd = K * Fin;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)(Kd + 1) * Fout;
    if (idx >= total) return;
    const size_t slab = (size_t)(Kd + 1) * Fout;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        s0 += partial[(size_t)(s + 0) * slab + idx];
        s1 += partial[(size_t)(s + 1) * slab + idx];
        s2 += partial[(size_t)(s + 2) * slab + idx];
        s3 += partial[(size_t)(s + 3) * slab + idx];
    }
    for (; s < S; ++s) s0 += partial[(size_t)s * slab + idx];
    const float v = (s0 + s1) + (s2 + s3);
    const int kd = (int)(idx / Fout), o = (int)(idx - (long)kd * Fout);
    if (kd == Kd) {
        if (db != nullptr) st1<BF16>(db, o, v);
    } else {
        const int k = kd / Fin, f = kd - k * Fin;
        st1<BF16>(dW, ((size_t)f * K + k) * Fout + o, v);
    }
}

}  // namespace

// ------------------------------- host-side launchers (internal) ------------------------------
int dsw_mix_fwd_launch(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                       int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P;
    P.A0 = X; P.A1 = T; P.a_plane_stride = (size_t)N * Fin; P.lda = (int)Fin;
    P.n_planes_a = (int)K; P.kd_per_plane = (int)Fin;
    P.Bsrc = W; P.b_sp = Fout; P.b_sq = 0; P.b_skd = K * Fout; P.b_sn = 1;
    P.C0 = Y; P.C1 = Y; P.c_plane_stride = 0; P.ldc = (int)Fout; P.n_planes_c = 1; P.n_per_plane = (int)Fout;
    P.bias = bias; P.M = N;
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0) && (K == 1 || ((uintptr_t)T & am) == 0);
    P.b_vec = (Fout % 4 == 0) && (((uintptr_t)W & am) == 0);
    dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)((Fout + BN - 1) / BN));
    if (dtype == DSW_F32) hipLaunchKernelGGL(ts_gemm_kernel<false>, grid, dim3(256), 0, stream, P);
    else if (dtype == DSW_BF16) hipLaunchKernelGGL(ts_gemm_kernel<true>, grid, dim3(256), 0, stream, P);
    else return DSW_ERR_BAD_DTYPE;
    return dsw_check_launch();
}

// G_0 -> dX buffer, G_1.. -> Gws planes
int dsw_mix_dgrad_launch(const void* dY, const void* W, void* G0, void* Grest, int64_t N, int64_t Fin,
                         int64_t Fout, int64_t K, int dtype, hipStream_t stream) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P;
    P.A0 = dY; P.A1 = dY; P.a_plane_stride = 0; P.lda = (int)Fout; P.n_planes_a = 1; P.kd_per_plane = (int)Fout;
    P.Bsrc = W; P.b_sp = 0; P.b_sq = Fout; P.b_skd = 1; P.b_sn = K * Fout;
    P.C0 = G0; P.C1 = Grest; P.c_plane_stride = (size_t)N * Fin; P.ldc = (int)Fin; P.n_planes_c = (int)K;
    P.n_per_plane = (int)Fin;
    P.bias = nullptr; P.M = N;
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0);
    P.b_vec = 0;
    const int ntile = (int)((Fin + BN - 1) / BN);
    dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)(K * ntile));
    if (dtype == DSW_F32) hipLaunchKernelGGL(ts_gemm_kernel<false>, grid, dim3(256), 0, stream, P);
    else if (dtype == DSW_BF16) hipLaunchKernelGGL(ts_gemm_kernel<true>, grid, dim3(256), 0, stream, P);
    else return DSW_ERR_BAD_DTYPE;
    return dsw_check_launch();
}

// number of row slabs used by wgrad for a given N (also sizes the partial workspace)
int64_t dsw_wgrad_slabs(int64_t N, int64_t* rows_per_slab) {
    const int64_t target = 1024;                       // ~4 workgroups per CU
    int64_t rps = (N + target - 1) / target;
    rps = ((rps + WR - 1) / WR) * WR;
    if (rps < 4 * WR) rps = 4 * WR;
    if (rows_per_slab) *rows_per_slab = rps;
    return N > 0 ? (N + rps - 1) / rps : 0;
}

int dsw_wgrad_launch(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                     int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream) {
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    int64_t rps = 0;
    const int64_t S = dsw_wgrad_slabs(N, &rps);
    if (S > 0) {
        WgradParams P;
        P.X = X; P.T = T; P.plane_stride = (size_t)N * Fin; P.dY = dY; P.partial = partial;
        P.N = N; P.Fin = (int)Fin; P.Fout = (int)Fout; P.K = (int)K; P.rows_per_slab = rps;
        P.tiles_per_plane = (int)((Fin + 31) / 32);
        const uintptr_t am = (uintptr_t)(4 * es) - 1;
        P.t_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0) && (K == 1 || ((uintptr_t)T & am) == 0);
        P.dy_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0);
        const int ntiles = (int)K * P.tiles_per_plane;
        dim3 grid((unsigned)S, (unsigned)((ntiles + 3) / 4), (unsigned)((Fout + BN - 1) / BN));
        if (dtype == DSW_F32) hipLaunchKernelGGL(cheb_wgrad_kernel<false>, grid, dim3(256), 0, stream, P);
        else hipLaunchKernelGGL(cheb_wgrad_kernel<true>, grid, dim3(256), 0, stream, P);
        int rc = dsw_check_launch();
        if (rc != DSW_OK) return rc;
    }
    const long total = (long)(K * Fin + 1) * Fout;
    dim3 rgrid((unsigned)((total + 255) / 256));
    if (dtype == DSW_F32)
        hipLaunchKernelGGL(cheb_wgrad_reduce_kernel<false>, rgrid, dim3(256), 0, stream, partial, (int)S,
                           (int)Fin, (int)Fout, (int)K, dW, db);
    else
        hipLaunchKernelGGL(cheb_wgrad_reduce_kernel<true>, rgrid, dim3(256), 0, stream, partial, (int)S,
                           (int)Fin, (int)Fout, (int)K, dW, db);
    return dsw_check_launch();
}
