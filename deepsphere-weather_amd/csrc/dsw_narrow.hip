// Channel mix of NARROW layers on the vector ALU (fp32): K * Fout <= 16 output columns per node row - the model's output
// layer (64 -> 2 fields, my_models_graph.py:560-564) and its residual linear map.  A 64-column MFMA tile would carry 2-6
// live columns; these contractions are a few dozen FMAs per 256-byte row, i.e. plain streaming kernels:
//
//   forward  Z_q[n, o] = sum_f X[n, f] W[f, q, o] (+ bias[o] on plane 0, optional ReLU)          q < K, o < Fout
//   dgrad    dX[n, f]  = sum_{q, o} D_q[n, o] W[f, q, o]                                          (D_0 = dY)
//   wgrad    dW[f, q, o] = sum_n X[n, f] D_q[n, o],  db[o] = sum_n dY[n, o]      per-block partials -> deterministic reduce
//
// W is [Fin, K, Fout] (the reference's parameter layout, layers.py:243-244): the K * Fout weights of one input channel are
// contiguous.  A row of X is split over Fin / 4 lanes (16 bytes each, Fin / 4 a power of two <= 64); every lane keeps the
// 4 x KF weights of its channels in registers.
#include <type_traits>
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

int dsw_wgrad_reduce_launch(const float* partial, int64_t S, int64_t Fin, int64_t Fout, int64_t K, void* dW, void* db,
                            int64_t K_out, int64_t k_off, int db_cols, int dtype, hipStream_t stream, int accumulate);

namespace {

constexpr int MAXKF = 16;   // widest narrow side; kernels are compiled for 8 and 16 (register budget)
constexpr int NTH = 256;

struct NarrowArgs {
    const float* X;       // [N, Fin]
    const float* W;       // [Fin, KF]
    const float* bias;    // [Fout] or null (forward, plane 0)
    const float* D0;      // plane 0 of the narrow side ([N, Fout]): dY (backward) / unused (forward)
    const float* Drest;   // planes 1..K-1
    float* Z0;            // forward outputs: plane 0, planes 1..K-1
    float* Zrest;
    float* dX;            // dgrad output [N, Fin]
    float* partial;       // wgrad: [blocks][Fin + 1][KF]
    long N;
    int Fin, Fout, K, relu;
};

// sum over the LPR consecutive lanes of a row (every lane gets it): DPP butterflies inside a 16-lane row (vector-ALU rate),
// cross-row steps through the permute network
template <int LPR>
static __device__ __forceinline__ float lane_sum(float v) {
    auto dpp = [](const float x, auto C) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), decltype(C)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});                          // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});                          // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8) v += dpp(v, std::integral_constant<int, 0x141>{}); // row_half_mirror
    if constexpr (LPR >= 16) v += dpp(v, std::integral_constant<int, 0x140>{}); // row_mirror
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPR >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// the K * Fout values of the narrow side for one row (plane q = j / Fout)
template <int KFM>
static __device__ __forceinline__ void load_d(const NarrowArgs& P, const long r, const int KF, float (&d)[KFM]) {
#pragma unroll
    for (int j = 0; j < KFM; ++j) {
        if (j < KF) {
            const int q = j / P.Fout, o = j - q * P.Fout;
            d[j] = q == 0 ? P.D0[(size_t)r * P.Fout + o] : P.Drest[(size_t)(q - 1) * P.N * P.Fout + (size_t)r * P.Fout + o];
        } else {
            d[j] = 0.f;
        }
    }
}

template <int LPR, int KFM>
__global__ __launch_bounds__(NTH) void narrow_fwd_kernel(const NarrowArgs P) {
    constexpr int RPB = NTH / LPR;                              // rows per block and pass
    const int KF = P.K * P.Fout;
    const int c = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    float w[4][KFM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KFM; ++j) w[i][j] = j < KF ? P.W[(size_t)(4 * c + i) * KF + j] : 0.f;
    // the reduced sums land on every lane of the row: lane c stores the output columns j = c, c + LPR, ...
    constexpr int NS = (KFM + LPR - 1) / LPR;
    float bias[NS];
    float* dst[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int j = t * LPR + c;
        const int jq = j < KF ? j / P.Fout : 0, jo = j < KF ? j - jq * P.Fout : 0;
        bias[t] = (j < KF && jq == 0 && P.bias != nullptr) ? P.bias[jo] : 0.f;
        dst[t] = (j < KF) ? (jq == 0 ? P.Z0 : P.Zrest + (size_t)(jq - 1) * P.N * P.Fout) + jo : nullptr;
    }
    // two rows per thread and pass (independent load / reduction chains in flight together)
    const long stride = (long)gridDim.x * RPB;
    for (long r = (long)blockIdx.x * RPB + rl; r < P.N; r += 2 * stride) {
        const long r2 = r + stride < P.N ? r + stride : r;       // tail: the second row repeats the first (same stores)
        const float4 x = *reinterpret_cast<const float4*>(P.X + (size_t)r * P.Fin + 4 * c);
        const float4 y = *reinterpret_cast<const float4*>(P.X + (size_t)r2 * P.Fin + 4 * c);
        float mine[NS], mine2[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) { mine[t] = 0.f; mine2[t] = 0.f; }
#pragma unroll
        for (int j = 0; j < KFM; ++j) {
            if (j < KF) {                                       // uniform
                float a = fmaf(x.x, w[0][j], fmaf(x.y, w[1][j], fmaf(x.z, w[2][j], x.w * w[3][j])));
                float b = fmaf(y.x, w[0][j], fmaf(y.y, w[1][j], fmaf(y.z, w[2][j], y.w * w[3][j])));
                a = lane_sum<LPR>(a);
                b = lane_sum<LPR>(b);
                if (c == j % LPR) { mine[j / LPR] = a; mine2[j / LPR] = b; }
            }
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (dst[t] != nullptr) {
                float a = mine[t] + bias[t], b = mine2[t] + bias[t];
                if (P.relu) { a = a < 0.f ? 0.f : a; b = b < 0.f ? 0.f : b; }   // NaN stays NaN (torch.relu)
                dst[t][(size_t)r * P.Fout] = a;
                dst[t][(size_t)r2 * P.Fout] = b;
            }
        }
    }
}

template <int LPR, int KFM>
__global__ __launch_bounds__(NTH) void narrow_dgrad_kernel(const NarrowArgs P) {
    constexpr int RPB = NTH / LPR;
    const int KF = P.K * P.Fout;
    const int c = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    float w[4][KFM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KFM; ++j) w[i][j] = j < KF ? P.W[(size_t)(4 * c + i) * KF + j] : 0.f;
    // two rows per thread and pass: their (dependent-free) loads are in flight together - the loop is latency-bound
    const long stride = (long)gridDim.x * RPB;
    for (long r = (long)blockIdx.x * RPB + rl; r < P.N; r += 2 * stride) {
        const long r2 = r + stride < P.N ? r + stride : r;       // tail: the second row repeats the first (same store)
        float d[KFM], e[KFM];
        load_d<KFM>(P, r, KF, d);
        load_d<KFM>(P, r2, KF, e);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < KFM; ++j) {
            if (j < KF) {
                a.x = fmaf(d[j], w[0][j], a.x); a.y = fmaf(d[j], w[1][j], a.y);
                a.z = fmaf(d[j], w[2][j], a.z); a.w = fmaf(d[j], w[3][j], a.w);
                b.x = fmaf(e[j], w[0][j], b.x); b.y = fmaf(e[j], w[1][j], b.y);
                b.z = fmaf(e[j], w[2][j], b.z); b.w = fmaf(e[j], w[3][j], b.w);
            }
        }
        *reinterpret_cast<float4*>(P.dX + (size_t)r * P.Fin + 4 * c) = a;
        *reinterpret_cast<float4*>(P.dX + (size_t)r2 * P.Fin + 4 * c) = b;
    }
}

// partial[block][f][j] = sum over the block's rows of X[n, f] D_j[n]; row Fin = column sums of the D planes
template <int LPR, int KFM>
__global__ __launch_bounds__(NTH) void narrow_wgrad_kernel(const NarrowArgs P) {
    extern __shared__ float red[];                              // [RPB][Fin + 1][KF]
    constexpr int RPB = NTH / LPR;
    const int KF = P.K * P.Fout;
    const int c = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    float acc[4][KFM], cs[KFM];
#pragma unroll
    for (int j = 0; j < KFM; ++j) {
        cs[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = 0.f;
    }
    // a block owns a CONTIGUOUS range of rows (fixed by the grid): the summation order never depends on timing
    const long per = (P.N + gridDim.x - 1) / gridDim.x;
    const long r_begin = (long)blockIdx.x * per, r_end = r_begin + per < P.N ? r_begin + per : P.N;
    for (long r = r_begin + rl; r < r_end; r += 2 * RPB) {       // two rows per pass (loads in flight together)
        const bool two = r + RPB < r_end;
        const long r2 = two ? r + RPB : r;
        const float4 x = *reinterpret_cast<const float4*>(P.X + (size_t)r * P.Fin + 4 * c);
        float4 y = *reinterpret_cast<const float4*>(P.X + (size_t)r2 * P.Fin + 4 * c);
        float d[KFM], e[KFM];
        load_d<KFM>(P, r, KF, d);
        load_d<KFM>(P, r2, KF, e);
        if (!two) y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < KFM; ++j) {
            if (j < KF) {
                const float ej = two ? e[j] : 0.f;
                acc[0][j] = fmaf(y.x, ej, fmaf(x.x, d[j], acc[0][j])); acc[1][j] = fmaf(y.y, ej, fmaf(x.y, d[j], acc[1][j]));
                acc[2][j] = fmaf(y.z, ej, fmaf(x.z, d[j], acc[2][j])); acc[3][j] = fmaf(y.w, ej, fmaf(x.w, d[j], acc[3][j]));
                cs[j] += d[j] + ej;
            }
        }
    }
    const int slab = (P.Fin + 1) * KF;
    float* mine = red + (size_t)rl * slab;
#pragma unroll
    for (int j = 0; j < KFM; ++j) {
        if (j < KF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[(4 * c + i) * KF + j] = acc[i][j];
            if (c == 0) mine[P.Fin * KF + j] = cs[j];
        }
    }
    __syncthreads();
    float* out = P.partial + (size_t)blockIdx.x * slab;
    for (int e = threadIdx.x; e < slab; e += NTH) {
        float v = 0.f;
        for (int g = 0; g < RPB; ++g) v += red[(size_t)g * slab + e];   // fixed order
        out[e] = v;
    }
}

static bool narrow_ok(const void* X, const void* W, int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    if (dtype != DSW_F32 || N <= 0 || K * Fout > MAXKF || K * Fout < 1) return false;
    if (Fin < 16 || Fin > 256 || (Fin & (Fin - 1)) != 0) return false;     // Fin / 4 lanes: a power of two, 4..64
    return dsw_aligned16(X) && W != nullptr;
}

static unsigned narrow_grid(int64_t N, int rpb) {
    int64_t g = (N + rpb - 1) / rpb;
    return (unsigned)(g < 1024 ? g : 1024);   // 4 blocks per CU, every thread walks several rows: the weights load once
}

#define DSW_NARROW_LPR(KERNEL, LPR_, KF_, ...)                                               \
    if ((KF_) <= 8) {                                                                        \
        switch (LPR_) {                                                                      \
            case 4: DSW_LAUNCH((KERNEL<4, 8>), __VA_ARGS__); break;                  \
            case 8: DSW_LAUNCH((KERNEL<8, 8>), __VA_ARGS__); break;                  \
            case 16: DSW_LAUNCH((KERNEL<16, 8>), __VA_ARGS__); break;                \
            case 32: DSW_LAUNCH((KERNEL<32, 8>), __VA_ARGS__); break;                \
            default: DSW_LAUNCH((KERNEL<64, 8>), __VA_ARGS__); break;                \
        }                                                                                    \
    } else {                                                                                 \
        switch (LPR_) {                                                                      \
            case 4: DSW_LAUNCH((KERNEL<4, 16>), __VA_ARGS__); break;                 \
            case 8: DSW_LAUNCH((KERNEL<8, 16>), __VA_ARGS__); break;                 \
            case 16: DSW_LAUNCH((KERNEL<16, 16>), __VA_ARGS__); break;               \
            case 32: DSW_LAUNCH((KERNEL<32, 16>), __VA_ARGS__); break;               \
            default: DSW_LAUNCH((KERNEL<64, 16>), __VA_ARGS__); break;               \
        }                                                                                    \
    }

}  // namespace

// Each returns 1 if it took the call (*rc = status), 0 if the caller must use the matrix-core path.
int dsw_narrow_fwd_try(const void* X, const void* W, const void* bias, void* Z0, void* Zrest, int64_t N, int64_t Fin,
                       int64_t Fout, int64_t K, int dtype, hipStream_t stream, int relu, int* rc) {
    if (!narrow_ok(X, W, N, Fin, Fout, K, dtype) || !Z0 || (K > 1 && !Zrest)) return 0;
    NarrowArgs A{};
    A.X = static_cast<const float*>(X); A.W = static_cast<const float*>(W); A.bias = static_cast<const float*>(bias);
    A.Z0 = static_cast<float*>(Z0); A.Zrest = static_cast<float*>(Zrest);
    A.N = N; A.Fin = (int)Fin; A.Fout = (int)Fout; A.K = (int)K; A.relu = relu;
    DSW_NARROW_LPR(narrow_fwd_kernel, (int)(Fin / 4), K * Fout, dim3(narrow_grid(N, NTH / (int)(Fin / 4))), dim3(NTH), 0, stream, A)
    *rc = dsw_check_launch();
    return 1;
}

int dsw_narrow_dgrad_try(const void* dY, const void* D, const void* W, void* dX, int64_t N, int64_t Fin, int64_t Fout,
                         int64_t K, int dtype, hipStream_t stream, int* rc) {
    if (!narrow_ok(dX, W, N, Fin, Fout, K, dtype) || !dY || (K > 1 && !D)) return 0;
    NarrowArgs A{};
    A.W = static_cast<const float*>(W); A.D0 = static_cast<const float*>(dY); A.Drest = static_cast<const float*>(D);
    A.dX = static_cast<float*>(dX);
    A.N = N; A.Fin = (int)Fin; A.Fout = (int)Fout; A.K = (int)K;
    DSW_NARROW_LPR(narrow_dgrad_kernel, (int)(Fin / 4), K * Fout, dim3(narrow_grid(N, NTH / (int)(Fin / 4))), dim3(NTH), 0, stream, A)
    *rc = dsw_check_launch();
    return 1;
}

// `max_blocks`: slabs the caller's partial buffer holds ([max_blocks][Fin + 1][K * Fout] floats)
int dsw_narrow_wgrad_try(const void* X, const void* dY, const void* D, void* dW, void* db, float* partial,
                         int64_t max_blocks, int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                         hipStream_t stream, int* rc, int accumulate) {
    if (!narrow_ok(X, dW, N, Fin, Fout, K, dtype) || !dY || (K > 1 && !D) || !partial || max_blocks < 1) return 0;
    const int rpb = NTH / (int)(Fin / 4);
    const size_t lds = (size_t)rpb * (Fin + 1) * K * Fout * 4;
    if (lds > 64 * 1024) return 0;
    NarrowArgs A{};
    A.X = static_cast<const float*>(X); A.D0 = static_cast<const float*>(dY); A.Drest = static_cast<const float*>(D);
    A.partial = partial;
    A.N = N; A.Fin = (int)Fin; A.Fout = (int)Fout; A.K = (int)K;
    int64_t blocks = (N + 8 * rpb - 1) / (8 * rpb);            // >= 8 passes per block
    if (blocks > 1024) blocks = 1024;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    DSW_NARROW_LPR(narrow_wgrad_kernel, (int)(Fin / 4), K * Fout, dim3((unsigned)blocks), dim3(NTH), lds, stream, A)
    *rc = dsw_check_launch();
    if (*rc != DSW_OK) return 1;
    // partial rows are (f, j = q * Fout + o) = dW's own [Fin, K, Fout] order: reduce as a K = 1 layer of width K * Fout
    *rc = dsw_wgrad_reduce_launch(partial, blocks, Fin, K * Fout, 1, dW, db, 1, 0, (int)Fout, dtype, stream, accumulate);
    return 1;
}
