// Max-value pooling / unpooling over a sparse remap matrix (reference: modules/layers.py:1040-1103).
//
//   pool     for every coarse cell d, sample b and channel f:  p* = argmax_p  w[p] * x[b, col[p], f]  over the
//            non-zeros p of row d of the pooling matrix (first maximum on ties, NaN counts as maximal - torch.argmax);
//            y[b,d,f] = x[b, col[p*], f] (the UNWEIGHTED value),  sel[b,d,f] = col[p*]
//   pool^T   dx[b,v,f] = sum over { d : sel[b,d,f] == v } dy[b,d,f]        (autograd of the reference's torch.gather)
//   unpool   y[b,v,f] = x[b,d,f] for the LARGEST d with sel[b,d,f] == v, else 0
//            (the reference's torch.index_put without accumulate into zeros: on CPU the last write wins, and writes run
//            in increasing d for a fixed sample / channel)
//   unpool^T dx[b,d,f] = dy[b, sel[b,d,f], f]                               (autograd of index_put: a plain gather)
//
// The reference does the pooling with a Python Counter loop over the rows of the matrix and materialises a
// [nnz, B*F] gather (layers.py:1054-1072) and a [2, B*F*Vd] int64 index tensor; here the selection is ONE int32 per
// output element and every kernel is a streaming pass in the native [B, V, C] layout: a lane owns VEC consecutive
// channels of one output row (16 bytes when the rows allow it), so the gathers are whole row segments.  HBM-bound
// integer / compare work: no LDS, no MFMA.  pool^T is evaluated as a GATHER over the transposed matrix (deterministic,
// no atomics); unpool resolves duplicates with an integer atomicMax on the coarse index (deterministic as well).
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

namespace {

constexpr int PT = 256;

template <bool BF16, int VEC>
struct Px {
    // VEC consecutive channels at element offset i, widened to fp32
    static __device__ __forceinline__ void load(const void* p, size_t i, float (&v)[VEC]) {
        if constexpr (VEC == 1) {
            v[0] = BF16 ? bf16_to_f32(static_cast<const uint16_t*>(p)[i]) : static_cast<const float*>(p)[i];
        } else if constexpr (BF16) {   // VEC == 8: 16 bytes
            const uint4 t = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p) + i);
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
        } else {                       // VEC == 4: 16 bytes
            const float4 t = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
    }
    static __device__ __forceinline__ void store(void* p, size_t i, const float (&v)[VEC]) {
        if constexpr (VEC == 1) {
            if constexpr (BF16) static_cast<uint16_t*>(p)[i] = f32_to_bf16(v[0]);
            else static_cast<float*>(p)[i] = v[0];
        } else if constexpr (BF16) {
            uint4 t;
            t.x = f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16); t.y = f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            t.z = f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16); t.w = f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p) + i) = t;
        } else {
            *reinterpret_cast<float4*>(static_cast<float*>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
};

template <int VEC>
static __device__ __forceinline__ void load_sel(const int* p, size_t i, int (&s)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = p[i + j];
}

// one lane = (sample, output row, VEC channels); lanes of a row are adjacent -> row segments are read whole
template <bool BF16, int VEC>
__global__ __launch_bounds__(PT) void maxval_pool_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ colind,
                                                             const float* __restrict__ vals, const void* __restrict__ X,
                                                             void* __restrict__ Y, int* __restrict__ sel, long v_out,
                                                             long v_in, long B, int C) {
    const int cpr = C / VEC;
    const long total = B * v_out * cpr;
    for (long t = (long)blockIdx.x * PT + threadIdx.x; t < total; t += (long)gridDim.x * PT) {
        const long row = t / cpr;                 // b * v_out + d
        const int c0 = (int)(t - row * cpr) * VEC;
        const long b = row / v_out, d = row - b * v_out;
        const int p0 = rowptr[d], p1 = rowptr[d + 1];
        float best[VEC], out[VEC];
        int arg[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { best[j] = 0.f; out[j] = 0.f; arg[j] = -1; }
        for (int p = p0; p < p1; ++p) {
            const int col = colind[p];
            const float w = vals[p];
            float x[VEC];
            Px<BF16, VEC>::load(X, ((size_t)b * v_in + col) * C + c0, x);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float s = w * x[j];
                // first maximum wins; NaN is maximal and sticks (torch.argmax semantics)
                const bool take = arg[j] < 0 || s > best[j] || (s != s && best[j] == best[j]);
                if (take) { best[j] = s; out[j] = x[j]; arg[j] = col; }
            }
        }
        Px<BF16, VEC>::store(Y, (size_t)row * C + c0, out);
#pragma unroll
        for (int j = 0; j < VEC; ++j) sel[(size_t)row * C + c0 + j] = arg[j];
    }
}

// dx[b,v,:] = sum_{d in row v of M^T, sel[b,d,:] == v} dy[b,d,:]   (rowptr_t / colind_t: CSR of the transposed matrix)
template <bool BF16, int VEC>
__global__ __launch_bounds__(PT) void maxval_pool_bwd_kernel(const int* __restrict__ rowptr_t, const int* __restrict__ colind_t,
                                                             const void* __restrict__ dY, const int* __restrict__ sel,
                                                             void* __restrict__ dX, long v_fine, long v_coarse, long B, int C) {
    const int cpr = C / VEC;
    const long total = B * v_fine * cpr;
    for (long t = (long)blockIdx.x * PT + threadIdx.x; t < total; t += (long)gridDim.x * PT) {
        const long row = t / cpr;
        const int c0 = (int)(t - row * cpr) * VEC;
        const long b = row / v_fine, v = row - b * v_fine;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int p = rowptr_t[v]; p < rowptr_t[v + 1]; ++p) {
            const size_t o = ((size_t)b * v_coarse + colind_t[p]) * C + c0;
            float g[VEC];
            int s[VEC];
            Px<BF16, VEC>::load(dY, o, g);
            load_sel<VEC>(sel, o, s);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += (s[j] == (int)v) ? g[j] : 0.f;
        }
        Px<BF16, VEC>::store(dX, (size_t)row * C + c0, acc);
    }
}

// winner[b, sel[b,d,f], f] = max(d)
__global__ __launch_bounds__(PT) void maxval_unpool_mark_kernel(const int* __restrict__ sel, int* __restrict__ winner,
                                                                long v_coarse, long v_fine, long B, int C) {
    const long total = B * v_coarse * C;
    for (long t = (long)blockIdx.x * PT + threadIdx.x; t < total; t += (long)gridDim.x * PT) {
        const long row = t / C;
        const int f = (int)(t - row * C);
        const long b = row / v_coarse, d = row - b * v_coarse;
        const int v = sel[t];
        if (v >= 0 && v < v_fine) atomicMax(&winner[((size_t)b * v_fine + v) * C + f], (int)d);
    }
}

template <bool BF16>
__global__ __launch_bounds__(PT) void maxval_unpool_fill_kernel(const int* __restrict__ winner, const void* __restrict__ X,
                                                                void* __restrict__ Y, long v_coarse, long v_fine, long B, int C) {
    const long total = B * v_fine * C;
    for (long t = (long)blockIdx.x * PT + threadIdx.x; t < total; t += (long)gridDim.x * PT) {
        const long row = t / C;
        const int f = (int)(t - row * C);
        const long b = row / v_fine;
        const int d = winner[t];
        float v[1] = {0.f};
        if (d >= 0) Px<BF16, 1>::load(X, ((size_t)b * v_coarse + d) * C + f, v);
        Px<BF16, 1>::store(Y, (size_t)t, v);
    }
}

template <bool BF16>
__global__ __launch_bounds__(PT) void maxval_unpool_bwd_kernel(const int* __restrict__ sel, const void* __restrict__ dY,
                                                               void* __restrict__ dX, long v_coarse, long v_fine, long B, int C) {
    const long total = B * v_coarse * C;
    for (long t = (long)blockIdx.x * PT + threadIdx.x; t < total; t += (long)gridDim.x * PT) {
        const long row = t / C;
        const int f = (int)(t - row * C);
        const long b = row / v_coarse;
        const int v = sel[t];
        float g[1] = {0.f};
        if (v >= 0 && v < v_fine) Px<BF16, 1>::load(dY, ((size_t)b * v_fine + v) * C + f, g);
        Px<BF16, 1>::store(dX, (size_t)t, g);
    }
}

static unsigned grid_for(long total) {
    long g = (total + PT - 1) / PT;
    if (g < 1) g = 1;
    if (g > 256L * 32) g = 256L * 32;     // grid-stride beyond 32 blocks per CU
    return (unsigned)g;
}

}  // namespace

extern "C" {

int dsw_maxval_pool_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                        const void* X, void* Y, int32_t* sel, int64_t B, int64_t C, int dtype, dsw_stream_t stream) {
    if (v_out < 0 || v_in < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (v_out == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !colind || !vals || !X || !Y || !sel) return DSW_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == DSW_BF16 ? 8 : 4;
    const bool wide = (C % vec == 0) && dsw_aligned16(X) && dsw_aligned16(Y);
    const long total = B * v_out * (wide ? C / vec : C);
#define DSW_POOL_FWD(BF, V_)                                                                                             \
    DSW_LAUNCH((maxval_pool_fwd_kernel<BF, V_>), dim3(grid_for(total)), dim3(PT), 0, s, rowptr, colind, vals, X, \
                       Y, sel, (long)v_out, (long)v_in, (long)B, (int)C)
    if (dtype == DSW_BF16) { if (wide) DSW_POOL_FWD(true, 8); else DSW_POOL_FWD(true, 1); }
    else { if (wide) DSW_POOL_FWD(false, 4); else DSW_POOL_FWD(false, 1); }
#undef DSW_POOL_FWD
    return dsw_check_launch();
}

int dsw_maxval_pool_bwd(const int32_t* rowptr_t, const int32_t* colind_t, int64_t v_fine, int64_t v_coarse,
                        const void* dY, const int32_t* sel, void* dX, int64_t B, int64_t C, int dtype,
                        dsw_stream_t stream) {
    if (v_fine < 0 || v_coarse < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (v_fine == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr_t || !colind_t || !dY || !sel || !dX) return DSW_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == DSW_BF16 ? 8 : 4;
    const bool wide = (C % vec == 0) && dsw_aligned16(dY) && dsw_aligned16(dX);
    const long total = B * v_fine * (wide ? C / vec : C);
#define DSW_POOL_BWD(BF, V_)                                                                                               \
    DSW_LAUNCH((maxval_pool_bwd_kernel<BF, V_>), dim3(grid_for(total)), dim3(PT), 0, s, rowptr_t, colind_t, dY, sel, \
                       dX, (long)v_fine, (long)v_coarse, (long)B, (int)C)
    if (dtype == DSW_BF16) { if (wide) DSW_POOL_BWD(true, 8); else DSW_POOL_BWD(true, 1); }
    else { if (wide) DSW_POOL_BWD(false, 4); else DSW_POOL_BWD(false, 1); }
#undef DSW_POOL_BWD
    return dsw_check_launch();
}

int64_t dsw_maxval_unpool_workspace_bytes(int64_t B, int64_t v_fine, int64_t C) {
    if (B < 0 || v_fine < 0 || C < 0) return DSW_ERR_BAD_ARG;
    return B * v_fine * C * 4;
}

int dsw_maxval_unpool_fwd(const int32_t* sel, const void* X, void* Y, void* workspace, int64_t workspace_bytes,
                          int64_t B, int64_t v_coarse, int64_t v_fine, int64_t C, int dtype, dsw_stream_t stream) {
    if (v_fine < 0 || v_coarse < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (v_fine == 0 || B == 0 || C == 0) return DSW_OK;
    if (!Y || (v_coarse > 0 && (!sel || !X))) return DSW_ERR_BAD_ARG;
    const int64_t need = dsw_maxval_unpool_workspace_bytes(B, v_fine, C);
    if (!workspace || workspace_bytes < need) return DSW_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int* winner = static_cast<int*>(workspace);
    if (hipMemsetAsync(winner, 0xFF, (size_t)need, s) != hipSuccess) return DSW_ERR_LAUNCH;   // -1 everywhere
    if (v_coarse > 0)
        DSW_LAUNCH(maxval_unpool_mark_kernel, dim3(grid_for(B * v_coarse * C)), dim3(PT), 0, s, sel, winner,
                           (long)v_coarse, (long)v_fine, (long)B, (int)C);
    if (dtype == DSW_BF16)
        DSW_LAUNCH(maxval_unpool_fill_kernel<true>, dim3(grid_for(B * v_fine * C)), dim3(PT), 0, s, winner, X, Y,
                           (long)v_coarse, (long)v_fine, (long)B, (int)C);
    else
        DSW_LAUNCH(maxval_unpool_fill_kernel<false>, dim3(grid_for(B * v_fine * C)), dim3(PT), 0, s, winner, X, Y,
                           (long)v_coarse, (long)v_fine, (long)B, (int)C);
    return dsw_check_launch();
}

int dsw_maxval_unpool_bwd(const int32_t* sel, const void* dY, void* dX, int64_t B, int64_t v_coarse, int64_t v_fine,
                          int64_t C, int dtype, dsw_stream_t stream) {
    if (v_fine < 0 || v_coarse < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (v_coarse == 0 || B == 0 || C == 0) return DSW_OK;
    if (!sel || !dY || !dX) return DSW_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DSW_BF16)
        DSW_LAUNCH(maxval_unpool_bwd_kernel<true>, dim3(grid_for(B * v_coarse * C)), dim3(PT), 0, s, sel, dY, dX,
                           (long)v_coarse, (long)v_fine, (long)B, (int)C);
    else
        DSW_LAUNCH(maxval_unpool_bwd_kernel<false>, dim3(grid_for(B * v_coarse * C)), dim3(PT), 0, s, sel, dY, dX,
                           (long)v_coarse, (long)v_fine, (long)B, (int)C);
    return dsw_check_launch();
}

}  // extern "C"
