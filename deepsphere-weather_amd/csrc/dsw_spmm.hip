// CSR SpMM over node-major [B, V, C] activations with the fused Chebyshev axpby epilogue:
//
//     Y[b, r, :] = alpha * sum_p vals[p] * X[b, colind[p], :] + beta * Z[b, r, :] + gamma * Z2[b, r, :]
//
// One launch covers T_1 = L x (alpha=1), T_k = 2 L T_{k-1} - T_{k-2} (alpha=2, beta=-1), the adjoint
// steps G_{j-1} = 2 L^T G_j + G_{j-1} - G_{j+1}, and the rectangular pooling remap (alpha=1).
// Replaces the torch.sparse.mm call sites of /root/reference/modules/layers.py:164,167,962 and
// the permute/contiguous/cat traffic around them (layers.py:158-173).
//
// HBM-bound (~0.5 flop/byte).  Layout choices:
//   * a row of C channels is split over C/VEC lanes, each lane moving 16 B (float4 / 8 x bf16), so a
//     neighbour gather is one contiguous C*s-byte segment (128 B for 32 fp32 channels);
//   * a thread keeps NB samples of the batch in flight for the same (row, chunk): the CSR row is
//     read once per NB samples and NB independent gathers are outstanding per non-zero;
//   * rows are consecutive in a workgroup, so in HEALPix nested order (Morton curve per face) the
//     gathered neighbours of a 32..256-row tile mostly hit the CU's L1 / the XCD's L2.
#include "dsw_common.h"

namespace {

template <bool BF16, int VEC>
struct Vec;

template <>
struct Vec<false, 4> {
    static __device__ __forceinline__ void load(const void* base, size_t elem, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(void* base, size_t elem, const float (&v)[4]) {
        *reinterpret_cast<float4*>(static_cast<float*>(base) + elem) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct Vec<false, 1> {
    static __device__ __forceinline__ void load(const void* base, size_t elem, float (&v)[1]) {
        v[0] = static_cast<const float*>(base)[elem];
    }
    static __device__ __forceinline__ void store(void* base, size_t elem, const float (&v)[1]) {
        static_cast<float*>(base)[elem] = v[0];
    }
};
template <>
struct Vec<true, 8> {
    static __device__ __forceinline__ void load(const void* base, size_t elem, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(base) + elem);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(void* base, size_t elem, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(base) + elem) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <>
struct Vec<true, 1> {
    static __device__ __forceinline__ void load(const void* base, size_t elem, float (&v)[1]) {
        v[0] = bf16_to_f32(static_cast<const uint16_t*>(base)[elem]);
    }
    static __device__ __forceinline__ void store(void* base, size_t elem, const float (&v)[1]) {
        static_cast<uint16_t*>(base)[elem] = f32_to_bf16(v[0]);
    }
};

// One thread = (row, VEC-wide channel chunk) x NB samples.
template <bool BF16, int VEC, int NB>
__global__ __launch_bounds__(256) void spmm_csr_rowsplit(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, const void* Z2,
    float alpha, float beta, float gamma,
    int v_out, int v_in, int C, int cpr, int B) {
    using V = Vec<BF16, VEC>;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(gid / cpr);
    if (row >= v_out) return;
    const int c0 = (int)(gid - (long)row * cpr) * VEC;
    const int b0 = blockIdx.y * NB;

    float acc[NB][VEC];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[i][j] = 0.f;

    const size_t xs = (size_t)v_in * C;  // sample stride of X
    const int s = rowptr[row], e = rowptr[row + 1];
    int p = s;
    for (; p + 2 <= e; p += 2) {
        const int col0 = colind[p], col1 = colind[p + 1];
        const float a0 = vals[p], a1 = vals[p + 1];
        float x0[NB][VEC], x1[NB][VEC];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = (b0 + i < B) ? b0 + i : B - 1;
            V::load(X, (size_t)b * xs + (size_t)col0 * C + c0, x0[i]);
            V::load(X, (size_t)b * xs + (size_t)col1 * C + c0, x1[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[i][j] = fmaf(a0, x0[i][j], acc[i][j]);
                acc[i][j] = fmaf(a1, x1[i][j], acc[i][j]);
            }
    }
    if (p < e) {
        const int col0 = colind[p];
        const float a0 = vals[p];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = (b0 + i < B) ? b0 + i : B - 1;
            float x0[VEC];
            V::load(X, (size_t)b * xs + (size_t)col0 * C + c0, x0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[i][j] = fmaf(a0, x0[j], acc[i][j]);
        }
    }

    const size_t ys = (size_t)v_out * C;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (b0 + i >= B) break;
        const size_t off = (size_t)(b0 + i) * ys + (size_t)row * C + c0;
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = alpha * acc[i][j];
        if (Z != nullptr) {
            float z[VEC];
            V::load(Z, off, z);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = fmaf(beta, z[j], o[j]);
        }
        if (Z2 != nullptr) {
            float z[VEC];
            V::load(Z2, off, z);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = fmaf(gamma, z[j], o[j]);
        }
        V::store(Y, off, o);
    }
}

template <bool BF16, int VEC, int NB>
int launch_rowsplit(const int* rowptr, const int* colind, const float* vals, const void* X, void* Y,
                    const void* Z, const void* Z2, float alpha, float beta, float gamma, int v_out,
                    int v_in, int C, int B, hipStream_t stream) {
    const int cpr = C / VEC;
    const long threads = (long)v_out * cpr;
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)((B + NB - 1) / NB));
    hipLaunchKernelGGL((spmm_csr_rowsplit<BF16, VEC, NB>), grid, dim3(256), 0, stream, rowptr, colind,
                       vals, X, Y, Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, B);
    return dsw_check_launch();
}

}  // namespace

// Internal C++ entry used by dsw_api.hip
int dsw_spmm_launch(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                    const void* X, void* Y, int64_t B, int64_t C, float alpha, const void* Z, float beta,
                    const void* Z2, float gamma, int dtype, hipStream_t stream) {
    if (v_out <= 0 || v_in <= 0 || B <= 0 || C <= 0) return (v_out == 0 || B == 0) ? DSW_OK : DSW_ERR_BAD_ARG;
    if (v_out > INT32_MAX || v_in > INT32_MAX || C > INT32_MAX || B > 65535 * 4) return DSW_ERR_BAD_ARG;
    if (Z == nullptr) beta = 0.f;
    if (Z2 == nullptr) gamma = 0.f;
    const bool al = dsw_aligned16(X) && dsw_aligned16(Y) && (Z == nullptr || dsw_aligned16(Z)) &&
                    (Z2 == nullptr || dsw_aligned16(Z2));
    const int vo = (int)v_out, vi = (int)v_in, c = (int)C, b = (int)B;
    if (dtype == DSW_F32) {
        if (al && c % 4 == 0) {
            if (b >= 4) return launch_rowsplit<false, 4, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
            if (b >= 2) return launch_rowsplit<false, 4, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
            return launch_rowsplit<false, 4, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
        }
        if (b >= 4) return launch_rowsplit<false, 1, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
        return launch_rowsplit<false, 1, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
    }
    if (dtype == DSW_BF16) {
        if (al && c % 8 == 0) {
            if (b >= 2) return launch_rowsplit<true, 8, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
            return launch_rowsplit<true, 8, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
        }
        if (b >= 4) return launch_rowsplit<true, 1, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
        return launch_rowsplit<true, 1, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream);
    }
    return DSW_ERR_BAD_DTYPE;
}
