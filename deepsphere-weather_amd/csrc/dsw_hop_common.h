// Device helpers shared by the LDS-staged sparse kernels (dsw_spmm2.hip: two hops fused; dsw_spmm1s.hip: one hop on
// staged neighbourhoods): 16-byte row segments widened to fp32, nontemporal tile-row accesses, the ELL row gather.
#pragma once
#include <type_traits>
#include "dsw_common.h"

namespace {
// tile-row operands are touched once per launch (Z2 read, Y1 / Y2 written): nontemporal; the gathered operands (U on
// the 2-ring, Z1 on the 1-ring) are shared by neighbouring tiles and stay on the cached path
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void st16(char* p, const uint4& v) {
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(p));
}
template <typename T4>
static __device__ __forceinline__ T4 ld16_once(const char* p) {
    return __builtin_bit_cast(T4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p)));
}


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs

typedef float f32x2 __attribute__((ext_vector_type(2)));       // v_pk_fma_f32 / v_pk_mul_f32 operands
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));      // v_cvt_pk_bf16_f32 result

static __device__ __forceinline__ f32x2 fmav(const f32x2 a, const f32x2 b, const f32x2 c) {
    return __builtin_elementwise_fma(a, b, c);
}
static __device__ __forceinline__ float fmav(const float a, const float b, const float c) { return fmaf(a, b, c); }

// 16 bytes of a row widened to fp32: bf16 -> 4 register PAIRS (packed fp32 math halves the VALU work of the
// 8-channel lane), fp32 -> 4 scalars (pairs would only cost registers there)
template <bool BF16>
struct Row16 {
    static constexpr int N = 4;
    using V = typename std::conditional<BF16, f32x2, float>::type;
    static __device__ __forceinline__ V splat(const float a) {
        if constexpr (BF16) return f32x2{a, a};
        else return a;
    }
    static __device__ __forceinline__ void unpack(const uint4 t, V (&v)[N]) {
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (BF16) v[i] = f32x2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
            else v[i] = __uint_as_float(w[i]);
        }
    }
    static __device__ __forceinline__ uint4 pack(const V (&v)[N]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (BF16)   // round-to-nearest-even, one v_cvt_pk_bf16_f32 per pair
                w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[i], bf16x2));
            else w[i] = __float_as_uint(v[i]);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// acc += sum_j val[j] * buf[idx[j]] over the first W entries of one ELL row (padded with {own row, 0}); the row
// INDICES (position in the staged list, u16) and the VALUES (fp32) are two separate arrays: 6 bytes per entry instead
// of 8 - what lets the k = 20 stencil (23 entries on 155 rows) keep two workgroups on a CU.  bufc = staging buffer +
// this lane's byte offset inside a row.  The chain index -> address -> data -> fma is a sequence of dependent LDS
// round trips, so 8 (then 4, then 2) entries are fetched per batch and their data rows requested back to back.
// BYTEOFF: the u16 entries are BYTE offsets of the rows (list position x row bytes < 64 KiB: the add folds the 16-bit
// select, v_add_u32_sdwa); otherwise they are list positions and the address costs a multiply-add.
template <bool BF16, bool BYTEOFF, int GB = 8>
static __device__ __forceinline__ void gather_ell(const unsigned short* __restrict__ row_idx, const float* __restrict__ row_val,
                                                  const int W, const unsigned row_bytes,
                                                  const unsigned char* __restrict__ bufc,
                                                  typename Row16<BF16>::V (&acc)[Row16<BF16>::N]) {
    using R = Row16<BF16>;
    using VT = typename R::V;
    constexpr int N = R::N;
    int j = 0;
    // GB = 4: batches of four only - 16 registers less, what the k = 20 adjoint variant needs to stay under 128
    for (; GB >= 8 && j + 8 <= W; j += 8) {
        const uint2 i0 = *reinterpret_cast<const uint2*>(row_idx + j), i1 = *reinterpret_cast<const uint2*>(row_idx + j + 4);
        const float4 v0 = *reinterpret_cast<const float4*>(row_val + j), v1 = *reinterpret_cast<const float4*>(row_val + j + 4);
        const unsigned ix[8] = {i0.x & 0xffffu, i0.x >> 16, i0.y & 0xffffu, i0.y >> 16,
                                i1.x & 0xffffu, i1.x >> 16, i1.y & 0xffffu, i1.y >> 16};
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint4 d[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) d[t] = *reinterpret_cast<const uint4*>(bufc + (BYTEOFF ? ix[t] : ix[t] * row_bytes));
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            VT x[N];
            R::unpack(d[t], x);
            const VT v = R::splat(vv[t]);
#pragma unroll
            for (int c = 0; c < N; ++c) acc[c] = fmav(v, x[c], acc[c]);
        }
    }
    for (; j + 4 <= W; j += 4) {
        const uint2 i0 = *reinterpret_cast<const uint2*>(row_idx + j);
        const float4 v0 = *reinterpret_cast<const float4*>(row_val + j);
        const unsigned ix[4] = {i0.x & 0xffffu, i0.x >> 16, i0.y & 0xffffu, i0.y >> 16};
        const float vv[4] = {v0.x, v0.y, v0.z, v0.w};
        uint4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = *reinterpret_cast<const uint4*>(bufc + (BYTEOFF ? ix[t] : ix[t] * row_bytes));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            VT x[N];
            R::unpack(d[t], x);
            const VT v = R::splat(vv[t]);
#pragma unroll
            for (int c = 0; c < N; ++c) acc[c] = fmav(v, x[c], acc[c]);
        }
    }
    if (j + 2 <= W) {   // a pair
        const unsigned i0 = *reinterpret_cast<const unsigned*>(row_idx + j);
        const float2 v0 = *reinterpret_cast<const float2*>(row_val + j);
        const uint4 d0 = *reinterpret_cast<const uint4*>(bufc + (BYTEOFF ? (i0 & 0xffffu) : (i0 & 0xffffu) * row_bytes));
        const uint4 d1 = *reinterpret_cast<const uint4*>(bufc + (BYTEOFF ? (i0 >> 16) : (i0 >> 16) * row_bytes));
        VT x0[N], x1[N];
        R::unpack(d0, x0); R::unpack(d1, x1);
        const VT va = R::splat(v0.x), vb = R::splat(v0.y);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            acc[c] = fmav(va, x0[c], acc[c]);
            acc[c] = fmav(vb, x1[c], acc[c]);
        }
        j += 2;
    }
    if (j < W) {        // odd loop length (HEALPix k = 8: 9 entries in 96 % of the rows): a single last entry
        const unsigned i0 = row_idx[j];
        const uint4 d0 = *reinterpret_cast<const uint4*>(bufc + (BYTEOFF ? i0 : i0 * row_bytes));
        VT x0[N];
        R::unpack(d0, x0);
        const VT va = R::splat(row_val[j]);
#pragma unroll
        for (int c = 0; c < N; ++c) acc[c] = fmav(va, x0[c], acc[c]);
    }
}

}  // namespace
