// Shared definitions of the tall-skinny MFMA contractions (dsw_gemm.hip, dsw_gemm_x3.hip).
#pragma once
#include "dsw_common.h"
#include <cstdlib>
#include <type_traits>

namespace dsw_gemm {

#ifndef DSW_GEMM_PF
#define DSW_GEMM_PF 3   // A-chunk prefetch ring depth of the resident-panel GEMM
#endif
#ifndef DSW_WGRAD_PF
#define DSW_WGRAD_PF 3  // chunk prefetch ring depth of wgrad
#endif

constexpr int BM = 128;       // rows of the tall operand per workgroup (4 waves x 32)
constexpr int BN = 64;        // output columns per workgroup (2 MFMA tiles per wave)
constexpr int BK = 32;        // reduction chunk staged in LDS
constexpr int LDA = BK + 4;   // +4 floats: conflict-free ds_read_b128 of 16 rows (stride 36 words)

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
    // output: n_planes_c planes of [M, ldc]; plane 0 = C0, plane q>0 = C1 + (q-1)*c_plane_stride.
    // Output columns are addressed flattened: j = q * n_per_plane + n.
    void* C0;
    void* C1;
    size_t c_plane_stride;
    int ldc;
    int n_planes_c;
    int n_per_plane;        // valid columns per output plane
    const void* bias;       // [n_per_plane] or null (same dtype as the data)
    int bias_plane0;        // 1: the bias belongs to output plane 0 only (mix-first forward: Z_0 = X W_0 + b)
    long M;
    int a_vec;              // 1 if float4/bf16x4 loads of A are legal
    int dbg;                // ablation builds only (-DDSW_ABLATION + DSW_DBG env): 1 = skip the epilogue stores, 2 = skip the MFMAs; else 0
    int relu;               // 1: ReLU in the epilogue (ConvBlock: conv -> relu, my_models_graph.py:108-114), after the bias
    // Residual-block epilogue (my_models_graph.py:205-216: `x_out *= rezero_weight; x_out += res_connection(x)`) and
    // gradient accumulation:  C = act(scale * (acc + bias) + R)
    const void* scale;      // ONE device scalar of the data dtype (the ReZero parameter) applied to every output plane, or null
    const void* R;          // [M, ldr] operand of the data dtype added to output PLANE 0 (later planes: nothing), or null
    int ldr;
    // Scratch for a per-call image of the small operand (caller-owned device memory, may be null): the streaming-W kernel
    // (dsw_gemm_x3s.hip) splits W into its three bf16 terms ONCE per call into it instead of once per workgroup and
    // chunk; launchers without that path park a folded copy of W there.
    void* pre_ws;
    long pre_bytes;
    // >= 0: output plane fold_q is produced with B(fold_q) - B(fold_q + 2) - the top of the adjoint / Clenshaw recurrence
    // subtracts the raw plane K-1 from plane K-3, and both come out of this GEMM (dsw_api.hip, dsw_fold_w_launch).  -1: none.
    int fold_q;
    // Balanced decomposition of the streaming-W kernel (dsw_gemm_x3s.hip, set by its launcher only): one partial tile per
    // workgroup + one ready flag per workgroup, carved out of pre_ws behind the image.  null: whole tiles per workgroup.
    float* sk_part;
    unsigned* sk_flags;
};

// epilogue activation; NaN stays NaN like torch.relu (fmaxf would turn it into 0)
static __device__ __forceinline__ float epi_act(const float v, const int relu) { return (relu && v < 0.f) ? 0.f : v; }

// act(scale * (acc + bias) + R[row]): `res` points at (row 0, this lane's column) of R or is null, `roff` = row * ldr.
// scale == 1 and res == null reproduce acc + bias bit for bit (x * 1.0f is exact).
template <bool BF16>
static __device__ __forceinline__ float epi_out(const float acc, const float bias, const float scale, const char* res,
                                                const size_t roff, const int relu) {
    float v = (acc + bias) * scale;
    if (res != nullptr) v += BF16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(res)[roff])
                                  : reinterpret_cast<const float*>(res)[roff];
    return epi_act(v, relu);
}
// The 16 residual values of one 32 x 32 accumulator tile, requested BACK TO BACK before any store of the tile: a load
// issued after a store to memory the compiler cannot prove distinct stays behind it, i.e. one exposed memory latency
// per element (measured: 56 -> 133 us on a 98 304 x 128 tile pass).
template <bool BF16>
static __device__ __forceinline__ void epi_res_load(float (&rv)[16], const char* res, const long rbase, const int ldr,
                                                    const long M) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        long r = rbase + (i & 3) + 8 * (i >> 2);
        r = r < M ? r : M - 1;                       // clamped: unconditional loads
        rv[i] = res != nullptr ? (BF16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(res)[(size_t)r * ldr])
                                       : reinterpret_cast<const float*>(res)[(size_t)r * ldr]) : 0.f;
    }
}
static __device__ __forceinline__ float epi_fin(const float acc, const float bias, const float scale, const float rv, const int relu) {
    return epi_act((acc + bias) * scale + rv, relu);
}
// per-column setup of the residual operand: plane-0 columns only
template <bool BF16>
static __device__ __forceinline__ const char* epi_res_ptr(const TsGemmParams& P, const bool col_ok, const int q, const int n) {
    return (P.R != nullptr && col_ok && q == 0) ? static_cast<const char*>(P.R) + (size_t)n * (BF16 ? 2 : 4) : nullptr;
}
template <bool BF16>
static __device__ __forceinline__ float epi_scale(const TsGemmParams& P) {
    return P.scale != nullptr ? ld1<BF16>(P.scale, 0) : 1.f;
}


// Position of a persistent workgroup in its (row tile, A plane, chunk) sequence, advanced incrementally: the
// equivalent  it / total, it % total, c / chunks  are 64-bit divisions that cost ~100 scalar instructions each,
// and the CU's single scalar unit is shared by all its waves (measured: the GEMM main loops were SALU-bound).
struct TileChunkIter {
    long row0;      // first row of the current row tile
    long it;        // linear iteration index
    int p, kc, c;   // A plane, chunk inside the plane, c = p * chunks + kc
    __device__ __forceinline__ void init(const long first_row) { row0 = first_row; it = 0; p = 0; kc = 0; c = 0; }
    // start in the middle of a tile: chunk c0 of the tile's `total` chunks
    __device__ __forceinline__ void init_at(const long first_row, const int c0, const int chunks) {
        row0 = first_row; it = 0; c = c0; p = c0 / chunks; kc = c0 - p * chunks;
    }
    __device__ __forceinline__ void next(const int chunks, const int total, const long row_step) {
        ++it; ++c; ++kc;
        if (kc == chunks) { kc = 0; ++p; }
        if (c == total) { c = 0; p = 0; row0 += row_step; }
    }
    // stays on the last valid iteration (tail prefetches re-read it harmlessly)
    __device__ __forceinline__ void next_clamped(const long n_iter, const int chunks, const int total, const long row_step) {
        if (it + 1 < n_iter) next(chunks, total, row_step);
    }
};

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
    // mix-first backward (dW_k = X^T D_k): the dY side has `dy_planes` planes (plane 0 = dY, plane z >= 1 =
    // dY1 + (z-1) * dy_plane_stride elements) and K == 1 on the T side; blockIdx.z = z * otiles + o-tile and the
    // partial rows of plane z are (z * Fin + f).  dy_planes <= 1: plain wgrad.
    const void* dY1;
    size_t dy_plane_stride;
    int dy_planes, otiles;
    // fused dgrad (cheb_wgrad_x3_kernel<..., FUSE>): the workgroup also writes G_k[n, f] = sum_o dY[n, o] W[f, k, o] for
    // the rows it streams (plane 0 -> G0, plane k >= 1 -> Grest + (k-1) * plane_stride elements)
    const void* W;
    void* G0;
    void* Grest;
    int fold;               // 1: dgrad plane K-3 with W[:, K-3, :] - W[:, K-1, :] (dsw_fold_w_launch done while the panel is filled)
    int dbg;                // reserved for ablation builds; 0
};


}  // namespace dsw_gemm
