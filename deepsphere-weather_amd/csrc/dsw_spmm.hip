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
#include "../../include/dsw_hip.h"
#include <cstdlib>
#include <cstring>

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
    // streaming (non-temporal) forms for data that this kernel touches exactly once (Z, Z2, Y): they
    // should not evict the gathered X rows, which are re-read k+1 times, from L2
    typedef float f4v __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load_nt(const void* base, size_t elem, float (&v)[4]) {
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(static_cast<const float*>(base) + elem));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store_nt(void* base, size_t elem, const float (&v)[4]) {
        f4v t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(static_cast<float*>(base) + elem));
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
    static __device__ __forceinline__ void load_nt(const void* b, size_t e, float (&v)[1]) { load(b, e, v); }
    static __device__ __forceinline__ void store_nt(void* b, size_t e, const float (&v)[1]) { store(b, e, v); }
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
    static __device__ __forceinline__ void load_nt(const void* b, size_t e, float (&v)[8]) { load(b, e, v); }
    static __device__ __forceinline__ void store_nt(void* b, size_t e, const float (&v)[8]) { store(b, e, v); }
};
template <>
struct Vec<true, 1> {
    static __device__ __forceinline__ void load(const void* base, size_t elem, float (&v)[1]) {
        v[0] = bf16_to_f32(static_cast<const uint16_t*>(base)[elem]);
    }
    static __device__ __forceinline__ void store(void* base, size_t elem, const float (&v)[1]) {
        static_cast<uint16_t*>(base)[elem] = f32_to_bf16(v[0]);
    }
    static __device__ __forceinline__ void load_nt(const void* b, size_t e, float (&v)[1]) { load(b, e, v); }
    static __device__ __forceinline__ void store_nt(void* b, size_t e, const float (&v)[1]) { store(b, e, v); }
};

// Long rows of spmm_csr_rowsplit: one WAVE per (row, sample).  Lane = (channel chunk c = lane % cpr, part p = lane / cpr);
// part p takes entries p, p + parts, ..; the parts of a chunk are added up by xor-shuffles and part 0 runs the epilogue.
// cpr is a power of two <= 64 (checked by the launcher).
template <bool BF16, int VEC>
static __device__ __forceinline__ void spmm_long_row_one(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, const void* Z2, float alpha, float beta, float gamma,
    int v_out, int v_in, int C, int cpr, int ldx, int ldy, int ldz, const int row, const int b) {
    using V = Vec<BF16, VEC>;
    const int lane = threadIdx.x & 63;
    const int c0 = (lane & (cpr - 1)) * VEC;
    const int part = lane / cpr, parts = 64 / cpr;
    const size_t xs = (size_t)v_in * ldx;
    const int s = rowptr[row], e = rowptr[row + 1];
    const size_t xb = (size_t)b * xs + c0;
    float acc[VEC];
#pragma unroll
    for (int jj = 0; jj < VEC; ++jj) acc[jj] = 0.f;
    int q = s + part;
    for (; q + 3 * parts < e; q += 4 * parts) {      // four entries per step: four gathers in flight
        int col[4];
        float a[4], x[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) { col[u] = colind[q + u * parts]; a[u] = vals[q + u * parts]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) V::load(X, xb + (size_t)col[u] * ldx, x[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jj = 0; jj < VEC; ++jj) acc[jj] = fmaf(a[u], x[u][jj], acc[jj]);
    }
    for (; q < e; q += parts) {
        float x[VEC];
        V::load(X, xb + (size_t)colind[q] * ldx, x);
        const float a = vals[q];
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) acc[jj] = fmaf(a, x[jj], acc[jj]);
    }
    for (int m = cpr; m < 64; m <<= 1) {
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) acc[jj] += __shfl_xor(acc[jj], m, 64);
    }
    if (part == 0) {
        const size_t off = ((size_t)b * v_out + row) * (size_t)C + c0;
        float o[VEC];
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) o[jj] = alpha * acc[jj];
        if (Z != nullptr) {
            float z[VEC];
            V::load(Z, ((size_t)b * v_out + row) * (size_t)ldz + c0, z);
#pragma unroll
            for (int jj = 0; jj < VEC; ++jj) o[jj] = fmaf(beta, z[jj], o[jj]);
        }
        if (Z2 != nullptr) {
            float z[VEC];
            V::load(Z2, off, z);
#pragma unroll
            for (int jj = 0; jj < VEC; ++jj) o[jj] = fmaf(gamma, z[jj], o[jj]);
        }
        V::store(Y, ((size_t)b * v_out + row) * (size_t)ldy + c0, o);
    }
}

// Which rows are long?  Without a plan the waves scan the row lengths 64 rows at a time (ballot; row r belongs to wave
// r % n_waves), so a launch without long rows pays a few microseconds of a handful of extra blocks.  With a plan of the
// operator (dsw_remap_plan: the remap matrices of the pooling layers) the long rows are LISTED: wave task t = (listed row
// t / B, sample t % B), exactly as many blocks as the list needs, no scan.
template <bool BF16, int VEC>
static __device__ __forceinline__ void spmm_long_rows(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, const void* Z2, float alpha, float beta, float gamma,
    int v_out, int v_in, int C, int cpr, int B, int flags, int ldx, int ldy, int ldz, int long_thr, long lblock,
    long lblocks, const int* __restrict__ long_list, int n_long) {
    const int lane = threadIdx.x & 63;
    const long all_waves = lblocks * (blockDim.x >> 6);
    const long gwave = lblock * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (long_list != nullptr) {
        const long tasks = (long)n_long * B;
        for (long t = gwave; t < tasks; t += all_waves)
            spmm_long_row_one<BF16, VEC>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, ldx, ldy,
                                         ldz, long_list[t / B], (int)(t % B));
        return;
    }
    // work item = (long row, sample): ONE sample per wave keeps this path within the registers of the main path (the
    // kernel's allocation is the maximum of both) and gives B waves to every long row
    const int bsplit = (all_waves >= 2L * B) ? B : 1;
    const long n_waves = all_waves / bsplit;
    const long wave = gwave / bsplit;
    const int b_first = bsplit > 1 ? (int)(gwave % bsplit) : 0;
    const int b_step = bsplit > 1 ? B : 1;               // with bsplit > 1 the sample loop below runs once
    if (wave >= n_waves) return;
    // row r belongs to wave r % n_waves: neighbouring long rows (the cells around a pole) go to different waves
    for (long kb = 0; kb * n_waves < v_out; kb += 64) {
        const long r = (kb + lane) * n_waves + wave;
        const int len = r < v_out ? rowptr[r + 1] - rowptr[r] : 0;
        unsigned long long todo = __ballot(len > long_thr);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int row = (int)((kb + j) * n_waves + wave);
            for (int b = b_first; b < B; b += b_step)
                spmm_long_row_one<BF16, VEC>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, ldx, ldy,
                                             ldz, row, b);
        }
    }
}

// One thread = (row, VEC-wide channel chunk) x NB samples.
template <bool BF16, int VEC, int NB>
__global__ __launch_bounds__(256) void spmm_csr_rowsplit(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, const void* Z2,
    float alpha, float beta, float gamma,
    int v_out, int v_in, int C, int cpr, int B, long row_blocks, int xcd_swizzle, int ldx, int ldy, int ldz,
    int long_thr, long main_blocks, const int* __restrict__ long_list, int n_long) {
    // ldx / ldy: elements between consecutive rows of X / Y (>= C: a channel slice of a wider node-major tensor, e.g.
    // one half of the decoder's concatenation buffer); ldz: the same for Z; Z2 is always dense [B, v_out, C]
    using V = Vec<BF16, VEC>;
    // XCD-aware block order: hardware block i runs on XCD i % 8 (each XCD has a private 4 MiB L2).
    // Remap so that every XCD walks ONE contiguous range of (batch group, row block) pairs: the
    // neighbour rows a block gathers were then fetched into the same L2 by its predecessors,
    // instead of every XCD pulling its own copy of every halo row from HBM / Infinity Cache.
    // Rows longer than long_thr entries (0: none are treated specially) are left to the blocks in front of the main ones, which
    // give each such row a whole wave: with one lane group per row a 300-entry row (the polar cells of a cross-sampling
    // pooling matrix) is a 300-step dependent chain that the rest of the launch waits for.
    const long lblocks = (long)gridDim.x - main_blocks;   // they come FIRST in the grid (a multiple of 8: the XCD phase stays)
    if ((long)blockIdx.x < lblocks) {
        spmm_long_rows<BF16, VEC>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, B,
                                      xcd_swizzle, ldx, ldy, ldz, long_thr, (long)blockIdx.x, lblocks, long_list, n_long);
        return;
    }
    const long nwg = main_blocks;
    const long orig = (long)blockIdx.x - lblocks;
    long wg = orig;
    if (xcd_swizzle & 1) {
        const long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const long bgrp = wg / row_blocks;
    const long rblk = wg - bgrp * row_blocks;
    const long gid = rblk * blockDim.x + threadIdx.x;
    const int row = (int)(gid / cpr);
    if (row >= v_out) return;
    const int c0 = (int)(gid - (long)row * cpr) * VEC;
    const int b0 = (int)bgrp * NB;

    float acc[NB][VEC];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[i][j] = 0.f;

    const size_t xs = (size_t)v_in * ldx;  // sample stride of X
    const int s = rowptr[row], e = rowptr[row + 1];
    if (long_thr > 0 && e - s > long_thr) return;   // a wave of the trailing blocks owns this row
    int p = s;
    for (; p + 2 <= e; p += 2) {
        const int col0 = colind[p], col1 = colind[p + 1];
        const float a0 = vals[p], a1 = vals[p + 1];
        float x0[NB][VEC], x1[NB][VEC];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = (b0 + i < B) ? b0 + i : B - 1;
            V::load(X, (size_t)b * xs + (size_t)col0 * ldx + c0, x0[i]);
            V::load(X, (size_t)b * xs + (size_t)col1 * ldx + c0, x1[i]);
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
            V::load(X, (size_t)b * xs + (size_t)col0 * ldx + c0, x0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[i][j] = fmaf(a0, x0[j], acc[i][j]);
        }
    }

    const size_t ys = (size_t)v_out * C;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (b0 + i >= B) break;
        const size_t off = (size_t)(b0 + i) * ys + (size_t)row * C + c0;
        const size_t offy = ((size_t)(b0 + i) * v_out + row) * (size_t)ldy + c0;
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = alpha * acc[i][j];
        if (Z != nullptr) {
            float z[VEC];
            const size_t offz = ((size_t)(b0 + i) * v_out + row) * (size_t)ldz + c0;
            if (xcd_swizzle & 2) V::load_nt(Z, offz, z); else V::load(Z, offz, z);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = fmaf(beta, z[j], o[j]);
        }
        if (Z2 != nullptr) {
            float z[VEC];
            if (xcd_swizzle & 2) V::load_nt(Z2, off, z); else V::load(Z2, off, z);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = fmaf(gamma, z[j], o[j]);
        }
        if (xcd_swizzle & 4) V::store_nt(Y, offy, o); else V::store(Y, offy, o);
    }
}

// LDS-tiled variant: a workgroup owns R consecutive rows of ONE sample.  It first copies the
// R x C block X[b, r0:r0+R, :] (contiguous in HBM, fully coalesced 16-B loads) into LDS, then every
// (row, chunk) lane gathers its neighbours from LDS when the column falls inside the tile and from
// global memory (L2) otherwise.  On HEALPix nested order a 256-row tile is a 16x16 pixel patch, so
// ~90 % (k=8) of the gathers are LDS reads and each input row crosses the L1/L2 path once instead
// of k+1 times.  Requires a square-ish operator only in the sense that tile rows index X rows
// (v_in >= v_out is not needed: columns outside [r0, r0+R) simply take the global path).
template <bool BF16, int VEC>
__global__ __launch_bounds__(256) void spmm_csr_tiled(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, const void* Z2,
    float alpha, float beta, float gamma,
    int v_out, int v_in, int C, int cpr, int R, int B) {
    using V = Vec<BF16, VEC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    constexpr int ES = BF16 ? 2 : 4;
    const int b = blockIdx.x % B;             // samples of one tile are neighbours in launch order
    const int r0 = (blockIdx.x / B) * R;
    const int rows_here = min(R, v_out - r0);
    const size_t xs = (size_t)v_in * C;
    const char* xb = static_cast<const char*>(X) + (size_t)b * xs * ES;

    // phase 1: stage the tile's own input rows (those that exist in X)
    const int stage_rows = max(0, min(R, v_in - r0));
    const int n16 = stage_rows * C * ES / 16;
    const uint4* src = reinterpret_cast<const uint4*>(xb + (size_t)r0 * C * ES);
    uint4* dst = reinterpret_cast<uint4*>(tile_raw);
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __syncthreads();

    const int rows_per_pass = 256 / cpr;
    const int lrow = threadIdx.x / cpr;
    const int c0 = (threadIdx.x - lrow * cpr) * VEC;
    if (lrow >= rows_per_pass) return;  // 256 % cpr leftover lanes
    const size_t ys = (size_t)v_out * C;
    for (int rr = lrow; rr < rows_here; rr += rows_per_pass) {
        const int row = r0 + rr;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        const int s = rowptr[row], e = rowptr[row + 1];
        int p = s;
        for (; p + 2 <= e; p += 2) {
            const int col0 = colind[p], col1 = colind[p + 1];
            const float a0 = vals[p], a1 = vals[p + 1];
            float x0[VEC], x1[VEC];
            const unsigned l0 = (unsigned)(col0 - r0), l1 = (unsigned)(col1 - r0);
            if (l0 < (unsigned)stage_rows) V::load(tile_raw, (size_t)l0 * C + c0, x0);
            else V::load(xb, (size_t)col0 * C + c0, x0);
            if (l1 < (unsigned)stage_rows) V::load(tile_raw, (size_t)l1 * C + c0, x1);
            else V::load(xb, (size_t)col1 * C + c0, x1);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[j] = fmaf(a0, x0[j], acc[j]);
                acc[j] = fmaf(a1, x1[j], acc[j]);
            }
        }
        if (p < e) {
            const int col0 = colind[p];
            const float a0 = vals[p];
            float x0[VEC];
            const unsigned l0 = (unsigned)(col0 - r0);
            if (l0 < (unsigned)stage_rows) V::load(tile_raw, (size_t)l0 * C + c0, x0);
            else V::load(xb, (size_t)col0 * C + c0, x0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(a0, x0[j], acc[j]);
        }
        const size_t off = (size_t)b * ys + (size_t)row * C + c0;
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = alpha * acc[j];
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

template <bool BF16, int VEC>
int launch_tiled(const int* rowptr, const int* colind, const float* vals, const void* X, void* Y,
                 const void* Z, const void* Z2, float alpha, float beta, float gamma, int v_out, int v_in,
                 int C, int B, int R, hipStream_t stream) {
    const int cpr = C / VEC;
    const int es = BF16 ? 2 : 4;
    const long tiles = (v_out + R - 1) / R;
    dim3 grid((unsigned)(tiles * B));
    const size_t lds = (size_t)R * C * es;
    DSW_LAUNCH((spmm_csr_tiled<BF16, VEC>), grid, dim3(256), lds, stream, rowptr, colind, vals, X, Y,
                       Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, R, B);
    return dsw_check_launch();
}

// long rows of a launch: `list` (device, n_long rows, threshold thr) from a plan of the operator, or nullptr = scan
struct LongRows { const int* list; int n_long; int thr; };

template <bool BF16, int VEC, int NB>
int launch_rowsplit(const int* rowptr, const int* colind, const float* vals, const void* X, void* Y,
                    const void* Z, const void* Z2, float alpha, float beta, float gamma, int v_out,
                    int v_in, int C, int B, hipStream_t stream, int hints = 0, int ldx = 0, int ldy = 0, int ldz = 0,
                    LongRows lrw = {nullptr, 0, 0}) {
    const int cpr = C / VEC;
    const long threads = (long)v_out * cpr;
    static const char* bs_env = dsw_diag_env("DSW_SPMM_BLOCK");   // diagnostics: threads per block (64..1024)
    const int bs = bs_env ? (atoi(bs_env) > 256 ? 256 : atoi(bs_env)) : 256;
    const long row_blocks = (threads + bs - 1) / bs;
    const long bgroups = (B + NB - 1) / NB;
    static const char* sw = dsw_diag_env("DSW_SPMM_XCD");  // "0" disables the XCD-aware block order (diagnostics)
    // bit0: XCD block order, bit1: nt loads of Z/Z2, bit2: nt stores of Y (env overrides the caller's hints)
    const int swz = sw ? atoi(sw) : (1 | (hints & 6));
    // long rows (see spmm_long_rows): blocks in front of the main ones give each a whole wave.  Listed by a plan: one wave
    // task per (listed row, sample), four per block; otherwise a few blocks scan the row lengths
    const bool lr = (cpr & (cpr - 1)) == 0 && cpr <= 64 && bs % 64 == 0;
    const bool listed = lr && lrw.list != nullptr;
    const int long_thr = listed ? lrw.thr : lr ? 64 : 0;
    long lblocks;
    if (listed) {
        lblocks = (((long)lrw.n_long * B + (bs / 64) - 1) / (bs / 64) + 7) & ~7L;   // (a multiple of 8: the XCD phase stays)
        if (lblocks > 4096) lblocks = 4096;
    } else {
        lblocks = lr ? (((long)v_out / 256 + 7) & ~7L) : 0;   // one wave per ~64 rows: a single scan step each
        if (lblocks > 256) lblocks = 256;
        if (lr && lblocks < 8) lblocks = 8;
    }
    const long main_blocks = row_blocks * bgroups;
    dim3 grid((unsigned)(main_blocks + lblocks));
    DSW_LAUNCH((spmm_csr_rowsplit<BF16, VEC, NB>), grid, dim3(bs), 0, stream, rowptr, colind,
                       vals, X, Y, Z, Z2, alpha, beta, gamma, v_out, v_in, C, cpr, B, row_blocks, swz, ldx > 0 ? ldx : C,
                       ldy > 0 ? ldy : C, ldz > 0 ? ldz : C, long_thr, main_blocks, listed ? lrw.list : nullptr,
                       listed ? lrw.n_long : 0);
    return dsw_check_launch();
}

// ---- interpolation pooling between the levels of a REGULAR hierarchy (HEALPix nested: 4 children per parent) -------------
// The remap matrix of such a pooling has m entries per row in columns m r .. m r + m - 1 (GROUPS: pooling, and the
// transposed unpooling), its counterpart one entry per row in column r / m (BROADCAST: unpooling, and the transposed
// pooling).  Both are pure streaming: no row pointers, no column indices, no dependent index -> address -> data chain -
// 16-byte lanes, m row loads and one store (or one load and m stores), the weights read from the CSR values (any values:
// the structure is what the plan certifies, dsw_amd/functional.py: CsrOperator.remap_plan).
struct RemapArgs {
    const float* vals;
    const void* X;
    void* Y;
    const void* Z;      // optional epilogue operand (the other consumer's gradient of a forked tensor), row stride ldz
    float beta;
    int rows;           // GROUPS: output rows; BROADCAST: input rows
    int m, C, cpr;
    int ldx, ldy, ldz;
};

template <bool BF16, int VEC, int M>   // M: 4 = the HEALPix hierarchy, 0 = run-time m
__global__ __launch_bounds__(256) void remap_groups_kernel(const RemapArgs P) {
    using V = Vec<BF16, VEC>;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(gid / P.cpr);
    if (r >= P.rows) return;
    const int c0 = (int)(gid - (long)r * P.cpr) * VEC;
    const int b = blockIdx.y;
    const int m = M > 0 ? M : P.m;
    const size_t xrow = ((size_t)b * P.rows + r) * (size_t)m;      // first input row of the group (v_in = m rows)
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if constexpr (M == 4) {
        const float4 w = *reinterpret_cast<const float4*>(P.vals + (size_t)r * 4);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        float x[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) V::load_nt(P.X, (xrow + u) * (size_t)P.ldx + c0, x[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wv[u], x[u][j], acc[j]);
    } else {
        for (int u = 0; u < m; ++u) {
            float x[VEC];
            V::load_nt(P.X, (xrow + u) * (size_t)P.ldx + c0, x);
            const float w = P.vals[(size_t)r * m + u];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, x[j], acc[j]);
        }
    }
    const size_t orow = (size_t)b * P.rows + r;
    if (P.Z != nullptr) {
        float z[VEC];
        V::load_nt(P.Z, orow * (size_t)P.ldz + c0, z);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(P.beta, z[j], acc[j]);
    }
    V::store(P.Y, orow * (size_t)P.ldy + c0, acc);
}

template <bool BF16, int VEC, int M>
__global__ __launch_bounds__(256) void remap_broadcast_kernel(const RemapArgs P) {
    using V = Vec<BF16, VEC>;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(gid / P.cpr);
    if (q >= P.rows) return;
    const int c0 = (int)(gid - (long)q * P.cpr) * VEC;
    const int b = blockIdx.y;
    const int m = M > 0 ? M : P.m;
    float x[VEC];
    V::load_nt(P.X, ((size_t)b * P.rows + q) * (size_t)P.ldx + c0, x);
    const size_t orow = ((size_t)b * P.rows + q) * (size_t)m;
    if constexpr (M == 4) {
        const float4 w = *reinterpret_cast<const float4*>(P.vals + (size_t)q * 4);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        float z[4][VEC];
        if (P.Z != nullptr) {
#pragma unroll
            for (int u = 0; u < 4; ++u) V::load_nt(P.Z, (orow + u) * (size_t)P.ldz + c0, z[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = wv[u] * x[j];
            if (P.Z != nullptr) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = fmaf(P.beta, z[u][j], o[j]);
            }
            V::store(P.Y, (orow + u) * (size_t)P.ldy + c0, o);
        }
    } else {
        for (int u = 0; u < m; ++u) {
            const float w = P.vals[(size_t)q * m + u];
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = w * x[j];
            if (P.Z != nullptr) {
                float z[VEC];
                V::load_nt(P.Z, (orow + u) * (size_t)P.ldz + c0, z);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = fmaf(P.beta, z[j], o[j]);
            }
            V::store(P.Y, (orow + u) * (size_t)P.ldy + c0, o);
        }
    }
}

// Generic remap rows of 6-30 entries (a pooling between two different samplings: 14 on average): PARTS lane groups share a
// row - part p takes entries p, p + PARTS, ... four at a time (four index loads, then four row loads in flight), the parts are
// added up by xor-shuffles and part 0 runs the epilogue.  With one lane group per row (spmm_csr_rowsplit) such a row is a chain
// of 7+ dependent index -> data round trips and the launch has 3 blocks per CU: 31 us for 95 MB in the C5 step.  Lane
// layout: chunk (cpr lanes) fastest, then part, then row; sample = blockIdx.y.  Listed long rows: the blocks in front.
template <bool BF16, int VEC, int NB>
__global__ __launch_bounds__(256) void remap_parts_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ colind, const float* __restrict__ vals,
    const void* __restrict__ X, void* Y, const void* Z, float beta, int v_out, int v_in, int C, int cpr, int parts, int B,
    int ldx, int ldy, int ldz, int long_thr, int lblocks, const int* __restrict__ long_list, int n_long) {
    using V = Vec<BF16, VEC>;
    // grid: lblocks * B blocks for the listed long rows (they come first), then row_blocks * ceil(B / NB) main blocks
    const long lb_all = (long)lblocks * B;
    if ((long)blockIdx.x < lb_all) {
        const int bl = (int)(blockIdx.x / lblocks);
        const long all_waves = (long)lblocks * 4;
        for (long t = ((long)blockIdx.x - (long)bl * lblocks) * 4 + (threadIdx.x >> 6); t < n_long; t += all_waves)
            spmm_long_row_one<BF16, VEC>(rowptr, colind, vals, X, Y, Z, nullptr, 1.f, beta, 0.f, v_out, v_in, C, cpr, ldx, ldy, ldz,
                                         long_list[t], bl);
        return;
    }
    // XCD-aware order of the main blocks (hardware block i runs on XCD i % 8, each with a private L2): every XCD walks ONE
    // contiguous range of (sample group, row block) pairs - neighbouring destination rows share their source rows (2.2
    // destinations per source row in the C5 pooling), which a round-robin deal would fetch into two or three L2s
    const long nwg = (long)gridDim.x - lb_all, orig = (long)blockIdx.x - lb_all;
    const long q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int bgroups = (B + NB - 1) / NB;
    const long row_blocks = nwg / bgroups;
    const int bg = (int)(wg / row_blocks);
    const int b0 = bg * NB;
    const int gl = cpr * parts;                               // lanes per row: a power of two <= 64
    const long gid = (wg - (long)bg * row_blocks) * 256 + threadIdx.x;
    const int row = (int)(gid / gl);
    if (row >= v_out) return;                                 // (whole lane groups leave together)
    const int lg = (int)(gid - (long)row * gl);
    const int part = lg / cpr;
    const int c0 = (lg - part * cpr) * VEC;
    const int s = rowptr[row], e = rowptr[row + 1];
    const bool mine = !(long_thr > 0 && e - s > long_thr);    // a wave of the front blocks owns a listed row
    const size_t xs = (size_t)v_in * ldx;
    size_t xb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) xb[i] = (size_t)(b0 + i < B ? b0 + i : B - 1) * xs + c0;
    float acc[NB][VEC];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[i][j] = 0.f;
    if (mine) {
        // four entries of this part per step: their indices and weights first, then 4 x NB row loads in flight - a row of the
        // C5 pooling (14 entries, 4 parts) is ONE step: the wave's life is two memory round trips, not eight
        for (int q = s + part; q < e; q += 4 * parts) {
            int col[4];
            float a[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + u * parts;
                const bool ok = qq < e;
                live[u] = ok;
                col[u] = ok ? colind[qq] : 0;
                a[u] = ok ? vals[qq] : 0.f;
            }
            float x[4][NB][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < NB; ++i) V::load(X, xb[i] + (size_t)col[u] * ldx, x[u][i]);   // (padding entries: row 0, weight 0)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < VEC; ++j)    // a padding slot read row 0: an Inf / NaN there must not reach this row (0 * Inf)
                        acc[i][j] = fmaf(a[u], live[u] ? x[u][i][j] : 0.f, acc[i][j]);
        }
    }
    for (int m = cpr; m < gl; m <<= 1) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[i][j] += __shfl_xor(acc[i][j], m, 64);
    }
    if (mine && part == 0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (b0 + i >= B) break;
            const size_t orow = (size_t)(b0 + i) * v_out + row;
            if (Z != nullptr) {
                float z[VEC];
                V::load_nt(Z, orow * (size_t)ldz + c0, z);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[i][j] = fmaf(beta, z[j], acc[i][j]);
            }
            V::store(Y, orow * (size_t)ldy + c0, acc[i]);
        }
    }
}

template <bool BF16, int VEC, int NB>
int launch_remap_parts_nb(const int* rowptr, const int* colind, const float* vals, const void* X, void* Y, const void* Z, float beta,
                          int v_out, int v_in, int C, int parts, int B, int ldx, int ldy, int ldz, LongRows lrw, hipStream_t stream) {
    const int cpr = C / VEC;
    const long threads = (long)v_out * cpr * parts;
    const bool listed = lrw.list != nullptr && lrw.n_long > 0;
    int lblocks = listed ? (int)((((long)lrw.n_long + 3) / 4 + 7) & ~7L) : 0;
    if (lblocks > 2048) lblocks = 2048;
    const long nblk = (threads + 255) / 256 * ((B + NB - 1) / NB) + (long)lblocks * B;
    if (nblk > 2147483647L) return DSW_ERR_BAD_ARG;
    dim3 grid((unsigned)nblk);
    DSW_LAUNCH((remap_parts_kernel<BF16, VEC, NB>), grid, dim3(256), 0, stream, rowptr, colind, vals, X, Y, Z, beta, v_out, v_in, C,
               cpr, parts, B, ldx, ldy, ldz, listed ? lrw.thr : 0, lblocks, listed ? lrw.list : nullptr, listed ? lrw.n_long : 0);
    return dsw_check_launch();
}

template <bool BF16, int VEC>
int launch_remap_parts(const int* rowptr, const int* colind, const float* vals, const void* X, void* Y, const void* Z, float beta,
                       int v_out, int v_in, int C, int parts, int B, int ldx, int ldy, int ldz, LongRows lrw, hipStream_t stream) {
    // samples per thread: the index loads are shared and 4 x NB row loads are in flight per lane; as many as leave the launch
    // at least ~4 blocks per CU
    const long rb = ((long)v_out * (C / VEC) * parts + 255) / 256;
    if (!BF16 && B >= 4 && rb * ((B + 3) / 4) >= 1024)
        return launch_remap_parts_nb<BF16, VEC, 4>(rowptr, colind, vals, X, Y, Z, beta, v_out, v_in, C, parts, B, ldx, ldy, ldz, lrw, stream);
    if (B >= 2 && rb * ((B + 1) / 2) >= 1024)
        return launch_remap_parts_nb<BF16, VEC, 2>(rowptr, colind, vals, X, Y, Z, beta, v_out, v_in, C, parts, B, ldx, ldy, ldz, lrw, stream);
    return launch_remap_parts_nb<BF16, VEC, 1>(rowptr, colind, vals, X, Y, Z, beta, v_out, v_in, C, parts, B, ldx, ldy, ldz, lrw, stream);
}

template <bool BF16, int VEC>
int launch_remap_regular(int kind, const RemapArgs& A, int B, hipStream_t stream) {
    const long threads = (long)A.rows * A.cpr;
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)B);
    if (kind == 1) {
        if (A.m == 4) DSW_LAUNCH((remap_groups_kernel<BF16, VEC, 4>), grid, dim3(256), 0, stream, A);
        else DSW_LAUNCH((remap_groups_kernel<BF16, VEC, 0>), grid, dim3(256), 0, stream, A);
    } else {
        if (A.m == 4) DSW_LAUNCH((remap_broadcast_kernel<BF16, VEC, 4>), grid, dim3(256), 0, stream, A);
        else DSW_LAUNCH((remap_broadcast_kernel<BF16, VEC, 0>), grid, dim3(256), 0, stream, A);
    }
    return dsw_check_launch();
}

}  // namespace

// Internal C++ entry used by dsw_api.hip
int dsw_spmm_launch_ld(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                       const void* X, int64_t ldx_, void* Y, int64_t ldy_, int64_t B, int64_t C, float alpha, const void* Z,
                       float beta, const void* Z2, float gamma, int dtype, hipStream_t stream, int hints, int64_t ldz_,
                       const int* long_list, int n_long, int long_thr) {
    const LongRows lrw = {long_list, n_long, long_thr};
    if (ldz_ <= 0) ldz_ = C;
    if (ldx_ < C || ldy_ < C || ldz_ < C || ldx_ > INT32_MAX || ldy_ > INT32_MAX || ldz_ > INT32_MAX) return DSW_ERR_BAD_ARG;
    const int ldx = (int)ldx_, ldy = (int)ldy_, ldz = (int)ldz_;
    const bool dense = ldx == C && ldy == C && ldz == C;
    if (v_out <= 0 || v_in <= 0 || B <= 0 || C <= 0) return (v_out == 0 || B == 0) ? DSW_OK : DSW_ERR_BAD_ARG;
    if (v_out > INT32_MAX || v_in > INT32_MAX || C > INT32_MAX || B > 65535 * 4) return DSW_ERR_BAD_ARG;
    if (Z == nullptr) beta = 0.f;
    if (Z2 == nullptr) gamma = 0.f;
    const int vec_ = dtype == DSW_BF16 ? 8 : 4;
    const bool al = dsw_aligned16(X) && dsw_aligned16(Y) && (Z == nullptr || dsw_aligned16(Z)) &&
                    (Z2 == nullptr || dsw_aligned16(Z2)) && ldx % vec_ == 0 && ldy % vec_ == 0 && ldz % vec_ == 0;
    const int vo = (int)v_out, vi = (int)v_in, c = (int)C, b = (int)B;
    // kernel choice: LDS-tiled when a >=64-row tile of one sample fits 32 KiB and lanes divide evenly
    static const char* force = dsw_diag_env("DSW_SPMM_KERNEL");  // "rowsplit" | "tiled" (diagnostics only)
    const int es = dtype == DSW_BF16 ? 2 : 4;
    const int vec = dtype == DSW_BF16 ? 8 : 4;
    int tile_rows = 0;
    if (dense && al && vo == vi && c % vec == 0 && c / vec <= 256) {  // square operators only: tile rows == tile columns
        static const char* tb = dsw_diag_env("DSW_SPMM_TILE_BYTES");
        const long tile_bytes = tb ? atol(tb) : 32768;
        long r = tile_bytes / ((long)c * es);
        if (r > 1024) r = 1024;
        const int rpp = 256 / (c / vec);
        r = (r / rpp) * rpp;
        if (r >= 64) tile_rows = (int)r;
    }
    const bool want_tiled = force ? (strcmp(force, "tiled") == 0) : false;
    if (tile_rows > 0 && want_tiled && (long)((vo + tile_rows - 1) / tile_rows) * b < 2147483647L) {
        if (dtype == DSW_F32)
            return launch_tiled<false, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, tile_rows, stream);
        if (dtype == DSW_BF16)
            return launch_tiled<true, 8>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, tile_rows, stream);
    }
    static const char* nbs = dsw_diag_env("DSW_SPMM_NB");  // diagnostics: samples per thread
    const int nbf = nbs ? atoi(nbs) : 0;
    if (dtype == DSW_F32) {
        if (al && c % 4 == 0) {
            if (nbf == 1) return launch_rowsplit<false, 4, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            if (nbf == 2) return launch_rowsplit<false, 4, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            // planned remap products with few output rows (a pooling to a coarse level: 12 288 rows of 8 lanes): fewer samples
            // per thread so that the launch has at least ~8 blocks per CU - the gathers are latency-bound chains, and with
            // 3 blocks per CU (NB = 4) the chip waits on them instead of hiding them
            if (lrw.list != nullptr) {
                int nb = b >= 4 ? 4 : b >= 2 ? 2 : 1;
                while (nb > 1 && ((long)vo * (c / 4) + 255) / 256 * ((b + nb - 1) / nb) < 2048) nb >>= 1;
                if (nb == 1) return launch_rowsplit<false, 4, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
                if (nb == 2) return launch_rowsplit<false, 4, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            }
            if (b >= 4) return launch_rowsplit<false, 4, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            if (b >= 2) return launch_rowsplit<false, 4, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            return launch_rowsplit<false, 4, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
        }
        if (b >= 4) return launch_rowsplit<false, 1, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
        return launch_rowsplit<false, 1, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
    }
    if (dtype == DSW_BF16) {
        if (al && c % 8 == 0) {
            if (b >= 2) return launch_rowsplit<true, 8, 2>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
            return launch_rowsplit<true, 8, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
        }
        if (b >= 4) return launch_rowsplit<true, 1, 4>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
        return launch_rowsplit<true, 1, 1>(rowptr, colind, vals, X, Y, Z, Z2, alpha, beta, gamma, vo, vi, c, b, stream, hints, ldx, ldy, ldz, lrw);
    }
    return DSW_ERR_BAD_DTYPE;
}

int dsw_spmm_launch(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                    const void* X, void* Y, int64_t B, int64_t C, float alpha, const void* Z, float beta,
                    const void* Z2, float gamma, int dtype, hipStream_t stream, int hints) {
    return dsw_spmm_launch_ld(rowptr, colind, vals, v_out, v_in, X, C, Y, C, B, C, alpha, Z, beta, Z2, gamma, dtype, stream,
                              hints, C, nullptr, 0, 0);
}

// Interpolation-pooling product Y = M X (+ beta Z) with a plan of M (include/dsw_hip.h: dsw_remap_plan).
int dsw_remap_launch(const dsw_remap_plan* plan, const int* rowptr, const int* colind, const float* vals, int64_t v_out,
                     int64_t v_in, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C, const void* Z,
                     int64_t ldz, float beta, int dtype, hipStream_t stream) {
    if (Z == nullptr) { beta = 0.f; ldz = C; }
    const int vec = dtype == DSW_BF16 ? 8 : 4;
    const bool al = dsw_aligned16(X) && dsw_aligned16(Y) && (Z == nullptr || dsw_aligned16(Z)) && dsw_aligned16(vals) &&
                    ldx % vec == 0 && ldy % vec == 0 && ldz % vec == 0 && C % vec == 0;
    const int kind = plan ? plan->kind : 0;
    if ((kind == 1 || kind == 2) && al && plan->m >= 1 && B <= 65535 && v_out > 0 && v_in > 0 && B > 0 &&
        ldx <= INT32_MAX && ldy <= INT32_MAX && ldz <= INT32_MAX && v_out <= INT32_MAX && v_in <= INT32_MAX &&
        ((kind == 1 && v_in == v_out * plan->m) || (kind == 2 && v_out == v_in * plan->m))) {
        RemapArgs A;
        A.vals = vals; A.X = X; A.Y = Y; A.Z = Z; A.beta = beta;
        A.rows = (int)(kind == 1 ? v_out : v_in); A.m = plan->m; A.C = (int)C; A.cpr = (int)(C / vec);
        A.ldx = (int)ldx; A.ldy = (int)ldy; A.ldz = (int)ldz;
        if (dtype == DSW_F32) return launch_remap_regular<false, 4>(kind, A, (int)B, stream);
        if (dtype == DSW_BF16) return launch_remap_regular<true, 8>(kind, A, (int)B, stream);
        return DSW_ERR_BAD_DTYPE;
    }
    const bool listed = plan != nullptr && plan->long_rows != nullptr && plan->long_thr > 0;
    // rows of 6+ entries on average (and every long row listed, or none long): lane groups share the rows
    if (plan != nullptr && kind == 0 && plan->parts > 1 && al && B <= 65535 && B > 0 && v_out > 0 && v_in > 0 &&
        ldx <= INT32_MAX && ldy <= INT32_MAX && ldz <= INT32_MAX && v_out <= INT32_MAX && v_in <= INT32_MAX) {
        const int cpr = (int)(C / vec);
        int parts = plan->parts;
        while (parts > 1 && cpr * parts > 64) parts >>= 1;
        if (parts > 1 && (cpr & (cpr - 1)) == 0 && (parts & (parts - 1)) == 0) {
            const LongRows lrw = {listed ? plan->long_rows : nullptr, listed ? plan->n_long : 0, listed ? plan->long_thr : 0};
            if (dtype == DSW_F32)
                return launch_remap_parts<false, 4>(rowptr, colind, vals, X, Y, Z, beta, (int)v_out, (int)v_in, (int)C, parts,
                                                    (int)B, (int)ldx, (int)ldy, (int)ldz, lrw, stream);
            if (dtype == DSW_BF16)
                return launch_remap_parts<true, 8>(rowptr, colind, vals, X, Y, Z, beta, (int)v_out, (int)v_in, (int)C, parts,
                                                   (int)B, (int)ldx, (int)ldy, (int)ldz, lrw, stream);
            return DSW_ERR_BAD_DTYPE;
        }
    }
    return dsw_spmm_launch_ld(rowptr, colind, vals, v_out, v_in, X, ldx, Y, ldy, B, C, 1.f, Z, beta, nullptr, 0.f, dtype, stream,
                              0, ldz, listed ? plan->long_rows : nullptr, listed ? plan->n_long : 0, listed ? plan->long_thr : 0);
}
