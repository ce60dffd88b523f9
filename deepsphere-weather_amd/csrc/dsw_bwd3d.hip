// WHOLE backward of a K = 3, 32 -> 64 (or 32 -> 32: one chunk phase, template parameter NCH) fp32 ConvCheb layer in ONE launch
// (+ the partial reduce), in the DUAL form:
//
//     U_0 = dY,   U_1 = L^T dY,   U_2 = 2 L^T U_1 - dY          (the Chebyshev basis of dY under L^T, 64 channels)
//     dX    = sum_k U_k W_k^T                                    (autograd of layers.py:163-178)
//     dW_k  = X^T U_k     ( = T_k(L X)^T dY = (T_k X)^T dY ),   db = 1^T dY
//
// Neither the forward's basis planes T_1 / T_2 nor the dgrad planes G_k exist in this form: the launch reads dY and X
// and writes dX (+ one slab of dW / db partials per workgroup) - 4 tensor passes instead of the 12 of the separate route
// (fused wgrad + dgrad pass: X, T_1, T_2, dY in, G_0..G_2 out; adjoint pair: G_0..G_2 in, dX out), and the forward no
// longer has to store T_1 / T_2 (dsw_cheb_bwd_needs_basis).  What it pays: the two L^T hops run on the 64 channels of dY
// instead of the 32 of the dgrad planes - on-chip work (LDS gathers), which a launch has room for that moves a third
// of the bytes.
//
// Structure = the one-launch forward (dsw_fwd3.hip: same tile plan - of L^T -, same LDS staging of the tile's two-ring,
// same gathers, same split images) run on dY in two 32-channel chunks per (tile, sample), with a different matrix phase:
//   * dX: A = W_k^T fragment of ONE chunk (16 dX channels x 32 dY channels, 3 planes x 3 terms in registers), B = the
//     tile rows of U_k from the split images.  The reduction over the 192 (k, o) pairs is cut in two between the wave
//     groups: waves 0-3 hold the weights of chunk 0 and work in chunk phase 0, waves 4-7 those of chunk 1; the first
//     partial travels through a private, L2-resident 8 KB slot of the workspace and the second group adds and stores.
//   * dW: while one group runs the dX products of a chunk, the OTHER group accumulates dW_k[:, chunk] = X^T U_k on the
//     matrix cores: both operands are read TRANSPOSED from the row-major split images (ds_read_b64_tr_b16, as
//     dsw_wgrad_x3.hip), the 16 x 16 accumulators stay in registers for the whole life of the (persistent) workgroup.
//     db rides along as U_0^T 1 (an all-ones B fragment: exact in bf16).
// Every product is the exact three-way bf16 split with six MFMA terms (as everywhere in this library: fp32 accuracy).
//
// LDS (80 KB, two workgroups per CU): two-ring rows of the dY chunk | U_1 on the one-ring | split images of U_0, U_1 |
// split image of the X tile rows (once per sample) | ELL of the tile.  The U_2 image aliases the two-ring buffer (dead
// after hop 1: the thread keeps its own U_0 row in registers for "- U_0"), which costs a FOURTH barrier per chunk - the
// next chunk's rows can only be written once everybody has left the matrix phase.
#include <cstdlib>
#if defined(DSW_ABL_D3_NODW) || defined(DSW_ABL_D3_NODX) || defined(DSW_ABL_D3_NOGATHER) || defined(DSW_ABL_D3_NOSPLIT) || \
    defined(DSW_ABL_D3_NOD) || defined(DSW_ABL_D3_NOLOAD) || defined(DSW_D3_TIMELINE)
#define DSW_ABLATION 1   // timing experiments (tools/build_variant1.sh): wrong results by design, refused by _native.load()
#endif
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);
int dsw_wgrad_reduce_launch(const float* partial, int64_t S, int64_t Fin, int64_t Fout, int64_t K, void* dW, void* db,
                            int64_t K_out, int64_t k_off, int db_cols, int dtype, hipStream_t stream, int accumulate);

namespace {

constexpr int NTHREADS = 512;
constexpr int RB = 128;             // bytes of one staged row (a 32-channel chunk of a dY row) and of a dX / X row
constexpr int RPP = 64;             // rows per pass of the 512 threads
constexpr int IMG_TERM = 64 * 64;   // one bf16 term of one image: [64 rows][32 bf16]
constexpr int IMG_PLANE = 3 * IMG_TERM;
constexpr int SLAB = (3 * 32 + 1) * 64;   // floats of one partial slab: [(k, f) | db][o] at Fout = 64 (what the scratch is sized for)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct DualArgs {
    const int* tile_meta;
    const int* s2_rows;
    const int* lrowptr;
    const unsigned short* lcol;
    const float* lval;
    const char* dY;
    const char* X;
    char* dX;            // may be null (the input needs no gradient)
    const float* W;      // [32][3][Fout]
    float* partial;      // [gridDim.x][SLAB] or null (no parameter gradients wanted)
    char* pscr;          // [gridDim.x][8 KB]: the chunk-0 partial of dX on its way from waves 0-3 to waves 4-7 (L2-resident)
    int V, n_tiles, max_n1, max_n2, bufx_rows;
    int B, n_chunks, spc, ell_w;
};

template <typename T4>
static __device__ __forceinline__ void st16_nt(char* p, const T4& v) {
    typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(p));
}
static __device__ __forceinline__ float trunc_bf16(float f) { return __uint_as_float(__float_as_uint(f) & 0xffff0000u); }
static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
static __device__ __forceinline__ void split3x8(const float (&f)[8], bf16x8_t& h, bf16x8_t& m, bf16x8_t& l) {
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r1[j] = f[j] - trunc_bf16(f[j]);
        r2[j] = r1[j] - trunc_bf16(r1[j]);
    }
    uint4 uh = {pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
    uint4 um = {pack2(r1[0], r1[1]), pack2(r1[2], r1[3]), pack2(r1[4], r1[5]), pack2(r1[6], r1[7])};
    uint4 ul = {pack2(r2[0], r2[1]), pack2(r2[2], r2[3]), pack2(r2[4], r2[5]), pack2(r2[6], r2[7])};
    h = __builtin_bit_cast(bf16x8_t, uh); m = __builtin_bit_cast(bf16x8_t, um); l = __builtin_bit_cast(bf16x8_t, ul);
}
// the three bf16 terms of 4 consecutive channels (quad c4 of 8) of image row `row` -> LDS, 8 bytes per term.  Image =
// [term][64 rows][64 B]; 16-byte chunk kc of row r sits at chunk kc ^ (2 * ((r >> 3) & 1)) (dsw_fwd3.hip): the row reads of
// the dX products AND the transposing reads of the dW products (4-row groups, one key per group) are conflict free.
static __device__ __forceinline__ void split_store(unsigned char* __restrict__ img, const int row, const unsigned c4,
                                                   const float (&f)[4]) {
#ifdef DSW_ABL_D3_NOSPLIT
    if (row >= 0) return;
#endif
    float r1[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r1[j] = f[j] - trunc_bf16(f[j]);
        r2[j] = r1[j] - trunc_bf16(r1[j]);
    }
    const unsigned off = (unsigned)row * 64u + ((((c4 >> 1) ^ (((unsigned)row >> 2) & 2u)) << 4) | ((c4 & 1u) << 3));
    unsigned char* base = img + off;
    *reinterpret_cast<uint2*>(base) = make_uint2(pack2(f[0], f[1]), pack2(f[2], f[3]));
    *reinterpret_cast<uint2*>(base + IMG_TERM) = make_uint2(pack2(r1[0], r1[1]), pack2(r1[2], r1[3]));
    *reinterpret_cast<uint2*>(base + 2 * IMG_TERM) = make_uint2(pack2(r2[0], r2[1]), pack2(r2[2], r2[3]));
}

// acc += sum_j val[j] * buf[pos[j]] over the first W entries of one ELL row (fp32 values + u8 list positions; dsw_fwd3.hip)
static __device__ __forceinline__ void gather_ell(const unsigned char* __restrict__ row_idx, const float* __restrict__ row_val,
                                                  const int W, const unsigned char* __restrict__ bufc, float (&acc)[4]) {
#ifdef DSW_ABL_D3_NOGATHER
    acc[0] = row_val[0]; return;
#endif
    int j = 0;
    for (; j + 4 <= W; j += 4) {
        const unsigned w = *reinterpret_cast<const unsigned*>(row_idx + j);
        const float4 v0 = *reinterpret_cast<const float4*>(row_val + j);
        const unsigned ix[4] = {(w & 0xffu) << 7, ((w >> 8) & 0xffu) << 7, ((w >> 16) & 0xffu) << 7, (w >> 24) << 7};
        const float vv[4] = {v0.x, v0.y, v0.z, v0.w};
        float4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = *reinterpret_cast<const float4*>(bufc + ix[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = fmaf(vv[t], d[t].x, acc[0]); acc[1] = fmaf(vv[t], d[t].y, acc[1]);
            acc[2] = fmaf(vv[t], d[t].z, acc[2]); acc[3] = fmaf(vv[t], d[t].w, acc[3]);
        }
    }
    if (j + 2 <= W) {
        const unsigned w = *reinterpret_cast<const unsigned short*>(row_idx + j);
        const float2 v0 = *reinterpret_cast<const float2*>(row_val + j);
        const float4 d0 = *reinterpret_cast<const float4*>(bufc + ((w & 0xffu) << 7));
        const float4 d1 = *reinterpret_cast<const float4*>(bufc + ((w >> 8) << 7));
        acc[0] = fmaf(v0.x, d0.x, acc[0]); acc[1] = fmaf(v0.x, d0.y, acc[1]);
        acc[2] = fmaf(v0.x, d0.z, acc[2]); acc[3] = fmaf(v0.x, d0.w, acc[3]);
        acc[0] = fmaf(v0.y, d1.x, acc[0]); acc[1] = fmaf(v0.y, d1.y, acc[1]);
        acc[2] = fmaf(v0.y, d1.z, acc[2]); acc[3] = fmaf(v0.y, d1.w, acc[3]);
        j += 2;
    }
    if (j < W) {
        const unsigned w = row_idx[j];
        const float v0 = row_val[j];
        const float4 d0 = *reinterpret_cast<const float4*>(bufc + (w << 7));
        acc[0] = fmaf(v0, d0.x, acc[0]); acc[1] = fmaf(v0, d0.y, acc[1]);
        acc[2] = fmaf(v0, d0.z, acc[2]); acc[3] = fmaf(v0, d0.w, acc[3]);
    }
}

#ifdef DSW_GATHER_N
// The loop form with the NEXT batch's position word and weights requested in front of this batch's row reads: one dependent
// LDS round trip per batch of four instead of two (positions -> rows).  ELL storage is padded to W % 4 == 0 entries per row,
// so the look-ahead read of the last batch stays inside the row (ell_w) or is skipped.
static __device__ __forceinline__ void gather_n(const unsigned char* __restrict__ row_idx, const float* __restrict__ row_val,
                                                const int W, const int Wpad, const unsigned char* __restrict__ bufc, float (&acc)[4]) {
    unsigned w = *reinterpret_cast<const unsigned*>(row_idx);
    float4 v0 = *reinterpret_cast<const float4*>(row_val);
    int j = 0;
    for (; j + 4 <= W; j += 4) {
        unsigned wn = 0u;
        float4 vn = {0.f, 0.f, 0.f, 0.f};
        if (j + 4 < Wpad) {
            wn = *reinterpret_cast<const unsigned*>(row_idx + j + 4);
            vn = *reinterpret_cast<const float4*>(row_val + j + 4);
        }
        const unsigned ix[4] = {(w & 0xffu) << 7, ((w >> 8) & 0xffu) << 7, ((w >> 16) & 0xffu) << 7, (w >> 24) << 7};
        const float vv[4] = {v0.x, v0.y, v0.z, v0.w};
        float4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = *reinterpret_cast<const float4*>(bufc + ix[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = fmaf(vv[t], d[t].x, acc[0]); acc[1] = fmaf(vv[t], d[t].y, acc[1]);
            acc[2] = fmaf(vv[t], d[t].z, acc[2]); acc[3] = fmaf(vv[t], d[t].w, acc[3]);
        }
        w = wn; v0 = vn;
    }
    const int rem = W - j;      // 0..3 entries left: their positions and weights are here already
    if (rem > 0) {
        const float4 d0 = *reinterpret_cast<const float4*>(bufc + ((w & 0xffu) << 7));
        float4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = d1;
        if (rem > 1) d1 = *reinterpret_cast<const float4*>(bufc + (((w >> 8) & 0xffu) << 7));
        if (rem > 2) d2 = *reinterpret_cast<const float4*>(bufc + (((w >> 16) & 0xffu) << 7));
        acc[0] = fmaf(v0.x, d0.x, acc[0]); acc[1] = fmaf(v0.x, d0.y, acc[1]);
        acc[2] = fmaf(v0.x, d0.z, acc[2]); acc[3] = fmaf(v0.x, d0.w, acc[3]);
        if (rem > 1) {
            acc[0] = fmaf(v0.y, d1.x, acc[0]); acc[1] = fmaf(v0.y, d1.y, acc[1]);
            acc[2] = fmaf(v0.y, d1.z, acc[2]); acc[3] = fmaf(v0.y, d1.w, acc[3]);
        }
        if (rem > 2) {
            acc[0] = fmaf(v0.z, d2.x, acc[0]); acc[1] = fmaf(v0.z, d2.y, acc[1]);
            acc[2] = fmaf(v0.z, d2.z, acc[2]); acc[3] = fmaf(v0.z, d2.w, acc[3]);
        }
    }
}
#define GATHER(idx_, val_, wt_, buf_, acc_) gather_n(idx_, val_, wt_, W, buf_, acc_)
#else
#define GATHER(idx_, val_, wt_, buf_, acc_) gather_ell(idx_, val_, wt_, buf_, acc_)
#endif

// 8 consecutive ROWS of one channel column of a row-major image: two transposing reads (rows +0..3 at p, +4..7 at p + 256)
static __device__ __forceinline__ bf16x8_t read_tr(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 64));
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, r);
}

// six leading terms of (ah + am + al) (bh + bm + bl), smallest first
#define DSW_MFMA6(acc_, ah_, am_, al_, bh_, bm_, bl_)                                   \
    do {                                                                                 \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, bh_, acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, bl_, acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_, bm_, acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_, bh_, acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, bm_, acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, bh_, acc_, 0, 0, 0);         \
    } while (0)

#ifdef DSW_D3_TIMELINE   // diagnostics build: cycle stamps of waves 0 and 4 of workgroup 0 (third sample of its first item) behind the scratch slots
#define DSW_STAMP(i_) do { if (blockIdx.x == 0 && b == b_begin + 2 && orig == blockIdx.x && (tid == 0 || tid == 256)) \
        reinterpret_cast<long long*>(P.pscr + (size_t)gridDim.x * 8192)[(tid >> 8) * 32 + c * 16 + (i_)] = clock64(); } while (0)
#else
#define DSW_STAMP(i_) do { } while (0)
#endif
// NST / NS1: register-stage slots per thread for the gather list (ceil(max_n2 / 64)) and for the one-ring (ceil(max_n1 / 64)).
// Every tile is FULL (64 rows: V % 64 == 0, tiles of consecutive rows) - a condition of eligibility.
// NCH: 32-channel chunks of a dY row (Fout = 32 NCH).  NCH = 1 (32 -> 32 layers): ONE chunk phase per sample - waves 0-3 run the
// whole dX reduction (no partial travels), waves 4-7 the dW products of the same phase.
template <int NST, int NS1, int NCH = 2>
__global__ __launch_bounds__(NTHREADS, 4) void cheb3_bwd_dual_kernel(const DualArgs P) {
    constexpr int YB = RB * NCH;            // bytes of a dY row in HBM
    constexpr int FO = 32 * NCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bufX = lds;                                              // [bufx_rows][128] dY chunk rows of the 2-ring; U_2 image after hop 1
    unsigned char* bufT = bufX + (size_t)P.bufx_rows * RB;                  // [max_n1][128] U_1 on the 1-ring; dX partial handoff after hop 2
    unsigned char* simg = bufT + (size_t)P.max_n1 * RB;                     // split images of U_0, U_1 on the tile rows
    unsigned char* ximg = simg + 2 * IMG_PLANE;                             // split image of the X tile rows (per sample)
    float* ell_val = reinterpret_cast<float*>(ximg + IMG_PLANE);            // [max_n1][W]
    unsigned char* ell_idx = reinterpret_cast<unsigned char*>(ell_val + (size_t)P.max_n1 * P.ell_w);     // [max_n1][W] u8
    int* rows = reinterpret_cast<int*>(ell_idx + (((size_t)P.max_n1 * P.ell_w + 3) & ~(size_t)3));   // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);
    float* dbs = reinterpret_cast<float*>(tile_w + 4);                      // [64] db of this workgroup (LDS accumulator)
    unsigned char* u2img = bufX;

    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t y_sample = (size_t)P.V * YB, x_sample = (size_t)P.V * RB;
    const int grp = tid >> 3;                       // row of a 64-row pass
    const unsigned c4 = (unsigned)(tid & 7);        // 16-byte chunk (4 channels) of this lane inside a staged row
    const unsigned cb = c4 * 16;

    // ---- roles in the matrix phases.  gsel = the dY chunk whose W fragments this wave holds: it runs the dX products in chunk
    // phase gsel and the dW products (of chunk 1 - gsel) in the other one.
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, kc = lane >> 4;
    const int gsel = wave >> 2, jw = wave & 3;
    const int cbk = jw & 1, rb0 = (jw >> 1) * 2;    // dX: 16-channel block of dX, row blocks rb0, rb0 + 1 of the tile
    const int fb = jw & 1, ob = jw >> 1;            // dW: 16-channel block of X, 16-channel block of the dY chunk
    bf16x8_t wh[3], wm[3], wl[3];                   // A fragments: lane holds W[f = 16 cbk + l15][s][o = 32 gsel + 8 kc + j]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float f[8];
        const float* src = P.W + ((size_t)(16 * cbk + l15) * 3 + s) * FO + 32 * (gsel < NCH ? gsel : 0) + 8 * kc;   // (NCH = 1: waves 4-7 never use theirs)
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        split3x8(f, wh[s], wm[s], wl[s]);
    }
    // (the per-lane offsets of the matrix phase - fragment rows, transposing-read addresses, handoff slot - are recomputed
    // there from an opaque copy of the thread id: as loop invariants they would sit in ~12 registers through the gather
    // phases, which have none to spare - the compiler spilled them and reloaded them with s_waitcnt vmcnt(0))
    f32x4_t dwa[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) dwa[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (tid < 64) dbs[tid] = 0.f;                    // (published by the first barrier of the item loop)

#ifdef DSW_D3_SKEW
    // A/B builds: the two workgroups of a CU start in step and stay in step - both in their LDS-bound hop phases, then both in
    // the matrix phase; one of them starts DSW_D3_SKEW cycles late so that one's hops run under the other's matrix phase.
    // DSW_D3_SKEW_MODE 0: the upper half of the grid (the second workgroup of every CU if the dispatcher fills CU slots in order),
    // 1: odd (blockIdx.x >> 3), 2: odd blockIdx.x >> 4
    {
#ifndef DSW_D3_SKEW_MODE
#define DSW_D3_SKEW_MODE 0
#endif
        const bool late = DSW_D3_SKEW_MODE == 0 ? (blockIdx.x >= gridDim.x / 2)
                          : DSW_D3_SKEW_MODE == 1 ? (((blockIdx.x >> 3) & 1u) != 0u) : (((blockIdx.x >> 4) & 1u) != 0u);
        if (late)
            for (int i = 0; i < DSW_D3_SKEW / 1024; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const long n_items = (long)P.n_tiles * P.n_chunks;
    const long q8 = n_items >> 3, r8 = n_items & 7;
    for (long orig = blockIdx.x; orig < n_items; orig += gridDim.x) {
        // XCD-aware order (see dsw_spmm2.hip): gridDim.x is a multiple of 8, so a workgroup stays on its XCD's contiguous range
        const long xcd = orig & 7;
        const long wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        const int tile = (int)(wg / P.n_chunks);
        const int chunk = (int)(wg - (long)tile * P.n_chunks);
        const int b_begin = chunk * P.spc;
        const int b_end = min(P.B, b_begin + P.spc);
        const int* meta = P.tile_meta + (size_t)tile * 6;
        const int s2_off = meta[0], n1 = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];

        __syncthreads();   // the previous item is over (row list, ELL, handoff buffer)
        int* lrp = reinterpret_cast<int*>(bufT);
        if (tid == 0) *tile_w = 2;
        for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
        for (int i = tid; i <= n1; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
        __syncthreads();

        const unsigned tile_off = (unsigned)rows[grp] * (unsigned)RB + cb;     // this thread's tile row in X / dX, sample-relative
        auto offU = [&](const int k) __attribute__((always_inline)) {         // ... and its gather-list rows in dY (index-clamped)
            return (unsigned)rows[min(grp + k * RPP, n2 - 1)] * (unsigned)YB + cb;
        };
        u32x4 su[NST];
        u32x4 xq = {0u, 0u, 0u, 0u};
        if (b_begin < b_end) {
#pragma unroll
            for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.dY + (size_t)b_begin * y_sample + offU(k));
            xq = *reinterpret_cast<const u32x4*>(P.X + (size_t)b_begin * x_sample + tile_off);
        }
        const int tile_nnz = lrp[n1];
        for (int t = tid; t < n1 * W; t += NTHREADS) {
            const int i = t / W, j = t - i * W;
            const int p0 = lrp[i], p1 = lrp[i + 1];
            unsigned col = 0;
            float val = 0.f;
            if (tile_nnz > 0) {
                const int p = max(0, min(p0 + j, tile_nnz - 1));
                col = P.lcol[nnz_off + p];
                val = P.lval[nnz_off + p];
            }
            if (j == 0 && p1 - p0 > 2) atomicMax(tile_w, p1 - p0);
            const bool live = p0 + j < p1;
            ell_idx[t] = (unsigned char)(live ? col : (unsigned)i);
            ell_val[t] = live ? val : 0.f;
        }
        __syncthreads();   // ELL complete (and lrp in bufT dead)
        const int Wt = *tile_w;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * RB + cb) = su[k];
        }

        for (int b = b_begin; b < b_end; ++b) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                DSW_STAMP(0);
                __syncthreads();   // A: the rows of (b, c) are complete in bufX
                DSW_STAMP(1);
                {   // next chunk's rows (and, in chunk phase 1, the next sample's X tile row): in flight under phases 1, 2 and 3
                    const int bn = c + 1 < NCH ? b : (b + 1 < b_end ? b + 1 : b);
                    const char* src = P.dY + (size_t)bn * y_sample + (c + 1 < NCH ? (c + 1) * RB : 0);
#pragma unroll
                    for (int k = 0; k < NST; ++k)
#ifdef DSW_ABL_D3_NOLOAD
                        su[k] = u32x4{(unsigned)bn, offU(k), 0u, 0u};
#else
                        su[k] = *reinterpret_cast<const u32x4*>(src + offU(k));
#endif
                }
                // ---- phase 1: U_1 = L^T U_0 on the one-ring; the tile rows (slot 0) also leave the split images of U_0 and U_1
                float u0[4];
#pragma unroll 1
                for (int k = 0; k < NS1; ++k) {
                    const int i = grp + k * RPP;
                    if (k == 0 || i < n1) {
                        float acc[4] = {0.f, 0.f, 0.f, 0.f};
                        GATHER(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, bufX + cb, acc);
                        *reinterpret_cast<uint4*>(bufT + (size_t)i * RB + cb) =
                            make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
                        if (k == 0) {
                            split_store(simg + IMG_PLANE, i, c4, acc);
                            const float4 xr = *reinterpret_cast<const float4*>(bufX + (size_t)i * RB + cb);
                            u0[0] = xr.x; u0[1] = xr.y; u0[2] = xr.z; u0[3] = xr.w;
                            split_store(simg, i, c4, u0);
                        }
                    }
                }
                DSW_STAMP(2);
                __syncthreads();   // B: U_1 complete; bufX is dead (the U_2 image takes its place)
                DSW_STAMP(3);
                // ---- phase 2: U_2 = 2 L^T U_1 - U_0 on the tile rows -> split image; chunk phase 0: the X tile rows -> split image
                {
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    GATHER(ell_idx + (size_t)grp * W, ell_val + (size_t)grp * W, Wt, bufT + cb, acc);
                    const float t2[4] = {fmaf(2.f, acc[0], -u0[0]), fmaf(2.f, acc[1], -u0[1]), fmaf(2.f, acc[2], -u0[2]), fmaf(2.f, acc[3], -u0[3])};
                    split_store(u2img, grp, c4, t2);
                    if (c == 0) {
                        const float xf[4] = {__uint_as_float(xq[0]), __uint_as_float(xq[1]), __uint_as_float(xq[2]), __uint_as_float(xq[3])};
                        split_store(ximg, grp, c4, xf);
                    }
                    if (c == NCH - 1) {
                        const int bn = b + 1 < b_end ? b + 1 : b;
                        xq = *reinterpret_cast<const u32x4*>(P.X + (size_t)bn * x_sample + ((unsigned)rows[grp] * (unsigned)RB + cb));
                    }
                }
                DSW_STAMP(4);
                __syncthreads();   // C: images complete; nobody reads bufT of this chunk any more
                DSW_STAMP(5);
                // ---- phase 3: matrix cores
                int tv = tid;
                asm volatile("" : "+v"(tv));          // opaque: everything below is recomputed here, not kept across the phases
                const unsigned lane_ = (unsigned)tv & 63u, l15_ = lane_ & 15u, kc_ = lane_ >> 4, jw_ = ((unsigned)tv >> 6) & 3u;
#ifdef DSW_ABL_D3_NODX
                if (gsel == c) { } else
#endif
#ifdef DSW_ABL_D3_NODW
                if (gsel != c) { } else
#endif
                if (gsel == c) {
                    // dX partial of this chunk: rows 16 (rb0 + r) + l15, dX channels 16 cbk + 4 kc .. + 3
                    // The reduction over (k, o) is cut between the wave groups: the chunk-0 partial travels from waves 0-3 to
                    // waves 4-7 through a private 8 KB slot in the workspace (rewritten every sample: it lives in L2; same
                    // workgroup, three barriers apart), requested here and added behind the products.
                    char* slot = P.pscr + (size_t)blockIdx.x * 8192 + ((jw_ * 2u) * 64u + lane_) * 16u;
                    f32x4_t h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
                    if (NCH == 2 && c == 1) {
                        h0 = *reinterpret_cast<const f32x4_t*>(slot);
                        h1 = *reinterpret_cast<const f32x4_t*>(slot + 1024);
                    }
                    f32x4_t pacc[2];
                    pacc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    pacc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    unsigned fro[2];                // B fragment (row 16 (rb0 + r) + l15, chunk kc) inside a term of an image
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const unsigned row = 16u * ((jw_ >> 1) * 2u + r) + l15_;
                        fro[r] = row * 64u + ((kc_ ^ ((row >> 2) & 2u)) << 4);
                    }
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const unsigned char* pb = s < 2 ? simg + (size_t)s * IMG_PLANE : u2img;
                        bf16x8_t th[2], tm[2], tl[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            th[r] = *reinterpret_cast<const bf16x8_t*>(pb + fro[r]);
                            tm[r] = *reinterpret_cast<const bf16x8_t*>(pb + IMG_TERM + fro[r]);
                            tl[r] = *reinterpret_cast<const bf16x8_t*>(pb + 2 * IMG_TERM + fro[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], th[r], pacc[r], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tl[r], pacc[r], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], tm[r], pacc[r], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], th[r], pacc[r], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tm[r], pacc[r], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r) pacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], th[r], pacc[r], 0, 0, 0);
                    }
                    if (NCH == 2 && c == 0) {
                        *reinterpret_cast<f32x4_t*>(slot) = pacc[0];
                        *reinterpret_cast<f32x4_t*>(slot + 1024) = pacc[1];
                    } else if (P.dX != nullptr) {
                        char* dst = P.dX + (size_t)b * x_sample;                  // uniform base + 32-bit lane offsets
                        const unsigned col = (16u * (jw_ & 1u) + 4u * kc_) * 4u, r0 = 32u * (jw_ >> 1) + l15_;
                        st16_nt(dst + ((unsigned)rows[r0] * (unsigned)RB + col), pacc[0] + h0);
                        st16_nt(dst + ((unsigned)rows[r0 + 16] * (unsigned)RB + col), pacc[1] + h1);
                    }
                } else {
                    // dW_s[f = 16 fb + l15][o = 32 c + 16 ob + 4 kc .. + 3] += sum over the 64 tile rows of U_s[row][o] X[row][f]:
                    // A = U_s^T (m = o), B = X^T (n = f), two k-steps of 32 rows; db: B = ones (the wave with fb == k-step)
                    // transposing reads: a 16-lane group kc addresses rows 8 kc + (i >> 2) (+ 4: second read; + 32: second k-step)
                    // and the 8-byte pieces (i & 3) of the 16-channel block; it receives column i = l15 of those rows
                    const unsigned trrow = (8u * kc_ + (l15_ >> 2)) * 64u + ((l15_ & 1u) << 3);
                    const unsigned trkey = 2u * (kc_ & 1u), trhalf = (l15_ & 3u) >> 1;
                    const unsigned xoff = trrow + (((2u * (jw_ & 1u) + trhalf) ^ trkey) << 4);
                    const unsigned uoff = trrow + (((2u * (jw_ >> 1) + trhalf) ^ trkey) << 4);
                    const uint4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
                    f32x4_t dba = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const unsigned char* xp = ximg + xoff + 2048 * ks;
                        const bf16x8_t xh = read_tr(xp), xm = read_tr(xp + IMG_TERM), xl = read_tr(xp + 2 * IMG_TERM);
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            const unsigned char* up = (s < 2 ? simg + (size_t)s * IMG_PLANE : u2img) + uoff + 2048 * ks;
                            const bf16x8_t uh = read_tr(up), um = read_tr(up + IMG_TERM), ul = read_tr(up + 2 * IMG_TERM);
                            DSW_MFMA6(dwa[s], uh, um, ul, xh, xm, xl);
                            if (s == 0 && (jw_ & 1u) == 0) {
                                dba = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ul, ones, dba, 0, 0, 0);
                                dba = __builtin_amdgcn_mfma_f32_16x16x32_bf16(um, ones, dba, 0, 0, 0);
                                dba = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uh, ones, dba, 0, 0, 0);
                            }
                        }
                    }
                    // db of this (tile, sample, chunk): one wave per 16-channel block owns these 16 sums for the whole launch
                    // (fixed order of additions: deterministic)
                    if ((jw_ & 1u) == 0 && l15_ == 0) {
                        float* d = dbs + 32 * c + 16 * (jw_ >> 1) + 4 * kc_;
                        const f32x4_t o = *reinterpret_cast<const f32x4_t*>(d);
                        *reinterpret_cast<f32x4_t*>(d) = o + dba;
                    }
                }
                DSW_STAMP(6);
#ifndef DSW_ABL_D3_NOD
                __syncthreads();   // D: everybody has left the matrix phase: images, bufX, handoff slot readable / writable
#endif
                DSW_STAMP(7);
                {   // next chunk's rows -> the (single) input buffer; the next barrier A publishes them
                    const unsigned sto = (unsigned)tv * 16u;          // = grp * RB + cb
#pragma unroll
                    for (int k = 0; k < NST; ++k)
                        if ((int)((unsigned)tv >> 3) + k * RPP < n2) *reinterpret_cast<u32x4*>(bufX + sto + (unsigned)(k * RPP * RB)) = su[k];
                }
            }
        }
    }

    // ---- the workgroup's slab of dW / db partials: [(s, f)][o] rows, then the db row
    if (P.partial != nullptr) {
        float* slab = P.partial + (size_t)blockIdx.x * ((3 * 32 + 1) * FO);
        const int cdw = 1 - gsel;               // the chunk whose dW this wave accumulated (NCH = 1: waves 0-3 none)
        if (cdw < NCH) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                *reinterpret_cast<f32x4_t*>(slab + (size_t)(s * 32 + 16 * fb + l15) * FO + 32 * cdw + 16 * ob + 4 * kc) = dwa[s];
        }
        __syncthreads();
        if (tid < FO) slab[(size_t)96 * FO + tid] = dbs[tid];
    }
}

size_t dual_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    const int bufx_rows = plan->max_n2 > 96 ? plan->max_n2 : 96;           // the U_2 image (12 KB) lives there after hop 1
    size_t s = (size_t)(bufx_rows + (size_t)plan->max_n1) * RB + 3 * IMG_PLANE;
    s += (size_t)plan->max_n1 * ell_w * 4 + (((size_t)plan->max_n1 * ell_w + 3) & ~(size_t)3);   // fp32 values + u8 positions
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16 + 64 * 4;
    return (s + 15) & ~(size_t)15;
}

template <int NST, int NS1, int NCH>
int launch_dual(const DualArgs& A, long nwg, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)cheb3_bwd_dual_kernel<NST, NS1, NCH>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DSW_ERR_LAUNCH;
    DSW_LAUNCH((cheb3_bwd_dual_kernel<NST, NS1, NCH>), dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A);
    return dsw_check_launch();
}
template <int NCH>
int launch_dual_n(const int nst, const int ns1, const DualArgs& A, long nwg, size_t lds, hipStream_t stream) {
    if (nst == 3 && ns1 == 2) return launch_dual<3, 2, NCH>(A, nwg, lds, stream);
    if (nst == 2 && ns1 <= 2) return launch_dual<2, 2, NCH>(A, nwg, lds, stream);
    if (nst == 3) return launch_dual<3, 3, NCH>(A, nwg, lds, stream);
    return launch_dual<4, 4, NCH>(A, nwg, lds, stream);
}

// persistent workgroups: two per CU, a multiple of 8 (one XCD per residue class), never more than there are items
long dual_grid(long n_items) {
#ifdef DSW_D3_WG_PER_CU     // A/B builds: how much of a workgroup's time is waiting that a second workgroup fills
    long g = DSW_D3_WG_PER_CU * dsw_device_cus();
#else
    long g = 2 * dsw_device_cus();
#endif
    g -= g % 8;
    if (g < 8) g = 8;
    return n_items < g ? n_items : g;
}

}  // namespace

// 1 if the one-launch dual backward exists for this layer shape and plan of L^T (pointer alignment aside)
int dsw_cheb3_bwd_dual_eligible(const dsw_hop2_plan* plan_t, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    static const char* env = dsw_diag_env("DSW_BWD_DUAL");   // "0": separate dgrad / wgrad / adjoint launches (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || (Fout != 64 && Fout != 32)) return 0;
#ifdef DSW_D3_NO_FOUT32      // A/B builds: 32 -> 32 layers on the route over the basis planes, as before round 6
    if (Fout == 32) return 0;
#endif
    if (!plan_t || plan_t->hops == 1 || plan_t->tile_rows != 64 || plan_t->explicit_tiles || !dsw_spmm2_supported(plan_t, Fin, dtype)) return 0;
    if (V <= 0 || V % 64 != 0 || (unsigned long long)V * 256ull >= (1ull << 32)) return 0;   // full tiles; 32-bit row offsets inside a sample
    if (plan_t->max_n2 > 255 || plan_t->max_n1 < 64) return 0;     // u8 list positions in the ELL; the handoff slot is 64 rows of the U_1 buffer
    if (dual_lds_bytes(plan_t) > 80 * 1024) return 0;              // two workgroups per CU or not at all
    const int nst = (plan_t->max_n2 + RPP - 1) / RPP, ns1 = (plan_t->max_n1 + RPP - 1) / RPP;
    return (nst > 4 || ns1 > nst) ? 0 : 1;
}

// bytes of scratch of the launch: one slab of dW / db partials and one 8 KB dX-partial slot per persistent workgroup
int64_t dsw_cheb3_bwd_dual_ws_bytes(void) { return (int64_t)(2 * dsw_device_cus() + 8) * (SLAB * 4 + 8192); }

// dX, dW, db from X, dY, W in one launch (+ the partial reduce) if the shape / plan allow it.  Returns 1 if it took the call
// (*rc = status), 0 if the caller must use the generic sequence.  dX or dW may be null (db only together with dW).
int dsw_cheb3_bwd_dual_try(const dsw_hop2_plan* plan_t, int64_t V, const void* X, const void* dY, const void* W, void* dX,
                           void* dW, void* db, float* partial, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                           hipStream_t stream, int* rc, int accumulate) {
    if (!dsw_cheb3_bwd_dual_eligible(plan_t, V, Fin, Fout, K, dtype)) return 0;
    if (!dsw_aligned16(dY) || !dsw_aligned16(X) || !dsw_aligned16(W) || (dX && !dsw_aligned16(dX)) || !dsw_aligned16(partial)) return 0;
    if (B <= 0 || (!dX && !dW)) { *rc = DSW_OK; return 1; }
    DualArgs A;
    A.tile_meta = plan_t->tile_meta; A.s2_rows = plan_t->s2_rows; A.lrowptr = plan_t->lrowptr;
    A.lcol = plan_t->lcol; A.lval = plan_t->lval;
    A.dY = static_cast<const char*>(dY); A.X = static_cast<const char*>(X); A.dX = static_cast<char*>(dX);
    A.W = static_cast<const float*>(W); A.partial = dW ? partial : nullptr;
    A.pscr = reinterpret_cast<char*>(partial) + (size_t)(2 * dsw_device_cus() + 8) * SLAB * 4;
    A.V = (int)V; A.n_tiles = plan_t->n_tiles; A.max_n1 = plan_t->max_n1; A.max_n2 = plan_t->max_n2;
    A.bufx_rows = plan_t->max_n2 > 96 ? plan_t->max_n2 : 96;
    A.B = (int)B; A.ell_w = (plan_t->reserved + 3) & ~3;
    // batch chunks: items = tiles x chunks over the persistent workgroups; rounds x (tile prologue + samples per chunk)
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long items = (long)plan_t->n_tiles * c;
            const long g = dual_grid(items);
            const long rounds = (items + g - 1) / g;
            const double cost = (double)rounds * (1.0 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = dual_grid((long)plan_t->n_tiles * A.n_chunks);
    const size_t lds = dual_lds_bytes(plan_t);
    const int nst = (plan_t->max_n2 + RPP - 1) / RPP, ns1 = (plan_t->max_n1 + RPP - 1) / RPP;
    int r = Fout == 64 ? launch_dual_n<2>(nst, ns1, A, nwg, lds, stream) : launch_dual_n<1>(nst, ns1, A, nwg, lds, stream);
    if (r == DSW_OK && dW != nullptr)
        r = dsw_wgrad_reduce_launch(partial, nwg, Fin, Fout, K, dW, db, K, 0, 1 << 30, dtype, stream, accumulate);
    *rc = r;
    return 1;
}
