// dX of a K = 3, 32-input-channel, 64-output-channel fp32 ConvCheb layer in ONE launch, straight from dY:
//
//     G_k = dY W_k^T  (k = 0, 1, 2),   dX = G_0 - G_2 + L^T (G_1 + 2 L^T G_2)          (autograd of layers.py:163-178)
//
// The separate route (dsw_wgrad_x3.hip FUSE + the adjoint pair of dsw_spmm2.hip) writes the three dgrad planes G_k to HBM
// and reads them back: 604 MB of the 1.86 GB the north-star step moves.  Here a workgroup owns (tile of 64 rows) x (the
// batch): per sample it streams the dY rows of the tile's TWO-RING (tile rows, 1-ring, 2-ring: the gather list of the
// two-hop plan of L^T) in chunks of 64 rows, forms G_2 on the 2-ring, G_1 on the 1-ring and G_0' = dY (W_0 - W_2)^T on the
// tile with the matrix cores - the planes live in LDS only - and then runs the two L^T hops from LDS exactly as the
// two-hop kernel does.  HBM sees dY in (with the halo re-reads, mostly L2 hits) and dX out.  The dgrad flops on the rings
// are redundant (2.6x the tile's), which the matrix pipe has room for: the kernel is priced by its HBM bytes.
//
// Matrix part (as dsw_fwd3.hip): fp32 product on the bf16 pipe by exact 3-way operand splitting, six MFMA terms,
// v_mfma_f32_16x16x32_bf16 in the orientation  G^T = W (dY)^T : A = W_k fragment (16 dX channels x 32 dY channels), B = 16
// rows x 32 dY channels, D = 4 consecutive dX channels of one row per lane.  Both operands come from LDS images with
// 128-byte rows ([term][row][64 bf16]); the 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7): a fragment read (16
// rows x one chunk column per ds_read_b128 lane group, two chunk columns per group) touches every bank once.  The split
// image of W (3 planes x 3 terms x 32 rows: 36 KB) is built once per workgroup; every dY row is split once, by the
// thread that loaded it.
//
// One workgroup of 16 waves per CU (147 KB of LDS at nside 64), SPECIALISED: waves 0-7 are the matrix waves, waves 8-15 the
// hop waves, and they work on DIFFERENT samples - while the hop waves run the two L^T hops of sample b out of one set of G
// buffers, the matrix waves fill the other set for sample b + 1.  A matrix wave owns two row blocks (16 list rows each) of the
// tile's two-ring for the whole batch: it loads the dY rows of a block straight into the B-fragment layout of the MFMA (lane
// (n, kg): 8 consecutive dY channels of row n), splits them in registers and multiplies them with the W fragments of the planes
// the rows need, read from a split image of the weights in LDS, both dX channel blocks interleaved (two accumulation chains).
// Two workgroup barriers per sample (between the hops / at the end), which the matrix waves pass between their two row blocks.
//
// How it got here: one 1024-thread workgroup staging dY through an LDS image in 64-row chunks, all waves in
// lockstep through five barriers per sample: 216 us; two 512-thread workgroups per CU with the W fragments in registers: 203 us
// (seven barriers, phases that do not overlap: the ablation's parts add up); no dY image, a row block per wave, three barriers:
// 168 us; the specialised form: see DESIGN.md.  The launches it replaces cost 157 us.
#include <cstdlib>
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);

namespace {
template <typename T4>
static __device__ __forceinline__ void st16_nt(char* p, const T4& v) {
    typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(p));
}

constexpr int XB = 128;          // bytes of a dX / G row in HBM (32 fp32 channels)
constexpr int YB = 256;          // bytes of a dY row (64 fp32 channels)
constexpr int GS = 144;          // LDS stride of a G row: 128 + 16 (the 16 rows of an accumulator store hit distinct banks)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

static __device__ __forceinline__ float trunc_bf16(float f) { return __uint_as_float(__float_as_uint(f) & 0xffff0000u); }
static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// byte offset of 16-byte chunk c of row r inside a [rows][128 B] image
static __device__ __forceinline__ unsigned img_off(const unsigned r, const unsigned c) { return r * 128u + ((c ^ ((r >> 1) & 7u)) << 4); }

// the three bf16 images of 4 consecutive channels (quad q of the row's 16) of one image row -> LDS (8 bytes per term)
static __device__ __forceinline__ void split_store4(unsigned char* __restrict__ img, const int term_stride, const unsigned row,
                                                    const unsigned q, const float (&f)[4]) {
#ifdef DSW_ABL_B3_NOSPLIT       // ablation builds only (refused by _native.load unless named by DSW_HIP_LIB)
    if (q < 64) return;
#endif
    float r1[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r1[j] = f[j] - trunc_bf16(f[j]);
        r2[j] = r1[j] - trunc_bf16(r1[j]);
    }
    unsigned char* base = img + img_off(row, q >> 1) + ((q & 1u) << 3);
    *reinterpret_cast<uint2*>(base) = make_uint2(pack2(f[0], f[1]), pack2(f[2], f[3]));
    *reinterpret_cast<uint2*>(base + term_stride) = make_uint2(pack2(r1[0], r1[1]), pack2(r1[2], r1[3]));
    *reinterpret_cast<uint2*>(base + 2 * term_stride) = make_uint2(pack2(r2[0], r2[1]), pack2(r2[2], r2[3]));
}

// acc += sum_j val[j] * buf[pos[j]] over the first W entries of one ELL row (fp32 values + u8 list positions, padded with
// {own row, 0}); bufc = G buffer + this lane's byte offset in a row; rows GS bytes apart
template <unsigned GS>
static __device__ __forceinline__ void gather_ell(const unsigned char* __restrict__ row_idx, const float* __restrict__ row_val,
                                                  const int W, const unsigned char* __restrict__ bufc, float (&acc)[4]) {
#ifdef DSW_ABL_B3_NOGATHER
    acc[0] = row_val[0]; return;
#endif
    int j = 0;
    for (; j + 4 <= W; j += 4) {
        const unsigned w = *reinterpret_cast<const unsigned*>(row_idx + j);
        const float4 v0 = *reinterpret_cast<const float4*>(row_val + j);
        const unsigned ix[4] = {(w & 0xffu) * GS, ((w >> 8) & 0xffu) * GS, ((w >> 16) & 0xffu) * GS, (w >> 24) * GS};
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
    for (; j < W; ++j) {
        const unsigned w = row_idx[j];
        const float v0 = row_val[j];
        const float4 d0 = *reinterpret_cast<const float4*>(bufc + w * GS);
        acc[0] = fmaf(v0, d0.x, acc[0]); acc[1] = fmaf(v0, d0.y, acc[1]);
        acc[2] = fmaf(v0, d0.z, acc[2]); acc[3] = fmaf(v0, d0.w, acc[3]);
    }
}

// six leading terms of the split product (smallest first), for TWO products that share the B operand (the two dX channel blocks of a row block), interleaved: two
// independent accumulation chains keep the matrix pipe issuing while a result is still in flight
static __device__ __forceinline__ void mfma6x2(const bf16x8_t (&a0)[3], const bf16x8_t (&a1)[3], const bf16x8_t (&b)[3],
                                               f32x4_t& c0, f32x4_t& c1) {
#ifdef DSW_ABL_B3_NOMFMA
    c0[0] += __builtin_bit_cast(f32x4_t, a0[0])[0] + __builtin_bit_cast(f32x4_t, b[0])[0];
    c1[0] += __builtin_bit_cast(f32x4_t, a1[0])[0]; return;
#endif
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[2], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[2], b[0], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0], b[2], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], b[2], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[1], b[1], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1], b[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[1], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1], b[0], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0], b[1], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], b[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], b[0], c1, 0, 0, 0);
}

constexpr int NTH3 = 1024;
constexpr int W3_PLANE = 3 * 32 * 128;       // [term][32 f][64 o bf16], 16-byte chunks swizzled by the row (img_off)
constexpr int W3_BYTES = 3 * W3_PLANE;       // 36 KB

struct Bwd3Args {
    const int* tile_meta;
    const int* s2_rows;
    const int* lrowptr;
    const unsigned short* lcol;
    const float* lval;
    const char* dY;
    char* dX;
    const float* W;      // [32][3][64]
    int V, n_tiles, max_n1, max_n2;
    int B, n_chunks, spc, ell_w;
    int explicit_tiles;
};

static __device__ __forceinline__ void split8(const u32x4 lo, const u32x4 hi, bf16x8_t (&t)[3]) {
    const float f[8] = {__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(lo[2]), __uint_as_float(lo[3]),
                        __uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(hi[2]), __uint_as_float(hi[3])};
#ifdef DSW_ABL_B3_NOSPLIT
    const uint4 u0 = {pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
    t[0] = t[1] = t[2] = __builtin_bit_cast(bf16x8_t, u0);
    return;
#endif
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r1[j] = f[j] - trunc_bf16(f[j]);
        r2[j] = r1[j] - trunc_bf16(r1[j]);
    }
    const uint4 uh = {pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
    const uint4 um = {pack2(r1[0], r1[1]), pack2(r1[2], r1[3]), pack2(r1[4], r1[5]), pack2(r1[6], r1[7])};
    const uint4 ul = {pack2(r2[0], r2[1]), pack2(r2[2], r2[3]), pack2(r2[4], r2[5]), pack2(r2[6], r2[7])};
    t[0] = __builtin_bit_cast(bf16x8_t, uh); t[1] = __builtin_bit_cast(bf16x8_t, um); t[2] = __builtin_bit_cast(bf16x8_t, ul);
}

__global__ __launch_bounds__(NTH3, 4) void cheb3_bwd_fused_kernel(const Bwd3Args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wimg = lds;                                              // [3 planes][3 terms][32 f][128 B]
    const size_t gset = ((size_t)P.max_n2 + P.max_n1 + 64) * GS;           // one set of G buffers: G_2 | G_1 (H_1) | G_0'
    unsigned char* gbuf = wimg + W3_BYTES;                                  // [2 sets]
    float* ell_val = reinterpret_cast<float*>(gbuf + 2 * gset);             // [max_n1][W]
    unsigned char* ell_idx = reinterpret_cast<unsigned char*>(ell_val + (size_t)P.max_n1 * P.ell_w);   // [max_n1][W] u8
    int* rows = reinterpret_cast<int*>(ell_idx + (((size_t)P.max_n1 * P.ell_w + 3) & ~(size_t)3));   // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);

    const long nwg = gridDim.x, orig = blockIdx.x;                          // XCD-aware order (see dsw_spmm2.hip)
    const long q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], n1 = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];
    const int rt = P.explicit_tiles ? meta[5] : min(64, P.V - tile * 64);   // tile rows = the first rt list entries
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t y_sample = (size_t)P.V * YB, x_sample = (size_t)P.V * XB;

    int* lrp = reinterpret_cast<int*>(gbuf);         // local row pointers, parked in the G buffers until the ELL is built
    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTH3) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= n1; i += NTH3) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const bool matrix = wave < 8;                    // uniform per wave: waves 0-7 form the G planes, waves 8-15 run the hops
    // ---- matrix role: row blocks wave and wave + 8 (list rows 16 rb .. 16 rb + 15); lane (n = l & 15, kg = l >> 4)
    const unsigned l15 = (unsigned)(lane & 15), kc = (unsigned)(lane >> 4);
    const int pA = 16 * (wave & 7), pB = pA + 128;
    const bool hasA = matrix && pA < n2, hasB = matrix && pB < n2;
    const unsigned yoffA = (unsigned)rows[min(pA + (int)l15, n2 - 1)] * (unsigned)YB + kc * 32u;   // (+ 128: the second k-step)
    const unsigned yoffB = (unsigned)rows[min(pB + (int)l15, n2 - 1)] * (unsigned)YB + kc * 32u;
    u32x4 pyA[4], pyB[4];
    auto load4 = [&](const int b, const unsigned yoff, u32x4 (&py)[4]) __attribute__((always_inline)) {
#ifdef DSW_ABL_B3_NOLOAD
        py[0] = py[1] = py[2] = py[3] = u32x4{(unsigned)b, yoff, 0u, 0u};
#else
        const char* src = P.dY + (size_t)b * y_sample + yoff;
        py[0] = *reinterpret_cast<const u32x4*>(src); py[1] = *reinterpret_cast<const u32x4*>(src + 16);
        py[2] = *reinterpret_cast<const u32x4*>(src + 128); py[3] = *reinterpret_cast<const u32x4*>(src + 144);
#endif
    };
    if (b_begin < b_end) {
        if (hasA) load4(b_begin, yoffA, pyA);
        if (hasB) load4(b_begin, yoffB, pyB);
    }
    // CSR -> ELL of the tile + 1-ring rows (as dsw_fwd3.hip)
    const int tile_nnz = lrp[n1];
    for (int t = tid; t < n1 * W; t += NTH3) {
        const int i = t / W, j = t - i * W;
        const int q0 = lrp[i], q1 = lrp[i + 1];
        unsigned col = 0;
        float val = 0.f;
        if (tile_nnz > 0) {
            const int p = max(0, min(q0 + j, tile_nnz - 1));
            col = P.lcol[nnz_off + p];
            val = P.lval[nnz_off + p];
        }
        if (j == 0 && q1 - q0 > 2) atomicMax(tile_w, q1 - q0);
        const bool live = q0 + j < q1;
        ell_idx[t] = (unsigned char)(live ? col : (unsigned)i);
        ell_val[t] = live ? val : 0.f;
    }
    // split image of the weights: plane 0 = W_0 - W_2 (the subtraction of the raw plane K-1 at the top of the adjoint
    // recurrence, folded into the weights), planes 1, 2 = W_1, W_2; image row = dX channel f, columns = dY channels o
    for (int e = tid; e < 3 * 32 * 16; e += NTH3) {
        const int p = e / (32 * 16), f = (e / 16) & 31;
        const unsigned q = (unsigned)(e & 15);
        const float4 w = *reinterpret_cast<const float4*>(P.W + ((size_t)f * 3 + p) * 64 + 4 * q);
        float v[4] = {w.x, w.y, w.z, w.w};
        if (p == 0) {
            const float4 w2 = *reinterpret_cast<const float4*>(P.W + ((size_t)f * 3 + 2) * 64 + 4 * q);
            v[0] -= w2.x; v[1] -= w2.y; v[2] -= w2.z; v[3] -= w2.w;
        }
        split_store4(wimg + (size_t)p * W3_PLANE, 32 * 128, (unsigned)f, q, v);
    }
    __syncthreads();   // ELL and weight image complete (lrp in the G buffers dead)
    const int Wt = *tile_w;

    // A-fragment offsets in a plane of the weight image: dX channel blocks fb = 0 / 1, k-steps 0 / 1
    const unsigned a00 = img_off(l15, kc), a01 = img_off(l15, 4u + kc), a10 = img_off(16u + l15, kc), a11 = img_off(16u + l15, 4u + kc);
    const unsigned gcol = 4u * kc * 4u;              // 16 bytes of a G row: dX channels 16 fb + 4 kc .. + 3

    // G planes of one row block (first list row p0) of the sample held in py -> set gs; py then takes sample b_next
    auto form = [&](const int p0, const unsigned yoff, u32x4 (&py)[4], unsigned char* gs, const int b_next) __attribute__((always_inline)) {
        bf16x8_t b0[3], b1[3];
        split8(py[0], py[1], b0);
        split8(py[2], py[3], b1);
        load4(b_next, yoff, py);                     // in flight under the matrix work and the hops of a whole sample
        unsigned char* s2 = gs;
        unsigned char* s1 = gs + (size_t)P.max_n2 * GS;
        unsigned char* s0 = s1 + (size_t)P.max_n1 * GS;
        const unsigned grow_ = (unsigned)(p0 + (int)l15);
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) {             // plane 2 (two-ring), plane 1 (one-ring), plane 0' (tile)
            const int plane = 2 - pi;
            const bool need = plane == 2 ? true : plane == 1 ? p0 < n1 : p0 < rt;
            if (need) {
                const unsigned char* wp = wimg + (size_t)plane * W3_PLANE;
                f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                {
                    bf16x8_t a0[3], a1[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        a0[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a00);
                        a1[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a10);
                    }
                    mfma6x2(a0, a1, b0, c0, c1);
                }
                {
                    bf16x8_t a0[3], a1[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        a0[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a01);
                        a1[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a11);
                    }
                    mfma6x2(a0, a1, b1, c0, c1);
                }
                unsigned char* gd = (plane == 2 ? s2 : plane == 1 ? s1 : s0) + grow_ * GS + gcol;
                if ((int)grow_ < (plane == 2 ? n2 : plane == 1 ? n1 : 64)) {
                    *reinterpret_cast<f32x4_t*>(gd) = c0;
                    *reinterpret_cast<f32x4_t*>(gd + 64) = c1;
                }
            }
        }
    };

    // ---- hop role (waves 8-15): list positions hg and hg + 64, 16-byte chunk of the 128-byte row
    const int hg = (tid & 511) >> 3;
    const unsigned gcb = (unsigned)(tid & 7) * 16u;

    // pipeline prologue: the G planes of the first sample
    if (b_begin < b_end) {
        const int bn = b_begin + 1 < b_end ? b_begin + 1 : b_begin;
        if (hasA) form(pA, yoffA, pyA, gbuf, bn);
        if (hasB) form(pB, yoffB, pyB, gbuf, bn);
    }
    __syncthreads();
    for (int b = b_begin; b < b_end; ++b) {
        unsigned char* gs = gbuf + (size_t)((b - b_begin) & 1) * gset;          // set of sample b (read by the hops)
        unsigned char* gn = gbuf + (size_t)((b - b_begin + 1) & 1) * gset;      // set of sample b + 1 (written by the matrix waves)
        const bool more = b + 1 < b_end;
        const int bn = b + 2 < b_end ? b + 2 : b_end - 1;
        if (matrix) {
            if (more && hasA) form(pA, yoffA, pyA, gn, bn);
        } else {
            // ---- hop 1: H_1 = G_1 + 2 L^T G_2 on the tile + 1-ring rows, in place
            unsigned char* s2 = gs;
            unsigned char* s1 = gs + (size_t)P.max_n2 * GS;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = hg + 64 * k;
                if (i < n1) {
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    gather_ell<GS>(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, s2 + gcb, acc);
                    float4* hp = reinterpret_cast<float4*>(s1 + (size_t)i * GS + gcb);
                    const float4 g = *hp;
                    *hp = make_float4(fmaf(2.f, acc[0], g.x), fmaf(2.f, acc[1], g.y), fmaf(2.f, acc[2], g.z), fmaf(2.f, acc[3], g.w));
                }
            }
        }
        __syncthreads();   // H_1 of sample b complete (the matrix waves pass here between their two row blocks)
        if (matrix) {
            if (more && hasB) form(pB, yoffB, pyB, gn, bn);
        } else {
            // ---- hop 2: dX = G_0' + L^T H_1 on the tile rows -> HBM
            unsigned char* s1 = gs + (size_t)P.max_n2 * GS;
            unsigned char* s0 = s1 + (size_t)P.max_n1 * GS;
            if (hg < rt) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                gather_ell<GS>(ell_idx + (size_t)hg * W, ell_val + (size_t)hg * W, Wt, s1 + gcb, acc);
                const float4 g = *reinterpret_cast<const float4*>(s0 + (size_t)hg * GS + gcb);
                const float4 o = make_float4(g.x + acc[0], g.y + acc[1], g.z + acc[2], g.w + acc[3]);
                st16_nt(P.dX + (size_t)b * x_sample + (size_t)rows[hg] * XB + gcb, o);
            }
        }
        __syncthreads();   // set of sample b free, set of sample b + 1 complete
    }
}

size_t bwd3_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    size_t s = (size_t)W3_BYTES + 2 * ((size_t)plan->max_n2 + plan->max_n1 + 64) * GS;
    s += (size_t)plan->max_n1 * ell_w * 4 + (((size_t)plan->max_n1 * ell_w + 3) & ~(size_t)3);   // fp32 values + u8 positions
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;
    return (s + 15) & ~(size_t)15;
}

}  // namespace

// 1 if the one-launch dgrad + adjoint exists for this layer shape and plan of L^T (pointer alignment aside)
int dsw_cheb3_bwd_fused_eligible(const dsw_hop2_plan* plan_t, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    static const char* env = dsw_diag_env("DSW_BWD3_FUSED");   // "0": separate dgrad planes + adjoint pair (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || Fout != 64) return 0;
    if (!plan_t || plan_t->hops == 1 || plan_t->tile_rows != 64 || !dsw_spmm2_supported(plan_t, Fin, dtype)) return 0;
    if (plan_t->max_n2 > 255 || plan_t->max_n1 > 128) return 0;     // u8 list positions, 16 row blocks on 8 matrix waves; hop 1 in two passes
    if (bwd3_lds_bytes(plan_t) > 160 * 1024) return 0;
    return 1;
}

// dX from dY in one launch if the shape / plan allow it.  Returns 1 if it took the call (*rc = status), 0 if the caller
// must use the generic sequence (dgrad planes + adjoint recurrence).  (ws: unused since the weights image moved to LDS.)
int dsw_cheb3_bwd_fused_try(const dsw_hop2_plan* plan_t, int64_t V, const void* dY, const void* W, void* dX, int64_t B,
                            int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int* rc, void* ws,
                            int64_t ws_bytes) {
    (void)ws; (void)ws_bytes;
    if (!dsw_cheb3_bwd_fused_eligible(plan_t, Fin, Fout, K, dtype)) return 0;
    if (!dsw_aligned16(dY) || !dsw_aligned16(dX) || !dsw_aligned16(W)) return 0;
    if ((unsigned long long)V * YB >= (1ull << 32)) return 0;           // 32-bit row offsets inside a sample
    if (V <= 0 || B <= 0) { *rc = DSW_OK; return 1; }
    Bwd3Args A;
    A.tile_meta = plan_t->tile_meta; A.s2_rows = plan_t->s2_rows; A.lrowptr = plan_t->lrowptr;
    A.lcol = plan_t->lcol; A.lval = plan_t->lval;
    A.dY = static_cast<const char*>(dY); A.dX = static_cast<char*>(dX); A.W = static_cast<const float*>(W);
    A.V = (int)V; A.n_tiles = plan_t->n_tiles; A.max_n1 = plan_t->max_n1; A.max_n2 = plan_t->max_n2;
    A.B = (int)B; A.ell_w = (plan_t->reserved + 3) & ~3; A.explicit_tiles = plan_t->explicit_tiles;
    // batch chunks: one workgroup per CU; rounds x (plan + weight staging + pipeline fill, about 3 samples' worth, + samples per chunk)
    const long slots = dsw_device_cus();
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long rounds = (plan_t->n_tiles * c + slots - 1) / slots;
            const double cost = (double)rounds * (3.0 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan_t->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return 0;
    const size_t lds = bwd3_lds_bytes(plan_t);
    if (hipFuncSetAttribute((const void*)cheb3_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        *rc = DSW_ERR_LAUNCH;
        return 1;
    }
    DSW_LAUNCH(cheb3_bwd_fused_kernel, dim3((unsigned)nwg), dim3(NTH3), lds, stream, A);
    *rc = dsw_check_launch();
    return 1;
}
