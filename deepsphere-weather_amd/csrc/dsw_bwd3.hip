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
// 1024 threads, one workgroup per CU (141 KB of LDS at nside 64): 16 waves = the occupancy of the two 512-thread
// workgroups of the forward kernel.  Per sample: NCH chunk steps (split-store chunk c, one barrier, MFMAs of chunk c
// while the loads of the next sample's chunk c are in flight), then hop 1 and hop 2, each behind one barrier.
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

constexpr int NTH = 1024;
constexpr int XB = 128;          // bytes of a dX / G row in HBM (32 fp32 channels)
constexpr int YB = 256;          // bytes of a dY row (64 fp32 channels)
constexpr int GS = 144;          // LDS stride of a G row: 128 + 16 (the 16 rows of an accumulator store hit distinct banks)
constexpr int IMG_TERM = 64 * 128;           // one term of a 64-row chunk image
constexpr int IMG_BYTES = 3 * IMG_TERM;      // 24 KB
constexpr int WIMG_PLANE = 3 * 32 * 128;     // [term][32 f][64 o bf16]
constexpr int WIMG_BYTES = 3 * WIMG_PLANE;   // 36 KB
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

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

static __device__ __forceinline__ float trunc_bf16(float f) { return __uint_as_float(__float_as_uint(f) & 0xffff0000u); }
static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// byte offset of 16-byte chunk c of row r inside a [rows][128 B] image
static __device__ __forceinline__ unsigned img_off(const unsigned r, const unsigned c) { return r * 128u + ((c ^ ((r >> 1) & 7u)) << 4); }

// the three bf16 images of 4 consecutive channels (quad q of the row's 16) of one image row -> LDS (8 bytes per term)
static __device__ __forceinline__ void split_store4(unsigned char* __restrict__ img, const int term_stride, const unsigned row,
                                                    const unsigned q, const float (&f)[4]) {
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
static __device__ __forceinline__ void gather_ell(const unsigned char* __restrict__ row_idx, const float* __restrict__ row_val,
                                                  const int W, const unsigned char* __restrict__ bufc, float (&acc)[4]) {
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

// six leading terms of the split product, smallest first: acc += A (3 terms) x B (3 terms)
static __device__ __forceinline__ f32x4_t mfma6(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x4_t acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// NCH = ceil(max_n2 / 64): chunks of 64 list rows per sample
template <int NCH>
__global__ __launch_bounds__(NTH, 4) void cheb3_bwd_fused_kernel(const Bwd3Args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* img = lds;                                               // [2][3 terms][64 rows][128 B]
    unsigned char* wimg = img + 2 * IMG_BYTES;                              // [3 planes][3 terms][32 f][128 B]
    unsigned char* g2 = wimg + WIMG_BYTES;                                  // [max_n2][GS] G_2 on the 2-ring
    unsigned char* g1 = g2 + (size_t)P.max_n2 * GS;                         // [max_n1][GS] G_1, then H_1, on the 1-ring
    unsigned char* g0 = g1 + (size_t)P.max_n1 * GS;                         // [64][GS]     G_0' on the tile
    float* ell_val = reinterpret_cast<float*>(g0 + 64 * GS);                // [max_n1][W]
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

    int* lrp = reinterpret_cast<int*>(g1);           // local row pointers, parked in g1 until the ELL is built
    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTH) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= n1; i += NTH) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    // staging role: list position 64 c + srow of chunk c, 16-byte lane sq (dY channels 4 sq .. 4 sq + 3)
    const int srow = tid >> 4;
    const unsigned sq = (unsigned)(tid & 15);
    unsigned offY[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) offY[c] = (unsigned)rows[min(64 * c + srow, n2 - 1)] * (unsigned)YB + sq * 16u;
    u32x4 su[NCH];
    if (b_begin < b_end) {
        const size_t sb = (size_t)b_begin * y_sample;
#pragma unroll
        for (int c = 0; c < NCH; ++c) su[c] = *reinterpret_cast<const u32x4*>(P.dY + sb + offY[c]);
    }
    // CSR -> ELL of the tile + 1-ring rows (as dsw_fwd3.hip)
    const int tile_nnz = lrp[n1];
    for (int t = tid; t < n1 * W; t += NTH) {
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
    // split image of the weights: plane 0 = W_0 - W_2 (the subtraction of the raw plane K-1 at the top of the adjoint
    // recurrence, folded into the weights), planes 1, 2 = W_1, W_2; image row = dX channel f, columns = dY channels o
    for (int e = tid; e < 3 * 32 * 16; e += NTH) {
        const int p = e / (32 * 16), f = (e / 16) & 31;
        const unsigned q = (unsigned)(e & 15);
        const float4 w = *reinterpret_cast<const float4*>(P.W + ((size_t)f * 3 + p) * 64 + 4 * q);
        float v[4] = {w.x, w.y, w.z, w.w};
        if (p == 0) {
            const float4 w2 = *reinterpret_cast<const float4*>(P.W + ((size_t)f * 3 + 2) * 64 + 4 * q);
            v[0] -= w2.x; v[1] -= w2.y; v[2] -= w2.z; v[3] -= w2.w;
        }
        split_store4(wimg + (size_t)p * WIMG_PLANE, 32 * 128, (unsigned)f, q, v);
    }
    __syncthreads();   // ELL and weight image complete (lrp in g1 dead)
    const int Wt = *tile_w;

    // MFMA role of this wave: row block rb of a chunk (16 list rows), dX channel block fb (16 channels), plane group pg
    // (0: plane 2 = needed on the whole 2-ring; 1: plane 1 on the 1-ring and plane 0' on the tile)
    const int wave = tid >> 6, lane = tid & 63;
    const int rb = wave & 3, fb = (wave >> 2) & 1, pg = wave >> 3;
    const unsigned l15 = (unsigned)(lane & 15), kc = (unsigned)(lane >> 4);
    const unsigned a_off0 = img_off(16u * fb + l15, kc), a_off1 = img_off(16u * fb + l15, 4u + kc);   // k-steps 0 / 1 of the W image
    const unsigned b_off0 = img_off(16u * rb + l15, kc), b_off1 = img_off(16u * rb + l15, 4u + kc);   // ... of the chunk image
    const unsigned g_st = (16u * rb + l15) * GS + (16u * fb + 4u * kc) * 4u;   // accumulator -> G row of the chunk, 16 bytes

    // gather role (hops): list position grow, 16-byte chunk gc of the 128-byte row
    const int grow = tid >> 3;
    const unsigned gcb = (unsigned)(tid & 7) * 16u;

    for (int b = b_begin; b < b_end; ++b) {
        const size_t sb_next = (size_t)(b + 1 < b_end ? b + 1 : b) * y_sample;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned char* im = img + (size_t)(c & 1) * IMG_BYTES;
            {   // this sample's chunk c -> split image; its register then takes the next sample's chunk c
                const float f4[4] = {__uint_as_float(su[c][0]), __uint_as_float(su[c][1]), __uint_as_float(su[c][2]),
                                     __uint_as_float(su[c][3])};
                split_store4(im, IMG_TERM, (unsigned)srow, sq, f4);
                su[c] = *reinterpret_cast<const u32x4*>(P.dY + sb_next + offY[c]);
            }
            __syncthreads();   // image of chunk c complete; everybody is past the hops of the previous sample
            const int p0 = 64 * c + 16 * rb;                // first list position of this wave's row block
            const bool need = pg == 0 ? p0 < n2 : p0 < n1;  // uniform per wave
            if (need) {
                bf16x8_t bf0[3], bf1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    bf0[t] = *reinterpret_cast<const bf16x8_t*>(im + (size_t)t * IMG_TERM + b_off0);
                    bf1[t] = *reinterpret_cast<const bf16x8_t*>(im + (size_t)t * IMG_TERM + b_off1);
                }
                const int plane = pg == 0 ? 2 : 1;
                {
                    const unsigned char* wp = wimg + (size_t)plane * WIMG_PLANE;
                    bf16x8_t a0[3], a1[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        a0[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a_off0);
                        a1[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a_off1);
                    }
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mfma6(a0, bf0, acc);
                    acc = mfma6(a1, bf1, acc);
                    unsigned char* gdst = (pg == 0 ? g2 : g1) + (size_t)(64 * c) * GS + g_st;
                    if (p0 + (int)l15 < (pg == 0 ? n2 : n1)) *reinterpret_cast<f32x4_t*>(gdst) = acc;
                }
                if (pg == 1 && c == 0) {   // the tile rows: plane 0'
                    const unsigned char* wp = wimg;
                    bf16x8_t a0[3], a1[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        a0[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a_off0);
                        a1[t] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)t * (32 * 128) + a_off1);
                    }
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mfma6(a0, bf0, acc);
                    acc = mfma6(a1, bf1, acc);
                    *reinterpret_cast<f32x4_t*>(g0 + g_st) = acc;
                }
            }
        }
        __syncthreads();   // G_2, G_1, G_0' complete
        // ---- hop 1: H_1 = G_1 + 2 L^T G_2 on the tile + 1-ring rows, in place
        if (grow < n1) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            gather_ell(ell_idx + (size_t)grow * W, ell_val + (size_t)grow * W, Wt, g2 + gcb, acc);
            float4* hp = reinterpret_cast<float4*>(g1 + (size_t)grow * GS + gcb);
            const float4 g = *hp;
            *hp = make_float4(fmaf(2.f, acc[0], g.x), fmaf(2.f, acc[1], g.y), fmaf(2.f, acc[2], g.z), fmaf(2.f, acc[3], g.w));
        }
        __syncthreads();
        // ---- hop 2: dX = G_0' + L^T H_1 on the tile rows -> HBM
        if (grow < rt) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            gather_ell(ell_idx + (size_t)grow * W, ell_val + (size_t)grow * W, Wt, g1 + gcb, acc);
            const float4 g = *reinterpret_cast<const float4*>(g0 + (size_t)grow * GS + gcb);
            const float4 o = make_float4(g.x + acc[0], g.y + acc[1], g.z + acc[2], g.w + acc[3]);
            st16_nt(P.dX + (size_t)b * x_sample + (size_t)rows[grow] * XB + gcb, o);
        }
        // (no barrier here: the next sample's first chunk step has one before anything overwrites the G buffers)
    }
}

size_t bwd3_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    size_t s = 2 * (size_t)IMG_BYTES + WIMG_BYTES + ((size_t)plan->max_n2 + plan->max_n1 + 64) * GS;
    s += (size_t)plan->max_n1 * ell_w * 4 + (((size_t)plan->max_n1 * ell_w + 3) & ~(size_t)3);   // fp32 values + u8 positions
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;
    return (s + 15) & ~(size_t)15;
}

template <int NCH>
int launch_bwd3(const Bwd3Args& A, long nwg, size_t lds, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)cheb3_bwd_fused_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return DSW_ERR_LAUNCH;
    DSW_LAUNCH((cheb3_bwd_fused_kernel<NCH>), dim3((unsigned)nwg), dim3(NTH), lds, stream, A);
    return dsw_check_launch();
}

}  // namespace

// 1 if the one-launch dgrad + adjoint exists for this layer shape and plan of L^T (pointer alignment aside)
int dsw_cheb3_bwd_fused_eligible(const dsw_hop2_plan* plan_t, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    static const char* env = dsw_diag_env("DSW_BWD3_FUSED");   // "0": separate dgrad planes + adjoint pair (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || Fout != 64) return 0;
    if (!plan_t || plan_t->hops == 1 || plan_t->tile_rows != 64 || !dsw_spmm2_supported(plan_t, Fin, dtype)) return 0;
    if (plan_t->max_n2 > 255 || plan_t->max_n1 > 128) return 0;     // u8 list positions; hop 1 in one pass of the 1024 threads
    if (bwd3_lds_bytes(plan_t) > 160 * 1024) return 0;
    return 1;
}

// dX from dY in one launch if the shape / plan allow it.  Returns 1 if it took the call (*rc = status), 0 if the caller
// must use the generic sequence (dgrad planes + adjoint recurrence).
int dsw_cheb3_bwd_fused_try(const dsw_hop2_plan* plan_t, int64_t V, const void* dY, const void* W, void* dX, int64_t B,
                            int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int* rc) {
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
    // batch chunks: one workgroup per CU; rounds x (plan + weight image staging, about 2 samples' worth, + samples per chunk)
    const long slots = dsw_device_cus();
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long rounds = (plan_t->n_tiles * c + slots - 1) / slots;
            const double cost = (double)rounds * (2.0 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan_t->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return 0;
    const size_t lds = bwd3_lds_bytes(plan_t);
    const int nch = (plan_t->max_n2 + 63) / 64;
    switch (nch) {
        case 1: *rc = launch_bwd3<1>(A, nwg, lds, stream); break;
        case 2: *rc = launch_bwd3<2>(A, nwg, lds, stream); break;
        case 3: *rc = launch_bwd3<3>(A, nwg, lds, stream); break;
        default: *rc = launch_bwd3<4>(A, nwg, lds, stream); break;
    }
    return 1;
}
