// Whole ConvCheb forward of a K = 3, 32-input-channel fp32 layer in ONE launch:
//
//     T1 = L X,   T2 = 2 L T1 - X            (layers.py:163-169)
//     Y  = [X | T1 | T2] W + bias            (layers.py:171-178, :375)
//
// The two-hop SpMM of dsw_spmm2.hip (same tile plan, same LDS staging of the tile's 2-ring, same gather code) followed,
// per (tile, sample), by the channel mix on the matrix cores while X, T1 and T2 of the tile rows are still on chip.
// The unfused sequence reads X, T1, T2 back from HBM in the mix kernel (3 of its 5 tensor passes); here the launch moves
// X in, T1 / T2 out (once, for backward) and Y out.
//
// Matrix part: fp32 product on the bf16 pipe by exact 3-way operand splitting (x = h + m + l, 8 + 8 + 8 mantissa bits;
// six MFMA terms, as ts_gemm_x3), v_mfma_f32_16x16x32_bf16 in the SWAPPED orientation  Y^T = W^T T^T :
//   * the "A" operand is the W fragment (16 output channels of the wave's column block x one 32-channel plane), split
//     once per workgroup and held in REGISTERS (3 planes x 3 terms x 4 VGPRs = 36) - no LDS for the panel, so two
//     workgroups still share a CU;
//   * the "B" operand are the tile rows.  Every row is split ONCE, by the thread that produces it (the staging thread
//     for X, the phase-1 / phase-2 thread for T1 / T2: 5.5 VALU per element), into three bf16 images in LDS
//     ([plane][term][64 rows][64 B], 36 KB), from which each wave reads ready-made fragments (one ds_read_b128 per
//     term).  What this replaces, measured at the north-star shape: splitting on the fly in every wave that shares a
//     row (4 column blocks): +32 M VALU instructions, 181 us; v_mfma_f32_16x16x4_f32 (no split): 165 us - the fp32
//     MFMA issues at the VECTOR rate and did not overlap with the other workgroup's gather FMAs (removing the MFMAs
//     gave back 47 of its 61 us);
//   * the accumulator comes out as 4 CONSECUTIVE output channels of one row per lane: 16-byte stores, 64 contiguous
//     bytes per row and instruction.  One 32-channel plane is exactly one MFMA k-step.
// The 36 KB of split images are paid for by single-buffering the input rows: the next sample's rows (prefetched into
// registers right after barrier A, as in dsw_spmm2.hip) are written to LDS after barrier C, when phase 2 - the last
// reader of the input buffer - is over, and BEFORE the MFMA phase, which reads only the split images.  Three
// workgroup barriers per sample.  16-byte chunk kc of row r of a split image sits at chunk kc ^ (2 * ((r >> 3) & 1)):
// the fragment reads (16 rows x one chunk column per lane group) are bank-conflict free.
#include <cstdlib>
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);

namespace {
// streamed-once stores (T1 / T2 are read again only by the backward pass, Y by the next layer): nontemporal, so that
// they do not displace the gathered rows from L2 / Infinity Cache (NS step -1.4 % same-box)
template <typename T4>
static __device__ __forceinline__ void st16(char* p, const T4& v) {
#ifdef DSW_ABL_F3_NOSTORE
    if (p != nullptr) return;      // (never null here: keeps the operands alive)
#endif
    typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
#ifdef DSW_F3_PLAIN_STORE      // A/B builds: cached stores (partial lines of a row meet in L2)
    *reinterpret_cast<u32x4_nt*>(p) = __builtin_bit_cast(u32x4_nt, v);
#else
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(p));
#endif
}


constexpr int NTHREADS = 512;
constexpr int RB = 128;    // bytes of one activation row (32 fp32 channels), in HBM and in the LDS staging buffers
constexpr int RPP = 64;    // rows per pass of the 512 threads (8 lanes of 16 bytes per row)
constexpr int SPLIT_BYTES = 3 * 3 * 64 * 64;   // [plane][term][row][32 bf16]
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct Fwd3Args {
    const int* tile_meta;
    const int* s2_rows;
    const int* lrowptr;
    const unsigned short* lcol;
    const float* lval;
    const char* X;
    char* T1;            // may be null (inference: the basis is not kept)
    char* T2;
    char* Y;
    const float* W;      // [32][3][Fout]
    const float* bias;   // [Fout] or null
    int V, n_tiles, max_n1, max_n2;
    int B, n_chunks, spc, ell_w;
    int Fout;
    int relu;            // ReLU after the bias (ConvBlock)
    int explicit_tiles;  // the plan's tiles are explicit row sets (see dsw_hop2_plan)
    const unsigned char* ell2;   // per-tile ELL image in the LDS layout (dsw_hop2_plan::ell2) or null
    long ell2_stride;
};

static __device__ __forceinline__ float trunc_bf16(float f) { return __uint_as_float(__float_as_uint(f) & 0xffff0000u); }
// truncating fp32 -> bf16 of a pair: v_perm_b32 keeps the two high bytes of each value (lo in the low half)
static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// x = h + m + l exactly (8 + 8 + 8 mantissa bits by truncation; the residuals are exact in fp32)
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
// the three bf16 images of 4 consecutive channels of one tile row -> LDS (8 bytes per term)
static __device__ __forceinline__ void split_store(unsigned char* __restrict__ simg, const int plane, const int row,
                                                   const unsigned c4, const float (&f)[4]) {
#ifdef DSW_ABL_F3_NOSPLIT
    return;
#endif
    float r1[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r1[j] = f[j] - trunc_bf16(f[j]);
        r2[j] = r1[j] - trunc_bf16(r1[j]);
    }
    // byte offset inside the 64-byte row: 16-byte chunk (c4 >> 1) swizzled by the row, 8-byte half (c4 & 1)
    const unsigned off = (unsigned)row * 64u + ((((c4 >> 1) ^ (((unsigned)row >> 2) & 2u)) << 4) | ((c4 & 1u) << 3));
    unsigned char* base = simg + (size_t)plane * (3 * 64 * 64) + off;
    *reinterpret_cast<uint2*>(base) = make_uint2(pack2(f[0], f[1]), pack2(f[2], f[3]));
    *reinterpret_cast<uint2*>(base + 64 * 64) = make_uint2(pack2(r1[0], r1[1]), pack2(r1[2], r1[3]));
    *reinterpret_cast<uint2*>(base + 2 * 64 * 64) = make_uint2(pack2(r2[0], r2[1]), pack2(r2[2], r2[3]));
}

// acc += sum_j val[j] * buf[pos[j]] over the first W entries of one ELL row: fp32 values + u8 list POSITIONS of the rows
// in the staging buffer (W % 4 == 0 storage, any W <= that in use, padded with {own row, 0}); bufc = staging buffer + this
// lane's byte offset in a row.  One byte per index (the 2-ring of a 64-row tile has < 256 rows) is what keeps the
// workgroup under 80 KB at nside = 64 (115 / 175 rows in the fattest tile).
// 4 rows in flight per batch (the two-hop kernel takes 8): 36 registers hold the W fragments for the whole workgroup,
// and a spill costs a scratch access + s_waitcnt vmcnt(0) in the middle of the prefetch window.
static __device__ __forceinline__ void gather_ell(const unsigned char* __restrict__ row_idx, const float* __restrict__ row_val,
                                                  const int W, const unsigned char* __restrict__ bufc, float (&acc)[4]) {
#ifdef DSW_ABL_F3_NOGATHER      // ablation builds only (refused by _native.load unless named by DSW_HIP_LIB)
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
    if (j + 2 <= W) {   // a pair
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
    if (j < W) {        // odd loop length (HEALPix k = 8: 9 entries in 96 % of the rows): a single last entry
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

// NST / NS1: register-stage slots per thread for the gather list (ceil(max_n2 / 64)) and for S1 (ceil(max_n1 / 64));
// the tile is 64 rows = slot 0.  NCB = Fout / 16 column blocks; a wave owns ONE column block (its W fragments stay in
// registers) and RBW = NCB / 2 of the four 16-row blocks of the tile.
// FULL: every tile has 64 rows (V % 64 == 0) and the basis is kept (T != NULL): no store of the loop sits under a
// condition, so the compiler can count the VMEM operations behind the prefetch loads (s_waitcnt vmcnt(n), n > 0).
// KEEP: the basis planes T1 / T2 are stored (training with a backward that reads them); !KEEP with FULL: no T store at all
// (inference, or a backward in the dual form - dsw_bwd3d.hip - which needs X and dY only).
template <int NST, int NS1, int NCB, bool FULL, bool KEEP = true>
__global__ __launch_bounds__(NTHREADS, 4) void cheb3_fwd_fused_kernel(const Fwd3Args P) {
    constexpr int RBW = NCB / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bufX = lds;                                              // [max_n2][128] input rows of the 2-ring
    unsigned char* bufT = bufX + (size_t)P.max_n2 * RB;                     // [max_n1][128] T1 on the 1-ring
    unsigned char* simg = bufT + (size_t)P.max_n1 * RB;                     // split images of the tile rows (36 KB)
    float* ell_val = reinterpret_cast<float*>(simg + SPLIT_BYTES);          // [max_n1][W]
    unsigned char* ell_idx = reinterpret_cast<unsigned char*>(ell_val + (size_t)P.max_n1 * P.ell_w);     // [max_n1][W] u8
    int* rows = reinterpret_cast<int*>(ell_idx + (((size_t)P.max_n1 * P.ell_w + 3) & ~(size_t)3));   // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);

    const long nwg = gridDim.x, orig = blockIdx.x;                          // XCD-aware order (see dsw_spmm2.hip)
#ifdef DSW_F3_SKEW
    // A/B builds: the second workgroup of every CU (first round of the grid: blocks 256..511 on a 256-CU part) starts late, so
    // that its LDS-bound hop phases run under the other workgroup's matrix phase; later rounds inherit the offset
    if (blockIdx.x >= 256u && blockIdx.x < 512u)
        for (int i = 0; i < DSW_F3_SKEW / 1024; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], n1 = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];
    const int rt = P.explicit_tiles ? meta[5] : min(64, P.V - tile * 64);   // tile rows = the first rt list entries
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t sample_bytes = (size_t)P.V * RB;

    int* lrp = reinterpret_cast<int*>(bufT);
    constexpr bool image = FULL;                    // FULL launches carry the tile's ELL as an LDS image of the plan: ONE round of loads, one barrier
    if (tid == 0) *tile_w = image ? meta[5] : 2;
    for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
    if constexpr (image) {
        const unsigned char* src = P.ell2 + (size_t)tile * (size_t)P.ell2_stride;
        unsigned char* dst = reinterpret_cast<unsigned char*>(ell_val);          // values, then the u8 positions: contiguous in LDS
        const int nbytes = P.max_n1 * W * 5;
        for (int i = tid; i < (nbytes >> 4); i += NTHREADS)
            *reinterpret_cast<uint4*>(dst + 16 * i) = *reinterpret_cast<const uint4*>(src + 16 * i);
        for (int i = (nbytes & ~15) + tid; i < nbytes; i += NTHREADS) dst[i] = src[i];
    } else {
        for (int i = tid; i <= n1; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
    }
    __syncthreads();

    const int grp = tid >> 3;                       // row of a 64-row pass
    const unsigned c4 = (unsigned)(tid & 7);        // 16-byte chunk (4 channels) of this lane inside a row
    const unsigned cb = c4 * 16;
    const unsigned tile_off = (unsigned)rows[min(grp, rt - 1)] * (unsigned)RB + cb;    // this thread's tile row (slot 0), sample-relative
    // byte offset (sample-relative) of list position grp + k * 64, index-clamped so that every load is legal and
    // unconditional; re-read from LDS where needed instead of kept in registers (the W fragments need those)
    auto offU = [&](const int k) __attribute__((always_inline)) {
        return (unsigned)rows[min(grp + k * RPP, n2 - 1)] * (unsigned)RB + cb;
    };

    u32x4 su[NST];
    if (b_begin < b_end) {
        const size_t sb = (size_t)b_begin * sample_bytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.X + sb + offU(k));
    }
    if constexpr (!image) {
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
    }

    // ---- W fragments of this wave's column block, split once: lane l holds W[f = 8 (l >> 4) + j][plane][n = 16 cbk + (l & 15)]
    const int wave = tid >> 6, lane = tid & 63;
    const int cbk = wave % NCB;
    const int rb0 = (wave / NCB) * RBW;
    const int l15 = lane & 15, kc = lane >> 4;
    bf16x8_t wh[3], wm[3], wl[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = P.W[((size_t)(8 * kc + j) * 3 + s) * P.Fout + 16 * cbk + l15];
        split3x8(f, wh[s], wm[s], wl[s]);
    }
    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias != nullptr) bias4 = *reinterpret_cast<const f32x4_t*>(P.bias + 16 * cbk + 4 * kc);

    if constexpr (!image) __syncthreads();   // ELL complete (and lrp in bufT dead); with the image it was complete at the first barrier
    const int Wt = *tile_w;                           // gather loop length of this tile: its longest row (may be odd)
    // the first sample's rows (the loop writes the NEXT sample's rows after its barrier C)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = grp + k * RPP;
        if (i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * RB + cb) = su[k];
    }

    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();   // A: bufX(b) complete; everybody is past the MFMA phase of sample b-1 (split images, bufT free)
        const size_t sample = (size_t)b * sample_bytes;
        {   // next sample's input rows: in flight under phases 1 and 2
            const size_t sb = (size_t)(b + 1 < b_end ? b + 1 : b) * sample_bytes;
#pragma unroll
            for (int k = 0; k < NST; ++k)
#ifdef DSW_ABL_F3_NOLOAD
                su[k] = u32x4{(unsigned)sb, offU(k), 0u, 0u};
#else
                su[k] = *reinterpret_cast<const u32x4*>(P.X + sb + offU(k));
#endif
        }
        // ---- phase 1: T1 = L X on S1; the tile rows (slot 0) also leave their split images of X and T1
#ifdef DSW_GATHER_N
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int k = 0; k < NS1; ++k) {
            const int i = grp + k * RPP;
            if (k == 0 || i < n1) {               // slot 0 = the tile rows (64 <= n1 whenever the tile is full)
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                GATHER(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, bufX + cb, acc);
                const uint4 packed = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                __float_as_uint(acc[3]));
                *reinterpret_cast<uint4*>(bufT + (size_t)i * RB + cb) = packed;
                if (k == 0) {
                    // uniform 64-bit base + 32-bit lane offset: the store takes its address from an SGPR pair + one VGPR
                    if constexpr (FULL) { if constexpr (KEEP) st16(P.T1 + sample + tile_off, packed); }
                    else if (P.T1 != nullptr && i < rt) st16(P.T1 + sample + tile_off, packed);
                    split_store(simg, 1, i, c4, acc);
                    const float4 xr = *reinterpret_cast<const float4*>(bufX + (size_t)i * RB + cb);
                    const float xf[4] = {xr.x, xr.y, xr.z, xr.w};
                    split_store(simg, 0, i, c4, xf);
                }
            }
        }
        __syncthreads();   // B
        // ---- phase 2: T2 = 2 L T1 - X on the tile rows -> HBM and split images
        {
            const int i = grp;
            if (FULL || i < rt) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                GATHER(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, bufT + cb, acc);
                const float4 u = *reinterpret_cast<const float4*>(bufX + (size_t)i * RB + cb);
                const float t2[4] = {fmaf(2.f, acc[0], -u.x), fmaf(2.f, acc[1], -u.y), fmaf(2.f, acc[2], -u.z), fmaf(2.f, acc[3], -u.w)};
                if ((FULL && KEEP) || (!FULL && P.T2 != nullptr))
                    st16(P.T2 + sample + tile_off,
                         make_uint4(__float_as_uint(t2[0]), __float_as_uint(t2[1]), __float_as_uint(t2[2]), __float_as_uint(t2[3])));
                split_store(simg, 2, i, c4, t2);
            }
        }
        __syncthreads();   // C: nobody reads bufX / bufT of this sample any more; split images complete
#ifdef DSW_STAGE_EARLY   // A/B builds: next sample's rows -> the (single) input buffer in front of the matrix phase
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * RB + cb) = su[k];
        }
#endif
        // ---- phase 3: Y[tile rows, 16 cbk ..+16] = [X | T1 | T2] W + bias on the matrix cores, from the split images
        f32x4_t acc[RBW];
        unsigned fro[RBW];
#pragma unroll
        for (int r = 0; r < RBW; ++r) {
            const unsigned row = 16u * (rb0 + r) + l15;
            acc[r] = bias4;
            fro[r] = row * 64u + (((unsigned)kc ^ ((row >> 2) & 2u)) << 4);
        }
#ifndef DSW_ABL_F3_NOMFMA
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bf16x8_t th[RBW], tm[RBW], tl[RBW];
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
                const unsigned char* pb = simg + (size_t)s * (3 * 64 * 64) + fro[r];
                th[r] = *reinterpret_cast<const bf16x8_t*>(pb);
                tm[r] = *reinterpret_cast<const bf16x8_t*>(pb + 64 * 64);
                tl[r] = *reinterpret_cast<const bf16x8_t*>(pb + 2 * 64 * 64);
            }
            // six leading terms, smallest first; the RBW independent accumulators are interleaved
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], th[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tl[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], tm[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], th[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tm[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], th[r], acc[r], 0, 0, 0);
        }
#endif
#ifdef DSW_F3_YBOUNCE
        // A/B build: the Y tile through LDS (the dead input buffer; rows padded to Fout * 4 + 16 bytes: conflict-free 16-byte
        // writes), so that every wave instruction stores whole rows - 1 KiB contiguous - instead of 64-byte quarters of 16 rows
        if constexpr (FULL) {
            const unsigned yrb = (unsigned)P.Fout * 4u, ypad = yrb + 16u;
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
                const unsigned row = 16u * (rb0 + r) + l15;
                if (P.relu) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[r][t] = acc[r][t] < 0.f ? 0.f : acc[r][t];
                }
                *reinterpret_cast<f32x4_t*>(bufX + row * ypad + (16u * cbk + 4u * kc) * 4u) = acc[r];
            }
            __syncthreads();   // E: the tile is complete
            const unsigned lpr = yrb >> 4;                       // 16-byte lanes per row (8 or 16)
            char* ydst = P.Y + (size_t)b * ((size_t)P.V * P.Fout * 4);
            for (unsigned q = (unsigned)tid; q < 64u * lpr; q += NTHREADS) {
                const unsigned row = q / lpr, c16 = q - row * lpr;
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(bufX + row * ypad + c16 * 16u);
                st16(ydst + ((unsigned)rows[row] * yrb + c16 * 16u), v);
            }
            __syncthreads();   // F: the buffer is free for the next sample's rows
        } else
#endif
#pragma unroll
        for (int r = 0; r < RBW; ++r) {   // lane: row 16 (rb0 + r) + l15, output channels 16 cbk + 4 kc .. + 3
            const int row = 16 * (rb0 + r) + l15;
            if (P.relu) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = acc[r][t] < 0.f ? 0.f : acc[r][t];   // NaN stays NaN (torch.relu)
            }
            if (FULL || row < rt)
                st16(P.Y + (size_t)b * ((size_t)P.V * P.Fout * 4) +
                         (unsigned)(rows[FULL ? row : min(row, rt - 1)] * P.Fout * 4 + (16 * cbk + 4 * kc) * 4), acc[r]);
        }
#ifndef DSW_STAGE_EARLY
        // next sample's rows -> the (single) input buffer, BEHIND the matrix phase (bufX is free since barrier C, the next
        // barrier A publishes it): the loads get the whole sample to land
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * RB + cb) = su[k];
        }
#endif
    }
}

// ---- dense stencils (one-hop plans: the reference's default k = 20 graph, equiangular): hop 2 + channel mix in one launch ------
// The whole-forward kernel above needs the tile's TWO-ring in LDS, which a 21-32-entry stencil does not leave room for
// (114 KB).  Here hop 1 stays the staged launch of dsw_spmm1s.hip (T1 = L X through HBM / the Infinity Cache) and THIS kernel
// is the rest of the forward: per (tile, sample) the T1 rows of the tile + 1-ring are staged in LDS, T2 = 2 L T1 - X is
// gathered on the tile rows, X / T1 / T2 of the tile rows are split once into the bf16 images and the channel mix runs on
// the matrix cores as above.  What it replaces re-read X, T1 and T2 (3 of its 5 tensor passes) in a separate GEMM launch:
// NS at k = 20: hop 2 59 us + GEMM 112 us.  Two barriers per sample, 65 KB of LDS (two workgroups per CU).
template <int NST, int NCB, bool FULL>
__global__ __launch_bounds__(NTHREADS, 4) void cheb3_hop2mix_kernel(const Fwd3Args P) {
    constexpr int RBW = NCB / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bufT = lds;                                              // [max_n2][128] T1 on the tile + 1-ring
    unsigned char* simg = bufT + (size_t)P.max_n2 * RB;                     // split images of the tile rows (36 KB)
    float* ell_val = reinterpret_cast<float*>(simg + SPLIT_BYTES);          // [64][W]
    unsigned char* ell_idx = reinterpret_cast<unsigned char*>(ell_val + (size_t)64 * P.ell_w);          // [64][W] u8
    int* rows = reinterpret_cast<int*>(ell_idx + (size_t)64 * P.ell_w);     // [max_n2] global row ids (64 W is a multiple of 4)
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);

    const long nwg = gridDim.x, orig = blockIdx.x;                          // XCD-aware order (see dsw_spmm2.hip)
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], rt = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];   // one-hop plan: n1 = tile rows
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t sample_bytes = (size_t)P.V * RB;

    int* lrp = reinterpret_cast<int*>(simg);          // local row pointers of the tile rows, parked in the image area
    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= rt; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    const int grp = tid >> 3;                       // row of a 64-row pass
    const unsigned c4 = (unsigned)(tid & 7);        // 16-byte chunk (4 channels) of this lane inside a row
    const unsigned cb = c4 * 16;
    const unsigned tile_off = (unsigned)rows[min(grp, rt - 1)] * (unsigned)RB + cb;    // this thread's tile row, sample-relative
    auto offU = [&](const int k) __attribute__((always_inline)) {
        return (unsigned)rows[min(grp + k * RPP, n2 - 1)] * (unsigned)RB + cb;
    };
    const char* T1in = P.T1;
    u32x4 su[NST];
    u32x4 xr = {0u, 0u, 0u, 0u};
    if (b_begin < b_end) {
        const size_t sb = (size_t)b_begin * sample_bytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(T1in + sb + offU(k));
        xr = *reinterpret_cast<const u32x4*>(P.X + sb + tile_off);
    }
    const int tile_nnz = lrp[rt];
    for (int t = tid; t < rt * W; t += NTHREADS) {
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

    const int wave = tid >> 6, lane = tid & 63;
    const int cbk = wave % NCB;
    const int rb0 = (wave / NCB) * RBW;
    const int l15 = lane & 15, kc = lane >> 4;
    bf16x8_t wh[3], wm[3], wl[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = P.W[((size_t)(8 * kc + j) * 3 + s) * P.Fout + 16 * cbk + l15];
        split3x8(f, wh[s], wm[s], wl[s]);
    }
    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias != nullptr) bias4 = *reinterpret_cast<const f32x4_t*>(P.bias + 16 * cbk + 4 * kc);

    __syncthreads();   // ELL complete (and lrp in the image area dead)
    const int Wt = *tile_w;
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = grp + k * RPP;
        if (i < n2) *reinterpret_cast<u32x4*>(bufT + (size_t)i * RB + cb) = su[k];
    }

    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();   // A: bufT(b) complete; everybody is past the MFMA phase of sample b-1 (split images free)
        const size_t sample = (size_t)b * sample_bytes;
        u32x4 xn;
        {   // next sample's rows: in flight under the gather
            const size_t sb = (size_t)(b + 1 < b_end ? b + 1 : b) * sample_bytes;
#pragma unroll
            for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(T1in + sb + offU(k));
            xn = *reinterpret_cast<const u32x4*>(P.X + sb + tile_off);
        }
        // ---- T2 = 2 L T1 - X on the tile rows -> HBM; X, T1, T2 of the tile rows -> split images
        if (FULL || grp < rt) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            gather_ell(ell_idx + (size_t)grp * W, ell_val + (size_t)grp * W, Wt, bufT + cb, acc);
            const float xf[4] = {__uint_as_float(xr[0]), __uint_as_float(xr[1]), __uint_as_float(xr[2]), __uint_as_float(xr[3])};
            const float t2[4] = {fmaf(2.f, acc[0], -xf[0]), fmaf(2.f, acc[1], -xf[1]), fmaf(2.f, acc[2], -xf[2]), fmaf(2.f, acc[3], -xf[3])};
            st16(P.T2 + sample + tile_off,
                 make_uint4(__float_as_uint(t2[0]), __float_as_uint(t2[1]), __float_as_uint(t2[2]), __float_as_uint(t2[3])));
            split_store(simg, 2, grp, c4, t2);
            const float4 t1 = *reinterpret_cast<const float4*>(bufT + (size_t)grp * RB + cb);
            const float t1f[4] = {t1.x, t1.y, t1.z, t1.w};
            split_store(simg, 1, grp, c4, t1f);
            split_store(simg, 0, grp, c4, xf);
        }
        __syncthreads();   // C: nobody reads bufT of this sample any more; split images complete
#ifdef DSW_STAGE_EARLY
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufT + (size_t)i * RB + cb) = su[k];
        }
#endif
        xr = xn;
        // ---- Y[tile rows, 16 cbk ..+16] = [X | T1 | T2] W + bias on the matrix cores (as cheb3_fwd_fused_kernel)
        f32x4_t acc[RBW];
        unsigned fro[RBW];
#pragma unroll
        for (int r = 0; r < RBW; ++r) {
            const unsigned row = 16u * (rb0 + r) + l15;
            acc[r] = bias4;
            fro[r] = row * 64u + (((unsigned)kc ^ ((row >> 2) & 2u)) << 4);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bf16x8_t th[RBW], tm[RBW], tl[RBW];
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
                const unsigned char* pb = simg + (size_t)s * (3 * 64 * 64) + fro[r];
                th[r] = *reinterpret_cast<const bf16x8_t*>(pb);
                tm[r] = *reinterpret_cast<const bf16x8_t*>(pb + 64 * 64);
                tl[r] = *reinterpret_cast<const bf16x8_t*>(pb + 2 * 64 * 64);
            }
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], th[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tl[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], tm[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], th[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], tm[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], th[r], acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RBW; ++r) {
            const int row = 16 * (rb0 + r) + l15;
            if (P.relu) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = acc[r][t] < 0.f ? 0.f : acc[r][t];
            }
            if (FULL || row < rt)
                st16(P.Y + (size_t)b * ((size_t)P.V * P.Fout * 4) +
                         (unsigned)(rows[FULL ? row : min(row, rt - 1)] * P.Fout * 4 + (16 * cbk + 4 * kc) * 4), acc[r]);
        }
#ifndef DSW_STAGE_EARLY
        // next sample's T1 rows -> the (single) staging buffer: AFTER the matrix phase, so that the loads had the gather AND the
        // matrix phase to land (bufT is free since barrier C; the next barrier A publishes it)
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufT + (size_t)i * RB + cb) = su[k];
        }
#endif
    }
}

size_t hop2mix_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    size_t s = (size_t)plan->max_n2 * RB + SPLIT_BYTES + (size_t)64 * ell_w * 5;
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;
    return (s + 15) & ~(size_t)15;
}

template <int NST, bool FULL>
int launch_h2m(const Fwd3Args& A, long nwg, size_t lds, hipStream_t stream) {
#define DSW_H2M(N_)                                                                                                     \
    case N_: {                                                                                                          \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)cheb3_hop2mix_kernel<NST, N_, FULL>,                    \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return DSW_ERR_LAUNCH;                                                                                      \
        DSW_LAUNCH((cheb3_hop2mix_kernel<NST, N_, FULL>), dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A);         \
        break;                                                                                                          \
    }
    switch (A.Fout / 16) {
        DSW_H2M(2) DSW_H2M(4)
        default: return DSW_ERR_BAD_ARG;
    }
#undef DSW_H2M
    return dsw_check_launch();
}

size_t fwd3_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    size_t s = (size_t)(plan->max_n1 + (size_t)plan->max_n2) * RB + SPLIT_BYTES;
    s += (size_t)plan->max_n1 * ell_w * 4 + (((size_t)plan->max_n1 * ell_w + 3) & ~(size_t)3);   // fp32 values + u8 positions
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;
    return (s + 15) & ~(size_t)15;
}

template <int NST, int NS1, bool FULL, bool KEEP = true>
int launch_ncb(const Fwd3Args& A, long nwg, size_t lds, hipStream_t stream) {
#define DSW_F3(N_)                                                                                                      \
    case N_: {                                                                                                          \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)cheb3_fwd_fused_kernel<NST, NS1, N_, FULL, KEEP>,       \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return DSW_ERR_LAUNCH;                                                                                      \
        DSW_LAUNCH((cheb3_fwd_fused_kernel<NST, NS1, N_, FULL, KEEP>), dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A); \
        break;                                                                                                          \
    }
    switch (A.Fout / 16) {
        DSW_F3(2) DSW_F3(4)
        default: return DSW_ERR_BAD_ARG;
    }
#undef DSW_F3
    return dsw_check_launch();
}

}  // namespace

// 1 if the one-launch forward exists for this layer shape and plan (pointer alignment aside)
int dsw_cheb3_fwd_fused_eligible(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    static const char* env = dsw_diag_env("DSW_FWD_FUSED");   // "0": generic sequence (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || (Fout != 32 && Fout != 64)) return 0;   // (Fout = 128 compiles but spills)
    if (!plan || plan->tile_rows != 64 || !dsw_spmm2_supported(plan, Fin, dtype)) return 0;
    if (plan->max_n2 > 255) return 0;                       // u8 list positions in the ELL
    if (fwd3_lds_bytes(plan) > 80 * 1024) return 0;         // two workgroups per CU or not at all
    const int nst = (plan->max_n2 + RPP - 1) / RPP, ns1 = (plan->max_n1 + RPP - 1) / RPP;
    return (nst > 4 || ns1 > nst) ? 0 : 1;
}

// Runs the whole forward in one launch if the shape / plan allow it.  Returns 1 if it took the call (*rc = status),
// 0 if the caller must use the generic sequence (basis launches + channel-mix GEMM).
int dsw_cheb3_fwd_fused_try(const dsw_hop2_plan* plan, int64_t V, const void* X, const void* W, const void* bias, void* Y,
                            void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                            int* rc, int relu) {
    if (!dsw_cheb3_fwd_fused_eligible(plan, Fin, Fout, K, dtype)) return 0;
    if (!dsw_aligned16(X) || !dsw_aligned16(Y) || !dsw_aligned16(W) || (bias && !dsw_aligned16(bias)) ||
        (T && !dsw_aligned16(T)))
        return 0;
    const size_t lds = fwd3_lds_bytes(plan);
    const int nst = (plan->max_n2 + RPP - 1) / RPP, ns1 = (plan->max_n1 + RPP - 1) / RPP;
    if (V <= 0 || B <= 0) { *rc = DSW_OK; return 1; }
    Fwd3Args A;
    A.tile_meta = plan->tile_meta; A.s2_rows = plan->s2_rows; A.lrowptr = plan->lrowptr;
    A.lcol = plan->lcol; A.lval = plan->lval;
    A.X = static_cast<const char*>(X);
    const size_t planeb = (size_t)B * (size_t)V * RB;
    A.T1 = static_cast<char*>(T);
    A.T2 = T ? static_cast<char*>(T) + planeb : nullptr;
    A.Y = static_cast<char*>(Y);
    A.W = static_cast<const float*>(W); A.bias = static_cast<const float*>(bias);
    A.V = (int)V; A.n_tiles = plan->n_tiles; A.max_n1 = plan->max_n1; A.max_n2 = plan->max_n2;
    A.B = (int)B; A.ell_w = (plan->reserved + 3) & ~3; A.Fout = (int)Fout; A.relu = relu;
    A.explicit_tiles = plan->explicit_tiles;
    A.ell2 = plan->explicit_tiles ? nullptr : plan->ell2; A.ell2_stride = (long)plan->ell2_stride;

    // batch chunks: same cost model as the two-hop kernel (rounds x (staging + samples per chunk))
    const long slots = 256L * ((160 * 1024) / (long)lds > 0 ? (160 * 1024) / (long)lds : 1);
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long rounds = (plan->n_tiles * c + slots - 1) / slots;
            const double cost = (double)rounds * (1.5 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return 0;
    int r;
    const bool full = (V % 64 == 0) && !plan->explicit_tiles && plan->ell2 != nullptr;
    const bool keep = T != nullptr;
#define DSW_F3_PICK(A_, B_)                                                                           \
    r = !full ? launch_ncb<A_, B_, false>(A, nwg, lds, stream)                                        \
              : keep ? launch_ncb<A_, B_, true, true>(A, nwg, lds, stream) : launch_ncb<A_, B_, true, false>(A, nwg, lds, stream)
    if (nst == 3 && ns1 == 2) DSW_F3_PICK(3, 2);
    else if (nst == 2 && ns1 <= 2) DSW_F3_PICK(2, 2);
    else if (nst == 3) DSW_F3_PICK(3, 3);
    else DSW_F3_PICK(4, 4);
#undef DSW_F3_PICK
    *rc = r;
    return 1;
}

int dsw_spmm1s_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);
int dsw_spmm1s_launch(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z, const void* Z2, void* Y,
                      int64_t B, int64_t C, float a, float b, float c, int dtype, hipStream_t stream, int stream_out);

// 1 if the forward of this layer on a ONE-hop plan runs as: staged hop 1 (dsw_spmm1s.hip) + ONE launch for hop 2 and the channel mix
int dsw_cheb3_hop2mix_eligible(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    static const char* env = dsw_diag_env("DSW_HOP2MIX");   // "0": staged hops + separate GEMM (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || (Fout != 32 && Fout != 64)) return 0;
    if (!plan || plan->hops != 1 || plan->tile_rows != 64 || plan->reserved <= 0 || !dsw_spmm1s_supported(plan, Fin, dtype)) return 0;
    if (plan->max_n2 > 255 || plan->max_n1 > 64) return 0;     // u8 list positions; tile rows = one pass
    if (hop2mix_lds_bytes(plan) > 80 * 1024) return 0;          // two workgroups per CU or not at all
    return (plan->max_n2 + RPP - 1) / RPP <= 3 ? 1 : 0;       // (4 staging slots spill)
}

// Runs the forward as hop 1 + (hop 2 + channel mix) if the shape / plan allow it.  Returns 1 if it took the call (*rc = status
// of the launches, *stage = 1 after hop 1 for the caller's tracing), 0 if the caller must use the generic sequence.  T (the
// basis planes kept for backward) is REQUIRED: T1 travels through it.
int dsw_cheb3_hop2mix_try(const dsw_hop2_plan* plan, int64_t V, const void* X, const void* W, const void* bias, void* Y,
                          void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int* rc,
                          int relu, void (*after_hop1)(void*), void* ctx) {
    if (T == nullptr || !dsw_cheb3_hop2mix_eligible(plan, Fin, Fout, K, dtype)) return 0;
    if (!dsw_aligned16(X) || !dsw_aligned16(Y) || !dsw_aligned16(W) || (bias && !dsw_aligned16(bias)) || !dsw_aligned16(T)) return 0;
    if (V <= 0 || B <= 0) { *rc = DSW_OK; return 1; }
    const size_t planeb = (size_t)B * (size_t)V * RB;
    // hop 1: T1 = L X (cached stores: the kernel below gathers it)
    *rc = dsw_spmm1s_launch(plan, V, X, nullptr, nullptr, T, B, Fin, 1.f, 0.f, 0.f, dtype, stream, 0);
    if (after_hop1) after_hop1(ctx);
    if (*rc != DSW_OK) return 1;
    const size_t lds = hop2mix_lds_bytes(plan);
    const int nst = (plan->max_n2 + RPP - 1) / RPP;
    Fwd3Args A;
    A.tile_meta = plan->tile_meta; A.s2_rows = plan->s2_rows; A.lrowptr = plan->lrowptr;
    A.lcol = plan->lcol; A.lval = plan->lval;
    A.X = static_cast<const char*>(X);
    A.T1 = static_cast<char*>(T);
    A.T2 = static_cast<char*>(T) + planeb;
    A.Y = static_cast<char*>(Y);
    A.W = static_cast<const float*>(W); A.bias = static_cast<const float*>(bias);
    A.V = (int)V; A.n_tiles = plan->n_tiles; A.max_n1 = plan->max_n1; A.max_n2 = plan->max_n2;
    A.B = (int)B; A.ell_w = (plan->reserved + 3) & ~3; A.Fout = (int)Fout; A.relu = relu;
    A.explicit_tiles = plan->explicit_tiles;
    A.ell2 = nullptr; A.ell2_stride = 0;
    const long slots = 256L * ((160 * 1024) / (long)lds > 1 ? 2 : 1);
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long rounds = (plan->n_tiles * c + slots - 1) / slots;
            const double cost = (double)rounds * (1.5 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) { *rc = DSW_ERR_BAD_ARG; return 1; }
    const bool full = (V % 64 == 0) && !plan->explicit_tiles;
    int r;
    if (nst <= 2) r = full ? launch_h2m<2, true>(A, nwg, lds, stream) : launch_h2m<2, false>(A, nwg, lds, stream);
    else r = full ? launch_h2m<3, true>(A, nwg, lds, stream) : launch_h2m<3, false>(A, nwg, lds, stream);
    *rc = r;
    return 1;
}
