// Fused two-hop SpMM: two consecutive applications of one sparse operator in a single launch,
//
//     Y1 = a1 * (A U)  + b1 * Z1 + d1 * Z1b          (first hop, needed on the 1-ring of a tile)
//     Y2 = a2 * (A Y1) + b2 * U  + c2 * Z2           (second hop, on the tile rows)
//
// which covers a pair of Chebyshev steps  T_k = 2 L T_{k-1} - T_{k-2},  T_{k+1} = 2 L T_k - T_{k-1}
// (forward, layers.py:163-169) and a pair of adjoint steps (its autograd).  The intermediate Y1 of
// the halo rows lives only in LDS, so the K=3 forward recurrence moves 1 read + 2 writes of the
// activation tensor through HBM instead of 3 reads + 2 writes, and the K=3 adjoint 3 reads + 1
// write instead of 5 reads + 2 writes.
//
// A workgroup owns (tile of R consecutive rows) x (one sample).  The host-side plan
// (dsw_amd/hop2.py) gives for every tile the gather list S2 (tile rows first, then the 1-ring =
// S1, then the 2-ring) and the CSR of the S1 rows with columns rewritten as positions in that list.
//   phase 0: local CSR + gather list -> LDS (coalesced), then the U rows of S2 -> LDS (16 B / lane)
//   phase 1: Y1 on S1 from LDS gathers (all operands of the inner loop are LDS reads) -> LDS
//   phase 2: Y2 on the tile rows from LDS gathers -> HBM
// HBM-bound by construction: every U row is fetched ~|S2|/R times, but only the first fetch
// misses L2 (blocks of one XCD walk contiguous tiles), and the gathers - k+1 per row and hop,
// the limiter of the one-hop kernel on the L1/TA path - run on the LDS pipe at 4x the rate.
#include <cstdlib>
#include "dsw_hop_common.h"
#include "../../include/dsw_hip.h"

namespace {
constexpr int NTHREADS = 512;

struct Hop2Args {
    const int* tile_meta;       // [n_tiles][6]
    const int* s2_rows;
    const int* lrowptr;
    const unsigned short* lcol;
    const float* lval;
    const char* U;
    const char* Z1;
    const char* Z1b;
    const char* Z2;
    char* Y1;
    char* Y2;
    float a1, b1, d1, a2, b2, c2;
    int V, n_tiles, tile_rows, max_n1, max_n2, max_nnz;
    int row_bytes;              // bytes of one STAGED row: the whole row (C * element size, multiple of 16) or one 128-byte
                                // channel chunk of it
    int row_stride;             // bytes between consecutive rows in HBM (C * element size)
    int ncc;                    // channel chunks per row (row_stride / row_bytes); B counts (sample, chunk) pairs
    int lpr;                    // 16-byte lanes per row
    int B;
    int n_chunks;               // batch chunks per tile (grid = n_tiles * n_chunks)
    int spc;                    // samples per chunk
    int ell_w;                  // ELL width in LDS: max row length of the plan rounded up to 4
    int single_buf;             // 1: one input-row buffer (halves the staging LDS so that a second workgroup fits the CU)
    int explicit_tiles;         // 1: the tile's rows are the first meta[5] entries of its gather list (any row set)
};


// NST = ceil(max_n2 / rows-per-pass): register-stage slots per thread.  Slot k of a lane group is list
// position i = grp + k*rpp: its U row (every slot) and - S1 and the tile rows being prefixes of the
// gather list - also its Z1/Z1b row (i < n1) and its Z2 row (i < rt).
constexpr int MAXST = 8;

// One workgroup = (tile, chunk of the batch).  The tile's plan slice is expanded to ELL in LDS once and
// reused for every sample of the chunk.  Per sample: the epilogue operands Z* of THIS sample and then the
// U rows of the NEXT sample are requested in one unconditional, index-clamped burst right after the
// barrier; vmcnt completes in order, so waiting for Z* (first use: end of the first phase-1 task) leaves
// the U burst in flight under phases 1 and 2, and it lands in the other half of the double-buffered bufX
// at the top of the next iteration.
// HZB: a second first-hop operand Z1b may be present (K >= 4 adjoint steps); without it the K = 3 adjoint pair of the
// k = 20 stencil (3 S1 slots) fits 128 registers = two workgroups per CU.
template <bool BF16, int NST, bool HZA, bool HZ2, int NS1 = NST, int NS2 = NST, bool BYTEOFF = false, bool HZB = HZA>
__global__ __launch_bounds__(NTHREADS, ((HZA && HZB && NS1 > 2) ? 2 : 4)) void spmm2_fused_kernel(const Hop2Args P) {
    using R = Row16<BF16>;
    using VT = typename R::V;
    constexpr int N = R::N;
    constexpr int GBATCH = (HZA && !HZB && NS1 > 2) ? 4 : 8;   // the variant squeezed under 128 registers
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // LDS carve-up (all offsets multiples of 16)
    unsigned char* bufX0 = lds;                                            // [max_n2][row_bytes], sample s
    // sample s+1 (aliases bufX0 when single-buffered: the loop then ends with a barrier so that nobody still reads it)
    unsigned char* bufX1 = P.single_buf ? bufX0 : bufX0 + (size_t)P.max_n2 * P.row_bytes;
    unsigned char* bufT = bufX1 + (size_t)P.max_n2 * P.row_bytes;          // [max_n1][row_bytes]
    float* ell_val = reinterpret_cast<float*>(bufT + (size_t)P.max_n1 * P.row_bytes);           // [max_n1][W] values
    unsigned short* ell_idx = reinterpret_cast<unsigned short*>(ell_val + (size_t)P.max_n1 * P.ell_w);   // [max_n1][W] list positions
    int* rows = reinterpret_cast<int*>(ell_idx + (size_t)P.max_n1 * P.ell_w);                    // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);                            // longest local row of THIS tile

    // XCD-aware order: each XCD (hardware block id % 8) walks one contiguous range of (tile, batch chunk)
    const long nwg = gridDim.x, orig = blockIdx.x;
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], n1 = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];
    // tile rows = the first rt entries of the gather list (consecutive rows tile * tile_rows .. unless the plan carries
    // explicit row sets); every address below goes through that list
    const int rt = P.explicit_tiles ? meta[5] : min(P.tile_rows, P.V - tile * P.tile_rows);
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t sample_bytes = (size_t)P.V * P.row_stride;
    // "sample" b of the loops below = (real sample b / ncc, channel chunk b % ncc): wide rows are processed one
    // 128-byte channel chunk at a time (the operator acts on every channel alike), so that the staged neighbourhood is
    // as small as for a 32-channel layer whatever the layer's width
    // (advanced incrementally in the sample loop: a divide + 64-bit multiply per offset is ~40 scalar instructions, and
    // the scalar unit is shared by all waves of the CU)
    struct Cursor { size_t off; int cc; };
    auto cursor_at = [&](const int b) __attribute__((always_inline)) {
        const int bs = b / P.ncc;
        return Cursor{(size_t)bs * sample_bytes + (size_t)(b - bs * P.ncc) * P.row_bytes, b - bs * P.ncc};
    };
    auto advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.off += P.row_bytes;
        if (++c.cc == P.ncc) { c.cc = 0; c.off += sample_bytes - (size_t)P.ncc * P.row_bytes; }
    };
    Cursor c_cur = cursor_at(b_begin);

    // ---- the tile's plan slice -> LDS, once for all samples of this workgroup.  Ordered so that nothing waits on
    // a chain of dependent global loads: (1) gather list + local row pointers (parked in bufT, free until the first
    // phase 1), (2) the first sample's input rows are REQUESTED, (3) the CSR -> ELL expansion runs under them with
    // independent loads only (row pointers come from LDS).
    int* lrp = reinterpret_cast<int*>(bufT);          // [n1 + 1] local row pointers (relative to nnz_off)
    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= n1; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    const int lpr = P.lpr;
    const int rpp = NTHREADS / lpr;                 // rows per pass
    const int grp0 = tid / lpr;
    const bool lane_ok = grp0 < rpp;
    const int grp = lane_ok ? grp0 : rpp - 1;       // leftover lanes shadow the last group (never store)
    const int cb = ((tid - grp0 * lpr) * 16) % P.row_bytes;   // byte offset of this lane inside the row

    // per-slot byte offsets (sample-relative), clamped so that every load is legal and unconditional
    unsigned offU[NST], offZ1[NS1], offZ2[NS2];   // < 2^32: one sample of one tensor
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = grp + k * rpp;
        offU[k] = (unsigned)rows[min(i, n2 - 1)] * (unsigned)P.row_stride + cb;
        if (k < NS1) offZ1[k] = (unsigned)rows[min(i, n1 - 1)] * (unsigned)P.row_stride + cb;
        if (k < NS2) offZ2[k] = (unsigned)rows[min(i, rt - 1)] * (unsigned)P.row_stride + cb;   // = this slot's OUTPUT row
    }

    u32x4 su[NST];
    if (b_begin < b_end) {
        const size_t sb = c_cur.off;
#pragma unroll
        for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.U + sb + offU[k]);
    }
    // CSR -> ELL (entries hold the BYTE offset of the source row inside a staging buffer: one multiply less per
    // gather), padding = {own row, weight 0}
    const int tile_nnz = lrp[n1];
    for (int t = tid; t < n1 * W; t += NTHREADS) {
        const int i = t / W, j = t - i * W;
        const int p0 = lrp[i], p1 = lrp[i + 1];
        unsigned col = 0;
        float val = 0.f;
        if (tile_nnz > 0) {                         // uniform; the loads are unconditional and index-clamped
            const int p = max(0, min(p0 + j, tile_nnz - 1));
            col = P.lcol[nnz_off + p];
            val = P.lval[nnz_off + p];
        }
        if (j == 0 && p1 - p0 > 2) atomicMax(tile_w, p1 - p0);
        const bool live = p0 + j < p1;
        const unsigned pos = live ? col : (unsigned)i;
        ell_idx[t] = (unsigned short)(BYTEOFF ? pos * (unsigned)P.row_bytes : pos);
        ell_val[t] = live ? val : 0.f;
    }
    __syncthreads();   // ELL complete (and lrp in bufT dead) before the first phase 1
    const int Wt = *tile_w;                         // gather loop length of this tile: its longest row (<= W, may be odd)

    for (int b = b_begin; b < b_end; ++b) {
        unsigned char* bufX = ((b - b_begin) & 1) ? bufX1 : bufX0;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * P.row_bytes + cb) = su[k];
        }
        __syncthreads();   // bufX(b) complete; every wave is past phase 2 of sample b-1 (bufT reusable)
        const size_t sample = c_cur.off;
        if (b + 1 < b_end) advance(c_cur);   // tail: the next-sample burst re-reads this sample (harmless)
        // burst: this sample's epilogue operands first, then the next sample's U rows
        // NS1 / NS2: slots that can hold an S1 row / a tile row (ceil(max_n1 / rpp), ceil(tile_rows / rpp))
        u32x4 cz1[HZA ? NS1 : 1], cz1b[(HZA && HZB) ? NS1 : 1], cz2[HZ2 ? NS2 : 1];
        if constexpr (HZA) {
#pragma unroll
            for (int k = 0; k < NS1; ++k) cz1[k] = *reinterpret_cast<const u32x4*>(P.Z1 + sample + offZ1[k]);
        }
        if (HZA && HZB && P.Z1b != P.Z1) {   // uniform; Z1b aliases Z1 (weight 0) when the caller has only one
#pragma unroll
            for (int k = 0; k < NS1; ++k) cz1b[k] = *reinterpret_cast<const u32x4*>(P.Z1b + sample + offZ1[k]);
        }
        if constexpr (HZ2) {
#pragma unroll
            for (int k = 0; k < NS2; ++k) cz2[k] = ld16_once<u32x4>(P.Z2 + sample + offZ2[k]);
        }
        {
            const size_t sb = c_cur.off;
#pragma unroll
            for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.U + sb + offU[k]);
        }

        // ---- phase 1: Y1 on S1
#pragma unroll
        for (int k = 0; k < NS1; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < n1) {
                VT acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = R::splat(0.f);
                gather_ell<BF16, BYTEOFF, GBATCH>(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, (unsigned)P.row_bytes, bufX + cb, acc);
                VT o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = R::splat(P.a1) * acc[j];
                if constexpr (HZA) {
                    VT z[N];
                    R::unpack(__builtin_bit_cast(uint4, cz1[k]), z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.b1), z[j], o[j]);
                }
                if (HZA && HZB && P.Z1b != P.Z1) {
                    VT z[N];
                    R::unpack(__builtin_bit_cast(uint4, cz1b[k]), z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.d1), z[j], o[j]);
                }
                const uint4 packed = R::pack(o);
                *reinterpret_cast<uint4*>(bufT + (size_t)i * P.row_bytes + cb) = packed;
                if (k < NS2 && P.Y1 != nullptr && i < rt)
                    st16(P.Y1 + sample + offZ2[k < NS2 ? k : 0], packed);
            }
        }
        __syncthreads();

        // ---- phase 2: Y2 on the tile rows
#pragma unroll
        for (int k = 0; k < NS2; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < rt) {
                VT acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = R::splat(0.f);
                gather_ell<BF16, BYTEOFF, GBATCH>(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, (unsigned)P.row_bytes, bufT + cb, acc);
                VT u[N], o[N];
                R::unpack(*reinterpret_cast<const uint4*>(bufX + (size_t)i * P.row_bytes + cb), u);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.a2), acc[j], R::splat(P.b2) * u[j]);
                if constexpr (HZ2) {
                    VT z[N];
                    R::unpack(__builtin_bit_cast(uint4, cz2[k]), z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.c2), z[j], o[j]);
                }
                st16(P.Y2 + sample + offZ2[k], R::pack(o));
            }
        }
        // double-buffered: no barrier here - the next iteration writes the OTHER bufX, and its barrier orders bufT reuse
        if (P.single_buf) __syncthreads();   // phase 2 still read this bufX (the b2 * U term)
    }
}

}  // namespace

// LDS bytes the kernel needs for this plan and row size (must match the carve-up above)
// ELL width: the longest local row is not in the plan struct; bound it by max_nnz / rows is useless, so the
// plan carries it in `reserved` (hop2.py); 0 = unknown -> not supported.
static int hop2_ell_w(const dsw_hop2_plan* plan) { return (plan->reserved + 3) & ~3; }

static size_t hop2_lds_bytes(const dsw_hop2_plan* plan, int row_bytes, bool single_buf = false) {
    size_t s = (size_t)(plan->max_n1 + (single_buf ? 1 : 2) * (size_t)plan->max_n2) * row_bytes;   // bufT + bufX (x2)
    s += (size_t)plan->max_n1 * hop2_ell_w(plan) * 6;  // ELL: fp32 values + u16 list positions
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;   // row ids + the tile's loop length
    return (s + 15) & ~(size_t)15;
}

// 1 if the fused kernel can run this plan / shape (LDS fits, rows are whole 16-byte lanes)
int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype) {
    if (!plan || plan->struct_bytes != (int64_t)sizeof(dsw_hop2_plan)) return 0;   // (another version of the header: see dsw_hip.h)
    if (plan->hops == 1 || plan->n_tiles <= 0 || plan->reserved <= 0) return 0;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    int64_t row_bytes = C * es;
    if (row_bytes % 16 != 0) return 0;
    if (row_bytes > 128 && row_bytes % 128 == 0) row_bytes = 128;   // wide rows: one 128-byte channel chunk at a time
    if (row_bytes / 16 > NTHREADS) return 0;
    const int64_t rpp = NTHREADS / (row_bytes / 16);
    if ((plan->max_n2 + rpp - 1) / rpp > MAXST) return 0;   // register staging capacity
    return hop2_lds_bytes(plan, (int)row_bytes, true) <= 160 * 1024 ? 1 : 0;
}

namespace {
template <bool BF16, int NST, int NS1, int NS2, bool BYTEOFF = false>
int launch_h2(const Hop2Args& A, long nwg, size_t lds, hipStream_t stream) {
    const int sel = ((A.Z1 || A.Z1b) ? 1 : 0) | (A.Z2 ? 2 : 0);
    if constexpr (BYTEOFF) {   // the exact-slot variants also exist without the second first-hop operand
        if (NS1 > 2 && (A.Z1 || A.Z1b) && A.Z1b == A.Z1) {
#define DSW_H2_NOB(Z2_)                                                                                          \
    do {                                                                                                         \
        if (lds > 64 * 1024 &&                                                                                   \
            hipFuncSetAttribute((const void*)spmm2_fused_kernel<BF16, NST, true, Z2_, NS1, NS2, BYTEOFF, false>,          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
            return DSW_ERR_LAUNCH;                                                                               \
        DSW_LAUNCH((spmm2_fused_kernel<BF16, NST, true, Z2_, NS1, NS2, BYTEOFF, false>), dim3((unsigned)nwg),     \
                           dim3(NTHREADS), lds, stream, A);                                                      \
    } while (0)
            if (A.Z2) DSW_H2_NOB(true); else DSW_H2_NOB(false);
#undef DSW_H2_NOB
            return dsw_check_launch();
        }
    }
#define DSW_H2_SEL(S, ZA_, Z2_)                                                                                  \
    case S: {                                                                                                    \
        if (lds > 64 * 1024 &&                                                                                   \
            hipFuncSetAttribute((const void*)spmm2_fused_kernel<BF16, NST, ZA_, Z2_, NS1, NS2, BYTEOFF>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
            return DSW_ERR_LAUNCH;                                                                               \
        DSW_LAUNCH((spmm2_fused_kernel<BF16, NST, ZA_, Z2_, NS1, NS2, BYTEOFF>), dim3((unsigned)nwg),             \
                           dim3(NTHREADS), lds, stream, A);                                                      \
        break;                                                                                                   \
    }
    switch (sel) {
        DSW_H2_SEL(0, false, false)
        DSW_H2_SEL(1, true, false)
        DSW_H2_SEL(2, false, true)
        DSW_H2_SEL(3, true, true)
    }
#undef DSW_H2_SEL
    return dsw_check_launch();
}
}  // namespace

int dsw_spmm2_launch(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z1, const void* Z1b,
                     const void* Z2, void* Y1, void* Y2, int64_t B, int64_t C, float a1, float b1, float d1,
                     float a2, float b2, float c2, int dtype, hipStream_t stream) {
    if (!dsw_spmm2_supported(plan, C, dtype)) return DSW_ERR_BAD_ARG;
    if (V <= 0 || B <= 0) return DSW_OK;
    if (!U || !Y2) return DSW_ERR_BAD_ARG;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    Hop2Args A;
    A.tile_meta = plan->tile_meta; A.s2_rows = plan->s2_rows; A.lrowptr = plan->lrowptr;
    A.lcol = plan->lcol; A.lval = plan->lval;
    A.U = static_cast<const char*>(U); A.Z1 = static_cast<const char*>(Z1);
    A.Z1b = static_cast<const char*>(Z1b); A.Z2 = static_cast<const char*>(Z2);
    A.Y1 = static_cast<char*>(Y1); A.Y2 = static_cast<char*>(Y2);
    A.a1 = a1; A.b1 = Z1 ? b1 : 0.f; A.d1 = Z1b ? d1 : 0.f; A.a2 = a2; A.b2 = b2; A.c2 = Z2 ? c2 : 0.f;
    if (!A.Z1 && A.Z1b) { A.Z1 = A.Z1b; A.b1 = A.d1; A.Z1b = nullptr; A.d1 = 0.f; }
    if (A.Z1 && !A.Z1b) A.Z1b = A.Z1;   // the kernel skips the second operand when both alias
    A.V = (int)V; A.n_tiles = plan->n_tiles; A.tile_rows = plan->tile_rows; A.explicit_tiles = plan->explicit_tiles;
    A.max_n1 = plan->max_n1; A.max_n2 = plan->max_n2; A.max_nnz = plan->max_nnz;
    A.row_stride = (int)(C * es);
    A.row_bytes = (A.row_stride > 128 && A.row_stride % 128 == 0) ? 128 : A.row_stride;
    A.ncc = A.row_stride / A.row_bytes;
    B *= A.ncc;                                            // from here on B counts (sample, channel chunk) pairs
    A.lpr = A.row_bytes / 16; A.B = (int)B;
    A.ell_w = hop2_ell_w(plan);
    // staging buffers: double-buffered input rows unless that costs the second resident workgroup (<= 80 KiB each) or
    // does not fit at all - the k = 20 stencil's 2-ring is 4x the tile, and 8 waves per CU cannot hide the LDS latency
    const size_t lds2 = hop2_lds_bytes(plan, A.row_bytes, false), lds1 = hop2_lds_bytes(plan, A.row_bytes, true);
    const int rpp_ = NTHREADS / A.lpr;
    // see __launch_bounds__: only the exact-slot variants (byte offsets) come without the Z1b registers
    const bool byteoff_ = (long)plan->max_n2 * A.row_bytes <= 65535;
    const bool two_wg_regs = !((A.Z1 || A.Z1b) && (plan->max_n1 + rpp_ - 1) / rpp_ > 2 && !(byteoff_ && A.Z1b == A.Z1));
    A.single_buf = (lds2 <= 80 * 1024) ? 0 : (lds1 <= 80 * 1024 && two_wg_regs) ? 1 : (lds2 <= 160 * 1024) ? 0 : 1;
    { static const char* sb = dsw_diag_env("DSW_H2_SINGLE"); if (sb) A.single_buf = (sb[0] == '1') && lds1 <= 160 * 1024 ? 1 : (lds2 <= 160 * 1024 ? 0 : 1); }   // diagnostics
    const size_t lds = A.single_buf ? lds1 : lds2;
    // batch chunks.  A workgroup = (tile, chunk of the batch); the launch runs in ceil(n_tiles * chunks / slots) rounds
    // over the resident slots, each round costing the plan staging (about 1.5 samples' worth) plus the samples of a
    // chunk: pick the chunk count that minimises rounds * (1.5 + samples per chunk).  (NS: 768 tiles on 512 slots -> 2
    // chunks = exactly 3 rounds, 91 us; the former "about 4 rounds" rule gave 3 chunks = 4.5 rounds, 102 us.)
    long per_cu = (160 * 1024) / (long)lds > 0 ? (160 * 1024) / (long)lds : 1;
    if (!two_wg_regs) per_cu = 1;
    if (per_cu > 2) per_cu = 2;                              // 512-thread workgroups at <= 128 registers: two per CU
    const long slots = 256L * per_cu;
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;       // >= 2 samples per workgroup: the register double buffer
        for (long c = 1; c <= cmax && c <= 16; ++c) {
            const long rounds = (plan->n_tiles * c + slots - 1) / slots;
            const double cost = (double)rounds * (1.5 + (double)((B + c - 1) / c));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = c; }
        }
    }
    { static const char* ce = dsw_diag_env("DSW_H2_CHUNKS"); if (ce) chunks = atol(ce); }   // diagnostics
    if (chunks < 1) chunks = 1;
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return DSW_ERR_BAD_ARG;
    const int rpp = NTHREADS / A.lpr;
    const int nst = (plan->max_n2 + rpp - 1) / rpp;
    const int ns1 = (plan->max_n1 + rpp - 1) / rpp;
    const int ns2 = (plan->tile_rows + rpp - 1) / rpp;
    // the common shapes (64-row tiles, 8 lanes per row) get exact slot counts - fewer live registers in the adjoint
    // variants - and byte-offset ELL entries: k = 8 (158 / 108 / 64 rows = 3 / 2 / 1 passes of 64 rows) and k = 20
    // (280 / 155 / 64 rows = 5 / 3 / 1); everything else sizes all three by NST
    const bool byteoff = (long)plan->max_n2 * A.row_bytes <= 65535;
    if (byteoff && nst == 3 && ns1 <= 2 && ns2 <= 1)
        return dtype == DSW_BF16 ? launch_h2<true, 3, 2, 1, true>(A, nwg, lds, stream) : launch_h2<false, 3, 2, 1, true>(A, nwg, lds, stream);
    if (byteoff && (nst == 4 || nst == 5) && ns1 <= 3 && ns2 <= 1)
        return dtype == DSW_BF16 ? launch_h2<true, 5, 3, 1, true>(A, nwg, lds, stream) : launch_h2<false, 5, 3, 1, true>(A, nwg, lds, stream);
#define DSW_H2_NST(N_)                                                                                     \
    case N_:                                                                                               \
        return dtype == DSW_BF16 ? launch_h2<true, N_, N_, N_>(A, nwg, lds, stream)                        \
                                 : launch_h2<false, N_, N_, N_>(A, nwg, lds, stream);
    switch (nst) {
        DSW_H2_NST(1) DSW_H2_NST(2) DSW_H2_NST(3) DSW_H2_NST(4)
        DSW_H2_NST(5) DSW_H2_NST(6) DSW_H2_NST(7) DSW_H2_NST(8)
    }
#undef DSW_H2_NST
    return DSW_ERR_BAD_ARG;
}
