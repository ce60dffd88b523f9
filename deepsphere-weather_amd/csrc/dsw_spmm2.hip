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
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

namespace {

constexpr int NTHREADS = 512;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs

template <bool BF16>
struct Row16 {  // 16 bytes of a row: 4 fp32 or 8 bf16, widened to fp32 registers
    static constexpr int N = BF16 ? 8 : 4;
    static __device__ __forceinline__ void unpack(const uint4 t, float (&v)[N]) {
        if constexpr (BF16) {
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
            v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
            v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
        }
    }
    static __device__ __forceinline__ uint4 pack(const float (&v)[N]) {
        if constexpr (BF16) {
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
            return make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                              __float_as_uint(v[3]));
        }
    }
};

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
    int row_bytes;              // C * element size (multiple of 16)
    int lpr;                    // 16-byte lanes per row
    int B;
    int n_chunks;               // batch chunks per tile (grid = n_tiles * n_chunks)
    int spc;                    // samples per chunk
};


// acc += sum_p val[p] * buf[col[p]] over the local CSR row [p, e): 4 independent {entry, row} LDS reads
// in flight per batch (the loop is latency-bound: entry -> address -> data -> fma)
template <bool BF16>
static __device__ __forceinline__ void gather_row(const uint2* __restrict__ ent, int p, const int e,
                                                  const unsigned char* __restrict__ buf, const int row_bytes,
                                                  const int cb, float (&acc)[Row16<BF16>::N]) {
    using R = Row16<BF16>;
    constexpr int N = R::N;
    for (; p + 4 <= e; p += 4) {
        const uint2 e0 = ent[p], e1 = ent[p + 1], e2 = ent[p + 2], e3 = ent[p + 3];
        const uint4 d0 = *reinterpret_cast<const uint4*>(buf + (size_t)e0.x * row_bytes + cb);
        const uint4 d1 = *reinterpret_cast<const uint4*>(buf + (size_t)e1.x * row_bytes + cb);
        const uint4 d2 = *reinterpret_cast<const uint4*>(buf + (size_t)e2.x * row_bytes + cb);
        const uint4 d3 = *reinterpret_cast<const uint4*>(buf + (size_t)e3.x * row_bytes + cb);
        float x0[N], x1[N], x2[N], x3[N];
        R::unpack(d0, x0); R::unpack(d1, x1); R::unpack(d2, x2); R::unpack(d3, x3);
        const float v0 = __uint_as_float(e0.y), v1 = __uint_as_float(e1.y);
        const float v2 = __uint_as_float(e2.y), v3 = __uint_as_float(e3.y);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            acc[j] = fmaf(v0, x0[j], acc[j]);
            acc[j] = fmaf(v1, x1[j], acc[j]);
            acc[j] = fmaf(v2, x2[j], acc[j]);
            acc[j] = fmaf(v3, x3[j], acc[j]);
        }
    }
    for (; p < e; ++p) {
        const uint2 e0 = ent[p];
        float x0[N];
        R::unpack(*reinterpret_cast<const uint4*>(buf + (size_t)e0.x * row_bytes + cb), x0);
        const float v0 = __uint_as_float(e0.y);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = fmaf(v0, x0[j], acc[j]);
    }
}

// Rows of U staged per thread and sample (register double buffer): ceil(max_n2 * lpr / NTHREADS) <= MAXST
constexpr int MAXST = 8;

// One workgroup = (tile, chunk of the batch): the tile's plan slice is staged once and reused for every
// sample of the chunk; the next sample's U rows are loaded into registers while the current sample is
// processed out of LDS (bufX is double-buffered).
template <bool BF16>
__global__ __launch_bounds__(NTHREADS) void spmm2_fused_kernel(const Hop2Args P) {
    using R = Row16<BF16>;
    constexpr int N = R::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // LDS carve-up (all offsets multiples of 16)
    unsigned char* bufX0 = lds;                                            // [max_n2][row_bytes], sample s
    unsigned char* bufX1 = bufX0 + (size_t)P.max_n2 * P.row_bytes;         // [max_n2][row_bytes], sample s+1
    unsigned char* bufT = bufX1 + (size_t)P.max_n2 * P.row_bytes;          // [max_n1][row_bytes]
    uint2* ent = reinterpret_cast<uint2*>(bufT + (size_t)P.max_n1 * P.row_bytes);   // [max_nnz] {col, val}
    int* lrp = reinterpret_cast<int*>(ent + ((P.max_nnz + 1) & ~1));       // [max_n1 + 1]
    int* rows = lrp + ((P.max_n1 + 1 + 3) & ~3);                           // [max_n2] global row ids

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
    const int r0 = tile * P.tile_rows;
    const int rt = min(P.tile_rows, P.V - r0);
    const int tid = threadIdx.x;
    const size_t sample_bytes = (size_t)P.V * P.row_bytes;

    // ---- the tile's plan slice -> LDS, once for all samples of this workgroup
    for (int i = tid; i <= n1; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
    for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
    __syncthreads();
    const int nnz = lrp[n1];
    for (int i = tid; i < nnz; i += NTHREADS)
        ent[i] = make_uint2((unsigned)P.lcol[nnz_off + i], __float_as_uint(P.lval[nnz_off + i]));

    const int lpr = P.lpr;
    const int rpp = NTHREADS / lpr;                 // rows per pass
    const int grp = tid / lpr;                      // row slot of this lane group
    const int cb = (tid - grp * lpr) * 16;          // byte offset of this lane inside the row
    const bool lane_ok = grp < rpp;

    // register staging of the U rows of S2 (next sample's loads fly under this sample's compute)
    u32x4 stg[MAXST];
    auto stage_load = [&](int b) __attribute__((always_inline)) {
        const char* src = P.U + (size_t)b * sample_bytes + cb;
#pragma unroll
        for (int k = 0; k < MAXST; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < n2) stg[k] = *reinterpret_cast<const u32x4*>(src + (size_t)rows[i] * P.row_bytes);
        }
    };
    auto stage_store = [&](unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXST; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < n2) *reinterpret_cast<u32x4*>(dst + (size_t)i * P.row_bytes + cb) = stg[k];
        }
    };

    if (b_begin < b_end) stage_load(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
        unsigned char* bufX = ((b - b_begin) & 1) ? bufX1 : bufX0;
        stage_store(bufX);
        __syncthreads();   // bufX(b) complete; every wave is past phase 2 of sample b-1 (bufT reusable)
        if (b + 1 < b_end) stage_load(b + 1);
        const size_t sample = (size_t)b * sample_bytes;

        // ---- phase 1: Y1 on S1
        if (lane_ok) {
            for (int i = grp; i < n1; i += rpp) {
                const size_t goff = sample + (size_t)rows[i] * P.row_bytes + cb;
                uint4 z1 = make_uint4(0, 0, 0, 0), z1b = make_uint4(0, 0, 0, 0);
                if (P.Z1) z1 = *reinterpret_cast<const uint4*>(P.Z1 + goff);      // issued before the gathers
                if (P.Z1b) z1b = *reinterpret_cast<const uint4*>(P.Z1b + goff);
                float acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = 0.f;
                gather_row<BF16>(ent, lrp[i], lrp[i + 1], bufX, P.row_bytes, cb, acc);
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = P.a1 * acc[j];
                if (P.Z1) {
                    float z[N];
                    R::unpack(z1, z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmaf(P.b1, z[j], o[j]);
                }
                if (P.Z1b) {
                    float z[N];
                    R::unpack(z1b, z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmaf(P.d1, z[j], o[j]);
                }
                const uint4 packed = R::pack(o);
                *reinterpret_cast<uint4*>(bufT + (size_t)i * P.row_bytes + cb) = packed;
                if (P.Y1 != nullptr && i < rt)
                    *reinterpret_cast<uint4*>(P.Y1 + sample + (size_t)(r0 + i) * P.row_bytes + cb) = packed;
            }
        }
        __syncthreads();

        // ---- phase 2: Y2 on the tile rows
        if (lane_ok) {
            for (int i = grp; i < rt; i += rpp) {
                const size_t goff = sample + (size_t)(r0 + i) * P.row_bytes + cb;
                uint4 z2 = make_uint4(0, 0, 0, 0);
                if (P.Z2) z2 = *reinterpret_cast<const uint4*>(P.Z2 + goff);
                float acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = 0.f;
                gather_row<BF16>(ent, lrp[i], lrp[i + 1], bufT, P.row_bytes, cb, acc);
                float u[N], o[N];
                R::unpack(*reinterpret_cast<const uint4*>(bufX + (size_t)i * P.row_bytes + cb), u);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = fmaf(P.a2, acc[j], P.b2 * u[j]);
                if (P.Z2) {
                    float z[N];
                    R::unpack(z2, z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmaf(P.c2, z[j], o[j]);
                }
                *reinterpret_cast<uint4*>(P.Y2 + goff) = R::pack(o);
            }
        }
        // no barrier here: the next iteration writes the OTHER bufX, and its barrier orders bufT reuse
    }
}

}  // namespace

// LDS bytes the kernel needs for this plan and row size (must match the carve-up above)
static size_t hop2_lds_bytes(const dsw_hop2_plan* plan, int row_bytes) {
    size_t s = (size_t)(plan->max_n1 + 2 * (size_t)plan->max_n2) * row_bytes;   // bufX double-buffered
    s += (size_t)((plan->max_nnz + 1) & ~1) * 8;
    s += (size_t)((plan->max_n1 + 1 + 3) & ~3) * 4;
    s += (size_t)plan->max_n2 * 4;
    return (s + 15) & ~(size_t)15;
}

// 1 if the fused kernel can run this plan / shape (LDS fits, rows are whole 16-byte lanes)
int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype) {
    if (!plan || plan->n_tiles <= 0) return 0;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    const int64_t row_bytes = C * es;
    if (row_bytes % 16 != 0 || row_bytes / 16 > NTHREADS) return 0;
    const int64_t rpp = NTHREADS / (row_bytes / 16);
    if ((plan->max_n2 + rpp - 1) / rpp > MAXST) return 0;   // register staging capacity
    return hop2_lds_bytes(plan, (int)row_bytes) <= 160 * 1024 ? 1 : 0;
}

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
    A.V = (int)V; A.n_tiles = plan->n_tiles; A.tile_rows = plan->tile_rows;
    A.max_n1 = plan->max_n1; A.max_n2 = plan->max_n2; A.max_nnz = plan->max_nnz;
    A.row_bytes = (int)(C * es); A.lpr = A.row_bytes / 16; A.B = (int)B;
    const size_t lds = hop2_lds_bytes(plan, A.row_bytes);
    // batch chunks: enough workgroups for ~4 rounds over the resident slots, >= 2 samples per workgroup
    // so that the plan staging and the first U load are amortised and the register double buffer pays
    long slots = 256L * ((160 * 1024) / (long)lds > 0 ? (160 * 1024) / (long)lds : 1);
    long chunks = (4 * slots + plan->n_tiles - 1) / plan->n_tiles;
    if (chunks > (B + 1) / 2) chunks = (B + 1) / 2;
    if (chunks < 1) chunks = 1;
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return DSW_ERR_BAD_ARG;
    if (dtype == DSW_BF16) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)spmm2_fused_kernel<true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DSW_ERR_LAUNCH;
        hipLaunchKernelGGL(spmm2_fused_kernel<true>, dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A);
    } else {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)spmm2_fused_kernel<false>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DSW_ERR_LAUNCH;
        hipLaunchKernelGGL(spmm2_fused_kernel<false>, dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A);
    }
    return dsw_check_launch();
}
