// Whole ConvCheb forward of a K = 3, 32-input-channel fp32 layer in ONE launch:
//
//     T1 = L X,   T2 = 2 L T1 - X            (layers.py:163-169)
//     Y  = [X | T1 | T2] W + bias            (layers.py:171-178, :375)
//
// The two-hop SpMM of dsw_spmm2.hip (same tile plan, same LDS staging of the tile's 2-ring, same prefetch scheme)
// followed, per (tile, sample), by the channel mix on the matrix cores while X, T1 and T2 of the tile rows are still
// in LDS.  The unfused sequence reads X, T1, T2 back from HBM in the mix kernel (3 of its 5 tensor passes); here the
// launch moves X in, T1 / T2 out (once, for backward; not at all when the caller passes T = NULL) and Y out.
//
// Matrix part.  v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bitwise an fmaf chain, no operand splitting) in the
// SWAPPED orientation  Y^T = W^T T^T :  the "A" operand is the W fragment (16 output channels of the wave's column
// block), held in REGISTERS for the lifetime of the workgroup (3 planes x 8 k-steps = 24 VGPRs - no LDS for the
// panel, so two workgroups still share a CU), the "B" operand is read from the staged rows (lane = row, 16 bytes per
// read feed four k-steps), and the accumulator comes out as 4 CONSECUTIVE output channels of one row per lane:
// 16-byte stores, 64 contiguous bytes per row and instruction.  The fp32 MFMA runs at 1/16 of the bf16 rate (61 us of
// matrix-pipe time per launch at the north-star shape), but the pipe is otherwise idle in this latency-bound kernel
// and the 3-way bf16 split of the first version cost more VALU issue time (every wave re-split the rows it shares
// with the other column blocks: +32 M VALU instructions, 181 us) than the matrix time it saved.
// LDS rows stay 128 bytes apart; the 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7).  The row
// gathers of the hops read all 8 chunks of a row (any order is as good as another; the ELL entry carries the row's
// swizzle so that the address is ONE v_xad_u32), and the MFMA operand reads - 16 rows x one chunk column per lane
// group, a 4-way bank conflict in the plain layout - become conflict-free.  T2 of the tile rows is parked in the dead
// halo rows of the current input buffer, so the mix needs no LDS of its own and only ONE extra barrier per sample.
#include <cstdlib>
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

int dsw_spmm2_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);

namespace {

constexpr int NTHREADS = 512;
constexpr int RB = 128;    // bytes of one activation row in HBM (32 fp32 channels)
constexpr int LS = 128;    // row stride in LDS (16-byte chunks XOR-swizzled by the row, see above)
constexpr int LPR = 8;     // 16-byte lanes per row
constexpr int RPP = 64;    // rows per pass of the 512 threads
#ifndef DSW_FWD3_GB
#define DSW_FWD3_GB 6
#endif
constexpr int GB = DSW_FWD3_GB;   // gathered rows in flight per thread
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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
};

// byte offset of 16-byte chunk position (c ^ swizzle(row)) given c * 16 = cb: cb ^ swz(row)
static __device__ __forceinline__ unsigned swz(const int row) { return (unsigned)((row >> 1) & 7) << 4; }

// acc += sum_j val[j] * buf[off[j]] over the first W entries {byte offset, value} of one ELL row (see dsw_spmm2.hip)
// entries carry  row * 128 + swizzle(row);  the lane's chunk of that row is at  buf + (entry ^ cb)
static __device__ __forceinline__ void gather_ell(const uint2* __restrict__ row_ent, const int W,
                                                  const unsigned char* __restrict__ buf, const unsigned cb,
                                                  float (&acc)[4]) {
    const uint4* e4 = reinterpret_cast<const uint4*>(row_ent);   // {off0, val0, off1, val1}
    int j = 0;
    // 4 rows per batch (the two-hop kernel takes 8): the W fragments of the mix live in registers for the whole
    // workgroup, and a spilled address costs a scratch load + vmcnt(0) in the middle of the prefetch window
    for (; j + GB <= W; j += GB) {
        uint4 e[GB / 2], d[GB];
#pragma unroll
        for (int t = 0; t < GB / 2; ++t) e[t] = e4[(j >> 1) + t];
#pragma unroll
        for (int t = 0; t < GB / 2; ++t) {
            d[2 * t] = *reinterpret_cast<const uint4*>(buf + (e[t].x ^ cb));
            d[2 * t + 1] = *reinterpret_cast<const uint4*>(buf + (e[t].z ^ cb));
        }
#pragma unroll
        for (int t = 0; t < GB / 2; ++t) {
            const float v0 = __uint_as_float(e[t].y), v1 = __uint_as_float(e[t].w);
            const uint32_t a[4] = {d[2 * t].x, d[2 * t].y, d[2 * t].z, d[2 * t].w};
            const uint32_t b[4] = {d[2 * t + 1].x, d[2 * t + 1].y, d[2 * t + 1].z, d[2 * t + 1].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = fmaf(v0, __uint_as_float(a[c]), acc[c]);
                acc[c] = fmaf(v1, __uint_as_float(b[c]), acc[c]);
            }
        }
    }
    for (; j + 2 <= W; j += 2) {   // W is even
        const uint4 ea = e4[j >> 1];
        const uint4 d0 = *reinterpret_cast<const uint4*>(buf + (ea.x ^ cb));
        const uint4 d1 = *reinterpret_cast<const uint4*>(buf + (ea.z ^ cb));
        const float v0 = __uint_as_float(ea.y), v1 = __uint_as_float(ea.w);
        const uint32_t a[4] = {d0.x, d0.y, d0.z, d0.w}, b[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = fmaf(v0, __uint_as_float(a[c]), acc[c]);
            acc[c] = fmaf(v1, __uint_as_float(b[c]), acc[c]);
        }
    }
}

// NST / NS1: register-stage slots per thread for the gather list (ceil(max_n2 / 64)) and for S1 (ceil(max_n1 / 64));
// the tile is 64 rows = one slot.  NCB = Fout / 16 column blocks; a wave owns ONE column block (its W fragments stay in
// registers) and RBW = NCB / 2 of the four 16-row blocks of the tile.
template <int NST, int NS1, int NCB>
__global__ __launch_bounds__(NTHREADS, 4) void cheb3_fwd_fused_kernel(const Fwd3Args P) {
    constexpr int RBW = NCB / 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
    unsigned char* bufX0 = lds;                                             // [max_n2][LS]
    unsigned char* bufX1 = bufX0 + (size_t)P.max_n2 * LS;                   // [max_n2][LS]
    unsigned char* bufT = bufX1 + (size_t)P.max_n2 * LS;                    // [max_n1][LS]
    uint2* ell = reinterpret_cast<uint2*>(bufT + (size_t)P.max_n1 * LS);    // [max_n1][W] {byte offset, value}
    int* rows = reinterpret_cast<int*>(ell + (size_t)P.max_n1 * P.ell_w);   // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);

    const long nwg = gridDim.x, orig = blockIdx.x;                          // XCD-aware order (see dsw_spmm2.hip)
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], n1 = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];
    const int r0 = tile * 64;
    const int rt = min(64, P.V - r0);
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t sample_bytes = (size_t)P.V * RB;

    int* lrp = reinterpret_cast<int*>(bufT);
    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTHREADS) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= n1; i += NTHREADS) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    const int grp = tid >> 3;                       // row of a 64-row pass
    const unsigned cb = (unsigned)(tid & 7) * 16;   // byte offset of this lane inside a row
    unsigned offU[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) offU[k] = (unsigned)rows[min(grp + k * RPP, n2 - 1)] * (unsigned)RB + cb;

    u32x4 su[NST];
    if (b_begin < b_end) {
        const size_t sb = (size_t)b_begin * sample_bytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.X + sb + offU[k]);
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
        ell[t] = (p0 + j < p1) ? make_uint2(col * (unsigned)LS + swz((int)col), __float_as_uint(val))
                               : make_uint2((unsigned)i * (unsigned)LS + swz(i), 0u);
    }

    // ---- W fragments of this wave's column block, split once: lane l holds W[f = 8 (l >> 4) + j][plane][n = 16 cbk + (l & 15)]
    const int wave = tid >> 6, lane = tid & 63;
    const int cbk = wave % NCB;
    const int rb0 = (wave / NCB) * RBW;
    const int l15 = lane & 15, kc = lane >> 4;
    // k-step (s, q, t) contracts channel 16 q + 4 (lane >> 4) + t of plane s: a lane's 16-byte operand read feeds 4 steps
    float wreg[3][2][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                wreg[s][q2][t] = P.W[((size_t)(16 * q2 + 4 * kc + t) * 3 + s) * P.Fout + 16 * cbk + l15];
    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias != nullptr) bias4 = *reinterpret_cast<const f32x4_t*>(P.bias + 16 * cbk + 4 * kc);

    __syncthreads();   // ELL complete (and lrp in bufT dead)
    const int Wt = (*tile_w + 1) & ~1;

    for (int b = b_begin; b < b_end; ++b) {
        unsigned char* bufX = ((b - b_begin) & 1) ? bufX1 : bufX0;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * RPP;
            if (i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * LS + (cb ^ swz(i))) = su[k];
        }
        __syncthreads();   // A: bufX(b) complete; everybody is past the mix of sample b-1 (bufT reusable)
        const size_t sample = (size_t)b * sample_bytes;
        {   // next sample's input rows: in flight under all three phases
            const size_t sb = (size_t)(b + 1 < b_end ? b + 1 : b) * sample_bytes;
#pragma unroll
            for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.X + sb + offU[k]);
        }
        // ---- phase 1: T1 = L X on S1
#pragma unroll
        for (int k = 0; k < NS1; ++k) {
            const int i = grp + k * RPP;
            if (i < n1) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                gather_ell(ell + (size_t)i * W, Wt, bufX, cb, acc);
                const uint4 packed = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                __float_as_uint(acc[3]));
                *reinterpret_cast<uint4*>(bufT + (size_t)i * LS + (cb ^ swz(i))) = packed;
                if (P.T1 != nullptr && i < rt)
                    *reinterpret_cast<uint4*>(P.T1 + sample + (size_t)(r0 + i) * RB + cb) = packed;
            }
        }
        __syncthreads();   // B
        // ---- phase 2: T2 = 2 L T1 - X on the tile rows; parked in rows 64.. of bufX (halo rows, dead after phase 1)
        {
            const int i = grp;
            if (i < rt) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                gather_ell(ell + (size_t)i * W, Wt, bufT, cb, acc);
                const uint4 u = *reinterpret_cast<const uint4*>(bufX + (size_t)i * LS + (cb ^ swz(i)));
                const uint4 packed = make_uint4(__float_as_uint(fmaf(2.f, acc[0], -__uint_as_float(u.x))),
                                                __float_as_uint(fmaf(2.f, acc[1], -__uint_as_float(u.y))),
                                                __float_as_uint(fmaf(2.f, acc[2], -__uint_as_float(u.z))),
                                                __float_as_uint(fmaf(2.f, acc[3], -__uint_as_float(u.w))));
                *reinterpret_cast<uint4*>(bufX + (size_t)(64 + i) * LS + (cb ^ swz(i))) = packed;   // swz(64 + i) == swz(i)
                if (P.T2 != nullptr)
                    *reinterpret_cast<uint4*>(P.T2 + sample + (size_t)(r0 + i) * RB + cb) = packed;
            }
        }
        __syncthreads();   // C
        // ---- phase 3: Y[tile rows, 16 cbk ..+16] = [X | T1 | T2] W + bias on the matrix cores
        const unsigned char* plane[3] = {bufX, bufT, bufX + (size_t)64 * LS};
        f32x4_t acc[RBW];
        unsigned rowoff[RBW], rswz[RBW];
#pragma unroll
        for (int r = 0; r < RBW; ++r) {
            const int row = 16 * (rb0 + r) + l15;
            acc[r] = bias4;
            rowoff[r] = (unsigned)row * LS;
            rswz[r] = swz(row);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                f32x4_t x[RBW];
#pragma unroll
                for (int r = 0; r < RBW; ++r)
                    x[r] = *reinterpret_cast<const f32x4_t*>(plane[s] + rowoff[r] + (((unsigned)(4 * q2 + kc) << 4) ^ rswz[r]));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < RBW; ++r)   // independent accumulators interleaved: no dependent-MFMA stall
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s][q2][t], x[r][t], acc[r], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < RBW; ++r) {   // lane: row 16 (rb0 + r) + l15, output channels 16 cbk + 4 kc .. + 3
            const int row = 16 * (rb0 + r) + l15;
            if (P.relu) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = acc[r][t] < 0.f ? 0.f : acc[r][t];   // NaN stays NaN (torch.relu)
            }
            if (row < rt)
                *reinterpret_cast<f32x4_t*>(P.Y + ((size_t)b * P.V + (size_t)(r0 + row)) * (size_t)P.Fout * 4 +
                                            (size_t)(16 * cbk + 4 * kc) * 4) = acc[r];
        }
        // no barrier: the next iteration fills the OTHER input buffer, and its barrier A orders the reuse of bufT
    }
}

size_t fwd3_lds_bytes(const dsw_hop2_plan* plan) {
    const int ell_w = (plan->reserved + 3) & ~3;
    size_t s = (size_t)(plan->max_n1 + 2 * (size_t)plan->max_n2) * LS;
    s += (size_t)plan->max_n1 * ell_w * 8;
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;
    return (s + 15) & ~(size_t)15;
}

template <int NST, int NS1>
int launch_ncb(const Fwd3Args& A, long nwg, size_t lds, hipStream_t stream) {
#define DSW_F3(N_)                                                                                                      \
    case N_: {                                                                                                          \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)cheb3_fwd_fused_kernel<NST, NS1, N_>,                   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return DSW_ERR_LAUNCH;                                                                                      \
        hipLaunchKernelGGL((cheb3_fwd_fused_kernel<NST, NS1, N_>), dim3((unsigned)nwg), dim3(NTHREADS), lds, stream, A); \
        break;                                                                                                          \
    }
    switch (A.Fout / 16) {
        DSW_F3(2) DSW_F3(4) DSW_F3(8)
        default: return DSW_ERR_BAD_ARG;
    }
#undef DSW_F3
    return dsw_check_launch();
}

}  // namespace

// Runs the whole forward in one launch if the shape / plan allow it.  Returns 1 if it took the call (*rc = status),
// 0 if the caller must use the generic sequence (basis launches + channel-mix GEMM).
int dsw_cheb3_fwd_fused_try(const dsw_hop2_plan* plan, int64_t V, const void* X, const void* W, const void* bias, void* Y,
                            void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                            int* rc, int relu) {
    static const char* env = getenv("DSW_FWD_FUSED");   // "0": generic sequence (diagnostics / A-B)
    if (env && env[0] == '0') return 0;
    if (dtype != DSW_F32 || K != 3 || Fin != 32 || (Fout != 32 && Fout != 64)) return 0;   // (Fout = 128 compiles but spills)
    if (!plan || plan->tile_rows != 64 || !dsw_spmm2_supported(plan, Fin, dtype)) return 0;
    if (plan->max_n2 < 128) return 0;                       // T2 is parked in rows 64..127 of the input buffer
    if (!dsw_aligned16(X) || !dsw_aligned16(Y) || !dsw_aligned16(W) || (bias && !dsw_aligned16(bias)) ||
        (T && !dsw_aligned16(T)))
        return 0;
    const size_t lds = fwd3_lds_bytes(plan);
    if (lds > 80 * 1024) return 0;                          // two workgroups per CU or not at all
    const int nst = (plan->max_n2 + RPP - 1) / RPP, ns1 = (plan->max_n1 + RPP - 1) / RPP;
    if (nst > 4 || ns1 > nst) return 0;
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
    if (nst == 3 && ns1 == 2) r = launch_ncb<3, 2>(A, nwg, lds, stream);
    else if (nst == 2 && ns1 <= 2) r = launch_ncb<2, 2>(A, nwg, lds, stream);
    else if (nst == 3) r = launch_ncb<3, 3>(A, nwg, lds, stream);
    else r = launch_ncb<4, 4>(A, nwg, lds, stream);
    *rc = r;
    return 1;
}
