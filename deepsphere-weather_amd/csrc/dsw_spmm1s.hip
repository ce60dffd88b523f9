// LDS-staged single-hop SpMM for DENSE stencils (the reference's default graph: k = 20 neighbours, 21-23 entries per
// row; modules/utils_config.py:50), per sample:
//
//     Y = a * (A U) + b * Z + c * Z2            (one Chebyshev step, layers.py:164,167, or one adjoint step)
//
// Why a second staged kernel next to the fused two-hop one (dsw_spmm2.hip): fusing two hops needs the first hop on
// the whole 1-ring of the tile.  At 9 entries per row (k = 8) that redundancy is 1.6x and the fused pair wins - the
// intermediate never touches HBM.  At 21+ entries the 1-ring of a 64-row tile is 2.2-2.4x the tile and every row task
// costs 21+ LDS row gathers: per output row the pair spends 3.4 row tasks, the LDS pipe is busy 60 % of the launch and
// the vector ALU 43 % (profiles/r03_k20_pmc_summary.txt) - the launch is bound by its on-chip work (0.42 of the HBM
// roofline), not by bytes.  One hop per launch has NO redundant row task (1.0 per output row and hop instead of 1.7),
// stages only tile + 1-ring (128-row tiles: 1.9x the tile from L2, of which ~1.2x misses to HBM / Infinity Cache) and
// keeps the intermediate plane in the 256 MB Infinity Cache between the two launches.
//
// A workgroup owns (tile of R rows) x (chunk of the batch); the host plan (dsw_amd/hop2.py, hops = 1) lists the rows
// the tile gathers (tile rows first) and the tile rows' CSR re-indexed to list positions.  Per sample: the staged rows
// arrive in registers (requested one sample ahead, right after the barrier, behind this sample's epilogue operands -
// vmcnt completes in order), are written to LDS, and every lane group gathers its output rows from LDS.
#include <cstdlib>
#include "dsw_hop_common.h"
#include "../../include/dsw_hip.h"

namespace {
constexpr int NTHREADS1 = 512;

struct Hop1Args {
    const int* tile_meta;       // [n_tiles][6]: list offset, tile rows, list length, nnz offset, row-pointer offset, tile rows
    const int* s2_rows;
    const int* lrowptr;
    const unsigned short* lcol;
    const float* lval;
    const unsigned short* ell_pos;   // padded ELL image of the tile rows' stencils (LDS-DMA kernel), see dsw_hop2_plan
    const float* ell_val;
    const char* U;
    const char* Z;
    const char* Z2;
    char* Y;
    float a, b, c;
    int V, n_tiles, tile_rows, max_rt, max_n2;
    int row_bytes;              // bytes of one STAGED row (whole row, or one 128-byte channel chunk of a wide row)
    int row_stride;             // bytes between consecutive rows in HBM
    int ncc;                    // channel chunks per row; B counts (sample, chunk) pairs
    int lpr;                    // 16-byte lanes per row
    int B, n_chunks, spc;
    int ell_w;
    int explicit_tiles;
    int stream_out;             // 1: nontemporal stores (the result is not gathered by the next launch)
};

// Generic form (any tile height / row length / row width): staged rows through registers, ELL of the tile rows in LDS.
// NST staging slots (ceil(max list length / rows per pass)), NS2 output slots (ceil(tile rows / rows per pass)).
template <bool BF16, int NST, int NS2, bool HZ, bool HZ2, int OCC>
__global__ __launch_bounds__(NTHREADS1, OCC) void spmm1_staged_kernel(const Hop1Args P) {
    using R = Row16<BF16>;
    using VT = typename R::V;
    constexpr int N = R::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bufX = lds;                                                              // [max_n2][row_bytes]
    float* ell_val = reinterpret_cast<float*>(bufX + (size_t)P.max_n2 * P.row_bytes);       // [max_rt][W]
    unsigned short* ell_idx = reinterpret_cast<unsigned short*>(ell_val + (size_t)P.max_rt * P.ell_w);
    int* rows = reinterpret_cast<int*>(ell_idx + (size_t)P.max_rt * P.ell_w);               // [max_n2] global row ids
    int* tile_w = rows + ((P.max_n2 + 3) & ~3);
    int* lrp = tile_w + 4;                                                                  // [max_rt + 1]

    // XCD-aware order: each XCD (hardware block id % 8) walks one contiguous range of (tile, batch chunk)
    const long nwg = gridDim.x, orig = blockIdx.x;
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], rt = meta[1], n2 = meta[2], nnz_off = meta[3], rp_off = meta[4];
    const int tid = threadIdx.x;
    const int W = P.ell_w;
    const size_t sample_bytes = (size_t)P.V * P.row_stride;
    struct Cursor { size_t off; int cc; };   // incremental sample offsets, see spmm1_dma_kernel
    auto cursor_at = [&](const int b) __attribute__((always_inline)) {
        const int bs = b / P.ncc;
        return Cursor{(size_t)bs * sample_bytes + (size_t)(b - bs * P.ncc) * P.row_bytes, b - bs * P.ncc};
    };
    auto advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.off += P.row_bytes;
        if (++c.cc == P.ncc) { c.cc = 0; c.off += sample_bytes - (size_t)P.ncc * P.row_bytes; }
    };
    Cursor c_cur = cursor_at(b_begin);

    if (tid == 0) *tile_w = 2;
    for (int i = tid; i < n2; i += NTHREADS1) rows[i] = P.s2_rows[s2_off + i];
    for (int i = tid; i <= rt; i += NTHREADS1) lrp[i] = P.lrowptr[rp_off + i];
    __syncthreads();

    const int lpr = P.lpr;
    const int rpp = NTHREADS1 / lpr;
    const int grp0 = tid / lpr;
    const bool lane_ok = grp0 < rpp;
    const int grp = lane_ok ? grp0 : rpp - 1;
    const int cb = ((tid - grp0 * lpr) * 16) % P.row_bytes;

    unsigned offU[NST], offY[NS2];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = grp + k * rpp;
        offU[k] = (unsigned)rows[min(i, n2 - 1)] * (unsigned)P.row_stride + cb;
        if (k < NS2) offY[k] = (unsigned)rows[min(i, rt - 1)] * (unsigned)P.row_stride + cb;
    }

    u32x4 su[NST];
    if (b_begin < b_end) {
        const size_t sb = c_cur.off;
#pragma unroll
        for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.U + sb + offU[k]);
    }
    // CSR -> ELL of the tile rows (entries = byte offsets of the staged rows), padding = {own row, weight 0}
    const int tile_nnz = lrp[rt];
    for (int t = tid; t < rt * W; t += NTHREADS1) {
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
        const unsigned pos = live ? col : (unsigned)i;
        ell_idx[t] = (unsigned short)(pos * (unsigned)P.row_bytes);
        ell_val[t] = live ? val : 0.f;
    }
    __syncthreads();
    const int Wt = *tile_w;

    for (int b = b_begin; b < b_end; ++b) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < n2) *reinterpret_cast<u32x4*>(bufX + (size_t)i * P.row_bytes + cb) = su[k];
        }
        __syncthreads();   // bufX(b) complete
        const size_t sample = c_cur.off;
        if (b + 1 < b_end) advance(c_cur);   // tail: the next-sample burst re-reads this sample (harmless)
        // burst: this sample's epilogue operands first, then the next sample's rows
        u32x4 cz[HZ ? NS2 : 1], cz2[HZ2 ? NS2 : 1];
        if constexpr (HZ) {
#pragma unroll
            for (int k = 0; k < NS2; ++k) cz[k] = *reinterpret_cast<const u32x4*>(P.Z + sample + offY[k]);
        }
        if constexpr (HZ2) {
#pragma unroll
            for (int k = 0; k < NS2; ++k) cz2[k] = ld16_once<u32x4>(P.Z2 + sample + offY[k]);
        }
        {
            const size_t sb = c_cur.off;
#pragma unroll
            for (int k = 0; k < NST; ++k) su[k] = *reinterpret_cast<const u32x4*>(P.U + sb + offU[k]);
        }
#pragma unroll
        for (int k = 0; k < NS2; ++k) {
            const int i = grp + k * rpp;
            if (lane_ok && i < rt) {
                VT acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j) acc[j] = R::splat(0.f);
                gather_ell<BF16, true, 8>(ell_idx + (size_t)i * W, ell_val + (size_t)i * W, Wt, (unsigned)P.row_bytes, bufX + cb, acc);
                VT o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = R::splat(P.a) * acc[j];
                if constexpr (HZ) {
                    VT z[N];
                    R::unpack(__builtin_bit_cast(uint4, cz[k]), z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.b), z[j], o[j]);
                }
                if constexpr (HZ2) {
                    VT z[N];
                    R::unpack(__builtin_bit_cast(uint4, cz2[k]), z);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.c), z[j], o[j]);
                }
                const uint4 packed = R::pack(o);
                if (P.stream_out) st16(P.Y + sample + offY[k], packed);
                else *reinterpret_cast<uint4*>(P.Y + sample + offY[k]) = packed;
            }
        }
        __syncthreads();   // every wave is done reading bufX(b)
    }
}

// LDS-DMA form of the same launch for the shape that matters (the k = 20 stencil on 64-row tiles: one output row per
// lane group, <= 24 entries per row - 32 on samplings with a few longer rows, e.g. equiangular k = 20 near the poles:
// WREG -, rows of 16 / 32 / 64 / 128 bytes so that the wave's lanes are LDS-linear).
//   * the staged rows travel HBM/L2 -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging
//     registers, no ds_write pass - ds_write_b128 costs 13 LDS cycles per KiB against 4 per KiB read) into a RING OF
//     THREE buffers, two samples ahead of the gather: the first measured form (two buffers, one sample ahead) had 58 % of
//     the wave cycles waiting and moved 3.4 TB/s - one sample's gather (~0.5 us) cannot cover a loaded-memory latency
//     (2-3 us), and 2 workgroups x 20 KB in flight per CU is what Little's law gives 3.5 TB/s for.  ONE barrier per sample;
//   * the waits are COUNTED by hand (s_waitcnt vmcnt(N), N = the vector-memory operations this wave issued after the
//     ones it needs: the next sample's DMA pieces and the previous store); the loads are issued from inline asm so that
//     hipcc, which would drain the queue (vmcnt(0)) at the first use of an ordinary load while an LDS-DMA is in flight,
//     does not see them.  vmcnt retires in issue order (what LLVM's own waitcnt insertion assumes on gfx9 for loads and
//     stores alike);
//   * the epilogue operand Z (tile rows) arrives the same way, one sample ahead, in a ring of two 8 KiB buffers: an asm
//     load into a REGISTER would be fair game for the register allocator before its data has arrived; a second epilogue
//     operand (Z2: the middle steps of K >= 4 adjoints) is left to the generic kernel below;
//   * the stencil of the lane group's row sits in registers for every sample of the workgroup (24 weights as register
//     pairs + 12 packed row offsets): the per-sample gather is 24 independent ds_read_b128 in three batches + packed FMAs,
//     with no index / weight reads from LDS and no index -> address -> data chain.  They are loaded from the plan's padded
//     ELL image (dsw_hop2_plan.ell_pos / ell_val); the prologue uses no LDS and no barrier.
// Ring invariant at the top of iteration b: buffer b % 3 holds sample b once this wave's wait AND the barrier are passed;
// every wave has then left the gather of sample b - 1, whose buffer (b + 2) % 3 is refilled right after the barrier.
static __device__ __forceinline__ void glds16(const char* gsrc, const unsigned lds_dst) {
    unsigned keep;   // M0 carries the LDS destination of an LDS-DMA; it is compiler-reserved: saved and restored in ONE statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// wait until at most n (0..4) of this wave's vector-memory operations are outstanding
static __device__ __forceinline__ void wait_vm(const int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

template <bool BF16, int NST, bool HZ, int WREG>
__global__ __launch_bounds__(NTHREADS1, 4) void spmm1_dma_kernel(const Hop1Args P) {
    static_assert(NST <= 3, "wait_vm counts up to NST + 1 operations");
    static_assert(WREG % 8 == 0, "stencil entries are loaded and gathered in batches of 8");
    using R = Row16<BF16>;
    using VT = typename R::V;
    constexpr int N = R::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned buf_bytes = (unsigned)P.max_n2 * (unsigned)P.row_bytes;                  // multiple of 16
    unsigned char* bufX = lds;                                                              // [3][max_n2][row_bytes]
    unsigned char* bufZ = bufX + 3 * (size_t)buf_bytes;                                     // [2][NTHREADS1 * 16] (HZ only)

    const long nwg = gridDim.x, orig = blockIdx.x;
    const long q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const long wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int tile = (int)(wg / P.n_chunks);
    const int chunk = (int)(wg - (long)tile * P.n_chunks);
    const int b_begin = chunk * P.spc;
    const int b_end = min(P.B, b_begin + P.spc);
    const int* meta = P.tile_meta + (size_t)tile * 6;
    const int s2_off = meta[0], rt = meta[1], n2 = meta[2], Wt = meta[5];
    const int tid = threadIdx.x;
    const size_t sample_bytes = (size_t)P.V * P.row_stride;
    // byte offset of "sample" b = (real sample b / ncc, channel chunk b % ncc), advanced incrementally: the scalar unit is
    // shared by the CU's 16 waves, and a 64-bit divide + multiply per offset (three offsets per iteration) had the
    // loop issue as many scalar as vector instructions (SQ_INSTS_SALU = SQ_INSTS_VALU in the first profile)
    struct Cursor { size_t off; int cc; };
    auto cursor_at = [&](const int b) __attribute__((always_inline)) {
        const int bs = b / P.ncc;
        return Cursor{(size_t)bs * sample_bytes + (size_t)(b - bs * P.ncc) * P.row_bytes, b - bs * P.ncc};
    };
    auto advance = [&](Cursor& c) __attribute__((always_inline)) {
        c.off += P.row_bytes;
        if (++c.cc == P.ncc) { c.cc = 0; c.off += sample_bytes - (size_t)P.ncc * P.row_bytes; }
    };

    // lanes per row is a power of two here: thread tid holds bytes [16 tid, 16 tid + 16) of every pass (LDS-linear)
    const int lpr = P.lpr;
    const int rpp = NTHREADS1 / lpr;
    const int grp = tid / lpr;
    const int cb = (tid - grp * lpr) * 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wave_lds = (unsigned)wave << 10;                 // 1 KiB per wave and pass
    const int wave_row0 = (wave << 6) / lpr;                        // first list position of this wave in pass 0

    // Prologue without LDS and without a workgroup barrier: the lane's entries of the gather list and the stencil of its
    // row (padded ELL image of the plan: 6 + 3 independent 16-byte loads, the 8 lanes of a row read the same addresses)
    // come straight from global memory.  (The first form expanded the tile's CSR to ELL in LDS - two dependent rounds of
    // global loads and two barriers per workgroup of 8 samples: an ablation without ANY gather work still took 83 % of
    // the launch time.)
    unsigned offU[NST];
    bool live[NST];
    int n_dma = 0;                                                  // DMA instructions this wave issues per sample (uniform)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = grp + k * rpp;
        live[k] = i < n2;
        offU[k] = (unsigned)P.s2_rows[s2_off + min(i, n2 - 1)] * (unsigned)P.row_stride + cb;
        n_dma += (wave_row0 + k * rpp < n2) ? 1 : 0;
    }
    n_dma = __builtin_amdgcn_readfirstlane(n_dma);
    const bool out_ok = grp < rt;
    const int n_store = __builtin_amdgcn_readfirstlane(wave_row0 < rt ? 1 : 0);   // this wave stores (uniform)
    const unsigned offY = offU[0];                                  // tile rows lead the list (used under out_ok only)
    f32x2 rv[WREG / 2];
    unsigned ra[WREG / 2];
    {
        const size_t e0 = ((size_t)tile * 64 + (size_t)min(grp, 63)) * WREG;
        const float4* pv = reinterpret_cast<const float4*>(P.ell_val + e0);
        const uint4* pp = reinterpret_cast<const uint4*>(P.ell_pos + e0);
#pragma unroll
        for (int j = 0; j < WREG / 4; ++j) {
            const float4 v = pv[j];
            rv[2 * j] = f32x2{v.x, v.y}; rv[2 * j + 1] = f32x2{v.z, v.w};
        }
#pragma unroll
        for (int j = 0; j < WREG / 8; ++j) {
            const uint4 ix = pp[j];     // 8 list positions -> 8 byte offsets of staged rows (< 64 KiB)
            const unsigned rb = (unsigned)P.row_bytes;
            ra[4 * j] = (ix.x & 0xffffu) * rb | ((ix.x >> 16) * rb) << 16;
            ra[4 * j + 1] = (ix.y & 0xffffu) * rb | ((ix.y >> 16) * rb) << 16;
            ra[4 * j + 2] = (ix.z & 0xffffu) * rb | ((ix.z >> 16) * rb) << 16;
            ra[4 * j + 3] = (ix.w & 0xffffu) * rb | ((ix.w >> 16) * rb) << 16;
        }
    }

    auto request = [&](const size_t off, const unsigned buf) __attribute__((always_inline)) {
        const char* src = P.U + off;
        const unsigned dst = (unsigned)reinterpret_cast<uintptr_t>(bufX) + buf * buf_bytes + wave_lds;   // low half of a generic LDS address = LDS offset
#pragma unroll
        for (int k = 0; k < NST; ++k) {
#ifndef DSW_ABL_NODMA
            if (live[k]) glds16(src + offU[k], dst + (unsigned)k * (NTHREADS1 * 16));
#endif
        }
    };
    auto request_z = [&](const size_t off, const unsigned buf) __attribute__((always_inline)) {
        if (out_ok)
            glds16(P.Z + off + offY, (unsigned)reinterpret_cast<uintptr_t>(bufZ) + buf * (NTHREADS1 * 16) + wave_lds);
    };
    // issue order (what wait_vm counts on): U(b0), Z(b0), U(b0 + 1); then per iteration Z(b + 1), U(b + 2), store(b)
    Cursor c0 = cursor_at(b_begin);     // sample b of the loop below
    Cursor c1 = c0; advance(c1);        // b + 1
    Cursor c2 = c1; advance(c2);        // b + 2
    // the plan loads above are ordinary loads: they must have landed before the counted waits start counting
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (b_begin < b_end) {
        request(c0.off, 0);
        if constexpr (HZ) request_z(c0.off, 0);
        if (b_begin + 1 < b_end) request(c1.off, 1);
    }

    unsigned cur = 0;   // ring position of sample b
    for (int b = b_begin; b < b_end; ++b) {
        // this wave's pieces of sample b and its epilogue operands have landed; younger, still in flight: the pieces of
        // sample b + 1 and the store of sample b - 1
        wait_vm((b + 1 < b_end ? n_dma : 0) + (b > b_begin ? n_store : 0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const size_t sample = c0.off;
        const unsigned curz = (unsigned)(b - b_begin) & 1u;
        if constexpr (HZ) {
            if (b + 1 < b_end) request_z(c1.off, curz ^ 1u);
        }
        if (b + 2 < b_end) request(c2.off, cur >= 1 ? cur - 1 : 2);
        c0 = c1; c1 = c2; advance(c2);
        const unsigned char* bx = bufX + cur * buf_bytes + cb;
        cur = cur == 2 ? 0 : cur + 1;
        VT acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = R::splat(0.f);
#ifndef DSW_ABL_NOGATHER
#pragma unroll
        for (int j0 = 0; j0 < WREG; j0 += 8) {
            if (j0 < Wt) {   // uniform: batches beyond the tile's longest row hold padding only
                uint4 d[8];
                // opaque to the optimiser from here on: otherwise the 24 unpacked addresses and the 24 splat weight pairs
                // (loop invariants) are hoisted out of the sample loop - 72 registers more than the packed forms
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(ra[j0 / 2 + t]), "+v"(rv[j0 / 2 + t]));
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    d[t] = *reinterpret_cast<const uint4*>(bx + ((t & 1) ? (ra[(j0 + t) / 2] >> 16) : (ra[(j0 + t) / 2] & 0xffffu)));
                if constexpr (BF16) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        VT x[N];
                        R::unpack(d[t], x);
                        const f32x2 wp = rv[(j0 + t) / 2];
                        const VT v = R::splat((t & 1) ? wp.y : wp.x);
#pragma unroll
                        for (int c = 0; c < N; ++c) acc[c] = fmav(v, x[c], acc[c]);
                    }
                } else {
                    f32x2 a01 = f32x2{acc[0], acc[1]}, a23 = f32x2{acc[2], acc[3]};
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const f32x2 wp = rv[(j0 + t) / 2];
                        const float w = (t & 1) ? wp.y : wp.x;
                        const f32x2 ws = f32x2{w, w};
                        a01 = fmav(ws, f32x2{__uint_as_float(d[t].x), __uint_as_float(d[t].y)}, a01);
                        a23 = fmav(ws, f32x2{__uint_as_float(d[t].z), __uint_as_float(d[t].w)}, a23);
                    }
                    if constexpr (!BF16) { acc[0] = a01.x; acc[1] = a01.y; acc[2] = a23.x; acc[3] = a23.y; }
                }
            }
        }
#endif
        if (out_ok) {
            VT o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = R::splat(P.a) * acc[j];
            if constexpr (HZ) {
                VT z[N];
                R::unpack(*reinterpret_cast<const uint4*>(bufZ + curz * (NTHREADS1 * 16) + tid * 16), z);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = fmav(R::splat(P.b), z[j], o[j]);
            }
            const uint4 packed = R::pack(o);
            if (P.stream_out) st16(P.Y + sample + offY, packed);
            else *reinterpret_cast<uint4*>(P.Y + sample + offY) = packed;
        }
    }
}

// ring of three buffers of staged rows (+ two 8 KiB buffers of epilogue rows)
size_t hop1_dma_lds_bytes(const dsw_hop2_plan* plan, int row_bytes, bool hz) {
    return ((3 * (size_t)plan->max_n2 * row_bytes + (hz ? 2 * (size_t)NTHREADS1 * 16 : 0)) + 15) & ~(size_t)15;
}

size_t hop1_lds_bytes(const dsw_hop2_plan* plan, int row_bytes) {
    const int w = (plan->reserved + 3) & ~3;
    size_t s = (size_t)plan->max_n2 * row_bytes;            // staged rows
    s += (size_t)plan->max_n1 * w * 6;                      // ELL of the tile rows
    s += (size_t)((plan->max_n2 + 3) & ~3) * 4 + 16;        // row ids + loop length
    s += (size_t)(plan->max_n1 + 1) * 4;                    // local row pointers
    return (s + 15) & ~(size_t)15;
}

struct Shape1 { int nst, ns2; };
constexpr Shape1 kShapes[] = {{2, 1}, {3, 1}, {5, 2}, {7, 4}, {8, 4}};

int pick_shape(int nst, int ns2) {
    for (unsigned i = 0; i < sizeof(kShapes) / sizeof(kShapes[0]); ++i)
        if (kShapes[i].nst >= nst && kShapes[i].ns2 >= ns2) return (int)i;
    return -1;
}

template <bool BF16, int NST, int NS2, int OCC>
int launch_h1(const Hop1Args& A, long nwg, size_t lds, hipStream_t stream) {
#define DSW_H1_GO(HZ_, HZ2_)                                                                                      \
    do {                                                                                                          \
        if (lds > 64 * 1024 &&                                                                                    \
            hipFuncSetAttribute((const void*)spmm1_staged_kernel<BF16, NST, NS2, HZ_, HZ2_, OCC>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)              \
            return DSW_ERR_LAUNCH;                                                                                \
        DSW_LAUNCH((spmm1_staged_kernel<BF16, NST, NS2, HZ_, HZ2_, OCC>), dim3((unsigned)nwg),            \
                           dim3(NTHREADS1), lds, stream, A);                                                      \
    } while (0)
    if (A.Z && A.Z2) DSW_H1_GO(true, true);
    else if (A.Z) DSW_H1_GO(true, false);
    else DSW_H1_GO(false, false);
#undef DSW_H1_GO
    return dsw_check_launch();
}
}  // namespace

// 1 if the staged single-hop kernel can run this plan / shape
int dsw_spmm1s_supported(const dsw_hop2_plan* plan, int64_t C, int dtype) {
    if (!plan || plan->hops != 1 || plan->n_tiles <= 0 || plan->reserved <= 0) return 0;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    int64_t row_bytes = C * es;
    if (row_bytes % 16 != 0) return 0;
    if (row_bytes > 128 && row_bytes % 128 == 0) row_bytes = 128;
    if (row_bytes / 16 > NTHREADS1) return 0;
    if ((int64_t)plan->max_n2 * row_bytes > 65535) return 0;          // u16 byte offsets of the staged rows
    const int64_t rpp = NTHREADS1 / (row_bytes / 16);
    if (pick_shape((int)((plan->max_n2 + rpp - 1) / rpp), (int)((plan->tile_rows + rpp - 1) / rpp)) < 0) return 0;
    return hop1_lds_bytes(plan, (int)row_bytes) <= 160 * 1024 ? 1 : 0;
}

int dsw_spmm1s_launch(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z, const void* Z2, void* Y,
                      int64_t B, int64_t C, float a, float b, float c, int dtype, hipStream_t stream, int stream_out) {
    if (!dsw_spmm1s_supported(plan, C, dtype)) return DSW_ERR_BAD_ARG;
    if (V <= 0 || B <= 0) return DSW_OK;
    if (!U || !Y) return DSW_ERR_BAD_ARG;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    Hop1Args A;
    A.tile_meta = plan->tile_meta; A.s2_rows = plan->s2_rows; A.lrowptr = plan->lrowptr;
    A.lcol = plan->lcol; A.lval = plan->lval; A.ell_pos = plan->ell_pos; A.ell_val = plan->ell_val;
    A.U = static_cast<const char*>(U); A.Z = static_cast<const char*>(Z); A.Z2 = static_cast<const char*>(Z2);
    A.Y = static_cast<char*>(Y);
    A.a = a; A.b = Z ? b : 0.f; A.c = Z2 ? c : 0.f;
    if (!A.Z && A.Z2) { A.Z = A.Z2; A.b = A.c; A.Z2 = nullptr; A.c = 0.f; }
    A.V = (int)V; A.n_tiles = plan->n_tiles; A.tile_rows = plan->tile_rows; A.explicit_tiles = plan->explicit_tiles;
    A.max_rt = plan->max_n1; A.max_n2 = plan->max_n2;
    A.row_stride = (int)(C * es);
    A.row_bytes = (A.row_stride > 128 && A.row_stride % 128 == 0) ? 128 : A.row_stride;
    A.ncc = A.row_stride / A.row_bytes;
    B *= A.ncc;
    A.lpr = A.row_bytes / 16; A.B = (int)B;
    A.ell_w = (plan->reserved + 3) & ~3;
    A.stream_out = stream_out;
    const size_t lds = hop1_lds_bytes(plan, A.row_bytes);
    long per_cu = (160 * 1024) / (long)lds > 0 ? (160 * 1024) / (long)lds : 1;
    if (per_cu > 2) per_cu = 2;                              // 512-thread workgroups at <= 128 registers
    if ((plan->max_n2 + (NTHREADS1 / A.lpr) - 1) / (NTHREADS1 / A.lpr) > 5) per_cu = 1;   // the 7 / 8-slot shapes (see OCC below)
    const long slots = 256L * per_cu;
    long chunks = 1;
    {
        double best = -1.0;
        const long cmax = B > 1 ? (B + 1) / 2 : 1;
        for (long cc = 1; cc <= cmax && cc <= 16; ++cc) {
            const long rounds = (plan->n_tiles * cc + slots - 1) / slots;
            const double cost = (double)rounds * (1.0 + (double)((B + cc - 1) / cc));
            if (best < 0 || cost < best - 1e-9) { best = cost; chunks = cc; }
        }
    }
    { static const char* ce = dsw_diag_env("DSW_H1_CHUNKS"); if (ce) chunks = atol(ce); }   // diagnostics
    if (chunks < 1) chunks = 1;
    A.spc = (int)((B + chunks - 1) / chunks);
    A.n_chunks = (int)((B + A.spc - 1) / A.spc);
    const long nwg = (long)plan->n_tiles * A.n_chunks;
    if (nwg > 2147483647L) return DSW_ERR_BAD_ARG;
    const int rpp = NTHREADS1 / A.lpr;
    const int shape = pick_shape((plan->max_n2 + rpp - 1) / rpp, (plan->tile_rows + rpp - 1) / rpp);
    const bool bf = dtype == DSW_BF16;
    // one output row per lane group, <= 32 entries per row, LDS-linear lanes, at most one epilogue operand: the LDS-DMA kernel
    static const char* dma_env = dsw_diag_env("DSW_H1_DMA");   // diagnostics: "0" = register-staged kernel
    const int nst_ = (plan->max_n2 + rpp - 1) / rpp;
    const size_t ldsd = hop1_dma_lds_bytes(plan, A.row_bytes, A.Z != nullptr);
    if (!(dma_env && dma_env[0] == '0') && !A.Z2 && (plan->ell_w == 24 || plan->ell_w == 32) && plan->ell_pos && plan->ell_val &&
        (A.lpr & (A.lpr - 1)) == 0 && plan->tile_rows <= rpp && plan->tile_rows <= 64 && nst_ <= 3 && ldsd <= 80 * 1024) {
#define DSW_H1_DMA2(BF_, NST_, W_)                                                                                   \
    do {                                                                                                             \
        if (ldsd > 64 * 1024) {                                                                                      \
            if (hipFuncSetAttribute((const void*)spmm1_dma_kernel<BF_, NST_, true, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd) != hipSuccess || \
                hipFuncSetAttribute((const void*)spmm1_dma_kernel<BF_, NST_, false, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd) != hipSuccess)   \
                return DSW_ERR_LAUNCH;                                                                               \
        }                                                                                                            \
        if (A.Z) { DSW_LAUNCH((spmm1_dma_kernel<BF_, NST_, true, W_>), dim3((unsigned)nwg), dim3(NTHREADS1), ldsd, stream, A); }    \
        else { DSW_LAUNCH((spmm1_dma_kernel<BF_, NST_, false, W_>), dim3((unsigned)nwg), dim3(NTHREADS1), ldsd, stream, A); }       \
        return dsw_check_launch();                                                                                   \
    } while (0)
#define DSW_H1_DMA(BF_, NST_) do { if (plan->ell_w == 24) DSW_H1_DMA2(BF_, NST_, 24); else DSW_H1_DMA2(BF_, NST_, 32); } while (0)
        if (nst_ <= 2) { if (bf) DSW_H1_DMA(true, 2); else DSW_H1_DMA(false, 2); }
        else { if (bf) DSW_H1_DMA(true, 3); else DSW_H1_DMA(false, 3); }
#undef DSW_H1_DMA
#undef DSW_H1_DMA2
    }
    // OCC = waves per SIMD the registers must allow: two 8-wave workgroups per CU for the 64 / 128-row tiles; the
    // 256-row shapes hold 7-8 staged rows per lane and run one workgroup per CU (LDS) anyway
#define DSW_H1_SHAPE(I_, NST_, NS2_, OCC_)                                                             \
    case I_:                                                                                           \
        return bf ? launch_h1<true, NST_, NS2_, OCC_>(A, nwg, lds, stream) : launch_h1<false, NST_, NS2_, OCC_>(A, nwg, lds, stream);
    switch (shape) {
        DSW_H1_SHAPE(0, 2, 1, 4) DSW_H1_SHAPE(1, 3, 1, 4) DSW_H1_SHAPE(2, 5, 2, 4) DSW_H1_SHAPE(3, 7, 4, 2) DSW_H1_SHAPE(4, 8, 4, 2)
    }
#undef DSW_H1_SHAPE
    return DSW_ERR_BAD_ARG;
}
