// fp32 channel-mix GEMM on the bf16 matrix pipe (exact 3-way split, see dsw_gemm_x3.hip) for WIDE layers:
// the split W panel of a column tile (3 * Kd * 128 * 2 B, e.g. 1.2 MB at Kd = 1536) does not fit LDS, so W is
// streamed chunk by chunk next to the activations instead of staying resident.
//
//   C (M x n_total) = sum_p A[p] (M x kd) * B[p] (kd x n_total) (+ bias)        [TsGemmParams, fp32 storage]
//
// Workgroup = NWV waves, tile = (32 * NWV rows) x (32 * NT columns); a wave owns 32 rows and NT MFMA tiles.
// Per 32-wide reduction chunk:
//   A: 3-deep register prefetch ring -> the wave's private staging rows in LDS (no workgroup barrier), split
//      into bf16 terms in registers right before the MFMAs (as in the resident kernel); round 6, 8-wave form with the
//      pre-split W image (DIRECT): the ring IS the operand - a lane loads its 2 x 8 values of the MFMA layout, no LDS;
//   B: the chunk's [32 k][32 * NT cols] fp32 block of W is fetched one chunk ahead into registers, split ONCE per
//      workgroup into three bf16 planes and written to LDS as [plane][col][k] (k contiguous, 80-byte rows ->
//      conflict-free ds_read_b128), double buffered: one workgroup barrier per chunk.
// W is tiny (<= a few MB) and re-read by every workgroup: it lives in L2; HBM sees the A stream and the output.
// The shapes that take this path are MFMA-bound (arithmetic intensity Kd * n / (Kd + n) >> machine balance), so
// the tile is as large as registers allow: 8 waves x (32 x 128) -> 48 MFMAs per wave and chunk against 16 staged
// W elements per lane.
#include "dsw_gemm_common.h"
#include <cstdio>

using namespace dsw_gemm;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int SC1 = 16;   // cache-policy operand of the raw buffer builtins on gfx940+: bit 4 = sc1, device scope
constexpr int KSB = 40;   // bf16 elements per B row in LDS: 32 k + 8 pad (80 B)

static __device__ __forceinline__ bf16x8_t pack_trunc8(const float (&f)[8]) {
    uint4 u;
    u.x = __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
    u.y = __builtin_amdgcn_perm(__float_as_uint(f[3]), __float_as_uint(f[2]), 0x07060302u);
    u.z = __builtin_amdgcn_perm(__float_as_uint(f[5]), __float_as_uint(f[4]), 0x07060302u);
    u.w = __builtin_amdgcn_perm(__float_as_uint(f[7]), __float_as_uint(f[6]), 0x07060302u);
    return __builtin_bit_cast(bf16x8_t, u);
}
static __device__ __forceinline__ float trunc_bf16(float f) {
    return __uint_as_float(__float_as_uint(f) & 0xffff0000u);
}
// one LDS-DMA instruction: 16 bytes per lane from gsrc (per lane) to lds_dst (wave-uniform) + lane * 16.  M0 carries the LDS
// destination; it is compiler-reserved: saved and restored in ONE statement (as in dsw_spmm1s.hip)
static __device__ __forceinline__ void glds16(const char* gsrc, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Pre-split image of the small operand (TsGemmParams::pre_ws): for chunk c = a-plane * chunks + chunk, column tile t and
// slot e = col * 16 + kp of the tile, three dwords {hi, mid, lo}, each the bf16 pair (k = 2 kp, 2 kp + 1) of one term -
// exactly what a thread of the GEMM writes to LDS for that slot.  1.9x the fp32 bytes of W (with the row pad of the LDS
// image), built once per call (a few microseconds), instead of ~5.5 vector instructions per element in every workgroup and chunk.
template <int BNT>
__global__ void x3s_presplit_kernel(const TsGemmParams P, unsigned* __restrict__ img, const int chunks, const int col_tiles,
                                    unsigned* __restrict__ flags, const int n_flags) {
    if (blockIdx.x == 0)   // ready flags of the balanced decomposition: cleared here, in the launch that precedes the GEMM
        for (int i = threadIdx.x; i < n_flags; i += blockDim.x) flags[i] = 0u;
    const long n_slots = (long)P.n_planes_a * chunks * col_tiles * (16 * BNT);
    const int n_total = P.n_planes_c * P.n_per_plane;
    const float* B = static_cast<const float*>(P.Bsrc);
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (long)gridDim.x * blockDim.x) {
        const int e = (int)(s % (16 * BNT));
        long r = s / (16 * BNT);
        const int t = (int)(r % col_tiles);
        r /= col_tiles;
        const int kc = (int)(r % chunks), p = (int)(r / chunks);
        const int col = e >> 4, kp = e & 15;
        const int j = t * BNT + col;
        float a = 0.f, b = 0.f;
        if (j < n_total) {
            const int q = j / P.n_per_plane, n = j - q * P.n_per_plane;
            const long g = (long)p * P.b_sp + (long)q * P.b_sq + (long)n * P.b_sn + (long)(kc * BK + 2 * kp) * P.b_skd;
            a = B[g]; b = B[g + P.b_skd];
            if (q == P.fold_q) { a -= B[g + 2 * P.b_sq]; b -= B[g + 2 * P.b_sq + P.b_skd]; }
        }
        const float ah = trunc_bf16(a), bh = trunc_bf16(b);
        const float a1 = a - ah, b1 = b - bh;
        const float am = trunc_bf16(a1), bm = trunc_bf16(b1);
        const float al = a1 - am, bl = b1 - bm;
        // the image of (chunk, column tile) IS the LDS image [3 planes][BNT cols][KSB] (pad included): the GEMM copies it with
        // LDS-DMA, 1 KiB per wave instruction, no registers and no ds_write in between
        unsigned* dst = img + (s / (16 * BNT)) * (long)(3 * BNT * KSB / 2) + col * (KSB / 2) + kp;
        dst[0] = __builtin_amdgcn_perm(__float_as_uint(bh), __float_as_uint(ah), 0x07060302u);
        dst[BNT * KSB / 2] = __builtin_amdgcn_perm(__float_as_uint(bm), __float_as_uint(am), 0x07060302u);
        dst[BNT * KSB] = __builtin_amdgcn_perm(__float_as_uint(bl), __float_as_uint(al), 0x07060302u);
    }
}

// KFAST: the B operand is contiguous along the reduction index (dgrad: W^T), else along the columns (forward).
// PRE: the W chunk comes pre-split from TsGemmParams::pre_ws (x3s_presplit_kernel) - three ready-made dwords per slot.
template <int NT, int NWV, bool KFAST, bool RES = false, bool PRE = false>
#ifdef DSW_X3S_LB4     // A/B builds: 128 registers per lane, two 8-wave workgroups per CU (with -DDSW_X3S_FORCE_NT2: 64-column tiles)
#define DSW_X3S_WAVES_PER_EU(pre_, nwv_) 4
#else
#define DSW_X3S_WAVES_PER_EU(pre_, nwv_) (((pre_) || (nwv_) == 8) ? 2 : 1)
#endif
__global__ __launch_bounds__(64 * NWV, DSW_X3S_WAVES_PER_EU(PRE, NWV)) void ts_gemm_x3s_kernel(const TsGemmParams P) {
    constexpr int BMT = 32 * NWV;
    constexpr int NTH = 64 * NWV;
    constexpr int BNT = 32 * NT;
    constexpr int PF = 3;
    constexpr int NPAIR = (16 * BNT + NTH - 1) / NTH;  // (k, k+1) element pairs of one W chunk per thread
    constexpr bool BTAIL = (16 * BNT) % NTH != 0;      // last slot only partly populated (6- and 7-wave workgroups)
    constexpr int BPLANE = BNT * KSB;                  // one bf16 plane of one B buffer
    // DIRECT (pre-split W, 8 waves): the A rows never touch LDS.  A lane loads, straight from HBM / L2, exactly the 2 x 8 values of
    // its row that the MFMA operand layout gives it (k = 16 s2 + 8 half .. + 7) - the same four 16-byte loads per lane and chunk
    // as the staged form, but no ds_write pass, no fragment reads, no A buffers: 61 KB of LDS instead of 134, so TWO workgroups
    // share a CU and run under each other's waits.
#ifdef DSW_X3S_NO_DIRECT
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = PRE && NWV == 8;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                                               // [2][BMT][LDA] (not DIRECT)
    // A rows are wave-private (a wave stages and reads only its own 32 rows, LDS operations of a wave complete in order): the
    // 4-wave shape keeps ONE A buffer - chunk it+1 is stored behind the fragment reads of chunk it - and with 78 KB of LDS
    // two of its workgroups share a CU, each with its own barrier: one's staging runs under the other's MFMAs.
    constexpr int NABUF = DIRECT ? 0 : NWV == 4 ? 1 : 2;
    unsigned short* Bt = reinterpret_cast<unsigned short*>(smem + NABUF * BMT * LDA);   // [2][3][BNT][KSB]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int n_total = P.n_planes_c * P.n_per_plane;
    const int col0 = blockIdx.y * BNT;
    const int chunks = P.kd_per_plane / BK;
    const int total = P.n_planes_a * chunks;
    const long row_tiles = (P.M + BMT - 1) / BMT;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    bool col_ok[NT];
    char* col_ptr[NT];
    float col_bias[NT];
    const char* col_res[NT];
    const float escale = RES ? epi_scale<false>(P) : 1.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int j = col0 + 32 * nt + l31;
        col_ok[nt] = j < n_total;
        const int q = col_ok[nt] ? j / P.n_per_plane : 0;
        const int n = col_ok[nt] ? j - q * P.n_per_plane : 0;
        char* base = static_cast<char*>((q == 0) ? P.C0 : P.C1);
        const size_t cbase = (q == 0) ? 0 : (size_t)(q - 1) * P.c_plane_stride;
        col_ptr[nt] = base + (cbase + (size_t)n) * 4;
        col_bias[nt] = (P.bias != nullptr && !(P.bias_plane0 && q != 0)) ? static_cast<const float*>(P.bias)[n] : 0.f;
        col_res[nt] = RES ? epi_res_ptr<false>(P, col_ok[nt], q, n) : nullptr;
    }

    // ---- W chunk: this thread's NPAIR (column, k-pair) slots; the global offset of a slot inside chunk 0
    int bslot_lds[NPAIR];     // element offset of the pair inside a plane of a B buffer
    int bslot_g[NPAIR];       // BYTE offset of W(k = 2*kp, col) relative to the chunk origin; < 0: column out of range
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        int e = tid + NTH * i;
        const bool slot_ok = !BTAIL || e < 16 * BNT;
        if (!slot_ok) e = 16 * BNT - 1;
        const int col = (KFAST || PRE) ? e >> 4 : e % BNT;
        const int kp = (KFAST || PRE) ? e & 15 : e / BNT;
        bslot_lds[i] = slot_ok ? col * KSB + 2 * kp : -1;
        const int j = col0 + col;
        if (j < n_total) {
            const int q = j / P.n_per_plane, n = j - q * P.n_per_plane;
            bslot_g[i] = (int)(((long)q * P.b_sq + (long)n * P.b_sn + (long)(2 * kp) * P.b_skd) * 4);   // W is a few MB
        } else {
            bslot_g[i] = -1;
        }
    }
    const float* Bsrc = static_cast<const float*>(P.Bsrc);
    constexpr int RBW = 2;                               // dwords per slot in the W ring: two fp32 values
    float rb0[NPAIR][RBW], rb1[NPAIR][RBW], rb2[NPAIR][RBW];   // W ring (not PRE): same depth / slot numbering as the A ring
    // Work of this workgroup.  Whole tiles: row tiles blockIdx.x, + gridDim.x, ...  Balanced (P.sk_part): the column tile's
    // row_tiles * total chunk steps are cut into gridDim.x equal contiguous ranges, wherever the cuts fall; a tile cut by a
    // range boundary is finished by the workgroup that holds its FIRST chunk (it reaches it at the END of its own range, when
    // the others - who met their piece first, or had nothing else - have parked theirs), see the segment logic in stage().
    const bool sk = P.sk_part != nullptr;
    const long S_sk = row_tiles * total;
    long n_iter, row_first, row_step;
    int c_first = 0;
    if (sk) {
        const long s0 = (long)blockIdx.x * S_sk / gridDim.x, s1 = (long)(blockIdx.x + 1) * S_sk / gridDim.x;
        const long t0 = s0 / total;
        n_iter = s1 - s0; c_first = (int)(s0 - t0 * total); row_first = t0 * BMT; row_step = BMT;
    } else {
        n_iter = (row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x * total;
        row_first = (long)blockIdx.x * BMT; row_step = (long)gridDim.x * BMT;
    }
    TileChunkIter cur, pre;   // the chunk being multiplied / the chunk being prefetched (PF ahead, clamped)
    cur.init_at(row_first, c_first, chunks);
    pre.init_at(row_first, c_first, chunks);
    int bp = cur.p, bkc = cur.kc;   // (plane, chunk) of the NEXT W chunk to fetch: cycles through the reduction, no clamp
    // PRE: the pre-split image of (chunk, this column tile) is the LDS image itself - copied by LDS-DMA, 1 KiB pieces dealt
    // round-robin to the waves; no ring registers, no ds_write.  The compiler does not see these operations: its own vmcnt
    // waits (A ring) become stricter than needed, never laxer (the counter retires in order); the wait for the pieces is
    // hand-counted in stage().
    constexpr int IMG_BYTES = 3 * BPLANE * 2;
    constexpr int NPIECE = IMG_BYTES / 1024;
    static_assert(IMG_BYTES % 1024 == 0, "whole 1 KiB pieces");
#ifdef DSW_ABL_X3S_NOB
    int abl_nob = 0;
#endif
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned bt_lds = (unsigned)reinterpret_cast<uintptr_t>(Bt);
    auto dma_b = [&](const int buf) __attribute__((always_inline)) {
#ifdef DSW_ABL_X3S_NOB      // timing experiment (wrong results): the W image is copied for the first two chunks only - what does its stream cost?
        if (abl_nob >= 2) { if (++bkc == chunks) { bkc = 0; if (++bp == P.n_planes_a) bp = 0; } return; }
        ++abl_nob;
#endif
        const char* src = static_cast<const char*>(P.pre_ws) + (((long)bp * chunks + bkc) * gridDim.y + blockIdx.y) * (long)IMG_BYTES + lane * 16;
        const unsigned dst = bt_lds + (unsigned)buf * IMG_BYTES;
#pragma unroll
        for (int i = 0; i < (NPIECE + NWV - 1) / NWV; ++i) {
            const int j = wave_u + NWV * i;
            if (j < NPIECE) glds16(src + j * 1024, dst + (unsigned)j * 1024u);
        }
        if (++bkc == chunks) { bkc = 0; if (++bp == P.n_planes_a) bp = 0; }
    };
    auto fetch_b = [&](float (&rb)[NPAIR][RBW]) __attribute__((always_inline)) {
        if constexpr (PRE) {
            (void)rb;
            return;
        } else {
            // uniform chunk origin (and origin + one reduction step) + the slot's 32-bit byte offset
            const char* o0 = reinterpret_cast<const char*>(Bsrc + (long)bp * P.b_sp + (long)(bkc * BK) * P.b_skd);
            const char* o1 = o0 + P.b_skd * 4;
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                const unsigned g = bslot_g[i] < 0 ? 0u : (unsigned)bslot_g[i];
                rb[i][0] = *reinterpret_cast<const float*>(o0 + g);   // unconditional (clamped) loads: nothing may consume the value
                rb[i][1] = *reinterpret_cast<const float*>(o1 + g);   // here, or the compiler parks a vmcnt(0) right behind them
            }
            if (++bkc == chunks) { bkc = 0; if (++bp == P.n_planes_a) bp = 0; }
        }
    };
    auto store_b = [&](unsigned short* buf, const float (&rb)[NPAIR][RBW]) __attribute__((always_inline)) {
        if constexpr (PRE) return;
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
            const float a = bslot_g[i] < 0 ? 0.f : rb[i][0], b = bslot_g[i] < 0 ? 0.f : rb[i][1];
            const float ah = trunc_bf16(a), bh = trunc_bf16(b);
            const float a1 = a - ah, b1 = b - bh;
            const float am = trunc_bf16(a1), bm = trunc_bf16(b1);
            const float al = a1 - am, bl = b1 - bm;
            if (BTAIL && bslot_lds[i] < 0) continue;
            uint32_t* dst = reinterpret_cast<uint32_t*>(buf + bslot_lds[i]);
            dst[0] = __builtin_amdgcn_perm(__float_as_uint(bh), __float_as_uint(ah), 0x07060302u);
            dst[BPLANE / 2] = __builtin_amdgcn_perm(__float_as_uint(bm), __float_as_uint(am), 0x07060302u);
            dst[BPLANE] = __builtin_amdgcn_perm(__float_as_uint(bl), __float_as_uint(al), 0x07060302u);
        }
    };

    // ---- A ring
    const int ar = wave * 32 + (lane >> 3);
    const int ac4 = (lane & 7) * 4;
    f32x4 ra0[4], ra1[4], ra2[4];
    // Addresses of the A loads = a UNIFORM pointer (plane, chunk: scalar registers, advanced by the scalar unit) + this lane's
    // 32-bit byte offset inside the plane (row * lda + column, fixed for a row tile).  The first form recomputed
    // (row * lda + plane * stride + k) in 64-bit vector arithmetic for every load: an in-kernel cycle timeline
    // (tools/x3s_timeline.py) showed 750 of the 5 800 cycles of a chunk step in issuing its 12 loads.
    unsigned arow_off[4];
    auto fetch = [&](f32x4 (&dra)[4]) __attribute__((always_inline)) {   // A rows of chunk `pre`, then advance it
        if (pre.c == 0 || pre.it == 0) {                                // uniform: first chunk of a row tile / of the range
            if constexpr (DIRECT) {
                long r = pre.row0 + wave * 32 + l31;                    // this lane's row of the wave's 32; its k-halves: 8 half + 16 s2
                r = r < P.M ? r : P.M - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    arow_off[i] = (unsigned)(((size_t)r * P.lda + 8 * half + 16 * (i >> 1) + 4 * (i & 1)) * 4);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long r = pre.row0 + ar + 8 * i;
                    r = r < P.M ? r : P.M - 1;
                    arow_off[i] = (unsigned)(((size_t)r * P.lda + ac4) * 4);    // < 4 GiB (checked by the launcher)
                }
            }
        }
        const char* abase = reinterpret_cast<const char*>(static_cast<const float*>((pre.p == 0) ? P.A0 : P.A1) +
                                                          ((pre.p == 0) ? 0 : (size_t)(pre.p - 1) * P.a_plane_stride) + pre.kc * BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) dra[i] = *reinterpret_cast<const f32x4*>(abase + arow_off[i]);
        pre.next_clamped(n_iter, chunks, total, row_step);
    };
    if (n_iter <= 0) return;
    // Software pipeline.  LDS holds two (A, W) buffer pairs; chunk j lives in pair j & 1.  Stage `it` multiplies
    // chunk it out of pair it & 1 and, in the SAME barrier interval (so that the scheduler can hide the split /
    // LDS-store VALU work in the MFMA shadow), stages chunk it+1 from ring slot (it+1) % 3 into the other pair and
    // refills that slot with chunk it+4.  W loads are issued before the A loads of a stage: vmcnt completes in
    // order, and the W slot is needed one instruction earlier.
    auto store_a = [&](float* abuf, const f32x4 (&slot)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&abuf[(ar + 8 * i) * LDA + ac4]) = slot[i];
    };
    if constexpr (DIRECT) {
        dma_b(0);
        fetch(ra0); fetch(ra1); fetch(ra2);     // chunks 0, 1, 2: stage it multiplies ring slot it % 3 and refills it with chunk it + 3
    } else if constexpr (PRE) {
        dma_b(0);                      // chunk 0 -> pair 0; chunk it+1 follows at the top of stage it
        fetch(ra0); fetch(ra1); fetch(ra2);
        store_a(As, ra0);
        fetch(ra0);
    } else {
        fetch_b(rb0); fetch(ra0);
        fetch_b(rb1); fetch(ra1);
        fetch_b(rb2); fetch(ra2);
        store_b(Bt, rb0);
        store_a(As, ra0);
        fetch_b(rb0); fetch(ra0);
    }
    const long n_pad = (n_iter + PF - 1) / PF * PF;
    bool seg_first = c_first == 0;   // the current segment holds chunk 0 of its tile

    auto stage = [&](auto U, const long it) __attribute__((always_inline)) {
        constexpr int v = (decltype(U)::value + 1) % 3;     // ring slot of chunk it+1
        f32x4 (&slot)[4] = *[&]() -> f32x4 (*)[4] {
            if constexpr (v == 0) return &ra0;
            else if constexpr (v == 1) return &ra1;
            else return &ra2;
        }();
        float (&bslot)[NPAIR][RBW] = *[&]() -> float (*)[NPAIR][RBW] {
            if constexpr (v == 0) return &rb0;
            else if constexpr (v == 1) return &rb1;
            else return &rb2;
        }();
        const int par = (int)(it & 1);
        const float* abuf = As + (size_t)(NABUF == 2 ? par : 0) * BMT * LDA;
        const unsigned short* bbuf = Bt + (size_t)par * 3 * BPLANE;
        // PRE: this wave's pieces of chunk `it` have landed - younger than them, still in flight: the 4 A loads issued behind
        // them (more at the end of a tile, where the count only errs on the safe side)
        if constexpr (PRE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();   // pair `par` complete; every wave is done reading the other pair (stage it-1)
        if constexpr (PRE) dma_b(par ^ 1);     // chunk it+1 -> the other pair, under the MFMAs of this one
        const float* arow = &abuf[(wave * 32 + l31) * LDA + 8 * half];
        const unsigned short* brow = bbuf + (size_t)l31 * KSB + 8 * half;
        f32x4 (&cslot)[4] = *[&]() -> f32x4 (*)[4] {          // DIRECT: the ring slot of chunk `it` itself
            if constexpr (decltype(U)::value == 0) return &ra0;
            else if constexpr (decltype(U)::value == 1) return &ra1;
            else return &ra2;
        }();
#pragma unroll
        for (int s2 = 0; s2 < BK / 16; ++s2) {
            f32x4 x0, x1;
            if constexpr (DIRECT) {
                x0 = cslot[2 * s2]; x1 = cslot[2 * s2 + 1];
            } else {
                x0 = *reinterpret_cast<const f32x4*>(arow + 16 * s2);
                x1 = *reinterpret_cast<const f32x4*>(arow + 16 * s2 + 4);
            }
            const float f[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            float r1[8], r2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r1[j] = f[j] - trunc_bf16(f[j]);
                r2[j] = r1[j] - trunc_bf16(r1[j]);
            }
            const bf16x8_t ah = pack_trunc8(f), am = pack_trunc8(r1), al = pack_trunc8(r2);
#ifdef DSW_X3S_BFRAG_ALL
            // A/B build: every W fragment of this k-step requested before its first MFMA (48 registers of the 70 this kernel leaves
            // unused at two waves per SIMD), instead of one column tile ahead
            bf16x8_t fbh[NT], fbm[NT], fbl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned short* bp_ = brow + (size_t)(32 * nt) * KSB + 16 * s2;
                fbh[nt] = *reinterpret_cast<const bf16x8_t*>(bp_);
                fbm[nt] = *reinterpret_cast<const bf16x8_t*>(bp_ + BPLANE);
                fbl[nt] = *reinterpret_cast<const bf16x8_t*>(bp_ + 2 * BPLANE);
            }
#endif
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned short* bp_ = brow + (size_t)(32 * nt) * KSB + 16 * s2;
#ifdef DSW_X3S_BFRAG_ALL
                const bf16x8_t bh = fbh[nt], bm = fbm[nt], bl = fbl[nt];
                (void)bp_;
#else
                const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(bp_);
                const bf16x8_t bm = *reinterpret_cast<const bf16x8_t*>(bp_ + BPLANE);
                const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(bp_ + 2 * BPLANE);
#endif
                f32x16 a_ = acc[nt];
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a_, 0, 0, 0);
                acc[nt] = a_;
            }
        }
#ifdef DSW_X3S_BFRAG_ALL
        // order of the region: the 3 NT fragment reads of k-step 0 first; the reads of k-step 1 trickle in, one behind every
        // two MFMAs of k-step 0 (their registers free up as they go); then the MFMAs of k-step 1
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NT, 0);
#pragma unroll
        for (int i = 0; i < 3 * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT, 0);
#endif
        if constexpr (DIRECT) {
            fetch(cslot);       // chunk it + 3 into the slot just multiplied
        } else {
            // chunk it+1 -> the other pair, then refill its ring slot with chunk it+4
            store_b(Bt + (size_t)(par ^ 1) * 3 * BPLANE, bslot);
            store_a(As + (size_t)(NABUF == 2 ? (par ^ 1) : 0) * BMT * LDA, slot);
            fetch_b(bslot);
            fetch(slot);
        }
        const bool tile_end = cur.c == total - 1;
        bool finish = tile_end && it < n_iter;
        if (sk && it < n_iter && (tile_end || it == n_iter - 1) && !(seg_first && tile_end)) {
            // a segment that is not a whole tile ends here
            const long slot0 = (long)blockIdx.y * gridDim.x;
            if (!seg_first) {
                // A piece WITHOUT the tile's first chunk (always the first segment of a range): park it in this workgroup's
                // slot and raise its flag.  Device-scope (sc1, write-through) stores - the reader sits on another XCD with its
                // own L2.  Not a release FENCE: that writes back the whole L2, once per wave (measured: 65 -> 142 us on a
                // 6 144-row layer); not atomic stores / loads either: the compiler waits for each before it issues the next.
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    P.sk_part + (slot0 + blockIdx.x) * (long)(BMT * BNT), 0, BMT * BNT * 4, 0x00020000);
#ifdef DSW_ABLATION
                if (!(P.dbg & 4))      // the flag is still raised
#endif
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[nt][i]), rs, tid * 4, (nt * 16 + i) * NTH * 4, SC1);
                        acc[nt][i] = 0.f;
                    }
                // RELEASE: every wave drains ITS OWN stores before the barrier (the compiler emits only vmcnt(63) here - the
                // stores are fire-and-forget for it - and a gfx950 workgroup barrier does not wait for memory operations), so
                // the flag below cannot become visible before any byte of the piece
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(P.sk_flags + slot0 + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                finish = false;
            } else {
                // The head of a tile at the end of the range: add the pieces of the later ranges in workgroup order, then the
                // epilogue.  (Done after the loop, with all loads of a piece in flight, the kernel ran out of scalar
                // registers and got 25 % slower.)
                const long tile_stop = (cur.row0 / BMT + 1) * total;
                for (long b = blockIdx.x + 1; b < gridDim.x && b * S_sk / gridDim.x < tile_stop; ++b) {
#ifdef DSW_ABLATION
                    if (P.dbg & 8) break;
#endif
                    if (tid == 0)
                        while (__hip_atomic_load(P.sk_flags + slot0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                            __builtin_amdgcn_s_sleep(4);
                    __syncthreads();
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        P.sk_part + (slot0 + b) * (long)(BMT * BNT), 0, BMT * BNT * 4, 0x00020000);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {   // 16 loads in flight, not 64: the registers are full
                        unsigned t[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i)    // device-scope loads: never a stale line of this XCD's L2
                            t[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, (nt * 16 + i) * NTH * 4, SC1);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[nt][i] += __uint_as_float(t[i]);
                        asm volatile("" ::: "memory");
                    }
                }
                finish = true;
            }
        }
        if (it < n_iter && (tile_end || it == n_iter - 1)) seg_first = true;   // the next segment starts a tile
        if (finish) {
            const long rbase = cur.row0 + wave * 32 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float* C = reinterpret_cast<float*>(col_ptr[nt]);
                const float bias = col_bias[nt];
                float rv[16];
                if constexpr (RES) {
                    asm volatile("" ::: "memory");   // see dsw_gemm_x3.hip
                    epi_res_load<false>(rv, col_res[nt], rbase, P.ldr, P.M);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) rv[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long r = rbase + (i & 3) + 8 * (i >> 2);
                    if (col_ok[nt] && r < P.M) C[(size_t)r * P.ldc] = epi_fin(acc[nt][i], bias, escale, rv[i], P.relu);
                    acc[nt][i] = 0.f;
                }
            }
        }
        cur.next(chunks, total, row_step);
    };
    for (long base = 0; base < n_pad; base += PF) {
        stage(std::integral_constant<int, 0>{}, base);
        stage(std::integral_constant<int, 1>{}, base + 1);
        stage(std::integral_constant<int, 2>{}, base + 2);
    }
}

// bytes of balanced-decomposition scratch behind the W image: flags (one per workgroup, 4 KiB) + one partial tile per workgroup
constexpr long SK_FLAG_BYTES = 4096;
constexpr long SK_MAX_WG = 1024;

template <int NT, int NWV, bool KFAST, bool RES = false, bool PRE = false>
int launch_x3s(const TsGemmParams& P0, int col_tiles, hipStream_t stream, char* sk_ws, long sk_bytes, bool flags_clear) {
    TsGemmParams P = P0;
#ifdef DSW_ABLATION   // -DDSW_ABLATION -DDSW_DIAG + DSW_DBG: 4 = pieces are not parked, 8 = not collected (wrong results by design)
    { static const char* d = dsw_diag_env("DSW_DBG"); P.dbg = d ? atoi(d) : 0; }
#endif
    constexpr int BMT = 32 * NWV;
#ifdef DSW_X3S_NO_DIRECT
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = PRE && NWV == 8;
#endif
    const size_t lds = (size_t)(DIRECT ? 0 : NWV == 4 ? 1 : 2) * BMT * LDA * 4 + (size_t)2 * 3 * (32 * NT) * KSB * 2;
    const long row_tiles = (P.M + BMT - 1) / BMT;
    const void* kfn = (const void*)ts_gemm_x3s_kernel<NT, NWV, KFAST, RES, PRE>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DSW_ERR_LAUNCH;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * NWV, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const long n_cu = dsw_device_cus();      // the device the launch goes to (a partitioned / masked device has fewer than 256)
    long gx = n_cu * per_cu / col_tiles;
    if (gx < 1) gx = 1;
    if (col_tiles > 1 && gx >= 8) gx &= ~7L;   // column tiles of one row tile on one XCD (A re-reads hit its L2)
    // Whole tiles per workgroup leave CUs idle whenever the row tiles do not divide by the workgroups (384 tiles on 256 CUs:
    // two rounds for 1.5 rounds of work; 96 x 2 tiles: a quarter of the chip idle).  With scratch for one partial tile per
    // workgroup the chunk steps are divided evenly instead (kernel comment); at least SK_MIN_STEPS steps per workgroup keep
    // the pipeline prologue and the parked pieces small against the work.  Measured (tools/bench_gemm.py --ws, same box, whole
    // tiles -> balanced): 98 304 x 768 -> 128 199 -> 180 us, 24 576 x 1 536 -> 256 163 -> 150, 6 144 x 1 536 -> 256 62 -> 48;
    // parking + collecting the pieces costs 4-7 us of that (ablation), and the chip gives part of the rest back as clock (all
    // CUs busy: DVFS) - which is why 24 576 x 576 -> 256 (18 steps per tile, 4.5 saved) LOSES 4 us: short tiles stay whole.
    static const char* skenv = dsw_diag_env("DSW_X3S_BALANCED");   // "0": whole tiles only (diagnostics / A-B)
    constexpr long SK_MIN_STEPS = 6;
    const long steps = row_tiles * (long)(P.n_planes_a * (P.kd_per_plane / BK));
    long gsk = gx;
    if (gsk > steps / SK_MIN_STEPS) gsk = steps / SK_MIN_STEPS;
    if (col_tiles > 1 && gsk >= 8) gsk &= ~7L;
    const long tile_bytes = (long)BMT * (32 * NT) * 4;
    const long steps_tile = steps / row_tiles;
    const long whole = (row_tiles + (gx < row_tiles ? gx : row_tiles) - 1) / (gx < row_tiles ? gx : row_tiles) * steps_tile;
    const bool pays = steps_tile >= 24 && gsk >= 2 && (whole - steps / gsk) * 5 >= whole;   // saves >= 20 % of the steps
    // Forward progress of the balanced form: a workgroup parks the piece at the START of its range without waiting for
    // anybody and waits, at the END of its range, only for the first pieces of the FOLLOWING workgroups that share its last
    // tile (at most ceil(steps_tile / (steps / gsk)) of them).  Workgroups are dispatched in index order, so with other work
    // resident on the chip (RCCL kernels, side streams) the resident set is a prefix of the grid: every member whose
    // followers are resident finishes and frees its slot for the next index.  That argument needs room for one tile's
    // span of workgroups; a device with very few CUs (partition modes) takes whole tiles.
    const long span = (steps_tile + (steps / (gsk > 0 ? gsk : 1)) - 1) / ((steps / (gsk > 0 ? gsk : 1)) > 0 ? (steps / (gsk > 0 ? gsk : 1)) : 1) + 1;
    const bool progress_ok = n_cu >= 32 && span * 4 <= n_cu * per_cu;
    const bool balanced = !(skenv && skenv[0] == '0') && sk_ws != nullptr && pays && progress_ok && row_tiles % gsk != 0 &&
                          gsk * col_tiles <= SK_MAX_WG && sk_bytes >= SK_FLAG_BYTES + gsk * col_tiles * tile_bytes;
#ifdef DSW_DIAG
    { static const char* tr = dsw_diag_env("DSW_X3S_TRACE");
      if (tr) fprintf(stderr, "x3s M=%ld planes=%d kd=%d n=%d nt=%d nwv=%d col_tiles=%d row_tiles=%ld gx=%ld gsk=%ld steps_tile=%ld whole=%ld bal=%d kfast=%d pre=%d res=%d\n",
                      (long)P.M, P.n_planes_a, P.kd_per_plane, P.n_planes_c * P.n_per_plane, NT, NWV, col_tiles, row_tiles, gx, gsk, steps_tile, whole, (int)balanced, (int)KFAST, (int)PRE, (int)RES); }
#endif
    if (balanced) {
        gx = gsk;
        P.sk_flags = reinterpret_cast<unsigned*>(sk_ws);
        P.sk_part = reinterpret_cast<float*>(sk_ws + SK_FLAG_BYTES);
        if (!flags_clear && hipMemsetAsync(sk_ws, 0, (size_t)(gsk * col_tiles * 4), stream) != hipSuccess) return DSW_ERR_LAUNCH;
    } else if (gx > row_tiles) {
        gx = row_tiles;
        if (col_tiles > 1 && gx >= 8) gx &= ~7L;
    }
    dim3 grid((unsigned)gx, (unsigned)col_tiles);
    DSW_LAUNCH((ts_gemm_x3s_kernel<NT, NWV, KFAST, RES, PRE>), grid, dim3(64 * NWV), lds, stream, P);
    return dsw_check_launch();
}

}  // namespace

// Streaming-W x3 GEMM: fp32 storage, aligned operands (kd_per_plane % 32 == 0, 16-byte A rows).  Returns 1 when it
// took the launch (*rc = status).
int dsw_ts_gemm_x3s_try_launch(const TsGemmParams& P, hipStream_t stream, int* rc) {
    static const char* x3env = dsw_diag_env("DSW_GEMM_X3");   // "0": exact fp32 MFMA kernels (diagnostics / A-B)
    if (x3env && x3env[0] == '0') return 0;
    static const char* senv = dsw_diag_env("DSW_GEMM_X3S");   // "0": disable only the streaming variant
    if (senv && senv[0] == '0') return 0;
    if (!P.a_vec || P.kd_per_plane % BK != 0 || P.M <= 0) return 0;
    if ((unsigned long long)P.M * (unsigned long long)P.lda * 4ull >= (1ull << 32)) return 0;   // 32-bit row offsets inside a plane
    const int n_total = P.n_planes_c * P.n_per_plane;
    if (((long)P.n_planes_c * (P.b_sq < 0 ? -P.b_sq : P.b_sq) + (long)P.n_per_plane * (P.b_sn < 0 ? -P.b_sn : P.b_sn) +
         32L * (P.b_skd < 0 ? -P.b_skd : P.b_skd)) * 4 >= (1L << 31)) return 0;                 // 32-bit slot offsets inside a chunk
    const bool kfast = P.b_skd == 1;
    const bool res = P.R != nullptr || P.scale != nullptr;     // epilogue operands: separate instantiations
    // Tile shape.  256 x 128 (8 waves) is the fastest shape per CU (a chunk step costs 3.6 us there against 2.7-3.0 us for
    // the half-size shapes), but a launch that gives fewer than half of the CUs a workgroup is better served by more, smaller
    // tiles: 64-column tiles double the workgroups (A is re-read out of L2), 128-row tiles double them again.  Measured
    // (tools/bench_gemm.py, 6 144 rows): 512 -> 256 102 -> 66 us with 128 x 64 tiles, 256 -> 512 58 -> 47 us with 256 x 64.
    static const char* ntenv = dsw_diag_env("DSW_X3S_NT");     // diagnostics: force 64-column tiles ("2")
    static const char* nwvenv = dsw_diag_env("DSW_X3S_NWV");
    const long row_tiles8 = (P.M + 255) / 256;
    int nt = n_total > 64 ? 4 : 2;
    if (nt == 4 && row_tiles8 * ((n_total + 127) / 128) < 128) nt = 2;
    if (ntenv) nt = ntenv[0] == '2' ? 2 : (n_total > 64 ? 4 : 2);
#ifdef DSW_X3S_FORCE_NT2
    nt = 2;
#endif
    const int col_tiles = (n_total + 32 * nt - 1) / (32 * nt);
    const long tiles8 = row_tiles8 * col_tiles;
    // one column tile + the pre-split image: two 4-wave workgroups per CU (one A buffer: 78 KB of LDS each) beat one 8-wave
    // workgroup on short reductions (98 304 x 192 -> 128: 51.9 -> 47.2 us same box); with more column tiles, or on long
    // reductions (98 304 x 768 -> 128: 156 -> 162 us), they do not
    const bool two_wg = col_tiles == 1 && nt == 4 && P.pre_ws != nullptr && P.n_planes_a * (P.kd_per_plane / BK) < 24;
    const int nwv = nwvenv ? atoi(nwvenv) : ((tiles8 < 128 || two_wg) ? 4 : 8);
    // W split once per call into the caller's scratch (with the fold of output plane fold_q, if any) when there is room
    static const char* preenv = dsw_diag_env("DSW_X3S_PRE");   // "0": split W per workgroup and chunk (diagnostics / A-B)
    const int chunks_ = P.kd_per_plane / BK;
    const long img_bytes = (long)P.n_planes_a * chunks_ * col_tiles * (3L * 32 * nt * KSB * 2);   // LDS-shaped: 3 planes x cols x (32 k + pad)
    const bool pre = !(preenv && preenv[0] == '0') && P.pre_ws != nullptr && P.pre_bytes >= img_bytes &&
                     (((uintptr_t)P.pre_ws) & 15u) == 0;
    if (!pre && P.fold_q >= 0) return 0;       // the caller folds the weights itself and comes back without fold_q
    // scratch of the balanced decomposition: behind the image (whether or not this call builds one)
    const long img_room = (img_bytes + 255) / 256 * 256;
    char* sk_ws = nullptr;
    long sk_bytes = 0;
    if (P.pre_ws != nullptr && (((uintptr_t)P.pre_ws) & 15u) == 0 && P.pre_bytes > img_room + SK_FLAG_BYTES) {
        sk_ws = static_cast<char*>(P.pre_ws) + img_room;
        sk_bytes = P.pre_bytes - img_room;
    }
    unsigned* flags = reinterpret_cast<unsigned*>(sk_ws);
    const int n_flags = sk_ws ? (int)SK_MAX_WG : 0;
    const bool fc = pre && sk_ws != nullptr;   // the split launch clears the flags on its way
    if (pre) {
        const long n_slots = (long)P.n_planes_a * chunks_ * col_tiles * (16L * 32 * nt);
        const int blocks = (int)((n_slots + 255) / 256 < 2048 ? (n_slots + 255) / 256 : 2048);
        if (nt == 4) DSW_LAUNCH((x3s_presplit_kernel<128>), dim3(blocks), dim3(256), 0, stream, P, static_cast<unsigned*>(P.pre_ws), chunks_, col_tiles, flags, n_flags);
        else DSW_LAUNCH((x3s_presplit_kernel<64>), dim3(blocks), dim3(256), 0, stream, P, static_cast<unsigned*>(P.pre_ws), chunks_, col_tiles, flags, n_flags);
        if ((*rc = dsw_check_launch()) != DSW_OK) return 1;
    }
#define DSW_X3S(NT_, NWV_)                                                                                          \
    (*rc = pre ? (res ? launch_x3s<NT_, NWV_, false, true, true>(P, col_tiles, stream, sk_ws, sk_bytes, fc)                              \
                      : launch_x3s<NT_, NWV_, false, false, true>(P, col_tiles, stream, sk_ws, sk_bytes, fc))                            \
         : res ? (kfast ? launch_x3s<NT_, NWV_, true, true>(P, col_tiles, stream, sk_ws, sk_bytes, fc)                                   \
                        : launch_x3s<NT_, NWV_, false, true>(P, col_tiles, stream, sk_ws, sk_bytes, fc))                                 \
               : (kfast ? launch_x3s<NT_, NWV_, true>(P, col_tiles, stream, sk_ws, sk_bytes, fc) : launch_x3s<NT_, NWV_, false>(P, col_tiles, stream, sk_ws, sk_bytes, fc)))
#define DSW_X3S_NWV(NT_)                                                                     \
    if (nwv == 4) DSW_X3S(NT_, 4); else DSW_X3S(NT_, 8);
    if (nt == 4) { DSW_X3S_NWV(4) } else { DSW_X3S_NWV(2) }
#undef DSW_X3S_NWV
#undef DSW_X3S
    return 1;
}
