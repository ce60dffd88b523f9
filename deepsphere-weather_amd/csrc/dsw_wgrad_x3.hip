// wgrad on the bf16 matrix pipe:  partial[s][(k, f)][o] = sum_{n in slab s} T_k[n, f] * dY[n, o]
//
// Both MFMA operands of this contraction are "transposed": the reduction index n is the ROW index of the
// row-major activations, while v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction elements per lane.
// gfx950 has the instruction for exactly this: ds_read_b64_tr_b16 delivers, to lane l of a 16-lane group, column
// (l & 15) of a [4 n][16 ch] bf16 block whose rows the group's lanes address 8 bytes each - a free 4x4 transpose
// on the LDS read path.  So the staging pass keeps the NATURAL row-major layout: every fp32 element is split
// exactly into 3 bf16 terms (or, for bf16 storage, taken as is), four channels are packed into one 8-byte LDS
// write per plane, and the tile image is [plane][32 n][32 ch] with 64-byte rows (no padding: a 32-lane read
// group then covers 4 rows x 64 B = all 64 banks exactly once).  An operand fragment is two transpose reads.
// The six leading cross terms of the split accumulate in fp32 - same accuracy argument as dsw_gemm_x3.hip.
//
// Workgroup = NW waves; wave w owns the (k, f)-tile blockIdx.y * NW + w (32 channel rows of one T plane) times
// NO 32-column blocks of dY (the workgroup's dY tile is shared by its waves).  NW up to 8 and NO = 4 for
// Fout % 128 == 0 keep the re-reads of dY (once per workgroup row of the grid) and of T (once per column of the
// grid) low for wide layers.  Row slabs -> fp32 partials -> cheb_wgrad_reduce_kernel (deterministic), 3-deep
// register prefetch ring over 32-row chunks.
#include "dsw_gemm_common.h"

using namespace dsw_gemm;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BLK = 32 * 32;   // bf16 elements of one [32 n][32 ch] block (2 KiB, 64-byte rows)

static __device__ __forceinline__ float trunc_bf16(float f) {
    return __uint_as_float(__float_as_uint(f) & 0xffff0000u);
}

template <bool BF16IO, bool NT = false>
static __device__ __forceinline__ f32x4 load4(const void* p, size_t i) {
    if constexpr (BF16IO) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(p) + i);
        f32x4 v;
        v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
        v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
        return v;
    } else {
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(static_cast<const float*>(p) + i));
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(p) + i);
    }
}

// truncating pack of 4 fp32 -> 4 bf16 (8 bytes)
static __device__ __forceinline__ u32x2 pack4(const float a, const float b, const float c, const float d) {
    u32x2 r;
    r[0] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    r[1] = __builtin_amdgcn_perm(__float_as_uint(d), __float_as_uint(c), 0x07060302u);
    return r;
}

// split the 4 channels of v into NSPLIT bf16 terms and write them row-major: dst -> (row n, channel c) of plane 0
template <int NSPLIT>
static __device__ __forceinline__ void split_store(unsigned short* dst, const int plane_elems, const f32x4 v) {
    const float h0 = trunc_bf16(v[0]), h1 = trunc_bf16(v[1]), h2 = trunc_bf16(v[2]), h3 = trunc_bf16(v[3]);
    *reinterpret_cast<u32x2*>(dst) = pack4(h0, h1, h2, h3);
    if constexpr (NSPLIT == 3) {
        const float r0 = v[0] - h0, r1 = v[1] - h1, r2 = v[2] - h2, r3 = v[3] - h3;
        const float m0 = trunc_bf16(r0), m1 = trunc_bf16(r1), m2 = trunc_bf16(r2), m3 = trunc_bf16(r3);
        *reinterpret_cast<u32x2*>(dst + plane_elems) = pack4(m0, m1, m2, m3);
        *reinterpret_cast<u32x2*>(dst + 2 * plane_elems) = pack4(r0 - m0, r1 - m1, r2 - m2, r3 - m3);
        // (the residuals as register pairs - v_pk_add_f32, half the subtractions - measured +-0 on the NS step and the U-Net
        // in round 4: the split is not what this loop waits for)
    }
}

// MFMA operand fragment (8 consecutive n of this lane's column) out of a [32 n][32 ch] block: two transpose reads
// (rows +0..3 and +4..7 of the lane half's 8-row group).  `p` already carries the lane's offset (frag_off below).
static __device__ __forceinline__ bf16x8_t read_frag_tr(const unsigned short* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 32));
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, r);
}

// the same fragment from two explicit addresses (rows +0..3 and rows +4..7): swizzled images
static __device__ __forceinline__ bf16x8_t read_frag_tr2(const unsigned short* p0, const unsigned short* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, r);
}

// FUSE (fp32, one o-tile covering all of Fout, one (k,f)-group): the workgroup holds the dY rows it needs for dW
// anyway, so it also evaluates the dgrad of those rows, G_k[n, f] = sum_o dY[n, o] W[f, k, o]: wave w multiplies the
// staged dY image (row-major blocks: plain 16-byte A fragments) with its 32 columns of the pre-split W^T panel (LDS,
// [plane][col][o], 16-byte padded rows) and writes its 32 x 32 tile of G.  dY is then read from HBM once for the
// whole backward GEMM work instead of twice (NS: 200 MB less traffic, one launch less).
template <bool BF16IO, int NSPLIT, int NW, int NO, bool FUSE = false>
__global__ __launch_bounds__(64 * NW, FUSE ? 2 : 1) void cheb_wgrad_x3_kernel(const WgradParams P) {
    static_assert(!FUSE || (!BF16IO && NSPLIT == 3), "fused dgrad: fp32 storage");
    constexpr int NT_ = 64 * NW;
    constexpr int BNO = 32 * NO;                        // dY columns of the workgroup
    constexpr int CQ = BNO / 4;                         // column quads of the dY tile
    constexpr int DV = WR * CQ;                         // float4 of the dY tile
    constexpr int RD = (DV + NT_ - 1) / NT_;            // ... per thread
    constexpr int G = NT_ / CQ;                         // threads sharing a column quad (NT_ % CQ == 0)
    static_assert(NT_ % CQ == 0, "column-sum layout");
    constexpr int PF = FUSE ? 2 : 3;                    // the fused variant needs the registers of the third slot
    constexpr int TPLANE = BLK;                         // one (wave, plane) T tile
    // dY image: NO blocks [32 n][32 o] per plane, 64-byte rows.  Two things keep BOTH ways it is read off the bank
    // conflicts the plain layout had (41 % of the LDS-active cycles at the north-star shape, PMC round 2):
    //  * the 16-byte chunks of a row are XOR-swizzled with (n >> 2) & 3: the fused dgrad reads the image by ROWS (lane =
    //    row, 16 bytes at a fixed column), and rows n, n + 4, n + 8, n + 12 of a 64-byte-row image share their banks -
    //    a 4-way conflict on every read; with the swizzle the 16 rows of a read phase hit 16 distinct bank quads.  The
    //    transpose reads of the wgrad address 4-row groups (one key per group): still every bank exactly once.
    //  * blocks are DBLK = BLK + 64 elements apart: a 32-lane write group stores rows n, n + 1 of BOTH blocks, and with
    //    2 KiB between the blocks the two copies of a row fell on the same banks.
    constexpr int DBLK = BLK + 64;
    constexpr int DPLANE = NO * DBLK;                   // one dY plane
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
    unsigned short* TsT = xs;                                  // [NW][NSPLIT][32 n][32 f]
    unsigned short* DsT = xs + (size_t)NW * NSPLIT * TPLANE;   // [NSPLIT][NO][32 n][32 o]
    float* red = reinterpret_cast<float*>(DsT + (size_t)NSPLIT * DPLANE);   // [G][BNO] column-sum scratch
    constexpr int WKS = BNO + 8;                        // bf16 elements per W^T panel row (o contiguous, 16 B pad)
    unsigned short* Wp = reinterpret_cast<unsigned short*>(red + (size_t)G * BNO);   // FUSE: [3][NW * 32 cols][WKS]
    constexpr int WPLANE = NW * 32 * WKS;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.y * NW + wave;
    const int ntiles = P.K * P.tiles_per_plane;
    const bool active = tile < ntiles;
    const int k = active ? tile / P.tiles_per_plane : 0;
    const int f0 = active ? (tile - k * P.tiles_per_plane) * 32 : 0;
    const int zk = P.dy_planes > 1 ? blockIdx.z / P.otiles : 0;    // dY plane (mix-first backward), else 0
    const int o0 = (blockIdx.z - zk * P.otiles) * BNO;
    const int Kd = (P.dy_planes > 1 ? P.dy_planes : P.K) * P.Fin;     // rows of dW in one partial slab
    const int k_out = P.dy_planes > 1 ? zk : k;                     // dW plane this wave's tile belongs to
    const bool do_db = blockIdx.y == 0 && zk == 0;
    const long n_begin = (long)blockIdx.x * P.rows_per_slab;
    const long n_end = (n_begin + P.rows_per_slab < P.N) ? n_begin + P.rows_per_slab : P.N;
    const void* A = (k == 0) ? P.X : P.T;
    const size_t abase = (k == 0) ? 0 : (size_t)(k - 1) * P.plane_stride;
    const void* dYp = zk == 0 ? P.dY : P.dY1;
    const size_t dybase = zk == 0 ? 0 : (size_t)(zk - 1) * P.dy_plane_stride;

    if constexpr (FUSE) {
        // W^T panel of this workgroup's (k, f) columns, split once: Wp[plane][w * 32 + fl][o] = split(W[f0(w) + fl, k(w), o])
        const float* Wsrc = static_cast<const float*>(P.W);
        for (int e = tid; e < NW * 32 * BNO; e += NT_) {
            const int c = e / BNO, o = e - c * BNO;
            const int t_ = blockIdx.y * NW + (c >> 5);
            float v = 0.f;
            if (t_ < ntiles) {
                const int kk = t_ / P.tiles_per_plane, ff = (t_ - kk * P.tiles_per_plane) * 32 + (c & 31);
                v = Wsrc[((size_t)ff * P.K + kk) * P.Fout + o];
                if (P.fold && kk == P.K - 3) v -= Wsrc[((size_t)ff * P.K + (P.K - 1)) * P.Fout + o];
            }
            const float h = trunc_bf16(v), r1 = v - h, m = trunc_bf16(r1), l = r1 - m;
            Wp[c * WKS + o] = (unsigned short)(__float_as_uint(h) >> 16);
            Wp[WPLANE + c * WKS + o] = (unsigned short)(__float_as_uint(m) >> 16);
            Wp[2 * WPLANE + c * WKS + o] = (unsigned short)(__float_as_uint(l) >> 16);
        }
        // visible to every wave after the first __syncthreads() of the chunk loop
    }
    f32x16 acc[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float cs[RD][4];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[i][j] = 0.f;

    const int tr = lane >> 3, tc4 = (lane & 7) * 4;   // T tile (per wave): rows tr + 8*i (i<4), channels tc4..+3

    f32x4 rt0[4], rt1[4], rt2[4], rd0[RD], rd1[RD], rd2[RD];
    auto fetch = [&](long n0, f32x4 (&drt)[4], f32x4 (&drd)[RD]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            drt[i] = load4<BF16IO, FUSE>(A, abase + (size_t)(n0 + tr + 8 * i) * P.Fin + f0 + tc4);   // FUSE: every row is read by exactly one workgroup - nontemporal (NS step -2.7 % same-box)
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            int e = tid + NT_ * i;
            if (DV % NT_ != 0) e = e < DV ? e : DV - 1;
            drd[i] = load4<BF16IO, FUSE>(dYp, dybase + (size_t)(n0 + e / CQ) * P.Fout + o0 + (e % CQ) * 4);
        }
    };

    const long n_chunks = (n_end - n_begin) / WR;     // ALIGNED: whole chunks only
    if (n_chunks > 0) {
        fetch(n_begin, rt0, rd0);
        fetch(n_begin + (1 < n_chunks ? 1 : n_chunks - 1) * WR, rt1, rd1);
        if constexpr (PF == 3) fetch(n_begin + (2 < n_chunks ? 2 : n_chunks - 1) * WR, rt2, rd2);
    }
    const long n_pad = (n_chunks + PF - 1) / PF * PF;

    // this lane's element offset inside a block for the transpose reads: row (i >> 2) of its [4 n] group, 8 rows
    // further for the upper lane half, 8-byte chunk (i & 3) of the 16-column half (g & 1); i = lane & 15, g = lane >> 4
    const int frag_off = (((lane & 15) >> 2) + 8 * half) * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    // the same two reads in the swizzled dY image: the lane's 16-byte chunk ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1) moves
    // with the key of its row group - 2 * half for rows +0..3, 2 * half + 1 for rows +4..7 (+16 rows: same key)
    const int dchunk = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const int drow = (((lane & 15) >> 2) + 8 * half) * 32 + (lane & 1) * 4;
    const int dfrag0 = drow + ((dchunk ^ (2 * half)) << 3);
    const int dfrag1 = drow + 4 * 32 + ((dchunk ^ (2 * half + 1)) << 3);
    unsigned short* tw = TsT + (size_t)wave * NSPLIT * TPLANE;

    auto stage = [&](auto U, const long ci) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        f32x4 (&srt)[4] = *[&]() -> f32x4 (*)[4] {
            if constexpr (u == 0) return &rt0; else if constexpr (u == 1) return &rt1; else return &rt2; }();
        f32x4 (&srd)[RD] = *[&]() -> f32x4 (*)[RD] {
            if constexpr (u == 0) return &rd0; else if constexpr (u == 1) return &rd1; else return &rd2; }();
        __syncthreads();   // previous chunk's fragments fully consumed
        const bool live = ci < n_chunks;
        // split + row-major LDS image
#pragma unroll
        for (int i = 0; i < 4; ++i)
            split_store<NSPLIT>(tw + (tr + 8 * i) * 32 + tc4, TPLANE, srt[i]);
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            const int e = tid + NT_ * i;
            if (DV % NT_ == 0 || e < DV) {
                const int n = e / CQ, c4 = (e % CQ) * 4;
                split_store<NSPLIT>(DsT + (c4 >> 5) * DBLK + n * 32 + (((((c4 & 31) >> 3) ^ (n >> 2)) & 3) << 3) + (c4 & 7),
                                    DPLANE, srd[i]);
                if (live && do_db) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cs[i][j] += srd[i][j];
                }
            }
        }
        __syncthreads();
        {
            const long nx = ci + PF;
            fetch(n_begin + (nx < n_chunks ? nx : n_chunks - 1) * WR, srt, srd);
        }
        if (live && active) {
            const unsigned short* ta = tw + frag_off;
#pragma unroll
            for (int s2 = 0; s2 < WR / 16; ++s2) {
                const bf16x8_t ah = read_frag_tr(ta + 16 * s2 * 32);
                bf16x8_t am = ah, al = ah;
                if constexpr (NSPLIT == 3) {
                    am = read_frag_tr(ta + TPLANE + 16 * s2 * 32);
                    al = read_frag_tr(ta + 2 * TPLANE + 16 * s2 * 32);
                }
#pragma unroll
                for (int t = 0; t < NO; ++t) {
                    const unsigned short* bp = DsT + t * DBLK + 16 * s2 * 32;
                    const bf16x8_t bh = read_frag_tr2(bp + dfrag0, bp + dfrag1);
                    f32x16 a_ = acc[t];
                    if constexpr (NSPLIT == 3) {
                        const bf16x8_t bm = read_frag_tr2(bp + DPLANE + dfrag0, bp + DPLANE + dfrag1);
                        const bf16x8_t bl = read_frag_tr2(bp + 2 * DPLANE + dfrag0, bp + 2 * DPLANE + dfrag1);
                        a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a_, 0, 0, 0);
                        a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a_, 0, 0, 0);
                        a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a_, 0, 0, 0);
                        a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a_, 0, 0, 0);
                        a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a_, 0, 0, 0);
                    }
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a_, 0, 0, 0);
                    acc[t] = a_;
                }
            }
            if constexpr (FUSE) {
                // G tile of this wave: rows = the chunk's 32 nodes, cols = its 32 (k, f) channels, reduction over o
                f32x16 g;
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = 0.f;
                const unsigned short* da = DsT + (size_t)l31 * 32;                                // A: dY[n = l31][o ..]
                const int akey = (l31 >> 2) & 3;
                const unsigned short* wb = Wp + (size_t)(wave * 32 + l31) * WKS + 8 * half;        // B: W^T[col = l31][o ..]
#pragma unroll
                for (int s = 0; s < BNO / 16; ++s) {
                    const unsigned short* ap = da + ((16 * s) >> 5) * DBLK + ((((((16 * s) & 31) >> 3) + half) ^ akey) << 3);
                    const bf16x8_t xh = *reinterpret_cast<const bf16x8_t*>(ap);
                    const bf16x8_t xm = *reinterpret_cast<const bf16x8_t*>(ap + DPLANE);
                    const bf16x8_t xl = *reinterpret_cast<const bf16x8_t*>(ap + 2 * DPLANE);
                    const bf16x8_t wh = *reinterpret_cast<const bf16x8_t*>(wb + 16 * s);
                    const bf16x8_t wm = *reinterpret_cast<const bf16x8_t*>(wb + WPLANE + 16 * s);
                    const bf16x8_t wl = *reinterpret_cast<const bf16x8_t*>(wb + 2 * WPLANE + 16 * s);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wm, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wh, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wm, g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, g, 0, 0, 0);
                }
                float* Gp = (k == 0) ? static_cast<float*>(P.G0)
                                     : static_cast<float*>(P.Grest) + (size_t)(k - 1) * P.plane_stride;
                const long nrow = n_begin + ci * WR + 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    // plain stores: the adjoint pair reads these planes next, partly out of the Infinity Cache
                    Gp[(size_t)(nrow + (i & 3) + 8 * (i >> 2)) * P.Fin + f0 + l31] = g[i];
            }
        }
    };
    for (long cb = 0; cb < n_pad; cb += PF) {
        stage(std::integral_constant<int, 0>{}, cb);
        stage(std::integral_constant<int, 1>{}, cb + 1);
        if constexpr (PF == 3) stage(std::integral_constant<int, 2>{}, cb + 2);
    }

    float* out = P.partial + (size_t)blockIdx.x * (size_t)(Kd + 1) * P.Fout;
    if (active) {
#pragma unroll
        for (int t = 0; t < NO; ++t) {
            const int o = o0 + 32 * t + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = f0 + (i & 3) + 8 * (i >> 2) + 4 * half;
                out[(size_t)(k_out * P.Fin + f) * P.Fout + o] = acc[t][i];
            }
        }
    }
    if (do_db) {
        // column sums of dY (db).  Slot i of a thread is tile element tid + NT_*i; NT_ % CQ == 0, so all slots of a
        // thread belong to the same column quad tid % CQ: add them up, then combine the G threads of a quad.
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = cs[0][j];
#pragma unroll
            for (int ii = 1; ii < RD; ++ii) v += cs[ii][j];
            red[(tid / CQ) * BNO + (tid % CQ) * 4 + j] = v;
        }
        __syncthreads();
        for (int c = tid; c < BNO; c += NT_) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) v += red[g * BNO + c];
            out[(size_t)Kd * P.Fout + o0 + c] = v;
        }
    }
}


// bf16 storage: the operands ARE bf16, so the tile image is a raw copy - 64-row chunks (the same bytes per chunk
// and 16 B per lane as the fp32 path), ds_write_b128 staging, one MFMA term.  Same tiling / slabs / ring otherwise.
// FUSE: as in the fp32 kernel - the workgroup also writes the dgrad tile G_k[n, f] = sum_o dY[n, o] W[f, k, o] of the
// rows it streams (needs one o-tile covering all of Fout).  The W^T panel is a raw bf16 copy; the wave's G tile is
// converted with v_cvt_pk and transposed through the wave's own T rows (dead after the wgrad MFMAs of the chunk) so
// that it leaves as 16 bytes per lane.
template <int NW, int NO, bool FUSE = false>
__global__ __launch_bounds__(64 * NW, FUSE ? 2 : 1) void cheb_wgrad_bf16_kernel(const WgradParams P) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NT_ = 64 * NW;
    constexpr int BNO = 32 * NO;
    constexpr int WRB = 64;                             // rows per chunk
    constexpr int CO = BNO / 8;                         // column octets (16-byte vectors) per dY row
    constexpr int DV = WRB * CO;
    constexpr int RD = (DV + NT_ - 1) / NT_;
    constexpr int G = NT_ / CO;
    static_assert(NT_ % CO == 0, "column-sum layout");
    constexpr int PF = FUSE ? 2 : 3;
    constexpr int BLKB = WRB * 32;                      // one [64 n][32 ch] block (4 KiB)
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
    unsigned short* TsT = xs;                           // [NW][64 n][32 f]
    unsigned short* DsT = xs + (size_t)NW * BLKB;       // [NO][64 n][32 o]
    float* red = reinterpret_cast<float*>(DsT + (size_t)NO * BLKB);   // [G][BNO]
    constexpr int WKS = BNO + 8;                        // bf16 elements per W^T panel row
    unsigned short* Wp = reinterpret_cast<unsigned short*>(red + (size_t)G * BNO);   // FUSE: [NW * 32 cols][WKS]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.y * NW + wave;
    const int ntiles = P.K * P.tiles_per_plane;
    const bool active = tile < ntiles;
    const int k = active ? tile / P.tiles_per_plane : 0;
    const int f0 = active ? (tile - k * P.tiles_per_plane) * 32 : 0;
    const int zk = P.dy_planes > 1 ? blockIdx.z / P.otiles : 0;
    const int o0 = (blockIdx.z - zk * P.otiles) * BNO;
    const int Kd = (P.dy_planes > 1 ? P.dy_planes : P.K) * P.Fin;
    const int k_out = P.dy_planes > 1 ? zk : k;
    const bool do_db = blockIdx.y == 0 && zk == 0;
    const long n_begin = (long)blockIdx.x * P.rows_per_slab;
    const long n_end = (n_begin + P.rows_per_slab < P.N) ? n_begin + P.rows_per_slab : P.N;
    const uint16_t* A = static_cast<const uint16_t*>((k == 0) ? P.X : P.T) + ((k == 0) ? 0 : (size_t)(k - 1) * P.plane_stride);
    const uint16_t* dY = zk == 0 ? static_cast<const uint16_t*>(P.dY)
                                 : static_cast<const uint16_t*>(P.dY1) + (size_t)(zk - 1) * P.dy_plane_stride;

    if constexpr (FUSE) {
        const uint16_t* Wsrc = static_cast<const uint16_t*>(P.W);
        for (int e = tid; e < NW * 32 * BNO; e += NT_) {
            const int c = e / BNO, o = e - c * BNO;
            const int t_ = blockIdx.y * NW + (c >> 5);
            unsigned short v = 0;
            if (t_ < ntiles) {
                const int kk = t_ / P.tiles_per_plane, ff = (t_ - kk * P.tiles_per_plane) * 32 + (c & 31);
                v = Wsrc[((size_t)ff * P.K + kk) * P.Fout + o];
                if (P.fold && kk == P.K - 3)
                    v = f32_to_bf16(bf16_to_f32(v) - bf16_to_f32(Wsrc[((size_t)ff * P.K + (P.K - 1)) * P.Fout + o]));
            }
            Wp[c * WKS + o] = v;
        }
    }
    f32x16 acc[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float cs[RD][8];
#pragma unroll
    for (int i = 0; i < RD; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[i][j] = 0.f;

    const int tr = lane >> 2, tc8 = (lane & 3) * 8;   // T tile (per wave): rows tr + 16*i (i<4), channels tc8..+7
    u32x4 rt0[4], rt1[4], rt2[4], rd0[RD], rd1[RD], rd2[RD];   // (the third slot is dead code when PF == 2)
    auto fetch = [&](long n0, u32x4 (&drt)[4], u32x4 (&drd)[RD]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            drt[i] = *reinterpret_cast<const u32x4*>(A + (size_t)(n0 + tr + 16 * i) * P.Fin + f0 + tc8);
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            int e = tid + NT_ * i;
            if (DV % NT_ != 0) e = e < DV ? e : DV - 1;
            drd[i] = *reinterpret_cast<const u32x4*>(dY + (size_t)(n0 + e / CO) * P.Fout + o0 + (e % CO) * 8);
        }
    };
    const long n_chunks = (n_end - n_begin) / WRB;
    if (n_chunks > 0) {
        fetch(n_begin, rt0, rd0);
        fetch(n_begin + (1 < n_chunks ? 1 : n_chunks - 1) * WRB, rt1, rd1);
        if constexpr (PF == 3) fetch(n_begin + (2 < n_chunks ? 2 : n_chunks - 1) * WRB, rt2, rd2);
    }
    const long n_pad = (n_chunks + PF - 1) / PF * PF;
    const int frag_off = (((lane & 15) >> 2) + 8 * half) * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    unsigned short* tw = TsT + (size_t)wave * BLKB;

    auto stage = [&](auto U, const long ci) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        u32x4 (&srt)[4] = *[&]() -> u32x4 (*)[4] {
            if constexpr (u == 0) return &rt0; else if constexpr (u == 1) return &rt1; else return &rt2; }();
        u32x4 (&srd)[RD] = *[&]() -> u32x4 (*)[RD] {
            if constexpr (u == 0) return &rd0; else if constexpr (u == 1) return &rd1; else return &rd2; }();
        __syncthreads();
        const bool live = ci < n_chunks;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tw + (tr + 16 * i) * 32 + tc8) = srt[i];
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            const int e = tid + NT_ * i;
            if (DV % NT_ == 0 || e < DV) {
                const int n = e / CO, c8 = (e % CO) * 8;
                *reinterpret_cast<u32x4*>(DsT + (c8 >> 5) * BLKB + n * 32 + (c8 & 31)) = srd[i];
                if (live && do_db) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        cs[i][2 * j] += __uint_as_float(srd[i][j] << 16);
                        cs[i][2 * j + 1] += __uint_as_float(srd[i][j] & 0xffff0000u);
                    }
                }
            }
        }
        __syncthreads();
        {
            const long nx = ci + PF;
            fetch(n_begin + (nx < n_chunks ? nx : n_chunks - 1) * WRB, srt, srd);
        }
        if (live && active) {
            const unsigned short* ta = tw + frag_off;
            const unsigned short* db = DsT + frag_off;
#pragma unroll
            for (int s2 = 0; s2 < WRB / 16; ++s2) {
                const bf16x8_t a = read_frag_tr(ta + 16 * s2 * 32);
#pragma unroll
                for (int t = 0; t < NO; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, read_frag_tr(db + t * BLKB + 16 * s2 * 32), acc[t], 0, 0, 0);
            }
            if constexpr (FUSE) {
                f32x16 g0, g1;     // rows 0..31 / 32..63 of the chunk x this wave's 32 (k, f) channels
#pragma unroll
                for (int i = 0; i < 16; ++i) { g0[i] = 0.f; g1[i] = 0.f; }
                const unsigned short* wb = Wp + (size_t)(wave * 32 + l31) * WKS + 8 * half;
#pragma unroll
                for (int s = 0; s < BNO / 16; ++s) {
                    const unsigned short* ap = DsT + ((16 * s) >> 5) * BLKB + (size_t)l31 * 32 + ((16 * s) & 31) + 8 * half;
                    const bf16x8_t bw = *reinterpret_cast<const bf16x8_t*>(wb + 16 * s);
                    g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap), bw, g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + 32 * 32), bw, g1, 0, 0, 0);
                }
                // bf16 tile -> the wave's own T rows (its fragments of this chunk are consumed) -> 16 B per lane out
                typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i & 3) + 8 * (i >> 2) + 4 * half;
                    const f32x2_ v = {g0[i], g1[i]};
                    const bf16x2_ pk = __builtin_convertvector(v, bf16x2_);
                    const unsigned int u = __builtin_bit_cast(unsigned int, pk);
                    tw[rr * 32 + l31] = (unsigned short)(u & 0xffffu);
                    tw[(rr + 32) * 32 + l31] = (unsigned short)(u >> 16);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint16_t* Gp = (k == 0) ? static_cast<uint16_t*>(P.G0)
                                        : static_cast<uint16_t*>(P.Grest) + (size_t)(k - 1) * P.plane_stride;
                const long nrow = n_begin + ci * WRB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = (lane >> 2) + 16 * j, c8 = (lane & 3) * 8;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(tw + rr * 32 + c8);
                    *reinterpret_cast<u32x4*>(Gp + (size_t)(nrow + rr) * P.Fin + f0 + c8) = v;
                }
            }
        }
    };
    for (long cb = 0; cb < n_pad; cb += PF) {
        stage(std::integral_constant<int, 0>{}, cb);
        stage(std::integral_constant<int, 1>{}, cb + 1);
        if constexpr (PF == 3) stage(std::integral_constant<int, 2>{}, cb + 2);
    }

    float* out = P.partial + (size_t)blockIdx.x * (size_t)(Kd + 1) * P.Fout;
    if (active) {
#pragma unroll
        for (int t = 0; t < NO; ++t) {
            const int o = o0 + 32 * t + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = f0 + (i & 3) + 8 * (i >> 2) + 4 * half;
                out[(size_t)(k_out * P.Fin + f) * P.Fout + o] = acc[t][i];
            }
        }
    }
    if (do_db) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = cs[0][j];
#pragma unroll
            for (int ii = 1; ii < RD; ++ii) v += cs[ii][j];
            red[(tid / CO) * BNO + (tid % CO) * 8 + j] = v;
        }
        __syncthreads();
        for (int c = tid; c < BNO; c += NT_) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) v += red[g * BNO + c];
            out[(size_t)Kd * P.Fout + o0 + c] = v;
        }
    }
}

template <int NW, int NO, bool FUSE = false>
int launch_wbf16(WgradParams& P, int groups, int otiles, int64_t max_slabs, int64_t* S_out, hipStream_t stream) {
    constexpr int NT_ = 64 * NW;
    constexpr int BNO = 32 * NO;
    constexpr int G = NT_ / (BNO / 8);
    const size_t lds = ((size_t)NW + NO) * 64 * 32 * 2 + (size_t)G * BNO * 4 + (FUSE ? (size_t)NW * 32 * (BNO + 8) * 2 : 0);
    const void* kfn = (const void*)cheb_wgrad_bf16_kernel<NW, NO, FUSE>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DSW_ERR_LAUNCH;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, NT_, lds) != hipSuccess || occ < 1) occ = 1;
    const int zdim = otiles * (P.dy_planes > 1 ? P.dy_planes : 1);
    P.otiles = otiles;
    int64_t want = 256L * occ / ((int64_t)groups * zdim);
    if (want < 8) want = 8;   // many (k, f) x column tiles already fill the CUs: fewer, longer slabs = fewer partials to reduce
    if (want > max_slabs) want = max_slabs;
    int64_t rps = (P.N + want - 1) / want;
    rps = ((rps + 63) / 64) * 64;
    if (rps < 256) rps = 256;
    const int64_t S = (P.N + rps - 1) / rps;
    P.rows_per_slab = rps;
    *S_out = S;
    dim3 grid((unsigned)S, (unsigned)groups, (unsigned)zdim);
    DSW_LAUNCH((cheb_wgrad_bf16_kernel<NW, NO, FUSE>), grid, dim3(NT_), lds, stream, P);
    return dsw_check_launch();
}

template <bool BF16IO, int NSPLIT, int NW, int NO, bool FUSE = false>
int launch_wx3(WgradParams& P, int groups, int otiles, int64_t max_slabs, int64_t* S_out, hipStream_t stream) {
    constexpr int NT_ = 64 * NW;
    constexpr int BNO = 32 * NO;
    constexpr int G = NT_ / (BNO / 4);
    const size_t lds = ((size_t)NW * NSPLIT * BLK + (size_t)NSPLIT * NO * (BLK + 64)) * 2 + (size_t)G * BNO * 4 +
                       (FUSE ? (size_t)3 * NW * 32 * (BNO + 8) * 2 : 0);
    const void* kfn = (const void*)cheb_wgrad_x3_kernel<BF16IO, NSPLIT, NW, NO, FUSE>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DSW_ERR_LAUNCH;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, NT_, lds) != hipSuccess || occ < 1) occ = 1;
    const int zdim = otiles * (P.dy_planes > 1 ? P.dy_planes : 1);
    P.otiles = otiles;
    int64_t want = 256L * occ / ((int64_t)groups * zdim);
    if (want < 8) want = 8;   // many (k, f) x column tiles already fill the CUs: fewer, longer slabs = fewer partials to reduce
    if (want > max_slabs) want = max_slabs;
    int64_t rps = (P.N + want - 1) / want;
    rps = ((rps + WR - 1) / WR) * WR;
    if (rps < 4 * WR) rps = 4 * WR;
    const int64_t S = (P.N + rps - 1) / rps;
    P.rows_per_slab = rps;
    *S_out = S;
    dim3 grid((unsigned)S, (unsigned)groups, (unsigned)zdim);
    DSW_LAUNCH((cheb_wgrad_x3_kernel<BF16IO, NSPLIT, NW, NO, FUSE>), grid, dim3(NT_), lds, stream, P);
    return dsw_check_launch();
}

}  // namespace

// Takes the launch (returns 1) for aligned problems: Fin % 32 == 0, Fout % 64 == 0, N % 32 == 0, 16-byte rows.
// The (k, f)-tile / column tiling is chosen here: up to 8 waves per workgroup, 128 columns when Fout % 128 == 0.
int dsw_wgrad_x3_try_launch(WgradParams& P, int bf16, int64_t max_slabs, int64_t* S_out, hipStream_t stream, int* rc) {
    static const char* x3env = dsw_diag_env("DSW_GEMM_X3");   // "0": exact fp32 MFMA kernels (diagnostics / A-B)
    if (x3env && x3env[0] == '0') return 0;
    const int ntiles = P.K * P.tiles_per_plane;
    const int groups = (ntiles + 7) / 8;
    const int nw = (ntiles + groups - 1) / groups;
    const bool wide = P.Fout % 128 == 0 && nw >= 3;   // few waves: the 128-column dY tile would not fit their registers
    const bool slim = !bf16 && P.Fout % 64 != 0;       // fp32, Fout = 32 (mod 64): 32-column o-tiles
    if (P.Fout % (slim ? 32 : 64) != 0) return 0;
    const int otiles = wide ? P.Fout / 128 : slim ? P.Fout / 32 : P.Fout / 64;
    if (slim) {
#define DSW_WX3_SLIM(NW_)                                                                                         \
    case NW_:                                                                                                     \
        *rc = launch_wx3<false, 3, NW_, 1>(P, groups, otiles, max_slabs, S_out, stream);                            \
        return 1;
        switch (nw) {
            DSW_WX3_SLIM(1) DSW_WX3_SLIM(2) DSW_WX3_SLIM(3) DSW_WX3_SLIM(4) DSW_WX3_SLIM(5) DSW_WX3_SLIM(6) DSW_WX3_SLIM(7)
            DSW_WX3_SLIM(8)
        }
#undef DSW_WX3_SLIM
        return 0;
    }
    // bf16 storage, whole 64-row chunks, 16-byte aligned 8-channel groups: raw-copy staging kernel
    const bool raw = bf16 && P.N % 64 == 0 && P.Fin % 8 == 0 && P.Fout % 8 == 0 && dsw_aligned16(P.X) &&
                     (P.K == 1 || dsw_aligned16(P.T)) && dsw_aligned16(P.dY) && P.plane_stride % 8 == 0 &&
                     (P.dy_planes <= 1 || (dsw_aligned16(P.dY1) && P.dy_plane_stride % 8 == 0));
    if (raw) {
#define DSW_WB(NW_)                                                                                               \
    case NW_:                                                                                                     \
        *rc = wide ? launch_wbf16<NW_, 4>(P, groups, otiles, max_slabs, S_out, stream)                              \
                   : launch_wbf16<NW_, 2>(P, groups, otiles, max_slabs, S_out, stream);                             \
        return 1;
        switch (nw) {
            DSW_WB(1) DSW_WB(2) DSW_WB(3) DSW_WB(4) DSW_WB(5) DSW_WB(6) DSW_WB(7) DSW_WB(8)
        }
#undef DSW_WB
    }
#define DSW_WX3(NW_)                                                                                              \
    case NW_:                                                                                                     \
        *rc = bf16 ? (wide ? launch_wx3<true, 1, NW_, 4>(P, groups, otiles, max_slabs, S_out, stream)               \
                           : launch_wx3<true, 1, NW_, 2>(P, groups, otiles, max_slabs, S_out, stream))              \
                   : (wide ? launch_wx3<false, 3, NW_, 4>(P, groups, otiles, max_slabs, S_out, stream)              \
                           : launch_wx3<false, 3, NW_, 2>(P, groups, otiles, max_slabs, S_out, stream));            \
        return 1;
#define DSW_WX3_NARROW(NW_)                                                                                       \
    case NW_:                                                                                                     \
        *rc = bf16 ? launch_wx3<true, 1, NW_, 2>(P, groups, otiles, max_slabs, S_out, stream)                       \
                   : launch_wx3<false, 3, NW_, 2>(P, groups, otiles, max_slabs, S_out, stream);                     \
        return 1;
    switch (nw) {
        DSW_WX3_NARROW(1) DSW_WX3_NARROW(2) DSW_WX3(3) DSW_WX3(4) DSW_WX3(5) DSW_WX3(6) DSW_WX3(7) DSW_WX3(8)
    }
#undef DSW_WX3
#undef DSW_WX3_NARROW
    return 0;
}

// wgrad with the dgrad of the same rows fused in (fp32, aligned, plain basis-first backward).  Applies when one
// workgroup sees all of Fout (64 or 128 columns) and all (k, f) tiles (<= 4 waves) and the W^T panel keeps two
// workgroups per CU: the small layers of the path (north-star shape 32 -> 64, K = 3).  Returns 1 when it took the launch.
int dsw_wgrad_dgrad_fused_try_launch(WgradParams& P, int bf16, int64_t max_slabs, int64_t* S_out, hipStream_t stream,
                                     int* rc) {
    static const char* x3env = dsw_diag_env("DSW_GEMM_X3");
    if (x3env && x3env[0] == '0') return 0;
    static const char* fenv = dsw_diag_env("DSW_BWD_FUSED");   // "0": separate dgrad and wgrad launches (A-B)
    if (fenv && fenv[0] == '0') return 0;
    if (P.dy_planes > 1 || !P.W || !P.G0 || (P.K > 1 && !P.Grest)) return 0;
    const int ntiles = P.K * P.tiles_per_plane;
    if (P.Fin % 32 != 0) return 0;
    if (bf16) {
        // raw-copy kernel: whole 64-row chunks, one o-tile = all of Fout (64 or 128 columns), 3..8 waves per group
        if (P.N % 64 != 0 || (P.Fout != 64 && P.Fout != 128)) return 0;
        const int groups = (ntiles + 7) / 8;
        const int nw = (ntiles + groups - 1) / groups;
        if (nw < 3) return 0;
        const bool wide = P.Fout == 128;
#define DSW_WFB(NW_)                                                                                       \
    case NW_:                                                                                              \
        *rc = wide ? launch_wbf16<NW_, 4, true>(P, groups, 1, max_slabs, S_out, stream)                    \
                   : launch_wbf16<NW_, 2, true>(P, groups, 1, max_slabs, S_out, stream);                   \
        return 1;
        switch (nw) {
            DSW_WFB(3) DSW_WFB(4) DSW_WFB(5) DSW_WFB(6) DSW_WFB(7) DSW_WFB(8)
        }
#undef DSW_WFB
        return 0;
    }
    if (ntiles < 3 || ntiles > 4) return 0;               // fp32: 1-2 wave workgroups would spill registers
    if (P.Fout != 64 && P.Fout != 32) return 0;          // one 64- or 32-column o-tile (128 would need 2x the panel and registers)
#define DSW_WF(NW_)                                                                                        \
    case NW_:                                                                                              \
        *rc = P.Fout == 64 ? launch_wx3<false, 3, NW_, 2, true>(P, 1, 1, max_slabs, S_out, stream)         \
                           : launch_wx3<false, 3, NW_, 1, true>(P, 1, 1, max_slabs, S_out, stream);        \
        return 1;
    switch (ntiles) {
        DSW_WF(3) DSW_WF(4)
    }
#undef DSW_WF
    return 0;
}
