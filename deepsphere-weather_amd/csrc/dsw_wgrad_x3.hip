// wgrad on the bf16 matrix pipe:  partial[s][(k, f)][o] = sum_{n in slab s} T_k[n, f] * dY[n, o]
//
// Both MFMA operands of this contraction are "transposed": the reduction index n is the ROW index of
// the row-major activations, while v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction elements per
// lane.  So the staging pass does the transpose together with the precision split: every fp32 element
// is split exactly into 3 bf16 terms (or, for bf16 storage, taken as is) and written to LDS as
// [plane][channel][n] with n contiguous; operand fragments are then plain 8-byte LDS reads.  The split is
// done once per element by the thread that staged it (the dY tile is shared by all waves of the
// workgroup), the six leading cross terms accumulate in fp32 - same accuracy argument as dsw_gemm_x3.hip.
// Structure otherwise as cheb_wgrad_kernel: NW waves = NW (k, f)-tiles of 32 rows x one 64-column o-tile,
// row slabs -> fp32 partials -> cheb_wgrad_reduce_kernel (deterministic), 3-deep register prefetch ring.
#include "dsw_gemm_common.h"

using namespace dsw_gemm;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int KSN = 36;   // bf16 elements per LDS row (32 n + 4 pad): 72-byte rows -> conflict-free b64 reads

static __device__ __forceinline__ float trunc_bf16(float f) {
    return __uint_as_float(__float_as_uint(f) & 0xffff0000u);
}

template <bool BF16IO>
static __device__ __forceinline__ f32x4 load4(const void* p, size_t i) {
    if constexpr (BF16IO) {
        const u32x2 t = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(p) + i);
        f32x4 v;
        v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
        v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
        return v;
    } else {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(p) + i);
    }
}

// split v into NSPLIT bf16 terms and scatter them to plane[p][(ch + j) * KSN + n], j = 0..3
template <int NSPLIT>
static __device__ __forceinline__ void split_store(unsigned short* base, const int plane_elems, const int ch,
                                                   const int n, const f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = v[j];
        const float h = trunc_bf16(x);
        unsigned short* dst = base + (size_t)(ch + j) * KSN + n;
        dst[0] = (unsigned short)(__float_as_uint(h) >> 16);
        if constexpr (NSPLIT == 3) {
            const float r1 = x - h, m = trunc_bf16(r1), l = r1 - m;
            dst[plane_elems] = (unsigned short)(__float_as_uint(m) >> 16);
            dst[2 * plane_elems] = (unsigned short)(__float_as_uint(l) >> 16);
        }
    }
}

static __device__ __forceinline__ bf16x8_t read_frag(const unsigned short* p) {
    const u32x2 a = *reinterpret_cast<const u32x2*>(p);
    const u32x2 b = *reinterpret_cast<const u32x2*>(p + 4);
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ u = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8_t, u);
}

template <bool BF16IO, int NSPLIT, int NW>
__global__ __launch_bounds__(64 * NW) void cheb_wgrad_x3_kernel(const WgradParams P) {
    constexpr int NT_ = 64 * NW;
    constexpr int RD = (WR * BN / 4 + NT_ - 1) / NT_;   // float4 of the dY tile per thread
    constexpr int PF = 3;
    constexpr int TPLANE = 32 * KSN;                    // one (wave, plane) T^T tile: [32 f][KSN]
    constexpr int DPLANE = BN * KSN;                    // one dY^T plane: [64 o][KSN]
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];
    unsigned short* TsT = xs;                                  // [NW][NSPLIT][32][KSN]
    unsigned short* DsT = xs + (size_t)NW * NSPLIT * TPLANE;   // [NSPLIT][64][KSN]
    float* red = reinterpret_cast<float*>(DsT + (size_t)NSPLIT * DPLANE);   // [NT_/16][64] column-sum scratch

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.y * NW + wave;
    const int ntiles = P.K * P.tiles_per_plane;
    const bool active = tile < ntiles;
    const int k = active ? tile / P.tiles_per_plane : 0;
    const int f0 = active ? (tile - k * P.tiles_per_plane) * 32 : 0;
    const int o0 = blockIdx.z * BN;
    const int Kd = P.K * P.Fin;
    const long n_begin = (long)blockIdx.x * P.rows_per_slab;
    const long n_end = (n_begin + P.rows_per_slab < P.N) ? n_begin + P.rows_per_slab : P.N;
    const void* A = (k == 0) ? P.X : P.T;
    const size_t abase = (k == 0) ? 0 : (size_t)(k - 1) * P.plane_stride;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
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
            drt[i] = load4<BF16IO>(A, abase + (size_t)(n0 + tr + 8 * i) * P.Fin + f0 + tc4);
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            int e = tid + NT_ * i;
            if ((WR * BN / 4) % NT_ != 0) e = e < WR * BN / 4 ? e : WR * BN / 4 - 1;
            drd[i] = load4<BF16IO>(P.dY, (size_t)(n0 + (e >> 4)) * P.Fout + o0 + (e & 15) * 4);
        }
    };

    const long n_chunks = (n_end - n_begin) / WR;     // ALIGNED: whole chunks only
    if (n_chunks > 0) {
        fetch(n_begin, rt0, rd0);
        fetch(n_begin + (1 < n_chunks ? 1 : n_chunks - 1) * WR, rt1, rd1);
        fetch(n_begin + (2 < n_chunks ? 2 : n_chunks - 1) * WR, rt2, rd2);
    }
    const long n_pad = (n_chunks + PF - 1) / PF * PF;

    auto stage = [&](auto U, const long ci) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        f32x4 (&srt)[4] = *[&]() -> f32x4 (*)[4] {
            if constexpr (u == 0) return &rt0; else if constexpr (u == 1) return &rt1; else return &rt2; }();
        f32x4 (&srd)[RD] = *[&]() -> f32x4 (*)[RD] {
            if constexpr (u == 0) return &rd0; else if constexpr (u == 1) return &rd1; else return &rd2; }();
        __syncthreads();   // previous chunk's fragments fully consumed
        const bool live = ci < n_chunks;
        // transpose + split into LDS
        if (P.dbg != 3)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            split_store<NSPLIT>(TsT + (size_t)wave * NSPLIT * TPLANE, TPLANE, tc4, tr + 8 * i, srt[i]);
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            const int e = tid + NT_ * i;
            if (e < WR * BN / 4 && P.dbg != 3) {
                split_store<NSPLIT>(DsT, DPLANE, (e & 15) * 4, e >> 4, srd[i]);
                if (live && blockIdx.y == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cs[i][j] += srd[i][j];
                }
            }
        }
        __syncthreads();
        if (P.dbg != 4) {
            const long nx = ci + PF;
            fetch(n_begin + (nx < n_chunks ? nx : n_chunks - 1) * WR, srt, srd);
        }
        if (live && active && P.dbg != 2) {
            const unsigned short* ta = TsT + (size_t)wave * NSPLIT * TPLANE + (size_t)l31 * KSN + 8 * half;
            const unsigned short* db = DsT + (size_t)l31 * KSN + 8 * half;
#pragma unroll
            for (int s2 = 0; s2 < WR / 16; ++s2) {
                const bf16x8_t ah = read_frag(ta + 16 * s2);
                const bf16x8_t b0h = read_frag(db + 16 * s2);
                const bf16x8_t b1h = read_frag(db + (size_t)32 * KSN + 16 * s2);
                if constexpr (NSPLIT == 3) {
                    const bf16x8_t am = read_frag(ta + TPLANE + 16 * s2), al = read_frag(ta + 2 * TPLANE + 16 * s2);
                    const bf16x8_t b0m = read_frag(db + DPLANE + 16 * s2), b0l = read_frag(db + 2 * DPLANE + 16 * s2);
                    const bf16x8_t b1m = read_frag(db + DPLANE + (size_t)32 * KSN + 16 * s2);
                    const bf16x8_t b1l = read_frag(db + 2 * DPLANE + (size_t)32 * KSN + 16 * s2);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0m, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1m, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0h, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1h, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0m, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1m, acc1, 0, 0, 0);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
            }
        }
    };
    for (long cb = 0; cb < n_pad; cb += PF) {
        stage(std::integral_constant<int, 0>{}, cb);
        stage(std::integral_constant<int, 1>{}, cb + 1);
        stage(std::integral_constant<int, 2>{}, cb + 2);
    }

    float* out = P.partial + (size_t)blockIdx.x * (size_t)(Kd + 1) * P.Fout;
    if (active) {
        const int oA = o0 + l31, oB = o0 + 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = f0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            const size_t off = (size_t)(k * P.Fin + f) * P.Fout;
            out[off + oA] = acc0[i];
            out[off + oB] = acc1[i];
        }
    }
    if (blockIdx.y == 0) {
        // column sums of dY (db): thread e summed rows (e>>4) + multiples; combine the NT_/16 row groups
        __syncthreads();
        constexpr int G = NT_ / 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = cs[0][j];
#pragma unroll
            for (int ii = 1; ii < RD; ++ii) v += cs[ii][j];   // same columns (NT_ % 16 == 0), other rows
            red[(tid >> 4) * BN + (tid & 15) * 4 + j] = v;
        }
        __syncthreads();
        if (tid < BN) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) v += red[g * BN + tid];
            out[(size_t)Kd * P.Fout + o0 + tid] = v;
        }
    }
}

template <bool BF16IO, int NSPLIT, int NW>
int launch_wx3(WgradParams& P, int groups, int otiles, int64_t max_slabs, int64_t* S_out, hipStream_t stream) {
    constexpr int NT_ = 64 * NW;
    const size_t lds = ((size_t)NW * NSPLIT * 32 * KSN + (size_t)NSPLIT * BN * KSN) * 2 + (size_t)(NT_ / 16) * BN * 4;
    const void* kfn = (const void*)cheb_wgrad_x3_kernel<BF16IO, NSPLIT, NW>;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, NT_, lds) != hipSuccess || occ < 1) occ = 1;
    int64_t want = 256L * occ / ((int64_t)groups * otiles);
    if (want < 32) want = 32;
    if (want > max_slabs) want = max_slabs;
    int64_t rps = (P.N + want - 1) / want;
    rps = ((rps + WR - 1) / WR) * WR;
    if (rps < 4 * WR) rps = 4 * WR;
    const int64_t S = (P.N + rps - 1) / rps;
    P.rows_per_slab = rps;
    { static const char* d = getenv("DSW_DBG"); P.dbg = d ? atoi(d) : 0; }
    *S_out = S;
    dim3 grid((unsigned)S, (unsigned)groups, (unsigned)otiles);
    hipLaunchKernelGGL((cheb_wgrad_x3_kernel<BF16IO, NSPLIT, NW>), grid, dim3(NT_), lds, stream, P);
    return dsw_check_launch();
}

}  // namespace

// Takes the launch (returns 1) for aligned problems: Fin % 32 == 0, Fout % 64 == 0, N % 32 == 0, 16-byte rows.
int dsw_wgrad_x3_try_launch(WgradParams& P, int nw, int groups, int otiles, int bf16, int64_t max_slabs, int64_t* S_out,
                            hipStream_t stream, int* rc) {
    static const char* x3env = getenv("DSW_GEMM_X3");   // "0": exact fp32 MFMA kernels (diagnostics / A-B)
    if (x3env && x3env[0] == '0') return 0;
#define DSW_WX3(NW_)                                                                                              \
    case NW_:                                                                                                     \
        *rc = bf16 ? launch_wx3<true, 1, NW_>(P, groups, otiles, max_slabs, S_out, stream)                         \
                   : launch_wx3<false, 3, NW_>(P, groups, otiles, max_slabs, S_out, stream);                       \
        return 1;
    switch (nw) {
        DSW_WX3(1) DSW_WX3(2) DSW_WX3(3) DSW_WX3(4)
    }
#undef DSW_WX3
    return 0;
}
