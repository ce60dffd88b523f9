// fp32 channel-mix GEMM evaluated on the bf16 matrix pipe by exact 3-way operand splitting.
#include "dsw_gemm_common.h"

using namespace dsw_gemm;

namespace {
// STREAM variants (IO bit 3): the A operand is larger than half of the 256 MB Infinity Cache - its rows are read with
// nontemporal loads (they cannot stay resident anyway, and streaming them past the caches leaves those to the W panel
// and to what the neighbouring kernels exchange).  Same-box: C3 -1.8 %, C5 -1.7 %, k = 20 north-star -2.3 %; the UNet's
// layers (A <= 75 MB, re-read by up to three column tiles out of L2) keep cached loads (+0.3 % with nontemporal ones).
template <bool NT>
static __device__ __forceinline__ f32x4 lda16(const void* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}


// ---------------------------------------------------------------------------------------------
// fp32 contraction on the bf16 matrix pipe ("x3 split"): the fp32 MFMA runs at the VALU rate
// (157 TFLOP/s, 64 cycles per 32x32x2), 16x slower than v_mfma_f32_32x32x16_bf16.  Each fp32
// operand is split EXACTLY into three bf16 terms  x = x_h + x_m + x_l  (8+8+8 mantissa bits, by
// truncation, residuals computed exactly in fp32) and the product is evaluated with the six
// leading terms  a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_m b_m + a_l b_h  in fp32 accumulators.
// The dropped terms are O(2^-24 |a||b|), i.e. fp32-rounding class: the result is an fp32 GEMM
// (measured max-rel error vs fp64 is within the same 2e-6 bound the tests apply to the exact-fp32
// path) at 6/16 of the matrix-pipe time.  Same tiling / ring / wave-local staging as the resident
// ts_gemm kernel; the B panel is pre-split once into three bf16 planes in LDS, stored [plane][col][k]
// (k contiguous, rows padded by 16 B -> conflict-free ds_read_b128).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

static __device__ __forceinline__ bf16x8_t pack_bf16x8(const float (&f)[8]) {
    // truncating fp32 -> bf16 of 8 values: v_perm_b32 picks the two high bytes of each pair
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

// NSPLIT = 3: fp32 storage, exact 3-way split (6 MFMA terms).  NSPLIT = 1: bf16 storage - the operands
// ARE bf16, one MFMA term is exact (fp32 accumulate), i.e. the plain bf16 tensor-core GEMM of the path.
// IO bit 0: bf16 rows (else fp32).  IO bit 1 (bf16 only): 64-element chunks - the same 128-byte row segments /
// 16 B per lane as the fp32 path, staged in LDS as raw bf16 and fed to the MFMA without any conversion (needs
// Fin % 64 == 0; otherwise 32-element chunks, 8 B per lane).  IO bit 2 (bf16, NT even): packed epilogue - each
// pair of 32-column MFMA tiles holds the INTERLEAVED columns (2l, 2l+1) of a 64-column group, so a lane converts
// its two accumulators with one v_cvt_pk_bf16_f32, the wave transposes the group through its private staging
// rows in LDS and writes whole 128-byte output rows with 16 B per lane (instead of 2-byte scalar stores, which
// cost a third of the bf16 mix and dgrad kernels).
// NWV waves per workgroup (each owns 32 rows of the BM = 32 * NWV row tile): 8 when the W panel is so large that
// only one workgroup fits a CU, so that the CU still has 8 waves of loads in flight.
template <int NT, int NSPLIT, int IO, int NWV>
__global__ __launch_bounds__(64 * NWV) void ts_gemm_x3_kernel(const TsGemmParams P) {
    constexpr int BM = 32 * NWV;                     // shadows the namespace constant
    constexpr int NTH = 64 * NWV;
    constexpr bool BF16IO = (IO & 1) != 0;
    constexpr bool WIDE = (IO & 2) != 0;
    constexpr bool EPI = (IO & 4) != 0;
    constexpr bool RES = (IO & 16) != 0;     // epilogue operands (scale, residual): their own instantiation - the extra
                                             // live registers must not cost the plain kernels their second wave per SIMD
    constexpr bool STREAM = (IO & 8) != 0;
    static_assert(!EPI || (BF16IO && NT % 2 == 0), "packed epilogue: bf16 rows, tile pairs");
    constexpr int BK = WIDE ? 64 : dsw_gemm::BK;     // reduction elements per chunk (shadows the namespace constant)
    constexpr int BNT = 32 * NT;
    constexpr int PF = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                                  // [BM][LDA] fp32 staging
    unsigned short* Bt = reinterpret_cast<unsigned short*>(smem + BM * LDA);  // [NSPLIT][BNT][KS] bf16

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int n_total = P.n_planes_c * P.n_per_plane;
    const int col0 = blockIdx.y * BNT;
    const int chunks = P.kd_per_plane / BK;             // ALIGNED: kd_per_plane % 32 == 0
    const int total = P.n_planes_a * chunks;
    const int KS = total * BK + 8;                      // padded k-stride of a B row (bf16 elements)
    const long row_tiles = (P.M + BM - 1) / BM;

    // tile column jj (MFMA tile jj / 32, lane jj % 32) -> output column of this workgroup's panel
    auto tile_col = [](const int jj) __attribute__((always_inline)) {
        if constexpr (EPI) return 64 * (jj >> 6) + 2 * (jj & 31) + ((jj >> 5) & 1);
        else return jj;
    };
    // B panel: split once
    for (int e = tid; e < total * BK * BNT; e += NTH) {
        const int r = e / BNT, jj = e - r * BNT;        // r = flattened reduction index, jj = tile column
        const int pa = r / P.kd_per_plane, kd = r - pa * P.kd_per_plane;
        const int j = col0 + tile_col(jj);
        float v = 0.f;
        if (j < n_total) {
            const int q = j / P.n_per_plane, n = j - q * P.n_per_plane;
            v = ld1<BF16IO>(P.Bsrc, (size_t)((long)pa * P.b_sp + (long)q * P.b_sq + (long)kd * P.b_skd + (long)n * P.b_sn));
        }
        const float h = trunc_bf16(v), r1 = v - h, m = trunc_bf16(r1), l = trunc_bf16(r1 - m);
        Bt[(0 * BNT + jj) * KS + r] = (unsigned short)(__float_as_uint(h) >> 16);
        if constexpr (NSPLIT == 3) {
            Bt[(1 * BNT + jj) * KS + r] = (unsigned short)(__float_as_uint(m) >> 16);
            Bt[(2 * BNT + jj) * KS + r] = (unsigned short)(__float_as_uint(l) >> 16);
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    bool col_ok[NT];
    char* col_ptr[NT];
    float col_bias[NT];
    const char* col_res[NT];
    const float escale = RES ? epi_scale<BF16IO>(P) : 1.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int j = col0 + tile_col(32 * nt + l31);
        col_ok[nt] = j < n_total;
        const int q = col_ok[nt] ? j / P.n_per_plane : 0;
        const int n = col_ok[nt] ? j - q * P.n_per_plane : 0;
        char* base = static_cast<char*>((q == 0) ? P.C0 : P.C1);
        const size_t cbase = (q == 0) ? 0 : (size_t)(q - 1) * P.c_plane_stride;
        col_ptr[nt] = base + (cbase + (size_t)n) * (BF16IO ? 2 : 4);
        col_bias[nt] = (P.bias != nullptr && !(P.bias_plane0 && q != 0)) ? ld1<BF16IO>(P.bias, n) : 0.f;
        col_res[nt] = RES ? epi_res_ptr<BF16IO>(P, col_ok[nt], q, n) : nullptr;
    }

    const int ar = wave * 32 + (lane >> 3);           // wave-local staging rows ar + 8*i
    const int ac4 = (tid & 7) * 4;                    // staging column, in 4-byte units of the 128 B payload
    f32x4 ra0[4], ra1[4], ra2[4];   // ring slots as separate arrays: guaranteed register-resident
    const long my_tiles = (row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const long n_iter = my_tiles * total;

    auto fetch = [&](long it, f32x4 (&dra)[4]) __attribute__((always_inline)) {
        const long ti = it / total;
        const int c = (int)(it - ti * total);
        const long row0 = ((long)blockIdx.x + ti * gridDim.x) * BM;
        const int p = c / chunks;
        const int k0 = (c - p * chunks) * BK;
        const void* A = (p == 0) ? P.A0 : P.A1;
        const size_t abase = (p == 0) ? 0 : (size_t)(p - 1) * P.a_plane_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long r = row0 + ar + 8 * i;
            r = r < P.M ? r : P.M - 1;
            const size_t off = abase + (size_t)r * P.lda + k0 + (WIDE ? 2 * ac4 : ac4);
            if constexpr (WIDE) {
                dra[i] = lda16<STREAM>(static_cast<const uint16_t*>(A) + off);   // 8 raw bf16
            } else if constexpr (BF16IO) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 t = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(A) + off);
                f32x4 v;
                v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
                v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
                dra[i] = v;
            } else {
                dra[i] = lda16<STREAM>(static_cast<const float*>(A) + off);
            }
        }
    };

    __syncthreads();   // B panel complete (the only workgroup barrier)
    if (n_iter <= 0) return;
    fetch(0, ra0);
    fetch(1 < n_iter ? 1 : n_iter - 1, ra1);
    fetch(2 < n_iter ? 2 : n_iter - 1, ra2);
    const long n_pad = (n_iter + PF - 1) / PF * PF;

    auto stage = [&](auto U, const long it) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        f32x4 (&slot)[4] = *[&]() -> f32x4 (*)[4] {
            if constexpr (u == 0) return &ra0;
            else if constexpr (u == 1) return &ra1;
            else return &ra2;
        }();
        const long ti = it / total;
        const int c = (int)(it - ti * total);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(&As[(ar + 8 * i) * LDA + ac4]) = slot[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const long nx = it + PF;
            fetch(nx < n_iter ? nx : n_iter - 1, slot);
        }
        const float* arow = &As[(wave * 32 + l31) * LDA + (WIDE ? 4 : 8) * half];
        const unsigned short* brow = Bt + (size_t)l31 * KS + c * BK + 8 * half;
#pragma unroll
        for (int s2 = 0; s2 < BK / 16; ++s2) {
            bf16x8_t ah, am, al;
            if constexpr (WIDE) {
                ah = *reinterpret_cast<const bf16x8_t*>(arow + 8 * s2);   // 16 k-values = 32 B per step
                am = ah; al = ah;
            } else {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(arow + 16 * s2);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(arow + 16 * s2 + 4);
                const float f[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                float r1[8], r2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    r1[j] = f[j] - trunc_bf16(f[j]);
                    r2[j] = r1[j] - trunc_bf16(r1[j]);
                }
                ah = pack_bf16x8(f);
                am = ah; al = ah;
                if constexpr (NSPLIT == 3) { am = pack_bf16x8(r1); al = pack_bf16x8(r2); }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned short* bp = brow + (size_t)(32 * nt) * KS + 16 * s2;
                const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(bp);
                f32x16 a_ = acc[nt];
                if constexpr (NSPLIT == 3) {
                    const bf16x8_t bm = *reinterpret_cast<const bf16x8_t*>(bp + (size_t)BNT * KS);
                    const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(bp + (size_t)2 * BNT * KS);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a_, 0, 0, 0);
                }
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a_, 0, 0, 0);
                acc[nt] = a_;
            }
        }

        if (c == total - 1 && it < n_iter && P.dbg != 1) {
            const long row0 = ((long)blockIdx.x + ti * gridDim.x) * BM;
            const bool full_rows = row0 + BM <= P.M;
            if constexpr (EPI) {
                unsigned char* wrows = reinterpret_cast<unsigned char*>(&As[(wave * 32) * LDA]);   // 32 x 144 B, wave-private
#pragma unroll
                for (int h = 0; h < NT / 2; ++h) {
                    const int j0 = col0 + 64 * h;                   // 64 consecutive columns of one output plane
                    if (j0 >= n_total) break;
                    const int q = j0 / P.n_per_plane, n0 = j0 - q * P.n_per_plane;
                    unsigned short* Cg = static_cast<unsigned short*>((q == 0) ? P.C0 : P.C1) +
                                         ((q == 0) ? 0 : (size_t)(q - 1) * P.c_plane_stride) + n0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const f32x2 v = {epi_act(acc[2 * h][i] + col_bias[2 * h], P.relu), epi_act(acc[2 * h + 1][i] + col_bias[2 * h + 1], P.relu)};
                        const int rl = 4 * half + (i & 3) + 8 * (i >> 2);
                        *reinterpret_cast<uint32_t*>(wrows + rl * (LDA * 4) + 4 * l31) =
                            __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rl = (lane >> 3) + 8 * t, c16 = lane & 7;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(wrows + rl * (LDA * 4) + 16 * c16);
                        const long r = row0 + wave * 32 + rl;
                        if (full_rows || r < P.M) *reinterpret_cast<f32x4*>(Cg + (size_t)r * P.ldc + 8 * c16) = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                char* C = col_ptr[nt];
                const float bias = col_bias[nt];
                const long rbase = row0 + wave * 32 + 4 * half;
                float rv[16];
                if constexpr (RES) {
                    asm volatile("" ::: "memory");   // the residual loads of tile nt + 1 stay behind the stores of tile nt
                    epi_res_load<BF16IO>(rv, col_res[nt], rbase, P.ldr, P.M);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) rv[i] = 0.f;
                }
                if (full_rows) {
                    if (col_ok[nt]) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            st1<BF16IO>(C, (size_t)(rbase + (i & 3) + 8 * (i >> 2)) * P.ldc, epi_fin(acc[nt][i], bias, escale, rv[i], P.relu));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const long r = rbase + (i & 3) + 8 * (i >> 2);
                        if (col_ok[nt] && r < P.M) st1<BF16IO>(C, (size_t)r * P.ldc, epi_fin(acc[nt][i], bias, escale, rv[i], P.relu));
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            }
            }
        }
    };
    for (long base = 0; base < n_pad; base += PF) {
        stage(std::integral_constant<int, 0>{}, base);
        stage(std::integral_constant<int, 1>{}, base + 1);
        stage(std::integral_constant<int, 2>{}, base + 2);
    }
}


template <int NT, int NSPLIT, int IO, int NWV>
int launch_x3(const TsGemmParams& P, int col_tiles, size_t lds, hipStream_t stream) {
    constexpr int BM = 32 * NWV;
    const long row_tiles = (P.M + BM - 1) / BM;
    const void* kfn = (const void*)ts_gemm_x3_kernel<NT, NSPLIT, IO, NWV>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DSW_ERR_LAUNCH;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * NWV, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    long gx = 256L * per_cu / col_tiles;
    if (gx < 1) gx = 1;
    if (gx > row_tiles) gx = row_tiles;
    // block (x, y) lands on XCD (x + y * gx) % 8: with gx a multiple of 8 the column tiles of one row tile share an
    // XCD, so only the first of them fetches the A rows from HBM and the others hit that XCD's L2
    if (col_tiles > 1 && gx >= 8) gx &= ~7L;
    dim3 grid((unsigned)gx, (unsigned)col_tiles);
    DSW_LAUNCH((ts_gemm_x3_kernel<NT, NSPLIT, IO, NWV>), grid, dim3(64 * NWV), lds, stream, P);
    return dsw_check_launch();
}

}  // namespace

// Takes the launch (returns 1, *rc = status) when the aligned problem's (split) W panel fits LDS.
// fp32 storage -> 3-way split; bf16 storage -> plain bf16 MFMA.
int dsw_ts_gemm_x3_try_launch(const TsGemmParams& P, int nt, int col_tiles, int bf16, hipStream_t stream, int* rc) {
    static const char* x3env = dsw_diag_env("DSW_GEMM_X3");   // "0": exact fp32 MFMA path (diagnostics / A-B)
    if (x3env && x3env[0] == '0') return 0;
    const size_t ks = (size_t)P.n_planes_a * P.kd_per_plane + 8;
    const bool wide = bf16 && P.kd_per_plane % 64 == 0;
    // packed epilogue: 64-column groups are whole, contiguous, 16-byte aligned row segments of one output plane
    const bool res = P.R != nullptr || P.scale != nullptr;
    if (res && bf16) return 0;     // epilogue operands: fp32 instantiations only (bf16 takes the generic MFMA kernel)
    const bool epi = bf16 && nt % 2 == 0 && P.n_per_plane % 64 == 0 && P.ldc % 8 == 0 && P.c_plane_stride % 8 == 0 &&
                     dsw_aligned16(P.C0) && (P.n_planes_c == 1 || dsw_aligned16(P.C1));
    const int nsplit = bf16 ? 1 : 3;
    const size_t panel = (size_t)nsplit * (size_t)(32 * nt) * ks * 2;
    size_t lds = (size_t)BM * LDA * 4 + panel;
    if (lds > 160 * 1024) return 0;
    // one workgroup per CU only -> 8-wave workgroups (256-row tiles) if they still fit
    const bool big = 2 * lds > 160 * 1024 && (size_t)2 * BM * LDA * 4 + panel <= 160 * 1024;
    if (big) lds = (size_t)2 * BM * LDA * 4 + panel;
    const size_t a_bytes = (size_t)P.M * (size_t)P.n_planes_a * P.kd_per_plane * (bf16 ? 2 : 4);
    const bool strm = a_bytes >= ((size_t)128 << 20);
#define DSW_X3_L(NT_, NS_, IO_, NWV_)                                                      \
    (strm ? launch_x3<NT_, NS_, (IO_) | 8, NWV_>(P, col_tiles, lds, stream)               \
          : launch_x3<NT_, NS_, IO_, NWV_>(P, col_tiles, lds, stream))
#define DSW_X3_IO(NT_, NWV_)                                                               \
    (wide ? DSW_X3_L(NT_, 1, 3, NWV_) : bf16 ? DSW_X3_L(NT_, 1, 1, NWV_)                   \
          : res ? DSW_X3_L(NT_, 3, 16, NWV_) : DSW_X3_L(NT_, 3, 0, NWV_))
#define DSW_X3_EPI(NT_, NWV_)                                                              \
    (wide ? DSW_X3_L(NT_, 1, 7, NWV_) : DSW_X3_L(NT_, 1, 5, NWV_))
#define DSW_X3_CASE(NT_)                                                                   \
    case NT_:                                                                              \
        *rc = big ? DSW_X3_IO(NT_, 8) : DSW_X3_IO(NT_, 4);                                 \
        return 1;
#define DSW_X3_CASE_E(NT_)                                                                 \
    case NT_:                                                                              \
        *rc = epi ? (big ? DSW_X3_EPI(NT_, 8) : DSW_X3_EPI(NT_, 4))                        \
                  : (big ? DSW_X3_IO(NT_, 8) : DSW_X3_IO(NT_, 4));                         \
        return 1;
    switch (nt) {
        DSW_X3_CASE(1) DSW_X3_CASE_E(2) DSW_X3_CASE(3) DSW_X3_CASE_E(4)
    }
#undef DSW_X3_EPI
#undef DSW_X3_L
#undef DSW_X3_CASE_E
#undef DSW_X3_IO
#undef DSW_X3_CASE
    return 0;
}
