// Channel-mix contractions of the Chebyshev convolution on the matrix cores (MFMA), gfx950.
//
//   forward   Y[n, o]      = bias[o] + sum_{k,f} T_k[n, f] * W[f, k, o]          (layers.py:171-178)
//   dgrad     G_k[n, f]    = sum_o dY[n, o] * W[f, k, o]                         (autograd of :177)
//   wgrad     dW[f, k, o]  = sum_n T_k[n, f] * dY[n, o],  db[o] = sum_n dY[n, o]
//
// n runs over the N = B*V node rows (hundreds of thousands), the other two extents are channel
// counts (2..1536), i.e. every contraction is "tall and skinny".  The T_k are K separate
// [N, Fin] planes (T_0 is the caller's x itself) so the reference's [B*V, Fin*K] operand
// (layers.py:171-173: view/permute/contiguous = one full extra copy) is never materialised.
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TFLOP/s peak on MI355X).
// The summation order over the reduction index differs from a k-ascending loop (operands are
// consumed in 8-wide groups, lane-half h taking elements 4h..4h+3) - irrelevant at fp32 tolerance.
// bf16 storage is supported by widening to fp32 while staging through LDS (fp32 accumulate).
#include "dsw_gemm_common.h"

using namespace dsw_gemm;

// x3-split fp32 GEMM on the bf16 matrix pipe (dsw_gemm_x3.hip); returns 1 if it took the launch
int dsw_ts_gemm_x3_try_launch(const TsGemmParams& P, int nt, int col_tiles, int bf16, hipStream_t stream, int* rc);

// streaming-W variant for wide fp32 layers (dsw_gemm_x3s.hip); returns 1 if it took the launch
int dsw_ts_gemm_x3s_try_launch(const TsGemmParams& P, hipStream_t stream, int* rc);
// narrow layers (K * Fout <= 16) on the vector ALU (dsw_narrow.hip)
int dsw_narrow_fwd_try(const void* X, const void* W, const void* bias, void* Z0, void* Zrest, int64_t N, int64_t Fin,
                       int64_t Fout, int64_t K, int dtype, hipStream_t stream, int relu, int* rc);
int dsw_narrow_dgrad_try(const void* dY, const void* D, const void* W, void* dX, int64_t N, int64_t Fin, int64_t Fout,
                         int64_t K, int dtype, hipStream_t stream, int* rc);
int dsw_narrow_wgrad_try(const void* X, const void* dY, const void* D, void* dW, void* db, float* partial,
                         int64_t max_blocks, int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                         hipStream_t stream, int* rc, int accumulate);
int64_t dsw_wgrad_slabs(int64_t N, int64_t Fin, int64_t Fout, int64_t K);

// wgrad on the bf16 matrix pipe (dsw_wgrad_x3.hip); returns 1 if it took the launch
int dsw_wgrad_x3_try_launch(WgradParams& P, int bf16, int64_t max_slabs, int64_t* S_out, hipStream_t stream, int* rc);

namespace {

// C (M x n_total) = sum_p A[p] (M x kd) * B[p] (kd x n_total) (+ bias),  n_total = n_planes_c * n_per_plane.
// Workgroup tile: 128 rows x 32*NT columns; wave w owns rows [32w, 32w+32) and NT 32x32 MFMA tiles.
// RESIDENT: the whole B panel of the column tile lives in LDS for the lifetime of the workgroup,
// which then walks row tiles blockIdx.x, +gridDim.x, ... with a register prefetch that runs across
// tile boundaries.  Otherwise B is streamed chunk by chunk next to A.
// ALIGNED: A rows are 16-byte aligned and kd_per_plane % 32 == 0 (no column bounds checks).
// PF: depth of the register prefetch ring for A (chunks in flight per workgroup).  One chunk is
// 16 KB; HBM latency under load is ~2-3 us while a chunk's MFMAs take < 1 us, so a single
// chunk in flight leaves the matrix pipe idle most of the time (measured: 3 TB/s effective).
template <bool BF16, int NT, bool RESIDENT, bool ALIGNED, bool RES = false>
__global__ __launch_bounds__(256) void ts_gemm_kernel(const TsGemmParams P) {
    constexpr int BNT = 32 * NT;
    constexpr int PF = RESIDENT ? DSW_GEMM_PF : 2;   // streaming B doubles the ring's registers: keep it shallow
    constexpr int NRB = RESIDENT ? 1 : (BK * BNT) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;               // [BM][LDA]
    float* Bs = smem + BM * LDA;    // RESIDENT: [n_planes_a * chunks * BK][BNT]; else [BK][BNT]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int n_total = P.n_planes_c * P.n_per_plane;
    const int col0 = blockIdx.y * BNT;
    const int chunks = (P.kd_per_plane + BK - 1) / BK;
    const int total = P.n_planes_a * chunks;            // reduction chunks per row tile
    const long row_tiles = (P.M + BM - 1) / BM;

    // B element for (reduction chunk it_c, row kk of the chunk, column jj of the tile)
    auto load_b = [&](int it_c, int kk, int jj) -> float {
        const int p = it_c / chunks;
        const int kd = (it_c - p * chunks) * BK + kk;
        const int j = col0 + jj;
        if (kd >= P.kd_per_plane || j >= n_total) return 0.f;
        const int q = j / P.n_per_plane, n = j - q * P.n_per_plane;
        return ld1<BF16>(P.Bsrc, (size_t)((long)p * P.b_sp + (long)q * P.b_sq + (long)kd * P.b_skd + (long)n * P.b_sn));
    };

    if constexpr (RESIDENT) {
        const int nelem = total * BK * BNT;
        for (int e = tid; e < nelem; e += 256) {
            const int r = e / BNT, jj = e - r * BNT;
            Bs[e] = load_b(r / BK, r % BK, jj);
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // per-lane output column of each MFMA tile: plane, in-plane column, bias (loaded once, so the
    // epilogue issues no loads and therefore never drains the prefetch ring)
    bool col_ok[NT];
    char* col_ptr[NT];
    float col_bias[NT];
    const char* col_res[NT];
    const float escale = RES ? epi_scale<BF16>(P) : 1.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int j = col0 + 32 * nt + l31;
        col_ok[nt] = j < n_total;
        const int q = col_ok[nt] ? j / P.n_per_plane : 0;
        const int n = col_ok[nt] ? j - q * P.n_per_plane : 0;
        char* base = static_cast<char*>((q == 0) ? P.C0 : P.C1);
        const size_t cbase = (q == 0) ? 0 : (size_t)(q - 1) * P.c_plane_stride;
        col_ptr[nt] = base + (cbase + (size_t)n) * (BF16 ? 2 : 4);
        col_bias[nt] = (P.bias != nullptr && !(P.bias_plane0 && q != 0)) ? ld1<BF16>(P.bias, n) : 0.f;
        col_res[nt] = RES ? epi_res_ptr<BF16>(P, col_ok[nt], q, n) : nullptr;
    }

    // A staging.  RESIDENT: every wave stages exactly the 32 rows it consumes (rows wave*32 + (lane>>3)
    // + 8*i), so the LDS hand-off is wave-local and the main loop needs NO workgroup barrier - waves
    // drift apart and overlap each other's load / LDS / MFMA phases.  Streaming-B mode shares the B
    // chunk across waves and keeps the two barriers (rows ar + 32*i).
    const int ar = RESIDENT ? (wave * 32 + (lane >> 3)) : (tid >> 3);
    constexpr int AR_STEP = RESIDENT ? 8 : 32;
    const int ac4 = (tid & 7) * 4;
    float4 ra[PF][4];
    float rb[PF][NRB];

    const long my_tiles = RESIDENT ? (row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 1;
    const long n_iter = my_tiles * total;

    // Branch-free in the ALIGNED case (row index clamped instead of predicated): hipcc only keeps
    // counted s_waitcnt vmcnt(N) - i.e. leaves the younger ring slots in flight - in straight-line code.
    auto fetch = [&](long it, float4 (&dra)[4], float (&drb)[NRB]) __attribute__((always_inline)) {
        const long ti = it / total;
        const int c = (int)(it - ti * total);
        const long row0 = ((long)blockIdx.x + ti * gridDim.x) * BM;
        const int p = c / chunks;
        const int k0 = (c - p * chunks) * BK;
        const void* A = (p == 0) ? P.A0 : P.A1;
        const size_t abase = (p == 0) ? 0 : (size_t)(p - 1) * P.a_plane_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long r = row0 + ar + AR_STEP * i;
            const int kc = k0 + ac4;
            if constexpr (ALIGNED) {
                r = r < P.M ? r : P.M - 1;  // rows >= M are computed on a copy of the last row, never stored
                dra[i] = ld4<BF16>(A, abase + (size_t)r * P.lda + kc);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < P.M) {
                    const size_t off = abase + (size_t)r * P.lda + kc;
                    if (P.a_vec && kc + 3 < P.kd_per_plane) {
                        v = ld4<BF16>(A, off);
                    } else {
                        if (kc + 0 < P.kd_per_plane) v.x = ld1<BF16>(A, off + 0);
                        if (kc + 1 < P.kd_per_plane) v.y = ld1<BF16>(A, off + 1);
                        if (kc + 2 < P.kd_per_plane) v.z = ld1<BF16>(A, off + 2);
                        if (kc + 3 < P.kd_per_plane) v.w = ld1<BF16>(A, off + 3);
                    }
                }
                dra[i] = v;
            }
        }
        if constexpr (!RESIDENT) {
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int e = tid + 256 * i;
                drb[i] = load_b(c, e / BNT, e % BNT);
            }
        }
    };

    if constexpr (RESIDENT) __syncthreads();   // B panel written by all waves (the only workgroup barrier)
    if (n_iter <= 0) return;
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u < n_iter ? u : n_iter - 1, ra[u], rb[u]);

    const long n_pad = (n_iter + PF - 1) / PF * PF;   // padded iterations redo the last chunk; never stored
    // one ring stage; `U` is a compile-time slot index so that ra[u] / rb[u] stay in registers
    auto stage = [&](auto U, const long it) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const long ti = it / total;
        const int c = (int)(it - ti * total);
        if constexpr (RESIDENT) {
            // wave-local WAR: this wave's fragment reads of the previous chunk precede the overwrite
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();  // previous chunk fully consumed by every wave
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[(ar + AR_STEP * i) * LDA + ac4]) = ra[u][i];
        if constexpr (RESIDENT) {
            // wave-local RAW: the ds_writes above are visible to this wave's lanes before the reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
#pragma unroll
            for (int i = 0; i < NRB; ++i) Bs[tid + 256 * i] = rb[u][i];
            __syncthreads();
        }
        {   // refill this ring slot (clamped: the tail re-reads the last chunk instead of branching)
            const long nx = it + PF;
            fetch(nx < n_iter ? nx : n_iter - 1, ra[u], rb[u]);
        }

        // LDS -> registers for the whole chunk first (A: 4 x b128, B: 16*NT x b32), then the MFMAs run
        // back to back; just-in-time operand reads would expose the LDS latency 16 times per chunk.
        const float* arow = &As[(wave * 32 + l31) * LDA + 4 * half];
        const float* bchunk = (RESIDENT ? (Bs + (size_t)c * BK * BNT) : Bs) + (4 * half) * BNT + l31;
        float4 a4[BK / 8];
        float bv[BK / 8][4][NT];
#pragma unroll
        for (int cc = 0; cc < BK / 8; ++cc) a4[cc] = *reinterpret_cast<const float4*>(arow + 8 * cc);
#pragma unroll
        for (int cc = 0; cc < BK / 8; ++cc)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[cc][t][nt] = bchunk[(8 * cc + t) * BNT + 32 * nt];
        __builtin_amdgcn_sched_barrier(0);   // keep the operand reads clustered ahead of the MFMA burst
        if (P.dbg != 2)
#pragma unroll
        for (int cc = 0; cc < BK / 8; ++cc) {
            const float av[4] = {a4[cc].x, a4[cc].y, a4[cc].z, a4[cc].w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[cc][t][nt], acc[nt], 0, 0, 0);
        }

        if (c == total - 1 && it < n_iter && P.dbg != 1) {
            // epilogue: C/D layout of a 32x32 tile: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
            const long row0 = ((long)blockIdx.x + ti * gridDim.x) * BM;
            const bool full_rows = row0 + BM <= P.M;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool jok = col_ok[nt];
                char* C = col_ptr[nt];          // points at (row 0, this lane's column) of the right plane
                const float bias = col_bias[nt];
                const long rbase = row0 + wave * 32 + 4 * half;
                float rv[16];
                if constexpr (RES) {
                    asm volatile("" ::: "memory");   // the residual loads of tile nt + 1 stay behind the stores of tile nt
                    epi_res_load<BF16>(rv, col_res[nt], rbase, P.ldr, P.M);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) rv[i] = 0.f;
                }
                if (full_rows) {
                    if (jok) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            st1<BF16>(C, (size_t)(rbase + (i & 3) + 8 * (i >> 2)) * P.ldc, epi_fin(acc[nt][i], bias, escale, rv[i], P.relu));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const long r = rbase + (i & 3) + 8 * (i >> 2);
                        if (jok && r < P.M) st1<BF16>(C, (size_t)r * P.ldc, epi_fin(acc[nt][i], bias, escale, rv[i], P.relu));
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            }
        }
    };
    for (long base = 0; base < n_pad; base += PF) {
        stage(std::integral_constant<int, 0>{}, base);
        stage(std::integral_constant<int, 1>{}, base + 1);
        if constexpr (PF > 2) stage(std::integral_constant<int, 2>{}, base + 2);
        if constexpr (PF > 3) stage(std::integral_constant<int, 3>{}, base + 3);
        if constexpr (PF > 4) stage(std::integral_constant<int, 4>{}, base + 4);
        if constexpr (PF > 5) stage(std::integral_constant<int, 5>{}, base + 5);
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad: partial[s][(k*Fin + f)][o] = sum_{n in slab s} T_k[n, f] * dY[n, o]; row Kd = column sums
// ---------------------------------------------------------------------------------------------
// NW waves per workgroup = number of 32-row (k, f) tiles it covers (1..4): no idle waves when
// K * ceil(Fin/32) is not a multiple of 4 (north-star: 3 tiles -> 3-wave workgroups).
// ALIGNED: Fin % 32 == 0, Fout-tile full and 16-byte aligned rows (no column bounds checks).
template <bool BF16, int NW, bool ALIGNED>
__global__ __launch_bounds__(64 * NW) void cheb_wgrad_kernel(const WgradParams P) {
    constexpr int NT_ = 64 * NW;                       // threads
    constexpr int RD = (WR * BN / 4 + NT_ - 1) / NT_;  // float4 of the dY tile per thread
    constexpr int PF = DSW_WGRAD_PF;                   // prefetch ring depth (chunks in flight)
    __shared__ __attribute__((aligned(16))) float Ts[NW][WR * 32];
    __shared__ __attribute__((aligned(16))) float Ds[WR * BN];

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
    float colsum = 0.f;

    const int tr = lane >> 3, tc4 = (lane & 7) * 4;   // T tile (per wave): rows tr + 8*i (i<4)

    float4 rt[PF][4], rd[PF][RD];
    // ALIGNED: N % 32 == 0, so every chunk is complete and loads are unconditional (inactive waves
    // load tile 0's data and discard it) -> straight-line code, counted vmcnt, ring stays in flight.
    auto fetch = [&](long n0, float4 (&drt)[4], float4 (&drd)[RD]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = n0 + tr + 8 * i;
            const int f = f0 + tc4;
            if constexpr (ALIGNED) {
                drt[i] = ld4<BF16>(A, abase + (size_t)r * P.Fin + f);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (active && r < n_end) {
                    const size_t off = abase + (size_t)r * P.Fin + f;
                    if (P.t_vec && f + 3 < P.Fin) {
                        v = ld4<BF16>(A, off);
                    } else {
                        if (f + 0 < P.Fin) v.x = ld1<BF16>(A, off + 0);
                        if (f + 1 < P.Fin) v.y = ld1<BF16>(A, off + 1);
                        if (f + 2 < P.Fin) v.z = ld1<BF16>(A, off + 2);
                        if (f + 3 < P.Fin) v.w = ld1<BF16>(A, off + 3);
                    }
                }
                drt[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            int e = tid + NT_ * i;                       // float4 index inside the [WR][BN] tile
            if constexpr (ALIGNED) {
                if ((WR * BN / 4) % NT_ != 0) e = e < WR * BN / 4 ? e : WR * BN / 4 - 1;
                const int dr = e >> 4, dc4 = (e & 15) * 4;
                drd[i] = ld4<BF16>(P.dY, (size_t)(n0 + dr) * P.Fout + o0 + dc4);
            } else {
                const int dr = e >> 4, dc4 = (e & 15) * 4;
                const long r = n0 + dr;
                const int o = o0 + dc4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.dy_planes > 1) {
                    // narrow mix-first layers: column j of the tile = channel j % F0 of plane j / F0 (plane 0 = dY,
                    // plane z >= 1 = dY1 + (z - 1) * dy_plane_stride): all K planes of [N, F0] in ONE pass over X
                    if (e < WR * BN / 4 && r < n_end) {
                        const int F0 = P.Fout / P.dy_planes;
                        float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int j = o + t;
                            if (j < P.Fout) {
                                const int z = j / F0, oo = j - z * F0;
                                const size_t off = (size_t)r * F0 + oo;
                                t4[t] = z == 0 ? ld1<BF16>(P.dY, off) : ld1<BF16>(P.dY1, (size_t)(z - 1) * P.dy_plane_stride + off);
                            }
                        }
                        v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    }
                } else if (e < WR * BN / 4 && r < n_end) {
                    const size_t off = (size_t)r * P.Fout + o;
                    if (P.dy_vec && o + 3 < P.Fout) {
                        v = ld4<BF16>(P.dY, off);
                    } else {
                        if (o + 0 < P.Fout) v.x = ld1<BF16>(P.dY, off + 0);
                        if (o + 1 < P.Fout) v.y = ld1<BF16>(P.dY, off + 1);
                        if (o + 2 < P.Fout) v.z = ld1<BF16>(P.dY, off + 2);
                        if (o + 3 < P.Fout) v.w = ld1<BF16>(P.dY, off + 3);
                    }
                }
                drd[i] = v;
            }
        }
    };

    const long n_chunks = (n_end - n_begin + WR - 1) / WR;
    if (n_chunks > 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            fetch(n_begin + (long)(u < n_chunks ? u : n_chunks - 1) * WR, rt[u], rd[u]);
    }
    const long n_pad = (n_chunks + PF - 1) / PF * PF;   // padded chunks re-read the last one, unused
    auto stage = [&](auto U, const long ci) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&Ts[wave][(tr + 8 * i) * 32 + tc4]) = rt[u][i];
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            const int e = tid + NT_ * i;
            if (e < WR * BN / 4) *reinterpret_cast<float4*>(&Ds[e * 4]) = rd[u][i];
        }
        __syncthreads();
        {
            const long nx = ci + PF;
            fetch(n_begin + (nx < n_chunks ? nx : n_chunks - 1) * WR, rt[u], rd[u]);
        }
        if (ci < n_chunks) {
            if (active) {
#pragma unroll
                for (int s = 0; s < WR / 2; ++s) {
                    const int n = 2 * s + half;
                    const float a = Ts[wave][n * 32 + l31];       // A^T[f][n]
                    const float b0 = Ds[n * BN + l31];             // dY[n][o]
                    const float b1 = Ds[n * BN + 32 + l31];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                }
            }
            if (blockIdx.y == 0 && tid < BN) {
                float cs = 0.f;
#pragma unroll 8
                for (int r = 0; r < WR; ++r) cs += Ds[r * BN + tid];
                colsum += cs;
            }
        }
    };
    for (long cb = 0; cb < n_pad; cb += PF) {
        stage(std::integral_constant<int, 0>{}, cb);
        stage(std::integral_constant<int, 1>{}, cb + 1);
        if constexpr (PF > 2) stage(std::integral_constant<int, 2>{}, cb + 2);
        if constexpr (PF > 3) stage(std::integral_constant<int, 3>{}, cb + 3);
        if constexpr (PF > 4) stage(std::integral_constant<int, 4>{}, cb + 4);
    }

    float* out = P.partial + (size_t)blockIdx.x * (size_t)(Kd + 1) * P.Fout;
    if (active) {
        const int oA = o0 + l31, oB = o0 + 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = f0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (f < P.Fin) {
                const size_t off = (size_t)(k * P.Fin + f) * P.Fout;
                if (oA < P.Fout) out[off + oA] = acc0[i];
                if (oB < P.Fout) out[off + oB] = acc1[i];
            }
        }
    }
    if (blockIdx.y == 0 && tid < BN && o0 + tid < P.Fout) out[(size_t)Kd * P.Fout + o0 + tid] = colsum;
}

// dW[f, k, o] = sum_s partial[s][k*Fin + f][o];  db[o] = sum_s partial[s][Kd][o]
// 32 * NG threads = 32 outputs x NG slab groups; fixed summation order -> bit-reproducible.  NG = 8 for wide layers
// (hundreds of blocks), 32 for narrow ones: a 64 -> 2 layer has 130 outputs = 5 blocks, each walking ~770 slabs.
template <bool BF16, int NG = 8>
__global__ __launch_bounds__(32 * NG) void cheb_wgrad_reduce_kernel(const float* __restrict__ partial, int S,
                                                                    int Fin, int Fout, int K, void* dW,
                                                                    void* db, int K_out, int k_off, int db_cols = 1 << 30,
                                                                    int accumulate = 0) {
    __shared__ float red[NG][32];
    const int Kd = K * Fin;
    const long total = (long)(Kd + 1) * Fout;
    const int lane_o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + lane_o;
    const size_t slab = (size_t)(Kd + 1) * Fout;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
        int s = grp;
        for (; s + 3 * NG < S; s += 4 * NG) {     // four loads in flight per lane
            s0 += partial[(size_t)s * slab + idx];
            s1 += partial[(size_t)(s + NG) * slab + idx];
            s2 += partial[(size_t)(s + 2 * NG) * slab + idx];
            s3 += partial[(size_t)(s + 3 * NG) * slab + idx];
        }
        for (; s < S; s += NG) s0 += partial[(size_t)s * slab + idx];
    }
    red[grp][lane_o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && idx < total) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) v += red[g][lane_o];
        const int kd = (int)(idx / Fout), o = (int)(idx - (long)kd * Fout);
        // accumulate: dW / db are gradient buffers that already hold the contributions of other uses of the parameter (an
        // autoregressive step applies every layer 7 times): add instead of leaving 38 x 7 tiny `add` launches to autograd
        if (kd == Kd) {
            if (db != nullptr && o < db_cols) st1<BF16>(db, o, accumulate ? v + ld1<BF16>(db, o) : v);   // (planes side by side: only plane 0 is dY)
        } else {
            const int k = kd / Fin, f = kd - k * Fin;
            const size_t at = ((size_t)f * K_out + k_off + k) * Fout + o;        // dW is [Fin, K_out, Fout]
            st1<BF16>(dW, at, accumulate ? v + ld1<BF16>(dW, at) : v);
        }
    }
}

}  // namespace

// ------------------------------- host-side launchers (internal) ------------------------------
int dsw_wgrad_reduce_launch(const float* partial, int64_t S, int64_t Fin, int64_t Fout, int64_t K, void* dW, void* db,
                            int64_t K_out, int64_t k_off, int db_cols, int dtype, hipStream_t stream, int accumulate);
template <bool BF16, int NT>
static int launch_ts_gemm_nt(const TsGemmParams& P, int col_tiles, hipStream_t stream) {
    constexpr int BNT = 32 * NT;
    const int chunks = (P.kd_per_plane + BK - 1) / BK;
    const long row_tiles = (P.M + BM - 1) / BM;
    const size_t a_bytes = (size_t)BM * LDA * 4;
    const size_t b_res = (size_t)P.n_planes_a * chunks * BK * BNT * 4;
    const bool resident = b_res <= 44 * 1024;   // A tile (18 KiB) + B panel stay under the 64 KiB default LDS limit
    const bool aligned = P.a_vec && (P.kd_per_plane % BK == 0);
    {
        int rc = DSW_OK;
        if (aligned && dsw_ts_gemm_x3_try_launch(P, NT, col_tiles, BF16 ? 1 : 0, stream, &rc)) return rc;
    }
    const bool res = P.R != nullptr || P.scale != nullptr;
#define DSW_TSG(RESIDENT_, ALIGNED_)                                                                                  \
    (res ? (const void*)ts_gemm_kernel<BF16, NT, RESIDENT_, ALIGNED_, true> : (const void*)ts_gemm_kernel<BF16, NT, RESIDENT_, ALIGNED_, false>)
#define DSW_TSG_LAUNCH(RESIDENT_, ALIGNED_, GRID_, LDS_)                                                              \
    do {                                                                                                              \
        if (res) DSW_LAUNCH((ts_gemm_kernel<BF16, NT, RESIDENT_, ALIGNED_, true>), GRID_, dim3(256), LDS_, stream, P);   \
        else DSW_LAUNCH((ts_gemm_kernel<BF16, NT, RESIDENT_, ALIGNED_, false>), GRID_, dim3(256), LDS_, stream, P);      \
    } while (0)
    if (resident) {
        const size_t lds = a_bytes + b_res;
        int per_cu = 0;
        const void* kfn = aligned ? DSW_TSG(true, true) : DSW_TSG(true, false);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, lds) != hipSuccess || per_cu < 1)
            per_cu = 1;   // grid = resident workgroups only: the persistent loop has no tail wave
        long gx = 256L * per_cu / col_tiles;
        if (gx < 1) gx = 1;
        if (gx > row_tiles) gx = row_tiles;
        if (col_tiles > 1 && gx >= 8) gx &= ~7L;   // column tiles of one row tile on one XCD: A re-reads hit its L2
        dim3 grid((unsigned)gx, (unsigned)col_tiles);
        if (aligned) DSW_TSG_LAUNCH(true, true, grid, lds);
        else DSW_TSG_LAUNCH(true, false, grid, lds);
    } else {
        const size_t lds = a_bytes + (size_t)BK * BNT * 4;
        dim3 grid((unsigned)row_tiles, (unsigned)col_tiles);
        if (aligned) DSW_TSG_LAUNCH(false, true, grid, lds);
        else DSW_TSG_LAUNCH(false, false, grid, lds);
    }
#undef DSW_TSG
#undef DSW_TSG_LAUNCH
    return dsw_check_launch();
}

constexpr int DSW_NEED_FOLD = 1;   // internal: launch_ts_gemm wants TsGemmParams::fold_q resolved by the caller (fold_then_launch)

template <bool BF16>
static int launch_ts_gemm(const TsGemmParams& P, hipStream_t stream) {
    const int n_total = P.n_planes_c * P.n_per_plane;
    // bf16-pipe path with narrower column tiles when the (split) W panel of the natural tile does not fit LDS:
    // re-reading A once per extra column tile is cheaper than running on the 16x slower fp32 MFMA
    if (P.a_vec && (P.kd_per_plane % BK == 0) && n_total > 32) {
        const int nat = n_total <= 128 ? (n_total + 31) / 32 : 4;
        if constexpr (!BF16) {
            // wide fp32 layers: stream W chunk by chunk (dsw_gemm_x3s.hip) instead of narrowing the column tile
            const size_t ks = (size_t)P.n_planes_a * P.kd_per_plane + 8;
            const size_t lds_nat = (size_t)BM * LDA * 4 + (size_t)3 * (size_t)(32 * nat) * ks * 2;
            int rc = DSW_OK;
            // ... and, with caller scratch (pre-split image by LDS-DMA), also the shapes whose panel would fit when the output
            // has two or more 128-column tiles: 24 576 x 128 -> 256 32.6 -> 28.3 us, x 128 -> 512 43.5 -> 40.4, 98 304 x 64 -> 256
            // 43.8 -> 40.1 (narrower outputs: the resident panel wins or ties)
            const bool stream_w = lds_nat > 160 * 1024 || (P.pre_ws != nullptr && n_total >= 256);
            if (stream_w && dsw_ts_gemm_x3s_try_launch(P, stream, &rc)) return rc;
        }
        if (P.fold_q >= 0) return DSW_NEED_FOLD;   // only the streaming kernel folds while it splits W: the caller folds and retries
        for (int nt = nat - 1; nt >= 1; --nt) {
            int rc = DSW_OK;
            // only when the natural width fails: probe it first through the normal path below
            const int chunks = P.kd_per_plane / BK;
            const size_t ks = (size_t)P.n_planes_a * chunks * BK + 8;
            const size_t lds_nat = (size_t)BM * LDA * 4 + (size_t)(BF16 ? 1 : 3) * (size_t)(32 * nat) * ks * 2;
            if (lds_nat <= 160 * 1024) break;
            const int tiles = (n_total + 32 * nt - 1) / (32 * nt);
            if (dsw_ts_gemm_x3_try_launch(P, nt, tiles, BF16 ? 1 : 0, stream, &rc)) return rc;
        }
    }
    if (P.fold_q >= 0) return DSW_NEED_FOLD;
    if (n_total <= 32) return launch_ts_gemm_nt<BF16, 1>(P, 1, stream);
    if (n_total <= 64) return launch_ts_gemm_nt<BF16, 2>(P, 1, stream);
    if (n_total <= 96) return launch_ts_gemm_nt<BF16, 3>(P, 1, stream);
    if (n_total <= 128) return launch_ts_gemm_nt<BF16, 4>(P, 1, stream);
    // wide outputs: 128-column tiles, or 96 when that wastes fewer padded columns
    const int t128 = (n_total + 127) / 128, t96 = (n_total + 95) / 96;
    if (t96 * 96 < t128 * 128) return launch_ts_gemm_nt<BF16, 3>(P, t96, stream);
    return launch_ts_gemm_nt<BF16, 4>(P, t128, stream);
}

// optional epilogue operands of the channel-mix launchers (see TsGemmParams): C = act(scale * (acc + bias) + R) on output
// plane 0 (R) / every plane (scale); ldc > 0 overrides the row stride of the output (a channel slice of a wider tensor)
struct DswEpiExtra {
    const void* scale;
    const void* R;
    int64_t ldr;
    int64_t ldc;
    void* ws;            // scratch for an image of W (pre-split terms, or a folded copy), see TsGemmParams::pre_ws
    int64_t ws_bytes;
    int fold;            // 1: produce plane K-3 with W[:, K-3, :] - W[:, K-1, :] (launchers whose output planes are the orders k)
};
static inline bool epi_extra_set(const DswEpiExtra* e) { return e && (e->scale || e->R || e->ldc > 0); }
static inline void epi_extra_apply(TsGemmParams& P, const DswEpiExtra* e) {
    P.fold_q = -1;
    if (!e) return;
    P.scale = e->scale; P.R = e->R; P.ldr = (int)e->ldr;
    if (e->ldc > 0) P.ldc = (int)e->ldc;
    P.pre_ws = e->ws; P.pre_bytes = (long)e->ws_bytes;
}
int dsw_fold_w_launch(const void* W, void* Wf, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t s);

// launch_ts_gemm for the launchers whose output planes are the Chebyshev orders (dgrad planes G_k, mix-first planes Z_k):
// with `fold` the streaming kernel folds while it splits W (pre_ws); every other kernel gets a folded COPY of W in pre_ws.
static int fold_then_launch(TsGemmParams& P, const DswEpiExtra* e, const void* W, int64_t Fin, int64_t Fout, int64_t K,
                            int dtype, hipStream_t stream) {
    if (e && e->fold && K >= 3) {
        if (!P.pre_ws) return DSW_ERR_WORKSPACE;
        P.fold_q = (int)K - 3;
    }
    int rc = dtype == DSW_F32 ? launch_ts_gemm<false>(P, stream) : launch_ts_gemm<true>(P, stream);
    if (rc != DSW_NEED_FOLD) return rc;
    if (P.pre_bytes < Fin * K * Fout * (dtype == DSW_BF16 ? 2 : 4)) return DSW_ERR_WORKSPACE;
    rc = dsw_fold_w_launch(W, P.pre_ws, Fin, Fout, K, dtype, stream);
    if (rc != DSW_OK) return rc;
    P.Bsrc = P.pre_ws; P.pre_ws = nullptr; P.pre_bytes = 0; P.fold_q = -1;
    return dtype == DSW_F32 ? launch_ts_gemm<false>(P, stream) : launch_ts_gemm<true>(P, stream);
}

int dsw_mix_fwd_launch(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                       int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int relu,
                       const DswEpiExtra* extra) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (K == 1 && !epi_extra_set(extra)) {
        int rcn = DSW_OK;
        if (dsw_narrow_fwd_try(X, W, bias, Y, nullptr, N, Fin, Fout, 1, dtype, stream, relu, &rcn)) return rcn;
    }
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P{};
    P.A0 = X; P.A1 = T; P.a_plane_stride = (size_t)N * Fin; P.lda = (int)Fin;
    P.n_planes_a = (int)K; P.kd_per_plane = (int)Fin;
    P.Bsrc = W; P.b_sp = Fout; P.b_sq = 0; P.b_skd = K * Fout; P.b_sn = 1;
    P.C0 = Y; P.C1 = Y; P.c_plane_stride = 0; P.ldc = (int)Fout; P.n_planes_c = 1; P.n_per_plane = (int)Fout;
    P.bias = bias; P.M = N; P.relu = relu;
    epi_extra_apply(P, extra);
#ifdef DSW_ABLATION   // build with -DDSW_ABLATION to enable the DSW_DBG ablation knobs (they produce wrong results by design)
    { static const char* d = dsw_diag_env("DSW_DBG"); P.dbg = d ? atoi(d) : 0; }
#endif
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0) && (K == 1 || ((uintptr_t)T & am) == 0);
    if (dtype == DSW_F32) return launch_ts_gemm<false>(P, stream);
    if (dtype == DSW_BF16) return launch_ts_gemm<true>(P, stream);
    return DSW_ERR_BAD_DTYPE;
}

// G_0 -> dX buffer, G_1.. -> Gws planes
int dsw_mix_dgrad_launch(const void* dY, const void* W, void* G0, void* Grest, int64_t N, int64_t Fin,
                         int64_t Fout, int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (K == 1 && !epi_extra_set(extra)) {
        int rcn = DSW_OK;
        if (dsw_narrow_dgrad_try(dY, nullptr, W, G0, N, Fin, Fout, 1, dtype, stream, &rcn)) return rcn;
    }
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P{};
    P.A0 = dY; P.A1 = dY; P.a_plane_stride = 0; P.lda = (int)Fout; P.n_planes_a = 1; P.kd_per_plane = (int)Fout;
    P.Bsrc = W; P.b_sp = 0; P.b_sq = Fout; P.b_skd = 1; P.b_sn = K * Fout;
    P.C0 = G0; P.C1 = Grest; P.c_plane_stride = (size_t)N * Fin; P.ldc = (int)Fin; P.n_planes_c = (int)K;
    P.n_per_plane = (int)Fin;
    P.bias = nullptr; P.M = N;
    epi_extra_apply(P, extra);
#ifdef DSW_ABLATION   // build with -DDSW_ABLATION to enable the DSW_DBG ablation knobs (they produce wrong results by design)
    { static const char* d = dsw_diag_env("DSW_DBG"); P.dbg = d ? atoi(d) : 0; }
#endif
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    return fold_then_launch(P, extra, W, Fin, Fout, K, dtype, stream);
}


// ---- mix-first evaluation order (Fout <= Fin / 2):  Y = sum_k T_k(L) (X W_k)  - the recurrence runs on Fout channels
// Z_k = X W_k (+ bias on k = 0):  Z_0 -> Z0 buffer (the caller's Y), Z_1.. -> Zrest planes of [N, Fout]
int dsw_zmix_launch(const void* X, const void* W, const void* bias, void* Z0, void* Zrest, int64_t N, int64_t Fin,
                    int64_t Fout, int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (!epi_extra_set(extra) && !(extra && extra->fold && K >= 3)) {   // (the vector-ALU kernel does not fold)
        int rcn = DSW_OK;
        if (dsw_narrow_fwd_try(X, W, bias, Z0, Zrest, N, Fin, Fout, K, dtype, stream, 0, &rcn)) return rcn;
    }
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P{};
    P.A0 = X; P.A1 = X; P.a_plane_stride = 0; P.lda = (int)Fin; P.n_planes_a = 1; P.kd_per_plane = (int)Fin;
    P.Bsrc = W; P.b_sp = 0; P.b_sq = Fout; P.b_skd = K * Fout; P.b_sn = 1;      // element (q = k, kd = f, n = o) = W[f, k, o]
    P.C0 = Z0; P.C1 = Zrest; P.c_plane_stride = (size_t)N * Fout; P.ldc = (int)Fout; P.n_planes_c = (int)K;
    P.n_per_plane = (int)Fout;
    P.bias = bias; P.bias_plane0 = 1; P.M = N;
    epi_extra_apply(P, extra);
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    return fold_then_launch(P, extra, W, Fin, Fout, K, dtype, stream);
}

// dX = sum_k D_k W_k^T with D_0 = dY and D_1.. = planes of [N, Fout] (the Chebyshev basis of dY under L^T)
int dsw_zdgrad_launch(const void* dY, const void* D, const void* W, void* dX, int64_t N, int64_t Fin, int64_t Fout,
                      int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra) {
    if (N == 0) return DSW_OK;
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (!epi_extra_set(extra)) {
        int rcn = DSW_OK;
        if (dsw_narrow_dgrad_try(dY, D, W, dX, N, Fin, Fout, K, dtype, stream, &rcn)) return rcn;
    }
    const int es = dtype == DSW_BF16 ? 2 : 4;
    TsGemmParams P{};
    P.A0 = dY; P.A1 = D; P.a_plane_stride = (size_t)N * Fout; P.lda = (int)Fout; P.n_planes_a = (int)K;
    P.kd_per_plane = (int)Fout;
    P.Bsrc = W; P.b_sp = Fout; P.b_sq = 0; P.b_skd = 1; P.b_sn = K * Fout;      // element (p = k, kd = o, n = f) = W[f, k, o]
    P.C0 = dX; P.C1 = dX; P.c_plane_stride = 0; P.ldc = (int)Fin; P.n_planes_c = 1; P.n_per_plane = (int)Fin;
    P.bias = nullptr; P.M = N;
    epi_extra_apply(P, extra);
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    P.a_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0) && (K == 1 || ((uintptr_t)D & am) == 0);
    if (dtype == DSW_F32) return launch_ts_gemm<false>(P, stream);
    if (dtype == DSW_BF16) return launch_ts_gemm<true>(P, stream);
    return DSW_ERR_BAD_DTYPE;
}

// Upper bound of row slabs wgrad may use (sizes the partial workspace; the launch picks the actual
// count from the kernel's occupancy so that every slab's workgroup is resident: no tail wave).
// Partials are capped at 64 MiB: wide layers get their parallelism from the (k,f) x o tiling instead.
static int64_t wgrad_max_slabs(int64_t Fin, int64_t Fout, int64_t K) {
    const int64_t slab_bytes = (K * Fin + 1) * Fout * 4;
    int64_t m = (64LL << 20) / slab_bytes;
    if (m > 2048) m = 2048;
    if (m < 1) m = 1;
    return m;
}

static int64_t wgrad_rows_per_slab(int64_t N, int64_t target_slabs, int64_t max_slabs) {
    if (target_slabs > max_slabs) target_slabs = max_slabs;
    if (target_slabs < 1) target_slabs = 1;
    int64_t rps = (N + target_slabs - 1) / target_slabs;
    rps = ((rps + WR - 1) / WR) * WR;
    if (rps < 4 * WR) rps = 4 * WR;
    return rps;
}

int64_t dsw_wgrad_slabs(int64_t N, int64_t Fin, int64_t Fout, int64_t K) {
    const int64_t m = wgrad_max_slabs(Fin, Fout, K);
    const int64_t rps = wgrad_rows_per_slab(N, m, m);
    return N > 0 ? (N + rps - 1) / rps : 0;
}

// K_out / k_off: the launch fills planes k_off .. k_off + K - 1 of a dW laid out [Fin, K_out, Fout] (the plain
// wgrad has K_out = K, k_off = 0; the mix-first backward issues one K = 1 launch per Chebyshev order)
int dsw_wgrad_launch_ex(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                        int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                        int64_t K_out, int64_t k_off, int accumulate);

int dsw_wgrad_launch(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                     int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int accumulate) {
    return dsw_wgrad_launch_ex(X, T, dY, dW, db, partial, N, Fin, Fout, K, dtype, stream, K, 0, accumulate);
}

// dy_planes > 1 (narrow mix-first layers, K * F0 <= 64): the dY side is `dy_planes` planes of [N, F0] (plane 0 = dY, the
// others at dY1 + (z - 1) * N * F0) laid side by side as ONE virtual [N, dy_planes * F0] operand - Fout is that total,
// K must be 1 - so that dW[f, k, o] = sum_n X[n, f] D_k[n, o] comes out of one pass over X, already in [Fin, K, F0] order.
static int dsw_wgrad_launch_impl(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                                 int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                                 int64_t K_out, int64_t k_off, const void* dY1, int dy_planes, int accumulate);

int dsw_wgrad_launch_ex(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                        int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                        int64_t K_out, int64_t k_off, int accumulate) {
    return dsw_wgrad_launch_impl(X, T, dY, dW, db, partial, N, Fin, Fout, K, dtype, stream, K_out, k_off, nullptr, 1, accumulate);
}

static int dsw_wgrad_launch_impl(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                                 int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                                 int64_t K_out, int64_t k_off, const void* dY1, int dy_planes, int accumulate) {
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (N > 0 && K == 1 && K_out == 1 && k_off == 0 && dy_planes <= 1 && Fout <= 16) {   // narrow dense map (residual linear)
        int rcn = DSW_OK;
        if (dsw_narrow_wgrad_try(X, dY, nullptr, dW, db, partial, dsw_wgrad_slabs(N, Fin, Fout, 1), N, Fin, Fout, 1, dtype,
                                 stream, &rcn, accumulate))
            return rcn;
    }
    const int es = dtype == DSW_BF16 ? 2 : 4;
    int64_t rps = 0;
    int64_t S = N > 0 ? 1 : 0;
    if (S > 0) {
        WgradParams P{};
        P.X = X; P.T = T; P.plane_stride = (size_t)N * Fin; P.dY = dY; P.partial = partial;
        P.N = N; P.Fin = (int)Fin; P.Fout = (int)Fout; P.K = (int)K; P.rows_per_slab = rps;
        P.tiles_per_plane = (int)((Fin + 31) / 32);
        const uintptr_t am = (uintptr_t)(4 * es) - 1;
        P.t_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0) && (K == 1 || ((uintptr_t)T & am) == 0);
        P.dy_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0);
        if (dy_planes > 1) {
            P.dY1 = dY1; P.dy_planes = dy_planes; P.dy_plane_stride = (size_t)N * (size_t)(Fout / dy_planes);
            P.dy_vec = 0;
        }
        const int ntiles = (int)K * P.tiles_per_plane;
        // waves per workgroup: spread the (k, f) tiles evenly over the fewest groups of <= 4
        const int groups = (ntiles + 3) / 4;
        const int nw = (ntiles + groups - 1) / groups;
        const bool aligned = dy_planes <= 1 && P.t_vec && P.dy_vec && (Fin % 32 == 0) && (Fout % (dtype == DSW_F32 ? 32 : BN) == 0) && (N % WR == 0);
        const int64_t otiles = (Fout + BN - 1) / BN;
        {   // bf16 matrix pipe (3-way split for fp32 storage) when the problem is aligned
            int rc3 = DSW_OK;
            int64_t S3 = 0;
            if (aligned && dsw_wgrad_x3_try_launch(P, dtype == DSW_BF16 ? 1 : 0, wgrad_max_slabs(Fin, Fout, K), &S3,
                                                   stream, &rc3)) {
                if (rc3 != DSW_OK) return rc3;
                S = S3;
                goto reduce;
            }
        }
        // slabs = resident workgroups (occupancy query), so the launch is a single full wave of work
#define DSW_WGRAD_LAUNCH(BF, NW_)                                                                          \
    do {                                                                                                   \
        const void* kfn = aligned ? (const void*)cheb_wgrad_kernel<BF, NW_, true>                          \
                                  : (const void*)cheb_wgrad_kernel<BF, NW_, false>;                        \
        int occ = 0;                                                                                       \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 64 * NW_, 0) != hipSuccess || occ < 1) \
            occ = 1;                                                                                       \
        int64_t want = 256L * occ / (groups * otiles);                                                     \
        if (want < 32) want = 32;                                                                          \
        rps = wgrad_rows_per_slab(N, want, wgrad_max_slabs(Fin, Fout, K));                                                                \
        S = (N + rps - 1) / rps;                                                                           \
        P.rows_per_slab = rps;                                                                             \
        dim3 grid((unsigned)S, (unsigned)groups, (unsigned)otiles);                                        \
        if (aligned) DSW_LAUNCH((cheb_wgrad_kernel<BF, NW_, true>), grid, dim3(64 * NW_), 0, stream, P); \
        else DSW_LAUNCH((cheb_wgrad_kernel<BF, NW_, false>), grid, dim3(64 * NW_), 0, stream, P);  \
    } while (0)
        if (dtype == DSW_F32) {
            if (nw == 1) DSW_WGRAD_LAUNCH(false, 1);
            else if (nw == 2) DSW_WGRAD_LAUNCH(false, 2);
            else if (nw == 3) DSW_WGRAD_LAUNCH(false, 3);
            else DSW_WGRAD_LAUNCH(false, 4);
        } else {
            if (nw == 1) DSW_WGRAD_LAUNCH(true, 1);
            else if (nw == 2) DSW_WGRAD_LAUNCH(true, 2);
            else if (nw == 3) DSW_WGRAD_LAUNCH(true, 3);
            else DSW_WGRAD_LAUNCH(true, 4);
        }
#undef DSW_WGRAD_LAUNCH
        int rc = dsw_check_launch();
        if (rc != DSW_OK) return rc;
    }
reduce:
    const int db_cols = dy_planes > 1 ? (int)(Fout / dy_planes) : (int)Fout;
    return dsw_wgrad_reduce_launch(partial, S, Fin, Fout, K, dW, db, K_out, k_off, db_cols, dtype, stream, accumulate);
}

// partial [S][K * Fin + 1][Fout] -> dW / db (used by dsw_narrow.hip as well)
// accumulate (dsw_cheb_bwd_res, accumulate_dw; an explicit argument of every weight-gradient launcher): the reduce stage
// adds to dW / db instead of overwriting them.
int dsw_wgrad_reduce_launch(const float* partial, int64_t S, int64_t Fin, int64_t Fout, int64_t K, void* dW, void* db,
                            int64_t K_out, int64_t k_off, int db_cols, int dtype, hipStream_t stream, int accumulate) {
    const long total = (long)(K * Fin + 1) * Fout;
    const int acc = accumulate ? 1 : 0;
    dim3 rgrid((unsigned)((total + 31) / 32));
    if (total <= 16384 && S > 64) {   // few outputs (<= 512 blocks), many slabs: 32 slab groups per block
        if (dtype == DSW_F32)
            DSW_LAUNCH((cheb_wgrad_reduce_kernel<false, 32>), rgrid, dim3(1024), 0, stream, partial, (int)S,
                               (int)Fin, (int)Fout, (int)K, dW, db, (int)K_out, (int)k_off, db_cols, acc);
        else
            DSW_LAUNCH((cheb_wgrad_reduce_kernel<true, 32>), rgrid, dim3(1024), 0, stream, partial, (int)S,
                               (int)Fin, (int)Fout, (int)K, dW, db, (int)K_out, (int)k_off, db_cols, acc);
        return dsw_check_launch();
    }
    if (dtype == DSW_F32)
        DSW_LAUNCH(cheb_wgrad_reduce_kernel<false>, rgrid, dim3(256), 0, stream, partial, (int)S,
                           (int)Fin, (int)Fout, (int)K, dW, db, (int)K_out, (int)k_off, db_cols, acc);
    else
        DSW_LAUNCH(cheb_wgrad_reduce_kernel<true>, rgrid, dim3(256), 0, stream, partial, (int)S,
                           (int)Fin, (int)Fout, (int)K, dW, db, (int)K_out, (int)k_off, db_cols, acc);
    return dsw_check_launch();
}

// mix-first backward: dW[f, k, o] = sum_n X[n, f] * D_k[n, o] (D_0 = dY, D_1.. = planes of [N, Fout]), db = column sums
// of dY.  One launch over (slab, f-tiles, (k, o-tile)) when the problem is aligned (bf16-pipe kernels); otherwise one
// plain K = 1 wgrad per Chebyshev order.
int dsw_wgrad_launch_ex(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                        int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                        int64_t K_out, int64_t k_off, int accumulate);
static int64_t wgrad_max_slabs(int64_t Fin, int64_t Fout, int64_t K);

int dsw_wgrad_mixfirst_launch(const void* X, const void* dY, const void* D, void* dW, void* db, float* partial,
                              int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int accumulate) {
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int es = dtype == DSW_BF16 ? 2 : 4;
    const uintptr_t am = (uintptr_t)(4 * es) - 1;
    if (N > 0 && K > 1) {
        WgradParams P{};
        P.X = X; P.T = nullptr; P.plane_stride = 0; P.dY = dY; P.partial = partial;
        P.N = N; P.Fin = (int)Fin; P.Fout = (int)Fout; P.K = 1;
        P.tiles_per_plane = (int)((Fin + 31) / 32);
        P.dY1 = D; P.dy_plane_stride = (size_t)N * Fout; P.dy_planes = (int)K;
        P.t_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0);
        P.dy_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0) && (((uintptr_t)D & am) == 0);
        const bool aligned = P.t_vec && P.dy_vec && (Fin % 32 == 0) && (Fout % (dtype == DSW_F32 ? 32 : BN) == 0) && (N % WR == 0);
        int rc3 = DSW_OK;
        int64_t S3 = 0;
        if (aligned && dsw_wgrad_x3_try_launch(P, dtype == DSW_BF16 ? 1 : 0, wgrad_max_slabs(Fin, Fout, K), &S3, stream,
                                               &rc3)) {
            if (rc3 != DSW_OK) return rc3;
            return dsw_wgrad_reduce_launch(partial, S3, Fin, Fout, K, dW, db, K, 0, 1 << 30, dtype, stream, accumulate);
        }
    }
    if (N > 0 && K > 1 && K * Fout <= 16) {  // a handful of output columns: vector-ALU kernel (dsw_narrow.hip)
        int rcn = DSW_OK;
        if (dsw_narrow_wgrad_try(X, dY, D, dW, db, partial, dsw_wgrad_slabs(N, Fin, K * Fout, 1), N, Fin, Fout, K, dtype,
                                 stream, &rcn, accumulate))
            return rcn;
    }
    if (N > 0 && K > 1 && K * Fout <= BN)   // narrow output (e.g. the model's last layer, 64 -> 2): all K planes in ONE pass over X
        return dsw_wgrad_launch_impl(X, nullptr, dY, dW, db, partial, N, Fin, K * Fout, 1, dtype, stream, 1, 0, D, (int)K, accumulate);
    int rc = DSW_OK;
    const size_t dplane = (size_t)N * Fout * es;
    for (int64_t k = 0; k < K && rc == DSW_OK; ++k)
        rc = dsw_wgrad_launch_ex(X, nullptr, k == 0 ? dY : static_cast<const void*>(static_cast<const char*>(D) + (k - 1) * dplane),
                                 dW, k == 0 ? db : nullptr, partial, N, Fin, Fout, 1, dtype, stream, K, k, accumulate);
    return rc;
}

// Backward GEMM work of the plain (basis-first) layer in one launch: dW / db partials AND the dgrad planes G_k
// (dsw_wgrad_x3.hip, FUSE variant) followed by the partial reduce.  Returns 1 when it took the work (*rc = status).
int dsw_wgrad_dgrad_fused_try_launch(WgradParams& P, int bf16, int64_t max_slabs, int64_t* S_out, hipStream_t stream,
                                     int* rc);

int dsw_bwd_gemm_fused_try(const void* X, const void* T, const void* W, const void* dY, void* dW, void* db, void* G0,
                           void* Grest, float* partial, int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                           hipStream_t stream, int* rc, int accumulate, int fold) {
    if ((dtype != DSW_F32 && dtype != DSW_BF16) || N <= 0 || !dW || !G0) return 0;
    WgradParams P{};
    P.X = X; P.T = T; P.plane_stride = (size_t)N * Fin; P.dY = dY; P.partial = partial;
    P.N = N; P.Fin = (int)Fin; P.Fout = (int)Fout; P.K = (int)K;
    P.tiles_per_plane = (int)((Fin + 31) / 32);
    P.W = W; P.G0 = G0; P.Grest = Grest; P.fold = (fold && K >= 3) ? 1 : 0;
    const uintptr_t am = 15;
    P.t_vec = (Fin % 4 == 0) && (((uintptr_t)X & am) == 0) && (K == 1 || ((uintptr_t)T & am) == 0);
    P.dy_vec = (Fout % 4 == 0) && (((uintptr_t)dY & am) == 0);
    const bool aligned = P.t_vec && P.dy_vec && (Fin % 32 == 0) && (Fout % (dtype == DSW_F32 ? 32 : BN) == 0) && (N % WR == 0) &&
                         (((uintptr_t)G0 & am) == 0) && (K == 1 || ((uintptr_t)Grest & am) == 0);
    if (!aligned || (dtype == DSW_BF16 && (Fin % 8 != 0 || P.plane_stride % 8 != 0))) return 0;
    int64_t S = 0;
    if (!dsw_wgrad_dgrad_fused_try_launch(P, dtype == DSW_BF16 ? 1 : 0, wgrad_max_slabs(Fin, Fout, K), &S, stream, rc)) return 0;
    if (*rc != DSW_OK) return 1;
    *rc = dsw_wgrad_reduce_launch(partial, S, Fin, Fout, K, dW, db, K, 0, 1 << 30, dtype, stream, accumulate);
    return 1;
}
