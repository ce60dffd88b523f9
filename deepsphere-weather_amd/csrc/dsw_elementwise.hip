// Elementwise pieces of the residual block around the Chebyshev convolutions (my_models_graph.py:205-216):
//   forward   y = w * c + r            (ReZero scale of the conv stack's output + residual branch), one pass
//   backward  grad_c = w * g,  grad_w = sum(g * c)   (one pass over g and c + a tiny deterministic second stage);
//             grad_r = g needs no kernel.
// The reference does this with an in-place mul and an in-place add (two passes forward; mul, mul, reduce backward).
// Pure streaming kernels: 16 bytes per lane, grid-stride, HBM-bound.
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 2048;   // also the size of the partial-sum buffer the caller provides (floats)

template <bool BF16>
struct Ew {
    static constexpr int V = BF16 ? 8 : 4;   // elements per 16 bytes
    static __device__ __forceinline__ void load(const void* p, size_t i, float (&v)[V]) {
        const uint4 t = *reinterpret_cast<const uint4*>(static_cast<const char*>(p) + i * (BF16 ? 2 : 4));
        if constexpr (BF16) {
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
        } else {
            v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
        }
    }
    static __device__ __forceinline__ void store(void* p, size_t i, const float (&v)[V]) {
        uint4 t;
        if constexpr (BF16) {
            t.x = f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16); t.y = f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            t.z = f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16); t.w = f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
        } else {
            t = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        }
        *reinterpret_cast<uint4*>(static_cast<char*>(p) + i * (BF16 ? 2 : 4)) = t;
    }
    static __device__ __forceinline__ float load1(const void* p, size_t i) {
        if constexpr (BF16) return bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
        else return static_cast<const float*>(p)[i];
    }
    static __device__ __forceinline__ void store1(void* p, size_t i, float v) {
        if constexpr (BF16) static_cast<uint16_t*>(p)[i] = f32_to_bf16(v);
        else static_cast<float*>(p)[i] = v;
    }
};

template <bool BF16>
__global__ __launch_bounds__(EW_THREADS) void rezero_fwd_kernel(const void* c, const void* r, const void* wp, void* y, long n,
                                                                int vec) {
    using E = Ew<BF16>;
    constexpr int V = E::V;
    const float w = E::load1(wp, 0);
    const long nv = vec ? n / V : 0;   // vec = 0: some base pointer is not 16-byte aligned -> every element takes the scalar path
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (long)gridDim.x * EW_THREADS) {
        float a[V], b[V], o[V];
        E::load(c, (size_t)i * V, a);
        E::load(r, (size_t)i * V, b);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = fmaf(w, a[j], b[j]);
        E::store(y, (size_t)i * V, o);
    }
    for (long i = nv * V + (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS)
        E::store1(y, i, fmaf(w, E::load1(c, i), E::load1(r, i)));
}

// same with a row stride on the output: y[row * ldy + j] = w * c[row * C + j] + r[row * C + j] - the block writes straight
// into its half of the decoder's concatenation buffer (my_models_graph.py:528-545 builds that buffer with torch.cat)
template <bool BF16>
__global__ __launch_bounds__(EW_THREADS) void rezero_fwd_ld_kernel(const void* c, const void* r, const void* wp, void* y,
                                                                   long rows, int cpr, long ldy) {
    using E = Ew<BF16>;
    constexpr int V = E::V;
    const float w = E::load1(wp, 0);
    const long nv = rows * cpr;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (long)gridDim.x * EW_THREADS) {
        const long row = i / cpr;
        const int j = (int)(i - row * cpr);
        float a[V], b[V], o[V];
        E::load(c, (size_t)i * V, a);
        E::load(r, (size_t)i * V, b);
#pragma unroll
        for (int t = 0; t < V; ++t) o[t] = fmaf(w, a[t], b[t]);
        E::store(y, (size_t)row * ldy + (size_t)j * V, o);
    }
}

// grad_c = w * g; partial[block] = sum over the block's elements of g * c (fixed order -> reproducible)
template <bool BF16>
__global__ __launch_bounds__(EW_THREADS) void rezero_bwd_kernel(const void* g, const void* c, const void* wp, void* gc,
                                                                float* partial, long n, int vec) {
    using E = Ew<BF16>;
    constexpr int V = E::V;
    __shared__ float red[EW_THREADS / 64];
    const float w = E::load1(wp, 0);
    const long nv = vec ? n / V : 0;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (long)gridDim.x * EW_THREADS) {
        float a[V], b[V], o[V];
        E::load(g, (size_t)i * V, a);
        E::load(c, (size_t)i * V, b);
#pragma unroll
        for (int j = 0; j < V; ++j) { o[j] = w * a[j]; acc = fmaf(a[j], b[j], acc); }
        if (gc != nullptr) E::store(gc, (size_t)i * V, o);
    }
    for (long i = nv * V + (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
        const float a = E::load1(g, i);
        if (gc != nullptr) E::store1(gc, i, w * a);
        acc = fmaf(a, E::load1(c, i), acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < EW_THREADS / 64; ++k) s += red[k];
        partial[blockIdx.x] = s;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void rezero_bwd_final_kernel(const float* partial, int nblocks, void* gw) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) Ew<BF16>::store1(gw, 0, red[0]);
}

// Parameter gradients behind a fused  out = s * conv(x) + r  (ReZero tail of a residual block): the backward kernels
// ran on g = d out with the UNSCALED weight gradients dW_raw = T^T g, db_raw = sum g; then
//     dW = s * dW_raw,   db = s * db_raw,   ds = <W, dW_raw> + <bias, db_raw>      (= sum g * conv(x), without conv(x))
// ONE launch of up to 64 workgroups: each scales its slice and leaves a partial sum in the workspace; the workgroup that
// draws the last ticket adds the partials in index order (bit-identical reruns) and re-arms the ticket.
constexpr int RPG_BLOCKS = 64;
template <bool BF16>
__global__ __launch_bounds__(256) void rezero_param_grads_kernel(const void* W, const void* bias, const void* dW_raw,
                                                                 const void* db_raw, const void* sp, void* dW, void* db,
                                                                 void* ds, long n_w, long n_b, float* ws) {
    using E = Ew<BF16>;
    __shared__ float red[4];
    __shared__ int last;
    const float s = E::load1(sp, 0);
    float acc = 0.f;
    const long n = n_w + n_b;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (i < n_w) {
            const float g = E::load1(dW_raw, i);
            acc = fmaf(E::load1(W, i), g, acc);
            E::store1(dW, i, s * g);
        } else {
            const long j = i - n_w;
            const float g = E::load1(db_raw, j);
            acc = fmaf(E::load1(bias, j), g, acc);
            E::store1(db, j, s * g);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(ws), 1u);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float t = 0.f;
        for (unsigned k = 0; k < gridDim.x; ++k) t += reinterpret_cast<volatile float*>(ws)[1 + k];
        E::store1(ds, 0, t);
        *reinterpret_cast<unsigned*>(ws) = 0u;        // re-armed for the next launch on this workspace
    }
}

// y = relu(y) in place (the mix-first layers end in an SpMM, not in a GEMM epilogue);  g_out = y > 0 ? g : 0
template <bool BF16, bool BWD>
__global__ __launch_bounds__(EW_THREADS) void relu_kernel(const void* g, const void* y, void* out, long n, int vec) {
    using E = Ew<BF16>;
    constexpr int V = E::V;
    const long nv = vec ? n / V : 0;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nv; i += (long)gridDim.x * EW_THREADS) {
        float a[V], b[V], o[V];
        E::load(y, (size_t)i * V, b);
        if constexpr (BWD) E::load(g, (size_t)i * V, a);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = BWD ? (b[j] > 0.f ? a[j] : 0.f) : (b[j] < 0.f ? 0.f : b[j]);
        E::store(out, (size_t)i * V, o);
    }
    for (long i = nv * V + (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
        const float b = E::load1(y, i);
        E::store1(out, i, BWD ? (b > 0.f ? E::load1(g, i) : 0.f) : (b < 0.f ? 0.f : b));
    }
}

static int ew_blocks(long n, int v) {
    long b = ((n + v - 1) / v + EW_THREADS - 1) / EW_THREADS;
    if (b < 1) b = 1;
    if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    return (int)b;
}

}  // namespace

namespace {
// W' = W with plane K-3 replaced by W[:, K-3, :] - W[:, K-1, :]   (W: [Fin][K][Fout])
template <bool BF16>
__global__ void fold_w_kernel(const void* __restrict__ W, void* __restrict__ Wf, const int Fin, const int K, const int Fout) {
    const int n = Fin * K * Fout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i % Fout, k = (i / Fout) % K, f = i / (Fout * K);
        float v = Ew<BF16>::load1(W, i);
        if (k == K - 3) v -= Ew<BF16>::load1(W, ((size_t)f * K + (K - 1)) * Fout + o);
        if constexpr (BF16) static_cast<uint16_t*>(Wf)[i] = f32_to_bf16(v);
        else static_cast<float*>(Wf)[i] = v;
    }
}
}  // namespace

// The top of the adjoint (and of the Clenshaw) recurrence subtracts the RAW plane K-1 from plane K-3:
// G'_{K-3} = G_{K-3} + c L^T G'_{K-2} - G_{K-1}.  Both planes come out of one GEMM with the layer's weights, so the
// subtraction is folded into the weights of plane K-3 (a few KB of parameters) instead of a pass over a [B, V, C] plane:
// one epilogue operand less in that step of the recurrence (for K = 3: in its last step, which is then a one-operand step).
int dsw_fold_w_launch(const void* W, void* Wf, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t s) {
    if (K < 3 || !W || !Wf) return DSW_ERR_BAD_ARG;
    const int64_t n = Fin * K * Fout;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    if (dtype == DSW_BF16) DSW_LAUNCH(fold_w_kernel<true>, dim3(blocks), dim3(256), 0, s, W, Wf, (int)Fin, (int)K, (int)Fout);
    else DSW_LAUNCH(fold_w_kernel<false>, dim3(blocks), dim3(256), 0, s, W, Wf, (int)Fin, (int)K, (int)Fout);
    return dsw_check_launch();
}

int dsw_relu_inplace_launch(void* y, int64_t n, int dtype, hipStream_t s) {
    if (n <= 0) return DSW_OK;
    const int vec = dsw_aligned16(y) ? 1 : 0;
    if (dtype == DSW_F32)
        DSW_LAUNCH((relu_kernel<false, false>), dim3(ew_blocks(n, vec ? 4 : 1)), dim3(EW_THREADS), 0, s, nullptr, y, y, (long)n, vec);
    else
        DSW_LAUNCH((relu_kernel<true, false>), dim3(ew_blocks(n, vec ? 8 : 1)), dim3(EW_THREADS), 0, s, nullptr, y, y, (long)n, vec);
    return dsw_check_launch();
}

extern "C" {

int dsw_relu_bwd(const void* dY, const void* Y, void* dYm, int64_t n, int dtype, dsw_stream_t stream) {
    if (n < 0) return DSW_ERR_BAD_ARG;
    if (n == 0) return DSW_OK;
    if (!dY || !Y || !dYm) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int vec = (dsw_aligned16(dY) && dsw_aligned16(Y) && dsw_aligned16(dYm)) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    DswTraceScope trace_(s, DSW_ROLE_ELEMENTWISE, n, 1, 0);
    if (dtype == DSW_F32)
        DSW_LAUNCH((relu_kernel<false, true>), dim3(ew_blocks(n, vec ? 4 : 1)), dim3(EW_THREADS), 0, s, dY, Y, dYm, (long)n, vec);
    else
        DSW_LAUNCH((relu_kernel<true, true>), dim3(ew_blocks(n, vec ? 8 : 1)), dim3(EW_THREADS), 0, s, dY, Y, dYm, (long)n, vec);
    return dsw_check_launch();
}


int64_t dsw_rezero_residual_workspace_bytes(void) { return (int64_t)EW_MAX_BLOCKS * 4; }

int dsw_rezero_residual_fwd(const void* c, const void* r, const void* w, void* y, int64_t n, int dtype,
                            dsw_stream_t stream) {
    if (n < 0) return DSW_ERR_BAD_ARG;
    if (n == 0) return DSW_OK;
    if (!c || !r || !w || !y) return DSW_ERR_BAD_ARG;
    // contiguous views with a storage offset (batch slices, an Identity residual of a sliced input) need not be 16-byte
    // aligned: they take the scalar path of the same kernel (the reference simply works there)
    const int vec = (dsw_aligned16(c) && dsw_aligned16(r) && dsw_aligned16(y)) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    DswTraceScope trace_(s, DSW_ROLE_ELEMENTWISE, n, 2, 0);
    if (dtype == DSW_F32)
        DSW_LAUNCH(rezero_fwd_kernel<false>, dim3(ew_blocks(n, vec ? 4 : 1)), dim3(EW_THREADS), 0, s, c, r, w, y,
                           (long)n, vec);
    else if (dtype == DSW_BF16)
        DSW_LAUNCH(rezero_fwd_kernel<true>, dim3(ew_blocks(n, vec ? 8 : 1)), dim3(EW_THREADS), 0, s, c, r, w, y,
                           (long)n, vec);
    else
        return DSW_ERR_BAD_DTYPE;
    return dsw_check_launch();
}

int dsw_rezero_residual_fwd_ld(const void* c, const void* r, const void* w, void* y, int64_t rows, int64_t C, int64_t ldy,
                               int dtype, dsw_stream_t stream) {
    if (rows < 0 || C < 0 || ldy < C) return DSW_ERR_BAD_ARG;
    if (rows == 0 || C == 0) return DSW_OK;
    if (!c || !r || !w || !y) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int V = dtype == DSW_BF16 ? 8 : 4;
    if (C % V != 0 || ldy % V != 0 || !dsw_aligned16(c) || !dsw_aligned16(r) || !dsw_aligned16(y)) return DSW_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    DswTraceScope trace_(s, DSW_ROLE_ELEMENTWISE, rows * C, 2, 0);
    const int cpr = (int)(C / V);
    if (dtype == DSW_F32)
        DSW_LAUNCH(rezero_fwd_ld_kernel<false>, dim3(ew_blocks(rows * C, V)), dim3(EW_THREADS), 0, s, c, r, w, y,
                           (long)rows, cpr, (long)ldy);
    else
        DSW_LAUNCH(rezero_fwd_ld_kernel<true>, dim3(ew_blocks(rows * C, V)), dim3(EW_THREADS), 0, s, c, r, w, y,
                           (long)rows, cpr, (long)ldy);
    return dsw_check_launch();
}

}  // extern "C"

int dsw_rezero_param_grads_launch(const void* W, const void* bias, const void* dW_raw, const void* db_raw, const void* scale,
                                  void* dW, void* db, void* dscale, int64_t n_w, int64_t n_b, void* workspace, int dtype,
                                  hipStream_t stream) {
    const long n = n_w + n_b;
    int nb = (int)((n + 2047) / 2048);
    nb = nb < 1 ? 1 : (nb > RPG_BLOCKS ? RPG_BLOCKS : nb);
    float* ws = static_cast<float*>(workspace);
    if (dtype == DSW_F32)
        DSW_LAUNCH(rezero_param_grads_kernel<false>, dim3(nb), dim3(256), 0, stream, W, bias, dW_raw, db_raw, scale, dW,
                           db, dscale, (long)n_w, (long)n_b, ws);
    else
        DSW_LAUNCH(rezero_param_grads_kernel<true>, dim3(nb), dim3(256), 0, stream, W, bias, dW_raw, db_raw, scale, dW,
                           db, dscale, (long)n_w, (long)n_b, ws);
    return dsw_check_launch();
}
int64_t dsw_rezero_param_grads_ws_bytes_impl() { return (int64_t)(1 + RPG_BLOCKS) * 4; }

extern "C" {

int dsw_rezero_residual_bwd(const void* g, const void* c, const void* w, void* grad_c, void* grad_w, void* workspace,
                            int64_t workspace_bytes, int64_t n, int dtype, dsw_stream_t stream) {
    if (n < 0) return DSW_ERR_BAD_ARG;
    if (!g || !c || !w || !grad_w) return DSW_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < dsw_rezero_residual_workspace_bytes()) return DSW_ERR_WORKSPACE;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int vec = (dsw_aligned16(g) && dsw_aligned16(c) && (!grad_c || dsw_aligned16(grad_c))) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    DswTraceScope trace_(s, DSW_ROLE_ELEMENTWISE, n, 3, 0);
    float* partial = static_cast<float*>(workspace);
    const int nb = ew_blocks(n, !vec ? 1 : dtype == DSW_BF16 ? 8 : 4);
    if (dtype == DSW_F32) {
        DSW_LAUNCH(rezero_bwd_kernel<false>, dim3(nb), dim3(EW_THREADS), 0, s, g, c, w, grad_c, partial, (long)n, vec);
        DSW_LAUNCH(rezero_bwd_final_kernel<false>, dim3(1), dim3(256), 0, s, partial, nb, grad_w);
    } else {
        DSW_LAUNCH(rezero_bwd_kernel<true>, dim3(nb), dim3(EW_THREADS), 0, s, g, c, w, grad_c, partial, (long)n, vec);
        DSW_LAUNCH(rezero_bwd_final_kernel<true>, dim3(1), dim3(256), 0, s, partial, nb, grad_w);
    }
    return dsw_check_launch();
}

}  // extern "C"
