// extern "C" entry points of libdsw_hip.so (see include/dsw_hip.h for the contract).
#include "dsw_common.h"
#include "../../include/dsw_hip.h"

// internal launchers (dsw_spmm.hip / dsw_gemm.hip)
int dsw_spmm_launch(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                    const void* X, void* Y, int64_t B, int64_t C, float alpha, const void* Z, float beta,
                    const void* Z2, float gamma, int dtype, hipStream_t stream, int hints = 0);
// hints: bit1 = the epilogue operands Z / Z2 are cold (not produced by the previous launch): stream them
// with non-temporal loads so they do not evict the gathered rows from L2 / Infinity Cache
#define DSW_SPMM_HINT_COLD_Z 2
int dsw_mix_fwd_launch(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                       int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream);
int dsw_mix_dgrad_launch(const void* dY, const void* W, void* G0, void* Grest, int64_t N, int64_t Fin,
                         int64_t Fout, int64_t K, int dtype, hipStream_t stream);
int64_t dsw_wgrad_slabs(int64_t N, int64_t Fin, int64_t Fout, int64_t K);
int dsw_wgrad_launch(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                     int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream);

static inline int64_t elem_size(int dtype) { return dtype == DSW_BF16 ? 2 : 4; }
static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

extern "C" {

int dsw_version(void) { return DSW_VERSION; }

const char* dsw_strerror(int code) {
    switch (code) {
        case DSW_OK: return "ok";
        case DSW_ERR_BAD_ARG: return "bad argument (shape / null pointer)";
        case DSW_ERR_BAD_DTYPE: return "unsupported dtype (expected DSW_F32 or DSW_BF16)";
        case DSW_ERR_WORKSPACE: return "workspace too small or null";
        case DSW_ERR_LAUNCH: return "HIP kernel launch failed";
        case DSW_ERR_ALIGN: return "pointer not sufficiently aligned";
        default: return "unknown error";
    }
}

int dsw_spmm_csr(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out,
                 int64_t v_in, int64_t nnz, const void* X, void* Y, int64_t B, int64_t C, float alpha,
                 const void* Z, float beta, const void* Z2, float gamma, int dtype, dsw_stream_t stream) {
    if (v_out < 0 || v_in < 0 || nnz < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    if (v_out == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !X || !Y || (nnz > 0 && (!colind || !vals))) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    return dsw_spmm_launch(rowptr, colind, vals, v_out, v_in, X, Y, B, C, alpha, Z, beta, Z2, gamma, dtype,
                           (hipStream_t)stream);
}

int dsw_cheb_basis_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V,
                       int64_t nnz, const void* X, void* T, int64_t B, int64_t C, int64_t K, int dtype,
                       dsw_stream_t stream) {
    if (K <= 1) return K == 1 ? DSW_OK : DSW_ERR_BAD_ARG;
    if (V < 0 || B < 0 || C < 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (V == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !X || !T) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t plane = B * V * C * elem_size(dtype);
    char* t = static_cast<char*>(T);
    int rc = dsw_spmm_launch(rowptr, colind, vals, V, V, X, t, B, C, 1.f, nullptr, 0.f, nullptr, 0.f, dtype,
                             (hipStream_t)stream);
    for (int64_t k = 2; k < K && rc == DSW_OK; ++k) {
        const void* prev = t + (k - 2) * plane;                     // T_{k-1}
        const void* prev2 = (k == 2) ? X : (t + (k - 3) * plane);    // T_{k-2}
        rc = dsw_spmm_launch(rowptr, colind, vals, V, V, prev, t + (k - 1) * plane, B, C, 2.f, prev2, -1.f,
                             nullptr, 0.f, dtype, (hipStream_t)stream);
    }
    return rc;
}

int dsw_cheb_basis_adj(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                       int64_t nnz, void* G0, void* Grest, int64_t B, int64_t C, int64_t K, int dtype,
                       dsw_stream_t stream) {
    if (K <= 1) return K == 1 ? DSW_OK : DSW_ERR_BAD_ARG;
    if (V < 0 || B < 0 || C < 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (V == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr_t || !G0 || !Grest) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t plane = B * V * C * elem_size(dtype);
    char* G = static_cast<char*>(Grest);
    int rc = DSW_OK;
    for (int64_t j = K - 1; j >= 1 && rc == DSW_OK; --j) {
        void* gm1 = (j == 1) ? G0 : static_cast<void*>(G + (j - 2) * plane);   // G_{j-1}
        const void* gj = G + (j - 1) * plane;                                 // G_j
        const void* gp1 = (j + 1 <= K - 1) ? (G + j * plane) : nullptr;       // G_{j+1}
        // G_{j-1} / G_{j+1} were written by the dgrad GEMM, not by the previous launch: cold operands
        rc = dsw_spmm_launch(rowptr_t, colind_t, vals_t, V, V, gj, gm1, B, C, (j == 1) ? 1.f : 2.f, gm1, 1.f, gp1,
                             -1.f, dtype, (hipStream_t)stream, DSW_SPMM_HINT_COLD_Z);
    }
    return rc;
}

int dsw_cheb_mix_fwd(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                     int64_t Fin, int64_t Fout, int64_t K, int dtype, dsw_stream_t stream) {
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (N == 0) return DSW_OK;
    if (!X || !W || !Y || (K > 1 && !T)) return DSW_ERR_BAD_ARG;
    return dsw_mix_fwd_launch(X, T, W, bias, Y, N, Fin, Fout, K, dtype, (hipStream_t)stream);
}

int dsw_cheb_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                 const void* X, const void* W, const void* bias, void* Y, void* T, int64_t B, int64_t Fin,
                 int64_t Fout, int64_t K, int dtype, dsw_stream_t stream) {
    if (K <= 0) return DSW_ERR_BAD_ARG;
    int rc = DSW_OK;
    if (K > 1) {
        rc = dsw_cheb_basis_fwd(rowptr, colind, vals, V, nnz, X, T, B, Fin, K, dtype, stream);
        if (rc != DSW_OK) return rc;
    }
    return dsw_cheb_mix_fwd(X, T, W, bias, Y, B * V, Fin, Fout, K, dtype, stream);
}

int64_t dsw_cheb_bwd_workspace_bytes(int64_t B, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    const int64_t N = B * V;
    const int64_t g = round_up((K - 1) * N * Fin * elem_size(dtype), 256);
    const int64_t S = dsw_wgrad_slabs(N, Fin, Fout, K);
    const int64_t p = round_up((S > 0 ? S : 1) * (K * Fin + 1) * Fout * 4, 256);
    return g + p + 256;
}

int dsw_cheb_bwd(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                 int64_t nnz, const void* X, const void* T, const void* W, const void* dY, void* dX, void* dW,
                 void* db, void* workspace, int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout,
                 int64_t K, int dtype, dsw_stream_t stream) {
    if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0 || K <= 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t need = dsw_cheb_bwd_workspace_bytes(B, V, Fin, Fout, K, dtype);
    if (!workspace || workspace_bytes < need) return DSW_ERR_WORKSPACE;
    if (!dY || !X || !W || (K > 1 && !T)) return DSW_ERR_BAD_ARG;
    const int64_t N = B * V;
    hipStream_t s = (hipStream_t)stream;
    // carve the workspace (256-byte aligned base)
    char* ws = reinterpret_cast<char*>(round_up((int64_t)(uintptr_t)workspace, 256));
    const int64_t plane = N * Fin * elem_size(dtype);
    char* G = ws;                                                    // G_1 .. G_{K-1}
    float* partial = reinterpret_cast<float*>(ws + round_up((K - 1) * plane, 256));
    int rc = DSW_OK;
    if (dX != nullptr && N > 0) {
        if (K > 1 && !rowptr_t) return DSW_ERR_BAD_ARG;
        rc = dsw_mix_dgrad_launch(dY, W, dX, G, N, Fin, Fout, K, dtype, s);
        if (rc == DSW_OK && K > 1)
            rc = dsw_cheb_basis_adj(rowptr_t, colind_t, vals_t, V, nnz, dX, G, B, Fin, K, dtype, stream);
        if (rc != DSW_OK) return rc;
    }
    if (dW != nullptr) {
        rc = dsw_wgrad_launch(X, T, dY, dW, db, partial, N, Fin, Fout, K, dtype, s);
    }
    return rc;
}

}  // extern "C"
