/* cheb_oracle.c - plain C restatement of the ConvCheb / RemapBlock arithmetic.  TEST INFRASTRUCTURE ONLY.
 *
 * Third, independent restatement of the reference algorithm (besides the torch op-sequence and the
 * numpy fp64 closed form in cheb_oracle.py): scalar loops, double accumulation, no dependencies.
 * Follows /root/reference/modules/layers.py:113-180 (conv_cheb: T0 = x, T1 = L x, Tk = 2 L T(k-1) - T(k-2),
 * y = [T0|..|T(K-1)] W with W indexed [f][k][o]), :375 (bias) and :956-964 (RemapBlock), in the node-major
 * [B, V, C] layout.  Backward is the hand-derived adjoint, valid for non-symmetric L.
 * Checked against the golden fixtures recorded from the reference (tests/test_oracle_golden.py).
 * Only tests / smoke / the bench cpu_baseline leg may load it; the product never does.
 *
 * build: gcc -O2 -shared -fPIC -o _build/libcheb_oracle.so cheb_oracle.c   (see Makefile)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y[b, r, :] = sum_p vals[p] * x[b, colind[p], :]   (one CSR application per sample) */
static void spmm(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                 const double* x, double* y, int64_t B, int64_t C) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t r = 0; r < v_out; ++r) {
            double* yr = y + (b * v_out + r) * C;
            memset(yr, 0, (size_t)C * sizeof(double));
            for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                const double a = (double)vals[p];
                const double* xr = x + (b * v_in + colind[p]) * C;
                for (int64_t c = 0; c < C; ++c) yr[c] += a * xr[c];
            }
        }
}

/* RemapBlock.forward (layers.py:956-964): y = M x per sample; x [B, v_in, C] -> y [B, v_out, C] (double) */
int oracle_remap(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                 const double* x, double* y, int64_t B, int64_t C) {
    spmm(rowptr, colind, vals, v_out, v_in, x, y, B, C);
    return 0;
}

/* ConvCheb forward.  x [B,V,Fin], w [Fin,K,Fout], bias [Fout] or NULL -> y [B,V,Fout];
 * basis (optional, [K,B,V,Fin]) receives T_0..T_{K-1}. */
int oracle_cheb_forward(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, const double* x,
                        const double* w, const double* bias, double* y, double* basis, int64_t B, int64_t Fin,
                        int64_t Fout, int64_t K) {
    const int64_t plane = B * V * Fin;
    double* T = basis ? basis : (double*)malloc((size_t)(K * plane) * sizeof(double));
    if (!T) return -1;
    memcpy(T, x, (size_t)plane * sizeof(double));
    if (K > 1) spmm(rowptr, colind, vals, V, V, T, T + plane, B, Fin);
    for (int64_t k = 2; k < K; ++k) {
        double* tk = T + k * plane;
        spmm(rowptr, colind, vals, V, V, T + (k - 1) * plane, tk, B, Fin);
        const double* tm2 = T + (k - 2) * plane;
        for (int64_t i = 0; i < plane; ++i) tk[i] = 2.0 * tk[i] - tm2[i];
    }
    for (int64_t n = 0; n < B * V; ++n)
        for (int64_t o = 0; o < Fout; ++o) {
            double acc = bias ? bias[o] : 0.0;
            for (int64_t k = 0; k < K; ++k) {
                const double* t = T + k * plane + n * Fin;
                for (int64_t f = 0; f < Fin; ++f) acc += t[f] * w[(f * K + k) * Fout + o];
            }
            y[n * Fout + o] = acc;
        }
    if (!basis) free(T);
    return 0;
}

/* ConvCheb backward.  rowptr_t/colind_t/vals_t: CSR of L^T.  basis: T_0..T_{K-1} from the forward.
 * dW[f,k,o] = sum_n T_k[n,f] dY[n,o]; db[o] = sum_n dY[n,o]; G_k = dY W_k^T;
 * for j = K-1..1: G_{j-1} += (j > 1 ? 2 : 1) L^T G_j - G_{j+1}; dX = G_0. */
int oracle_cheb_backward(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                         const double* basis, const double* w, const double* dy, double* dx, double* dw, double* db,
                         int64_t B, int64_t Fin, int64_t Fout, int64_t K) {
    const int64_t N = B * V, plane = N * Fin;
    double* G = (double*)calloc((size_t)((K + 1) * plane), sizeof(double)); /* G_0..G_{K-1}, scratch */
    if (!G) return -1;
    double* tmp = G + K * plane;
    memset(dw, 0, (size_t)(Fin * K * Fout) * sizeof(double));
    if (db) memset(db, 0, (size_t)Fout * sizeof(double));
    for (int64_t n = 0; n < N; ++n) {
        const double* g = dy + n * Fout;
        if (db) for (int64_t o = 0; o < Fout; ++o) db[o] += g[o];
        for (int64_t k = 0; k < K; ++k) {
            const double* t = basis + k * plane + n * Fin;
            double* gk = G + k * plane + n * Fin;
            for (int64_t f = 0; f < Fin; ++f) {
                const double* wr = w + (f * K + k) * Fout;
                double acc = 0.0;
                for (int64_t o = 0; o < Fout; ++o) {
                    acc += g[o] * wr[o];
                    dw[(f * K + k) * Fout + o] += t[f] * g[o];
                }
                gk[f] = acc;
            }
        }
    }
    for (int64_t j = K - 1; j >= 1; --j) {
        spmm(rowptr_t, colind_t, vals_t, V, V, G + j * plane, tmp, B, Fin);
        double* gm1 = G + (j - 1) * plane;
        const double c = (j > 1) ? 2.0 : 1.0;
        const double* gp1 = (j + 1 <= K - 1) ? G + (j + 1) * plane : NULL;
        for (int64_t i = 0; i < plane; ++i) gm1[i] += c * tmp[i] - (gp1 ? gp1[i] : 0.0);
    }
    memcpy(dx, G, (size_t)plane * sizeof(double));
    free(G);
    return 0;
}
