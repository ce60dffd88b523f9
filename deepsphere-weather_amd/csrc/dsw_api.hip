// extern "C" entry points of libdsw_hip.so (see include/dsw_hip.h for the contract).
#include "dsw_common.h"
#include "../../include/dsw_hip.h"
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <vector>

// ---- launch tracing (dsw_trace_begin / dsw_trace_end; mechanism: dsw_common.h) ----------------------------------------
namespace {
struct DswTraceRec {
    int role;                 // -1: a kernel launch; 0: start of an entry point; > 0: end of that role
    int a0, a1, a2;
    const char* name;         // kernel launches: the kernel expression of the launch site (static string)
    hipEvent_t e0, e1;
};
struct DswTrace {
    std::mutex mu;
    std::atomic<bool> on{false};
    std::vector<hipEvent_t> ev;      // pool: two per kernel launch
    std::vector<DswTraceRec> rec;
    size_t n_ev = 0;
    bool overflow = false;
};
DswTrace g_trace;
std::atomic<int> g_build_flags{0};

static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
    return cs != hipStreamCaptureStatusNone;
}
static void trace_event(hipStream_t s, int role, int64_t a0, int64_t a1, int64_t a2) {
    if (!g_trace.on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_trace.mu);
    if (!g_trace.on.load(std::memory_order_relaxed) || capturing(s)) return;
    g_trace.rec.push_back(DswTraceRec{role, (int)a0, (int)a1, (int)a2, nullptr, nullptr, nullptr});
}
// start marker of an entry point (role 0: closes nothing) / end of a role (owns the kernels since the previous marker)
static inline void trace_start(dsw_stream_t s) { trace_event((hipStream_t)s, 0, 0, 0, 0); }
static inline void trace_mark(dsw_stream_t s, int role, int64_t a0 = 0, int64_t a1 = 0, int64_t a2 = 0) { trace_event((hipStream_t)s, role, a0, a1, a2); }
}  // namespace

bool dsw_trace_kernel(const char* name, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_trace.on.load(std::memory_order_relaxed)) return false;
    std::lock_guard<std::mutex> lk(g_trace.mu);
    if (!g_trace.on.load(std::memory_order_relaxed) || capturing(s)) return false;
    if (g_trace.n_ev + 2 > g_trace.ev.size()) { g_trace.overflow = true; return false; }
    *e0 = g_trace.ev[g_trace.n_ev++];
    *e1 = g_trace.ev[g_trace.n_ev++];
    g_trace.rec.push_back(DswTraceRec{-1, 0, 0, 0, name, *e0, *e1});
    return true;
}

void dsw_trace_point(hipStream_t s, int role, int64_t a0, int64_t a1, int64_t a2) { trace_event(s, role, a0, a1, a2); }

// every translation unit reports the diagnostics switches it was compiled with (dsw_common.h: DSW_TU_BUILD_FLAGS)
int dsw_register_build_flags(int flags) {
    g_build_flags.fetch_or(flags, std::memory_order_relaxed);
    return flags;
}

// internal launchers (dsw_spmm.hip / dsw_gemm.hip)
int dsw_spmm_launch_ld(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                       const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C, float alpha, const void* Z,
                       float beta, const void* Z2, float gamma, int dtype, hipStream_t stream, int hints = 0,
                       int64_t ldz = 0, const int* long_list = nullptr, int n_long = 0, int long_thr = 0);
int dsw_remap_launch(const dsw_remap_plan* plan, const int* rowptr, const int* colind, const float* vals, int64_t v_out,
                     int64_t v_in, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C, const void* Z,
                     int64_t ldz, float beta, int dtype, hipStream_t stream);
int dsw_spmm_launch(const int* rowptr, const int* colind, const float* vals, int64_t v_out, int64_t v_in,
                    const void* X, void* Y, int64_t B, int64_t C, float alpha, const void* Z, float beta,
                    const void* Z2, float gamma, int dtype, hipStream_t stream, int hints = 0);
// hints: bit1 = the epilogue operands Z / Z2 are cold (not produced by the previous launch): stream them
// with non-temporal loads so they do not evict the gathered rows from L2 / Infinity Cache
#define DSW_SPMM_HINT_COLD_Z 2
struct DswEpiExtra {   // optional epilogue operands / scratch of the channel-mix launchers (dsw_gemm.hip)
    const void* scale;
    const void* R;
    int64_t ldr;
    int64_t ldc;
    void* ws;            // scratch for an image of W (pre-split terms, or a folded copy)
    int64_t ws_bytes;
    int fold;            // 1: plane K-3 with W[:, K-3, :] - W[:, K-1, :]
};
// bytes of that scratch for a layer: the pre-split image of the streaming GEMM (1.5x the fp32 weights, columns padded to the
// 128-column tile) or a copy of the weights, whichever is larger
// Can one of the layer's four channel-mix forms take the streaming-W GEMM (dsw_gemm.hip: launch_ts_gemm, fp32 only:
// outputs of >= 256 columns, or a split W panel beyond LDS)?  Only then is the scratch of its balanced decomposition needed.
static inline bool w_may_stream(int64_t red, int64_t cols) {
    if (cols <= 32) return false;
    const int64_t nat = cols <= 128 ? (cols + 31) / 32 * 32 : 128;
    return cols >= 256 || 3 * nat * (red + 8) * 2 > 100 * 1024;    // (the launcher's own bound is ~140 KB of panel: conservative)
}
static inline int64_t w_image_bytes(int64_t Fin, int64_t Fout, int64_t K, int dtype = DSW_F32) {
    const int64_t fwd = (K * Fin + 31) / 32 * 32 * ((Fout + 127) / 128 * 128) * 15 / 2;      // reduction K Fin, columns Fout
    const int64_t bwd = (Fout + 31) / 32 * 32 * ((K * Fin + 127) / 128 * 128) * 15 / 2;      // reduction Fout, columns K Fin
    const int64_t zmx = (Fin + 31) / 32 * 32 * ((K * Fout + 127) / 128 * 128) * 15 / 2;      // mix-first planes: columns K Fout
    const int64_t zdg = (K * Fout + 31) / 32 * 32 * ((Fin + 127) / 128 * 128) * 15 / 2;      // mix-first dX: reduction K Fout, columns Fin
    int64_t m = fwd > bwd ? fwd : bwd;
    if (zmx > m) m = zmx;
    if (zdg > m) m = zdg;
    const int64_t copy = Fin * K * Fout * 4;
    // + the scratch of the streaming GEMM's balanced decomposition (dsw_gemm_x3s.hip: 4 KiB of flags, one partial 256 x 128
    // fp32 tile per workgroup, 256 workgroups) - only for layers one of whose GEMMs can take that kernel (the
    // 32-channel layers paid 32 MiB of scratch per backward call for a kernel they never launch)
    const bool streams = dtype == DSW_F32 && (w_may_stream(K * Fin, Fout) || w_may_stream(Fout, K * Fin) ||
                                               w_may_stream(Fin, K * Fout) || w_may_stream(K * Fout, Fin));
    return (m > copy ? m : copy) + 512 + (streams ? 4096 + 256 * (256 * 128 * 4) : 0);
}
int dsw_mix_fwd_launch(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                       int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int relu = 0,
                       const DswEpiExtra* extra = nullptr);
int dsw_relu_inplace_launch(void* y, int64_t n, int dtype, hipStream_t s);
int dsw_fold_w_launch(const void* W, void* Wf, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t s);
int dsw_mix_dgrad_launch(const void* dY, const void* W, void* G0, void* Grest, int64_t N, int64_t Fin,
                         int64_t Fout, int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra = nullptr);
int dsw_rezero_param_grads_launch(const void* W, const void* bias, const void* dW_raw, const void* db_raw, const void* scale,
                                  void* dW, void* db, void* dscale, int64_t n_w, int64_t n_b, void* workspace, int dtype,
                                  hipStream_t stream);
int64_t dsw_rezero_param_grads_ws_bytes_impl();
int64_t dsw_wgrad_slabs(int64_t N, int64_t Fin, int64_t Fout, int64_t K);
int dsw_spmm2_launch(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z1, const void* Z1b,
                     const void* Z2, void* Y1, void* Y2, int64_t B, int64_t C, float a1, float b1, float d1,
                     float a2, float b2, float c2, int dtype, hipStream_t stream);
int dsw_spmm1s_launch(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z, const void* Z2, void* Y,
                      int64_t B, int64_t C, float a, float b, float c, int dtype, hipStream_t stream, int stream_out);
int dsw_spmm1s_supported(const dsw_hop2_plan* plan, int64_t C, int dtype);
int dsw_wgrad_launch(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                     int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int accumulate);
int dsw_wgrad_launch_ex(const void* X, const void* T, const void* dY, void* dW, void* db, float* partial,
                        int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                        int64_t K_out, int64_t k_off, int accumulate);
int dsw_wgrad_mixfirst_launch(const void* X, const void* dY, const void* D, void* dW, void* db, float* partial,
                              int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int accumulate);
int dsw_bwd_gemm_fused_try(const void* X, const void* T, const void* W, const void* dY, void* dW, void* db, void* G0,
                           void* Grest, float* partial, int64_t N, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                           hipStream_t stream, int* rc, int accumulate, int fold);
int dsw_zmix_launch(const void* X, const void* W, const void* bias, void* Z0, void* Zrest, int64_t N, int64_t Fin,
                    int64_t Fout, int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra = nullptr);
int dsw_cheb3_fwd_fused_try(const dsw_hop2_plan* plan, int64_t V, const void* X, const void* W, const void* bias, void* Y,
                            void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream,
                            int* rc, int relu);
int dsw_cheb3_fwd_fused_eligible(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype);
int dsw_cheb3_hop2mix_eligible(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype);
int dsw_cheb3_hop2mix_try(const dsw_hop2_plan* plan, int64_t V, const void* X, const void* W, const void* bias, void* Y,
                          void* T, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype, hipStream_t stream, int* rc,
                          int relu, void (*after_hop1)(void*), void* ctx);
int dsw_cheb3_bwd_dual_eligible(const dsw_hop2_plan* plan_t, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype);
int64_t dsw_cheb3_bwd_dual_ws_bytes(void);
int dsw_cheb3_bwd_dual_try(const dsw_hop2_plan* plan_t, int64_t V, const void* X, const void* dY, const void* W, void* dX,
                           void* dW, void* db, float* partial, int64_t B, int64_t Fin, int64_t Fout, int64_t K, int dtype,
                           hipStream_t stream, int* rc, int accumulate);
int dsw_zdgrad_launch(const void* dY, const void* D, const void* W, void* dX, int64_t N, int64_t Fin, int64_t Fout,
                      int64_t K, int dtype, hipStream_t stream, const DswEpiExtra* extra = nullptr);

static inline int64_t elem_size(int dtype) { return dtype == DSW_BF16 ? 2 : 4; }
static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Evaluation order of the layer.  Y = sum_k T_k(L) X W_k can run the recurrence on the Fin channels of X and mix
// afterwards (basis-first), or mix first (Z_k = X W_k) and run the Clenshaw-form recurrence
// c_{K-1} = Z_{K-1}, c_j = a_j L c_{j+1} + Z_j - c_{j+2} (a_j = 2, a_0 = 1), Y = c_0 on the Fout channels.
// Same result (association differs: fp32 rounding only), same GEMM flops; the SpMM hops - the HBM/gather-bound
// part - scale with the channel count they run on, so layers that shrink the channel count (the decoder half of the
// UNet, 64 -> 2 output layer) take the mix-first order.  Its backward is the dual: Chebyshev basis of dY under L^T,
// dX = sum_k D_k W_k^T, dW_k = X^T D_k - and needs nothing saved from the forward except X.
static bool mix_first(int64_t Fin, int64_t Fout, int64_t K) {
    static const char* env = dsw_diag_env("DSW_MIX_FIRST");   // "0": always basis-first (diagnostics / A-B)
    if (env && env[0] == '0') return false;
    return K >= 2 && 2 * Fout <= Fin;
}

extern "C" {

// a plan built against another version of include/dsw_hip.h (the struct grew): refuse it instead of reading past its end
#define DSW_PLAN_CHECK(p_) do { if ((p_) != nullptr && (p_)->struct_bytes != (int64_t)sizeof(dsw_hop2_plan)) return DSW_ERR_BAD_ARG; } while (0)
#define DSW_PLAN_OK(p_) ((p_) == nullptr || (p_)->struct_bytes == (int64_t)sizeof(dsw_hop2_plan))
int dsw_version(void) { return DSW_VERSION; }

int dsw_build_flags(void) { return g_build_flags.load(std::memory_order_relaxed); }

int dsw_trace_begin(int capacity) {
    if (capacity <= 0 || capacity > (1 << 20)) return DSW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_trace.mu);
    if (g_trace.on.load()) return DSW_ERR_BAD_ARG;          // one trace at a time
    g_trace.ev.assign((size_t)capacity * 2, nullptr);
    for (size_t i = 0; i < g_trace.ev.size(); ++i)
        if (hipEventCreate(&g_trace.ev[i]) != hipSuccess) {
            for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(g_trace.ev[j]);
            g_trace.ev.clear();
            return DSW_ERR_LAUNCH;
        }
    g_trace.rec.clear();
    g_trace.rec.reserve((size_t)capacity * 2);
    g_trace.n_ev = 0; g_trace.overflow = false;
    g_trace.on.store(true);
    return DSW_OK;
}

int dsw_trace_end(int32_t* call, int32_t* roles, int32_t* aux0, int32_t* aux1, int32_t* aux2, float* us, char* names,
                  int name_stride, int cap) {
    std::lock_guard<std::mutex> lk(g_trace.mu);
    if (!g_trace.on.load()) return DSW_ERR_BAD_ARG;
    g_trace.on.store(false);
    int out = 0, rc = DSW_OK, n_call = 0;
    // kernels between two markers belong to the role of the closing marker; kernels launched outside every entry point's
    // markers (none today) are dropped at the next start marker
    size_t first = 0;     // first kernel record of the open interval
    for (size_t i = 0; i < g_trace.rec.size() && rc == DSW_OK; ++i) {
        const DswTraceRec& r = g_trace.rec[i];
        if (r.role == -1) continue;
        if (r.role > 0) {
            bool any = false;
            for (size_t j = first; j < i && rc == DSW_OK; ++j) {
                const DswTraceRec& k = g_trace.rec[j];
                if (k.role != -1) continue;
                float ms = 0.f;
                if (hipEventSynchronize(k.e1) != hipSuccess || hipEventElapsedTime(&ms, k.e0, k.e1) != hipSuccess) { rc = DSW_ERR_LAUNCH; break; }
                if (out < cap && roles && us) {
                    if (call) call[out] = n_call;
                    roles[out] = r.role;
                    if (aux0) aux0[out] = r.a0;
                    if (aux1) aux1[out] = r.a1;
                    if (aux2) aux2[out] = r.a2;
                    us[out] = ms * 1000.f;
                    if (names && name_stride > 1) {
                        int c = 0;
                        for (; k.name && k.name[c] && c < name_stride - 1; ++c) names[(size_t)out * name_stride + c] = k.name[c];
                        names[(size_t)out * name_stride + c] = 0;
                    }
                }
                ++out;
                any = true;
            }
            if (any) ++n_call;
        }
        first = i + 1;
    }
    for (hipEvent_t e : g_trace.ev) (void)hipEventDestroy(e);
    g_trace.ev.clear(); g_trace.rec.clear();
    const bool ovf = g_trace.overflow;
    g_trace.n_ev = 0;
    if (rc != DSW_OK) { (void)hipGetLastError(); return rc; }
    return ovf ? DSW_ERR_WORKSPACE : out;
}

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
    trace_start(stream);
    const int rc = dsw_spmm_launch(rowptr, colind, vals, v_out, v_in, X, Y, B, C, alpha, Z, beta, Z2, gamma, dtype,
                                   (hipStream_t)stream);
    trace_mark(stream, DSW_ROLE_SPMM, v_out, v_in, C);
    return rc;
}

int dsw_spmm_csr_ld(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out, int64_t v_in,
                    int64_t nnz, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C, float alpha,
                    const void* Z, int64_t ldz, float beta, const void* Z2, float gamma, int dtype, dsw_stream_t stream) {
    if (v_out < 0 || v_in < 0 || nnz < 0 || B < 0 || C < 0 || ldx < C || ldy < C || (Z && ldz < C)) return DSW_ERR_BAD_ARG;
    if (v_out == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !X || !Y || (nnz > 0 && (!colind || !vals))) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    trace_start(stream);
    const int rc = dsw_spmm_launch_ld(rowptr, colind, vals, v_out, v_in, X, ldx, Y, ldy, B, C, alpha, Z, beta, Z2, gamma, dtype,
                                      (hipStream_t)stream, 0, Z ? ldz : C);
    trace_mark(stream, DSW_ROLE_SPMM, v_out, v_in, C);
    return rc;
}

int dsw_remap_csr(const dsw_remap_plan* plan, const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t v_out,
                  int64_t v_in, int64_t nnz, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t B, int64_t C,
                  const void* Z, int64_t ldz, float beta, int dtype, dsw_stream_t stream) {
    if (v_out < 0 || v_in < 0 || nnz < 0 || B < 0 || C < 0 || ldx < C || ldy < C || (Z && ldz < C)) return DSW_ERR_BAD_ARG;
    if (v_out == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !X || !Y || (nnz > 0 && (!colind || !vals))) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (plan != nullptr) {
        if (plan->kind < 0 || plan->kind > 2) return DSW_ERR_BAD_ARG;
        // a regular plan certifies the STRUCTURE of this very matrix: shapes that cannot have it are refused, not reinterpreted
        if (plan->kind == 1 && (plan->m < 1 || v_in != v_out * plan->m || nnz != v_in)) return DSW_ERR_BAD_ARG;
        if (plan->kind == 2 && (plan->m < 1 || v_out != v_in * plan->m || nnz != v_out)) return DSW_ERR_BAD_ARG;
        if (plan->kind == 0 && plan->n_long > 0 && (!plan->long_rows || plan->long_thr <= 0)) return DSW_ERR_BAD_ARG;
    }
    trace_start(stream);
    const int rc = dsw_remap_launch(plan, rowptr, colind, vals, v_out, v_in, X, ldx, Y, ldy, B, C, Z, Z ? ldz : C, beta, dtype,
                                    (hipStream_t)stream);
    trace_mark(stream, DSW_ROLE_SPMM, v_out, v_in, C);
    return rc;
}

int dsw_spmm2_fused(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z1, const void* Z1b,
                    const void* Z2, void* Y1, void* Y2, int64_t B, int64_t C, float a1, float b1, float d1,
                    float a2, float b2, float c2, int dtype, dsw_stream_t stream) {
    DSW_PLAN_CHECK(plan);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (V < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    trace_start(stream);
    const int rc = dsw_spmm2_launch(plan, V, U, Z1, Z1b, Z2, Y1, Y2, B, C, a1, b1, d1, a2, b2, c2, dtype,
                                    (hipStream_t)stream);
    trace_mark(stream, DSW_ROLE_SPMM2, V, V, C);
    return rc;
}

int dsw_spmm_staged(const dsw_hop2_plan* plan, int64_t V, const void* U, const void* Z, const void* Z2, void* Y,
                    int64_t B, int64_t C, float a, float b, float c, int dtype, dsw_stream_t stream, int stream_out) {
    DSW_PLAN_CHECK(plan);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (V < 0 || B < 0 || C < 0) return DSW_ERR_BAD_ARG;
    trace_start(stream);
    const int rc = dsw_spmm1s_launch(plan, V, U, Z, Z2, Y, B, C, a, b, c, dtype, (hipStream_t)stream, stream_out);
    trace_mark(stream, DSW_ROLE_SPMM_STAGED, V, V, C);
    return rc;
}

int dsw_spmm_staged_supported(const dsw_hop2_plan* plan, int64_t C, int dtype) {
    return DSW_PLAN_OK(plan) ? dsw_spmm1s_supported(plan, C, dtype) : 0;
}

static int cheb_basis_fwd_impl(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V,
                               int64_t nnz, const void* X, void* T, int64_t B, int64_t C, int64_t K, int dtype,
                               dsw_stream_t stream, const dsw_hop2_plan* plan) {
    if (K <= 1) return K == 1 ? DSW_OK : DSW_ERR_BAD_ARG;
    if (V < 0 || B < 0 || C < 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (V == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr || !X || !T) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t plane = B * V * C * elem_size(dtype);
    char* t = static_cast<char*>(T);
    hipStream_t s = (hipStream_t)stream;
    auto Tk = [&](int64_t k) -> const void* { return k == 0 ? X : static_cast<const void*>(t + (k - 1) * plane); };
    // hop pairs run fused whenever a supported plan is given (DSW_HOP2_FWD=0 forces one launch per hop)
    static const char* fwd_env = dsw_diag_env("DSW_HOP2_FWD");
    const bool fused = plan != nullptr && !(fwd_env != nullptr && fwd_env[0] == '0') && dsw_spmm2_supported(plan, C, dtype);
    const bool staged = plan != nullptr && dsw_spmm1s_supported(plan, C, dtype);   // dense stencils: one staged launch per hop
    int rc = DSW_OK;
    int64_t k = 1;   // next basis index to produce
    while (k < K && rc == DSW_OK) {
        if (staged) {
            // the plane is gathered by the next hop (cached stores) unless it is the last one
            rc = dsw_spmm1s_launch(plan, V, Tk(k - 1), k > 1 ? Tk(k - 2) : nullptr, nullptr, t + (k - 1) * plane, B, C,
                                   k == 1 ? 1.f : 2.f, -1.f, 0.f, dtype, s, k + 1 < K ? 0 : 1);
            k += 1;
        } else if (fused && k + 1 < K) {
            // T_k = c L T_{k-1} - [k>1] T_{k-2};  T_{k+1} = 2 L T_k - T_{k-1}   (c = 1 for k = 1, else 2)
            rc = dsw_spmm2_launch(plan, V, Tk(k - 1), k > 1 ? Tk(k - 2) : nullptr, nullptr, nullptr,
                                  t + (k - 1) * plane, t + k * plane, B, C, k == 1 ? 1.f : 2.f, -1.f, 0.f, 2.f,
                                  -1.f, 0.f, dtype, s);
            k += 2;
        } else {
            rc = dsw_spmm_launch(rowptr, colind, vals, V, V, Tk(k - 1), t + (k - 1) * plane, B, C,
                                 k == 1 ? 1.f : 2.f, k > 1 ? Tk(k - 2) : nullptr, -1.f, nullptr, 0.f, dtype, s);
            k += 1;
        }
    }
    return rc;
}

// folded != 0: the caller produced plane K-3 as G_{K-3} - G_{K-1} (dsw_fold_w_launch), so the step that computes
// G'_{K-3} does not subtract plane K-1 again
static int cheb_basis_adj_impl(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                               int64_t nnz, void* G0, void* Grest, int64_t B, int64_t C, int64_t K, int dtype,
                               dsw_stream_t stream, const dsw_hop2_plan* plan_t, void* spare, int folded) {
    if (K <= 1) return K == 1 ? DSW_OK : DSW_ERR_BAD_ARG;
    if (V < 0 || B < 0 || C < 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (V == 0 || B == 0 || C == 0) return DSW_OK;
    if (!rowptr_t || !G0 || !Grest) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t plane = B * V * C * elem_size(dtype);
    char* G = static_cast<char*>(Grest);
    hipStream_t s = (hipStream_t)stream;
    const bool fused = plan_t != nullptr && dsw_spmm2_supported(plan_t, C, dtype) && (K < 4 || spare != nullptr);
    // G'_{j-1} = G_{j-1} + c_j L^T G'_j - G'_{j+1}   (c_j = 2 for j >= 2, 1 for j = 1; G'_K = 0)
    // loc[j]: where G'_j currently lives (its own plane, or one of the two spare planes)
    auto own = [&](int64_t j) -> char* { return j == 0 ? static_cast<char*>(G0) : G + (j - 1) * plane; };
    char* loc_j = own(K - 1);        // G'_{K-1} = G_{K-1}
    char* loc_jp1 = nullptr;         // G'_{K}   = 0
    int spare_sel = 0;
    int rc = DSW_OK;
    int64_t j = K - 1;
    const bool staged = plan_t != nullptr && dsw_spmm1s_supported(plan_t, C, dtype);
    while (j >= 1 && rc == DSW_OK) {
        if (staged) {
            // single step on staged neighbourhoods: G'_{j-1} = c_j L^T G'_j + G_{j-1} - G'_{j+1}, in place on plane j-1
            char* gm1 = own(j - 1);
            rc = dsw_spmm1s_launch(plan_t, V, loc_j, gm1, (folded && j == K - 2) ? nullptr : loc_jp1, gm1, B, C,
                                   (j == 1) ? 1.f : 2.f, 1.f, -1.f, dtype, s, j > 1 ? 0 : 1);
            loc_jp1 = loc_j;
            loc_j = gm1;
            j -= 1;
        } else if (fused && j >= 2) {
            // pair (j, j-1):  Y1 = G'_{j-1} = 2 L^T G'_j + G_{j-1} - G'_{j+1}
            //                 Y2 = G'_{j-2} = c L^T Y1 + G_{j-2} - G'_j        (c = 1 if j-1 == 1 else 2)
            const bool last = (j - 2 == 0);
            // Y1 is needed later only if another step follows (as its G'_{j+1}); it cannot be written over
            // G_{j-1} in place (neighbouring tiles read that plane as Z1), so it goes to a spare plane.
            char* y1 = last ? nullptr : static_cast<char*>(spare) + (size_t)spare_sel * plane;
            char* y2 = own(j - 2);   // in place: Z2 = G_{j-2} is read on the writer's own rows only
            rc = dsw_spmm2_launch(plan_t, V, loc_j, own(j - 1), loc_jp1, own(j - 2), y1, y2, B, C, 2.f, 1.f, -1.f,
                                  (j - 1 == 1) ? 1.f : 2.f, (folded && j == K - 1) ? 0.f : -1.f, 1.f, dtype, s);
            loc_jp1 = y1;
            loc_j = y2;
            spare_sel ^= 1;
            j -= 2;
        } else {
            // single step: G'_{j-1} = c_j L^T G'_j + G_{j-1} - G'_{j+1}, in place on plane j-1
            char* gm1 = own(j - 1);
            rc = dsw_spmm_launch(rowptr_t, colind_t, vals_t, V, V, loc_j, gm1, B, C, (j == 1) ? 1.f : 2.f, gm1, 1.f,
                                 (folded && j == K - 2) ? nullptr : loc_jp1, -1.f, dtype, s, DSW_SPMM_HINT_COLD_Z);
            loc_jp1 = loc_j;
            loc_j = gm1;
            j -= 1;
        }
    }
    return rc;
}

int dsw_cheb_basis_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V,
                       int64_t nnz, const void* X, void* T, int64_t B, int64_t C, int64_t K, int dtype,
                       dsw_stream_t stream, const dsw_hop2_plan* plan) {
    DSW_PLAN_CHECK(plan);
    trace_start(stream);
    const int rc = cheb_basis_fwd_impl(rowptr, colind, vals, V, nnz, X, T, B, C, K, dtype, stream, plan);
    trace_mark(stream, DSW_ROLE_BASIS_FWD, V, C, K);
    return rc;
}

int dsw_cheb_basis_adj(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                       int64_t nnz, void* G0, void* Grest, int64_t B, int64_t C, int64_t K, int dtype,
                       dsw_stream_t stream, const dsw_hop2_plan* plan_t, void* spare) {
    DSW_PLAN_CHECK(plan_t);
    trace_start(stream);
    const int rc = cheb_basis_adj_impl(rowptr_t, colind_t, vals_t, V, nnz, G0, Grest, B, C, K, dtype, stream, plan_t, spare, 0);
    trace_mark(stream, DSW_ROLE_BASIS_ADJ, V, C, K);
    return rc;
}

int dsw_cheb_mix_fwd(const void* X, const void* T, const void* W, const void* bias, void* Y, int64_t N,
                     int64_t Fin, int64_t Fout, int64_t K, int dtype, dsw_stream_t stream) {
    if (N < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (N == 0) return DSW_OK;
    if (!X || !W || !Y || (K > 1 && !T)) return DSW_ERR_BAD_ARG;
    trace_start(stream);
    const int rc = dsw_mix_fwd_launch(X, T, W, bias, Y, N, Fin, Fout, K, dtype, (hipStream_t)stream);
    trace_mark(stream, DSW_ROLE_MIX_FWD, N, Fin, Fout);
    return rc;
}

int dsw_cheb_mix_first(int64_t Fin, int64_t Fout, int64_t K) { return mix_first(Fin, Fout, K) ? 1 : 0; }

int dsw_cheb_bwd_needs_basis(const dsw_hop2_plan* plan_t, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    DSW_PLAN_CHECK(plan_t);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (K == 1 || mix_first(Fin, Fout, K)) return 0;
    return dsw_cheb3_bwd_dual_eligible(plan_t, V, Fin, Fout, K, dtype) ? 0 : 1;
}

int dsw_cheb_fwd_path(const dsw_hop2_plan* plan, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    DSW_PLAN_CHECK(plan);
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (mix_first(Fin, Fout, K)) return DSW_FWD_MIX_FIRST;
    if (dsw_cheb3_fwd_fused_eligible(plan, Fin, Fout, K, dtype)) return DSW_FWD_ONE_LAUNCH;
    if (dsw_cheb3_hop2mix_eligible(plan, Fin, Fout, K, dtype)) return DSW_FWD_HOP1_THEN_ONE_LAUNCH;   // (given the basis buffer T)
    if (K > 1 && plan != nullptr && dsw_spmm1s_supported(plan, Fin, dtype)) return DSW_FWD_STAGED_HOPS;
    if (K > 2 && plan != nullptr && dsw_spmm2_supported(plan, Fin, dtype)) return DSW_FWD_FUSED_PAIRS;
    return DSW_FWD_PLAIN_HOPS;
}

static int cheb_fwd_impl(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                         const void* X, const void* W, const void* bias, void* Y, void* T, int64_t B, int64_t Fin,
                         int64_t Fout, int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act,
                         const void* scale, const void* R, int64_t ldr, int64_t ldy, void* workspace = nullptr,
                         int64_t workspace_bytes = 0) {
    if (K <= 0 || (act != DSW_ACT_NONE && act != DSW_ACT_RELU)) return DSW_ERR_BAD_ARG;
    if (workspace != nullptr && (Fin <= 0 || Fout <= 0 || workspace_bytes < w_image_bytes(Fin, Fout, K, dtype))) return DSW_ERR_WORKSPACE;
    if (R != nullptr && ldr < Fout) return DSW_ERR_BAD_ARG;
    if (ldy != 0 && ldy < Fout) return DSW_ERR_BAD_ARG;
    const int relu = act == DSW_ACT_RELU;
    const bool extras = scale != nullptr || R != nullptr || (ldy != 0 && ldy != Fout);
    // workspace (optional, dsw_cheb_fwd_workspace_bytes): scratch for a per-call image of the weights - the streaming-W GEMM
    // of wide fp32 layers splits W once per call into it
    char* wsa = workspace ? reinterpret_cast<char*>(round_up((int64_t)(uintptr_t)workspace, 256)) : nullptr;
    const int64_t wsb = workspace ? workspace_bytes - (wsa - static_cast<char*>(workspace)) : 0;
    DswEpiExtra ex = {scale, R, ldr, (ldy != 0 && ldy != Fout) ? ldy : 0, wsa, wsb, 0};
    const bool use_ex = extras || wsa != nullptr;
    int rc = DSW_OK;
    trace_start(stream);
    if (mix_first(Fin, Fout, K)) {
        if (ldy != 0 && ldy != Fout) return DSW_ERR_BAD_ARG;   // the recurrence on the output planes runs on dense [N, Fout]
        // T is scratch here: (K-1) planes of [N, Fin] hold the K-1 (+2 spare for K >= 4) planes of [N, Fout]
        if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0) return DSW_ERR_BAD_ARG;
        if (B == 0 || V == 0) return DSW_OK;
        if (!X || !W || !Y || !T || !rowptr) return DSW_ERR_BAD_ARG;
        if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
        const int64_t N = B * V;
        // scale and residual ride in the epilogue of the plane GEMM: Y = sum_k T_k(L) (s X W_k) + (s b + R) is linear in
        // the planes, so Z_k = s (X W_k) (+ s b + R on plane 0) and the recurrence is unchanged
        char* spare = static_cast<char*>(T) + (K - 1) * N * Fout * elem_size(dtype);
        // plane K-3 is produced as Z_{K-3} - Z_{K-1} (folded weights, parked behind the planes in the T scratch when it has
        // the room - it does unless the batch is a handful of nodes): one epilogue operand less at the top of the recurrence
        const void* Wz = W;
        int folded = 0;
        // (only where it saves a pass: the staged one-hop kernels read that plane from HBM; inside a fused pair it is the
        // staged input of the first hop anyway, and the fold would only add a launch)
        if (K >= 3 && plan != nullptr && plan->hops == 1 && wsa != nullptr) {
            ex.fold = 1;        // the plane GEMM folds (while it splits W, or through a folded copy in the workspace)
            folded = 1;
        } else if (K >= 3 && plan != nullptr && plan->hops == 1) {
            const int64_t used = round_up((K - 1 + (K >= 4 ? 2 : 0)) * N * Fout * elem_size(dtype), 256);
            const int64_t have = (K - 1) * N * Fin * elem_size(dtype);
            if (have - used >= Fin * K * Fout * elem_size(dtype)) {
                void* Wf = static_cast<char*>(T) + used;
                rc = dsw_fold_w_launch(W, Wf, Fin, Fout, K, dtype, (hipStream_t)stream);
                if (rc != DSW_OK) return rc;
                Wz = Wf;
                folded = 1;
            }
        }
        rc = dsw_zmix_launch(X, Wz, bias, Y, T, N, Fin, Fout, K, dtype, (hipStream_t)stream, use_ex ? &ex : nullptr);
        trace_mark(stream, DSW_ROLE_ZMIX, V, Fin, Fout);
        if (rc != DSW_OK) return rc;
        // the Clenshaw recurrence has exactly the form of the adjoint recurrence (with L instead of L^T)
        rc = cheb_basis_adj_impl(rowptr, colind, vals, V, nnz, Y, T, B, Fout, K, dtype, stream, plan,
                                 K >= 4 ? spare : nullptr, folded);
        trace_mark(stream, DSW_ROLE_CLENSHAW_FWD, V, Fout, K);
        // the mix-first order ends in an SpMM: the activation is one in-place pass over the (Fout-channel) output
        if (rc == DSW_OK && relu) {
            rc = dsw_relu_inplace_launch(Y, N * Fout, dtype, (hipStream_t)stream);
            trace_mark(stream, DSW_ROLE_ELEMENTWISE, V, Fout, 0);
        }
        return rc;
    }
    if (K == 3 && !extras && X && W && Y && rowptr && B >= 0 && V >= 0) {
        // K = 3, 32 input channels, fp32: both hops AND the channel mix in one launch (dsw_fwd3.hip)
        int rcf = DSW_OK;
        if (dsw_cheb3_fwd_fused_try(plan, V, X, W, bias, Y, T, B, Fin, Fout, K, dtype, (hipStream_t)stream, &rcf, relu)) {
            trace_mark(stream, DSW_ROLE_FWD_ONE_LAUNCH, V, Fin, Fout);
            return rcf;
        }
        // ... on a one-hop plan (dense stencils): staged hop 1, then hop 2 + channel mix in one launch
        struct Hop1Ctx { dsw_stream_t s; int64_t V, Fin; } hc = {stream, V, Fin};
        auto hop1_done = [](void* c) { auto* h = static_cast<Hop1Ctx*>(c); trace_mark(h->s, DSW_ROLE_BASIS_FWD, h->V, h->Fin, 2); };
        if (dsw_cheb3_hop2mix_try(plan, V, X, W, bias, Y, T, B, Fin, Fout, K, dtype, (hipStream_t)stream, &rcf, relu, hop1_done, &hc)) {
            trace_mark(stream, DSW_ROLE_FWD_HOP2_MIX, V, Fin, Fout);
            return rcf;
        }
    }
    if (K > 1) {
        rc = cheb_basis_fwd_impl(rowptr, colind, vals, V, nnz, X, T, B, Fin, K, dtype, stream, plan);
        trace_mark(stream, DSW_ROLE_BASIS_FWD, V, Fin, K);
        if (rc != DSW_OK) return rc;
    }
    if (B * V < 0 || Fin <= 0 || Fout <= 0) return DSW_ERR_BAD_ARG;
    if (B * V == 0) return DSW_OK;
    if (!X || !W || !Y || (K > 1 && !T)) return DSW_ERR_BAD_ARG;
    rc = dsw_mix_fwd_launch(X, T, W, bias, Y, B * V, Fin, Fout, K, dtype, (hipStream_t)stream, relu, use_ex ? &ex : nullptr);
    trace_mark(stream, DSW_ROLE_MIX_FWD, B * V, K * Fin, Fout);
    return rc;
}

int64_t dsw_cheb_fwd_workspace_bytes(int64_t B, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    return w_image_bytes(Fin, Fout, K, dtype) + 256;
}

int dsw_cheb_fwd_ws(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                    const void* X, const void* W, const void* bias, void* Y, int64_t ldy, void* T, int64_t B, int64_t Fin,
                    int64_t Fout, int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act,
                    const void* scale, const void* R, int64_t ldr, void* workspace, int64_t workspace_bytes) {
    DSW_PLAN_CHECK(plan);
    return cheb_fwd_impl(rowptr, colind, vals, V, nnz, X, W, bias, Y, T, B, Fin, Fout, K, dtype, stream, plan, act,
                         scale, R, ldr, ldy, workspace, workspace_bytes);
}

int dsw_cheb_fwd_act(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                     const void* X, const void* W, const void* bias, void* Y, void* T, int64_t B, int64_t Fin,
                     int64_t Fout, int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act) {
    DSW_PLAN_CHECK(plan);
    return cheb_fwd_impl(rowptr, colind, vals, V, nnz, X, W, bias, Y, T, B, Fin, Fout, K, dtype, stream, plan, act,
                         nullptr, nullptr, 0, 0);
}

int dsw_cheb_fwd_res(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                     const void* X, const void* W, const void* bias, void* Y, int64_t ldy, void* T, int64_t B, int64_t Fin,
                     int64_t Fout, int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan, int act,
                     const void* scale, const void* R, int64_t ldr) {
    DSW_PLAN_CHECK(plan);
    return cheb_fwd_impl(rowptr, colind, vals, V, nnz, X, W, bias, Y, T, B, Fin, Fout, K, dtype, stream, plan, act,
                         scale, R, ldr, ldy);
}

int dsw_cheb_fwd(const int32_t* rowptr, const int32_t* colind, const float* vals, int64_t V, int64_t nnz,
                 const void* X, const void* W, const void* bias, void* Y, void* T, int64_t B, int64_t Fin,
                 int64_t Fout, int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan) {
    return dsw_cheb_fwd_act(rowptr, colind, vals, V, nnz, X, W, bias, Y, T, B, Fin, Fout, K, dtype, stream, plan,
                            DSW_ACT_NONE);
}

int64_t dsw_cheb_bwd_workspace_bytes(int64_t B, int64_t V, int64_t Fin, int64_t Fout, int64_t K, int dtype) {
    if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0 || K <= 0) return DSW_ERR_BAD_ARG;
    const int64_t N = B * V;
    if (mix_first(Fin, Fout, K)) {   // D_1..D_{K-1} planes of [N, Fout] + the partials of one K = 1 wgrad launch
        const int64_t d = round_up((K - 1) * N * Fout * elem_size(dtype), 256);
        const int64_t S1 = dsw_wgrad_slabs(N, Fin, Fout, 1), SK = dsw_wgrad_slabs(N, Fin, Fout, K);
        const int64_t p1 = (S1 > 0 ? S1 : 1) * (Fin + 1) * Fout * 4;        // one K = 1 launch per order (fallback)
        const int64_t pk = (SK > 0 ? SK : 1) * (K * Fin + 1) * Fout * 4;    // single launch over all orders
        int64_t pm = p1 > pk ? p1 : pk;
        if (K * Fout <= 64) {   // narrow output: the K planes side by side, one K = 1 launch of width K * Fout
            const int64_t Sn = dsw_wgrad_slabs(N, Fin, K * Fout, 1);
            const int64_t pn = (Sn > 0 ? Sn : 1) * (Fin + 1) * K * Fout * 4;
            if (pn > pm) pm = pn;
        }
        return d + round_up(pm, 256) + round_up(w_image_bytes(Fin, Fout, K, dtype), 256) + 256;   // + the weight image / scratch of the dX GEMM
    }
    // G_1..G_{K-1} planes, plus two spare planes for the pairwise fused adjoint when K >= 4
    const int64_t g = round_up((K - 1 + (K >= 4 ? 2 : 0)) * N * Fin * elem_size(dtype), 256);
    const int64_t S = dsw_wgrad_slabs(N, Fin, Fout, K);
    const int64_t p = round_up((S > 0 ? S : 1) * (K * Fin + 1) * Fout * 4, 256);
    const int64_t wf = round_up(w_image_bytes(Fin, Fout, K, dtype), 256);   // pre-split / folded image of the weights for the dgrad GEMM
    int64_t total = g + p + wf + 256;
    // the one-launch dual backward (dsw_bwd3d.hip) parks one slab of dW / db partials per persistent workgroup at the base
    if (dtype == DSW_F32 && K == 3 && Fin == 32 && Fout == 64 && V % 64 == 0) {
        const int64_t d = round_up(dsw_cheb3_bwd_dual_ws_bytes(), 256) + 256;
        if (d > total) total = d;
    }
    return total;
}

static int cheb_bwd_impl(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                 int64_t nnz, const void* X, const void* T, const void* W, const void* dY, void* dX, void* dW,
                 void* db, void* workspace, int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout,
                 int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan_t, const void* scale,
                 const void* dX_add, int64_t ld_add, int accumulate) {
    if (dX_add != nullptr && ld_add < Fin) return DSW_ERR_BAD_ARG;
    const bool extras = scale != nullptr || dX_add != nullptr;
    const DswEpiExtra ex = {scale, dX_add, ld_add, 0, nullptr, 0, 0};
    if (B < 0 || V < 0 || Fin <= 0 || Fout <= 0 || K <= 0 || nnz < 0) return DSW_ERR_BAD_ARG;
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    const int64_t need = dsw_cheb_bwd_workspace_bytes(B, V, Fin, Fout, K, dtype);
    if (!workspace || workspace_bytes < need) return DSW_ERR_WORKSPACE;
    const bool mf = mix_first(Fin, Fout, K);
    const int64_t N = B * V;
    hipStream_t s = (hipStream_t)stream;
    if (N > 0 && (!dY || !X || !W)) return DSW_ERR_BAD_ARG;
    if (N == 0) {   // an empty batch shard (B < world size): the parameter gradients are exactly zero, not "unwritten"
        if (dW && hipMemsetAsync(dW, 0, (size_t)(Fin * K * Fout * elem_size(dtype)), s) != hipSuccess) return DSW_ERR_LAUNCH;
        if (db && hipMemsetAsync(db, 0, (size_t)(Fout * elem_size(dtype)), s) != hipSuccess) return DSW_ERR_LAUNCH;
        return DSW_OK;
    }
    // carve the workspace (256-byte aligned base)
    char* ws = reinterpret_cast<char*>(round_up((int64_t)(uintptr_t)workspace, 256));
    trace_start(stream);
    if (mf) {
        if (N == 0 || (!dX && !dW)) return DSW_OK;
        if (!rowptr_t) return DSW_ERR_BAD_ARG;
        const int64_t dplane = N * Fout * elem_size(dtype);
        char* D = ws;                                                     // D_1 .. D_{K-1}
        float* part = reinterpret_cast<float*>(ws + round_up((K - 1) * dplane, 256));
        // scratch of the dX GEMM (per-call weight image, balanced decomposition) behind everything else: need - wf from the base
        DswEpiExtra exw = ex;
        exw.ws = ws + (need - 256 - round_up(w_image_bytes(Fin, Fout, K, dtype), 256));
        exw.ws_bytes = w_image_bytes(Fin, Fout, K, dtype);
        int rcm = cheb_basis_fwd_impl(rowptr_t, colind_t, vals_t, V, nnz, dY, D, B, Fout, K, dtype, stream, plan_t);
        trace_mark(stream, DSW_ROLE_BASIS_DUAL, V, Fout, K);     // the dual: Chebyshev basis of dY under L^T
        if (rcm == DSW_OK && dX != nullptr) {
            rcm = dsw_zdgrad_launch(dY, D, W, dX, N, Fin, Fout, K, dtype, s, &exw);
            trace_mark(stream, DSW_ROLE_BWD_DGRAD, V, Fin, Fout);
        }
        if (rcm == DSW_OK && dW != nullptr) {
            rcm = dsw_wgrad_mixfirst_launch(X, dY, D, dW, db, part, N, Fin, Fout, K, dtype, s, accumulate);
            trace_mark(stream, DSW_ROLE_BWD_WGRAD, V, Fin, Fout);
        }
        return rcm;
    }
    if (!extras && (dX != nullptr || dW != nullptr)) {
        // K = 3, 32 -> 64, fp32, two-hop plan of L^T: the whole backward in ONE launch in the dual form (dsw_bwd3d.hip) - the
        // Chebyshev basis of dY under L^T on chip, dX = sum_k U_k W_k^T and dW_k = X^T U_k from it: neither T nor dgrad planes
        int rcd = DSW_OK;
        if (dsw_cheb3_bwd_dual_try(plan_t, V, X, dY, W, dX, dW, db, reinterpret_cast<float*>(ws), B, Fin, Fout, K, dtype, s, &rcd,
                                   accumulate)) {
            trace_mark(stream, DSW_ROLE_BWD_DUAL, V, Fin, Fout);
            return rcd;
        }
    }
    // every other route reads the forward's basis planes.  T == NULL is the contract of the dual form only (include/dsw_hip.h:
    // dsw_cheb_bwd_needs_basis() == 0 AND 16-byte aligned X, dY, W, dX, workspace): say which half of it was broken
    if (K > 1 && !T)
        return (!extras && dsw_cheb_bwd_needs_basis(plan_t, V, Fin, Fout, K, dtype) == 0) ? DSW_ERR_ALIGN : DSW_ERR_BAD_ARG;
    const int64_t plane = N * Fin * elem_size(dtype);
    char* G = ws;                                                    // G_1 .. G_{K-1}
    char* spare = G + (K - 1) * plane;
    float* partial = reinterpret_cast<float*>(ws + round_up((K - 1 + (K >= 4 ? 2 : 0)) * plane, 256));
    int rc = DSW_OK;
    // dgrad weights with plane K-3 folded (G_{K-3} - G_{K-1} out of the GEMM): the adjoint recurrence then has one
    // epilogue operand less at its top - for K = 3 every step is a one-operand step
    // (staged one-hop plans only: inside a fused pair the subtracted plane is the staged input of the first hop - no pass
    // is saved there and the fold would only add work)
    const int folded = (K >= 3 && dX != nullptr && N > 0 && plan_t != nullptr && plan_t->hops == 1) ? 1 : 0;
    if (dX != nullptr && dW != nullptr && N > 0 && !extras) {
        // small aligned fp32 layers: dgrad planes and dW partials from ONE pass over dY (dsw_wgrad_x3.hip, FUSE); the fold is
        // applied while that kernel fills its W^T panel
        if (K > 1 && !rowptr_t) return DSW_ERR_BAD_ARG;
        int rcf = DSW_OK;
        if (dsw_bwd_gemm_fused_try(X, T, W, dY, dW, db, dX, G, partial, N, Fin, Fout, K, dtype, s, &rcf, accumulate, folded)) {
            trace_mark(stream, DSW_ROLE_BWD_GEMM_FUSED, V, Fin, Fout);
            if (rcf != DSW_OK) return rcf;
            if (K > 1) {
                rcf = cheb_basis_adj_impl(rowptr_t, colind_t, vals_t, V, nnz, dX, G, B, Fin, K, dtype, stream, plan_t,
                                          K >= 4 ? spare : nullptr, folded);
                trace_mark(stream, DSW_ROLE_BASIS_ADJ, V, Fin, K);
            }
            return rcf;
        }
    }
    if (dX != nullptr && N > 0) {
        if (K > 1 && !rowptr_t) return DSW_ERR_BAD_ARG;
        // scale multiplies every dgrad plane (the recurrence is linear); dX_add joins plane 0 = the dX buffer, onto which
        // the adjoint recurrence then accumulates.  The GEMM gets scratch for a per-call image of the weights (the streaming
        // kernel splits W once per call into it, folding plane K-3 on the way; other kernels take a folded copy)
        const int64_t S_ = dsw_wgrad_slabs(N, Fin, Fout, K);
        DswEpiExtra exw = ex;
        exw.ws = reinterpret_cast<char*>(partial) + round_up((S_ > 0 ? S_ : 1) * (K * Fin + 1) * Fout * 4, 256);
        exw.ws_bytes = w_image_bytes(Fin, Fout, K, dtype);
        exw.fold = folded;
        rc = dsw_mix_dgrad_launch(dY, W, dX, G, N, Fin, Fout, K, dtype, s, &exw);
        trace_mark(stream, DSW_ROLE_BWD_DGRAD, V, Fin, Fout);
        if (rc == DSW_OK && K > 1) {
            rc = cheb_basis_adj_impl(rowptr_t, colind_t, vals_t, V, nnz, dX, G, B, Fin, K, dtype, stream, plan_t,
                                     K >= 4 ? spare : nullptr, folded);
            trace_mark(stream, DSW_ROLE_BASIS_ADJ, V, Fin, K);
        }
        if (rc != DSW_OK) return rc;
    }
    if (dW != nullptr) {
        rc = dsw_wgrad_launch(X, T, dY, dW, db, partial, N, Fin, Fout, K, dtype, s, accumulate);
        trace_mark(stream, DSW_ROLE_BWD_WGRAD, V, Fin, Fout);
    }
    return rc;
}

int dsw_cheb_bwd(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                 int64_t nnz, const void* X, const void* T, const void* W, const void* dY, void* dX, void* dW,
                 void* db, void* workspace, int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout,
                 int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan_t) {
    DSW_PLAN_CHECK(plan_t);
    return cheb_bwd_impl(rowptr_t, colind_t, vals_t, V, nnz, X, T, W, dY, dX, dW, db, workspace, workspace_bytes, B, Fin,
                         Fout, K, dtype, stream, plan_t, nullptr, nullptr, 0, 0);
}

int dsw_cheb_bwd_res(const int32_t* rowptr_t, const int32_t* colind_t, const float* vals_t, int64_t V,
                     int64_t nnz, const void* X, const void* T, const void* W, const void* dY, void* dX, void* dW,
                     void* db, void* workspace, int64_t workspace_bytes, int64_t B, int64_t Fin, int64_t Fout,
                     int64_t K, int dtype, dsw_stream_t stream, const dsw_hop2_plan* plan_t, const void* scale,
                     const void* dX_add, int64_t ld_add, int accumulate_dw) {
    DSW_PLAN_CHECK(plan_t);
    if (accumulate_dw && B * V == 0) return DSW_OK;        // an empty shard adds nothing
    return cheb_bwd_impl(rowptr_t, colind_t, vals_t, V, nnz, X, T, W, dY, dX, dW, db, workspace, workspace_bytes, B,
                         Fin, Fout, K, dtype, stream, plan_t, scale, dX_add, ld_add, accumulate_dw ? 1 : 0);
}

int64_t dsw_rezero_param_grads_workspace_bytes(void) { return dsw_rezero_param_grads_ws_bytes_impl(); }

int dsw_rezero_param_grads(const void* W, const void* bias, const void* dW_raw, const void* db_raw, const void* scale,
                           void* dW, void* db, void* dscale, int64_t n_w, int64_t n_b, void* workspace,
                           int64_t workspace_bytes, int dtype, dsw_stream_t stream) {
    if (dtype != DSW_F32 && dtype != DSW_BF16) return DSW_ERR_BAD_DTYPE;
    if (n_w < 0 || n_b < 0 || !W || !dW_raw || !scale || !dW || !dscale || (n_b > 0 && (!bias || !db_raw || !db)))
        return DSW_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < dsw_rezero_param_grads_ws_bytes_impl()) return DSW_ERR_WORKSPACE;
    DswTraceScope trace_((hipStream_t)stream, DSW_ROLE_ELEMENTWISE, n_w + n_b, 4, 0);
    return dsw_rezero_param_grads_launch(W, bias, dW_raw, db_raw, scale, dW, db, dscale, n_w, n_b, workspace, dtype,
                                         (hipStream_t)stream);
}

}  // extern "C"
