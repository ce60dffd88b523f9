// A client of libdsw_hip.so that is NOT Python: plain HIP runtime + the C ABI of include/dsw_hip.h, checked against
// the plain-C oracle (oracle/cheb_oracle.c, linked in as the checker).  Shows the drop-in boundary for what it is:
// extern "C", device pointers, sizes, a stream - no torch types anywhere.  Built and run by
// tests/test_hip_parity.py::test_c_abi_from_a_c_client (GPU box only).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dsw_hip.h"

extern "C" {
int oracle_cheb_forward(const int32_t*, const int32_t*, const float*, int64_t, const double*, const double*, const double*,
                        double*, double*, int64_t, int64_t, int64_t, int64_t);
int oracle_cheb_backward(const int32_t*, const int32_t*, const float*, int64_t, const double*, const double*, const double*,
                         double*, double*, double*, int64_t, int64_t, int64_t, int64_t);
}

#define HIP_OK(x)                                                                    \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)

static uint32_t rng_state = 12345u;
static double urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffffff) / double(1 << 24) - 0.5; }

template <typename T>
static T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    HIP_OK(hipMalloc(&d, h.size() * sizeof(T) + 16));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static double max_rel(const std::vector<float>& a, const std::vector<double>& ref) {
    double m = 0, e = 0;
    for (size_t i = 0; i < ref.size(); ++i) { m = std::fmax(m, std::fabs(ref[i])); e = std::fmax(e, std::fabs(a[i] - ref[i])); }
    return e / (m > 0 ? m : 1);
}

// non-symmetric banded-plus-random operator: row r holds r-2..r+2 (cyclic) and one far column
static void make_operator(int V, std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<float>& va) {
    rp.assign(V + 1, 0);
    for (int r = 0; r < V; ++r) {
        int cols[6] = {(r + V - 2) % V, (r + V - 1) % V, r, (r + 1) % V, (r + 2) % V, (int)((r * 7919L + 13) % V)};
        std::vector<int> c(cols, cols + 6);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        for (int col : c) { ci.push_back(col); va.push_back((float)(0.3 * urand())); }
        rp[r + 1] = (int32_t)ci.size();
    }
}

static void transpose_csr(int V, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci, const std::vector<float>& va,
                          std::vector<int32_t>& trp, std::vector<int32_t>& tci, std::vector<float>& tva) {
    trp.assign(V + 1, 0);
    for (int32_t c : ci) trp[c + 1]++;
    for (int i = 0; i < V; ++i) trp[i + 1] += trp[i];
    tci.resize(ci.size()); tva.resize(ci.size());
    std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
    for (int r = 0; r < V; ++r)
        for (int p = rp[r]; p < rp[r + 1]; ++p) { int q = fill[ci[p]]++; tci[q] = r; tva[q] = va[p]; }
}

static int run_case(int V, int B, int Fin, int Fout, int K) {
    const int64_t N = (int64_t)B * V;
    std::vector<int32_t> rp, ci, trp, tci;
    std::vector<float> va, tva;
    make_operator(V, rp, ci, va);
    transpose_csr(V, rp, ci, va, trp, tci, tva);
    std::vector<float> x(N * Fin), w((size_t)Fin * K * Fout), b(Fout), gy(N * Fout);
    for (auto& v : x) v = (float)urand();
    for (auto& v : w) v = (float)(urand() / std::sqrt((double)Fin * K));
    for (auto& v : b) v = (float)(0.2 * urand());
    for (auto& v : gy) v = (float)urand();

    // ---- checker (fp64, CPU)
    std::vector<double> xd(x.begin(), x.end()), wd(w.begin(), w.end()), bd(b.begin(), b.end()), gd(gy.begin(), gy.end());
    std::vector<double> y64(N * Fout), basis((size_t)K * N * Fin), dx64(N * Fin), dw64(w.size()), db64(Fout);
    oracle_cheb_forward(rp.data(), ci.data(), va.data(), V, xd.data(), wd.data(), bd.data(), y64.data(), basis.data(), B, Fin, Fout, K);
    oracle_cheb_backward(trp.data(), tci.data(), tva.data(), V, basis.data(), wd.data(), gd.data(), dx64.data(), dw64.data(),
                         db64.data(), B, Fin, Fout, K);

    // ---- the library, through its C ABI
    int32_t *d_rp = to_device(rp), *d_ci = to_device(ci), *d_trp = to_device(trp), *d_tci = to_device(tci);
    float *d_va = to_device(va), *d_tva = to_device(tva), *d_x = to_device(x), *d_w = to_device(w), *d_b = to_device(b),
          *d_gy = to_device(gy);
    float *d_y, *d_T, *d_dx, *d_dw, *d_db;
    HIP_OK(hipMalloc(&d_y, N * Fout * 4)); HIP_OK(hipMalloc(&d_T, (size_t)(K > 1 ? K - 1 : 1) * N * Fin * 4));
    HIP_OK(hipMalloc(&d_dx, N * Fin * 4)); HIP_OK(hipMalloc(&d_dw, w.size() * 4)); HIP_OK(hipMalloc(&d_db, Fout * 4));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    int rc = dsw_cheb_fwd(d_rp, d_ci, d_va, V, (int64_t)ci.size(), d_x, d_w, d_b, d_y, d_T, B, Fin, Fout, K, DSW_F32, stream, nullptr);
    if (rc) { std::printf("dsw_cheb_fwd: %s\n", dsw_strerror(rc)); return 1; }
    const int64_t wsb = dsw_cheb_bwd_workspace_bytes(B, V, Fin, Fout, K, DSW_F32);
    if (wsb < 0) { std::printf("workspace: %s\n", dsw_strerror((int)wsb)); return 1; }
    void* d_ws;
    HIP_OK(hipMalloc(&d_ws, wsb));
    rc = dsw_cheb_bwd(d_trp, d_tci, d_tva, V, (int64_t)tci.size(), d_x, dsw_cheb_mix_first(Fin, Fout, K) ? nullptr : d_T, d_w, d_gy,
                      d_dx, d_dw, d_db, d_ws, wsb, B, Fin, Fout, K, DSW_F32, stream, nullptr);
    if (rc) { std::printf("dsw_cheb_bwd: %s\n", dsw_strerror(rc)); return 1; }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> y(N * Fout), dx(N * Fin), dw(w.size()), db(Fout);
    HIP_OK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dx.data(), d_dx, dx.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dw.data(), d_dw, dw.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(db.data(), d_db, db.size() * 4, hipMemcpyDeviceToHost));
    const double ey = max_rel(y, y64), edx = max_rel(dx, dx64), edw = max_rel(dw, dw64), edb = max_rel(db, db64);
    const double tol = 2e-6;
    const bool ok = ey <= tol && edx <= tol && edw <= 2 * tol && edb <= 2 * tol;
    std::printf("V=%d B=%d %d->%d K=%d %s: y %.2e dx %.2e dw %.2e db %.2e  %s\n", V, B, Fin, Fout, K,
                dsw_cheb_mix_first(Fin, Fout, K) ? "mix-first" : "basis-first", ey, edx, edw, edb, ok ? "ok" : "FAIL");
    for (void* p : {(void*)d_rp, (void*)d_ci, (void*)d_trp, (void*)d_tci, (void*)d_va, (void*)d_tva, (void*)d_x, (void*)d_w,
                    (void*)d_b, (void*)d_gy, (void*)d_y, (void*)d_T, (void*)d_dx, (void*)d_dw, (void*)d_db, d_ws})
        HIP_OK(hipFree(p));
    HIP_OK(hipStreamDestroy(stream));
    return ok ? 0 : 1;
}

int main() {
    if (dsw_version() < 100) { std::printf("unexpected library version\n"); return 1; }
    int bad = 0;
    bad += run_case(1024, 3, 32, 64, 3);    // north-star channel shape, basis-first
    bad += run_case(1024, 2, 64, 32, 3);    // mix-first
    bad += run_case(700, 2, 18, 10, 4);     // unaligned everything
    bad += run_case(512, 1, 48, 16, 1);     // K = 1
    std::printf(bad ? "C-ABI CLIENT: FAIL\n" : "C-ABI CLIENT: PASS\n");
    return bad ? 1 : 0;
}
