// Shared device helpers for the dsw HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define DSW_VERSION 101  // 0.1.1: dsw_hop2_plan carries its own size (struct_bytes)

// error codes returned by every C-ABI entry point (0 = ok)
#define DSW_OK 0
#define DSW_ERR_BAD_ARG (-1)
#define DSW_ERR_BAD_DTYPE (-2)
#define DSW_ERR_WORKSPACE (-3)
#define DSW_ERR_LAUNCH (-4)
#define DSW_ERR_ALIGN (-5)

#define DSW_F32 0
#define DSW_BF16 1

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

static __device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}
static __device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (uint16_t)(u >> 16);
}

// Diagnostic switches (A/B runs, ablations, kernel-selection overrides) exist only in builds made with -DDSW_DIAG
// (`DSW_BUILD_DIAG=1 python -m dsw_amd.build`): the product library reads NO environment variable - a process-wide,
// read-once switch that silently changes the evaluation order has no place in it.
#ifdef DSW_DIAG
#include <cstdlib>
static inline const char* dsw_diag_env(const char* name) { return getenv(name); }
#else
static inline const char* dsw_diag_env(const char*) { return nullptr; }
#endif

// Build flags of this translation unit, collected at load time (dsw_build_flags()): a library any of whose objects was
// compiled with a diagnostics / ablation switch is refused by dsw_amd._native.load() unless it was asked for by path
// (DSW_HIP_LIB) - a variant build is one -D away from a wrong-by-design product library.
#define DSW_FLAG_DIAG 1
#define DSW_FLAG_ABLATION 2
#if defined(DSW_DIAG)
#define DSW_TU_F_DIAG DSW_FLAG_DIAG
#else
#define DSW_TU_F_DIAG 0
#endif
#if defined(DSW_ABLATION) || defined(DSW_ABL_NODMA) || defined(DSW_ABL_NOGATHER) || defined(DSW_ABL_F3_NOGATHER) || \
    defined(DSW_ABL_F3_NOSPLIT) || defined(DSW_F3_PLAIN_STORE) || defined(DSW_ABL_F3_NOLOAD) || defined(DSW_ABL_F3_NOSTORE) || defined(DSW_ABL_F3_NOMFMA) || defined(DSW_ABL_B3_NOSPLIT) || \
    defined(DSW_ABL_B3_NOGATHER) || defined(DSW_ABL_B3_NOMFMA) || defined(DSW_STAGE_EARLY) || defined(DSW_ABL_B3_NOLOAD) || defined(DSW_ABL_B3_NOFRAG) || \
    defined(DSW_GATHER_N) || defined(DSW_F3_YBOUNCE) || defined(DSW_F3_SKEW) || defined(DSW_D3_SKEW) || defined(DSW_D3_WG_PER_CU) || \
    defined(DSW_D3_NO_FOUT32) || defined(DSW_X3S_NO_DIRECT) || defined(DSW_X3S_BFRAG_ALL) || defined(DSW_ABL_X3S_NOB) || defined(DSW_X3S_LB4) || defined(DSW_X3S_FORCE_NT2)   /* round-6 A/B switches (correct results, other schedules) */
#define DSW_TU_F_ABL DSW_FLAG_ABLATION
#else
#define DSW_TU_F_ABL 0
#endif
int dsw_register_build_flags(int flags);                                            // dsw_api.hip
namespace { const int dsw_tu_build_flags_registered = dsw_register_build_flags(DSW_TU_F_DIAG | DSW_TU_F_ABL); }

// Launch tracing (dsw_trace_begin / dsw_trace_end, dsw_api.hip).  While a trace is open every kernel of the library is
// launched with a start / stop event pair ATTACHED TO ITS DISPATCH (hipExtLaunchKernelGGL: the timestamps of the kernel's own
// completion signal - no marker packets between the kernels, so the launch sequence runs as it does without the trace), and
// the entry points drop role markers (host-side sequence points, nothing on the stream) that say which role the kernels
// since the previous marker belong to.  Off: one relaxed atomic load per launch.
bool dsw_trace_kernel(const char* name, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1);
void dsw_trace_point(hipStream_t s, int role, int64_t a0, int64_t a1, int64_t a2);     // role 0 = start of an entry point
#define DSW_LAUNCH(kern, grid, block, lds, stream, ...)                                                 \
    do {                                                                                                  \
        hipEvent_t dsw_e0_ = nullptr, dsw_e1_ = nullptr;                                                  \
        if (dsw_trace_kernel(#kern, (stream), &dsw_e0_, &dsw_e1_))                                        \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, dsw_e0_, dsw_e1_, 0, __VA_ARGS__);      \
        else                                                                                              \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                              \
    } while (0)
struct DswTraceScope {   // start marker now, the role's end marker when the entry point returns
    hipStream_t s; int role; int64_t a0, a1, a2;
    DswTraceScope(hipStream_t s_, int role_, int64_t a0_ = 0, int64_t a1_ = 0, int64_t a2_ = 0)
        : s(s_), role(role_), a0(a0_), a1(a1_), a2(a2_) { dsw_trace_point(s, 0, 0, 0, 0); }
    ~DswTraceScope() { dsw_trace_point(s, role, a0, a1, a2); }
};

static inline int dsw_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DSW_OK : DSW_ERR_LAUNCH;
}

// compute units of the current device (cached per device ordinal; 256 on a whole MI355X, fewer in partition modes)
static inline long dsw_device_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

static inline bool dsw_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
