// Shared device helpers for the dsw HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DSW_VERSION 100  // 0.1.0

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

static inline int dsw_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DSW_OK : DSW_ERR_LAUNCH;
}

static inline bool dsw_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
