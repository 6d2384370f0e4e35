// Shared helpers for the polara_hip kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/polara_hip.h"

#define PK_WAVE 64

void pk_set_error(const char *fmt, ...);

#define PK_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            pk_set_error(__VA_ARGS__);        \
            return PK_E_INVALID;              \
        }                                     \
    } while (0)

#define PK_CHECK_LAUNCH(name)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            pk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return PK_E_LAUNCH;                                                        \
        }                                                                              \
    } while (0)

static inline hipStream_t pk_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t pk_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave-level reductions (64 lanes) -----------------------------------------------------
__device__ __forceinline__ double pk_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float pk_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int pk_lane() { return threadIdx.x & 63; }
