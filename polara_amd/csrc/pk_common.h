// Shared helpers for the polara_hip kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/polara_hip.h"

#define PK_WAVE 64

void pk_set_error(const char *fmt, ...);
// process-wide options set by explicit calls (pk_set_option, api.cpp) — the hooks the tests use to force a code path on small
// inputs; the library reads NO environment variable.  Returns `dflt` while the option is unset.
int pk_option(const char *name, int dflt);

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

// ---- "done once" flags for PER-DEVICE function attributes (hipFuncSetAttribute raises the dynamic-LDS limit of a kernel
// on the CURRENT device only; the coarse ABI allows contexts on several devices in one process).  One bit per device
// ordinal; two threads racing on the first call both set the attribute, which is idempotent.
struct PkDeviceOnce {
    std::atomic<uint64_t> mask{0};
    uint64_t bit() const {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool pending() const { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

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

// ---- value of lane (lane ^ J) without touching the LDS pipe ---------------------------------------
// __shfl_xor compiles to ds_bpermute_b32: ~100+ cycles of LDS latency per dependent step, which made
// the 21-stage wave sorts of the scoring kernels latency-bound (6.3K cycles per flush).  DPP modifiers
// cover J = 1, 2 (quad_perm), 4 (row_half_mirror then quad reversal), 8 (row_ror:8); gfx950's
// v_permlane16_swap / v_permlane32_swap cover J = 16, 32.
template <int J>
__device__ __forceinline__ int pk_lane_xor(int v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "pk_lane_xor: J must be a power of two < 64");
    if constexpr (J == 1) {
        return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        const int y = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);   // row_half_mirror: i -> 7 - i
        return __builtin_amdgcn_update_dpp(0, y, 0x1B, 0xf, 0xf, true);            // quad_perm [3,2,1,0]
    } else if constexpr (J == 8) {
        return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);  // row_ror:8
    } else if constexpr (J == 16) {
        // (a, b) <- swap(odd rows of a, even rows of b): a = (v0 v0 v2 v2), b = (v1 v1 v3 v3)
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return ((threadIdx.x >> 4) & 1) ? (int)r[0] : (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return ((threadIdx.x >> 5) & 1) ? (int)r[0] : (int)r[1];
    }
}
template <int J>
__device__ __forceinline__ float pk_lane_xor(float v) {
    return __int_as_float(pk_lane_xor<J>(__float_as_int(v)));
}
template <int J>
__device__ __forceinline__ double pk_lane_xor(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = pk_lane_xor<J>((int)(unsigned)(b & 0xffffffffll)), hi = pk_lane_xor<J>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
