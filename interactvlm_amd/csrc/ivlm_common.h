// Shared device/host helpers for libivlm_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ivlm_hip.h"

#define IVLM_WAVE 64

#define IVLM_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return IVLM_ERR_INVALID_ARG; \
    } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (torch, RCCL): clear it on entry so that we only ever report our own launches.
static inline void ivlm_enter() { (void)hipGetLastError(); }

extern "C" void ivlm_set_last_hip_error(int code, const char* where);

#define ivlm_launch_status() ivlm_launch_status_at(__FILE__, __LINE__)
static inline int ivlm_launch_status_at(const char* file, int line) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return IVLM_OK;
    char buf[256];
    const char* base = file;
    for (const char* c = file; *c; ++c)
        if (*c == '/') base = c + 1;
    snprintf(buf, sizeof(buf), "%s:%d", base, line);
    ivlm_set_last_hip_error((int)e, buf);
    return IVLM_ERR_LAUNCH;
}

// run a HIP runtime call; on failure record the detail and return IVLM_ERR_LAUNCH from the caller
#define IVLM_HIP_TRY(expr)                                        \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) {                                   \
            ivlm_set_last_hip_error((int)e_, #expr);              \
            return IVLM_ERR_LAUNCH;                               \
        }                                                         \
    } while (0)

static inline hipStream_t ivlm_stream(ivlm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- bf16 <-> f32 (round-to-nearest-even), raw 16-bit storage -------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ---- wave64 reductions (fixed butterfly order => deterministic) ------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }
