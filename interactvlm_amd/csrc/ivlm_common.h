// Shared device/host helpers for libivlm_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

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

// Function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) are per DEVICE: a process that drives several GPUs must set them
// once on each (process-wide `static bool` flags left the second GPU's launches of > 64 KB LDS failing).  One bit per
// device id in a static mask next to the kernel; the bit is published AFTER the attribute is set, so a concurrent first launch from
// another thread at worst sets the attribute twice.
typedef std::atomic<uint64_t> ivlm_dev_mask_t;
static inline uint64_t ivlm_dev_bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return 1ull << (dev & 63);
}
static inline bool ivlm_dev_pending(const ivlm_dev_mask_t& m) { return !(m.load(std::memory_order_acquire) & ivlm_dev_bit()); }
static inline void ivlm_dev_done(ivlm_dev_mask_t& m) { m.fetch_or(ivlm_dev_bit(), std::memory_order_release); }

// Launch helper of the kernels whose duration bench.py reports (GEMM / GEMV / lift families): when the caller armed
// ivlm_profile_launches(start, stop), the HIP events are attached to the KERNEL (hipExtLaunchKernelGGL: recorded by the command
// processor at its first / last wave) instead of being separate records around it - an event pair around one ~17 us launch reads
// 5-12 us too long.
bool ivlm_profile_events(hipEvent_t* start, hipEvent_t* stop);
template <typename K, typename... A>
static inline void ivlm_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
    hipEvent_t ev0, ev1;
    if (ivlm_profile_events(&ev0, &ev1)) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, ev0, ev1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, args...);
}

// ---- bf16 <-> f32 (round-to-nearest-even), raw 16-bit storage -------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16 through the hardware converter of gfx950 (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN): one VALU
// op per PAIR instead of ~6 per value for the integer add-and-shift formulation - the bf16 epilogues of the GEMMs, the norm
// outputs and the hi + lo operand split of the fp32-activation kernels were VALU-bound on the conversion.
typedef __attribute__((ext_vector_type(2))) __bf16 ivlm_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float ivlm_f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // (a compiler-visible conversion, not inline asm: the hazard recognizer must see it - after a v_dot2c accumulation chain an
    //  opaque asm read the accumulator too early)
    const ivlm_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ivlm_bf16x2_t));
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// two fp32 -> packed IEEE half pair (round to nearest even).  NOT saturating: a value past +-65504 becomes inf and the result of the
// call NaN - which the model's guard sees and recomputes with bf16 operands (model.py evaluate); a clamp would return a plausible,
// wrong answer instead
typedef __attribute__((ext_vector_type(2))) _Float16 ivlm_f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const ivlm_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ivlm_f16x2_t));
}
// 16-bit output pair of the GEMM epilogues: bf16 or (f16 != 0) fp16
__device__ __forceinline__ uint32_t pack_16x2(float lo, float hi, int f16) { return f16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }

// one 16-bit storage element <-> fp32, operand kind chosen at compile time: F16 = IEEE half (overflow -> inf, see pack_f16x2), else bf16
template <bool F16>
__device__ __forceinline__ float h16_to_f32(uint16_t v) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
    else return bf16_to_f32(v);
}
template <bool F16>
__device__ __forceinline__ uint16_t f32_to_h16(float f) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)f);
    else return f32_to_bf16(f);
}
template <bool F16>
__device__ __forceinline__ float pair_lo_f32(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(ivlm_f16x2_t, u)[0];
    else return __uint_as_float(u << 16);
}
template <bool F16>
__device__ __forceinline__ float pair_hi_f32(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(ivlm_f16x2_t, u)[1];
    else return __uint_as_float(u & 0xffff0000u);
}

// hi + lo bf16 split of fp32 values (x = hi + lo to 2^-17 relative): packed pairs, hi = RNE(x), lo = RNE(x - hi)
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// the same with the element type chosen at run time: f16 != 0 -> hi + lo IEEE halves (x = hi + lo to 2^-22 while lo stays normal)
__device__ __forceinline__ void split_16x2(float a, float b, uint32_t& hi, uint32_t& lo, int f16) {
    if (f16) {
        hi = pack_f16x2(a, b);
        const ivlm_f16x2_t h = __builtin_bit_cast(ivlm_f16x2_t, hi);
        lo = pack_f16x2(a - (float)h[0], b - (float)h[1]);
    } else {
        split_bf16x2(a, b, hi, lo);
    }
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
