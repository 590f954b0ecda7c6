// Skinny GEMM / GEMV for gfx950: out[M<=8, N] = act(x[M,K] . W[N,K]^T + bias) + residual.
//
// Batch-1 autoregressive decode (HF greedy search driven by InteractVLM.evaluate, model/InteractVLM.py:524-531)
// is pure weight streaming: 13.5 GB of bf16 weights per generated token for LLaMA-7B.  No MFMA (M is 1): waves own whole
// weight rows, stream them with non-temporal 16-byte loads (8 in flight per lane: 1 KB contiguous per wave instruction), dot
// them against the activation row(s) staged once per block in LDS and reduce with a wave butterfly.  Two kernels:
//   gemv1_kernel  M = 1, fp32 x (the decode path proper): 1024-thread blocks, ONE row per wave, no persistence - see below;
//   gemv_kernel   M <= 8, bf16 or fp32 x: persistent blocks, two rows per wave (SwiGLU over the row-interleaved gate / up
//                 weights completes inside the wave), software-pipelined over (row group, chunk batch) steps.
#include <stdlib.h>

#include <algorithm>

#include "kernels.h"

namespace ivlm {
namespace {

constexpr int kMaxM = 8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // native vector: nontemporal-loadable

__device__ __forceinline__ float dot8(const u32x4_t& w, const u32x4_t& x) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc += __uint_as_float(w[j] << 16) * __uint_as_float(x[j] << 16);
        acc += __uint_as_float(w[j] & 0xffff0000u) * __uint_as_float(x[j] & 0xffff0000u);
    }
    return acc;
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return x < 0.0f ? 0.0f : x;  // (torch.relu semantics: a NaN stays a NaN; fmaxf would turn it into 0)
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

// ROWS weight rows per wave (2: enables the SwiGLU epilogue and doubles the loads in flight; 1: small N, more
// waves).  XLDS: the activation rows (optionally RMS-normalised: x * gamma) are staged ONCE per block in LDS
// together with their row scale, instead of every wave re-deriving them from global memory for every weight row.
// AF32: the activations are fp32 (fp32 residual stream / fp32 hidden states): they stay fp32 in LDS and every product
// bf16 weight x fp32 activation is exact - the decode path carries no operand rounding at all (it is HBM-bound: the
// extra LDS bytes and the 8 FMAs per 16-byte weight chunk instead of 8 packed multiplies are free).
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

__device__ __forceinline__ float dot8f(const u32x4_t& w, const f32x4v_t& xa, const f32x4v_t& xb) {
    float acc = __uint_as_float(w[0] << 16) * xa[0];
    acc = fmaf(__uint_as_float(w[0] & 0xffff0000u), xa[1], acc);
    acc = fmaf(__uint_as_float(w[1] << 16), xa[2], acc);
    acc = fmaf(__uint_as_float(w[1] & 0xffff0000u), xa[3], acc);
    acc = fmaf(__uint_as_float(w[2] << 16), xb[0], acc);
    acc = fmaf(__uint_as_float(w[2] & 0xffff0000u), xb[1], acc);
    acc = fmaf(__uint_as_float(w[3] << 16), xb[2], acc);
    acc = fmaf(__uint_as_float(w[3] & 0xffff0000u), xb[3], acc);
    return acc;
}

__device__ __forceinline__ float gemm_residual_f32_or_bf16(const GemmArgs& g, int64_t col) {  // row 0 (M = 1)
    return g.res_f32 ? reinterpret_cast<const float*>(g.residual)[col] : bf16_to_f32(g.residual[col]);
}

template <int M, int ROWS, bool RMS, bool XLDS, bool AF32>
__global__ __launch_bounds__(256) void gemv_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[4][M];
    __shared__ float s_rstd[M];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;  // 8-element chunks per row (16 bytes of weights)
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);      // bf16 image [M][nchunk] (XLDS && !AF32)
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);    // fp32 image [M][2 planes][nchunk] float4 (XLDS && AF32)
    const float* Af = reinterpret_cast<const float*>(g.A);

#ifndef IVLM_GEMV_UMUL
#define IVLM_GEMV_UMUL 1
#endif
    constexpr int U = (ROWS == 2 ? 4 : 8) * IVLM_GEMV_UMUL;  // 8 x 16-byte weight loads in flight per lane per step
    const int wave_global = blockIdx.x * 4 + wave;
    const int nwaves = gridDim.x * 4;
    const int ngroups = (g.N + ROWS - 1) / ROWS;
    const int nbatch = (nchunk + 64 * U - 1) / (64 * U);
    const int my_groups = wave_global < ngroups ? (ngroups - wave_global + nwaves - 1) / nwaves : 0;
    const int nsteps = my_groups * nbatch;

    auto issue = [&](u32x4_t (&wa)[U], u32x4_t (&wb)[U], int step) __attribute__((always_inline)) {
        const int pr = wave_global + (step / nbatch) * nwaves;
        const int n0 = ROWS * pr, n1 = (ROWS == 2 && n0 + 1 < g.N) ? n0 + 1 : n0;
        const u32x4_t* w0 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n0 * g.ldw);
        const u32x4_t* w1 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n1 * g.ldw);
        const int c0 = (step % nbatch) * (64 * U) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cc = c0 + u * 64;
            cc = cc < nchunk ? cc : nchunk - 1;  // clamped (unconditional) loads keep the buffers in registers
            wa[u] = __builtin_nontemporal_load(w0 + cc);
            if (ROWS == 2) wb[u] = __builtin_nontemporal_load(w1 + cc);
        }
    };

    // ---- block prologue: stage x (x*gamma) in LDS, reduce sum(x^2) -----------------------------------
    auto prologue = [&]() __attribute__((always_inline)) {
        float ssq[M];
#pragma unroll
        for (int m = 0; m < M; ++m) ssq[m] = 0.0f;
        for (int c = threadIdx.x; c < nchunk; c += 256) {
            u32x4_t gv;
            if (RMS) gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + c);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (AF32) {
                    const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(Af + (int64_t)m * g.lda) + 2 * c;
                    f32x4v_t xa = xp[0], xb = xp[1];
                    if (RMS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) ssq[m] += xa[j] * xa[j] + xb[j] * xb[j];
                        xa[0] *= __uint_as_float(gv[0] << 16); xa[1] *= __uint_as_float(gv[0] & 0xffff0000u);
                        xa[2] *= __uint_as_float(gv[1] << 16); xa[3] *= __uint_as_float(gv[1] & 0xffff0000u);
                        xb[0] *= __uint_as_float(gv[2] << 16); xb[1] *= __uint_as_float(gv[2] & 0xffff0000u);
                        xb[2] *= __uint_as_float(gv[3] << 16); xb[3] *= __uint_as_float(gv[3] & 0xffff0000u);
                    }
                    xf[(2 * m) * nchunk + c] = xa;  // two planes: consecutive lanes read consecutive 16 bytes
                    xf[(2 * m + 1) * nchunk + c] = xb;
                } else {
                    u32x4_t xv = *(reinterpret_cast<const u32x4_t*>(g.A + (int64_t)m * g.lda) + c);
                    if (RMS) {
                        ssq[m] += dot8(xv, xv);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            xv[j] = pack_bf16x2(__uint_as_float(xv[j] << 16) * __uint_as_float(gv[j] << 16),
                                                __uint_as_float(xv[j] & 0xffff0000u) * __uint_as_float(gv[j] & 0xffff0000u));
                    }
                    xs[m * nchunk + c] = xv;
                }
            }
        }
        if (RMS) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(ssq[m]);
                if (lane == 0) s_red[wave][m] = v;
            }
        }
        __syncthreads();
        if (RMS) {
            if (threadIdx.x < M)
                s_rstd[threadIdx.x] = rsqrtf((s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] +
                                              s_red[3][threadIdx.x]) / (float)g.K + g.rms_eps);
            __syncthreads();
        }
    };

    float a0[M], a1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.0f;

    auto consume = [&](const u32x4_t (&wa)[U], const u32x4_t (&wb)[U], int step) __attribute__((always_inline)) {
        const int pr = wave_global + (step / nbatch) * nwaves;
        const int c0 = (step % nbatch) * (64 * U) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c0 + u * 64;
            if (cc < nchunk) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    if (AF32) {
                        f32x4v_t xa, xb;
                        if (XLDS) {
                            xa = xf[(2 * m) * nchunk + cc];
                            xb = xf[(2 * m + 1) * nchunk + cc];
                        } else {
                            const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(Af + (int64_t)m * g.lda) + 2 * cc;
                            xa = xp[0];
                            xb = xp[1];
                        }
                        a0[m] += dot8f(wa[u], xa, xb);
                        if (ROWS == 2) a1[m] += dot8f(wb[u], xa, xb);
                    } else {
                        const u32x4_t xv = XLDS ? xs[m * nchunk + cc]
                                                : *(reinterpret_cast<const u32x4_t*>(g.A + (int64_t)m * g.lda) + cc);
                        a0[m] += dot8(wa[u], xv);
                        if (ROWS == 2) a1[m] += dot8(wb[u], xv);
                    }
                }
            }
        }
        if ((step % nbatch) != nbatch - 1) return;
        // ---- last batch of this row group: reduce + epilogue ------------------------------------------
        const int n0 = ROWS * pr, n1 = (ROWS == 2 && n0 + 1 < g.N) ? n0 + 1 : n0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            a0[m] = wave_sum(a0[m]);
            if (ROWS == 2) a1[m] = wave_sum(a1[m]);
            if (RMS) {
                a0[m] *= s_rstd[m];
                a1[m] *= s_rstd[m];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v0 = a0[m] + (g.bias ? bf16_to_f32(g.bias[n0]) : 0.0f);
                float v1 = a1[m] + (g.bias ? bf16_to_f32(g.bias[n1]) : 0.0f);
                if (ROWS == 2 && g.act == ACT_SWIGLU) {
                    const float o = (v0 / (1.0f + __expf(-v0))) * v1;
                    const int64_t idx = (int64_t)m * g.ldc + pr;
                    if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
                    else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
                    continue;
                }
                v0 = act_apply(v0, g.act);
                v1 = act_apply(v1, g.act);
                if (g.residual) {
                    const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
                    if (g.res_f32) {
                        const float* R = reinterpret_cast<const float*>(g.residual);
                        v0 += R[rrow * g.ldr + n0];
                        v1 += R[rrow * g.ldr + n1];
                    } else {
                        v0 += bf16_to_f32(g.residual[rrow * g.ldr + n0]);
                        v1 += bf16_to_f32(g.residual[rrow * g.ldr + n1]);
                    }
                }
                if (g.out_f32) {
                    float* C = static_cast<float*>(g.C) + (int64_t)m * g.ldc;
                    C[n0] = v0;
                    if (n1 != n0) C[n1] = v1;
                } else {
                    bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc;
                    C[n0] = f32_to_bf16(v0);
                    if (n1 != n0) C[n1] = f32_to_bf16(v1);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.0f;
    };

    // ---- software pipeline over (row group, chunk batch) steps: the loads of step s+1 are in flight while
    //      step s is consumed, including across row groups and across the block prologue ---------------------
    u32x4_t wa0[U], wb0[U], wa1[U], wb1[U];
    if (nsteps > 0) issue(wa0, wb0, 0);
    if (XLDS) prologue();
    for (int s2 = 0; s2 < nsteps; s2 += 2) {
        if (s2 + 1 < nsteps) issue(wa1, wb1, s2 + 1);
        consume(wa0, wb0, s2);
        if (s2 + 1 < nsteps) {
            if (s2 + 2 < nsteps) issue(wa0, wb0, s2 + 2);
            consume(wa1, wb1, s2 + 1);
        }
    }
}

// ---- batch-1 decode GEMV (M = 1, fp32 x): the kernel every generated token runs 4 x 32 + 1 times --------------------
// Measured against pure streaming reads of the same matrices (tools/experiments/exp_access_pattern.hip: 6.4 TB/s for one
// row per wave, 1 KB per wave instruction), the persistent software-pipelined kernel above reaches 80 %; the SIMPLEST
// shape does better: 1024-thread blocks, ONE weight row per wave, one pass, no persistence - the hardware dispatcher
// balances the CUs, 32 resident waves per CU hide the latency, and x is staged once per 16 rows (16 KB of L2 reads per
// 128 KB of weights).  Same arithmetic as gemv_kernel<1, 1, RMS, true, true> (fp32 x in two LDS planes, exact products,
// per-lane ascending chunk order), so results are identical up to the block-level sum(x^2) order of the RMS prologue.
constexpr int kG1Waves = 16;

// (KS = 2 waves per weight row - K halves, partial sums through LDS - was measured for the 4096-row matrices, which put only
//  16 waves on each CU: 3-6 % slower on every shape.  KS stays a template parameter for that A/B: ivlm_gemv1_tuning.)
template <bool RMS, int KS>
__global__ __launch_bounds__(64 * kG1Waves, 8) void gemv1_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[kG1Waves];
    __shared__ float s_val[kG1Waves];
    constexpr int U = 8 / KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);  // [2 planes][nchunk] float4
    const int row = blockIdx.x * (kG1Waves / KS) + wave / KS;
    const bool live = row < g.N;
    const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)(live ? row : g.N - 1) * g.ldw);
    // this wave's chunk range (multiples of 64 chunks = 1 KB of the row per wave instruction)
    const int cpp = KS == 1 ? nchunk : ((nchunk + 64 * KS - 1) / (64 * KS)) * 64;
    const int c_lo = (wave % KS) * cpp, c_hi = min(nchunk, c_lo + cpp);
    u32x4_t w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min(c_lo + lane + 64 * u, nchunk - 1));
    // ---- stage x (x * gamma) in LDS, sum(x^2) ----
    float ssq = 0.0f;
    for (int c = threadIdx.x; c < nchunk; c += 64 * kG1Waves) {
        const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(g.A) + 2 * c;
        f32x4v_t xa = xp[0], xb = xp[1];
        if (RMS) {
            const u32x4_t gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssq += xa[j] * xa[j] + xb[j] * xb[j];
            xa[0] *= __uint_as_float(gv[0] << 16); xa[1] *= __uint_as_float(gv[0] & 0xffff0000u);
            xa[2] *= __uint_as_float(gv[1] << 16); xa[3] *= __uint_as_float(gv[1] & 0xffff0000u);
            xb[0] *= __uint_as_float(gv[2] << 16); xb[1] *= __uint_as_float(gv[2] & 0xffff0000u);
            xb[2] *= __uint_as_float(gv[3] << 16); xb[3] *= __uint_as_float(gv[3] & 0xffff0000u);
        }
        xf[c] = xa;
        xf[nchunk + c] = xb;
    }
    if (RMS) {
        ssq = wave_sum(ssq);
        if (lane == 0) s_red[wave] = ssq;
    }
    __syncthreads();
    float acc = 0.0f;
    // rows longer than one batch of U chunks per lane (down_proj: K = 11008): each half of the batch is refilled with the NEXT batch's
    // chunks as soon as it is consumed, so the wave always has U / 2 .. U loads in flight instead of a full round trip per batch
    // (same chunk order: identical sums)
    constexpr int UH = U / 2;
    for (int c = c_lo + lane; c < c_hi; c += 64 * U) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int u = hf * UH; u < (hf + 1) * UH; ++u) {
                const int cc = c + 64 * u;
                if (cc < c_hi) acc += dot8f(w[u], xf[cc], xf[nchunk + cc]);
            }
#pragma unroll
            for (int u = hf * UH; u < (hf + 1) * UH; ++u) {
                const int nc = c + 64 * (U + u);
                if (nc < c_hi) w[u] = __builtin_nontemporal_load(wp + nc);
            }
        }
    }
    acc = wave_sum(acc);
    if (KS > 1) {  // partial sums of the row's KS waves, added in part order
        if (lane == 0) s_val[wave] = acc;
        __syncthreads();
        if (wave % KS) return;
        acc = s_val[wave];
#pragma unroll
        for (int i = 1; i < KS; ++i) acc += s_val[wave + i];
    }
    if (RMS) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < kG1Waves; ++i) q += s_red[i];
        acc *= rsqrtf(q / (float)g.K + g.rms_eps);
    }
    float v = acc + ((g.bias && live) ? bf16_to_f32(g.bias[row]) : 0.0f);
    if (KS == 1 && g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: the even wave finishes the pair
        if (lane == 0) s_val[wave] = v;
        __syncthreads();
        if (lane != 0 || (wave & 1) || !live) return;
        const float o = (v / (1.0f + __expf(-v))) * s_val[wave + 1];
        const int64_t idx = row >> 1;
        if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
        else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
        return;
    }
    if (lane != 0 || !live) return;
    v = act_apply(v, g.act);
    if (g.residual) v += gemm_residual_f32_or_bf16(g, row);
    if (g.out_f32) static_cast<float*>(g.C)[row] = v;
    else static_cast<bf16_t*>(g.C)[row] = f32_to_bf16(v);
}

// ---- e4m3 WEIGHTS for the batch-1 decode linears (BASELINE.json configs[4]; opt-in variant) -----------------------------------
// The decode step is HBM-bound on the weight bytes, so this is where fp8 pays most: half the bytes per token.  W is a byte
// matrix [N, K] of OCP e4m3 values with ONE per-tensor scale (W ~ q * *scale_w: the tensor the fp8 prefill GEMM also uses); the
// activation row stays fp32 in LDS and the products q x x are exact in fp32 (the only error is the weight quantisation).  Same
// shape as gemv1_kernel: 1024-thread blocks, one row per wave, 16-byte non-temporal loads (16 weights each), fused RMSNorm
// prologue, SwiGLU / residual epilogues.  Rows of K bytes: K % 16 == 0.
typedef __attribute__((ext_vector_type(2))) float f32x2v_t;

__device__ __forceinline__ float dot16_fp8(const u32x4_t& w, const f32x4v_t& a0, const f32x4v_t& a1, const f32x4v_t& b0,
                                           const f32x4v_t& b1, float acc) {
    // 16 e4m3 weights = elements 16c .. 16c+15 of the row against x chunks 2c (a0 | a1) and 2c + 1 (b0 | b1)
    const f32x4v_t* xs[4] = {&a0, &a1, &b0, &b1};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2v_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[j], false);
        const f32x2v_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[j], true);
        const f32x4v_t& x = *xs[j];
        acc = fmaf(lo[0], x[0], acc);
        acc = fmaf(lo[1], x[1], acc);
        acc = fmaf(hi[0], x[2], acc);
        acc = fmaf(hi[1], x[3], acc);
    }
    return acc;
}

template <bool RMS>
__global__ __launch_bounds__(64 * kG1Waves, 8) void gemv1_fp8w_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[kG1Waves];
    __shared__ float s_val[kG1Waves];
    constexpr int U = 4;  // 4 x 16 bytes in flight per lane: one pass over a 4096-byte row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;   // fp32 x chunks of 8 elements (two 16-byte planes)
    const int nw16 = g.K >> 4;     // 16-byte weight chunks per row
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);
    const int row = blockIdx.x * kG1Waves + wave;
    const bool live = row < g.N;
    const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint8_t*>(g.W) + (int64_t)(live ? row : g.N - 1) * g.ldw);
    u32x4_t w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min(lane + 64 * u, nw16 - 1));
    float ssq = 0.0f;
    for (int c = threadIdx.x; c < nchunk; c += 64 * kG1Waves) {
        const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(g.A) + 2 * c;
        f32x4v_t xa = xp[0], xb = xp[1];
        if (RMS) {
            const u32x4_t gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssq += xa[j] * xa[j] + xb[j] * xb[j];
            xa[0] *= __uint_as_float(gv[0] << 16); xa[1] *= __uint_as_float(gv[0] & 0xffff0000u);
            xa[2] *= __uint_as_float(gv[1] << 16); xa[3] *= __uint_as_float(gv[1] & 0xffff0000u);
            xb[0] *= __uint_as_float(gv[2] << 16); xb[1] *= __uint_as_float(gv[2] & 0xffff0000u);
            xb[2] *= __uint_as_float(gv[3] << 16); xb[3] *= __uint_as_float(gv[3] & 0xffff0000u);
        }
        xf[c] = xa;
        xf[nchunk + c] = xb;
    }
    if (RMS) {
        ssq = wave_sum(ssq);
        if (lane == 0) s_red[wave] = ssq;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int c = lane; c < nw16; c += 64 * U) {
        if (c != lane) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c + 64 * u < nw16) w[u] = __builtin_nontemporal_load(wp + c + 64 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + 64 * u;
            if (cc < nw16) acc = dot16_fp8(w[u], xf[2 * cc], xf[nchunk + 2 * cc], xf[2 * cc + 1], xf[nchunk + 2 * cc + 1], acc);
        }
    }
    acc = wave_sum(acc) * (*g.scale_w);
    if (RMS) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < kG1Waves; ++i) q += s_red[i];
        acc *= rsqrtf(q / (float)g.K + g.rms_eps);
    }
    float v = acc + ((g.bias && live) ? bf16_to_f32(g.bias[row]) : 0.0f);
    if (g.act == ACT_SWIGLU) {
        if (lane == 0) s_val[wave] = v;
        __syncthreads();
        if (lane != 0 || (wave & 1) || !live) return;
        const float o = (v / (1.0f + __expf(-v))) * s_val[wave + 1];
        const int64_t idx = row >> 1;
        if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
        else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
        return;
    }
    if (lane != 0 || !live) return;
    v = act_apply(v, g.act);
    if (g.residual) v += gemm_residual_f32_or_bf16(g, row);
    if (g.out_f32) static_cast<float*>(g.C)[row] = v;
    else static_cast<bf16_t*>(g.C)[row] = f32_to_bf16(v);
}

int g_gemv1_ksplit = 0;  // 0 = rule below (A/B hook: ivlm_gemv1_tuning)
int g_gemv1_lds_floor = 0;  // dynamic LDS requested by grids of <= one block per CU (A/B hook: ivlm_gemv1_lds_floor)

template <bool RMS, int KS>
static void launch_gemv1_ks(const GemmArgs& g, hipStream_t st) {
    static ivlm_dev_mask_t set{0};
    auto kfn = gemv1_kernel<RMS, KS>;
    if (ivlm_dev_pending(set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        ivlm_dev_done(set);
    }
    const int rows_per_block = kG1Waves / KS;
    const int blocks = (g.N + rows_per_block - 1) / rows_per_block;
    size_t lds = (size_t)g.K * 4;
    if (blocks <= 256 && (size_t)g_gemv1_lds_floor > lds) lds = (size_t)g_gemv1_lds_floor;
    ivlm_launch(kfn, dim3(blocks), dim3(64 * kG1Waves), lds, st, g);
}

static int launch_gemv1(const GemmArgs& g, hipStream_t st) {
    int ks = g_gemv1_ksplit > 0 ? g_gemv1_ksplit : 1;
    if (g.act == ACT_SWIGLU || g.K < 2048) ks = 1;
    if (ks == 2) {
        if (g.rms_w) launch_gemv1_ks<true, 2>(g, st);
        else launch_gemv1_ks<false, 2>(g, st);
    } else {
        if (g.rms_w) launch_gemv1_ks<true, 1>(g, st);
        else launch_gemv1_ks<false, 1>(g, st);
    }
    return ivlm_launch_status();
}

// first index of the row maximum (torch.argmax tie rule), one 1024-thread block per row, 16-byte loads
// (the lm_head logits row is 128 KB: ~5 us instead of 38 us for the 256-thread scalar version)
// bump != NULL: bump[row] += 1 by the same launch (the decode graphs step their device-side positions here instead of with a launch of
// their own)
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, int cols, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ bump) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* r = x + (int64_t)blockIdx.x * cols;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    };
    const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 2);  // scalars up to the first 16-byte boundary
    const int nhead = head < cols ? head : cols;
    if ((int)threadIdx.x < nhead) take(r[threadIdx.x], threadIdx.x);
    const int nvec = (cols - nhead) >> 2;
    const float4* rv = reinterpret_cast<const float4*>(r + nhead);
    for (int i = threadIdx.x; i < nvec; i += 1024) {
        const float4 v = rv[i];
        const int c = nhead + 4 * i;
        take(v.x, c);
        take(v.y, c + 1);
        take(v.z, c + 2);
        take(v.w, c + 3);
    }
    for (int i = nhead + 4 * nvec + threadIdx.x; i < cols; i += 1024) take(r[i], i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = best;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        out[blockIdx.x] = bi;
        if (bump) bump[blockIdx.x] += 1;
    }
}

}  // namespace

// Grid shaping (measured on the LLaMA-7B decode shapes, tools/bench_gemv.py): every block must be resident at once (a second
// dispatch round runs at a fraction of the occupancy) and every wave should own the SAME number of row groups (the launch
// ends when the most loaded wave does): with `slots` = CUs x resident blocks per CU (LDS-limited: the fp32 x image of
// down_proj is 44 KB), each wave takes g = ceil(groups / (4 * slots)) groups and the grid is ceil(groups / (4 g)) blocks.
int g_gemv_max_blocks_per_cu = 0, g_gemv_rows2_min_n = 0;  // experiment hooks (0 = defaults)
int g_gemv_legacy = 0;  // 1: M = 1 fp32 rows also take the persistent kernel (A/B hook of tools/bench_decode.py)
static int gemv_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = 256;
    }
    return cus;
}

template <int M>
static int launch_gemv(const GemmArgs& g, hipStream_t st) {
    const size_t xbytes = (size_t)M * g.K * (g.a_f32 ? 4 : 2);
    // (x staged in LDS also when the GEMV shares the CUs with the encoder's 128-KB-LDS GEMM blocks: never staging it costs
    //  12 ms end to end, staging only K = 4096 vectors 1 ms - LDS room is not what slows the decode next to the encoder)
    const bool xlds = xbytes <= (g.a_f32 ? 60 : 48) * 1024;
    if (g.rms_w && !xlds) return IVLM_ERR_UNSUPPORTED;  // (such shapes take the skinny MFMA kernel)
    const int rows2_min = g_gemv_rows2_min_n > 0 ? g_gemv_rows2_min_n : 8192;
    const bool rows2 = g.act == ACT_SWIGLU || g.N > rows2_min;  // small N: one row per wave = twice the waves
    const int ngroups = rows2 ? (g.N + 1) / 2 : g.N;
    const size_t lds = xlds ? xbytes : 0;
    int per_cu = g_gemv_max_blocks_per_cu > 0 ? g_gemv_max_blocks_per_cu : 4;
    if (lds > 0) per_cu = std::min<int>(per_cu, std::max<int>(1, (int)((160 * 1024) / (lds + 512))));
    const int slots = gemv_cu_count() * per_cu;
    const int gpw = std::max(1, (ngroups + 4 * slots - 1) / (4 * slots));  // row groups per wave
    const int blocks = (ngroups + 4 * gpw - 1) / (4 * gpw);
#define IVLM_GEMV_GO(ROWS, RMS, XL, AF)                                                                               \
    do {                                                                                                              \
        auto kfn = gemv_kernel<M, ROWS, RMS, XL, AF>;                                                                 \
        if (lds > 48 * 1024) {                                                                                        \
            static ivlm_dev_mask_t set{0};                                                                            \
            if (ivlm_dev_pending(set)) {                                                                              \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          64 * 1024);                                                                 \
                ivlm_dev_done(set);                                                                                   \
            }                                                                                                         \
        }                                                                                                             \
        ivlm_launch(kfn, dim3(blocks), dim3(256), lds, st, g);                                                        \
    } while (0)
#define IVLM_GEMV_ROWS(ROWS)                                                                   \
    if (g.a_f32) {                                                                             \
        if (g.rms_w) IVLM_GEMV_GO(ROWS, true, true, true);                                     \
        else if (xlds) IVLM_GEMV_GO(ROWS, false, true, true);                                  \
        else IVLM_GEMV_GO(ROWS, false, false, true);                                           \
    } else {                                                                                   \
        if (g.rms_w) IVLM_GEMV_GO(ROWS, true, true, false);                                    \
        else if (xlds) IVLM_GEMV_GO(ROWS, false, true, false);                                 \
        else IVLM_GEMV_GO(ROWS, false, false, false);                                          \
    }
    if (rows2) { IVLM_GEMV_ROWS(2) } else { IVLM_GEMV_ROWS(1) }
#undef IVLM_GEMV_ROWS
#undef IVLM_GEMV_GO
    return ivlm_launch_status();
}

int gemv_bf16(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.M > kMaxM || g.N <= 0 || g.K <= 0) return IVLM_ERR_INVALID_ARG;
    if ((g.K & 7) || (g.lda & (g.a_f32 ? 3 : 7)) || (g.ldw & 7) || g.batch != 1) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    if (g.M == 1 && g.a_f32 && (size_t)g.K * 4 <= 60 * 1024 && !g_gemv_legacy) return launch_gemv1(g, st);
    switch (g.M) {
        case 1: return launch_gemv<1>(g, st);
        case 2: return launch_gemv<2>(g, st);
        case 3: return launch_gemv<3>(g, st);
        case 4: return launch_gemv<4>(g, st);
        case 5: return launch_gemv<5>(g, st);
        case 6: return launch_gemv<6>(g, st);
        case 7: return launch_gemv<7>(g, st);
        default: return launch_gemv<8>(g, st);
    }
}

// M = 1, fp32 x, e4m3 weight bytes [N, K] (row stride ldw BYTES), per-tensor scale in device memory
int gemv1_fp8w(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || !g.scale_w || g.M != 1 || !g.a_f32 || g.N <= 0 || g.K <= 0) return IVLM_ERR_INVALID_ARG;
    if ((g.K & 15) || (g.ldw & 15) || (size_t)g.K * 4 > 60 * 1024) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    static ivlm_dev_mask_t set0{0}, set1{0};
    if (g.rms_w) {
        auto kfn = gemv1_fp8w_kernel<true>;
        if (ivlm_dev_pending(set1)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            ivlm_dev_done(set1);
        }
        ivlm_launch(kfn, dim3((g.N + kG1Waves - 1) / kG1Waves), dim3(64 * kG1Waves), (size_t)g.K * 4, st, g);
    } else {
        auto kfn = gemv1_fp8w_kernel<false>;
        if (ivlm_dev_pending(set0)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            ivlm_dev_done(set0);
        }
        ivlm_launch(kfn, dim3((g.N + kG1Waves - 1) / kG1Waves), dim3(64 * kG1Waves), (size_t)g.K * 4, st, g);
    }
    return ivlm_launch_status();
}

int argmax_f32(const float* x, int rows, int cols, int32_t* out, hipStream_t st, int32_t* bump) {
    if (!x || !out || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    argmax_kernel<<<rows, 1024, 0, st>>>(x, cols, out, bump);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_gemv_tuning(int max_blocks_per_cu, int rows2_min_n) {  // benchmark hook: 0 = defaults
    ivlm::g_gemv_legacy = max_blocks_per_cu < 0 ? 1 : 0;  // negative: the persistent kernel also for M = 1 fp32 rows
    ivlm::g_gemv_max_blocks_per_cu = max_blocks_per_cu > 0 ? max_blocks_per_cu : 0;
    ivlm::g_gemv_rows2_min_n = rows2_min_n;
    return 0;
}

extern "C" void ivlm_gemv1_tuning(int ksplit) { ivlm::g_gemv1_ksplit = ksplit; }
extern "C" void ivlm_gemv1_lds_floor(int bytes) { ivlm::g_gemv1_lds_floor = bytes < 0 ? 0 : (bytes > 128 * 1024 ? 128 * 1024 : bytes); }

extern "C" int ivlm_gemv_fp8w(const float* x, const void* Wq, int64_t ldw, const float* scale_w, void* C, const void* bias,
                              const void* residual, int N, int K, int act, int out_f32, const void* rms_w, float rms_eps, int flags,
                              ivlm_stream_t stream) {
    ivlm_enter();
    ivlm::GemmArgs g;
    g.A = reinterpret_cast<const bf16_t*>(x);
    g.a_f32 = 1;
    g.W = static_cast<const bf16_t*>(Wq);
    g.ldw = ldw;
    g.scale_w = scale_w;
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    g.M = 1; g.N = N; g.K = K;
    g.lda = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = rms_eps;
    return ivlm::gemv1_fp8w(g, ivlm_stream(stream));
}

extern "C" int ivlm_argmax_f32(const float* x, int rows, int cols, int32_t* out, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::argmax_f32(x, rows, cols, out, ivlm_stream(stream), nullptr);
}

extern "C" int ivlm_argmax_f32_bump(const float* x, int rows, int cols, int32_t* out, int32_t* bump, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::argmax_f32(x, rows, cols, out, ivlm_stream(stream), bump);
}
