// Internal C++ launch API shared by the C-ABI wrappers and the stage runners (not installed).
#pragma once
#include "ivlm_common.h"

namespace ivlm {

// ---- epilogue activation codes (also exposed through the C ABI) --------------------------------
enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICK_GELU = 2, ACT_RELU = 3, ACT_SILU = 4, ACT_SWIGLU = 5 };

struct GemmArgs {
    const bf16_t* A = nullptr;  // activations [M,K], row stride lda (elements)
    const bf16_t* W = nullptr;  // weights     [N,K], row stride ldw  (nn.Linear layout)
    void* C = nullptr;          // output      [M,N] (or [M,N/2] for ACT_SWIGLU), row stride ldc
    const bf16_t* bias = nullptr;      // [N] or null
    const bf16_t* residual = nullptr;  // [*,N] added after the activation, row stride ldr, or null
    int64_t lda = 0, ldw = 0, ldc = 0, ldr = 0;
    int res_mod = 0;   // >0: residual row = m % res_mod (broadcast tables such as pos_embed)
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    int out_f32 = 0;   // 1: C is float
    // batched (strided) variant: blockIdx.z = batch
    int batch = 1;
    int64_t strideA = 0, strideW = 0, strideC = 0, strideR = 0;
};

// bf16 x bf16 -> fp32-accumulate MFMA GEMM with fused bias/activation/residual epilogue.
int gemm_bf16(const GemmArgs& g, hipStream_t st);

// ---- normalisation ---------------------------------------------------------------------------
// y = (x-mean)/sqrt(var+eps)*w+b over the last dim (rows x cols); bf16 in/out, fp32 statistics.
int layernorm_bf16(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int64_t rows, int cols, float eps,
                   hipStream_t st);
// y = x * rsqrt(mean(x^2)+eps) * w  (LLaMA RMSNorm; fp32 statistics, HF casts back before the weight multiply)
int rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int cols, float eps, hipStream_t st);

}  // namespace ivlm
