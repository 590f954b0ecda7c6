// Pieces shared by the GEMM tile kernels: activation, LDS-DMA helper, XCD-aware tile raster, fused epilogue.
// (A 4-phase / counted-vmcnt schedule of the 256x256 tile was built and measured in round 1: correct but 5-8 % slower
// than the one-barrier-per-K-tile loop of gemm.hip, so it is not shipped)
#pragma once
#include "kernels.h"

namespace ivlm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// one MX fp8 step: the two 16-byte fragments a lane holds for the two bf16 k-steps of a 128-byte K tile are, as bytes, 32 e4m3
// values of ONE 16x16x128 step (both operands use the same lane -> k-subset map, so the sum over k is the same sum)
__device__ __forceinline__ f32x4_t mfma_fp8_128(const bf16x8_t& a0, const bf16x8_t& a1, const bf16x8_t& b0, const bf16x8_t& b1,
                                                const f32x4_t& c) {
    struct P { bf16x8_t lo, hi; };
    const P pa{a0, a1}, pb{b0, b1};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_bit_cast(i32x8_t, pa), __builtin_bit_cast(i32x8_t, pb), c,
                                                            0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  // e4m3 x e4m3, scales 2^0
}

// exact-GELU x * Phi(x) with Phi through erfc (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7, no cancellation for x < 0):
// ~14 VALU ops instead of ~45 for ocml erff - the GELU epilogue of the 16384x5120x1280 SAM MLP GEMM was 30 % of its time
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = 0.5f * p * t * __expf(-z * z);  // 0.5 * erfc(z)
    return x * (x < 0.0f ? q : 1.0f - q);
}

__device__ __forceinline__ float gemm_act(float x, int act) {
    switch (act) {
        case ACT_GELU: return gelu_erf_fast(x);
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

// 16 zero bytes: DMA source for the K tail (K % 64 != 0) so that partial tiles contribute nothing
__device__ __attribute__((aligned(16))) const uint32_t kGemmZeroChunk[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// XCD-aware, grouped tile raster (see gemm.hip): block id -> tile origin
__device__ __forceinline__ void gemm_tile_origin(const GemmArgs& g, int BM, int BN, int& m0, int& n0) {
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    constexpr int kXcd = 8, kGroupM = 8;
    const int bid = blockIdx.x;
    const int xcd = bid % kXcd, loc = bid / kXcd;
    const int q = nwg / kXcd, r = nwg % kXcd;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per_group = kGroupM * tiles_n;
    const int first_m = (lin / per_group) * kGroupM;
    const int gsz = tiles_m - first_m < kGroupM ? tiles_m - first_m : kGroupM;
    const int in_group = lin % per_group;
    m0 = (first_m + in_group % gsz) * BM;
    n0 = (in_group / gsz) * BN;
}

// one residual element (bf16 or fp32 stream)
__device__ __forceinline__ float gemm_residual_at(const GemmArgs& g, const bf16_t* R, int64_t idx) {
    return g.res_f32 ? reinterpret_cast<const float*>(R)[idx] : bf16_to_f32(R[idx]);
}

// Epilogue for one accumulator fragment: the lane owns C[m][n .. n+3] (operands were swapped so that the four
// registers are consecutive N).  bias -> activation (or SwiGLU) -> residual -> store.
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& g, int bz, int m, int n, const f32x4_t& acc) {
    if (m >= g.M || n >= g.N) return;
    if (g.out_rows) {  // scatter epilogue: destination (and residual) row from the map; negative = dropped
        m = g.out_rows[m];
        if (m < 0) return;
    }
    const bf16_t* __restrict__ bias = g.bias;
    const bf16_t* __restrict__ R = g.residual ? g.residual + (int64_t)bz * g.strideR * (g.res_f32 ? 2 : 1) : nullptr;
    const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (g.fp8) {  // per-tensor dequantisation scales (device scalars)
        const float alpha = (*g.scale_a) * (*g.scale_w);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= alpha;
    }
    if ((g.N & 3) == 0) {
        if (bias) {
            const uint2 b2 = *reinterpret_cast<const uint2*>(bias + n);
            v[0] += bf16_to_f32((bf16_t)(b2.x & 0xffff));
            v[1] += bf16_to_f32((bf16_t)(b2.x >> 16));
            v[2] += bf16_to_f32((bf16_t)(b2.y & 0xffff));
            v[3] += bf16_to_f32((bf16_t)(b2.y >> 16));
        }
        if (ACT == ACT_SWIGLU) {
            // rows interleaved (gate_j, up_j): out[j] = silu(gate_j) * up_j, two outputs per lane
            const float o0 = (v[0] / (1.0f + __expf(-v[0]))) * v[1];
            const float o1 = (v[2] / (1.0f + __expf(-v[2]))) * v[3];
            const int64_t o = (int64_t)m * g.ldc + (n >> 1);
            if (OUT_F32) {
                float* C = static_cast<float*>(g.C) + (int64_t)bz * g.strideC;
                *reinterpret_cast<float2*>(C + o) = make_float2(o0, o1);
            } else {
                bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
                *reinterpret_cast<uint32_t*>(C + o) = pack_bf16x2(o0, o1);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gemm_act(v[j], ACT);
        if (R) {
            if (g.res_f32) {  // fp32 residual stream
                const float4 r4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(R) + rrow * g.ldr + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            } else {
                const uint2 r2 = *reinterpret_cast<const uint2*>(R + rrow * g.ldr + n);
                v[0] += bf16_to_f32((bf16_t)(r2.x & 0xffff));
                v[1] += bf16_to_f32((bf16_t)(r2.x >> 16));
                v[2] += bf16_to_f32((bf16_t)(r2.y & 0xffff));
                v[3] += bf16_to_f32((bf16_t)(r2.y >> 16));
            }
        }
        const int64_t o = g.c_panel ? (int64_t)(n >> 6) * g.c_panel + (int64_t)m * 64 + (n & 63) : (int64_t)m * g.ldc + n;
        if (g.out_fp8) {  // e4m3 output with the consumer's calibrated per-tensor scale (mlp1 -> mlp2)
            const float inv = 1.0f / (*g.scale_out);
            uint32_t w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
            *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(g.C) + (int64_t)bz * g.strideC + o) = w;
        } else if (OUT_F32) {
            float* C = static_cast<float*>(g.C) + (int64_t)bz * g.strideC;
            *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
            *reinterpret_cast<uint2*>(C + o) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    } else {  // ragged N: scalar tail (never on the hot shapes)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= g.N) break;
            float x = v[j] + (bias ? bf16_to_f32(bias[n + j]) : 0.0f);
            x = gemm_act(x, ACT);
            if (R) x += gemm_residual_at(g, R, rrow * g.ldr + n + j);
            const int64_t o = (int64_t)m * g.ldc + n + j;
            if (OUT_F32)
                (static_cast<float*>(g.C) + (int64_t)bz * g.strideC)[o] = x;
            else
                (static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC)[o] = f32_to_bf16(x);
        }
    }
}

}  // namespace ivlm
