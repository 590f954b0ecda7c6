// Batch-1 decode GEMV over LOSSLESSLY packed bf16 weights ("bf12": 12 bits per weight) for gfx950.
//
// The decode step of the greedy search under InteractVLM.evaluate (model/InteractVLM.py:524-531) is pure weight streaming: 13.5 GB of
// bf16 weights per generated token for LLaMA-7B, 60 % of an image's time.  The 8 exponent bits of a weight matrix carry ~2.7 bits
// of information: inside one row nearly every weight lies within 15 binades of the row's largest (a Gaussian row: all but 1e-4).
// So a row is stored as
//     P  [K]   bytes   sign << 7 | mantissa (7 bits)
//     E  [K/2] bytes   two 4-bit codes: exponent field - ebase[row] in 1 .. 15; 0 = the weight is zero or lies outside the window
//     ebase    int32   per row (row maximum of the exponent field - 15, clamped at 0)
//     patches  CSR     (column, bf16 value) of the nonzero weights outside the window (subnormals, the far tail)
// = 1.5 bytes per weight instead of 2, and EVERY weight is reconstructed bit for bit (ivlm_unpack_bf12 is the proof: tests).
// The kernel never rebuilds the bf16 value: a lane turns (P byte, code) into the fp32 number 1.m x 2^(code - 127) with four integer
// operations (byte permute, mask, bit-field extract, shift-add; code 0 gives exactly 0.0), multiplies it with x * 2^100 (x is staged
// once per block in LDS, pre-scaled by an exact power of two so that the tiny products stay normal fp32 numbers) and the row sum is
// scaled back by 2^(ebase - 100) - exact power-of-two scalings, so the arithmetic is that of gemv1_kernel (bf16 weight x fp32
// activation products, exact; fp32 accumulation) up to the summation order.  Same shape as gemv1_kernel: 1024-thread blocks, one row
// per wave, non-temporal loads (16 B of P + 8 B of E per lane and step = 16 weights), fused RMSNorm prologue, SwiGLU / residual
// epilogues.
#include <algorithm>

#include "kernels.h"

namespace ivlm {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

constexpr int kWaves = 16;
constexpr int kXScaleExp = 100;  // x is staged as x * 2^100

struct P12 {
    const uint8_t* P;   // [N][ldp]
    const uint8_t* E;   // [N][lde]
    int64_t ldp, lde;
    const int32_t* ebase;      // [N]
    const int32_t* patch_ptr;  // [N+1]
    const int32_t* patch_col;
    const bf16_t* patch_val;
};

// one weight: [b_k, b_k, 0, 0] puts the byte's sign on bit 31 and its mantissa on bits 22 .. 16 (bits 30 .. 23 are masked away); the code
// becomes the exponent field: 1.m x 2^(code - 127); code 0 (with P = 0) is exactly 0.0.  Four integer operations + the FMA.
// (The packed fp32 FMA - two products per instruction - was measured SLOWER: 3.07 vs 2.56 ms per token; register-pair moves.)
template <int K>
__device__ __forceinline__ float w_p12(uint32_t pw, uint32_t e16) {
    const uint32_t t = __builtin_amdgcn_perm(0u, pw, (uint32_t)((K << 24) | (K << 16) | 0x0c0cu));
    const uint32_t f0 = t & 0x807f0000u;
    const uint32_t n = __builtin_amdgcn_ubfe(e16, 4 * K, 4);
    return __uint_as_float((n << 23) + f0);
}

// four weights of one 32-bit word of P against four x values; e16: their four codes in bits 0 .. 15
__device__ __forceinline__ float dot4_p12(uint32_t pw, uint32_t e16, const f32x4v_t& x, float acc) {
    acc = fmaf(w_p12<0>(pw, e16), x[0], acc);
    acc = fmaf(w_p12<1>(pw, e16), x[1], acc);
    acc = fmaf(w_p12<2>(pw, e16), x[2], acc);
    acc = fmaf(w_p12<3>(pw, e16), x[3], acc);
    return acc;
}

__device__ __forceinline__ float act1(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return x < 0.0f ? 0.0f : x;  // (torch.relu semantics: a NaN stays a NaN; fmaxf would turn it into 0)
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

template <bool RMS>
__global__ __launch_bounds__(64 * kWaves, 8) void gemv1_p12_kernel(GemmArgs g, P12 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[kWaves];
    __shared__ float s_val[kWaves];
    constexpr int U = 4;  // 4 x (16 + 8) bytes in flight per lane: one pass over a 4096-weight row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;  // fp32 x chunks of 8 elements (two 16-byte planes)
    const int nw16 = g.K >> 4;    // 16-weight chunks per row
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);
    const int row = blockIdx.x * kWaves + wave;
    const bool live = row < g.N;
    const int rr = live ? row : g.N - 1;
    const u32x4_t* pp = reinterpret_cast<const u32x4_t*>(p.P + (int64_t)rr * p.ldp);
    const u32x2_t* ep = reinterpret_cast<const u32x2_t*>(p.E + (int64_t)rr * p.lde);
    u32x4_t w[U];
    u32x2_t e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = min(lane + 64 * u, nw16 - 1);
        w[u] = __builtin_nontemporal_load(pp + c);
        e[u] = __builtin_nontemporal_load(ep + c);
    }
    const int eb = p.ebase[rr];
    const int p0 = p.patch_ptr[rr], p1 = p.patch_ptr[rr + 1];
    // ---- stage x * 2^100 (x * gamma * 2^100) in LDS, sum(x^2) of the unscaled row ----
    const float xs = __builtin_ldexpf(1.0f, kXScaleExp);
    float ssq = 0.0f;
    for (int c = threadIdx.x; c < nchunk; c += 64 * kWaves) {
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
        xf[c] = xa * xs;
        xf[nchunk + c] = xb * xs;
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
                if (c + 64 * u < nw16) {
                    w[u] = __builtin_nontemporal_load(pp + c + 64 * u);
                    e[u] = __builtin_nontemporal_load(ep + c + 64 * u);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + 64 * u;
            if (cc < nw16) {  // weights 16 cc .. 16 cc + 15 against x chunks 2 cc (planes a | b) and 2 cc + 1
                acc = dot4_p12(w[u][0], e[u][0], xf[2 * cc], acc);
                acc = dot4_p12(w[u][1], e[u][0] >> 16, xf[nchunk + 2 * cc], acc);
                acc = dot4_p12(w[u][2], e[u][1], xf[2 * cc + 1], acc);
                acc = dot4_p12(w[u][3], e[u][1] >> 16, xf[nchunk + 2 * cc + 1], acc);
            }
        }
    }
    // the nonzero weights outside the row's exponent window (exact bf16 values; usually none, on average < 1 per row)
    float pacc = 0.0f;
    for (int i = p0 + lane; i < p1; i += 64) {
        const int col = p.patch_col[i];
        const int ch = col >> 3, wi = col & 7;
        const float xv = reinterpret_cast<const float*>(xf)[((wi < 4 ? ch : nchunk + ch) << 2) + (wi & 3)];
        pacc = fmaf(bf16_to_f32(p.patch_val[i]), xv, pacc);
    }
    acc = __builtin_ldexpf(wave_sum(acc), eb - kXScaleExp) + __builtin_ldexpf(wave_sum(pacc), -kXScaleExp);
    if (RMS) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < kWaves; ++i) q += s_red[i];
        acc *= rsqrtf(q / (float)g.K + g.rms_eps);
    }
    float v = acc + ((g.bias && live) ? bf16_to_f32(g.bias[row]) : 0.0f);
    if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: the even wave finishes the pair
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
    v = act1(v, g.act);
    if (g.residual) v += g.res_f32 ? reinterpret_cast<const float*>(g.residual)[row] : bf16_to_f32(g.residual[row]);
    if (g.out_f32) static_cast<float*>(g.C)[row] = v;
    else static_cast<bf16_t*>(g.C)[row] = f32_to_bf16(v);
}

// exact reconstruction of the bf16 matrix (the losslessness check of the tests; not on the path): one thread per weight
__global__ __launch_bounds__(256) void unpack_p12_kernel(P12 p, int N, int K, bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / K), c = (int)(i - (int64_t)r * K);
        const uint32_t b = p.P[(int64_t)r * p.ldp + c];
        const uint32_t code = (p.E[(int64_t)r * p.lde + (c >> 1)] >> ((c & 1) * 4)) & 0xfu;
        out[i] = code ? (bf16_t)(((b & 0x80u) << 8) | ((uint32_t)(p.ebase[r] + (int)code) << 7) | (b & 0x7fu)) : (bf16_t)0;
    }
}
__global__ __launch_bounds__(256) void unpack_p12_patch_kernel(P12 p, int N, int K, bf16_t* __restrict__ out) {
    const int r = blockIdx.x;
    for (int i = p.patch_ptr[r] + threadIdx.x; i < p.patch_ptr[r + 1]; i += 256) out[(int64_t)r * K + p.patch_col[i]] = p.patch_val[i];
}

}  // namespace
}  // namespace ivlm

using namespace ivlm;

static bool p12_ok(const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase, const int32_t* patch_ptr,
                   const int32_t* patch_col, const void* patch_val, int N, int K) {
    return P && E && ebase && patch_ptr && patch_col && patch_val && N > 0 && K > 0 && !(K & 15) && ldp >= K && lde >= K / 2 &&
           !(ldp & 15) && !(lde & 7) && !(reinterpret_cast<uintptr_t>(P) & 15) && !(reinterpret_cast<uintptr_t>(E) & 7);
}

extern "C" int ivlm_gemv1_bf12(const float* x, const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase,
                               const int32_t* patch_ptr, const int32_t* patch_col, const void* patch_val, void* C, const void* bias,
                               const void* residual, int N, int K, int act, int out_f32, const void* rms_w, float rms_eps, int flags,
                               ivlm_stream_t stream) {
    ivlm_enter();
    if (!x || !C || !p12_ok(P, ldp, E, lde, ebase, patch_ptr, patch_col, patch_val, N, K)) return IVLM_ERR_INVALID_ARG;
    if ((size_t)K * 4 > 60 * 1024) return IVLM_ERR_UNSUPPORTED;  // the fp32 x image lives in LDS
    if (act == ACT_SWIGLU && ((N & 1) || residual)) return IVLM_ERR_UNSUPPORTED;
    GemmArgs g;
    g.A = reinterpret_cast<const bf16_t*>(x);
    g.a_f32 = 1;
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
    P12 p{static_cast<const uint8_t*>(P), static_cast<const uint8_t*>(E), ldp, lde, ebase, patch_ptr, patch_col,
          static_cast<const bf16_t*>(patch_val)};
    hipStream_t st = ivlm_stream(stream);
    static bool set0 = false, set1 = false;
    const dim3 grid((N + kWaves - 1) / kWaves), block(64 * kWaves);
    if (rms_w) {
        auto kfn = gemv1_p12_kernel<true>;
        if (!set1) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            set1 = true;
        }
        ivlm_launch(kfn, grid, block, (size_t)K * 4, st, g, p);
    } else {
        auto kfn = gemv1_p12_kernel<false>;
        if (!set0) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            set0 = true;
        }
        ivlm_launch(kfn, grid, block, (size_t)K * 4, st, g, p);
    }
    return ivlm_launch_status();
}

extern "C" int ivlm_unpack_bf12(const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase, const int32_t* patch_ptr,
                                const int32_t* patch_col, const void* patch_val, int N, int K, void* w_out, ivlm_stream_t stream) {
    ivlm_enter();
    if (!w_out || !p12_ok(P, ldp, E, lde, ebase, patch_ptr, patch_col, patch_val, N, K)) return IVLM_ERR_INVALID_ARG;
    P12 p{static_cast<const uint8_t*>(P), static_cast<const uint8_t*>(E), ldp, lde, ebase, patch_ptr, patch_col,
          static_cast<const bf16_t*>(patch_val)};
    hipStream_t st = ivlm_stream(stream);
    const int64_t total = (int64_t)N * K;
    unpack_p12_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 8), 256, 0, st>>>(p, N, K, static_cast<bf16_t*>(w_out));
    unpack_p12_patch_kernel<<<N, 256, 0, st>>>(p, N, K, static_cast<bf16_t*>(w_out));
    return ivlm_launch_status();
}
