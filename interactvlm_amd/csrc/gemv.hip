// Skinny GEMM / GEMV for gfx950: out[M<=8, N] = act(x[M,K] . W[N,K]^T + bias) + residual.
//
// Batch-1 autoregressive decode (HF greedy search driven by InteractVLM.evaluate, model/InteractVLM.py:524-531)
// is pure weight streaming: 13.5 GB of bf16 weights per generated token for LLaMA-7B.  No LDS round trip
// (nothing is shared between waves), no MFMA (M is 1): each wave owns two weight rows at a time, streams
// them with non-temporal 16-byte loads (8 in flight per lane), dots them against the L1-resident
// activation rows and reduces with a wave butterfly.  Two rows per wave also lets the SwiGLU epilogue
// (row-interleaved gate/up weights) complete inside the wave.
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
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

template <int M, bool RMS>
__global__ __launch_bounds__(256) void gemv_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nchunk = g.K >> 3;  // 16-byte chunks per row
    const int npairs = (g.N + 1) >> 1;
    for (int pr = wave_global; pr < npairs; pr += nwaves) {
        const int n0 = 2 * pr, n1 = (n0 + 1 < g.N) ? n0 + 1 : n0;
        const u32x4_t* w0 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n0 * g.ldw);
        const u32x4_t* w1 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n1 * g.ldw);
        float a0[M], a1[M], ssq[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a0[m] = a1[m] = ssq[m] = 0.0f;
        for (int c = lane; c < nchunk; c += 256) {  // 4 chunks x 2 rows = 8 loads in flight per lane
            u32x4_t wa[4], wb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c + u * 64;
                if (cc < nchunk) {
                    wa[u] = __builtin_nontemporal_load(w0 + cc);
                    wb[u] = __builtin_nontemporal_load(w1 + cc);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c + u * 64;
                if (cc < nchunk) {
                    u32x4_t gv;
                    if (RMS) gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + cc);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        u32x4_t xv = *(reinterpret_cast<const u32x4_t*>(g.A + (int64_t)m * g.lda) + cc);
                        if (RMS) {  // x * gamma in fp32 (rounded once to bf16); the row scale rstd is applied at the end
                            ssq[m] += dot8(xv, xv);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                xv[j] = pack_bf16x2(__uint_as_float(xv[j] << 16) * __uint_as_float(gv[j] << 16),
                                                    __uint_as_float(xv[j] & 0xffff0000u) *
                                                        __uint_as_float(gv[j] & 0xffff0000u));
                        }
                        a0[m] += dot8(wa[u], xv);
                        a1[m] += dot8(wb[u], xv);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            a0[m] = wave_sum(a0[m]);
            a1[m] = wave_sum(a1[m]);
            if (RMS) {
                const float rstd = rsqrtf(wave_sum(ssq[m]) / (float)g.K + g.rms_eps);
                a0[m] *= rstd;
                a1[m] *= rstd;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v0 = a0[m] + (g.bias ? bf16_to_f32(g.bias[n0]) : 0.0f);
                float v1 = a1[m] + (g.bias ? bf16_to_f32(g.bias[n1]) : 0.0f);
                if (g.act == ACT_SWIGLU) {
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
                    v0 += bf16_to_f32(g.residual[rrow * g.ldr + n0]);
                    v1 += bf16_to_f32(g.residual[rrow * g.ldr + n1]);
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
    }
}

// first index of the row maximum (torch.argmax tie rule), one block per row
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, int cols, int32_t* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const float* r = x + (int64_t)blockIdx.x * cols;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < cols; i += 256) {
        const float v = r[i];
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    }
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
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        out[blockIdx.x] = bi;
    }
}

}  // namespace

int gemv_bf16(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.M > kMaxM || g.N <= 0 || g.K <= 0) return IVLM_ERR_INVALID_ARG;
    if ((g.K & 7) || (g.lda & 7) || (g.ldw & 7) || g.batch != 1) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    const int npairs = (g.N + 1) / 2;
    int blocks = (npairs + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    switch (g.M) {
#define IVLM_GEMV_CASE(MM)                                                   \
    case MM:                                                                 \
        if (g.rms_w) gemv_kernel<MM, true><<<blocks, 256, 0, st>>>(g);       \
        else gemv_kernel<MM, false><<<blocks, 256, 0, st>>>(g);              \
        break;
        IVLM_GEMV_CASE(1) IVLM_GEMV_CASE(2) IVLM_GEMV_CASE(3) IVLM_GEMV_CASE(4)
        IVLM_GEMV_CASE(5) IVLM_GEMV_CASE(6) IVLM_GEMV_CASE(7) IVLM_GEMV_CASE(8)
#undef IVLM_GEMV_CASE
    }
    return ivlm_launch_status();
}

int argmax_f32(const float* x, int rows, int cols, int32_t* out, hipStream_t st) {
    if (!x || !out || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    argmax_kernel<<<rows, 256, 0, st>>>(x, cols, out);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_argmax_f32(const float* x, int rows, int cols, int32_t* out, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::argmax_f32(x, rows, cols, out, ivlm_stream(stream));
}
