// Optional heads of the reference's ModifiedSAM (model/InteractVLM.py:20-44; off in every released configuration):
//   * UncertaintyModule (model/components.py:40-78): a per-pixel 256 -> 64 -> 16 -> 1 MLP (ReLU, ReLU, Softplus) over the SAM image
//     embeddings.  The module casts its input to bf16 and runs inside the bf16 model: every linear's fp32 accumulation is rounded
//     to bf16 once (bias included), and so are the activations - restated here with the same rounding points.
//   * the bilinear resize (align_corners=False) of its map to the label / original size (InteractVLM.py:446-447, 615-616), with
//     fp32 interpolation weights (ATen's GPU kernel; its CPU kernel rounds them to bf16 - see oracle/nn.py uncertainty_resize).
// Both are a few microseconds of VALU work on 16384 pixels: one thread per pixel, weights broadcast from LDS.
#include "bilinear.h"

namespace ivlm {
namespace {

__device__ __forceinline__ float bf16_round(float x) { return bf16_to_f32(f32_to_bf16(x)); }

constexpr int kC = 256, kH1 = 64, kH2 = 16;

__global__ __launch_bounds__(256) void uncertainty_mlp_kernel(const float* __restrict__ emb, int64_t rows, const bf16_t* __restrict__ w1,
                                                              const bf16_t* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                              const bf16_t* __restrict__ b2, const bf16_t* __restrict__ w3,
                                                              const bf16_t* __restrict__ b3, float* __restrict__ out) {
    __shared__ float W1t[kC][kH1];   // [k][j] = linear1.weight[j][k]
    __shared__ float W2t[kH1][kH2];  // [j][i] = linear2.weight[i][j]
    __shared__ float W3[kH2], B1[kH1], B2[kH2];
    for (int i = threadIdx.x; i < kC * kH1; i += 256) W1t[i % kC][i / kC] = bf16_to_f32(w1[i]);
    for (int i = threadIdx.x; i < kH1 * kH2; i += 256) W2t[i % kH1][i / kH1] = bf16_to_f32(w2[i]);
    if (threadIdx.x < kH1) B1[threadIdx.x] = bf16_to_f32(b1[threadIdx.x]);
    if (threadIdx.x < kH2) {
        B2[threadIdx.x] = bf16_to_f32(b2[threadIdx.x]);
        W3[threadIdx.x] = bf16_to_f32(w3[threadIdx.x]);
    }
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float4* x4 = reinterpret_cast<const float4*>(emb + r * kC);
    float h1[kH1];
#pragma unroll
    for (int j = 0; j < kH1; ++j) h1[j] = 0.f;
    for (int k4 = 0; k4 < kC / 4; ++k4) {
        const float4 xv = x4[k4];
        const float xs[4] = {bf16_round(xv.x), bf16_round(xv.y), bf16_round(xv.z), bf16_round(xv.w)};  // x.bfloat16()
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4* wrow = reinterpret_cast<const float4*>(&W1t[k4 * 4 + kk][0]);
#pragma unroll
            for (int j4 = 0; j4 < kH1 / 4; ++j4) {
                const float4 wv = wrow[j4];
                h1[j4 * 4 + 0] = fmaf(xs[kk], wv.x, h1[j4 * 4 + 0]);
                h1[j4 * 4 + 1] = fmaf(xs[kk], wv.y, h1[j4 * 4 + 1]);
                h1[j4 * 4 + 2] = fmaf(xs[kk], wv.z, h1[j4 * 4 + 2]);
                h1[j4 * 4 + 3] = fmaf(xs[kk], wv.w, h1[j4 * 4 + 3]);
            }
        }
    }
    float h2[kH2];
#pragma unroll
    for (int i = 0; i < kH2; ++i) h2[i] = 0.f;
#pragma unroll
    for (int j = 0; j < kH1; ++j) {
        const float a0 = bf16_round(h1[j] + B1[j]), a = a0 < 0.f ? 0.f : a0;  // relu(linear1) on bf16 values (NaN stays NaN)
#pragma unroll
        for (int i = 0; i < kH2; ++i) h2[i] = fmaf(a, W2t[j][i], h2[i]);
    }
    float y = 0.f;
#pragma unroll
    for (int i = 0; i < kH2; ++i) {
        const float a2 = bf16_round(h2[i] + B2[i]);
        y = fmaf(a2 < 0.f ? 0.f : a2, W3[i], y);
    }
    y = bf16_round(y + bf16_to_f32(b3[0]));
    // nn.Softplus(beta = 1, threshold = 20) on the bf16 value, fp32 inside, rounded to bf16
    out[r] = bf16_round(y > 20.f ? y : log1pf(expf(y)));
}

// dst[n, y, x] = bilinear(src[n], align_corners = False) - fp32 taps and weights, result fp32 or (OUT_BF16) rounded to bf16
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, int h, int w, void* __restrict__ dst,
                                                              int oh, int ow) {
    using namespace ivlm_bilinear;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
    if (x >= ow) return;
    const Axis ay = axis_src(y, (float)h / (float)oh, h), ax = axis_src(x, (float)w / (float)ow, w);
    const float* p = src + (size_t)n * h * w;
    const float t = p[(size_t)ay.i0 * w + ax.i0] * ax.l0 + p[(size_t)ay.i0 * w + ax.i1] * ax.l1;
    const float b = p[(size_t)ay.i1 * w + ax.i0] * ax.l0 + p[(size_t)ay.i1 * w + ax.i1] * ax.l1;
    const float v = t * ay.l0 + b * ay.l1;
    const size_t o = ((size_t)n * oh + y) * ow + x;
    if (OUT_BF16) static_cast<bf16_t*>(dst)[o] = f32_to_bf16(v);
    else static_cast<float*>(dst)[o] = v;
}

}  // namespace
}  // namespace ivlm

extern "C" int ivlm_uncertainty_mlp(const float* embeddings, int64_t rows, const void* w1, const void* b1, const void* w2, const void* b2,
                                    const void* w3, const void* b3, float* out, ivlm_stream_t stream) {
    ivlm_enter();
    IVLM_CHECK_ARG(embeddings && w1 && b1 && w2 && b2 && w3 && b3 && out && rows > 0);
    IVLM_CHECK_ARG(((uintptr_t)embeddings & 15) == 0);
    using namespace ivlm;
    uncertainty_mlp_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, ivlm_stream(stream)>>>(
        embeddings, rows, static_cast<const bf16_t*>(w1), static_cast<const bf16_t*>(b1), static_cast<const bf16_t*>(w2),
        static_cast<const bf16_t*>(b2), static_cast<const bf16_t*>(w3), static_cast<const bf16_t*>(b3), out);
    return ivlm_launch_status();
}

extern "C" int ivlm_resize_bilinear(const float* src, int n, int h, int w, void* dst, int dst_dtype, int oh, int ow, ivlm_stream_t stream) {
    ivlm_enter();
    IVLM_CHECK_ARG(src && dst && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && oh <= 65535 && n <= 65535);
    if (dst_dtype != IVLM_F32 && dst_dtype != IVLM_BF16) return IVLM_ERR_UNSUPPORTED;
    using namespace ivlm;
    const dim3 grid((ow + 255) / 256, oh, n);
    if (dst_dtype == IVLM_BF16) resize_bilinear_kernel<true><<<grid, 256, 0, ivlm_stream(stream)>>>(src, h, w, dst, oh, ow);
    else resize_bilinear_kernel<false><<<grid, 256, 0, ivlm_stream(stream)>>>(src, h, w, dst, oh, ow);
    return ivlm_launch_status();
}
