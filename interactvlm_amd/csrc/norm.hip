// Row normalisations for gfx950 (bf16 in/out, fp32 statistics), one wave per row, 16-byte loads.
//   LayerNorm  : SAM ViT (eps 1e-6, image_encoder.py:158,172), SAM decoder (eps 1e-5, transformer.py:134-144),
//                LayerNorm2d over channels when activations are kept NHWC (common.py:32-42), CLIP (HF, 1e-5)
//   RMSNorm    : HF LlamaRMSNorm (x * rsqrt(mean(x^2)+eps) cast to bf16, THEN times weight)
// HBM-bound: one read + one write of the row; the row lives in registers between the passes.
#include "kernels.h"

namespace ivlm {
namespace {

constexpr int kMaxChunks = 16;  // 16 chunks x 8 elements x 64 lanes = 8192 columns max

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
}

template <bool RMS, int MAXC, bool GELU>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int64_t rows,
                                                   int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols >> 3;  // cols % 8 == 0
    const uint4* xr = reinterpret_cast<const uint4*>(x + row * cols);
    float v[MAXC][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 64 + lane;
        if (idx < nchunk) {
            unpack8(xr[idx], v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += RMS ? v[c][j] * v[c][j] : v[c][j];
        }
    }
    s = wave_sum(s);
    float mean = 0.0f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c * 64 + lane < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[c][j] - mean;
                    q += d * d;
                }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    uint4* yr = reinterpret_cast<uint4*>(y + row * cols);
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    const uint4* br = reinterpret_cast<const uint4*>(b);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 64 + lane;
        if (idx < nchunk) {
            float wv[8], o[8];
            unpack8(wr[idx], wv);
            if (RMS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = bf16_to_f32(f32_to_bf16(v[c][j] * rstd)) * wv[j];
            } else {
                float bv[8];
                unpack8(br[idx], bv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[j] = (v[c][j] - mean) * rstd * wv[j] + bv[j];
                    if (GELU) o[j] = 0.5f * o[j] * (1.0f + erff(o[j] * 0.70710678118654752f));
                }
            }
            yr[idx] = pack8(o);
        }
    }
}

}  // namespace

int layernorm_bf16(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int64_t rows, int cols, float eps,
                   hipStream_t st, int gelu) {
    if (!x || !w || !b || !y || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    if ((cols & 7) || cols > kMaxChunks * 512) return IVLM_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (gelu) {
        if (cols > 4 * 512) return IVLM_ERR_UNSUPPORTED;
        norm_kernel<false, 4, true><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps);
    } else if (cols <= 4 * 512) {
        norm_kernel<false, 4, false><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps);
    } else {
        norm_kernel<false, kMaxChunks, false><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps);
    }
    return ivlm_launch_status();
}

int rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int cols, float eps, hipStream_t st) {
    if (!x || !w || !y || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    if ((cols & 7) || cols > kMaxChunks * 512) return IVLM_ERR_UNSUPPORTED;
    if (cols <= 4 * 512)
        norm_kernel<true, 4, false><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(x, w, nullptr, y, rows, cols, eps);
    else
        norm_kernel<true, kMaxChunks, false><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(x, w, nullptr, y, rows, cols, eps);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {

int ivlm_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int64_t rows, int cols, float eps,
                        int gelu, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::layernorm_bf16(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w),
                                static_cast<const bf16_t*>(b), static_cast<bf16_t*>(y), rows, cols, eps,
                                ivlm_stream(stream), gelu);
}

int ivlm_rmsnorm_bf16(const void* x, const void* w, void* y, int64_t rows, int cols, float eps,
                      ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::rmsnorm_bf16(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), static_cast<bf16_t*>(y),
                              rows, cols, eps, ivlm_stream(stream));
}

}  // extern "C"
