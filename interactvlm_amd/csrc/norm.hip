// Row normalisations for gfx950 (bf16 or fp32 in/out, fp32 statistics), one wave per row, 16-byte loads.
// The residual streams of the three transformers are fp32 (GEMM residual epilogues write fp32): the norms read fp32 rows and
// write the bf16 MFMA operand (or fp32 again where the normalised row is itself a stream: CLIP pre_layrnorm, SAM decoder).
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

template <bool RMS, int MAXC, bool GELU, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void norm_kernel(const void* __restrict__ xv, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ b, void* __restrict__ yv, int64_t rows,
                                                   int cols, float eps, const int32_t* __restrict__ out_rows,
                                                   const float* __restrict__ fp8_scale, int y_split) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols >> 3;  // cols % 8 == 0
    const uint4* xr = reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(xv) + row * cols);
    const float4* xr4 = reinterpret_cast<const float4*>(static_cast<const float*>(xv) + row * cols);
    float v[MAXC][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int idx = c * 64 + lane;
        if (idx < nchunk) {
            if (XF32) {
                const float4 a = xr4[2 * idx], bq = xr4[2 * idx + 1];
                v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
                v[c][4] = bq.x; v[c][5] = bq.y; v[c][6] = bq.z; v[c][7] = bq.w;
            } else {
                unpack8(xr[idx], v[c]);
            }
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
    const int64_t orow = out_rows ? (int64_t)out_rows[row] : row;
    // y_split (bf16 output only): the row is written as [hi(cols) | lo(cols)], the A operand of an fp32-activation GEMM
    uint4* yr = reinterpret_cast<uint4*>(static_cast<bf16_t*>(yv) + orow * ((y_split & 1) ? 2 * cols : cols));
    float4* yr4 = reinterpret_cast<float4*>(static_cast<float*>(yv) + orow * cols);
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
                for (int j = 0; j < 8; ++j)  // HF casts the normalised row to the input dtype before the weight multiply
                    o[j] = (XF32 ? v[c][j] * rstd : bf16_to_f32(f32_to_bf16(v[c][j] * rstd))) * wv[j];
            } else {
                float bv[8];
                unpack8(br[idx], bv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[j] = (v[c][j] - mean) * rstd * wv[j] + bv[j];
                    if (GELU) o[j] = 0.5f * o[j] * (1.0f + erff(o[j] * 0.70710678118654752f));
                }
            }
            if (!YF32 && fp8_scale) {  // e4m3 operand of an fp8 GEMM, calibrated per-tensor scale
                const float inv = 1.0f / *fp8_scale;
                float cl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) cl[j] = fminf(fmaxf(o[j] * inv, -448.0f), 448.0f);
                uint32_t a = 0, bq = 0;
                a = __builtin_amdgcn_cvt_pk_fp8_f32(cl[0], cl[1], a, false);
                a = __builtin_amdgcn_cvt_pk_fp8_f32(cl[2], cl[3], a, true);
                bq = __builtin_amdgcn_cvt_pk_fp8_f32(cl[4], cl[5], bq, false);
                bq = __builtin_amdgcn_cvt_pk_fp8_f32(cl[6], cl[7], bq, true);
                reinterpret_cast<uint2*>(static_cast<uint8_t*>(yv) + orow * cols)[idx] = make_uint2(a, bq);
            } else if (YF32) {
                yr4[2 * idx] = make_float4(o[0], o[1], o[2], o[3]);
                yr4[2 * idx + 1] = make_float4(o[4], o[5], o[6], o[7]);
            } else if (y_split == 2) {  // IEEE fp16 row (operand of an fp16 GEMM)
                yr[idx] = make_uint4(pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3]), pack_f16x2(o[4], o[5]), pack_f16x2(o[6], o[7]));
            } else if (y_split) {  // 1: [hi | lo] bf16 halves, 3: [hi | lo] IEEE halves
                uint32_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_16x2(o[2 * j], o[2 * j + 1], h[j], l[j], y_split == 3);
                yr[idx] = make_uint4(h[0], h[1], h[2], h[3]);
                yr[nchunk + idx] = make_uint4(l[0], l[1], l[2], l[3]);
            } else {
                yr[idx] = pack8(o);
            }
        }
    }
}

}  // namespace

template <bool RMS, int MAXC, bool GELU>
static void launch_norm(const void* x, int x_f32, const bf16_t* w, const bf16_t* b, void* y, int y_f32, int64_t rows, int cols,
                        float eps, hipStream_t st, const int32_t* out_rows = nullptr, const float* fp8_scale = nullptr) {
    const unsigned grid = (unsigned)((rows + 3) / 4);
    // (y kind: 0 bf16, 1 fp32, 2 split bf16 [hi | lo], 4 fp16, 5 split fp16 [hi | lo])
    const int y_split = y_f32 == 2 ? 1 : (y_f32 == 4 ? 2 : (y_f32 == 5 ? 3 : 0));
    if (x_f32 && y_f32 == 1) norm_kernel<RMS, MAXC, GELU, true, true><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps, out_rows, fp8_scale, 0);
    else if (x_f32) norm_kernel<RMS, MAXC, GELU, true, false><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps, out_rows, fp8_scale, y_split);
    else if (y_f32 == 1) norm_kernel<RMS, MAXC, GELU, false, true><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps, out_rows, fp8_scale, 0);
    else norm_kernel<RMS, MAXC, GELU, false, false><<<grid, 256, 0, st>>>(x, w, b, y, rows, cols, eps, out_rows, fp8_scale, y_split);
}

int layernorm(const void* x, int x_f32, const bf16_t* w, const bf16_t* b, void* y, int y_f32, int64_t rows, int cols, float eps,
              hipStream_t st, int gelu, const int32_t* out_rows, const float* fp8_scale) {
    if (!x || !w || !b || !y || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    if ((cols & 7) || cols > kMaxChunks * 512) return IVLM_ERR_UNSUPPORTED;
    if (gelu) {
        if (cols > 4 * 512) return IVLM_ERR_UNSUPPORTED;
        launch_norm<false, 4, true>(x, x_f32, w, b, y, y_f32, rows, cols, eps, st, out_rows);
    } else if (cols <= 4 * 512) {
        launch_norm<false, 4, false>(x, x_f32, w, b, y, y_f32, rows, cols, eps, st, out_rows, fp8_scale);
    } else {
        launch_norm<false, kMaxChunks, false>(x, x_f32, w, b, y, y_f32, rows, cols, eps, st, out_rows);
    }
    return ivlm_launch_status();
}

int rmsnorm(const void* x, int x_f32, const bf16_t* w, void* y, int y_f32, int64_t rows, int cols, float eps, hipStream_t st,
            const float* fp8_scale) {
    if (!x || !w || !y || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    if ((cols & 7) || cols > kMaxChunks * 512) return IVLM_ERR_UNSUPPORTED;
    if (fp8_scale && y_f32) return IVLM_ERR_INVALID_ARG;  // (fp8_scale: y is e4m3 bytes of the normalised row / *fp8_scale)
    if (cols <= 4 * 512) launch_norm<true, 4, false>(x, x_f32, w, nullptr, y, y_f32, rows, cols, eps, st, nullptr, fp8_scale);
    else launch_norm<true, kMaxChunks, false>(x, x_f32, w, nullptr, y, y_f32, rows, cols, eps, st, nullptr, fp8_scale);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {

int ivlm_layernorm(const void* x, int x_dtype, const void* w, const void* b, void* y, int y_dtype, int64_t rows, int cols,
                   float eps, int gelu, const int32_t* out_rows, const float* fp8_scale, ivlm_stream_t stream) {
    ivlm_enter();
    if ((y_dtype == IVLM_FP8) != (fp8_scale != nullptr)) return IVLM_ERR_INVALID_ARG;
    if (y_dtype == IVLM_FP8 && (gelu || cols > 2048)) return IVLM_ERR_UNSUPPORTED;
    return ivlm::layernorm(x, x_dtype == IVLM_F32, static_cast<const bf16_t*>(w), static_cast<const bf16_t*>(b), y,
                           y_dtype == IVLM_F32 ? 1 : (y_dtype == IVLM_BF16_SPLIT ? 2 : (y_dtype == IVLM_F16 ? 4 : (y_dtype == IVLM_F16_SPLIT ? 5 : 0))),
                           rows, cols, eps, ivlm_stream(stream), gelu, out_rows, fp8_scale);
}

int ivlm_rmsnorm_fp8(const void* x, int x_dtype, const void* w, void* y, int64_t rows, int cols, float eps, const float* fp8_scale,
                     ivlm_stream_t stream) {
    ivlm_enter();
    if (!fp8_scale) return IVLM_ERR_INVALID_ARG;
    return ivlm::rmsnorm(x, x_dtype == IVLM_F32, static_cast<const bf16_t*>(w), y, 0, rows, cols, eps, ivlm_stream(stream), fp8_scale);
}

int ivlm_rmsnorm(const void* x, int x_dtype, const void* w, void* y, int y_dtype, int64_t rows, int cols, float eps,
                 ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::rmsnorm(x, x_dtype == IVLM_F32, static_cast<const bf16_t*>(w), y,
                         y_dtype == IVLM_F32 ? 1 : (y_dtype == IVLM_BF16_SPLIT ? 2 : (y_dtype == IVLM_F16 ? 4 : (y_dtype == IVLM_F16_SPLIT ? 5 : 0))),
                         rows, cols, eps, ivlm_stream(stream));
}

}  // extern "C"
