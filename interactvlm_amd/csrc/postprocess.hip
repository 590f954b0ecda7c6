// Sam.postprocess_masks for gfx950 (model/segment_anything/modeling/sam.py:137-172):
//   bilinear(align_corners=False) h x w -> img x img, crop [:in_h,:in_w], bilinear -> (oh,ow), fp32.
// One pass, no img x img intermediate: every output pixel composes the two resizes on the fly
// (4 taps when the second resize is the identity, 16 otherwise).  The low-res source (256 KB per
// view) stays L2-resident; the kernel is bound by the fp32 store stream (16-byte stores).
#include "bilinear.h"

namespace {
using namespace ivlm_bilinear;

template <typename T, bool IDENT2, bool SIGMOID>
__global__ __launch_bounds__(256) void postprocess_kernel(const T* __restrict__ low, int h, int w, int img, int in_h,
                                                          int in_w, int oh, int ow, float* __restrict__ out,
                                                          const float* __restrict__ gt, float ignore_label) {
    const int n = blockIdx.z;
    const int y = blockIdx.y;
    const T* lowp = low + (size_t)n * h * w;
    float* orow = out + ((size_t)n * oh + y) * ow;
    const float* grow = (SIGMOID && gt) ? gt + ((size_t)n * oh + y) * ow : nullptr;  // sigmoid only where gt != ignore_label
    const float s1y = (float)h / (float)img, s1x = (float)w / (float)img;
    const float s2y = (float)in_h / (float)oh, s2x = (float)in_w / (float)ow;
    const Axis by = IDENT2 ? Axis{y, y, 1.0f, 0.0f} : axis_src(y, s2y, in_h);
    for (int x4 = (blockIdx.x * 256 + threadIdx.x) * 4; x4 < ow; x4 += gridDim.x * 256 * 4) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x4 + j;
            float val = 0.0f;
            if (x < ow) {
                if (IDENT2) {
                    val = stage1(lowp, h, w, s1y, s1x, y, x);
                } else {
                    const Axis bx = axis_src(x, s2x, in_w);
                    const float a = stage1(lowp, h, w, s1y, s1x, by.i0, bx.i0);
                    const float b = stage1(lowp, h, w, s1y, s1x, by.i0, bx.i1);
                    const float c = stage1(lowp, h, w, s1y, s1x, by.i1, bx.i0);
                    const float d = stage1(lowp, h, w, s1y, s1x, by.i1, bx.i1);
                    val = (a * bx.l0 + b * bx.l1) * by.l0 + (c * bx.l0 + d * bx.l1) * by.l1;
                }
                if (SIGMOID && (!grow || grow[x] != ignore_label)) val = sigmoid_f32(val);
            }
            r[j] = val;
        }
        if (x4 + 3 < ow && (ow & 3) == 0) {
            *reinterpret_cast<float4*>(orow + x4) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x4 + j < ow) orow[x4 + j] = r[j];
        }
    }
}

template <typename T>
int launch_postprocess(const T* low, int n, int h, int w, int img, int in_h, int in_w, int oh, int ow, int sig,
                       float* out, hipStream_t st, const float* gt = nullptr, float ignore_label = 0.0f) {
    const bool ident2 = (in_h == oh && in_w == ow);
    int gx = (ow + 1023) / 1024;
    dim3 grid(gx, oh, n);
    if (ident2) {
        if (sig) postprocess_kernel<T, true, true><<<grid, 256, 0, st>>>(low, h, w, img, in_h, in_w, oh, ow, out, gt, ignore_label);
        else postprocess_kernel<T, true, false><<<grid, 256, 0, st>>>(low, h, w, img, in_h, in_w, oh, ow, out, nullptr, 0.0f);
    } else {
        if (sig) postprocess_kernel<T, false, true><<<grid, 256, 0, st>>>(low, h, w, img, in_h, in_w, oh, ow, out, gt, ignore_label);
        else postprocess_kernel<T, false, false><<<grid, 256, 0, st>>>(low, h, w, img, in_h, in_w, oh, ow, out, nullptr, 0.0f);
    }
    return ivlm_launch_status();
}

}  // namespace

extern "C" int ivlm_postprocess_masks(const void* low, int dtype, int n, int h, int w, int img, int in_h, int in_w,
                                      int oh, int ow, int apply_sigmoid, float* out, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(low && out);
    IVLM_CHECK_ARG(n > 0 && h > 0 && w > 0 && img > 0 && oh > 0 && ow > 0);
    IVLM_CHECK_ARG(in_h > 0 && in_w > 0 && in_h <= img && in_w <= img && oh <= 65535 && n <= 65535);
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    if (dtype == IVLM_F32)
        return launch_postprocess(static_cast<const float*>(low), n, h, w, img, in_h, in_w, oh, ow, apply_sigmoid,
                                  out, st);
    if (dtype == IVLM_BF16)
        return launch_postprocess(static_cast<const bf16_t*>(low), n, h, w, img, in_h, in_w, oh, ow, apply_sigmoid,
                                  out, st);
    return IVLM_ERR_UNSUPPORTED;
}

// postprocess + the in-place sigmoid of InteractVLM.py:452-456 ('oafford' samples with 'HM' object views): sigmoid on the pixels
// whose ground-truth mask is not the ignore label, raw logits elsewhere.  gt f32 [n,oh,ow].
extern "C" int ivlm_postprocess_masks_valid(const void* low, int dtype, int n, int h, int w, int img, int in_h, int in_w, int oh,
                                            int ow, const float* gt, float ignore_label, float* out, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(low && out && gt);
    IVLM_CHECK_ARG(n > 0 && h > 0 && w > 0 && img > 0 && oh > 0 && ow > 0);
    IVLM_CHECK_ARG(in_h > 0 && in_w > 0 && in_h <= img && in_w <= img && oh <= 65535 && n <= 65535);
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    if (dtype == IVLM_F32)
        return launch_postprocess(static_cast<const float*>(low), n, h, w, img, in_h, in_w, oh, ow, 1, out, st, gt, ignore_label);
    if (dtype == IVLM_BF16)
        return launch_postprocess(static_cast<const bf16_t*>(low), n, h, w, img, in_h, in_w, oh, ow, 1, out, st, gt, ignore_label);
    return IVLM_ERR_UNSUPPORTED;
}
