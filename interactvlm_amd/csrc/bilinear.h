// Bilinear (align_corners=False) sampling shared by postprocess.hip and the fused low-res lift in lift.hip.
// Restates torch's area_pixel_compute_source_index + upsample_bilinear2d (sam.py:161-171 call sites).
#pragma once
#include "ivlm_common.h"

namespace ivlm_bilinear {

struct Axis {
    int i0, i1;
    float l0, l1;
};

// torch area_pixel_compute_source_index + upsample_bilinear2d index/lambda rule
__device__ __forceinline__ Axis axis_src(int dst, float scale, int n_in) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    int i0 = (int)s;
    i0 = i0 > n_in - 1 ? n_in - 1 : i0;
    Axis a;
    a.i0 = i0;
    a.i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    a.l1 = s - (float)i0;
    a.l0 = 1.0f - a.l1;
    return a;
}

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// value of the (virtual) img x img intermediate at integer pixel (yy, xx)
template <typename T>
__device__ __forceinline__ float stage1(const T* __restrict__ low, int h, int w, float s1y, float s1x, int yy,
                                        int xx) {
    const Axis ay = axis_src(yy, s1y, h), ax = axis_src(xx, s1x, w);
    const T* r0 = low + (size_t)ay.i0 * w;
    const T* r1 = low + (size_t)ay.i1 * w;
    const float t = ld(r0 + ax.i0) * ax.l0 + ld(r0 + ax.i1) * ax.l1;
    const float b = ld(r1 + ax.i0) * ax.l0 + ld(r1 + ax.i1) * ax.l1;
    return t * ay.l0 + b * ay.l1;
}


// value of Sam.postprocess_masks(low)[y, x]: first resize h x w -> img x img, crop, second resize (in_h,in_w)->(oh,ow)
template <typename T>
__device__ __forceinline__ float postprocess_at(const T* __restrict__ low, int h, int w, int img, int in_h, int in_w,
                                                int oh, int ow, int y, int x) {
    const float s1y = (float)h / (float)img, s1x = (float)w / (float)img;
    if (in_h == oh && in_w == ow) return stage1(low, h, w, s1y, s1x, y, x);
    const float s2y = (float)in_h / (float)oh, s2x = (float)in_w / (float)ow;
    const Axis by = axis_src(y, s2y, in_h), bx = axis_src(x, s2x, in_w);
    const float a = stage1(low, h, w, s1y, s1x, by.i0, bx.i0);
    const float b = stage1(low, h, w, s1y, s1x, by.i0, bx.i1);
    const float c = stage1(low, h, w, s1y, s1x, by.i1, bx.i0);
    const float d = stage1(low, h, w, s1y, s1x, by.i1, bx.i1);
    return (a * bx.l0 + b * bx.l1) * by.l0 + (c * bx.l0 + d * bx.l1) * by.l1;
}

}  // namespace ivlm_bilinear
