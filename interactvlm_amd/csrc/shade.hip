// Hard Phong shading of a rasterised mesh: the colour renders the reference feeds to SAM for object meshes
// (utils/demo_utils.py:146-168 `render_mesh`: pytorch3d MeshRenderer + HardPhongShader + one PointLights; vertex colours
// from TexturesVertex; preprocess_data/render_mesh_utils.py:177-198 is the same for the offline data).
//
// One thread per pixel, everything it needs comes from the rasteriser's own outputs: the three vertex ids of the winning
// face (pixel_to_vertices_map) and its barycentrics interpolate the vertex normals, world positions and colours
// (pytorch3d `interpolate_face_attributes`); then `phong_shading`:
//     colour = (ambient + diffuse * relu(n.l)) * texel + specular * (relu(v.r) * [n.l > 0]) ^ shininess
// with l = normalize(light - p), v = normalize(camera - p), r = -l + 2 (n.l) n, all normalisations with eps 1e-6 like
// F.normalize; background pixels take `bg` (hard_rgb_blend, BlendParams default white).  Output uint8 HWC, the
// (image * 255).astype(uint8) truncation of render_mesh included.  Gather-bound (36 B of ids/bary in, 3 B out per pixel +
// 27 floats of vertex data from L2): HBM, trivially small next to the encoder that consumes the image.
#include "kernels.h"

namespace ivlm {
namespace {

struct ShadeArgs {
    const int32_t* p2v;
    const float* bary;
    const float* verts;
    const float* normals;
    const float* colors;
    int npix;
    float light[3], cam[3];
    float ambient, diffuse, specular, shininess;
    float bg[3];
    uint8_t* out;
};

__device__ __forceinline__ float inv_norm3(float x, float y, float z) { return 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-6f); }

__global__ __launch_bounds__(256) void phong_shade_kernel(ShadeArgs a) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.npix) return;
    const int i0 = a.p2v[3 * p], i1 = a.p2v[3 * p + 1], i2 = a.p2v[3 * p + 2];
    float c[3];
    if (i0 < 0) {
        c[0] = a.bg[0]; c[1] = a.bg[1]; c[2] = a.bg[2];
    } else {
        const float b0 = a.bary[3 * p], b1 = a.bary[3 * p + 1], b2 = a.bary[3 * p + 2];
        float n[3], q[3], t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n[k] = b0 * a.normals[3 * i0 + k] + b1 * a.normals[3 * i1 + k] + b2 * a.normals[3 * i2 + k];
            q[k] = b0 * a.verts[3 * i0 + k] + b1 * a.verts[3 * i1 + k] + b2 * a.verts[3 * i2 + k];
            t[k] = b0 * a.colors[3 * i0 + k] + b1 * a.colors[3 * i1 + k] + b2 * a.colors[3 * i2 + k];
        }
        const float rn = inv_norm3(n[0], n[1], n[2]);
        float l[3] = {a.light[0] - q[0], a.light[1] - q[1], a.light[2] - q[2]};
        const float rl = inv_norm3(l[0], l[1], l[2]);
        float v[3] = {a.cam[0] - q[0], a.cam[1] - q[1], a.cam[2] - q[2]};
        const float rv = inv_norm3(v[0], v[1], v[2]);
        float cosang = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n[k] *= rn; l[k] *= rl; v[k] *= rv;
            cosang += n[k] * l[k];
        }
        float vr = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) vr += v[k] * (-l[k] + 2.0f * cosang * n[k]);
        const float alpha = cosang > 0.0f ? fmaxf(vr, 0.0f) : 0.0f;
        const float spec = a.specular * powf(alpha, a.shininess);
        const float lit = a.ambient + a.diffuse * fmaxf(cosang, 0.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = lit * t[k] + spec;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = c[k] * 255.0f;  // numpy's float -> uint8 cast of an in-range value: truncation
        a.out[3 * p + k] = (uint8_t)(int)fminf(fmaxf(x, 0.0f), 255.0f);
    }
}

}  // namespace

int phong_shade(const int32_t* p2v, const float* bary, const float* verts, const float* normals, const float* colors, int npix,
                const float* light3_host, const float* cam3_host, float ambient, float diffuse, float specular, float shininess,
                const float* bg3_host, uint8_t* out, hipStream_t st) {
    if (!p2v || !bary || !verts || !normals || !colors || !light3_host || !cam3_host || !bg3_host || !out || npix <= 0)
        return IVLM_ERR_INVALID_ARG;
    ShadeArgs a;
    a.p2v = p2v; a.bary = bary; a.verts = verts; a.normals = normals; a.colors = colors; a.npix = npix;
    for (int k = 0; k < 3; ++k) { a.light[k] = light3_host[k]; a.cam[k] = cam3_host[k]; a.bg[k] = bg3_host[k]; }
    a.ambient = ambient; a.diffuse = diffuse; a.specular = specular; a.shininess = shininess; a.out = out;
    phong_shade_kernel<<<(npix + 255) / 256, 256, 0, st>>>(a);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_phong_shade(const int32_t* p2v, const float* bary, const float* verts, const float* normals,
                                const float* colors, int npix, const float* light3_host, const float* cam3_host, float ambient,
                                float diffuse, float specular, float shininess, const float* bg3_host, uint8_t* out_rgb,
                                ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::phong_shade(p2v, bary, verts, normals, colors, npix, light3_host, cam3_host, ambient, diffuse, specular,
                             shininess, bg3_host, out_rgb, ivlm_stream(stream));
}
