// "Render" of Render-Localize-Lift for gfx950: mesh / point rasterisation -> pixel->vertex(+barycentric) and
// pixel->point lift tables.  Replaces the pytorch3d calls of
//   preprocess_data/render_mesh_utils.py:115-174 (MeshRasterizer, faces_per_pixel=1, blur_radius=0)
//   preprocess_data/utils_obj_pc.py:28-42,88-113 (PointsRasterizer, keep the nearest point)
//   utils/demo_utils.py:171-257 (per-object tables at demo time)
// pytorch3d (un-vendored, unpinned "@stable") is restated from its published conventions: row-vector cameras
// X_view = X_world.R + T, FoV 60 deg, NDC +X left / +Y up, pixel centres at 1-(2i+1)/H, view-space z as depth,
// strict inside test on the un-corrected barycentrics, perspective-correct barycentrics as output, nearest z wins,
// equal z -> lower face index.
//
// Design: one thread per primitive walks its screen bounding box and takes the per-pixel minimum of a packed
// 64-bit key (float-ordered z << 32 | primitive index) with atomicMin (L2 atomics); a resolve pass turns the
// winning key into the output tables.  Integer/atomic + HBM bound; no MFMA by design.
#include "kernels.h"

namespace ivlm {
namespace {

constexpr float kEps = 1e-8f;
constexpr unsigned long long kEmpty = 0xffffffffffffffffull;

struct Cam {
    float R[9];  // row-major, X_view = X_world . R + T
    float T[3];
    float s;     // 1 / tan(fov/2)
};

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// world -> (x_ndc, y_ndc, z_view)
__global__ void project_kernel(const float* __restrict__ verts, int n, Cam c, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    const float vx = x * c.R[0] + y * c.R[3] + z * c.R[6] + c.T[0];
    const float vy = x * c.R[1] + y * c.R[4] + z * c.R[7] + c.T[1];
    const float vz = x * c.R[2] + y * c.R[5] + z * c.R[8] + c.T[2];
    out[3 * i] = c.s * vx / vz;
    out[3 * i + 1] = c.s * vy / vz;
    out[3 * i + 2] = vz;
}

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return 1.0f - (2.0f * (float)i + 1.0f) / (float)S; }
// inverse, for bounding boxes: ndc -> fractional pixel index
__device__ __forceinline__ float ndc_to_pix(float v, int S) { return ((1.0f - v) * (float)S - 1.0f) * 0.5f; }

__global__ void raster_faces_kernel(const float* __restrict__ sv /*[Nv,3] screen verts*/,
                                    const int32_t* __restrict__ faces, int nf, int H, int W,
                                    unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float x0 = sv[3 * i0], y0 = sv[3 * i0 + 1], z0 = sv[3 * i0 + 2];
    const float x1 = sv[3 * i1], y1 = sv[3 * i1 + 1], z1 = sv[3 * i1 + 2];
    const float x2 = sv[3 * i2], y2 = sv[3 * i2 + 1], z2 = sv[3 * i2 + 2];
    if (fmaxf(z0, fmaxf(z1, z2)) < kEps) return;  // entirely behind the camera
    const float area = edge_fn(x2, y2, x0, y0, x1, y1);
    if (area <= kEps && area >= -kEps) return;  // degenerate face
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    // NDC +X left / +Y up: larger ndc -> smaller pixel index
    int j_lo = (int)floorf(ndc_to_pix(xmax, W)), j_hi = (int)ceilf(ndc_to_pix(xmin, W));
    int i_lo = (int)floorf(ndc_to_pix(ymax, H)), i_hi = (int)ceilf(ndc_to_pix(ymin, H));
    j_lo = j_lo < 0 ? 0 : j_lo;
    i_lo = i_lo < 0 ? 0 : i_lo;
    j_hi = j_hi > W - 1 ? W - 1 : j_hi;
    i_hi = i_hi > H - 1 ? H - 1 : i_hi;
    const float inv_area = 1.0f / (area + kEps);
    for (int i = i_lo; i <= i_hi; ++i) {
        const float py = pix_to_ndc(i, H);
        for (int j = j_lo; j <= j_hi; ++j) {
            const float px = pix_to_ndc(j, W);
            const float b0 = edge_fn(px, py, x1, y1, x2, y2) * inv_area;
            const float b1 = edge_fn(px, py, x2, y2, x0, y0) * inv_area;
            const float b2 = edge_fn(px, py, x0, y0, x1, y1) * inv_area;
            if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) continue;  // blur_radius = 0: strictly inside
            // perspective-correct barycentrics, then depth
            const float t0 = b0 * z1 * z2, t1 = z0 * b1 * z2, t2 = z0 * z1 * b2;
            const float den = fmaxf(t0 + t1 + t2, kEps);
            const float pz = (t0 * z0 + t1 * z1 + t2 * z2) / den;
            if (pz < 0.0f) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)f;
            atomicMin(&zbuf[(size_t)i * W + j], key);
        }
    }
}

__global__ void resolve_faces_kernel(const float* __restrict__ sv, const int32_t* __restrict__ faces, int H, int W,
                                     const unsigned long long* __restrict__ zbuf, int32_t* __restrict__ p2v,
                                     float* __restrict__ bary, int32_t* __restrict__ pix_to_face) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const unsigned long long key = zbuf[p];
    if (key == kEmpty) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p2v[3 * p + k] = -1;     // render_mesh_utils.py:146
            bary[3 * p + k] = -1.f;  // pytorch3d empty-pixel convention
        }
        if (pix_to_face) pix_to_face[p] = -1;
        return;
    }
    const int f = (int)(key & 0xffffffffu);
    const int i = p / W, j = p - i * W;
    const float px = pix_to_ndc(j, W), py = pix_to_ndc(i, H);
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float x0 = sv[3 * i0], y0 = sv[3 * i0 + 1], z0 = sv[3 * i0 + 2];
    const float x1 = sv[3 * i1], y1 = sv[3 * i1 + 1], z1 = sv[3 * i1 + 2];
    const float x2 = sv[3 * i2], y2 = sv[3 * i2 + 1], z2 = sv[3 * i2 + 2];
    const float inv_area = 1.0f / (edge_fn(x2, y2, x0, y0, x1, y1) + kEps);
    const float b0 = edge_fn(px, py, x1, y1, x2, y2) * inv_area;
    const float b1 = edge_fn(px, py, x2, y2, x0, y0) * inv_area;
    const float b2 = edge_fn(px, py, x0, y0, x1, y1) * inv_area;
    const float t0 = b0 * z1 * z2, t1 = z0 * b1 * z2, t2 = z0 * z1 * b2;
    const float den = fmaxf(t0 + t1 + t2, kEps);
    p2v[3 * p] = i0;
    p2v[3 * p + 1] = i1;
    p2v[3 * p + 2] = i2;
    bary[3 * p] = t0 / den;
    bary[3 * p + 1] = t1 / den;
    bary[3 * p + 2] = t2 / den;
    if (pix_to_face) pix_to_face[p] = f;
}

__global__ void raster_points_kernel(const float* __restrict__ sp /*[Np,3] screen points*/, int np, float radius,
                                     int H, int W, unsigned long long* __restrict__ zbuf) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= np) return;
    const float x = sp[3 * q], y = sp[3 * q + 1], z = sp[3 * q + 2];
    if (z < 0.0f) return;
    int j_lo = (int)floorf(ndc_to_pix(x + radius, W)), j_hi = (int)ceilf(ndc_to_pix(x - radius, W));
    int i_lo = (int)floorf(ndc_to_pix(y + radius, H)), i_hi = (int)ceilf(ndc_to_pix(y - radius, H));
    j_lo = j_lo < 0 ? 0 : j_lo;
    i_lo = i_lo < 0 ? 0 : i_lo;
    j_hi = j_hi > W - 1 ? W - 1 : j_hi;
    i_hi = i_hi > H - 1 ? H - 1 : i_hi;
    const float r2 = radius * radius;
    const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)q;
    for (int i = i_lo; i <= i_hi; ++i) {
        const float dy = pix_to_ndc(i, H) - y;
        for (int j = j_lo; j <= j_hi; ++j) {
            const float dx = pix_to_ndc(j, W) - x;
            if (dx * dx + dy * dy < r2) atomicMin(&zbuf[(size_t)i * W + j], key);
        }
    }
}

__global__ void resolve_points_kernel(const unsigned long long* __restrict__ zbuf, int n, int32_t* __restrict__ map) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned long long key = zbuf[p];
    map[p] = key == kEmpty ? -1 : (int32_t)(key & 0xffffffffu);
}

Cam make_cam(const float* cam12, float fov_deg) {
    Cam c;
    for (int i = 0; i < 9; ++i) c.R[i] = cam12[i];
    for (int i = 0; i < 3; ++i) c.T[i] = cam12[9 + i];
    c.s = 1.0f / tanf(fov_deg * 0.5f * 3.14159265358979323846f / 180.0f);
    return c;
}

}  // namespace

size_t raster_workspace_bytes(int n_prims_verts, int H, int W) {
    return sizeof(unsigned long long) * (size_t)H * W + sizeof(float) * 3 * (size_t)n_prims_verts + 256;
}

int rasterize_mesh(const float* verts, int nv, const int32_t* faces, int nf, const float* cam12_host, float fov_deg,
                   int H, int W, int32_t* p2v, float* bary, int32_t* pix_to_face, void* ws, size_t ws_bytes,
                   hipStream_t st) {
    if (!verts || !faces || !cam12_host || !p2v || !bary || !ws || nv <= 0 || nf <= 0 || H <= 0 || W <= 0)
        return IVLM_ERR_INVALID_ARG;
    if (ws_bytes < raster_workspace_bytes(nv, H, W)) return IVLM_ERR_WORKSPACE;
    unsigned long long* zbuf = static_cast<unsigned long long*>(ws);
    float* sv = reinterpret_cast<float*>(zbuf + (size_t)H * W);
    IVLM_HIP_TRY(hipMemsetAsync(zbuf, 0xff, sizeof(unsigned long long) * (size_t)H * W, st));
    const Cam c = make_cam(cam12_host, fov_deg);
    project_kernel<<<(nv + 255) / 256, 256, 0, st>>>(verts, nv, c, sv);
    raster_faces_kernel<<<(nf + 63) / 64, 64, 0, st>>>(sv, faces, nf, H, W, zbuf);
    resolve_faces_kernel<<<(H * W + 255) / 256, 256, 0, st>>>(sv, faces, H, W, zbuf, p2v, bary, pix_to_face);
    return ivlm_launch_status();
}

int rasterize_points(const float* pts, int np, const float* cam12_host, float fov_deg, float radius, int H, int W,
                     int32_t* map, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!pts || !cam12_host || !map || !ws || np <= 0 || H <= 0 || W <= 0 || !(radius > 0.0f))
        return IVLM_ERR_INVALID_ARG;
    if (ws_bytes < raster_workspace_bytes(np, H, W)) return IVLM_ERR_WORKSPACE;
    unsigned long long* zbuf = static_cast<unsigned long long*>(ws);
    float* sp = reinterpret_cast<float*>(zbuf + (size_t)H * W);
    IVLM_HIP_TRY(hipMemsetAsync(zbuf, 0xff, sizeof(unsigned long long) * (size_t)H * W, st));
    const Cam c = make_cam(cam12_host, fov_deg);
    project_kernel<<<(np + 255) / 256, 256, 0, st>>>(pts, np, c, sp);
    raster_points_kernel<<<(np + 63) / 64, 64, 0, st>>>(sp, np, radius, H, W, zbuf);
    resolve_points_kernel<<<(H * W + 255) / 256, 256, 0, st>>>(zbuf, H * W, map);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {

size_t ivlm_raster_workspace_bytes(int n_verts_or_points, int H, int W) {
    return ivlm::raster_workspace_bytes(n_verts_or_points, H, W);
}

int ivlm_rasterize_mesh(const float* verts, int nv, const int32_t* faces, int nf, const float* cam12_host,
                        float fov_deg, int H, int W, int32_t* p2v, float* bary, int32_t* pix_to_face, void* workspace,
                        size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::rasterize_mesh(verts, nv, faces, nf, cam12_host, fov_deg, H, W, p2v, bary, pix_to_face, workspace,
                                workspace_bytes, ivlm_stream(stream));
}

int ivlm_rasterize_points(const float* pts, int np, const float* cam12_host, float fov_deg, float radius, int H, int W,
                          int32_t* map, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::rasterize_points(pts, np, cam12_host, fov_deg, radius, H, W, map, workspace, workspace_bytes,
                                  ivlm_stream(stream));
}

}  // extern "C"
