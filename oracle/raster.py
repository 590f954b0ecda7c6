"""NumPy restatement of the pytorch3d rasterisation the reference uses for its lift tables.

TEST INFRASTRUCTURE ONLY.  **Parity unpinned**: pytorch3d (`pytorch3d@stable`, requirements.txt:28) is not
installed here and the reference holds no golden tables, so this follows pytorch3d's published conventions (camera
`look_at_view_transform` + `FoVPerspectiveCameras`, `rasterize_meshes` naive path, `rasterize_points`) and is checked
by self-consistency only.  Call sites restated: preprocess_data/render_mesh_utils.py:115-174,
preprocess_data/utils_obj_pc.py:28-42,88-113, utils/demo_utils.py:128-143,171-257.
Second witness [r6]: `oracle/raycast.py` computes pix_to_face + barycentrics by another algorithm (fp64 world-space rays) from the
same conventions; `tests/test_raster.py::test_oracle_raster_agrees_with_an_independent_fp64_ray_caster` holds the two (and the
HIP kernel) together on the four HUMAN_VIEW_DICT cameras.  That guards against a slip in THIS restatement; it is not a pin to
pytorch3d, and the label above stays.
"""
from __future__ import annotations

import numpy as np

F = np.float32
EPS = F(1e-8)


def look_at_view_transform(dist, elev, azim, tx=0.0, ty=0.0):
    """R [3,3] (row-vector convention X_view = X_world @ R + T) and T [3]; degrees; at = origin, up = +Y.
    render_mesh_utils.py:115-119 adds (tx, ty) to T afterwards; utils_obj_pc.py:31 adds only ty."""
    e, a = np.deg2rad(F(elev)), np.deg2rad(F(azim))
    C = F(dist) * np.array([np.cos(e) * np.sin(a), np.sin(e), np.cos(e) * np.cos(a)], dtype=F)
    z = -C / max(np.linalg.norm(C), 1e-5)                      # normalize(at - C)
    up = np.array([0, 1, 0], dtype=F)
    x = np.cross(up, z)
    if np.allclose(x, 0, atol=5e-3):                            # degenerate (looking along +-Y): pytorch3d's fallback
        x = np.cross(z, np.cross(up, z) + np.array([0, 0, 1], dtype=F))
        x = np.cross(up + np.array([0, 0, 1e-3], dtype=F), z)
    x = x / max(np.linalg.norm(x), 1e-5)
    y = np.cross(z, x)
    y = y / max(np.linalg.norm(y), 1e-5)
    R = np.stack([x, y, z], axis=1).astype(F)                   # columns
    T = (-(C @ R)).astype(F)
    T[0] += F(tx)
    T[1] += F(ty)
    return R, T


def project(verts, R, T, fov_deg=60.0):
    v = verts.astype(F) @ R + T
    s = F(1.0 / np.tan(np.deg2rad(fov_deg) / 2.0))
    return np.stack([s * v[:, 0] / v[:, 2], s * v[:, 1] / v[:, 2], v[:, 2]], axis=1).astype(F)


def _ndc(i, S):
    return (F(1.0) - (F(2.0) * i.astype(F) + F(1.0)) / F(S)).astype(F)


def rasterize_mesh(verts, faces, R, T, H, W, fov_deg=60.0):
    """-> (p2v int64 [H,W,3] (-1 bg), bary f32 [H,W,3] (-1 bg), pix_to_face int64 [H,W])."""
    sv = project(verts, R, T, fov_deg)
    zbest = np.full((H, W), np.inf, dtype=F)
    fbest = np.full((H, W), -1, dtype=np.int64)
    bbest = np.full((H, W, 3), -1, dtype=F)
    ys, xs = _ndc(np.arange(H), H), _ndc(np.arange(W), W)
    for f, (i0, i1, i2) in enumerate(faces):
        (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = sv[i0], sv[i1], sv[i2]
        if max(z0, z1, z2) < EPS:
            continue
        area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
        if -EPS <= area <= EPS:
            continue
        xmn, xmx, ymn, ymx = min(x0, x1, x2), max(x0, x1, x2), min(y0, y1, y2), max(y0, y1, y2)
        jj = np.nonzero((xs >= xmn) & (xs <= xmx))[0]
        ii = np.nonzero((ys >= ymn) & (ys <= ymx))[0]
        if len(jj) == 0 or len(ii) == 0:
            continue
        px, py = xs[jj][None, :], ys[ii][:, None]
        inv = F(1.0) / (area + EPS)
        b0 = ((px - x1) * (y2 - y1) - (py - y1) * (x2 - x1)) * inv
        b1 = ((px - x2) * (y0 - y2) - (py - y2) * (x0 - x2)) * inv
        b2 = ((px - x0) * (y1 - y0) - (py - y0) * (x1 - x0)) * inv
        inside = (b0 > 0) & (b1 > 0) & (b2 > 0)
        if not inside.any():
            continue
        t0, t1, t2 = b0 * z1 * z2, z0 * b1 * z2, z0 * z1 * b2
        den = np.maximum(t0 + t1 + t2, EPS)
        pz = (t0 * z0 + t1 * z1 + t2 * z2) / den
        sub_z = zbest[np.ix_(ii, jj)]
        win = inside & (pz >= 0) & (pz < sub_z)          # strict <: equal depth keeps the lower face index
        if not win.any():
            continue
        wi, wj = np.nonzero(win)
        gi, gj = ii[wi], jj[wj]
        zbest[gi, gj] = pz[wi, wj]
        fbest[gi, gj] = f
        bbest[gi, gj, 0] = (t0 / den)[wi, wj]
        bbest[gi, gj, 1] = (t1 / den)[wi, wj]
        bbest[gi, gj, 2] = (t2 / den)[wi, wj]
    p2v = np.full((H, W, 3), -1, dtype=np.int64)
    hit = fbest >= 0
    p2v[hit] = np.asarray(faces, dtype=np.int64)[fbest[hit]]
    return p2v, bbest, fbest


def rasterize_points(pts, R, T, radius, H, W, fov_deg=60.0):
    """-> pixel->point map int64 [H,W] (-1 none): nearest (smallest view z) point whose disc covers the pixel."""
    sp = project(pts, R, T, fov_deg)
    zbest = np.full((H, W), np.inf, dtype=F)
    best = np.full((H, W), -1, dtype=np.int64)
    ys, xs = _ndc(np.arange(H), H), _ndc(np.arange(W), W)
    r2 = F(radius) * F(radius)
    for q, (x, y, z) in enumerate(sp):
        if z < 0:
            continue
        jj = np.nonzero(np.abs(xs - x) <= radius)[0]
        ii = np.nonzero(np.abs(ys - y) <= radius)[0]
        if len(jj) == 0 or len(ii) == 0:
            continue
        d2 = (xs[jj][None, :] - x) ** 2 + (ys[ii][:, None] - y) ** 2
        win = (d2 < r2) & (z < zbest[np.ix_(ii, jj)])
        wi, wj = np.nonzero(win)
        zbest[ii[wi], jj[wj]] = z
        best[ii[wi], jj[wj]] = q
    return best


def normalize_mesh(verts):
    """utils/demo_utils.py:128-143: centre on the bbox centre, scale by the longest bbox side."""
    v = verts.astype(F)
    lo, hi = v.min(0), v.max(0)
    return ((v - (lo + hi) / 2) / (hi - lo).max()).astype(F)


def icosphere(subdiv=3):
    """A closed test mesh (unit sphere)."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=F), np.array(f, dtype=np.int32)


# ---- shaded renders (utils/demo_utils.py:146-168, 196-216: HardPhongShader + PointLights + TexturesVertex) -------------
# Restates pytorch3d's published shading path: Meshes.verts_normals_packed (face cross products accumulated on the three
# vertices, F.normalize eps 1e-6), interpolate_face_attributes with the rasteriser's barycentrics, lighting.PointLights
# .diffuse / .specular, shading.phong_shading, blending.hard_rgb_blend (white background).  Parity unpinned (see header).
def camera_center(R, T):
    """World position of the camera of X_view = X_world @ R + T."""
    return (-(T.astype(F) @ R.T.astype(F))).astype(F)


def vertex_normals(verts, faces):
    v = verts.astype(F)
    f = np.asarray(faces, dtype=np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]).astype(F)
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    return (vn / np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), F(1e-6))).astype(F)


def _unit(x):
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), F(1e-6))


def phong_shade(p2v, bary, verts, normals, colors, light, cam, ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0,
                bg=(1.0, 1.0, 1.0)):
    """p2v int [H,W,3] (-1 background), bary f32 [H,W,3] -> uint8 [H,W,3] = (image * 255).astype(uint8)."""
    ids = np.asarray(p2v, dtype=np.int64)
    fg = ids[..., 0] >= 0
    idc = np.where(fg[..., None], ids, 0)
    b = bary.astype(F)[..., None]
    n = (b * normals.astype(F)[idc]).sum(-2)
    q = (b * verts.astype(F)[idc]).sum(-2)
    t = (b * colors.astype(F)[idc]).sum(-2)
    n = _unit(n)
    l = _unit(np.asarray(light, F) - q)
    v = _unit(np.asarray(cam, F) - q)
    cosang = (n * l).sum(-1)
    refl = -l + F(2.0) * cosang[..., None] * n
    alpha = np.maximum((v * refl).sum(-1), 0) * (cosang > 0)
    spec = F(specular) * np.power(alpha.astype(F), F(shininess))
    col = (F(ambient) + F(diffuse) * np.maximum(cosang, 0))[..., None] * t + spec[..., None]
    img = np.where(fg[..., None], col, np.asarray(bg, F)).astype(F)
    return (np.clip(img * F(255.0), 0, 255)).astype(np.uint8)
