"""Second witness for the rasteriser restatement (`oracle/raster.py`): an fp64 WORLD-SPACE RAY CASTER.

TEST INFRASTRUCTURE ONLY.  The rasteriser of the path (pytorch3d: preprocess_data/render_mesh_utils.py:115-174) is absent from this
image and the reference holds no golden tables, so `oracle/raster.py` is a single restatement and stays **parity unpinned**.  This
module removes the single-restatement risk without claiming a pin: it computes the same two tables - `pix_to_face` and the
per-pixel barycentrics - by a DIFFERENT algorithm from the SAME published conventions, and `tests/test_raster.py`
asks the two to agree on the four `HUMAN_VIEW_DICT` cameras:

  * raster.py projects the vertices to NDC in fp32, walks faces, evaluates 2-D edge functions per pixel and makes the
    barycentrics perspective-correct with the 1/z formula (pytorch3d's naive rasteriser);
  * this file never projects a triangle: it shoots one ray per pixel centre from the camera centre through the image plane IN
    WORLD SPACE (camera stated geometrically: eye, forward, left, up - not as the R, T matrices) and intersects it with the
    triangles in 3-D (Moeller-Trumbore, fp64).  The 3-D barycentrics of the hit point ARE the perspective-correct barycentrics,
    and the ray parameter is the view depth, so nearest-hit = the z test.

Shared conventions (both follow SURVEY.md 8c; if both are wrong about pytorch3d in the same way this does not show it):
eye C = d (cos e sin a, sin e, cos e cos a); looking at the origin; up +Y; +X LEFT, +Y UP in NDC; pixel (i, j) centre at
y = 1 - (2 i + 1) / H, x = 1 - (2 j + 1) / W; FoV 60 deg; the (tx, ty) of render_mesh_utils.py:115-119 shift the view-space
coordinates; strict inside test; no culling; nearest hit wins.
"""
from __future__ import annotations

import numpy as np


def camera(dist, elev, azim, tx=0.0, ty=0.0):
    """-> (eye, left, up, fwd) unit vectors in world space, fp64, and the view-space shift (tx, ty).
    A world point X has view coordinates ((X - eye).left + tx, (X - eye).up + ty, (X - eye).fwd)."""
    e, a = np.deg2rad(np.float64(elev)), np.deg2rad(np.float64(azim))
    eye = dist * np.array([np.cos(e) * np.sin(a), np.sin(e), np.cos(e) * np.cos(a)])
    fwd = -eye / np.linalg.norm(eye)
    left = np.cross([0.0, 1.0, 0.0], fwd)
    left /= np.linalg.norm(left)
    up = np.cross(fwd, left)
    return eye, left, up, fwd, (float(tx), float(ty))


def pixel_rays(cam, H, W, fov_deg=60.0):
    """origin [3], directions [H, W, 3] (fp64, NOT normalised: the ray parameter t is the view depth z)."""
    eye, left, up, fwd, (tx, ty) = cam
    s = 1.0 / np.tan(np.deg2rad(fov_deg) / 2.0)
    v = 1.0 - (2.0 * np.arange(H) + 1.0) / H
    u = 1.0 - (2.0 * np.arange(W) + 1.0) / W
    # view coordinates of the points of a pixel's ray: (u t / s, v t / s, t)  =>  world X = eye + (u t / s - tx) left + (v t / s - ty) up + t fwd
    origin = eye - tx * left - ty * up
    dirs = (u[None, :, None] / s) * left + (v[:, None, None] / s) * up + fwd
    return origin, dirs


def cast_mesh(verts, faces, cam, H, W, fov_deg=60.0, pad=1):
    """-> (pix_to_face int64 [H, W] (-1 none), bary f64 [H, W, 3] (-1 none), depth f64 [H, W] (inf none)).
    One Moeller-Trumbore test per (face, pixel of the face's padded pixel bounding box)."""
    V = np.asarray(verts, dtype=np.float64)
    Fc = np.asarray(faces, dtype=np.int64)
    eye, left, up, fwd, (tx, ty) = cam
    s = 1.0 / np.tan(np.deg2rad(fov_deg) / 2.0)
    origin, dirs = pixel_rays(cam, H, W, fov_deg)
    # candidate pixels per face: pixel-space bounding box of its three projected corners (used ONLY to bound the search)
    rel = V - eye
    vz = rel @ fwd
    with np.errstate(divide="ignore", invalid="ignore"):
        ux = s * (rel @ left + tx) / vz
        vy = s * (rel @ up + ty) / vz
    col = ((1.0 - ux) * W - 1.0) / 2.0  # inverse of u = 1 - (2 j + 1) / W
    row = ((1.0 - vy) * H - 1.0) / 2.0
    best_t = np.full((H, W), np.inf)
    best_f = np.full((H, W), -1, dtype=np.int64)
    best_b = np.full((H, W, 3), -1.0)
    for f, (i0, i1, i2) in enumerate(Fc):
        if min(vz[i0], vz[i1], vz[i2]) <= 1e-9:  # (a corner behind the eye: search the whole image)
            r0, r1, c0, c1 = 0, H - 1, 0, W - 1
        else:
            rr, cc = row[[i0, i1, i2]], col[[i0, i1, i2]]
            r0, r1 = int(np.floor(rr.min())) - pad, int(np.ceil(rr.max())) + pad
            c0, c1 = int(np.floor(cc.min())) - pad, int(np.ceil(cc.max())) + pad
            r0, c0, r1, c1 = max(r0, 0), max(c0, 0), min(r1, H - 1), min(c1, W - 1)
            if r0 > r1 or c0 > c1:
                continue
        D = dirs[r0: r1 + 1, c0: c1 + 1]
        p0, e1, e2 = V[i0], V[i1] - V[i0], V[i2] - V[i0]
        pv = np.cross(D, e2)
        det = pv @ e1
        ok = np.abs(det) > 1e-18
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = origin - p0
        b1 = (pv @ tv) * inv
        qv = np.cross(tv, e1)
        b2 = (D @ qv) * inv
        t = (e2 @ qv) * inv
        b0 = 1.0 - b1 - b2
        sub_t = best_t[r0: r1 + 1, c0: c1 + 1]
        win = ok & (b0 > 0) & (b1 > 0) & (b2 > 0) & (t > 0) & (t < sub_t)
        if not win.any():
            continue
        wi, wj = np.nonzero(win)
        gi, gj = wi + r0, wj + c0
        best_t[gi, gj] = t[wi, wj]
        best_f[gi, gj] = f
        best_b[gi, gj, 0] = b0[wi, wj]
        best_b[gi, gj, 1] = b1[wi, wj]
        best_b[gi, gj, 2] = b2[wi, wj]
    return best_f, best_b, best_t
