"""Demo-time "Render" for objects and offline table generation for the canonical body, on the GPU.

Mirrors utils/demo_utils.py:128-257 (normalize_mesh, 4 cameras, rasterise -> lift2d_dict) and
preprocess_data/render_mesh_utils.py:115-174 / utils_obj_pc.py:28-113, with the pytorch3d rasteriser replaced by
``ivlm_rasterize_mesh`` / ``ivlm_rasterize_points``.  Camera matrices are 12 floats of host math.
The Phong-shaded colour renders that demo_utils.py feeds to SAM for object meshes are produced by ``object_renders``
(``ivlm_phong_shade`` on the rasteriser's own outputs).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check
from .constants import HUMAN_VIEW_DICT, OBJS_VIEW_DICT

F = np.float32


def look_at_view_transform(dist, elev, azim, tx=0.0, ty=0.0):
    """pytorch3d.renderer.look_at_view_transform(dist, elev, azim) (degrees, at = origin, up = +Y), row-vector
    convention, with the reference's post-hoc translation T[0] += tx, T[1] += ty (render_mesh_utils.py:115-119)."""
    e, a = np.deg2rad(F(elev)), np.deg2rad(F(azim))
    C = F(dist) * np.array([np.cos(e) * np.sin(a), np.sin(e), np.cos(e) * np.cos(a)], dtype=F)
    z = -C / max(float(np.linalg.norm(C)), 1e-5)
    up = np.array([0, 1, 0], dtype=F)
    x = np.cross(up, z)
    if np.allclose(x, 0, atol=5e-3):
        x = np.cross(up + np.array([0, 0, 1e-3], dtype=F), z)
    x = x / max(float(np.linalg.norm(x)), 1e-5)
    y = np.cross(z, x)
    y = y / max(float(np.linalg.norm(y)), 1e-5)
    R = np.stack([x, y, z], axis=1).astype(F)
    T = (-(C @ R)).astype(F)
    T[0] += F(tx)
    T[1] += F(ty)
    return R, T


def _cam12(R, T):
    arr = np.concatenate([R.reshape(-1), T.reshape(-1)]).astype(F)
    return (ctypes.c_float * 12)(*arr.tolist())


def _ws(n, H, W, device):
    nbytes = _lib.load().ivlm_raster_workspace_bytes(int(n), H, W)
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def rasterize_mesh(verts, faces, cam_params, image_size=(1024, 1024), fov_deg=60.0, want_faces=False):
    """verts f32 [Nv,3] / faces i32 [Nf,3] GPU tensors, cam_params (d, elev, azim, tx, ty)
    -> (pixel_to_vertices_map i32 [H,W,3], bary f32 [H,W,3][, pix_to_face i32 [H,W]]) on the GPU."""
    lib = _lib.load()
    assert verts.is_cuda and verts.dtype == torch.float32 and faces.dtype == torch.int32
    verts, faces = verts.contiguous(), faces.contiguous()
    H, W = image_size
    R, T = look_at_view_transform(*cam_params)
    p2v = torch.empty(H, W, 3, dtype=torch.int32, device=verts.device)
    bary = torch.empty(H, W, 3, dtype=torch.float32, device=verts.device)
    p2f = torch.empty(H, W, dtype=torch.int32, device=verts.device) if want_faces else None
    ws, nbytes = _ws(verts.shape[0], H, W, verts.device)
    check(lib.ivlm_rasterize_mesh(verts.data_ptr(), verts.shape[0], faces.data_ptr(), faces.shape[0], _cam12(R, T),
                                  float(fov_deg), H, W, p2v.data_ptr(), bary.data_ptr(),
                                  0 if p2f is None else p2f.data_ptr(), ws.data_ptr(), nbytes,
                                  torch.cuda.current_stream().cuda_stream), "rasterize_mesh")
    return (p2v, bary, p2f) if want_faces else (p2v, bary)


def rasterize_points(pts, cam_params, radius, image_size=(1024, 1024), fov_deg=60.0):
    """pts f32 [Np,3] -> pixel->point map i32 [H,W] (-1 none).  Only the y translation is applied, like
    utils_obj_pc.py:28-32."""
    lib = _lib.load()
    assert pts.is_cuda and pts.dtype == torch.float32
    pts = pts.contiguous()
    H, W = image_size
    d, e, a, _tx, ty = cam_params
    R, T = look_at_view_transform(d, e, a, 0.0, ty)
    m = torch.empty(H, W, dtype=torch.int32, device=pts.device)
    ws, nbytes = _ws(pts.shape[0], H, W, pts.device)
    check(lib.ivlm_rasterize_points(pts.data_ptr(), pts.shape[0], _cam12(R, T), float(fov_deg), float(radius), H, W,
                                    m.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
          "rasterize_points")
    return m


def normalize_mesh(verts):
    """utils/demo_utils.py:128-143."""
    lo, hi = verts.min(0).values, verts.max(0).values
    return (verts - (lo + hi) / 2) / (hi - lo).max()


def object_lift_tables(verts, faces, view_type="4MV-Z_HM_BM", image_size=(1024, 1024)):
    """generate_sam_inp_objs (utils/demo_utils.py:171-257) minus the shaded renders: normalise the mesh, rasterise
    the four object views (distance 1.5) -> (vid i32 [4,H,W,3], bary f32 [4,H,W,3], num_vertices)."""
    cams = OBJS_VIEW_DICT[view_type].get("mesh_cam_params") or OBJS_VIEW_DICT[view_type]["cam_params"]
    v = normalize_mesh(verts.float()).contiguous()
    out = [rasterize_mesh(v, faces, cams[n], image_size) for n in cams]
    return torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]), int(verts.shape[0])


def human_lift_tables(verts, faces, view_type="4MV-Z_Vitru", image_size=(1024, 1024)):
    """Offline body tables (preprocess_data/generate_damon_human_mask.py:112-139 -> pixel_to_vertex_map_1024.npz,
    bary_coords_map_1024.npz) for the four HUMAN_VIEW_DICT cameras."""
    cams = HUMAN_VIEW_DICT[view_type]["cam_params"]
    out = [rasterize_mesh(verts.float().contiguous(), faces, cams[n], image_size) for n in cams]
    return torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out])


def save_lift2d_dict(path, vid, bary, num_vertices):
    """Same joblib file the reference's ObjectMeshContact3DPredictor reads (components.py:392-398)."""
    import joblib

    joblib.dump({"pixel_to_vertices_map": [v.cpu().numpy().astype(np.int64) for v in vid],
                 "bary_coords_map": [b.cpu().numpy() for b in bary], "num_vertices": int(num_vertices)}, path)


# ---- shaded colour renders (the SAM inputs of the object path) ------------------------------------------------------------
LIGHT_LOCATIONS = [[0, 0, 3], [0, 0, 3], [0, 0, -3], [0, 0, -3]]  # utils/demo_utils.py:23, one per object view
YELLOW_VERTEX_COLOR = [1.00, 0.90, 0.30]                          # utils/demo_utils.py:25 ('grey' renders)


def vertex_normals(verts, faces):
    """pytorch3d Meshes.verts_normals_packed: face cross products accumulated on their vertices, normalised (eps 1e-6)."""
    f = faces.long()
    v0, v1, v2 = verts[f[:, 0]], verts[f[:, 1]], verts[f[:, 2]]
    fn = torch.cross(v1 - v0, v2 - v0, dim=1)
    vn = torch.zeros_like(verts)
    for k in range(3):
        vn.index_add_(0, f[:, k], fn)
    return (vn / vn.norm(dim=1, keepdim=True).clamp_min(1e-6)).contiguous()


def shade_mesh(verts, faces, colors, cam_params, light_location, image_size=(1024, 1024), fov_deg=60.0, normals=None,
               raster=None, ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0, background=(1.0, 1.0, 1.0)):
    """render_mesh (utils/demo_utils.py:146-168): HardPhongShader with one point light over vertex colours.
    verts f32 [Nv,3], faces i32 [Nf,3], colors f32 [Nv,3] on the GPU -> uint8 [H,W,3] RGB on the GPU.
    raster = (p2v, bary) of the same camera re-uses an existing rasterisation."""
    lib = _lib.load()
    verts = verts.float().contiguous()
    H, W = image_size
    p2v, bary = raster if raster is not None else rasterize_mesh(verts, faces, cam_params, image_size, fov_deg)
    normals = vertex_normals(verts, faces) if normals is None else normals
    colors = colors.to(device=verts.device, dtype=torch.float32).contiguous()
    R, T = look_at_view_transform(*cam_params)
    cam = (-(T @ R.T)).astype(F)
    f3 = lambda x: (ctypes.c_float * 3)(*[float(t) for t in x])
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=verts.device)
    check(lib.ivlm_phong_shade(p2v.data_ptr(), bary.data_ptr(), verts.data_ptr(), normals.data_ptr(), colors.data_ptr(),
                               H * W, f3(light_location), f3(cam), float(ambient), float(diffuse), float(specular),
                               float(shininess), f3(background), out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream), "phong_shade")
    return out


def object_renders(verts, faces, view_type="4MV-Z_HM_BM", colored=True, image_size=(1024, 1024)):
    """generate_sam_inp_objs (utils/demo_utils.py:171-257), all of it: normalise the mesh, colour it (xyz position colours
    for the 'color' renders, flat yellow for the 'grey' ones, both * 0.8 + 0.1), and for each of the four object cameras
    rasterise once and shade -> (renders uint8 [4,H,W,3], vid i32 [4,H,W,3], bary f32 [4,H,W,3], num_vertices)."""
    cams = OBJS_VIEW_DICT[view_type].get("mesh_cam_params") or OBJS_VIEW_DICT[view_type]["cam_params"]
    v = normalize_mesh(verts.float()).contiguous()
    if colored:
        lo, hi = v.min(0).values, v.max(0).values
        col = (v - lo) / (hi - lo)
    else:
        col = torch.tensor(YELLOW_VERTEX_COLOR, device=v.device).expand(v.shape[0], 3)
    col = (col * 0.8 + 0.1).contiguous()
    vn = vertex_normals(v, faces)
    imgs, vids, barys = [], [], []
    for i, name in enumerate(cams):
        p2v, bary = rasterize_mesh(v, faces, cams[name], image_size)
        imgs.append(shade_mesh(v, faces, col, cams[name], LIGHT_LOCATIONS[i % len(LIGHT_LOCATIONS)], image_size, normals=vn,
                               raster=(p2v, bary)))
        vids.append(p2v)
        barys.append(bary)
    return torch.stack(imgs), torch.stack(vids), torch.stack(barys), int(verts.shape[0])
