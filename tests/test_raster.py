"""Rasteriser: CPU self-consistency of the oracle (no GPU), and GPU kernels vs the oracle (-m gpu).
pytorch3d is absent ("parity unpinned"), so these are geometric / self-consistency checks."""
import numpy as np
import pytest

from oracle import raster as R


def test_camera_is_rigid_and_looks_at_origin():
    for d, e, a in [(2.0, 45.0, 315.0), (2.0, 315.0, 135.0), (1.5, 330.0, 225.0), (2.0, 0.0, 0.0)]:
        Rm, T = R.look_at_view_transform(d, e, a)
        assert np.allclose(Rm.T @ Rm, np.eye(3), atol=1e-5) and abs(np.linalg.det(Rm) - 1) < 1e-4
        # the origin sits on the optical axis at distance d
        assert np.allclose(np.zeros(3) @ Rm + T, [0, 0, d], atol=1e-4)


def test_oracle_sphere_render_properties():
    v, f = R.icosphere(2)
    Rm, T = R.look_at_view_transform(2.0, 45.0, 315.0)
    p2v, bary, p2f = R.rasterize_mesh(v * 0.5, f, Rm, T, 96, 96)
    hit = p2f >= 0
    # a radius-0.5 sphere at distance 2 with a 60 deg FoV covers a disc of the image centred in the middle
    frac = hit.mean()
    assert 0.12 < frac < 0.25
    ii, jj = np.nonzero(hit)
    assert abs(ii.mean() - 47.5) < 1.5 and abs(jj.mean() - 47.5) < 1.5
    # barycentrics are a partition of unity inside, -1 outside; ids are the face's vertices
    assert np.allclose(bary[hit].sum(-1), 1.0, atol=1e-4) and (bary[hit] > 0).all() and (bary[~hit] == -1).all()
    assert (p2v[~hit] == -1).all() and np.array_equal(p2v[hit], f[p2f[hit]])
    # visible faces point towards the camera (closed convex mesh => no back face can win the z test)
    C = -T @ Rm.T
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    ctr = v[f].mean(1) * 0.5
    facing = (n * (C - ctr)).sum(-1) > 0
    assert facing[np.unique(p2f[hit])].all()


def test_oracle_points_nearest_wins():
    pts = np.array([[0, 0, 0.5], [0, 0, -0.5], [0.3, 0, 0]], dtype=np.float32)
    Rm, T = R.look_at_view_transform(2.0, 0.0, 0.0)  # camera on +Z looking at the origin
    m = R.rasterize_points(pts, Rm, T, 0.1, 64, 64)
    assert m[32, 32] == 0 or m[31, 31] == 0  # the point nearer to the camera hides the one behind it
    assert (m == 1).sum() == 0 and (m == 2).sum() > 0


@pytest.mark.gpu
def test_gpu_mesh_raster_vs_oracle(hip_lib, cuda):
    import torch

    from interactvlm_amd import render

    v, f = R.icosphere(3)
    v = (v * np.array([0.45, 0.3, 0.35], dtype=np.float32)).astype(np.float32)  # an ellipsoid: less symmetric
    for cam in [(2.0, 45.0, 315.0, 0.0, 0.0), (2.0, 315.0, 135.0, 0.0, 0.3), (1.5, 330.0, 225.0, 0.0, 0.0)]:
        Rm, T = R.look_at_view_transform(*cam)
        e_p2v, e_bary, e_p2f = R.rasterize_mesh(v, f, Rm, T, 128, 128)
        p2v, bary, p2f = render.rasterize_mesh(torch.from_numpy(v).to(cuda), torch.from_numpy(f).to(cuda), cam,
                                               (128, 128), want_faces=True)
        p2f, p2v, bary = p2f.cpu().numpy(), p2v.cpu().numpy(), bary.cpu().numpy()
        same = p2f == e_p2f
        # silhouette / shared-edge pixels may flip with 1-ulp differences of the edge functions
        assert same.mean() > 0.997, same.mean()
        assert np.array_equal(p2v[same], e_p2v[same])
        assert np.abs(bary[same] - e_bary[same]).max() < 2e-4


@pytest.mark.gpu
def test_gpu_point_raster_and_tables_feed_the_lift(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops, render

    rng = np.random.default_rng(0)
    pts = rng.standard_normal((2048, 3)).astype(np.float32)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True) * 2.2
    cam = (2.0, 45.0, 45.0, 0.0, 0.0)
    Rm, T = R.look_at_view_transform(cam[0], cam[1], cam[2], 0.0, cam[4])
    exp = R.rasterize_points(pts, Rm, T, 0.05, 128, 128)
    got = render.rasterize_points(torch.from_numpy(pts).to(cuda), cam, 0.05, (128, 128)).cpu().numpy()
    assert (got == exp).mean() > 0.995
    # end to end: rasterised body-like tables -> lift plan -> contacts of a constant mask
    v, f = R.icosphere(4)
    vid, bary = render.human_lift_tables(torch.from_numpy(v * 0.5).to(cuda), torch.from_numpy(f).to(cuda),
                                         image_size=(256, 256))
    assert vid.shape == (4, 256, 256, 3)
    plan = ops.LiftPlan(vid.contiguous(), bary.contiguous(), v.shape[0])
    out, nv = ops.lift_mesh_plan(torch.full((1, 4, 256, 256), 1.5, device=cuda), plan, want_nviews=True)
    seen = nv > 0
    assert seen.float().mean() > 0.8  # four views see most of a sphere
    assert torch.allclose(out[seen], torch.full_like(out[seen], 1 / (1 + np.exp(-1.5))), atol=1e-5)
