"""GPU parity of the Render-Localize-Lift kernels (through the C ABI) against the CPU oracle and
the reference-generated golden vectors."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-6  # fp32 sums in a different (but fixed) order than the sequential oracle


def _t(a, dev, dtype=None):
    import torch

    t = torch.as_tensor(np.ascontiguousarray(a)).to(dev)
    return t.to(dtype) if dtype is not None else t


def _check_sets(got, exp, tol=TOL):
    for thr, ge in ((0.5, True), (0.3, False)):
        g = got >= thr if ge else got > thr
        e = exp >= thr if ge else exp > thr
        edge = np.abs(exp - thr) <= tol
        assert np.array_equal(g[~edge], e[~edge]), f"vertex-id set mismatch at threshold {thr}"


def test_lift_mesh_soft_golden(hip_lib, cuda, golden_dir):
    import torch

    from interactvlm_amd import ops

    for f in sorted(glob.glob(os.path.join(golden_dir, "lift_mesh_soft_s*.npz"))):
        d = np.load(f)
        nv = int(d["num_vertices"])
        logits = _t(d["logits"], cuda)
        vid, bary = _t(d["vid"], cuda, torch.int32), _t(d["bary"], cuda)
        plan = ops.LiftPlan(vid, bary, nv)
        out, nviews = ops.lift_mesh_plan(logits, plan, 0, 20.0, want_nviews=True)
        np.testing.assert_allclose(out.cpu().numpy(), d["expected"], atol=TOL, rtol=0)
        _check_sets(out.cpu().numpy(), d["expected"])
        out2, nviews2 = ops.lift_mesh_dense(logits, vid, bary, nv, 0, 20.0, want_nviews=True)
        np.testing.assert_allclose(out2.cpu().numpy(), d["expected"], atol=TOL, rtol=0)
        assert torch.equal(nviews, nviews2)


def test_lift_mesh_thresh_golden(hip_lib, cuda, golden_dir):
    import torch

    from interactvlm_amd import ops

    for f in sorted(glob.glob(os.path.join(golden_dir, "lift_mesh_thresh_s*.npz"))):
        d = np.load(f)
        nv = int(d["num_vertices"])
        vid, bary = _t(d["vid"], cuda, torch.int32), _t(d["bary"], cuda)
        plan = ops.LiftPlan(vid, bary, nv)
        for lk, ek in (("logits", "expected"), ("logits_partial", "expected_partial")):
            if lk not in d.files:
                continue
            logits = _t(d[lk][None], cuda)
            a = ops.lift_mesh_plan(logits, plan, 1, 0.3).cpu().numpy()
            b = ops.lift_mesh_dense(logits, vid, bary, nv, 1, 0.3).cpu().numpy()
            # a pixel whose p sits within 1 ulp of the 0.3 threshold may flip; none do on these seeds
            np.testing.assert_allclose(a, d[ek], atol=TOL, rtol=0)
            np.testing.assert_allclose(b, d[ek], atol=TOL, rtol=0)


def test_lift_points_golden(hip_lib, cuda, golden_dir):
    import torch

    from interactvlm_amd import ops

    for f in sorted(glob.glob(os.path.join(golden_dir, "lift_points_s*.npz"))):
        d = np.load(f)
        out = ops.lift_points(_t(d["probs"], cuda), _t(d["pid"], cuda, torch.int32), int(d["num_points"]))
        np.testing.assert_allclose(out.cpu().numpy(), d["expected"], atol=TOL, rtol=0)


def test_plan_is_sorted_and_reproducible(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops, synth

    vid, bary = synth.synth_mesh_tables(4, 128, 128, 997, fg=0.4, seed=3)
    vid_t, bary_t = _t(vid, cuda, torch.int32), _t(bary, cuda)
    p1 = ops.LiftPlan(vid_t, bary_t, 997)
    p2 = ops.LiftPlan(vid_t, bary_t, 997)
    assert p1.nnz == p2.nnz
    assert torch.equal(p1.row_ptr, p2.row_ptr) and torch.equal(p1.ent_pix, p2.ent_pix)
    assert torch.equal(p1.ent_w, p2.ent_w)
    # CSR content == the valid (pixel, slot) pairs of the dense table
    ok = ((vid >= 0) & (vid < 997)).all(-1)
    assert p1.nnz == 3 * int(ok.sum())
    logits = _t(synth.synth_normal("t/l", (3, 4, 128, 128), 4.0, 1), cuda)
    a = ops.lift_mesh_plan(logits, p1)
    b = ops.lift_mesh_plan(logits, p2)
    assert torch.equal(a, b), "plan path must be bit-reproducible"


@pytest.mark.parametrize("patch", [0, 6])
def test_full_size_vs_oracle(hip_lib, cuda, patch):
    """BASELINE size: 4 views x 1024^2 x 6890 vertices, random (patch=0) and clustered tables."""
    import torch

    from interactvlm_amd import ops, synth
    from oracle import cref

    V, H, W, NV = 4, 1024, 1024, 6890
    vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.4, seed=0, patch=patch)
    logits = synth.synth_normal("full/logits", (2, V, H, W), 4.0, seed=0)
    exp, exp_n = cref.lift_mesh_soft(logits, vid, bary, NV)
    vid_t, bary_t, lg = _t(vid, cuda, torch.int32), _t(bary, cuda), _t(logits, cuda)
    plan = ops.LiftPlan(vid_t, bary_t, NV)
    got, got_n = ops.lift_mesh_plan(lg, plan, 0, 20.0, want_nviews=True)
    np.testing.assert_allclose(got.cpu().numpy(), exp, atol=5e-6, rtol=0)
    assert np.array_equal(got_n.cpu().numpy(), exp_n), "visibility set (view_count>0) must be exact"
    _check_sets(got.cpu().numpy(), exp, 5e-6)
    got2 = ops.lift_mesh_dense(lg, vid_t, bary_t, NV, 0, 20.0)
    np.testing.assert_allclose(got2.cpu().numpy(), exp, atol=5e-6, rtol=0)
    # thresholded rule on the same tables
    exp_t, _ = cref.lift_mesh_thresh(logits[0], vid, bary, NV)
    got_t = ops.lift_mesh_plan(lg[:1], plan, 1, 0.3).cpu().numpy()
    bad = np.abs(got_t - exp_t) > 5e-6
    assert bad.mean() < 1e-3, "more than edge-of-threshold differences in the p>0.3 lift"


def test_lift_properties_full_size(hip_lib, cuda):
    """Size-independent properties: constant logits -> sigmoid(c) on every seen vertex; clamp;
    permutation of views leaves the result unchanged up to rounding."""
    import torch

    from interactvlm_amd import ops, synth

    V, H, W, NV = 4, 1024, 1024, 6890
    vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.3, seed=5, adversarial=False, patch=4)
    vid_t, bary_t = _t(vid, cuda, torch.int32), _t(bary, cuda)
    plan = ops.LiftPlan(vid_t, bary_t, NV)
    for c in (-30.0, -1.25, 0.0, 2.0, 50.0):
        lg = torch.full((1, V, H, W), c, device=cuda)
        out, nv = ops.lift_mesh_plan(lg, plan, 0, 20.0, want_nviews=True)
        want = 1.0 / (1.0 + np.exp(-np.clip(c, -20, 20)))
        seen = nv > 0
        assert torch.allclose(out[seen], torch.full_like(out[seen], want), atol=2e-6)
        assert (out[~seen] == 0).all()
    lg = _t(synth.synth_normal("prop/l", (1, V, H, W), 3.0, 2), cuda)
    perm = [2, 0, 3, 1]
    plan_p = ops.LiftPlan(vid_t[perm].contiguous(), bary_t[perm].contiguous(), NV)
    a = ops.lift_mesh_plan(lg, plan)
    b = ops.lift_mesh_plan(lg[:, perm].contiguous(), plan_p)
    assert torch.allclose(a, b, atol=2e-6)


def test_points_full_size_vs_oracle(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops, synth
    from oracle import cref

    B, V, H, W, NP = 2, 4, 1024, 1024, 2048
    pid = synth.synth_point_maps(B, V, H, W, NP, fg=0.3, seed=0)
    probs = synth.synth_uniform("full/probs", (B, V, H, W), 0, 1, seed=0)
    exp, exp_n = cref.lift_points(probs, pid, NP)
    got, got_n = ops.lift_points(_t(probs, cuda), _t(pid, cuda, torch.int32), NP, want_nviews=True)
    np.testing.assert_allclose(got.cpu().numpy(), exp, atol=2e-5, rtol=0)  # ~600-term fp32 sums, free order
    assert np.array_equal(got_n.cpu().numpy(), exp_n)
    # shared (un-batched) map
    got_s = ops.lift_points(_t(probs, cuda), _t(pid[0], cuda, torch.int32), NP)
    exp_s, _ = cref.lift_points(probs, np.stack([pid[0], pid[0]]), NP)
    np.testing.assert_allclose(got_s.cpu().numpy(), exp_s, atol=2e-5, rtol=0)
    # the point-major plan of a map (what the predictor builds when a p2pmap set comes back): same means, same visibility,
    # no atomics -> bit-reproducible
    for b in range(B):
        plan = ops.LiftPlan.from_points(_t(pid[b], cuda, torch.int32), NP)
        assert plan.nnz == int(((pid[b] >= 0) & (pid[b] < NP)).sum())
        gp, gn = ops.lift_points_plan(_t(probs[b: b + 1], cuda), plan, want_nviews=True)
        np.testing.assert_allclose(gp.cpu().numpy(), exp[b: b + 1], atol=2e-5, rtol=0)
        assert np.array_equal(gn.cpu().numpy(), exp_n[b: b + 1])
        assert torch.equal(gp, ops.lift_points_plan(_t(probs[b: b + 1], cuda), plan))
    # a map with unseen points and an empty view
    pid_e = pid[0].copy()
    pid_e[1] = -1
    pid_e[pid_e == 7] = -1
    plan = ops.LiftPlan.from_points(_t(pid_e, cuda, torch.int32), NP)
    ge, gne = ops.lift_points_plan(_t(probs[:1], cuda), plan, want_nviews=True)
    ee, ene = cref.lift_points(probs[:1], pid_e[None], NP)
    np.testing.assert_allclose(ge.cpu().numpy(), ee, atol=2e-5, rtol=0)
    assert np.array_equal(gne.cpu().numpy(), ene) and float(ge[0, 7]) == 0.0


def test_edge_cases(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops

    V, H, W, NV = 4, 8, 8, 11
    vid = torch.full((V, H, W, 3), -1, dtype=torch.int32, device=cuda)
    bary = torch.full((V, H, W, 3), -1.0, device=cuda)
    lg = torch.ones((2, V, H, W), device=cuda)
    plan = ops.LiftPlan(vid, bary, NV)  # empty plan
    assert plan.nnz == 0
    out, nv = ops.lift_mesh_plan(lg, plan, want_nviews=True)
    assert not out.any() and not nv.any()
    out = ops.lift_mesh_dense(lg, vid, bary, NV)
    assert not out.any()
    # zero-weight-only hit stays unseen; ids >= Nv drop the whole pixel
    vid[0, 0, 0] = torch.tensor([0, 1, 2], dtype=torch.int32)
    bary[0, 0, 0] = torch.tensor([0.0, 0.5, 0.5])
    vid[1, 0, 0] = torch.tensor([3, 4, NV], dtype=torch.int32)
    bary[1, 0, 0] = torch.tensor([0.3, 0.3, 0.4])
    plan = ops.LiftPlan(vid, bary, NV)
    out, nv = ops.lift_mesh_plan(torch.zeros((1, V, H, W), device=cuda), plan, want_nviews=True)
    assert nv[0].tolist() == [0, 1, 1] + [0] * 8
    assert torch.allclose(out[0, :3], torch.tensor([0.0, 0.5, 0.5], device=cuda))
    # a vertex count too large for LDS privatisation falls back to L2 atomics
    NVB = 50000
    vidb = torch.randint(0, NVB, (V, 32, 32, 3), dtype=torch.int32, device=cuda)
    baryb = torch.rand((V, 32, 32, 3), device=cuda)
    lgb = torch.randn((1, V, 32, 32), device=cuda)
    a = ops.lift_mesh_dense(lgb, vidb, baryb, NVB)
    b = ops.lift_mesh_plan(lgb, ops.LiftPlan(vidb, baryb, NVB))
    assert torch.allclose(a, b, atol=2e-6)


def test_long_rows_sort_fallback(hip_lib, cuda):
    """Rows longer than the LDS sort capacity (few vertices, many pixels)."""
    import torch

    from interactvlm_amd import ops, synth
    from oracle import cref

    V, H, W, NV = 2, 256, 256, 5
    vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.9, seed=1, adversarial=False)
    logits = synth.synth_normal("long/l", (1, V, H, W), 2.0, 0)
    exp, _ = cref.lift_mesh_soft(logits, vid, bary, NV)
    plan = ops.LiftPlan(_t(vid, cuda, torch.int32), _t(bary, cuda), NV)
    got = ops.lift_mesh_plan(_t(logits, cuda), plan).cpu().numpy()
    np.testing.assert_allclose(got, exp, atol=2e-5, rtol=0)
    # entries inside each row are sorted by (slot, pixel): pixel order is non-decreasing per slot run
    rp = plan.row_ptr.cpu().numpy()
    px = plan.ent_pix.cpu().numpy()
    for r in range(len(rp) - 1):
        seg = px[rp[r]: rp[r + 1]]
        drops = int((np.diff(seg) < 0).sum())
        assert drops <= 2, "row not sorted by (slot, pixel)"


def test_postprocess_vs_oracle(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops
    from oracle import cref

    x = np.random.default_rng(0).standard_normal((4, 1, 256, 256)).astype(np.float32)
    for ins, orig, tol in [((1024, 1024), (1024, 1024), 1e-6), ((1024, 683), (1500, 1000), 3e-4),
                           ((768, 1024), (600, 800), 3e-4), ((1024, 1024), (512, 512), 1e-5)]:
        exp = cref.postprocess_masks(x, ins, orig)
        got = ops.postprocess_masks(_t(x, cuda), ins, orig).cpu().numpy()
        assert got.shape == exp.shape and got.dtype == np.float32
        np.testing.assert_allclose(got, exp, atol=tol, rtol=0)
        got_b = ops.postprocess_masks(_t(x, cuda).to(torch.bfloat16), ins, orig).cpu().numpy()
        exp_b = cref.postprocess_masks(torch.from_numpy(x).to(torch.bfloat16).float().numpy(), ins, orig)
        np.testing.assert_allclose(got_b, exp_b, atol=tol, rtol=0)
    s = ops.postprocess_masks(_t(x, cuda), (1024, 1024), (1024, 1024), apply_sigmoid=True).cpu().numpy()
    np.testing.assert_allclose(s, 1 / (1 + np.exp(-cref.postprocess_masks(x, (1024, 1024), (1024, 1024)))), atol=1e-6)


@pytest.mark.parametrize("ins,orig", [((1024, 1024), (1024, 1024)), ((1024, 683), (750, 500)), ((768, 1024), (600, 800))])
def test_fused_lowres_lift_equals_two_step(hip_lib, cuda, ins, orig):
    """ivlm_lift_mesh_plan_lowres == ivlm_postprocess_masks -> ivlm_lift_mesh_plan, bit for bit (same arithmetic),
    and both match the oracle's postprocess + lift within TOL."""
    import torch

    from interactvlm_amd import ops, synth
    from oracle import cref

    V, NV, B = 4, 6890, 2
    H, W = orig
    vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.4, seed=3, patch=6)
    plan = ops.LiftPlan(_t(vid, cuda, torch.int32), _t(bary, cuda), NV)
    low = (np.random.default_rng(5).standard_normal((B, V, 256, 256)) * 6).astype(np.float32)
    for dt in (torch.float32, torch.bfloat16):
        low_t = _t(low, cuda).to(dt)
        full = ops.postprocess_masks(low_t.reshape(B * V, 1, 256, 256), ins, orig).reshape(B, V, H, W)
        for mode, param in ((0, 20.0), (1, 0.3)):
            two = ops.lift_mesh_plan(full, plan, mode=mode, param=param)
            one = ops.lift_mesh_plan_lowres(low_t, plan, ins, orig, 1024, mode=mode, param=param)
            assert torch.equal(one, two)
    exp, _ = cref.lift_mesh_soft(cref.postprocess_masks(low.reshape(B * V, 1, 256, 256), ins, orig).reshape(B, V, H, W),
                                 vid, bary, NV)
    got = ops.lift_mesh_plan_lowres(_t(low, cuda), plan, ins, orig, 1024).cpu().numpy()
    np.testing.assert_allclose(got, exp, atol=1e-3, rtol=0)
    _check_sets(got, exp, 1e-3)  # postprocess itself is only 3e-4-exact on non-identity second resizes


def test_predictor_modules_match_reference_api(hip_lib, cuda, tmp_path):
    """The nn.Module mirrors: same constructor files / forward signatures as model/components.py."""
    import joblib
    import torch

    from interactvlm_amd import components, synth
    from interactvlm_amd.constants import HUMAN_VIEW_DICT, view_names
    from oracle import cref

    V, H, W = 4, 64, 64
    NV = HUMAN_VIEW_DICT["4MV-Z_Vitru"]["num_vertices"]
    vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.5, seed=2)
    names = view_names(HUMAN_VIEW_DICT["4MV-Z_Vitru"])
    d = tmp_path / "data" / "hcontact_vitruvian"
    d.mkdir(parents=True)
    np.savez(d / "pixel_to_vertex_map_1024.npz", **{n: vid[i] for i, n in enumerate(names)})
    np.savez(d / "bary_coords_map_1024.npz", **{n: bary[i] for i, n in enumerate(names)})
    pred = components.HumanContact3DPredictor("4MV-Z_Vitru", V, metadata_root=str(tmp_path / "data"))
    logits = synth.synth_normal("mod/l", (2, V, H, W), 3.0, 0)
    out = pred([_t(logits[0], cuda), _t(logits[1], cuda)], ["hcontact", "other"])
    exp, _ = cref.lift_mesh_soft(logits, vid, bary, NV)
    assert out.shape == (2, NV)
    np.testing.assert_allclose(out[0].cpu().numpy(), exp[0], atol=TOL)
    assert not out[1].any()

    p = tmp_path / "lift2d_dict.pkl"
    joblib.dump({"pixel_to_vertices_map": [vid[i] for i in range(V)], "bary_coords_map": [bary[i] for i in range(V)],
                 "num_vertices": 333}, p)
    om = components.ObjectMeshContact3DPredictor("4MV-Z_HM", V)
    o1 = om([_t(logits[0], cuda)], ds_names=["ocontact"], lift2d_dict_path=str(p))
    o2 = om([_t(logits[0], cuda)], ds_names=["ocontact"], lift2d_dict_path=str(p))  # cached plan
    exp_t, _ = cref.lift_mesh_thresh(logits[0], vid, bary, 333)
    assert o1.shape == (1, 333) and torch.equal(o1, o2)
    np.testing.assert_allclose(o1.cpu().numpy(), exp_t, atol=TOL)
    assert om([_t(logits[0], cuda)], ds_names=["hcontact"]).shape == (1, 0)
    with pytest.raises(ValueError):
        om([_t(logits[0], cuda)], ds_names=["ocontact"])
    with pytest.raises(AssertionError):
        om([_t(logits[0], cuda)] * 2, ds_names=["ocontact", "ocontact"], lift2d_dict_path=str(p))

    pid = synth.synth_point_maps(2, V, H, W, 2048, seed=1)
    paths = []
    for b in range(2):
        row = []
        for v in range(V):
            mp = str(tmp_path / f"mask_{b}_{v}.png")
            np.savez(mp.replace("mask", "p2pmap")[:-4] + ".npz", mapping=pid[b, v])
            row.append(mp)
        paths.append(row)
    probs = synth.synth_uniform("mod/p", (2, V, H, W), 0, 1, 0)
    pc = components.ObjectPCAfford3DPredictor("4MV-Z_HM", V)
    o = pc([_t(probs[0], cuda), _t(probs[1], cuda)], None, paths)
    exp_p, _ = cref.lift_points(probs, pid, 2048)
    np.testing.assert_allclose(o.cpu().numpy(), exp_p, atol=2e-6)
    # the same p2pmap files again: the predictor inverts them once into point-major plans (deterministic gather) - same results
    o2 = pc([_t(probs[0], cuda), _t(probs[1], cuda)], None, paths)
    assert len(pc._plans) == 2
    np.testing.assert_allclose(o2.cpu().numpy(), exp_p, atol=2e-6)
    o3 = pc([_t(probs[1], cuda)], None, paths[1:])
    assert torch.equal(o3[0], o2[1])


def test_nan_logits_stay_nan_like_the_reference(hip_lib, cuda):
    """torch.clamp / the per-view ratio of the reference propagate a NaN mask logit into every vertex it votes for
    (components.py:242-277; the oracle: np.clip, votes / cnt); fminf / fmaxf and a `ratio >= 0` test would silently turn it into a
    plausible contact.  Plan, streaming and point-map kernels against the oracle, NaN positions included."""
    import torch

    from interactvlm_amd import ops
    from oracle import lift as OL

    V, H, W, NV = 4, 64, 64, 500
    g = np.random.default_rng(5)
    vid = g.integers(0, NV, (V, H, W, 3)).astype(np.int32)
    vid[g.random((V, H, W)) < 0.5] = -1
    bary = g.dirichlet((1, 1, 1), (V, H, W)).astype(np.float32)
    lg = (g.standard_normal((2, V, H, W)) * 3).astype(np.float32)
    lg[0, 1, 10:12, 5:40] = np.nan
    lg[1, 3, 33, :] = np.nan
    exp, exp_nv = OL.lift_mesh_soft(lg, vid, bary, NV)
    assert np.isnan(exp).any() and not np.isnan(exp).all()
    tv, tb, tl = _t(vid, cuda), _t(bary, cuda), _t(lg, cuda)
    got, nv1 = ops.lift_mesh_plan(tl, ops.LiftPlan(tv, tb, NV), 0, 20.0, want_nviews=True)
    got2 = ops.lift_mesh_dense(tl, tv, tb, NV, 0, 20.0)
    for o in (got, got2):
        o = o.cpu().numpy()
        assert np.array_equal(np.isnan(o), np.isnan(exp))
        np.testing.assert_allclose(o, exp, atol=TOL, rtol=0)  # (equal_nan)
    assert np.array_equal(nv1.cpu().numpy(), exp_nv)
    # point maps (mean of the probabilities over the pixels of a point)
    pid = g.integers(-1, 300, (V, H, W)).astype(np.int32)
    pr = 1.0 / (1.0 + np.exp(-lg[:1, :, :, :]))
    expp = OL.lift_points(pr, pid[None], 300)
    expp = expp[0] if isinstance(expp, tuple) else expp
    gp = ops.lift_points(_t(pr, cuda), _t(pid, cuda), 300).cpu().numpy()
    gpp = ops.lift_points_plan(_t(pr, cuda), ops.LiftPlan.from_points(_t(pid, cuda), 300)).cpu().numpy()
    assert np.array_equal(np.isnan(gpp), np.isnan(expp))
    np.testing.assert_allclose(gpp, expp, atol=2e-5, rtol=0)
    assert np.isnan(expp).any() and np.array_equal(np.isnan(gp), np.isnan(expp))
    np.testing.assert_allclose(gp, expp, atol=2e-5, rtol=0)
