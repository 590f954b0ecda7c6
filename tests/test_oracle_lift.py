"""Pins the CPU oracle (NumPy + C restatements) of the lift / postprocess path to the golden
vectors produced by the reference itself (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import cref, lift

TOL = 5e-7  # reference differs from the restatements only by the sigmoid's last ulp


def _files(golden_dir, pat):
    fs = sorted(glob.glob(os.path.join(golden_dir, pat)))
    assert fs, f"missing golden fixtures {pat}"
    return fs


@pytest.mark.parametrize("impl", [lift, cref], ids=["numpy", "c"])
def test_lift_mesh_soft_golden(golden_dir, impl):
    for f in _files(golden_dir, "lift_mesh_soft_s*.npz"):
        d = np.load(f)
        pred, nviews = impl.lift_mesh_soft(d["logits"], d["vid"], d["bary"], int(d["num_vertices"]))
        assert pred.shape == d["expected"].shape
        np.testing.assert_allclose(pred, d["expected"], atol=TOL, rtol=0)
        for name, thr in (("ge", 0.5), ("gt", 0.3)):
            got = pred >= thr if name == "ge" else pred > thr
            exp = d["expected"] >= thr if name == "ge" else d["expected"] > thr
            # sets are exact unless a value sits within TOL of the threshold
            edge = np.abs(d["expected"] - thr) <= TOL
            assert np.array_equal(got[~edge], exp[~edge])


@pytest.mark.parametrize("impl", [lift, cref], ids=["numpy", "c"])
def test_lift_mesh_thresh_golden(golden_dir, impl):
    for f in _files(golden_dir, "lift_mesh_thresh_s*.npz"):
        d = np.load(f)
        pred, _ = impl.lift_mesh_thresh(d["logits"], d["vid"], d["bary"], int(d["num_vertices"]))
        np.testing.assert_allclose(pred, d["expected"], atol=TOL, rtol=0)
        if "logits_partial" in d.files:  # three views select nothing (components.py:471-472)
            pred, nv = impl.lift_mesh_thresh(d["logits_partial"], d["vid"], d["bary"], int(d["num_vertices"]))
            np.testing.assert_allclose(pred, d["expected_partial"], atol=TOL, rtol=0)
            assert nv.max() <= 1.0


@pytest.mark.parametrize("impl", [lift, cref], ids=["numpy", "c"])
def test_lift_points_golden(golden_dir, impl):
    for f in _files(golden_dir, "lift_points_s*.npz"):
        d = np.load(f)
        pred, _ = impl.lift_points(d["probs"], d["pid"], int(d["num_points"]))
        np.testing.assert_allclose(pred, d["expected"], atol=TOL, rtol=0)


def test_zero_weight_hit_is_unseen():
    # a vertex touched only with weight 0 has cnt == 0 -> "unseen" (SURVEY Appendix A note)
    V, H, W, NV = 1, 2, 2, 5
    vid = np.full((V, H, W, 3), -1, np.int32)
    bary = np.full((V, H, W, 3), -1, np.float32)
    vid[0, 0, 0] = (0, 1, 2)
    bary[0, 0, 0] = (0.0, 0.5, 0.5)
    logits = np.zeros((1, V, H, W), np.float32)
    for impl in (lift, cref):
        pred, nviews = impl.lift_mesh_soft(logits, vid, bary, NV)
        assert nviews[0].tolist() == [0, 1, 1, 0, 0]
        np.testing.assert_allclose(pred[0], [0, 0.5, 0.5, 0, 0], atol=1e-7)


def test_empty_inputs():
    V, H, W, NV = 4, 8, 8, 11
    vid = np.full((V, H, W, 3), -1, np.int32)
    bary = np.full((V, H, W, 3), -1, np.float32)
    logits = np.ones((2, V, H, W), np.float32)
    for impl in (lift, cref):
        pred, nviews = impl.lift_mesh_soft(logits, vid, bary, NV)
        assert not pred.any() and not nviews.any()
        p2, n2 = impl.lift_points(logits, np.full((2, V, H, W), -1, np.int32), 7)
        assert not p2.any() and not n2.any()


def test_c_equals_numpy_large_random():
    from interactvlm_amd import synth

    vid, bary = synth.synth_mesh_tables(4, 128, 128, 997, fg=0.4, seed=7)
    logits = synth.synth_normal("t/logits", (2, 4, 128, 128), 4.0, seed=7)
    a, na = lift.lift_mesh_soft(logits, vid, bary, 997)
    b, nb = cref.lift_mesh_soft(logits, vid.astype(np.int32), bary, 997)
    np.testing.assert_allclose(a, b, atol=TOL, rtol=0)
    assert np.array_equal(na, nb)
    a, _ = lift.lift_mesh_thresh(logits[0], vid, bary, 997)
    b, _ = cref.lift_mesh_thresh(logits[0], vid.astype(np.int32), bary, 997)
    np.testing.assert_allclose(a, b, atol=TOL, rtol=0)


def test_postprocess_matches_torch_interpolate():
    import torch
    import torch.nn.functional as F

    x = np.random.default_rng(0).standard_normal((3, 1, 64, 64)).astype(np.float32)
    for ins, orig, tol in [((256, 256), (256, 256), 1e-6), ((256, 171), (375, 250), 2e-4), ((192, 256), (150, 200), 2e-4)]:
        t = F.interpolate(torch.from_numpy(x), (256, 256), mode="bilinear", align_corners=False)
        t = F.interpolate(t[..., : ins[0], : ins[1]], orig, mode="bilinear", align_corners=False).numpy()
        for impl in (lift, cref):
            got = impl.postprocess_masks(x, ins, orig, img_size=256)
            assert got.shape == t.shape
            # non-integer scales amplify the fp32 rounding of the source coordinate (~1e-4)
            np.testing.assert_allclose(got, t, atol=tol, rtol=0)
