"""Caller-side preprocessing (a17) vs the reference formulas restated with torch on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sam_and_clip_preprocess(hip_lib, cuda):
    import torch
    import torch.nn.functional as F

    from interactvlm_amd import preprocess as P
    from interactvlm_amd.constants import CLIP_MEAN, CLIP_STD, SAM_MEAN_PIXEL, SAM_STD_PIXEL

    rng = np.random.default_rng(0)
    for hw in [(1024, 1024), (768, 1024), (1024, 683)]:  # already at longest side 1024 => resize is the identity
        img = rng.integers(0, 256, size=hw + (3,), dtype=np.uint8)
        got, rs = P.sam_preprocess(img, cuda, dtype=torch.float32)
        assert rs == hw and got.shape == (3, 1024, 1024)
        x = torch.from_numpy(img).permute(2, 0, 1).float()  # run_demo.py:65-79
        ref = (x - torch.tensor(SAM_MEAN_PIXEL).view(-1, 1, 1)) / torch.tensor(SAM_STD_PIXEL).view(-1, 1, 1)
        ref = F.pad(ref, (0, 1024 - hw[1], 0, 1024 - hw[0]))
        assert torch.allclose(got.cpu(), ref, atol=1e-5)
        assert P.sam_preprocess(img, cuda)[0].dtype == torch.bfloat16
    # a down-scaling case: shape bookkeeping of ResizeLongestSide (transforms.py:102-113)
    img = rng.integers(0, 256, size=(600, 1500, 3), dtype=np.uint8)
    got, rs = P.sam_preprocess(img, cuda, dtype=torch.float32)
    assert rs == (410, 1024) and float(got[:, 410:, :].abs().max()) == 0.0
    # CLIP: 224x224 input => pure rescale + normalise
    img = rng.integers(0, 256, size=(224, 224, 3), dtype=np.uint8)
    got = P.clip_preprocess(img, cuda, dtype=torch.float32).cpu()
    x = torch.from_numpy(img).permute(2, 0, 1).float() / 255.0
    ref = (x - torch.tensor(CLIP_MEAN).view(-1, 1, 1)) / torch.tensor(CLIP_STD).view(-1, 1, 1)
    assert torch.allclose(got, ref, atol=1e-5)
    # centre crop geometry on a non-square image
    img = rng.integers(0, 256, size=(224, 448, 3), dtype=np.uint8)
    got = P.clip_preprocess(img, cuda, dtype=torch.float32).cpu()
    x = torch.from_numpy(img[:, 112:336]).permute(2, 0, 1).float() / 255.0
    ref = (x - torch.tensor(CLIP_MEAN).view(-1, 1, 1)) / torch.tensor(CLIP_STD).view(-1, 1, 1)
    assert torch.allclose(got, ref, atol=1e-5)


def test_preprocess_vs_reference_golden_non_identity_sizes(hip_lib, cuda, golden_dir):
    """a17 pinned to the reference: SAM inputs against the reference's own ``ResizeLongestSide(1024).apply_image`` +
    run_demo.preprocess, CLIP inputs against HF ``CLIPImageProcessor`` with the openai/clip-vit-large-patch14 settings, at sizes
    where the resizes are NOT identities (600x1500, 480x640, 333x500, 1500x600) - tests/golden/preprocess.npz from
    make_golden.py --only preprocess."""
    import os

    import torch

    from interactvlm_amd import preprocess as P

    d = np.load(os.path.join(golden_dir, "preprocess.npz"))
    for h, w in d["sizes"].tolist():
        img = np.random.default_rng(h * 10000 + w).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        got, rs = P.sam_preprocess(img, cuda, dtype=torch.float32)
        assert tuple(rs) == tuple(d[f"sam/{h}x{w}/resize"].tolist())
        g = got.cpu()
        assert torch.allclose(g[:, ::8, ::8], torch.from_numpy(d[f"sam/{h}x{w}/sub"]), atol=1e-5), (h, w)
        assert abs(float(g.double().sum()) - float(d[f"sam/{h}x{w}/sum"])) < 1e-2 * g.numel() ** 0.5
        c = P.clip_preprocess(img, cuda, dtype=torch.float32).cpu()
        assert c.shape == (3, 224, 224)
        assert torch.allclose(c[:, ::2, ::2], torch.from_numpy(d[f"clip/{h}x{w}/sub"]), atol=2e-5), (h, w)
        assert abs(float(c.double().sum()) - float(d[f"clip/{h}x{w}/sum"])) < 1e-2 * c.numel() ** 0.5
