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
