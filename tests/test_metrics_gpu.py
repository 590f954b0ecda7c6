"""§8f-2 on the device: F1/precision/recall and the SMPL->SMPL-X transfer as SpMV, vs the CPU restatement."""
import pytest

pytestmark = pytest.mark.gpu


def test_contact_prf_and_smplx_transfer(hip_lib, cuda):
    import torch

    from interactvlm_amd import ops
    from oracle import metrics as OM

    g = torch.Generator().manual_seed(0)
    pred = torch.rand(3, 6890, generator=g)
    gt = (torch.rand(3, 6890, generator=g) > 0.7).float()
    gt[2] = 0  # no positives: precision/recall/f1 collapse to 0 like the reference's epsilon form
    got = ops.contact_prf(gt.to(cuda), pred.to(cuda)).cpu()
    assert torch.allclose(got, OM.h_contact_metrics(gt, pred), atol=1e-6)
    # SMPL -> SMPL-X style matrix: 3 barycentric weights per row
    rows, cols = 10475, 6890
    M = torch.zeros(rows, cols)
    idx = torch.randint(0, cols, (rows, 3), generator=g)
    w = torch.rand(rows, 3, generator=g)
    w = w / w.sum(1, keepdim=True)
    M.scatter_add_(1, idx, w)
    sp = ops.SparseRows(M, cuda)
    y = sp.matvec(pred.to(cuda)).cpu()
    assert torch.allclose(y, OM.convert_contacts(pred, M), atol=1e-5)
