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


def test_metrics_vs_reference_golden(hip_lib, cuda, golden_dir):
    """Device F1 / precision / recall and the geodesic fp / fn distances against the numbers the reference's own
    get_h_contact_metrics / get_o_contact_metrics / get_h_geo_metric returned (tests/golden/make_golden.py gen_metrics):
    empty predictions, empty ground truth, the 0.5 boundary, non-binary gt values, a non-symmetric distance matrix."""
    import os

    import numpy as np
    import torch

    from interactvlm_amd import ops

    d = np.load(os.path.join(golden_dir, "metrics.npz"))
    pred, gt, dist = (torch.from_numpy(d[k]).to(cuda) for k in ("pred", "gt", "dist"))
    fp, fn, per = ops.h_geo_metric(pred, gt, dist)
    np.testing.assert_allclose(per.cpu().numpy(), d["geo_per_sample"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose([fp, fn], d["geo_batch"], rtol=2e-6)
    prf = ops.contact_prf(gt, pred).cpu().numpy()
    np.testing.assert_allclose(prf, d["prf_per_sample"], atol=1e-6)
    np.testing.assert_allclose(prf, d["prf_o_per_sample"], atol=1e-6)  # the object variant is the same arithmetic


def test_geo_metric_full_size_properties(hip_lib, cuda):
    """6890 x 6890 (the real matrix size, 190 MB): against the torch restatement, plus structural properties - a prediction
    equal to the ground truth has zero fp and fn distance on a metric with zero diagonal."""
    import torch

    from interactvlm_amd import ops
    from oracle import metrics as OM

    n = 6890
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(n, 3, generator=g)
    dist = torch.cdist(pts, pts).contiguous()
    dist.fill_diagonal_(0.0)  # (cdist's own diagonal is ~1e-4, not 0)
    gt = (torch.rand(2, n, generator=g) < 0.05).float()
    pred = torch.rand(2, n, generator=g) * 0.6
    pred[1] = gt[1]
    fp, fn, per = ops.h_geo_metric(pred.to(cuda), gt.to(cuda), dist.to(cuda))
    efp, efn, eper = OM.h_geo_metric(pred, gt, dist)
    assert torch.allclose(per.cpu(), eper, rtol=1e-5, atol=1e-6) and abs(fp - efp) < 1e-5 and abs(fn - efn) < 1e-5
    assert float(per[1, 0]) == 0.0 and float(per[1, 1]) == 0.0
