"""CPU restatement of the steps right after the path (TEST INFRASTRUCTURE ONLY):
utils/eval_utils.py:63-94 get_h_contact_metrics, :129-151 get_h_geo_metric, utils/utils.py:428-443 convert_contacts.
Pinned to the reference's own functions by tests/golden/metrics.npz (tests/golden/make_golden.py gen_metrics)."""
import torch


def h_contact_metrics(contact_gt, contact_pred, threshold=0.5):
    out = []
    for b in range(contact_gt.shape[0]):
        p = (contact_pred[b].float() >= threshold).float()
        g = (contact_gt[b].float() > 0).float()
        tp, pp, ap = (p * g).sum(), p.sum(), g.sum()
        pr, rc = tp / (pp + 1e-10), tp / (ap + 1e-10)
        out.append([float(2 * pr * rc / (pr + rc + 1e-10)), float(pr), float(rc)])
    return torch.tensor(out)


def convert_contacts(contacts, mapping_matrix):
    """contacts [B, 6890] -> [B, 10475]: torch.bmm(M[None].expand(B), contacts[..., None]).squeeze()."""
    B = contacts.shape[0]
    return torch.bmm(mapping_matrix[None].expand(B, -1, -1), contacts[..., None]).squeeze(-1)


def h_geo_metric(pred, gt, dist):
    """(fp_dist_avg, fn_dist_avg, per-sample [B,2]) - utils/eval_utils.py:129-151 with the distance matrix as an argument
    (the reference reads it from ./data/smpl_neutral_geodesic_dist.npy at import time)."""
    out = torch.zeros(gt.shape[0], 2)
    for b in range(gt.shape[0]):
        cols = gt[b] == 1
        rows = pred[b] >= 0.5
        e = dist[:, cols] if bool(cols.any()) else dist
        e = e[rows, :] if bool(rows.any()) else e
        out[b, 0] = e.min(dim=1)[0].mean()
        out[b, 1] = e.min(dim=0)[0].mean()
    return float(out[:, 0].mean()), float(out[:, 1].mean()), out
