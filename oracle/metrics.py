"""CPU restatement of the steps right after the path (TEST INFRASTRUCTURE ONLY):
utils/eval_utils.py:63-94 get_h_contact_metrics, utils/utils.py:428-443 convert_contacts."""
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
