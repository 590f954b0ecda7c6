"""NumPy restatement of the reference's 2D->3D lift predictors and ``postprocess_masks``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Each function follows the reference line by line (citations into /root/reference):
  * lift_mesh_soft    <- model/components.py:220-277  HumanContact3DPredictor
  * lift_mesh_thresh  <- model/components.py:392-424, 445-489  ObjectMeshContact3DPredictor
  * lift_points       <- model/components.py:289-347  ObjectPCAfford3DPredictor
                         (NumPy twin: preprocess_data/utils_obj_pc.py:47-86)
  * postprocess_masks <- model/segment_anything/modeling/sam.py:137-172
All arithmetic is fp32 and sequential in pixel order (np.add.at == torch CPU scatter_add_).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def sigmoid_f32(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32, copy=False)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def _bary_vote(values, ids, w, nv):
    """3x scatter_add of w_k*m and w_k (components.py:267-274) + per-view normalise."""
    votes = np.zeros(nv, dtype=F32)
    cnt = np.zeros(nv, dtype=F32)
    for k in range(3):
        np.add.at(votes, ids[:, k], (w[:, k] * values).astype(F32))
        np.add.at(cnt, ids[:, k], w[:, k])
    seen = cnt > 0
    votes[seen] = votes[seen] / cnt[seen]
    return votes, seen


def lift_mesh_soft(logits, vid, bary, num_vertices, clamp=20.0):
    """logits [B,V,H,W] f32, vid [V,H,W,3] int, bary [V,H,W,3] f32 -> (pred [B,Nv], nviews [B,Nv])."""
    logits = np.asarray(logits, dtype=F32)
    B, V = logits.shape[:2]
    nv = int(num_vertices)
    pred = np.zeros((B, nv), dtype=F32)
    nviews = np.zeros((B, nv), dtype=F32)
    for b in range(B):
        for v in range(V):
            m = sigmoid_f32(np.clip(logits[b, v], F32(-clamp), F32(clamp))).reshape(-1)  # :250-251
            ids = np.asarray(vid[v]).reshape(-1, 3).astype(np.int64)
            w = np.asarray(bary[v], dtype=F32).reshape(-1, 3)
            keep = ((ids >= 0) & (ids < nv)).all(axis=1)  # :258-259 whole triple valid
            if not keep.any():  # :264-265
                continue
            votes, seen = _bary_vote(m[keep], ids[keep], w[keep], nv)
            pred[b] = pred[b] + votes  # :276
            nviews[b] = nviews[b] + seen.astype(F32)  # :277
    valid = nviews > 0
    pred[valid] = pred[valid] / nviews[valid]  # :240-241
    pred = np.clip(pred, F32(0.0), F32(1.0))  # :242
    return pred, nviews


def lift_mesh_thresh(logits, vid, bary, num_vertices, threshold=0.3):
    """Object-mesh lift. logits [V,H,W] (batch is 1, components.py:436) -> (pred [1,Nv], nviews)."""
    logits = np.asarray(logits, dtype=F32)
    V = logits.shape[0]
    nv = int(num_vertices)
    pred = np.zeros((1, nv), dtype=F32)
    nviews = np.zeros((1, nv), dtype=F32)
    for v in range(V):
        p = sigmoid_f32(logits[v])  # :452 (no clamp)
        sel = p > F32(threshold)  # :453
        ids = np.asarray(vid[v])[sel].reshape(-1, 3).astype(np.int64)  # :456,461
        w = np.asarray(bary[v], dtype=F32)[sel].reshape(-1, 3)
        m = p[sel].reshape(-1)
        keep = ((ids >= 0) & (ids < nv)).all(axis=1)  # :465-466
        if not keep.any():  # :471-472
            continue
        votes, seen = _bary_vote(m[keep], ids[keep], w[keep], nv)
        pred[0] += votes
        nviews[0] += seen.astype(F32)  # :489
    valid = nviews > 0
    pred[valid] /= nviews[valid]  # :421-422 (no clamp)
    return pred, nviews


def lift_points(probs, pid, num_points):
    """probs [B,V,H,W] f32 (already sigmoid-ed by the caller), pid [B,V,H,W] int (-1 none)."""
    probs = np.asarray(probs, dtype=F32)
    B, V = probs.shape[:2]
    n = int(num_points)
    pred = np.zeros((B, n), dtype=F32)
    nviews = np.zeros((B, n), dtype=F32)
    for b in range(B):
        for v in range(V):
            mp = np.asarray(pid[b, v]).astype(np.int64)
            valid = mp != -1  # :329
            pts = mp[valid]
            vals = probs[b, v][valid]
            votes = np.zeros(n, dtype=F32)
            cnt = np.zeros(n, dtype=F32)
            np.add.at(votes, pts, vals)  # :338
            np.add.at(cnt, pts, F32(1.0))  # :339
            seen = cnt > 0
            votes[seen] /= cnt[seen]  # :342-343
            pred[b] += votes
            nviews[b] += seen.astype(F32)
    ok = nviews > 0
    pred[ok] /= nviews[ok]  # :313-314
    return pred, nviews


def _bilinear_axis(n_in: int, n_out: int):
    """torch F.interpolate(mode='bilinear', align_corners=False) source indices/weights."""
    scale = F32(n_in) / F32(n_out)
    dst = np.arange(n_out, dtype=F32)
    src = (dst + F32(0.5)) * scale - F32(0.5)
    src = np.maximum(src, F32(0.0))  # ATen area_pixel_compute_source_index clamps negatives
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    l0 = (F32(1.0) - l1).astype(F32)
    return i0, i1, l0, l1


def bilinear_resize(x, out_hw):
    """x [...,H,W] f32 -> [...,oh,ow]; same operation order as ATen's upsample_bilinear2d CPU
    kernel: out = l0y*(l0x*a + l1x*b) + l1y*(l0x*c + l1x*d)."""
    x = np.asarray(x, dtype=F32)
    H, W = x.shape[-2:]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if (H, W) == (oh, ow):
        return x.copy()
    y0, y1, ly0, ly1 = _bilinear_axis(H, oh)
    x0, x1, lx0, lx1 = _bilinear_axis(W, ow)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    t = top[..., x0] * lx0 + top[..., x1] * lx1
    b = bot[..., x0] * lx0 + bot[..., x1] * lx1
    return (t * ly0[:, None] + b * ly1[:, None]).astype(F32)


def postprocess_masks(low_res, input_size, original_size, img_size=1024):
    """low_res [V,C,h,w] (any float dtype) -> fp32 [V,C,*original_size] (sam.py:161-171)."""
    m = bilinear_resize(np.asarray(low_res, dtype=F32), (img_size, img_size))
    m = m[..., : int(input_size[0]), : int(input_size[1])]
    return bilinear_resize(m, original_size)


def contact_sets(pred, nviews):
    """The vertex-id sets that must match bit-exactly."""
    return {
        "seen": np.flatnonzero(np.asarray(nviews).reshape(-1) > 0),
        "ge_0.5": np.flatnonzero(np.asarray(pred).reshape(-1) >= 0.5),  # utils/eval_utils.py:75
        "gt_0.3": np.flatnonzero(np.asarray(pred).reshape(-1) > 0.3),  # run_demo.py:459
    }
