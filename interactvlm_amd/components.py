"""Host-side mirror of the reference's lift predictors (``model/components.py:195-489``).

Same class names, constructor arguments, ``forward`` signatures, return shapes and error behaviour
as the reference, so a caller can swap the import; all arithmetic runs in libivlm_hip.so.
What is different by design (MI355X-first):
  * constant tables are narrowed to int32, moved to HBM and inverted into a vertex-major CSR
    ("lift plan") ONCE, instead of the reference's ~150 MB host->device copy per view per call
    (components.py:253-254);
  * the object-mesh variant never syncs to the host per view (components.py:455-457): the p>thr
    selection is a predicate inside the kernel;
  * single-use tables (a fresh ``lift2d_dict.pkl``) stream through the dense atomic kernel, tables
    that repeat are cached as plans (keyed by path + mtime).
"""
from __future__ import annotations

import collections
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .constants import HUMAN_VIEW_DICT, OBJS_VIEW_DICT, view_names


def _stack_views(seg_maps: Sequence[torch.Tensor]) -> torch.Tensor:
    """list of [V,H,W] -> contiguous fp32 [B,V,H,W] (a view when B == 1 and already fp32)."""
    if len(seg_maps) == 1:
        t = seg_maps[0].unsqueeze(0)
    else:
        t = torch.stack(list(seg_maps), dim=0)
    return t.to(torch.float32).contiguous()


class HumanContact3DPredictor(torch.nn.Module):
    """Soft bary-weighted multi-view vote onto the 6890 SMPL-topology vertices (components.py:195-277)."""

    def __init__(self, hC_sam_view_type, multiview_channels, threshold=0.3, metadata_root="./data", tables=None,
                 device=None):
        super().__init__()
        self.hC_sam_view_type = hC_sam_view_type
        self.multiview_channels = multiview_channels
        self.threshold = threshold
        entry = HUMAN_VIEW_DICT[hC_sam_view_type]
        self.views = view_names(entry)
        self.num_vertices = entry["num_vertices"]
        if tables is None:  # same files / keys as the reference (components.py:203-218)
            folder = os.path.join(metadata_root, entry["folder"])
            p2v = np.load(os.path.join(folder, entry["pixel_to_vertex"]))
            bc = np.load(os.path.join(folder, entry["bary_coords"]))
            vid = np.stack([p2v[v] for v in self.views])
            bary = np.stack([bc[v] for v in self.views])
        else:
            vid, bary = tables
        # kept as plain attributes (not buffers) like the reference => absent from checkpoints
        self.pixel_to_vertex_map = vid if torch.is_tensor(vid) else torch.as_tensor(np.asarray(vid))
        self.bary_coord_map = bary if torch.is_tensor(bary) else torch.as_tensor(np.asarray(bary))
        self._plan = None
        self._plan_nv = None
        self._device = device

    def _get_plan(self, device) -> ops.LiftPlan:
        if self._plan is None or self._plan_nv != self.num_vertices or self._plan.row_ptr.device != device:
            vid = self.pixel_to_vertex_map.to(device=device, dtype=torch.int32).contiguous()
            bary = self.bary_coord_map.to(device=device, dtype=torch.float32).contiguous()
            self._plan = ops.LiftPlan(vid, bary, self.num_vertices)
            self._plan_nv = self.num_vertices
        return self._plan

    def forward(self, seg_maps, ds_names=None):
        batch_size = len(seg_maps)
        ds_names = ds_names if ds_names is not None else ["hcontact"] * batch_size
        device = seg_maps[0].device
        dtype = seg_maps[0].dtype
        logits = _stack_views(seg_maps)[:, : self.multiview_channels].contiguous()
        out = ops.lift_mesh_plan(logits, self._get_plan(device), mode=0, param=20.0)
        skip = [("hcontact" not in n) for n in ds_names]
        if any(skip):  # components.py:230-231: other samples keep zeros
            out[torch.tensor(skip, device=device)] = 0.0
        return out.to(dtype)


    def forward_lowres(self, low_res_list, input_size, original_size, img_size=1024):
        """Extension: same result as forward([postprocess_masks(l) ...]) without reading the
        full-resolution masks back: low_res_list = B x [V,1,h,w] (or [V,h,w]) low-res decoder outputs."""
        lows = [l.reshape(l.shape[0], l.shape[-2], l.shape[-1]) for l in low_res_list]
        low = torch.stack(lows, 0)[:, : self.multiview_channels].contiguous()
        return ops.lift_mesh_plan_lowres(low, self._get_plan(low.device), input_size, original_size, img_size)


class ObjectMeshContact3DPredictor(torch.nn.Module):
    """Hard-threshold (p > 0.3) bary vote onto an arbitrary object mesh (components.py:350-489)."""

    def __init__(self, oC_sam_view_type, multiview_channels, threshold=0.3):
        super().__init__()
        self.multiview_channels = multiview_channels
        self.view_names = view_names(OBJS_VIEW_DICT[oC_sam_view_type])
        self.threshold = threshold
        # lift plans of lift2d_dict.pkl files seen at least TWICE, least-recently-used first (a dataset evaluation visits one
        # pkl per object: a single-use table goes through the streaming dense kernel and is never inverted or kept; ~40 MB of
        # HBM per cached plan, bounded)
        self._plans = collections.OrderedDict()
        self._seen_once = collections.OrderedDict()
        self.max_plans = 8

    # -- table sources ---------------------------------------------------------------------
    @staticmethod
    def _load_lift2d(path):
        import joblib

        d = joblib.load(path)
        return (np.stack(d["pixel_to_vertices_map"]), np.stack(d["bary_coords_map"]), int(d["num_vertices"]))

    @staticmethod
    def _load_train(mask_paths):
        vids, barys, nv = [], [], None
        for mp in mask_paths:
            m = np.load(mp.replace("mask", "p2vmap").replace(".png", ".npz"))
            vids.append(m["pixel_to_vertices_map"])
            barys.append(m["bary_coords_map"])
            nv = int(m["num_vertices"])
        return np.stack(vids), np.stack(barys), nv

    def _lift(self, seg_maps, tables, cache_key=None):
        vid_np, bary_np, nv = tables
        device = seg_maps[0].device
        dtype = seg_maps[0].dtype
        logits = _stack_views(seg_maps)[:, : self.multiview_channels].contiguous()
        V = logits.shape[1]
        plan = self._plans.get(cache_key) if cache_key is not None else None
        if plan is not None:
            self._plans.move_to_end(cache_key)
            out = ops.lift_mesh_plan(logits, plan, mode=1, param=self.threshold)
        else:
            vid = torch.as_tensor(vid_np[:V]).to(device=device, dtype=torch.int32).contiguous()
            bary = torch.as_tensor(bary_np[:V]).to(device=device, dtype=torch.float32).contiguous()
            if cache_key is not None and cache_key in self._seen_once:  # the file came back: invert once, reuse from now on
                del self._seen_once[cache_key]
                plan = ops.LiftPlan(vid, bary, nv)
                self._plans[cache_key] = plan
                while len(self._plans) > self.max_plans:
                    self._plans.popitem(last=False)
                out = ops.lift_mesh_plan(logits, plan, mode=1, param=self.threshold)
            else:  # first sight (or no file identity): stream the dense tables once, keep nothing on the device
                if cache_key is not None:
                    self._seen_once[cache_key] = True
                    while len(self._seen_once) > 4096:
                        self._seen_once.popitem(last=False)
                out = ops.lift_mesh_dense(logits, vid, bary, nv, mode=1, param=self.threshold)
        return out.to(dtype)

    def forward_inference(self, seg_maps, device, dtype, ds_names=None, lift2d_dict_path=None):
        st = os.stat(lift2d_dict_path)
        key = ("lift2d", os.path.abspath(lift2d_dict_path), st.st_mtime_ns, st.st_size)
        if key in self._plans:
            return self._lift(seg_maps, (None, None, self._plans[key].num_vertices), cache_key=key)
        tables = self._load_lift2d(lift2d_dict_path)
        print(f"Num vertices: {tables[2]}")
        return self._lift(seg_maps, tables, cache_key=key)

    def forward_train(self, seg_maps, device, dtype, ds_names=None, mask_paths_list=None):
        return self._lift(seg_maps, self._load_train(mask_paths_list[0]))

    def forward(self, seg_maps, ds_names=None, mask_paths_list=None, lift2d_dict_path=None):
        device = seg_maps[0].device
        dtype = seg_maps[0].dtype
        if "ocontact" not in ds_names[0]:
            return torch.zeros((1, 0), device=device, dtype=dtype)
        batch_size = len(seg_maps)
        assert batch_size == 1, "Batch size should be 1 since different objects have different number of vertices"
        if lift2d_dict_path is not None:
            return self.forward_inference(seg_maps, device, dtype, ds_names, lift2d_dict_path)
        elif mask_paths_list is not None:
            return self.forward_train(seg_maps, device, dtype, ds_names, mask_paths_list)
        else:
            raise ValueError(
                "Either lift2d_dict_path or mask_paths_list must be provided for ObjectMeshContact3DPredictor")


class ObjectPCAfford3DPredictor(torch.nn.Module):
    """Pixel->point mean vote onto a 2048-point cloud (components.py:279-347)."""

    def __init__(self, oC_sam_view_type, multiview_channels, num_points=2048, threshold=0.3):
        super().__init__()
        self.num_points = num_points
        self.multiview_channels = multiview_channels
        self.threshold = threshold
        _ = OBJS_VIEW_DICT[oC_sam_view_type]["mask_size"]
        self._map_cache = {}
        # a p2pmap set that comes back (second sight of the same files) is inverted once into a point-major plan and kept (LRU of
        # 16): the plan gather is deterministic and moves ~1/10 of the bytes of the streaming kernel; single-use maps stream
        self._plans = {}
        self._seen = {}

    def _maps_for(self, mask_paths, device):
        key = tuple(mask_paths[: self.multiview_channels])
        t = self._map_cache.get(key)
        if t is None:
            maps = [np.load(mp.replace("mask", "p2pmap")[:-4] + ".npz")["mapping"] for mp in key]
            t = torch.as_tensor(np.stack(maps)).to(device=device, dtype=torch.int32).contiguous()
            if len(self._map_cache) > 64:
                self._map_cache.clear()
            self._map_cache[key] = t
        return t

    def forward(self, seg_maps, ds_names=None, mask_paths_list=None):
        device = seg_maps[0].device
        dtype = seg_maps[0].dtype
        batch_size = len(seg_maps)
        ds_names = ds_names if ds_names is not None else ["oafford"] * batch_size
        out = torch.zeros((batch_size, self.num_points), device=device, dtype=torch.float32)
        idx = [b for b, n in enumerate(ds_names) if "oafford" in n]
        if idx:
            probs = _stack_views([seg_maps[b] for b in idx])[:, : self.multiview_channels].contiguous()
            keys = [tuple(mask_paths_list[b][: self.multiview_channels]) for b in idx]
            stream_rows = []
            for r, (b, key) in enumerate(zip(idx, keys)):
                plan = self._plans.get(key)
                if plan is None and self._seen.get(key, 0) >= 1:  # second sight: invert once
                    plan = ops.LiftPlan.from_points(self._maps_for(mask_paths_list[b], device), self.num_points)
                    if len(self._plans) >= 16:
                        self._plans.pop(next(iter(self._plans)))
                    self._plans[key] = plan
                self._seen[key] = self._seen.get(key, 0) + 1
                if len(self._seen) > 4096:
                    self._seen.clear()
                if plan is not None:
                    out[b] = ops.lift_points_plan(probs[r: r + 1], plan)[0]
                else:
                    stream_rows.append(r)
            if stream_rows:
                pid = torch.stack([self._maps_for(mask_paths_list[idx[r]], device) for r in stream_rows]).contiguous()
                res = ops.lift_points(probs[stream_rows].contiguous() if len(stream_rows) != len(idx) else probs, pid, self.num_points)
                out[torch.tensor([idx[r] for r in stream_rows], device=device)] = res
        return out.to(dtype)
