"""View / camera constants and token conventions of the contact-inference path.

Restates (values only) preprocess_data/constants.py:138-382, datasets/base_contact_dataset.py:37-50,
utils/utils.py:12-23 of the reference.  Camera tuples are (distance, elevation°, azimuth°, x_trans, y_trans).
"""
from __future__ import annotations

import numpy as np

IGNORE_LABEL = -1            # utils/utils.py:17-19
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
SEG_TOKEN = "[SEG]"

SAM_MEAN_PIXEL = (123.675, 116.28, 103.53)   # run_demo.py:67-68 (0-255 RGB)
SAM_STD_PIXEL = (58.395, 57.12, 57.375)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # utils/utils.py:14-15 (0-1 RGB)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

_OBJ_CAMS = {
    "frontleft": (2.0, 45.0, 315.0, 0.0, 0.0),
    "frontright": (2.0, 45.0, 45.0, 0.0, 0.0),
    "backleft": (2.0, 330.0, 135.0, 0.0, 0.0),
    "backright": (2.0, 330.0, 225.0, 0.0, 0.0),
}
_OBJ_MESH_CAMS = {k: (1.5,) + v[1:] for k, v in _OBJ_CAMS.items()}  # utils/demo_utils.py:192-197
_HUMAN_CAMS = {
    "topfront": (2.0, 45.0, 315.0, 0.0, 0.0),
    "bottomfront": (2.0, 315.0, 315.0, 0.0, 0.3),
    "topback": (2.0, 45.0, 135.0, 0.0, 0.0),
    "bottomback": (2.0, 315.0, 135.0, 0.0, 0.3),
}


def _z4(names):
    return np.array([[[n]] for n in names])


def _human(folder, suffix=""):
    return {
        "order": "fix", "num_vertices": 6890, "grid_size": np.array([4, 1, 1]), "mask_size": 1024,
        "folder": folder,
        "pixel_to_vertex": "pixel_to_vertex_map_1024.npz", "bary_coords": "bary_coords_map_1024.npz",
        "contact_annot_f": f"contact_label_objectwise{suffix}.pkl",
        "body_parts_annot_f": f"body_parts_objectwise{suffix}.pkl",
        "names": _z4(list(_HUMAN_CAMS)), "ignore_keywords": ["supporting"] if suffix else [],
        "cam_params": dict(_HUMAN_CAMS),
    }


HUMAN_VIEW_DICT = {
    "4MV-Z_Vitru": _human("hcontact_vitruvian"),
    "4MV-Z_Vitru_mv2": _human("hcontact_vitruvian_mv2"),
    "4MV-Z_Vitru_FootGround": _human("hcontact_vitruvian", "_wFootGround"),
}


def _obj_z(mask_size, folder=None, mesh=False):
    d = {"order": "fix", "grid_size": np.array([4, 1, 1]), "mask_size": mask_size,
         "names": _z4(list(_OBJ_CAMS)), "ignore_keywords": [], "cam_params": dict(_OBJ_CAMS)}
    if folder:
        d["folder"] = folder
    if mesh:
        d["mesh_folder"] = "lowpoly_mesh_0507"
        d["mesh_cam_params"] = dict(_OBJ_MESH_CAMS)
    return d


OBJS_VIEW_DICT = {
    "4MV-Z_Fix": {**_obj_z(512, "rendered_points_0917"), "ignore_keywords": ["Refrigerator", "Baseballbat"]},
    "4MV-Z_HM": _obj_z(1024, "rendered_points_heatmap_1025"),
    "4MV-Z_HM1": _obj_z(1024, "rendered_points_heatmap_1102"),
    "4MV-Z_HM2": _obj_z(1024, "rendered_points_heatmap_AP1K0_1104"),
    "4MV-Z_HM_MeshInf": _obj_z(1024),
    "4MV-Z_HM_BM": _obj_z(1024, "rendered_points_heatmap_1025", mesh=True),
    "4MV-Z_HM_BM-L": _obj_z(1024, "rendered_points_heatmap_1025", mesh=True),
}


def normalize_cam_params(cam_params):
    """[d/10, e/360, a/360, (x+1)/2, (y+1)/2]; None -> zeros (base_contact_dataset.py:37-50)."""
    import torch

    if cam_params is None:
        return torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0])
    d, e, a, x, y = cam_params
    return torch.tensor([d / 10.0, e / 360.0, a / 360.0, (x + 1.0) / 2.0, (y + 1.0) / 2.0])


def view_names(view_dict_entry):
    return list(np.asarray(view_dict_entry["names"]).flatten())
