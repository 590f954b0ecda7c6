#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py [--only lift,sam,...]

Every fixture stores the seeded inputs it cannot re-derive plus the reference's outputs.
Weights are never stored: they are ``interactvlm_amd.synth.synth_param(<state-dict key>)``,
poured into the reference's own modules here and into the oracle / HIP path in the tests.
Nothing from the reference's source text is written anywhere — fixtures are data only.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _ref_shims  # noqa: E402
from interactvlm_amd import synth  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------
# F1-F3: lift predictors (model/components.py:195-489)
# ------------------------------------------------------------------------------------------
def gen_lift():
    import torch
    import joblib
    _ref_shims.install()
    import model.components as RC

    V, H, W, NV = 4, 64, 64, 257
    views = RC.HUMAN_VIEW_DICT["4MV-Z_Vitru"]["names"].flatten()
    cwd = os.getcwd()
    for seed in (0, 1, 2):
        vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.4, seed=seed, adversarial=True)
        B = 2 if seed == 0 else 1
        logits = synth.synth_normal(f"lift/logits/{seed}", (B, V, H, W), std=4.0, seed=seed)
        if seed == 1:  # exercise the +-20 clamp (components.py:250)
            logits.reshape(-1)[::97] = 35.0
            logits.reshape(-1)[5::101] = -28.0
        with tempfile.TemporaryDirectory() as td:
            d = os.path.join(td, "data", "hcontact_vitruvian")
            os.makedirs(d)
            np.savez(os.path.join(d, "pixel_to_vertex_map_1024.npz"), **{v: vid[i] for i, v in enumerate(views)})
            np.savez(os.path.join(d, "bary_coords_map_1024.npz"), **{v: bary[i] for i, v in enumerate(views)})
            os.chdir(td)
            try:
                pred = RC.HumanContact3DPredictor("4MV-Z_Vitru", V)
            finally:
                os.chdir(cwd)
        pred.num_vertices = NV  # small-fixture override of the 6890 constant (constants.py:318)
        seg = [torch.from_numpy(logits[b]) for b in range(B)]
        out = pred(seg).numpy()
        _save(f"lift_mesh_soft_s{seed}.npz", logits=logits, vid=vid.astype(np.int32), bary=bary,
              num_vertices=np.int64(NV), expected=out)

        # F2: object-mesh thresholded lift through lift2d_dict.pkl (components.py:392-424)
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "lift2d_dict.pkl")
            joblib.dump({"pixel_to_vertices_map": [vid[i] for i in range(V)],
                         "bary_coords_map": [bary[i] for i in range(V)],
                         "num_vertices": NV}, p)
            om = RC.ObjectMeshContact3DPredictor("4MV-Z_HM", V)
            out_t = om([torch.from_numpy(logits[0])], ds_names=["ocontact"], lift2d_dict_path=p).numpy()
            if seed == 2:  # empty-selection early return (components.py:471-472): all p <= 0.3
                lo = np.full_like(logits[0], -3.0)
                lo[1] = logits[0][1]
                out_e = om([torch.from_numpy(lo)], ds_names=["ocontact"], lift2d_dict_path=p).numpy()
            else:
                lo, out_e = None, None
        extra = {} if lo is None else {"logits_partial": lo, "expected_partial": out_e}
        _save(f"lift_mesh_thresh_s{seed}.npz", logits=logits[0], vid=vid.astype(np.int32), bary=bary,
              num_vertices=np.int64(NV), expected=out_t, **extra)


def gen_lift_points():
    import torch
    _ref_shims.install()
    import model.components as RC

    B, V, H, W, NP = 2, 4, 64, 64, 256
    for seed in (0, 1):
        pid = synth.synth_point_maps(B, V, H, W, NP, fg=0.3, seed=seed)
        probs = synth.synth_uniform(f"lift/probs/{seed}", (B, V, H, W), 0.0, 1.0, seed=seed)
        pc = RC.ObjectPCAfford3DPredictor("4MV-Z_HM", V, num_points=NP)
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for b in range(B):
                row = []
                for v in range(V):
                    mp = os.path.join(td, f"mask_{b}_{v}.png")
                    np.savez(mp.replace("mask", "p2pmap")[:-4] + ".npz", mapping=pid[b, v])
                    row.append(mp)
                paths.append(row)
            out = pc([torch.from_numpy(probs[b]) for b in range(B)], None, paths).numpy()
        # NumPy twin (preprocess_data/utils_obj_pc.py:47-86); its module imports pytorch3d/cv2 at
        # top level for unrelated functions, so those two names are stubbed for the import only.
        import types
        for n, attrs in (("cv2", ()), ("pytorch3d", ()), ("pytorch3d.renderer", (
                "look_at_view_transform", "FoVPerspectiveCameras", "PointsRasterizationSettings",
                "PointsRasterizer", "PointsRenderer", "AlphaCompositor", "NormWeightedCompositor"))):
            if n not in sys.modules:
                m = types.ModuleType(n)
                for a in attrs:
                    setattr(m, a, None)
                sys.modules[n] = m
        from preprocess_data.utils_obj_pc import lift_masks_to_pointcloud
        twin = np.stack([lift_masks_to_pointcloud(list(probs[b]), list(pid[b]), NP) for b in range(B)])
        assert np.array_equal(twin, out), "reference torch predictor and its NumPy twin disagree"
        _save(f"lift_points_s{seed}.npz", probs=probs, pid=pid.astype(np.int32),
              num_points=np.int64(NP), expected=out)


GENERATORS = {"lift": gen_lift, "lift_points": gen_lift_points}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    for name, fn in GENERATORS.items():
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()


if __name__ == "__main__":
    main()
