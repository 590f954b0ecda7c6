#!/usr/bin/env python3
"""Micro-benchmark of the lift / postprocess kernels (HIP events, per-launch averages)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import ops, synth  # noqa: E402

ALG_BYTES_MESH = 4 * 1024 * 1024 * 28 + 2 * 6890 * 4  # 117 495 696 B / image
ALG_BYTES_PC = 4 * 1024 * 1024 * 8
ALG_BYTES_POST = 4 * 256 * 256 * 4 + 4 * 1024 * 1024 * 4


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    V, H, W, NV, NP, B = 4, 1024, 1024, 6890, 2048, a.B
    res = {}
    from interactvlm_amd import synthetic
    for name, patch in (("random", 0), ("clustered6", 6), ("body_render", -1)):
        if patch < 0:  # the bench.py tables: rasterised 6890-vertex body, coherent vertex numbering
            vid_t, bary_t = synthetic.body_lift_tables(dev)
            vid_t, bary_t = vid_t.contiguous(), bary_t.contiguous()
        else:
            vid, bary = synth.synth_mesh_tables(V, H, W, NV, fg=0.4, seed=0, patch=patch)
            vid_t = torch.from_numpy(vid).to(dev, torch.int32)
            bary_t = torch.from_numpy(bary).to(dev)
        lg = torch.randn(B, V, H, W, device=dev) * 4
        t0 = timeit(lambda: ops.LiftPlan(vid_t, bary_t, NV), iters=3, warm=1)
        plan = ops.LiftPlan(vid_t, bary_t, NV)
        t_plan = timeit(lambda: ops.lift_mesh_plan(lg, plan), a.iters)
        t_dense = timeit(lambda: ops.lift_mesh_dense(lg, vid_t, bary_t, NV), a.iters)
        lowr = torch.randn(B, V, 256, 256, device=dev) * 4
        t_fused = timeit(lambda: ops.lift_mesh_plan_lowres(lowr, plan, (1024, 1024), (1024, 1024)), a.iters)
        t_post = timeit(lambda: ops.postprocess_masks(lowr.view(B * V, 1, 256, 256), (1024, 1024), (1024, 1024)), a.iters)
        big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)  # > 256 MB Infinity Cache: flush between launches
        def cold():
            big.zero_()
            ops.lift_mesh_plan(lg, plan)
        t_zero = timeit(lambda: big.zero_(), 10, 2)
        t_cold = timeit(cold, 10, 2) - t_zero
        res[name] = {
            "plan_cold_us": t_cold * 1e6, "plan_cold_alg_GBps": B * ALG_BYTES_MESH / t_cold / 1e9,
            "fused_lowres_us": t_fused * 1e6, "fused_lowres_alg_GBps": B * ALG_BYTES_MESH / t_fused / 1e9,
            "postprocess_us": t_post * 1e6,
            "plan_build_ms": t0 * 1e3, "nnz": plan.nnz, "plan_bytes": plan.bytes(),
            "plan_us": t_plan * 1e6, "plan_alg_GBps": B * ALG_BYTES_MESH / t_plan / 1e9,
            "plan_actual_GBps": (plan.bytes() + B * (plan.nnz * 4)) / t_plan / 1e9,
            "dense_us": t_dense * 1e6, "dense_alg_GBps": B * ALG_BYTES_MESH / t_dense / 1e9,
        }
    pid = torch.from_numpy(synth.synth_point_maps(B, V, H, W, NP, seed=0)).to(dev, torch.int32)
    pr = torch.rand(B, V, H, W, device=dev)
    t = timeit(lambda: ops.lift_points(pr, pid, NP), a.iters)
    res["points"] = {"us": t * 1e6, "alg_GBps": B * ALG_BYTES_PC / t / 1e9,
                     "note": "streaming kernel (single-use maps): memset + vote kernel + finalize, back to back"}
    # a RENDERED pixel -> point map (what the p2pmap files hold: utils_obj_pc.py:88-113 - splats of ~5 pixels, coherent along a row):
    # 2048 points of the stand-in body under the four object cameras, radius 0.005 NDC
    from interactvlm_amd import render
    from interactvlm_amd.constants import OBJS_VIEW_DICT
    bv, _ = synthetic.body_mesh()
    pts = render.normalize_mesh(bv[torch.linspace(0, bv.shape[0] - 1, NP).long()].to(dev)).contiguous()
    cams = OBJS_VIEW_DICT["4MV-Z_HM"]["cam_params"]
    pid_r = torch.stack([render.rasterize_points(pts, cams[n], 0.005, (H, W)) for n in list(cams)[:V]])[None].contiguous()
    t = timeit(lambda: ops.lift_points(pr[:1], pid_r, NP), a.iters)
    res["points_rendered_map"] = {"us": t * 1e6, "alg_GBps": ALG_BYTES_PC / t / 1e9, "foreground": float((pid_r >= 0).float().mean()),
                                  "note": "streaming kernel on a rasterised 2048-point cloud (runs of equal ids are summed in registers)"}
    pplan_r = ops.LiftPlan.from_points(pid_r[0], NP)
    t = timeit(lambda: ops.lift_points_plan(pr[:1], pplan_r), a.iters)
    res["points_rendered_map_plan"] = {"us": t * 1e6, "alg_GBps": ALG_BYTES_PC / t / 1e9, "nnz": pplan_r.nnz}
    pplan = ops.LiftPlan.from_points(pid[0], NP)
    tp = timeit(lambda: ops.lift_points_plan(pr[:1], pplan), a.iters)
    tb = timeit(lambda: ops.LiftPlan.from_points(pid[0], NP), iters=3, warm=1)
    res["points_plan"] = {"us": tp * 1e6, "alg_GBps": ALG_BYTES_PC / tp / 1e9, "nnz": pplan.nnz, "plan_bytes": pplan.bytes(),
                          "actual_GBps": (pplan.bytes() + pplan.nnz * 4) / tp / 1e9, "plan_build_ms": tb * 1e3,
                          "note": "point-major plan of a cached p2pmap set (second sight on): one gather launch, no atomics"}
    low = torch.randn(B * V, 1, 256, 256, device=dev)
    t = timeit(lambda: ops.postprocess_masks(low, (1024, 1024), (1024, 1024)), a.iters)
    res["postprocess"] = {"us": t * 1e6, "alg_GBps": B * ALG_BYTES_POST / t / 1e9}
    lowb = low.to(torch.bfloat16)
    t = timeit(lambda: ops.postprocess_masks(lowb, (1024, 683), (1500, 1000)), a.iters)
    res["postprocess_general"] = {"us": t * 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
