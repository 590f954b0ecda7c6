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
    res["points"] = {"us": t * 1e6, "alg_GBps": B * ALG_BYTES_PC / t / 1e9}
    low = torch.randn(B * V, 1, 256, 256, device=dev)
    t = timeit(lambda: ops.postprocess_masks(low, (1024, 1024), (1024, 1024)), a.iters)
    res["postprocess"] = {"us": t * 1e6, "alg_GBps": B * ALG_BYTES_POST / t / 1e9}
    lowb = low.to(torch.bfloat16)
    t = timeit(lambda: ops.postprocess_masks(lowb, (1024, 683), (1500, 1000)), a.iters)
    res["postprocess_general"] = {"us": t * 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
