"""Synthetic configurations, weights and inputs of the benchmark / smoke / parity workloads.

No checkpoint, tokenizer or dataset can be fetched here (no network, gated downloads — SURVEY.md §7), so the
workload of BASELINE.json configs[1] is reproduced in SHAPE: LLaMA-2-7B + CLIP ViT-L/14 + SAM ViT-H dims with
seeded random weights, a 75-id prompt with one image placeholder, and a forced 24-token answer containing
[SEG] (random weights never emit it; the forced schedule mirrors the reference's inference_type='forward').
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

from .constants import HUMAN_VIEW_DICT, normalize_cam_params
from .weights import ClipCfg, IvlmCfg, LlamaCfg, SamEncCfg, ivlm_spec

BF16 = torch.bfloat16


def config_7b() -> IvlmCfg:
    """interactvlm-3d-hcontact-damon shape: LLaVA-1.5-7B + ViT-L/14 + SAM-H (BASELINE.json configs[1])."""
    return IvlmCfg(llama=LlamaCfg(), clip=ClipCfg(), sam=SamEncCfg())


def config_13b() -> IvlmCfg:
    return IvlmCfg(llama=LlamaCfg(hidden=5120, layers=40, heads=40, inter=13824), clip=ClipCfg(), sam=SamEncCfg())


def config_tiny() -> IvlmCfg:
    """Small but structurally complete (smoke test): real head dims (128 / 64 / 80), windows, 64x64 grid."""
    return IvlmCfg(llama=LlamaCfg(hidden=256, layers=2, heads=2, inter=512, vocab=32003),
                   clip=ClipCfg(hidden=128, layers=3, heads=2, inter=256),
                   sam=SamEncCfg(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,)))


def device_weights(cfg: IvlmCfg, device, seed: int = 0, dtype=BF16):
    """Random weights generated ON THE DEVICE with the fan-in rule of synth.synth_param (not the same values:
    a 7B fp32 CPU draw would take minutes and 27 GB of host RAM)."""
    out = {}
    for key, shape in ivlm_spec(cfg).items():
        g = torch.Generator(device=device)
        g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        n = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if len(shape) == 1:
            t = 1.0 + 0.05 * n if key.endswith("weight") else 0.02 * n
        elif "rel_pos" in key or ("embed" in key and "proj" not in key and "patch_embedding" not in key):
            t = 0.5 * n
        elif "positional_encoding_gaussian_matrix" in key:
            t = n
        else:
            t = n / float(np.prod(shape[1:])) ** 0.5
        out[key] = t.to(dtype if "gaussian_matrix" not in key else torch.float32)
        del n
    return out


def prompt_ids(cfg: IvlmCfg, n_prompt: int = 75, n_answer: int = 24, seed: int = 0):
    """(prompt ids [1, n_prompt] with <im_start> -200 <im_end> at 35..37, forced answer ids [n_answer] with
    [SEG] at position 22 and EOS last) — SURVEY.md §8(d) config 2."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, 31000, size=n_prompt)
    ids[0] = 1
    ids[35], ids[36], ids[37] = cfg.im_start_idx, -200, cfg.im_end_idx
    ans = rng.integers(3, 31000, size=n_answer)
    ans[min(22, n_answer - 2)] = cfg.seg_token_idx
    ans[-1] = 2
    return torch.from_numpy(ids)[None], [int(t) for t in ans]


def human_cam_params(view_type="4MV-Z_Vitru"):
    cams = HUMAN_VIEW_DICT[view_type]["cam_params"]
    return torch.stack([normalize_cam_params(c) for c in cams.values()])[None]  # [1,V,5]


def images(cfg: IvlmCfg, device, seed: int = 0, batch: int = 1):
    """(images_clip [B,3,224,224], images [B,V,3,S,S]) bf16 on device, standard-normal (already 'normalised')."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    S = cfg.sam.img_size
    ic = torch.randn(batch, 3, cfg.clip.image_size, cfg.clip.image_size, generator=g, device=device).to(BF16)
    im = torch.randn(batch, cfg.multiview_channels, 3, S, S, generator=g, device=device).to(BF16)
    return ic, im


def body_mesh(n_rings: int = 82, n_seg: int = 84, semi_axes=(0.5, 0.85, 0.22)):
    """A closed 6890-vertex / 13776-face stand-in for the SMPL template (same counts: 2 + 82*84 vertices,
    2*84 + 81*84*2 faces): a lobed ellipsoid about the size of a vitruvian-pose body, vertices numbered ring by ring
    so that neighbouring ids are neighbours on the surface, as in SMPL.  -> (verts f32 [Nv,3], faces i32 [Nf,3])."""
    th = (np.arange(1, n_rings + 1, dtype=np.float64) / (n_rings + 1)) * np.pi
    ph = np.arange(n_seg, dtype=np.float64) / n_seg * 2 * np.pi
    T, P = np.meshgrid(th, ph, indexing="ij")
    lobes = 1.0 + 0.25 * np.cos(4 * P) * np.sin(T) ** 2  # limb-like bulges
    a, b, c = semi_axes
    ring = np.stack([a * lobes * np.sin(T) * np.cos(P), b * np.cos(T), c * lobes * np.sin(T) * np.sin(P)], -1)
    verts = np.concatenate([[[0.0, b, 0.0]], ring.reshape(-1, 3), [[0.0, -b, 0.0]]]).astype(np.float32)
    idx = lambda r, s: 1 + r * n_seg + (s % n_seg)
    faces = []
    for s in range(n_seg):
        faces.append((0, idx(0, s + 1), idx(0, s)))
        faces.append((len(verts) - 1, idx(n_rings - 1, s), idx(n_rings - 1, s + 1)))
    for r in range(n_rings - 1):
        for s in range(n_seg):
            faces.append((idx(r, s), idx(r, s + 1), idx(r + 1, s)))
            faces.append((idx(r, s + 1), idx(r + 1, s + 1), idx(r + 1, s)))
    return torch.from_numpy(verts), torch.tensor(faces, dtype=torch.int32)


def body_lift_tables(device, view_type="4MV-Z_Vitru", image_size=(1024, 1024)):
    """pixel_to_vertex_map_1024 / bary_coords_map_1024 of ``body_mesh`` under the four hcontact cameras, produced by
    the HIP rasteriser (render.human_lift_tables) -> (vid i32 [4,H,W,3], bary f32 [4,H,W,3]) on the device."""
    from . import render

    v, f = body_mesh()
    return render.human_lift_tables(v.to(device), f.to(device), view_type, image_size)
