"""Deterministic synthetic tensors keyed by state-dict name.

There are no checkpoints, tokenizers or datasets in this environment (no network, gated
downloads), so every parity test and the benchmark run on *seeded synthetic* weights and
inputs.  A tensor is a pure function of ``(key, shape, seed)`` so that the golden-fixture
generator (which fills the *reference's* modules, ``tests/golden/make_golden.py``), the CPU
oracle and the HIP path all see bit-identical weights without a multi-GB fixture ever
being committed.

Keys are the reference's state-dict names (``model.visual_model.mask_decoder...`` —
SURVEY.md §8b), which are also what a released checkpoint would carry.
"""
from __future__ import annotations

import zlib

import numpy as np

__all__ = ["synth_normal", "synth_uniform", "synth_param", "fill_state_dict"]


def _rng(key: str, seed: int) -> np.random.Generator:
    h = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.Philox(key=[h, seed & 0xFFFFFFFF]))


def synth_normal(key: str, shape, std: float = 1.0, seed: int = 0) -> np.ndarray:
    """fp32 N(0, std^2) tensor that depends only on (key, shape, seed)."""
    shape = tuple(int(s) for s in shape)
    return (_rng(key, seed).standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def synth_uniform(key: str, shape, lo: float = 0.0, hi: float = 1.0, seed: int = 0) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    u = _rng(key, seed).random(shape, dtype=np.float32)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def synth_param(key: str, shape, seed: int = 0) -> np.ndarray:
    """Synthetic value for one parameter / buffer of the model.

    * 1-D ``*.weight``  -> normalisation scale, 1 + 0.05 n   (only norms have 1-D weights)
    * 1-D anything else -> 0.02 n                             (biases, CLIP class embedding)
    * ``rel_pos_*``, ``pos_embed``, ``position_embedding``, embeddings of tokens -> 0.02 n ...
      except they must be O(1) to matter, so 0.5 n for rel_pos tables
    * >=2-D             -> n / sqrt(fan_in), fan_in = prod(shape[1:])  (keeps activations O(1)
      through 32-40 layers so that mask logits are not degenerate)
    """
    shape = tuple(int(s) for s in shape)
    if len(shape) == 0:
        return synth_normal(key, (), 1.0, seed)
    if len(shape) == 1:
        n = synth_normal(key, shape, 1.0, seed)
        if key.endswith("weight"):
            return (1.0 + 0.05 * n).astype(np.float32)
        return (0.02 * n).astype(np.float32)
    if "rel_pos" in key:
        return synth_normal(key, shape, 0.5, seed)
    if "positional_encoding_gaussian_matrix" in key:
        return synth_normal(key, shape, 1.0, seed)
    if "embed" in key and "proj" not in key and "patch_embedding" not in key:
        # token / position / learned-prompt embeddings (rows are vectors, not fan-in matrices)
        return synth_normal(key, shape, 0.5, seed)
    fan_in = int(np.prod(shape[1:]))
    return synth_normal(key, shape, 1.0 / np.sqrt(max(fan_in, 1)), seed)


def fill_state_dict(module, seed: int = 0, prefix: str = "") -> None:
    """Overwrite every parameter and buffer of a torch ``nn.Module`` with ``synth_param``.

    ``prefix`` is prepended to the module-relative name so that sub-modules built in
    isolation (e.g. only the mask decoder) receive the same values they would inside the
    full model.
    """
    import torch

    with torch.no_grad():
        # state_dict() = parameters + PERSISTENT buffers only (e.g. SAM's PE gaussian matrix); derived
        # non-persistent buffers such as rotary ``inv_freq`` must keep their computed values.
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            v = synth_param(prefix + name, tuple(t.shape), seed)
            t.copy_(torch.from_numpy(v).to(t.dtype))


# --------------------------------------------------------------------------------------
# synthetic lift tables (stand-ins for ./data/<folder>/pixel_to_vertex_map_1024.npz etc.)
# --------------------------------------------------------------------------------------
def synth_mesh_tables(V, H, W, num_vertices, fg=0.4, seed=0, adversarial=True, patch=0):
    """Random pixel->vertex-triple / barycentric tables in the reference's on-disk format.

    Returns (vid int64 [V,H,W,3], bary float32 [V,H,W,3]).  Background pixels carry ids -1 and
    bary -1 (preprocess_data/render_mesh_utils.py:146, pytorch3d empty-pixel convention).
    adversarial=True also plants (a) triples with one invalid id, (b) ids >= num_vertices and
    (c) zero-weight hits, exercising components.py:258-259 and the ``cnt > 0`` visibility rule.
    patch>0 makes ids piecewise-constant over patch x patch pixel blocks (spatial locality like
    a real rasterised mesh); patch=0 is i.i.d. per pixel (worst-case locality).
    """
    rng = _rng(f"mesh_tables/{V}x{H}x{W}/{num_vertices}/{fg}/{patch}", seed)
    if patch and patch > 1:
        hh, ww = (H + patch - 1) // patch, (W + patch - 1) // patch
        tri = rng.integers(0, num_vertices, size=(V, hh, ww, 3), dtype=np.int64)
        tri = np.repeat(np.repeat(tri, patch, axis=1), patch, axis=2)[:, :H, :W]
        fgm = rng.random((V, hh, ww)) < fg
        fgm = np.repeat(np.repeat(fgm, patch, axis=1), patch, axis=2)[:, :H, :W]
    else:
        tri = rng.integers(0, num_vertices, size=(V, H, W, 3), dtype=np.int64)
        fgm = rng.random((V, H, W)) < fg
    g = rng.gamma(1.0, 1.0, size=(V, H, W, 3)).astype(np.float32) + np.float32(1e-3)
    bary = (g / g.sum(-1, keepdims=True)).astype(np.float32)
    vid = np.where(fgm[..., None], tri, -1).astype(np.int64)
    bary = np.where(fgm[..., None], bary, np.float32(-1.0)).astype(np.float32)
    if adversarial:
        n = V * H * W
        flat_v = vid.reshape(n, 3)
        flat_b = bary.reshape(n, 3)
        k = max(4, n // 200)
        idx = rng.choice(n, size=3 * k, replace=False)
        flat_v[idx[:k], rng.integers(0, 3, size=k)] = -1            # partially invalid triple
        flat_v[idx[k:2 * k], rng.integers(0, 3, size=k)] = num_vertices + rng.integers(0, 5, size=k)
        zw = idx[2 * k:]
        flat_b[zw] = np.where(flat_v[zw] >= 0, np.float32(0.0), flat_b[zw])  # zero-weight hits
    return vid, bary


def synth_point_maps(B, V, H, W, num_points, fg=0.3, seed=0):
    """Random pixel->point maps (p2pmap_*.npz['mapping'] format): int64 [B,V,H,W], -1 = none."""
    rng = _rng(f"point_maps/{B}x{V}x{H}x{W}/{num_points}/{fg}", seed)
    pid = rng.integers(0, num_points, size=(B, V, H, W), dtype=np.int64)
    fgm = rng.random((B, V, H, W)) < fg
    return np.where(fgm, pid, -1).astype(np.int64)
