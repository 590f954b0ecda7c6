"""Optional heads of the reference's ``ModifiedSAM`` (model/InteractVLM.py:20-44), off in every released configuration
(scripts/run_train.sh passes neither ``--use_feat_fusion`` nor ``--use_uncertainty``) but part of the inference path when a
checkpoint was trained with them.  Both reference modules cast their inputs to bf16 and so only run inside the bf16 model: the
kernels round where those modules round.  All arithmetic in libivlm_hip.so.

  UncertaintyHead   components.py:40-78   per-pixel MLP 256 -> 64 -> 16 -> 1 (ReLU, ReLU, Softplus) + the caller's bilinear resize
  SamFusionHead     components.py:79-153  cross-attention of the SAM tokens over projected LLaVA hidden states, residual add
"""
from __future__ import annotations

import torch

from . import ops
from .weights import SAM_PREFIX

BF16, F32 = torch.bfloat16, torch.float32


def _dev(t, device):
    return t.to(device=device, dtype=BF16).contiguous()


class UncertaintyHead:
    def __init__(self, w, device, prefix=SAM_PREFIX + ".uncertainty", grid=64):
        self.grid = grid
        self.p = [_dev(w[f"{prefix}.linear{i}.{n}"], device) for i in (1, 2, 3) for n in ("weight", "bias")]

    def __call__(self, image_embeddings):
        """image_embeddings fp32 [V, grid*grid, 256] (channels last, as the encoder writes them) -> [V,1,grid,grid] fp32 holding the
        bf16 module's values (UncertaintyModule.forward)."""
        V = image_embeddings.shape[0]
        m = ops.uncertainty_mlp(image_embeddings.to(F32).contiguous(), *self.p)
        return m.view(V, 1, self.grid, self.grid)

    def resized(self, image_embeddings, size):
        """... followed by F.interpolate(size, bilinear, align_corners=False) -> bf16 [V,1,H,W] (InteractVLM.py:446-448)."""
        return ops.resize_bilinear(self(image_embeddings), size, dtype=BF16)


class SamFusionHead:
    def __init__(self, w, device, prefix=SAM_PREFIX + ".fusion", num_heads=8):
        g = lambda n: (_dev(w[f"{prefix}.{n}.weight"], device), _dev(w[f"{prefix}.{n}.bias"], device))
        self.sam_proj, self.llava_proj, self.output_proj = g("sam_proj"), g("llava_proj"), g("output_proj")
        self.q, self.k, self.v, self.o = (g("fusion." + n) for n in ("q_proj", "k_proj", "v_proj", "out_proj"))
        self.num_heads = num_heads
        self.llava_dim = self.llava_proj[0].shape[1]

    def __call__(self, image_embeddings, llava_features):
        """image_embeddings fp32 [V, HW, 256] (channels last), llava_features [L, hidden] (one sequence) -> fused embeddings fp32
        [V, HW, 256] holding bf16 values (LLaVASAMFusion.forward).  As in the reference the key / value rows are dealt to the V
        views in consecutive runs of L / V (``view(batch_size, -1, heads, head_dim)`` with the QUERY's batch size,
        components.py:93-96): V must divide L."""
        V, HW, C = image_embeddings.shape
        L = llava_features.shape[0]
        if llava_features.shape[1] != self.llava_dim:
            raise ops.IvlmError(f"fusion head: llava_proj expects hidden size {self.llava_dim}, the language model has "
                                f"{llava_features.shape[1]} (the reference constructs LLaVASAMFusion() with its 13B default)")
        if L % V:
            raise ops.IvlmError(f"fusion head: {L} LLaVA positions cannot be dealt to {V} views (the reference's "
                                f"view({V}, -1, heads, head_dim) raises here too: components.py:94-95)")
        E, H = self.q[0].shape[0], self.num_heads
        hd = E // H
        xb = ops.gather_rows(image_embeddings.reshape(V * HW, C), out_kind="bf16")  # sam_embeddings.bfloat16()
        # sam_proj sees a strided view (permute + reshape, components.py:134-137): at::linear's matmul + add route - the product is
        # rounded to bf16 BEFORE the bias is added (oracle/nn.py _linear_bf16(fused=False))
        sp = ops.add_rows(ops.linear(xb, self.sam_proj[0], None), self.sam_proj[1].view(1, -1), out_kind="bf16")
        lp = ops.linear(ops.gather_rows(llava_features.contiguous(), out_kind="bf16"), *self.llava_proj)
        f32 = lambda t: ops.gather_rows(t, out_kind="f32")  # bf16 values in fp32 containers for the fp32 attention kernel
        q = f32(ops.linear(sp, *self.q)).view(V, HW, H, hd).permute(0, 2, 1, 3)
        k = f32(ops.linear(lp, *self.k)).view(V, L // V, H, hd).permute(0, 2, 1, 3)
        v = f32(ops.linear(lp, *self.v)).view(V, L // V, H, hd).permute(0, 2, 1, 3)
        o = ops.attention_f32(q, k, v, hd ** -0.5)  # [V,H,HW,hd] view of a [V,HW,H,hd] buffer
        ob = ops.gather_rows(o.permute(0, 2, 1, 3).reshape(V * HW, E), out_kind="bf16")
        y = ops.linear(ops.linear(ob, *self.o), *self.output_proj)
        return ops.gather_rows(ops.add_rows(xb, y, out_kind="bf16"), out_kind="f32").view(V, HW, C)
