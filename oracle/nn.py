"""PyTorch-CPU fp32 restatement of the neural part of the contact-inference path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pure functions over a ``dict`` of weights
keyed by the reference's state-dict names.  Citations are into /root/reference.

  sam_image_encoder   model/segment_anything/modeling/image_encoder.py:110-426
  prompt_encoder_*    model/segment_anything/modeling/prompt_encoder.py:67-76,140-238
  mask_decoder        model/segment_anything/modeling/mask_decoder.py:75-191, transformer.py:16-242
  cam encoders        model/components.py:491-572,  process_embeddings  model/InteractVLM.py:268-294
  optional heads      model/components.py:40-153 (UncertaintyModule, LLaVASAMFusion), model/InteractVLM.py:20-44
  clip_vision         HF CLIPVisionModel as used by llava/model/multimodal_encoder/clip_encoder.py:31-60
  llama               HF LlamaModel as used by llava/model/language_model/llava_llama.py:55-135
  splice              llava/model/llava_arch.py:185-208 (mm_use_im_start_end branch)
  seg selection       model/InteractVLM.py:319-341,384-410 / 535-576
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def W(w, key):
    return w[key]


def linear(w, prefix, x):
    b = w.get(prefix + ".bias")
    return F.linear(x, w[prefix + ".weight"], b)


def layer_norm(w, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def layer_norm_2d(w, prefix, x, eps=1e-6):  # common.py:32-42, x [B,C,H,W]
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[prefix + ".weight"][:, None, None] * x + w[prefix + ".bias"][:, None, None]


# ------------------------------------------------------------------------------------------------
# SAM ViT image encoder
# ------------------------------------------------------------------------------------------------
def _get_rel_pos(q_size, k_size, rel_pos):  # image_encoder.py:292-322 (q_size == k_size here)
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    assert rel_pos.shape[0] == max_rel_dist
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.long()]


def _vit_attention(w, p, x, num_heads):  # image_encoder.py:235-260
    B, H, Wd, C = x.shape
    hd = C // num_heads
    qkv = linear(w, p + ".qkv", x).reshape(B, H * Wd, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * Wd, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = _get_rel_pos(H, H, w[p + ".rel_pos_h"])
    Rw = _get_rel_pos(Wd, Wd, w[p + ".rel_pos_w"])
    r_q = q.reshape(B * num_heads, H, Wd, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(-1, H, Wd, H, Wd) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(
        B * num_heads, H * Wd, H * Wd)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, num_heads, H, Wd, -1).permute(0, 2, 3, 1, 4).reshape(B, H, Wd, -1)
    return linear(w, p + ".proj", x)


def _window_partition(x, ws):  # image_encoder.py:263-288
    B, H, Wd, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - Wd % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, Wd + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def _window_unpartition(win, ws, pad_hw, hw):  # image_encoder.py:291-318
    Hp, Wp = pad_hw
    H, Wd = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :Wd, :].contiguous()


def sam_block(w, p, x, num_heads, window_size):  # image_encoder.py:177-193
    shortcut = x
    x = layer_norm(w, p + ".norm1", x, 1e-6)
    if window_size > 0:
        H, Wd = x.shape[1], x.shape[2]
        x, pad_hw = _window_partition(x, window_size)
    x = _vit_attention(w, p + ".attn", x, num_heads)
    if window_size > 0:
        x = _window_unpartition(x, window_size, pad_hw, (H, Wd))
    x = shortcut + x
    h = linear(w, p + ".mlp.lin2", F.gelu(linear(w, p + ".mlp.lin1", layer_norm(w, p + ".norm2", x, 1e-6))))
    return x + h


def sam_image_encoder(w, p, x, depth, num_heads, global_idx, window_size=14, patch=16):
    """x [B,3,S,S] -> [B,256,S/16,S/16]  (p = 'model.visual_model.image_encoder')."""
    x = F.conv2d(x, w[p + ".patch_embed.proj.weight"], w[p + ".patch_embed.proj.bias"], stride=patch)
    x = x.permute(0, 2, 3, 1) + w[p + ".pos_embed"]
    for i in range(depth):
        x = sam_block(w, f"{p}.blocks.{i}", x, num_heads, 0 if i in global_idx else window_size)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, w[p + ".neck.0.weight"])
    x = layer_norm_2d(w, p + ".neck.1", x)
    x = F.conv2d(x, w[p + ".neck.2.weight"], padding=1)
    return layer_norm_2d(w, p + ".neck.3", x)


# ------------------------------------------------------------------------------------------------
# SAM prompt encoder / mask decoder
# ------------------------------------------------------------------------------------------------
def dense_pe(w, p, size):  # prompt_encoder.py:67-76, 219-229
    h, wd = size
    g = w[p + ".pe_layer.positional_encoding_gaussian_matrix"]
    grid = torch.ones((h, wd), dtype=g.dtype)
    y = (grid.cumsum(0) - 0.5) / h
    x = (grid.cumsum(1) - 0.5) / wd
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ g)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1).unsqueeze(0)  # [1,C,h,w]


def prompt_encoder_text(w, p, text_embeds, size):  # prompt_encoder.py:140-186 with only text_embeds
    bs = text_embeds.shape[0]
    sparse = text_embeds
    dense = w[p + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, size[0], size[1])
    return sparse, dense


def _dec_attention(w, p, q, k, v, num_heads):  # transformer.py:220-242
    q, k, v = linear(w, p + ".q_proj", q), linear(w, p + ".k_proj", k), linear(w, p + ".v_proj", v)

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, num_heads, c // num_heads).transpose(1, 2)

    q, k, v = sep(q), sep(k), sep(v)
    attn = q @ k.permute(0, 1, 3, 2)
    attn = attn / math.sqrt(q.shape[-1])
    attn = torch.softmax(attn, dim=-1)
    out = attn @ v
    b, nh, nt, c = out.shape
    return linear(w, p + ".out_proj", out.transpose(1, 2).reshape(b, nt, nh * c))


def _two_way_block(w, p, queries, keys, query_pe, key_pe, skip_first_layer_pe, num_heads=8):  # transformer.py:151-182
    if skip_first_layer_pe:
        queries = _dec_attention(w, p + ".self_attn", queries, queries, queries, num_heads)
    else:
        q = queries + query_pe
        queries = queries + _dec_attention(w, p + ".self_attn", q, q, queries, num_heads)
    queries = layer_norm(w, p + ".norm1", queries, 1e-5)
    q = queries + query_pe
    k = keys + key_pe
    queries = queries + _dec_attention(w, p + ".cross_attn_token_to_image", q, k, keys, num_heads)
    queries = layer_norm(w, p + ".norm2", queries, 1e-5)
    mlp = linear(w, p + ".mlp.lin2", F.relu(linear(w, p + ".mlp.lin1", queries)))
    queries = layer_norm(w, p + ".norm3", queries + mlp, 1e-5)
    q = queries + query_pe
    k = keys + key_pe
    keys = keys + _dec_attention(w, p + ".cross_attn_image_to_token", k, q, queries, num_heads)
    keys = layer_norm(w, p + ".norm4", keys, 1e-5)
    return queries, keys


def _mlp3(w, p, x, n=3):  # mask_decoder.py:169-191
    for i in range(n):
        x = linear(w, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def mask_decoder(w, p, image_embeddings, image_pe, sparse, dense, num_mask_tokens=4, depth=2):
    """mask_decoder.py:116-164 + forward's mask slice (multimask_output=False).
    image_embeddings [V,C,h,w], image_pe [1,C,h,w], sparse [n,T,C], dense [n,C,h,w]
    -> low_res [V*n?,1,4h,4w], iou [.,1]  (batch broadcasting exactly as torch does it)."""
    out_tok = torch.cat([w[p + ".iou_token.weight"], w[p + ".mask_tokens.weight"]], dim=0)
    out_tok = out_tok.unsqueeze(0).expand(sparse.size(0), -1, -1)
    tokens = torch.cat((out_tok, sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0)
    src = src + dense
    pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, wd = src.shape
    # TwoWayTransformer.forward (transformer.py:62-106)
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos_src.flatten(2).permute(0, 2, 1)
    queries = tokens
    tp = p + ".transformer"
    for i in range(depth):
        queries, keys = _two_way_block(w, f"{tp}.layers.{i}", queries, keys, tokens, key_pe, i == 0)
    q = queries + tokens
    k = keys + key_pe
    queries = queries + _dec_attention(w, tp + ".final_attn_token_to_image", q, k, keys, 8)
    hs = layer_norm(w, tp + ".norm_final_attn", queries, 1e-5)
    src = keys
    iou_token_out = hs[:, 0, :]
    mask_tokens_out = hs[:, 1: 1 + num_mask_tokens, :]
    src = src.transpose(1, 2).view(b, c, h, wd)
    up = F.conv_transpose2d(src, w[p + ".output_upscaling.0.weight"], w[p + ".output_upscaling.0.bias"], stride=2)
    up = F.gelu(layer_norm_2d(w, p + ".output_upscaling.1", up))
    up = F.gelu(F.conv_transpose2d(up, w[p + ".output_upscaling.3.weight"], w[p + ".output_upscaling.3.bias"],
                                   stride=2))
    hyper = torch.stack([_mlp3(w, f"{p}.output_hypernetworks_mlps.{i}", mask_tokens_out[:, i, :])
                         for i in range(num_mask_tokens)], dim=1)
    b, c, h, wd = up.shape
    masks = (hyper @ up.view(b, c, h * wd)).view(b, num_mask_tokens, h, wd)
    iou = _mlp3(w, p + ".iou_prediction_head", iou_token_out)
    return masks[:, 0:1], iou[:, 0:1]


def postprocess_masks(masks, input_size, original_size, img_size=1024):  # sam.py:137-172
    m = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, tuple(original_size), mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------------
# camera-pose conditioning (components.py:491-572, InteractVLM.py:268-294)
# ------------------------------------------------------------------------------------------------
def cam_encode(w, p, kind, cam_params, V):
    """cam_params [V,5] -> view encodings [1,V,256] (kind in simple|view_index|vi_v1)."""
    if kind == "simple":
        return F.relu(linear(w, p + ".linear1", cam_params)).unsqueeze(0)
    outs = []
    for v in range(V):
        c = cam_params[[v]]
        if kind == "view_index":
            base = torch.sigmoid(linear(w, p + ".spatial_encoder.2", F.relu(linear(w, p + ".spatial_encoder.0", c))))
            outs.append(linear(w, f"{p}.view_transforms.{v}", base))
        else:  # vi_v1
            base = F.relu(linear(w, p + ".spatial_encoder.2", F.relu(linear(w, p + ".spatial_encoder.0", c))))
            outs.append(torch.sigmoid(linear(w, f"{p}.view_transforms.{v}", base)))
    return torch.stack(outs, dim=1)


def attention_splitter(w, p, x):  # components.py:155-193
    xp = linear(w, p + ".input_proj", x)
    k, v = linear(w, p + ".key", xp), linear(w, p + ".value", xp)
    outs = []
    for name in ("query_human", "query_object"):
        q = linear(w, p + "." + name, xp)
        a = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / (k.size(-1) ** 0.5), dim=-1)
        outs.append(linear(w, p + ".output_proj", torch.matmul(a, v)))
    return outs[0], outs[1]


# ------------------------------------------------------------------------------------------------
# optional heads of ModifiedSAM (InteractVLM.py:20-38,41-44; off in every released configuration).  Both modules cast their
# inputs to bf16 and therefore only run inside the bf16 model: the restatement keeps fp32 tensors holding bf16 VALUES and rounds
# where the bf16 modules round (after every linear / matmul / softmax / activation / add).
# ------------------------------------------------------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16).float()


def _linear_bf16(w, prefix, x, fused=True):
    """nn.Linear on bf16 tensors: fp32 accumulation, bias added before the one rounding (ATen's addmm route).  fused=False: the
    matmul result is rounded, then the bias is added and the sum rounded again - the route at::linear takes for a 3-D input
    that is not contiguous (matmul + add)."""
    if not fused:
        return _bf(_bf(F.linear(x, _bf(w[prefix + ".weight"]))) + _bf(w[prefix + ".bias"]))
    return _bf(F.linear(x, _bf(w[prefix + ".weight"]), _bf(w[prefix + ".bias"])))


def uncertainty_head(w, p, emb):  # components.py:40-78; emb [B,256,H,W] -> [B,1,H,W] (bf16 values)
    B, C, H, Wd = emb.shape
    x = _bf(emb).permute(0, 2, 3, 1).reshape(B * H * Wd, C)
    x = F.relu(_linear_bf16(w, p + ".linear1", x))
    x = F.relu(_linear_bf16(w, p + ".linear2", x))
    x = _bf(F.softplus(_linear_bf16(w, p + ".linear3", x)))
    return x.reshape(B, H, Wd, 1).permute(0, 3, 1, 2)


def uncertainty_resize(m, size, lambda_bf16=False):
    """InteractVLM.py:446-447 / 615-616: F.interpolate(bilinear, align_corners=False) on the bf16 map.  The four-tap sum runs in
    fp32 and is rounded to bf16 once; ATen's CPU kernel additionally rounds the interpolation weights to the tensor's dtype
    (lambda_bf16=True reproduces tests/golden/optional_heads.npz, made on the CPU, bit for bit), its GPU kernel keeps them in
    fp32 (the default here - the reference's deployed bf16 model runs on the GPU)."""
    h, wd = m.shape[-2:]

    def taps(o, i):
        x = ((torch.arange(o, dtype=torch.float32) + 0.5) * (i / o) - 0.5).clamp(min=0)
        i0 = x.floor().long().clamp(max=i - 1)
        i1 = (i0 + 1).clamp(max=i - 1)
        l1 = x - i0.float()
        if lambda_bf16:
            l1 = _bf(l1)
            return i0, i1, _bf(1 - l1), l1
        return i0, i1, 1 - l1, l1

    y0, y1, wy0, wy1 = taps(size[0], h)
    x0, x1, wx0, wx1 = taps(size[1], wd)
    mf = m.float()
    top = mf[..., y0, :][..., :, x0] * wx0 + mf[..., y0, :][..., :, x1] * wx1
    bot = mf[..., y1, :][..., :, x0] * wx0 + mf[..., y1, :][..., :, x1] * wx1
    return _bf(top * wy0[:, None] + bot * wy1[:, None])


def sam_fusion(w, p, sam, llava, num_heads=8):  # components.py:79-153; sam [B,256,H,W], llava [1,L,hidden] -> [B,256,H,W]
    B, C, H, Wd = sam.shape
    sam = _bf(sam)
    x = sam.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
    # (components.py:134-137: permute + reshape of [B,C,H,W] is a strided VIEW [B, H*W, C], not a copy -> the unfused route)
    sp = _linear_bf16(w, p + ".sam_proj", x, fused=False)
    lp = _linear_bf16(w, p + ".llava_proj", _bf(llava))
    E = sp.shape[-1]
    hd = E // num_heads
    # components.py:93-96: view(batch_size = the QUERY's batch, -1, heads, hd) - with one LLaVA sequence and B views the key /
    # value positions are dealt to the views in consecutive runs of L / B (an error in the reference when B does not divide L)
    q = _linear_bf16(w, p + ".fusion.q_proj", sp).view(B, -1, num_heads, hd).transpose(1, 2)
    k = _linear_bf16(w, p + ".fusion.k_proj", lp).view(B, -1, num_heads, hd).transpose(1, 2)
    v = _linear_bf16(w, p + ".fusion.v_proj", lp).view(B, -1, num_heads, hd).transpose(1, 2)
    a = _bf(_bf(torch.matmul(q, k.transpose(-2, -1))) / (hd ** 0.5))
    a = _bf(F.softmax(a, dim=-1))
    o = _bf(torch.matmul(a, v)).transpose(1, 2).contiguous().view(B, -1, E)
    o = _linear_bf16(w, p + ".output_proj", _linear_bf16(w, p + ".fusion.out_proj", o))
    return _bf(sam + o.reshape(B, H, Wd, C).permute(0, 3, 1, 2))


def process_embeddings(w, embedding, cam_params, token, cfg):
    """embedding [n_seg,V,256] (already repeated over views); cfg: multiview_cam_cond, cam_encoder_type,
    multiview_channels, base_token_type, hseg_token_idx, oseg_token_idx."""
    if cfg["multiview_cam_cond"]:
        enc = cam_encode(w, "cam_pose_encoder", cfg["cam_encoder_type"], cam_params, cfg["multiview_channels"])
        embedding = embedding + enc if cfg["cam_encoder_type"] == "simple" else embedding * enc
    if cfg["base_token_type"] == "Gen":
        return embedding
    if token == cfg["hseg_token_idx"]:
        return attention_splitter(w, "attention_splitter", embedding)[0]
    if token == cfg["oseg_token_idx"]:
        return attention_splitter(w, "attention_splitter", embedding)[1]
    return embedding


# ------------------------------------------------------------------------------------------------
# CLIP vision tower (HF CLIPVisionModel semantics), select_layer = -2, drop CLS
# ------------------------------------------------------------------------------------------------
def clip_vision(w, p, x, num_layers, num_heads, select_layer=-2, patch=14, eps=1e-5):
    """x [B,3,S,S] -> [B,(S/patch)^2,hidden]. p = 'model.vision_tower.vision_tower.vision_model'."""
    e = p + ".embeddings"
    pe = F.conv2d(x, w[e + ".patch_embedding.weight"], None, stride=patch).flatten(2).transpose(1, 2)
    cls = w[e + ".class_embedding"].expand(x.shape[0], 1, -1)
    h = torch.cat([cls, pe], dim=1) + w[e + ".position_embedding.weight"][None]
    h = layer_norm(w, p + ".pre_layrnorm", h, eps)
    hidden = [h]
    for i in range(num_layers):
        lp = f"{p}.encoder.layers.{i}"
        r = h
        y = layer_norm(w, lp + ".layer_norm1", h, eps)
        B, T, C = y.shape
        hd = C // num_heads
        q = (linear(w, lp + ".self_attn.q_proj", y) * hd ** -0.5).view(B, T, num_heads, hd).transpose(1, 2)
        k = linear(w, lp + ".self_attn.k_proj", y).view(B, T, num_heads, hd).transpose(1, 2)
        v = linear(w, lp + ".self_attn.v_proj", y).view(B, T, num_heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        h = r + linear(w, lp + ".self_attn.out_proj", a.transpose(1, 2).reshape(B, T, C))
        y = layer_norm(w, lp + ".layer_norm2", h, eps)
        y = linear(w, lp + ".mlp.fc1", y)
        y = y * torch.sigmoid(1.702 * y)  # quick_gelu
        h = h + linear(w, lp + ".mlp.fc2", y)
        hidden.append(h)
    return hidden[select_layer][:, 1:]


# ------------------------------------------------------------------------------------------------
# LLaMA (HF LlamaModel semantics: RMSNorm, rotate-half RoPE, SwiGLU, causal fp32 softmax)
# ------------------------------------------------------------------------------------------------
def rms_norm(wt, x, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return wt * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def rope_tables(T, hd, theta=10000.0, pos0=0):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    t = torch.arange(pos0, pos0 + T, dtype=torch.float32)
    fr = torch.outer(t, inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama(w, p, x, num_layers, num_heads, eps=1e-5, theta=10000.0):
    """x = inputs_embeds [B,T,H] -> last hidden state after the final norm [B,T,H]. p = 'model'."""
    B, T, C = x.shape
    hd = C // num_heads
    cos, sin = rope_tables(T, hd, theta)
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(num_layers):
        lp = f"{p}.layers.{i}"
        r = x
        y = rms_norm(w[lp + ".input_layernorm.weight"], x, eps)
        q = linear(w, lp + ".self_attn.q_proj", y).view(B, T, num_heads, hd).transpose(1, 2)
        k = linear(w, lp + ".self_attn.k_proj", y).view(B, T, num_heads, hd).transpose(1, 2)
        v = linear(w, lp + ".self_attn.v_proj", y).view(B, T, num_heads, hd).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        a = (q @ k.transpose(2, 3)) / math.sqrt(hd) + mask
        a = torch.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype) @ v
        x = r + linear(w, lp + ".self_attn.o_proj", a.transpose(1, 2).reshape(B, T, C))
        y = rms_norm(w[lp + ".post_attention_layernorm.weight"], x, eps)
        y = linear(w, lp + ".mlp.down_proj", F.silu(linear(w, lp + ".mlp.gate_proj", y)) * linear(w, lp + ".mlp.up_proj", y))
        x = x + y
    return rms_norm(w[p + ".norm.weight"], x, eps)


def splice_image_features(w, input_ids, image_features, image_token_index=-200):
    """llava_arch.py:185-208 (mm_use_im_start_end): ids [L] with one -200 flanked by <im_start>/<im_end>
    -> embeds [L-1+N_img, H].  image_features [N_img, H]."""
    emb = w["model.embed_tokens.weight"]
    pos = int((input_ids == image_token_index).nonzero()[0])
    return torch.cat([emb[input_ids[:pos]], image_features, emb[input_ids[pos + 1: pos + 2]],
                      emb[input_ids[pos + 2:]]], dim=0)


def seg_rows(ids, seg_ids, img_emb_len, model_forward=True):
    """Row indices (into the L+img_emb_len hidden sequence) selected as [SEG] embeddings.
    InteractVLM.py:331-341 (model_forward) / 545-549 (evaluate): for a seg token at id-index k the
    row is k - 1 + img_emb_len."""
    m = torch.zeros_like(ids, dtype=torch.bool)
    for s in seg_ids:
        if s is not None:
            m |= ids == s
    m = m[1:]
    if model_forward:
        m = torch.cat([m, torch.zeros(1, dtype=torch.bool)])
    m = torch.cat([torch.zeros(img_emb_len, dtype=torch.bool), m])
    return m


def text_hidden_fcs(w, x):  # InteractVLM.py:100-112
    return linear(w, "model.text_hidden_fcs.0.2", F.relu(linear(w, "model.text_hidden_fcs.0.0", x)))
