"""Caller-side image preprocessing of the path, mirroring run_demo.py:65-79,330-369:

  * SAM inputs: ``ResizeLongestSide(1024).apply_image`` (segment_anything/utils/transforms.py:17-34,102-113: PIL
    bilinear resize of the uint8 image to longest side 1024) -> ``preprocess``: (x - mean) / std, zero-pad to 1024^2.
  * CLIP input: HF ``CLIPImageProcessor`` defaults of openai/clip-vit-large-patch14: resize shortest side to 224
    (PIL bicubic), centre crop 224, rescale 1/255, normalise.

The resampling itself stays with PIL on the host exactly as in the reference (same library => same pixels; it is an
identity for the benchmark's 1024^2 inputs); normalise + crop + pad + cast run in one HIP kernel on the device.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check
from .constants import CLIP_MEAN, CLIP_STD, SAM_MEAN_PIXEL, SAM_STD_PIXEL


def get_preprocess_shape(oldh, oldw, long_side_length):
    scale = long_side_length * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def _resize_u8(image: np.ndarray, size_hw, resample) -> np.ndarray:
    if tuple(image.shape[:2]) == tuple(size_hw):
        return image
    from PIL import Image

    return np.asarray(Image.fromarray(image).resize((size_hw[1], size_hw[0]), resample=resample))


def _normalize_pad(u8_dev, crop, mean255, std255, out_hw, dtype):
    lib = _lib.load()
    H, W, _ = u8_dev.shape
    y0, x0, ch, cw = crop
    out = torch.empty(3, out_hw[0], out_hw[1], dtype=dtype, device=u8_dev.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean255])
    s = (ctypes.c_float * 3)(*[float(v) for v in std255])
    check(lib.ivlm_normalize_pad_u8(u8_dev.data_ptr(), H, W, y0, x0, ch, cw, m, s, out.data_ptr(),
                                    1 if dtype == torch.bfloat16 else 0, out_hw[0], out_hw[1],
                                    torch.cuda.current_stream().cuda_stream), "normalize_pad_u8")
    return out


def sam_preprocess(image_u8: np.ndarray, device, img_size=1024, dtype=torch.bfloat16):
    """uint8 RGB HWC -> ([3,img,img] model input, resize (h,w)) — transform.apply_image + preprocess()."""
    from PIL import Image

    newh, neww = get_preprocess_shape(image_u8.shape[0], image_u8.shape[1], img_size)
    r = _resize_u8(image_u8, (newh, neww), Image.BILINEAR)
    t = torch.from_numpy(np.ascontiguousarray(r)).to(device)
    return _normalize_pad(t, (0, 0, newh, neww), SAM_MEAN_PIXEL, SAM_STD_PIXEL, (img_size, img_size), dtype), (newh, neww)


def clip_preprocess(image_u8: np.ndarray, device, size=224, dtype=torch.bfloat16):
    """uint8 RGB HWC -> [3,size,size] (CLIPImageProcessor: shortest-edge bicubic resize, centre crop, rescale, norm)."""
    from PIL import Image

    h, w = image_u8.shape[:2]
    short = min(h, w)
    nh, nw = (size, int(size * w / h)) if h == short else (int(size * h / w), size)
    r = _resize_u8(image_u8, (nh, nw), Image.BICUBIC)
    y0, x0 = (nh - size) // 2, (nw - size) // 2
    t = torch.from_numpy(np.ascontiguousarray(r)).to(device)
    mean255 = [255.0 * m for m in CLIP_MEAN]
    std255 = [255.0 * s for s in CLIP_STD]
    return _normalize_pad(t, (y0, x0, size, size), mean255, std255, (size, size), dtype)
