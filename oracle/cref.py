"""ctypes binding of oracle/_build/liboracle.so (C restatement; TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lift_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def lift_mesh_soft(logits, vid, bary, num_vertices, clamp=20.0):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    vid = np.ascontiguousarray(vid, dtype=np.int32)
    bary = np.ascontiguousarray(bary, dtype=np.float32)
    B, V = logits.shape[:2]
    hw = int(np.prod(logits.shape[2:]))
    nv = int(num_vertices)
    pred = np.empty((B, nv), np.float32)
    nviews = np.empty((B, nv), np.float32)
    rc = lib().orc_lift_mesh_soft(_p(logits, C.c_float), _p(vid, C.c_int32), _p(bary, C.c_float), B, V,
                                  C.c_long(hw), nv, C.c_float(clamp), _p(pred, C.c_float), _p(nviews, C.c_float))
    assert rc == 0
    return pred, nviews


def lift_mesh_thresh(logits, vid, bary, num_vertices, threshold=0.3):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    vid = np.ascontiguousarray(vid, dtype=np.int32)
    bary = np.ascontiguousarray(bary, dtype=np.float32)
    V = logits.shape[0]
    hw = int(np.prod(logits.shape[1:]))
    nv = int(num_vertices)
    pred = np.empty((1, nv), np.float32)
    nviews = np.empty((1, nv), np.float32)
    rc = lib().orc_lift_mesh_thresh(_p(logits, C.c_float), _p(vid, C.c_int32), _p(bary, C.c_float), V,
                                    C.c_long(hw), nv, C.c_float(threshold), _p(pred, C.c_float),
                                    _p(nviews, C.c_float))
    assert rc == 0
    return pred, nviews


def lift_points(probs, pid, num_points):
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    pid = np.ascontiguousarray(pid, dtype=np.int32)
    B, V = probs.shape[:2]
    hw = int(np.prod(probs.shape[2:]))
    n = int(num_points)
    pred = np.empty((B, n), np.float32)
    nviews = np.empty((B, n), np.float32)
    rc = lib().orc_lift_points(_p(probs, C.c_float), _p(pid, C.c_int32), B, V, C.c_long(hw), n,
                               _p(pred, C.c_float), _p(nviews, C.c_float))
    assert rc == 0
    return pred, nviews


def postprocess_masks(low_res, input_size, original_size, img_size=1024):
    low = np.ascontiguousarray(low_res, dtype=np.float32)
    lead = low.shape[:-2]
    h, w = low.shape[-2:]
    n = int(np.prod(lead)) if lead else 1
    oh, ow = int(original_size[0]), int(original_size[1])
    out = np.empty(lead + (oh, ow), np.float32)
    rc = lib().orc_postprocess(_p(low, C.c_float), n, h, w, int(img_size), int(input_size[0]),
                               int(input_size[1]), oh, ow, _p(out, C.c_float))
    assert rc == 0
    return out
