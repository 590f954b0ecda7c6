"""Thin torch-tensor wrappers over the C ABI (``include/ivlm_hip.h``).

torch is plumbing here: device memory, the current HIP stream and nothing else.  Every function
requires CUDA(HIP) tensors and raises if the extension is missing — there is no CPU path.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import IvlmError, check

IVLM_F32, IVLM_BF16 = 0, 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Per-launch kernel durations for bench.py's roofline legs.  The HIP events are ATTACHED to the kernels
    (``ivlm_profile_launches`` -> hipExtLaunchKernelGGL: the command processor records the start event when the first kernel of
    the call begins and the stop event when its last kernel ends, on the stream the kernels run on) - an event pair recorded
    around a single ~17 us launch reads 5-12 us too long, which is why round 1's brackets sat below the rocprofv3 durations.
    Off by default: zero overhead."""

    def __init__(self):
        self.enabled = False
        self.records = {}  # name -> list[(start_event, end_event, work, tag)]
        self._hip = None

    def _rt(self):
        if self._hip is None:
            import ctypes
            import os
            self._hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
            self._hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        return self._hip

    def _event(self):
        import ctypes
        ev = ctypes.c_void_p()
        rc = self._rt().hipEventCreate(ctypes.byref(ev))
        if rc != 0:
            raise IvlmError(f"hipEventCreate failed ({rc})")
        return ev

    def _elapsed_ms(self, a, b):
        import ctypes
        ms = ctypes.c_float()
        rc = self._rt().hipEventElapsedTime(ctypes.byref(ms), a, b)
        if rc != 0:
            raise IvlmError(f"hipEventElapsedTime failed ({rc})")
        return ms.value

    def start(self):
        self._free()
        self.enabled, self.records = True, {}

    def stop(self):
        self.enabled = False

    def _free(self):
        if self._hip is not None:
            for rec in self.records.values():
                for a, b, _, _ in rec:
                    self._hip.hipEventDestroy(a)
                    self._hip.hipEventDestroy(b)
        self.records = {}

    def time(self, name, work, fn, tag=None):
        if not self.enabled:
            return fn()
        lib = _lib.load()
        a, b = self._event(), self._event()
        lib.ivlm_profile_launches(a, b)
        try:
            r = fn()
        finally:
            n = lib.ivlm_profile_launches(None, None)
        if n <= 0:
            raise IvlmError(f"{name}: no instrumented kernel launch inside the timed call")
        self.records.setdefault(name, []).append((a, b, work, tag))
        return r

    def by_tag(self, name):
        """per-tag (e.g. GEMM shape) totals of one record family: tag -> {launches, total_s, work}."""
        torch.cuda.synchronize()
        out = {}
        for a, b, w, tag in self.records.get(name, []):
            d = out.setdefault(tag, {"launches": 0, "total_s": 0.0, "work": 0.0})
            d["launches"] += 1
            d["total_s"] += self._elapsed_ms(a, b) * 1e-3
            d["work"] += w
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, rec in self.records.items():
            ms = [self._elapsed_ms(a, b) for a, b, _, _ in rec]
            out[name] = {"launches": len(rec), "total_s": sum(ms) * 1e-3, "avg_us": sum(ms) / len(ms) * 1e3,
                         "work": float(sum(r[2] for r in rec))}
        return out


TIMER = KernelTimer()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise IvlmError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise IvlmError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise IvlmError(f"{name}: tensor must be contiguous")
    return t


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return IVLM_F32
    if t.dtype == torch.bfloat16:
        return IVLM_BF16
    raise IvlmError(f"unsupported dtype {t.dtype}")


def _p(t):
    return 0 if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------
# lift
# --------------------------------------------------------------------------------------------
class LiftPlan:
    """Vertex-major CSR of constant pixel->vertex tables (built once on the GPU)."""

    def __init__(self, vid: torch.Tensor, bary: torch.Tensor, num_vertices: int):
        lib = _lib.load()
        vid = _req(vid, torch.int32, "vid")
        bary = _req(bary, torch.float32, "bary")
        assert vid.dim() == 4 and vid.shape[-1] == 3 and vid.shape == bary.shape, "tables must be [V,H,W,3]"
        V, H, W, _ = vid.shape
        self.V, self.HW, self.hw_shape, self.num_vertices = V, H * W, (H, W), int(num_vertices)
        dev = vid.device
        cap = 3 * V * H * W
        self.row_ptr = torch.empty(V * self.num_vertices + 1, dtype=torch.int32, device=dev)
        ent_pix = torch.empty(cap, dtype=torch.int32, device=dev)
        ent_w = torch.empty(cap, dtype=torch.float32, device=dev)
        nnz = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.ivlm_lift_plan_workspace_bytes(V, self.HW, self.num_vertices)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.ivlm_lift_plan_build(vid.data_ptr(), bary.data_ptr(), V, self.HW, self.num_vertices,
                                       self.row_ptr.data_ptr(), ent_pix.data_ptr(), ent_w.data_ptr(), cap,
                                       nnz.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "lift_plan_build")
        self.nnz = int(nnz.item())
        # trim to the real size (one-time copy) so the plan holds 8 B per entry, no slack
        keep = max(self.nnz, 1)  # an empty plan still needs valid (never dereferenced) pointers
        self.ent_pix = ent_pix[:keep].clone()
        self.ent_w = ent_w[:keep].clone()

    def bytes(self) -> int:
        return self.nnz * 8 + self.row_ptr.numel() * 4

    @classmethod
    def from_points(cls, pid: torch.Tensor, num_points: int):
        """Point-major CSR of a pixel -> point map (pid i32 [V,H,W], -1 = no point): weights 1, evaluated with mode 2."""
        lib = _lib.load()
        pid = _req(pid, torch.int32, "pid")
        assert pid.dim() == 3, "pid must be [V,H,W]"
        self = cls.__new__(cls)
        V, H, W = pid.shape
        self.V, self.HW, self.hw_shape, self.num_vertices = V, H * W, (H, W), int(num_points)
        dev = pid.device
        cap = V * H * W
        self.row_ptr = torch.empty(V * self.num_vertices + 1, dtype=torch.int32, device=dev)
        ent_pix = torch.empty(cap, dtype=torch.int32, device=dev)
        ent_w = torch.empty(cap, dtype=torch.float32, device=dev)
        nnz = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.ivlm_lift_plan_workspace_bytes(V, self.HW, self.num_vertices)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.ivlm_lift_points_plan_build(pid.data_ptr(), V, self.HW, self.num_vertices, self.row_ptr.data_ptr(),
                                              ent_pix.data_ptr(), ent_w.data_ptr(), cap, nnz.data_ptr(), ws.data_ptr(), ws_bytes,
                                              _stream()), "lift_points_plan_build")
        self.nnz = int(nnz.item())
        keep = max(self.nnz, 1)
        self.ent_pix = ent_pix[:keep].clone()
        self.ent_w = ent_w[:keep].clone()
        return self


def lift_mesh_plan(logits: torch.Tensor, plan: LiftPlan, mode: int = 0, param: float = 20.0, want_nviews=False):
    """logits f32 [B,V,H,W] -> contacts f32 [B,Nv] (and nviews) via the CSR plan."""
    lib = _lib.load()
    logits = _req(logits, torch.float32, "logits")
    B, V = logits.shape[0], logits.shape[1]
    assert V == plan.V and logits[0, 0].numel() == plan.HW, "logit shape does not match the lift plan"
    out = torch.empty(B, plan.num_vertices, dtype=torch.float32, device=logits.device)
    nviews = torch.empty_like(out) if want_nviews else None
    call = lambda: check(lib.ivlm_lift_mesh_plan(
        logits.data_ptr(), plan.row_ptr.data_ptr(), plan.ent_pix.data_ptr(), plan.ent_w.data_ptr(), B, V, plan.HW,
        plan.num_vertices, mode, float(param), out.data_ptr(), _p(nviews), _stream()), "lift_mesh_plan")
    if TIMER.enabled:  # algorithmic bytes of the dense formulation: 28 B/pixel + 8 B/vertex
        TIMER.time("lift_mesh_plan", float(B) * (V * plan.HW * 28 + 2 * plan.num_vertices * 4), call)
    else:
        call()
    return (out, nviews) if want_nviews else out


def lift_mesh_plan_lowres(low, plan: LiftPlan, input_size, original_size, img_size=1024, mode=0, param=20.0,
                          want_nviews=False):
    """low f32|bf16 [B,V,lh,lw] -> contacts [B,Nv]: == lift_mesh_plan(postprocess_masks(low)), fused."""
    lib = _lib.load()
    low = _req(low, None, "low")
    B, V, lh, lw = low.shape
    oh, ow = int(original_size[0]), int(original_size[1])
    assert V == plan.V and oh * ow == plan.HW, "plan was built for a different mask size"
    out = torch.empty(B, plan.num_vertices, dtype=torch.float32, device=low.device)
    nviews = torch.empty_like(out) if want_nviews else None
    call = lambda: check(lib.ivlm_lift_mesh_plan_lowres(
        low.data_ptr(), _dt(low), lh, lw, int(img_size), int(input_size[0]), int(input_size[1]), oh, ow,
        plan.row_ptr.data_ptr(), plan.ent_pix.data_ptr(), plan.ent_w.data_ptr(), B, V, plan.num_vertices, mode,
        float(param), out.data_ptr(), _p(nviews), _stream()), "lift_mesh_plan_lowres")
    if TIMER.enabled:
        TIMER.time("lift_mesh_plan", float(B) * (V * plan.HW * 28 + 2 * plan.num_vertices * 4), call)
    else:
        call()
    return (out, nviews) if want_nviews else out


_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def lift_mesh_dense(logits, vid, bary, num_vertices: int, mode: int = 0, param: float = 20.0, want_nviews=False):
    """Streaming (atomic) variant over dense tables: vid i32 [V,H,W,3], bary f32 [V,H,W,3]."""
    lib = _lib.load()
    logits = _req(logits, torch.float32, "logits")
    vid = _req(vid, torch.int32, "vid")
    bary = _req(bary, torch.float32, "bary")
    B, V = logits.shape[0], logits.shape[1]
    HW = logits[0, 0].numel()
    nv = int(num_vertices)
    out = torch.empty(B, nv, dtype=torch.float32, device=logits.device)
    nviews = torch.empty_like(out) if want_nviews else None
    nbytes = lib.ivlm_lift_mesh_dense_workspace_bytes(B, V, nv)
    ws = _workspace(nbytes, logits.device)
    check(lib.ivlm_lift_mesh_dense(logits.data_ptr(), vid.data_ptr(), bary.data_ptr(), B, V, HW, nv, mode,
                                   float(param), out.data_ptr(), _p(nviews), ws.data_ptr(), ws.numel(), _stream()),
          "lift_mesh_dense")
    return (out, nviews) if want_nviews else out


def lift_points(probs, pid, num_points: int, want_nviews=False):
    """probs f32 [B,V,H,W]; pid i32 [B,V,H,W] or [V,H,W] (shared) -> f32 [B,Np]."""
    lib = _lib.load()
    probs = _req(probs, torch.float32, "probs")
    pid = _req(pid, torch.int32, "pid")
    B, V = probs.shape[0], probs.shape[1]
    HW = probs[0, 0].numel()
    batched = 1 if pid.dim() == probs.dim() else 0
    n = int(num_points)
    out = torch.empty(B, n, dtype=torch.float32, device=probs.device)
    nviews = torch.empty_like(out) if want_nviews else None
    nbytes = lib.ivlm_lift_points_workspace_bytes(B, V, n)
    ws = _workspace(nbytes, probs.device)
    check(lib.ivlm_lift_points(probs.data_ptr(), pid.data_ptr(), batched, B, V, HW, n, out.data_ptr(), _p(nviews),
                               ws.data_ptr(), ws.numel(), _stream()), "lift_points")
    return (out, nviews) if want_nviews else out


def lift_points_plan(probs, plan: LiftPlan, want_nviews=False):
    """probs f32 [B,V,H,W] -> f32 [B,Np] through a point-major plan (``LiftPlan.from_points``): the deterministic, atomic-free
    counterpart of ``lift_points`` for a pixel -> point map that is used more than once."""
    return lift_mesh_plan(probs, plan, mode=2, param=0.0, want_nviews=want_nviews)


def postprocess_masks(low_res, input_size, original_size, img_size: int = 1024, apply_sigmoid: bool = False, sigmoid_gt=None,
                      ignore_label: float = -1.0):
    """low_res f32|bf16 [...,h,w] -> f32 [...,oh,ow] (Sam.postprocess_masks).  sigmoid_gt (f32, the output's shape): sigmoid on the
    pixels where it differs from ignore_label (InteractVLM.py:452-456), raw logits elsewhere."""
    lib = _lib.load()
    low_res = _req(low_res, None, "low_res")
    lead = tuple(low_res.shape[:-2])
    h, w = low_res.shape[-2:]
    n = 1
    for s in lead:
        n *= s
    oh, ow = int(original_size[0]), int(original_size[1])
    out = torch.empty(lead + (oh, ow), dtype=torch.float32, device=low_res.device)
    if sigmoid_gt is not None:
        gt = _req(sigmoid_gt, torch.float32, "sigmoid_gt")
        assert gt.numel() == out.numel()
        check(lib.ivlm_postprocess_masks_valid(low_res.data_ptr(), _dt(low_res), n, h, w, int(img_size), int(input_size[0]),
                                               int(input_size[1]), oh, ow, gt.data_ptr(), float(ignore_label), out.data_ptr(),
                                               _stream()), "postprocess_masks_valid")
        return out
    check(lib.ivlm_postprocess_masks(low_res.data_ptr(), _dt(low_res), n, h, w, int(img_size), int(input_size[0]),
                                     int(input_size[1]), oh, ow, 1 if apply_sigmoid else 0, out.data_ptr(),
                                     _stream()), "postprocess_masks")
    return out


# --------------------------------------------------------------------------------------------
# dense building blocks
# --------------------------------------------------------------------------------------------
ACT = {"none": 0, "gelu": 1, "quick_gelu": 2, "relu": 3, "silu": 4, "swiglu": 5, "sigmoid": 6}
BF16 = torch.bfloat16


SPLITK = os.environ.get("IVLM_SPLITK", "1") != "0"  # small-M GEMMs (prefill, CLIP): cut K so that >= ~3 tiles per CU are in flight
# fused reduction (the tile's last block sums the slices inside the GEMM launch: ivlm_gemm_bf16_splitk_fused) - bit-identical, one launch
# instead of two, and 2 - 3 x SLOWER on MI355X (round 5: prefill 12.0 -> 26.3 ms; down_proj 64 -> 192 us): the agent-scope release /
# acquire every block needs (L2 write-back + invalidate: the slices of a tile run on different XCDs) costs far more than the reduction
# launch it saves.  Opt-in, kept for the record and for single-XCD parts.
SPLITK_FUSED = os.environ.get("IVLM_SPLITK_FUSED", "0") == "1"
_SPLITK_CNT = {}
_SPLITK_CNT_CAPTURE = {}


def _splitk_counters(device):
    """IVLM_SPLITK_COUNTERS zeroed int32 words: the arrival counters of the fused split-K GEMMs (every launch leaves them at zero).
    Launches that may run concurrently must not share an array, so there is one per (device, stream).  While a HIP graph is being
    captured the array is created INSIDE the capture (memory of the graph's own pool, one array per capture, its zero fill is
    a node of the graph): two graphs - which may be replayed on different streams at the same time - never share counters, and
    the launches inside one captured stream are ordered."""
    if torch.cuda.is_current_stream_capturing():
        key = (device.index, _stream())
        c = _SPLITK_CNT_CAPTURE.get(key)
        if c is None:
            c = _SPLITK_CNT_CAPTURE[key] = torch.zeros(4096, dtype=torch.int32, device=device)
        return c
    if _SPLITK_CNT_CAPTURE:
        _SPLITK_CNT_CAPTURE.clear()  # (no capture in progress: the next capture gets arrays of its own)
    key = (device.index, _stream())
    c = _SPLITK_CNT.get(key)
    if c is None:
        c = _SPLITK_CNT[key] = torch.zeros(4096, dtype=torch.int32, device=device)
    return c


def _splitk_choice(M, N, K, act, rms):
    """number of K slices (1 = plain kernel): only where the 128x64 tiling leaves most of the 256 CUs idle."""
    if not SPLITK or M <= 8 or M > 1024 or act == "swiglu" or rms is not None or N % 4 or K % 64:
        return 1
    if M <= 16 and N >= 1024 and K >= 1024:
        return 1  # the skinny split-K MFMA kernel (csrc/gemv_mfma.hip) takes these
    if 128 < M <= 352 and N >= 8192:  # the row-stationary 176 x 128 tiles (gemm.hip): two K slices when they leave CUs idle
        t176 = ((M + 175) // 176) * ((N + 127) // 128)
        return 2 if (t176 < 256 and (K // 64) % 2 == 0 and K // 2 >= 512) else 1
    tiles = ((M + 127) // 128) * ((N + 63) // 64)
    if tiles >= 256:
        return 1
    best, k64 = 1, K // 64
    for sp in range(2, min(8, 1024 // tiles) + 1):
        if k64 % sp == 0 and K // sp >= 512:
            best = sp
    return best


GEMM_A_F32, GEMM_RES_F32, GEMM_A_SPLIT, GEMM_OUT_SPLIT, GEMM_F16, GEMM_OUT_F16, GEMM_W_PANEL = 1, 2, 4, 8, 16, 32, 64


def panel_weight(w):
    """[N, K] (K % 64 == 0) -> the K-panel layout [K / 64, N, 64] of the same values (IVLM_GEMM_W_PANEL): what ``linear`` takes as a
    3-D weight.  One wave DMA instruction of a K tile then reads 1 KB contiguous instead of 8 lines a row stride apart."""
    N, K = w.shape
    assert K % 64 == 0
    return w.view(N, K // 64, 64).permute(1, 0, 2).contiguous()
F16 = torch.float16
F32 = torch.float32


def linear(x, weight, bias=None, act="none", residual=None, res_mod=0, out=None, out_f32=False, rms=None, out_rows=None,
           a_rows=None, a_split=False, out_split=False, out_f16=False):
    """act(x @ weight.T + bias) + residual.  out_rows (int32 [M], tile GEMM path): scatter epilogue, see ivlm_hip.h.  x [..., K] bf16 - or fp32 for M <= 16 rows (weight-streaming kernels: exact
    products) - last dim contiguous, uniform row stride; weight [N, K] bf16; residual bf16 or fp32 (fp32 residual stream).
    "Parity" precision (tile GEMM, M > 16): a_split - x is [..., 2K] = [hi | lo] bf16 rows (fp32 activations, see split_rows)
    against the plain [N, K] weight; out_split - the fp32 result is written as [hi | lo] bf16 rows [..., 2 n_out]."""
    lib = _lib.load()
    w_panel = weight.dim() == 3  # K-panel layout [K / 64, N, 64] (``panel_weight``): tile GEMM only, same results as [N, K]
    if w_panel:
        assert weight.shape[2] == 64 and weight.is_contiguous()
        N, K = weight.shape[1], weight.shape[0] * 64
    else:
        N = weight.shape[0]
        K = weight.shape[1]
    f16 = weight.dtype == F16  # IEEE-half operands (tile GEMM): x must be fp16 too
    assert x.shape[-1] == (2 * K if a_split else K) and x.dtype in (BF16, F32, F16) and weight.dtype in (BF16, F16)
    assert (x.dtype == F16) == f16  # (f16 with a_split: [hi | lo] IEEE halves, the IVLM_F16_SPLIT rows of layernorm(out_split, out_f16))
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0] if a_rows is None else a_rows.numel()  # a_rows (int32 [M]): product row m reads x2[a_rows[m]]
    flags = 0
    if x.dtype == F32:
        if M > 16:
            raise IvlmError("linear: fp32 activations only on the M <= 16 weight-streaming paths (use split_rows + a_split)")
        flags |= GEMM_A_F32
    if a_split:
        assert x.dtype in (BF16, F16)
        flags |= GEMM_A_SPLIT
    if f16:
        flags |= GEMM_F16
    if out is not None and out.dtype == F16:
        out_f16 = True
    if out_f16:  # (with out_split: the fp32 result as [hi | lo] IEEE halves)
        assert not out_f32 or out_split
        flags |= GEMM_OUT_F16
    if w_panel:
        if M <= 16 or out_rows is not None or a_split or out_split:
            raise IvlmError("linear: K-panel weights serve the plain tile GEMMs (M > 16) only")
        flags |= GEMM_W_PANEL
    n_out = N // 2 if act == "swiglu" else N
    if out_split:
        flags |= GEMM_OUT_SPLIT
    n_cols = 2 * n_out if out_split else n_out
    if out is None:
        lead = x.shape[:-1] if a_rows is None else (M,)
        out = torch.empty(tuple(lead) + (n_cols,), dtype=F32 if (out_f32 and not out_split) else (F16 if out_f16 else BF16),
                          device=x.device)
    assert not out_split or out.dtype == (F16 if out_f16 else BF16)
    o2 = out.reshape(-1, n_cols)
    assert o2.stride(-1) == 1 and weight.stride(-1) == 1
    ldw = 64 if w_panel else weight.stride(0)
    r2, ldr = None, 0
    if residual is not None:
        r2 = residual.reshape(-1, N)
        assert r2.dtype in (BF16, F32) and r2.stride(-1) == 1
        ldr = r2.stride(0)
        if r2.dtype == F32:
            flags |= GEMM_RES_F32
    if bias is not None:
        assert bias.dtype == BF16 and bias.is_contiguous()
    call = lambda: check(lib.ivlm_gemm_bf16(
        x2.data_ptr(), x2.stride(0), weight.data_ptr(), ldw, o2.data_ptr(), o2.stride(0), _p(bias),
        _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out.dtype == F32 else 0, 1, 0, 0, 0, 0,
        _p(rms[0]) if rms else 0, float(rms[1]) if rms else 0.0, flags, _p(out_rows), _p(a_rows), _stream()), "gemm_bf16")
    splits = _splitk_choice(M, N, K, act, rms) if (x.dtype in (BF16, F16) and out_rows is None and a_rows is None) else 1
    if splits > 1 and o2.stride(0) % 4 == 0:
        ws = torch.empty(splits * M * N, dtype=F32, device=x.device)  # caching allocator: stream-safe
        if SPLITK_FUSED:  # reduction inside the GEMM launch (the tile's last block sums its slices): same values, one launch
            cnt = _splitk_counters(x.device)
            call = lambda: check(lib.ivlm_gemm_bf16_splitk_fused(
                x2.data_ptr(), x2.stride(0), weight.data_ptr(), ldw, o2.data_ptr(), o2.stride(0), _p(bias),
                _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out.dtype == F32 else 0, splits, ws.data_ptr(),
                ws.numel() * 4, cnt.data_ptr(), flags, _stream()), "gemm_bf16_splitk_fused")
        else:
            call = lambda: check(lib.ivlm_gemm_bf16_splitk(
                x2.data_ptr(), x2.stride(0), weight.data_ptr(), ldw, o2.data_ptr(), o2.stride(0), _p(bias),
                _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out.dtype == F32 else 0, splits, ws.data_ptr(),
                ws.numel() * 4, flags, _stream()), "gemm_bf16_splitk")
    if TIMER.enabled:  # work = algorithmic FLOPs (MFMA path) or weight bytes (GEMV path)
        if M > 16:  # (a split A operand doubles the MFMA work of the same algorithmic product: counted once)
            TIMER.time("gemm_bf16_mfma", 2.0 * M * N * K, call, tag=(M, N, K, act, "split" if a_split else ""))
        else:
            TIMER.time("gemv_bf16", 2.0 * N * K, call, tag=(M, N, K, act))
    else:
        call()
    return out


def _dtc(t):
    return IVLM_F32 if t.dtype == F32 else IVLM_BF16


IVLM_FP8 = 3
U8 = torch.uint8


def layernorm(x, weight, bias, eps=1e-5, gelu=False, out=None, out_f32=False, out_rows=None, fp8_scale=None, out_split=False,
              out_f16=False):
    """x bf16 or fp32 [..., cols] -> bf16 (the next GEMM's operand), fp32 (out_f32: the row is itself a stream) or, with
    fp8_scale (device fp32 scalar), e4m3 bytes of y / scale (uint8 tensor: the operand of linear_fp8); out_split: [hi | lo]
    bf16 rows [..., 2 cols] (the a_split operand of an fp32-activation GEMM)."""
    lib = _lib.load()
    x = _req(x, None, "x")
    cols = x.shape[-1]
    if out is None:
        if out_split:
            out = torch.empty(x.shape[:-1] + (2 * cols,), dtype=F16 if out_f16 else BF16, device=x.device)
        elif out_f16:
            out = torch.empty(x.shape, dtype=F16, device=x.device)
        else:
            out = torch.empty(x.shape, dtype=U8 if fp8_scale is not None else (F32 if out_f32 else BF16), device=x.device)
    y = out
    ydt = IVLM_FP8 if fp8_scale is not None else ((5 if out_f16 else 2) if out_split else (4 if out_f16 else _dtc(y)))
    check(lib.ivlm_layernorm(x.data_ptr(), _dtc(x), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), ydt,
                             x.numel() // cols, cols, float(eps), 1 if gelu else 0, _p(out_rows), _p(fp8_scale), _stream()),
          "layernorm")
    return y


def amax(x, out=None):
    """max |x| as a device fp32 scalar [1] (accumulates into ``out`` if given): per-tensor fp8 scale calibration."""
    lib = _lib.load()
    x = _req(x, None, "x")
    assert x.numel() % 8 == 0
    if out is None:
        out = torch.zeros(1, dtype=F32, device=x.device)
    check(lib.ivlm_amax(x.data_ptr(), _dtc(x), x.numel(), out.data_ptr(), _stream()), "amax")
    return out


def quantize_fp8(x, scale=None):
    """bf16 | fp32 [..., C] -> (e4m3 bytes uint8 [..., C], scale fp32 [1]) with x ~ q * scale; scale None: amax(x) / 448."""
    if scale is None:
        scale = amax(x) / 448.0  # (weights, once at load: a torch op on a device scalar)
        scale.clamp_(min=1e-12)
    C = x.shape[-1]
    q = gather_rows(x.reshape(-1, C), out_kind="fp8", scale=scale).view(x.shape)
    return q, scale


def linear_fp8(xq, wq, scale_a, scale_w, bias=None, act="none", residual=None, out=None, out_kind="bf16", scale_out=None,
               out_rows=None, a_rows=None):
    """act((xq . wq^T) * scale_a * scale_w + bias) + residual with e4m3 operands (uint8 tensors [M,K] / [N,K]) on the MX
    matrix instruction; out_kind 'bf16' | 'f32' | 'fp8' (act(...) / scale_out as e4m3).  Tile GEMM path only (M > 16)."""
    lib = _lib.load()
    K = xq.shape[-1]
    N = wq.shape[0]
    assert xq.dtype == U8 and wq.dtype == U8 and wq.shape[1] == K
    x2 = xq.reshape(-1, K)
    M = x2.shape[0] if a_rows is None else a_rows.numel()
    n_out = N // 2 if act == "swiglu" else N
    if out is None:
        dt = {"bf16": BF16, "f32": F32, "fp8": U8}[out_kind]
        out = torch.empty((M, n_out), dtype=dt, device=xq.device)
    else:
        out_kind = {BF16: "bf16", F32: "f32", U8: "fp8"}[out.dtype]
    o2 = out.reshape(-1, n_out)
    r2, ldr, flags = None, 0, 0
    if residual is not None:
        r2 = residual.reshape(-1, N)
        ldr = r2.stride(0)
        if r2.dtype == F32:
            flags |= GEMM_RES_F32
    kind = {"bf16": IVLM_BF16, "f32": IVLM_F32, "fp8": 2}[out_kind]
    call = lambda: check(lib.ivlm_gemm_fp8(
        x2.data_ptr(), x2.stride(0), wq.data_ptr(), wq.stride(0), o2.data_ptr(), o2.stride(0), _p(bias), _p(r2), ldr, M, N, K,
        ACT[act], kind, scale_a.data_ptr(), scale_w.data_ptr(), _p(scale_out), flags, _p(out_rows), _p(a_rows), _stream()),
        "gemm_fp8")
    if TIMER.enabled:
        TIMER.time("gemm_fp8_mfma", 2.0 * M * N * K, call, tag=(M, N, K, act))
    else:
        call()
    return out


def linear_fp8w(x, wq, scale_w, bias=None, act="none", residual=None, out_f32=True, rms=None):
    """Batch-1 decode linear with e4m3 weights: act((x . wq^T) * scale_w + bias) + residual; x fp32 [1, K] (exact products),
    wq uint8 [N, K], scale_w device fp32 [1]; rms = (weight, eps) fuses the preceding RMSNorm."""
    lib = _lib.load()
    x = _req(x, F32, "x")
    N, K = wq.shape
    assert x.numel() == K and wq.dtype == U8 and wq.stride(1) == 1
    n_out = N // 2 if act == "swiglu" else N
    out = torch.empty(1, n_out, dtype=F32 if out_f32 else BF16, device=x.device)
    flags = 0
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == N
        if residual.dtype == F32:
            flags |= GEMM_RES_F32
    call = lambda: check(lib.ivlm_gemv_fp8w(x.data_ptr(), wq.data_ptr(), wq.stride(0), scale_w.data_ptr(), out.data_ptr(), _p(bias),
                                            _p(residual), N, K, ACT[act], 1 if out_f32 else 0, _p(rms[0]) if rms else 0,
                                            float(rms[1]) if rms else 0.0, flags, _stream()), "gemv_fp8w")
    if TIMER.enabled:
        TIMER.time("gemv_fp8w", 1.0 * N * K, call, tag=(1, N, K, act))  # work = weight BYTES
    else:
        call()
    return out


class PackedBf12:
    """A bf16 weight matrix [N, K] in the lossless 12-bit layout of ``ivlm_gemv1_bf12`` (1.5 bytes per weight): P bytes (sign |
    mantissa), E nibbles (exponent code relative to the row's window), per-row exponent base, CSR patches for the (rare) nonzero
    weights outside the window.  Built once per matrix on the device (weight preparation, like the q|k|v concatenation)."""

    @staticmethod
    def takes(N, K):
        """shapes the fragment layout (the MFMA kernel) takes: 16-row blocks, 64-column step pairs, x as three bf16 planes in LDS"""
        return N % 16 == 0 and K % 64 == 0 and K * 6 <= 100 * 1024

    def __init__(self, w, fragments=True, pad_rows=False, packer="c"):
        """pad_rows: zero rows are appended up to a multiple of 16 so that the fragment layout takes the matrix (lm_head: 32003 rows);
        ``rows`` keeps the true count, ``linear_bf12`` returns that many outputs."""
        assert w.dtype == BF16 and w.dim() == 2 and w.is_cuda and w.shape[1] % 16 == 0
        self.rows = w.shape[0]
        n_rows = -(-w.shape[0] // 16) * 16
        if fragments and packer == "c" and (pad_rows or n_rows == w.shape[0]) and self.takes(n_rows, w.shape[1]):
            # the library's own packer (ivlm_pack_bf12m_*: what a C caller uses): two launches around one host read, no temporaries
            if not bool(torch.isfinite(w).all()):
                raise IvlmError("PackedBf12: inf / nan weights cannot be packed")
            lib, w, K = _lib.load(), w.contiguous(), w.shape[1]
            self.ebase = torch.empty(n_rows, dtype=torch.int32, device=w.device)
            self.patch_ptr = torch.empty(n_rows + 1, dtype=torch.int32, device=w.device)
            check(lib.ivlm_pack_bf12m_count(w.data_ptr(), self.rows, n_rows, K, self.ebase.data_ptr(), self.patch_ptr.data_ptr(),
                                            _stream()), "pack_bf12m_count")
            self.n_patches = int(self.patch_ptr[-1])
            self.patch_col = torch.zeros(max(1, self.n_patches), dtype=torch.int32, device=w.device)
            self.patch_val = torch.zeros(max(1, self.n_patches), dtype=BF16, device=w.device)
            self.P = torch.empty(n_rows // 16, K // 64, 4, 16, 2, 8, dtype=torch.uint8, device=w.device)
            self.E = torch.empty(n_rows // 16, K // 64, 4, 16, 2, 4, dtype=torch.uint8, device=w.device)
            check(lib.ivlm_pack_bf12m_fill(w.data_ptr(), self.rows, n_rows, K, self.ebase.data_ptr(), self.patch_ptr.data_ptr(),
                                           self.P.data_ptr(), self.E.data_ptr(), self.patch_col.data_ptr(), self.patch_val.data_ptr(),
                                           _stream()), "pack_bf12m_fill")
            self.shape, self.frag = (n_rows, K), True
            return
        if pad_rows and w.shape[0] % 16:
            w = torch.cat([w, torch.zeros(16 - w.shape[0] % 16, w.shape[1], dtype=BF16, device=w.device)])
        if not bool(torch.isfinite(w).all()):
            raise IvlmError("PackedBf12: inf / nan weights cannot be packed")
        N, K = w.shape
        bits = w.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        ef = (bits >> 7) & 0xFF
        eb = (ef.max(dim=1).values - 15).clamp_(min=0)
        code = ef - eb[:, None]
        inwin = code >= 1  # (code <= 15 by construction)
        esc = (~inwin) & ((bits & 0x7FFF) != 0)  # nonzero weights below the window (incl. bf16 subnormals): exact patches
        self.P = torch.where(inwin, ((bits >> 8) & 0x80) | (bits & 0x7F), torch.zeros_like(bits)).to(torch.uint8).contiguous()
        code = torch.where(inwin, code, torch.zeros_like(code))
        self.E = (code[:, 0::2] | (code[:, 1::2] << 4)).to(torch.uint8).contiguous()
        self.ebase = eb.to(torch.int32).contiguous()
        rows, cols = esc.nonzero(as_tuple=True)  # (row-major order: sorted by row)
        self.patch_ptr = torch.zeros(N + 1, dtype=torch.int32, device=w.device)
        if rows.numel():
            self.patch_ptr[1:] = torch.bincount(rows, minlength=N).cumsum(0).to(torch.int32)
        self.patch_col = (cols.to(torch.int32) if rows.numel() else torch.zeros(1, dtype=torch.int32, device=w.device)).contiguous()
        self.patch_val = (w[rows, cols] if rows.numel() else torch.zeros(1, dtype=BF16, device=w.device)).contiguous()
        self.shape, self.n_patches = (N, K), int(rows.numel())
        # fragment layout of the MFMA kernel (ivlm_gemv1_bf12m): the SAME bytes in the order the lanes consume them -
        # [N/16][K/64][lane = q*16 + r][h*8 + i] for weight k = sp*64 + h*32 + q*8 + i of row rb*16 + r.  Kept INSTEAD of the row layout.
        self.frag = fragments and self.takes(N, K)
        if self.frag:
            self.P = self.P.view(N // 16, 16, K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous()
            self.E = self.E.view(N // 16, 16, K // 64, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).contiguous()

    def bytes(self):
        return self.P.numel() + self.E.numel() + 4 * self.ebase.numel() + 4 * self.patch_ptr.numel() + 6 * self.n_patches

    def _rows(self):
        """(P [N, K], E [N, K/2]) in the row layout"""
        if not self.frag:
            return self.P, self.E
        N, K = self.shape
        return (self.P.permute(0, 3, 1, 4, 2, 5).reshape(N, K).contiguous(), self.E.permute(0, 3, 1, 4, 2, 5).reshape(N, K // 2).contiguous())

    def _args(self):
        P, E = self._rows()
        self._keep = (P, E)
        return (P.data_ptr(), P.stride(0), E.data_ptr(), E.stride(0), self.ebase.data_ptr(),
                self.patch_ptr.data_ptr(), self.patch_col.data_ptr(), self.patch_val.data_ptr())

    def _args_frag(self):
        return (self.P.data_ptr(), self.E.data_ptr(), self.ebase.data_ptr(), self.patch_ptr.data_ptr(), self.patch_col.data_ptr(),
                self.patch_val.data_ptr())

    def unpack(self):
        """-> bf16 [rows, K], bit-identical to the matrix that was packed."""
        out = torch.empty(self.shape, dtype=BF16, device=self.P.device)
        check(_lib.load().ivlm_unpack_bf12(*self._args(), self.shape[0], self.shape[1], out.data_ptr(), _stream()), "unpack_bf12")
        self._keep = None
        return out[: self.rows]


def linear_bf12(x, wp: PackedBf12, bias=None, act="none", residual=None, out_f32=True, rms=None, parts=None):
    """act(x @ W.T + bias) + residual for ONE fp32 row x against a ``PackedBf12`` weight: the batch-1 decode linears streaming
    1.5 instead of 2 bytes per weight (same exact bf16 x fp32 products and fp32 accumulation as ``linear`` on fp32 x)."""
    lib = _lib.load()
    N, K = wp.shape
    if parts is not None:  # x = the merge of the split-KV attention partials (parts tensor, head dim): fragment layout only
        pt, pD = parts
        assert x is None and wp.frag and rms is None and pt.dtype == F32 and pt.numel() >= (K // pD) * 4 * (pD + 4)
        dev = pt.device
    else:
        assert x.dtype == F32 and x.is_contiguous() and x.dim() == 2 and x.shape[1] == K and 1 <= x.shape[0] <= 16
        dev = x.device
    n_out = N // 2 if act == "swiglu" else N
    M = 1 if parts is not None else x.shape[0]
    out = torch.empty(M, n_out, dtype=F32 if out_f32 else BF16, device=dev)
    if M > 1:  # the batched decode step: M activation rows meet every rebuilt weight fragment (fragment layout only)
        assert wp.frag and wp.rows == N
        flags = 0
        if residual is not None:
            assert residual.dtype in (BF16, F32) and residual.is_contiguous() and tuple(residual.shape) == (M, N)
            flags = GEMM_RES_F32 if residual.dtype == F32 else 0
        call = lambda: check(lib.ivlm_gemv16_bf12m(x.data_ptr(), K, M, *wp._args_frag(), out.data_ptr(), n_out, _p(bias), _p(residual), N,
                                                   N, K, ACT[act], 1 if out_f32 else 0, _p(rms[0]) if rms else 0,
                                                   float(rms[1]) if rms else 0.0, flags, _stream()), "gemv16_bf12m")
        if TIMER.enabled:
            TIMER.time("gemv_bf16", 2.0 * N * K, call, tag=(M, N, K, act, "bf12"))
        else:
            call()
        return out
    flags = 0
    if residual is not None:
        assert residual.dtype in (BF16, F32) and residual.is_contiguous() and residual.numel() == N
        if residual.dtype == F32:
            flags |= GEMM_RES_F32
    if parts is not None:
        call = lambda: check(lib.ivlm_gemv1_bf12m_parts(pt.data_ptr(), pD, *wp._args_frag(), out.data_ptr(), _p(bias), _p(residual), N, K,
                                                        ACT[act], 1 if out_f32 else 0, flags, _stream()), "gemv1_bf12m_parts")
    elif wp.frag:  # MFMA kernel on the fragment layout
        call = lambda: check(lib.ivlm_gemv1_bf12m(x.data_ptr(), *wp._args_frag(), out.data_ptr(), _p(bias), _p(residual), N, K,
                                                  ACT[act], 1 if out_f32 else 0, _p(rms[0]) if rms else 0,
                                                  float(rms[1]) if rms else 0.0, flags, _stream()), "gemv1_bf12m")
    else:
        call = lambda: check(lib.ivlm_gemv1_bf12(x.data_ptr(), *wp._args(), out.data_ptr(), _p(bias), _p(residual), N, K, ACT[act],
                                                 1 if out_f32 else 0, _p(rms[0]) if rms else 0, float(rms[1]) if rms else 0.0,
                                                 flags, _stream()), "gemv1_bf12")
    if TIMER.enabled:  # work = the ALGORITHMIC bytes (the bf16 matrix), like the other decode linears
        TIMER.time("gemv_bf16", 2.0 * N * K, call, tag=(1, N, K, act, "bf12"))
    else:
        call()
    return out if wp.rows == N or act == "swiglu" else out[:, : wp.rows]


def rmsnorm(x, weight, eps=1e-5, out_f32=False, out_split=False, fp8_scale=None, out_f16=False):
    lib = _lib.load()
    x = _req(x, None, "x")
    cols = x.shape[-1]
    if fp8_scale is not None:  # e4m3 bytes of the normalised row / scale: the operand of linear_fp8
        y = torch.empty(x.shape, dtype=U8, device=x.device)
        check(lib.ivlm_rmsnorm_fp8(x.data_ptr(), _dtc(x), weight.data_ptr(), y.data_ptr(), x.numel() // cols, cols, float(eps),
                                   fp8_scale.data_ptr(), _stream()), "rmsnorm_fp8")
        return y
    if out_split:
        y = torch.empty(x.shape[:-1] + (2 * cols,), dtype=BF16, device=x.device)
    else:
        y = torch.empty(x.shape, dtype=F16 if out_f16 else (F32 if out_f32 else BF16), device=x.device)
    check(lib.ivlm_rmsnorm(x.data_ptr(), _dtc(x), weight.data_ptr(), y.data_ptr(), 2 if out_split else (4 if out_f16 else _dtc(y)),
                           x.numel() // cols, cols, float(eps), _stream()), "rmsnorm")
    return y


def bf16_to_f16(w):
    """bf16 tensor -> IEEE fp16 copy (RNE; out of range -> inf): the weight operands of ``linear`` on fp16 activations."""
    w = w.contiguous()
    assert w.dtype == torch.bfloat16
    out = torch.empty(w.shape, dtype=torch.float16, device=w.device)
    check(_lib.load().ivlm_bf16_to_f16(w.data_ptr(), out.data_ptr(), w.numel(), _stream()), "bf16_to_f16")
    return out


def f16_weight(w, name="weight"):
    """fp16 copy of a bf16 weight matrix for the fp16-operand GEMMs, checked: a bf16 value inside the fp16 normal range converts
    exactly; one below it moves by at most 2^-25; one above it (> 65504) cannot be represented - raise."""
    w16 = bf16_to_f16(w)
    err = float((w16.float() - w.float()).abs().max())
    if not (err <= 2.0 ** -24) or not bool(torch.isfinite(w16).all()):
        raise IvlmError(f"{name}: weights outside the fp16 range (max conversion error {err:.3g}) - use a split (hi + lo) "
                        f"precision mode instead of fp16 operands")
    return w16


def relpos_table64(tab_h, tab_w):
    """[rel_pos_h ; rel_pos_w ; zeros] as a bf16 [64, D] table: the operand of the attention kernels' TABLE MODE (rel_tab=...)."""
    n = tab_h.shape[0] + tab_w.shape[0]
    assert n <= 64
    t = torch.zeros(64, tab_h.shape[1], dtype=BF16, device=tab_h.device)
    t[: tab_h.shape[0]] = tab_h
    t[tab_h.shape[0]: n] = tab_w
    return t


def attention(q, k, v, scale, causal=False, q_pos0=0, rel=None, out=None, prescale_q=False, rel_tab=None, q_lo=None, q_lo_level=1):
    """q [B,H,Sq,D], k/v [Bk,H,Sk,D] (arbitrary strides, last dim contiguous; B % Bk == 0: K/V of batch
    b // (B//Bk)) -> o [B,H,Sq,D] as a view of a [B,Sq,H,D] buffer (so o.transpose(1,2) is contiguous).
    rel = (rel_h f32 [B*H,Sq,KH], rel_w f32 [B*H,Sq,KW]) adds SAM's decomposed rel-pos bias; rel_tab = (relpos_table64(...), side)
    lets the kernel compute those terms itself (14 x 14 windows)."""
    import ctypes

    lib = _lib.load()
    B, H, Sq, D = q.shape
    Bk, Sk = k.shape[0], k.shape[2]
    dt = q.dtype  # bf16, or IEEE fp16 (every tensor, incl. the table of table mode): the f16 matrix instruction
    assert dt in (BF16, F16) and k.dtype == dt and v.dtype == dt
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and B % Bk == 0
    if out is None:
        out = torch.empty(B, Sq, H, D, dtype=dt, device=q.device).permute(0, 2, 1, 3)
    assert out.stride(3) == 1 and out.dtype == dt
    st = (ctypes.c_int64 * 12)(q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                               v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2))
    rel_h = rel_w = None
    kh = kw = 0
    if rel is not None:
        rel_h, rel_w = rel
        assert rel_h.dtype == torch.float32 and rel_h.is_contiguous() and rel_w.is_contiguous()
        kh, kw = rel_h.shape[-1], rel_w.shape[-1]
    if rel_tab is not None:  # table mode: rel_h = the bf16 table, rel_w = NULL
        rel_h, kh = rel_tab
        kw = kh
        assert rel is None and rel_h.dtype == dt and rel_h.is_contiguous() and rel_h.shape[1] == D
        assert rel_h.shape[0] == 64 if 2 * kh <= 32 else (kh == 64 and rel_h.shape[0] >= 254)  # windows | the 64 x 64 grid
    if q_lo is not None:  # fp16 "exact q": q = q + q_lo as IEEE halves (SAM shapes; the strides of q)
        assert dt == F16 and q_lo.dtype == F16 and q_lo.stride() == q.stride() and q_lo.shape == q.shape and not causal and B == Bk
        check(lib.ivlm_attention_f16_qsplit(q.data_ptr(), q_lo.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                            ctypes.cast(st, ctypes.c_void_p), B, H, Sq, Sk, D, float(scale), _p(rel_h), _p(rel_w),
                                            kh, kw, int(q_lo_level), _stream()), "attention_f16_qsplit")
        return out
    fn = lib.ivlm_attention_f16 if dt == F16 else lib.ivlm_attention_bf16
    check(fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
             ctypes.cast(st, ctypes.c_void_p), B, H, Sq, Sk, D, float(scale), 1 if causal else 0,
             int(q_pos0), _p(rel_h), _p(rel_w), kh, kw, B // Bk,
             1 if (prescale_q or rel is not None or rel_tab is not None) else 0, _stream()), "attention")
    return out


def attention_split(q, q_lo, k, k_lo, v, v_lo, scale, causal=False, q_pos0=0, rel=None, out=None, prescale_q=False, rel_tab=None):
    """"Parity" precision of ``attention``: every operand as hi + lo bf16 planes (the *_lo tensors have the shapes and strides of
    the hi ones), fp32-operand arithmetic on the bf16 matrix cores.  out: bf16 [B, Sq, 2, H, D] buffer (default: allocated), the
    rows [hi(H*D) | lo(H*D)] of the next GEMM's a_split operand; returned as [B*Sq, 2*H*D]."""
    import ctypes

    lib = _lib.load()
    B, H, Sq, D = q.shape
    Bk, Sk = k.shape[0], k.shape[2]
    for t, tl in ((q, q_lo), (k, k_lo), (v, v_lo)):
        assert t.dtype == BF16 and tl.dtype == BF16 and t.stride() == tl.stride() and t.shape == tl.shape and t.stride(3) == 1
    assert B % Bk == 0
    if out is None:
        out = torch.empty(B, Sq, 2, H, D, dtype=BF16, device=q.device)
    assert out.shape == (B, Sq, 2, H, D) and out.stride(4) == 1
    oh, ol = out[:, :, 0].permute(0, 2, 1, 3), out[:, :, 1].permute(0, 2, 1, 3)  # [B,H,Sq,D] views
    st = (ctypes.c_int64 * 12)(q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                               v.stride(0), v.stride(1), v.stride(2), oh.stride(0), oh.stride(1), oh.stride(2))
    rel_h = rel_w = None
    kh = kw = 0
    if rel is not None:
        rel_h, rel_w = rel
        assert rel_h.dtype == torch.float32 and rel_h.is_contiguous() and rel_w.is_contiguous()
        kh, kw = rel_h.shape[-1], rel_w.shape[-1]
    if rel_tab is not None:  # table mode (see ``attention``)
        rel_h, kh = rel_tab
        kw = kh
        assert rel is None and rel_h.dtype == BF16 and rel_h.shape == (64, D) and rel_h.is_contiguous()
    check(lib.ivlm_attention_bf16_split(q.data_ptr(), q_lo.data_ptr(), k.data_ptr(), k_lo.data_ptr(), v.data_ptr(),
                                        v_lo.data_ptr(), oh.data_ptr(), ol.data_ptr(), ctypes.cast(st, ctypes.c_void_p), B, H,
                                        Sq, Sk, D, float(scale), 1 if causal else 0, int(q_pos0), _p(rel_h), _p(rel_w), kh, kw,
                                        B // Bk, 1 if (prescale_q or rel is not None or rel_tab is not None) else 0, _stream()),
          "attention_split")
    return out.view(B * Sq, 2 * H * D)


def relpos_bias_split(q, q_lo, tab_h, tab_w, SH, SW):
    """"Parity" precision of ``relpos_bias``: q = hi + lo planes, results unrounded fp32 (the VALU dot-product kernel)."""
    lib = _lib.load()
    B, H, S, D = q.shape
    if q_lo is None:
        q_lo = q  # bf16 q without a lo plane: only the rounding of the results to bf16 is dropped
    assert S == SH * SW and q.stride(3) == 1 and q.stride() == q_lo.stride() and tab_h.is_contiguous() and tab_w.is_contiguous()
    rel_h = torch.empty(B * H, S, SH, dtype=torch.float32, device=q.device)
    rel_w = torch.empty(B * H, S, SW, dtype=torch.float32, device=q.device)
    check(lib.ivlm_relpos_bias_split(q.data_ptr(), q_lo.data_ptr(), q.stride(0), q.stride(1), q.stride(2), tab_h.data_ptr(),
                                     tab_w.data_ptr(), B, H, SH, SW, D, rel_h.data_ptr(), rel_w.data_ptr(), _stream()),
          "relpos_bias_split")
    return rel_h, rel_w


def attention_f32(q, k, v, scale, out=None):
    """fp32 q [B,H,Sq,D], k/v [Bk,H,Sk,D] (D in 16, 32; arbitrary strides % 4) -> o [B,H,Sq,D] fp32 as a view of a
    [B,Sq,H,D] buffer; scores = (q.k) * scale, no operand rounding (SAM mask decoder)."""
    import ctypes

    lib = _lib.load()
    B, H, Sq, D = q.shape
    Bk, Sk = k.shape[0], k.shape[2]
    assert q.dtype == F32 and k.dtype == F32 and v.dtype == F32 and B % Bk == 0
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if out is None:
        out = torch.empty(B, Sq, H, D, dtype=F32, device=q.device).permute(0, 2, 1, 3)
    st = (ctypes.c_int64 * 12)(q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                               v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2))
    check(lib.ivlm_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), ctypes.cast(st, ctypes.c_void_p),
                                 B, H, Sq, Sk, D, float(scale), B // Bk, _stream()), "attention_f32")
    return out


def relpos_tables_cat(tab_h, tab_w):
    """[rel_pos_h ; rel_pos_w] padded with zero rows to a multiple of 64: the weight matrix of the GEMM formulation (global
    blocks: 254 -> 256 rows) and the table of the attention kernels' table mode (windows: 54 -> 64 rows)."""
    n = tab_h.shape[0] + tab_w.shape[0]
    cat = torch.zeros((n + 63) // 64 * 64, tab_h.shape[1], dtype=BF16, device=tab_h.device)
    cat[: tab_h.shape[0]] = tab_h
    cat[tab_h.shape[0]: n] = tab_w
    return cat


RELPOS_GEMM = True  # rel-pos operands through one batched MFMA GEMM + gather instead of the VALU dot-product kernel


def relpos_bias(q, tab_h, tab_w, SH, SW, cat=None, q_lo=None):
    """q [B,H,S=SH*SW,D] -> (rel_h f32 [B*H,S,SH], rel_w f32 [B*H,S,SW]).  cat = relpos_tables_cat(tab_h, tab_w)
    (pre-built once per block) selects the GEMM formulation when the q rows of all (b, s) are uniformly strided."""
    lib = _lib.load()
    B, H, S, D = q.shape
    assert S == SH * SW and q.stride(3) == 1 and tab_h.is_contiguous() and tab_w.is_contiguous()
    rel_h = torch.empty(B * H, S, SH, dtype=torch.float32, device=q.device)
    rel_w = torch.empty(B * H, S, SW, dtype=torch.float32, device=q.device)
    # (measured, SAM ViT-H: global 64x64 grid 242 -> 139 us; 14x14 windows 54 -> 62 us - the GEMM's N = 54 wastes half a tile and
    #  the gather moves as many bytes as the dot kernel writes - so only grids of 32x32 and up take this path)
    if q.dtype == F16:  # fp16 q (optionally + q_lo) against the fp16 table: fp32 G and terms, one call
        if not (cat is not None and cat.dtype == F16 and min(SH, SW) >= 32):
            raise IvlmError("relpos_bias: fp16 q needs the fp16 cat table and a grid of 32 x 32 or more (GEMM formulation)")
        npad, M = cat.shape[0], B * S
        assert q_lo is None or (q_lo.dtype == F16 and q_lo.stride() == q.stride())
        G = torch.empty(H, M, npad, dtype=F32, device=q.device)
        call = lambda: check(lib.ivlm_relpos_bias_f16(q.data_ptr(), _p(q_lo), q.stride(0), q.stride(1), q.stride(2), cat.data_ptr(), npad,
                                                      B, H, SH, SW, D, G.data_ptr(), G.numel() * 4, rel_h.data_ptr(), rel_w.data_ptr(),
                                                      _stream()), "relpos_bias_f16")
        if TIMER.enabled:
            TIMER.time("gemm_bf16_mfma", 2.0 * H * M * npad * D, call, tag=("relpos", M, npad, D))
        else:
            call()
        return rel_h, rel_w
    if (RELPOS_GEMM and cat is not None and min(SH, SW) >= 32 and q.dtype == cat.dtype and q.stride(0) == S * q.stride(2) and q.stride(2) % 8 == 0
            and q.stride(1) % 8 == 0 and D % 8 == 0 and q.data_ptr() % 16 == 0):
        npad, M = cat.shape[0], B * S
        G = torch.empty(H, M, npad, dtype=BF16, device=q.device)
        call = lambda: check(lib.ivlm_gemm_bf16(q.data_ptr(), q.stride(2), cat.data_ptr(), D, G.data_ptr(), npad, 0, 0, 0, 0,
                                                M, npad, D, 0, 0, H, q.stride(1), 0, M * npad, 0, 0, 0.0, 0, 0, 0, _stream()),
                             "relpos gemm")
        if TIMER.enabled:
            TIMER.time("gemm_bf16_mfma", 2.0 * H * M * npad * D, call, tag=("relpos", M, npad, D))
        else:
            call()
        check(lib.ivlm_relpos_gather(G.data_ptr(), M * npad, npad, B, H, SH, SW, rel_h.data_ptr(), rel_w.data_ptr(), _stream()),
              "relpos_gather")
        return rel_h, rel_w
    check(lib.ivlm_relpos_bias(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), tab_h.data_ptr(), tab_w.data_ptr(),
                               B, H, SH, SW, D, rel_h.data_ptr(), rel_w.data_ptr(), _stream()), "relpos_bias")
    return rel_h, rel_w


def argmax(logits, bump=None):
    """torch.argmax(logits, -1) as int32 [rows]; bump (int32 [rows] on the device): += 1 in the same launch (decode positions)."""
    lib = _lib.load()
    logits = _req(logits, torch.float32, "logits")
    rows, cols = logits.shape
    out = torch.empty(rows, dtype=torch.int32, device=logits.device)
    if bump is not None:
        assert bump.dtype == torch.int32 and bump.is_cuda and bump.numel() == rows and bump.is_contiguous()
        check(lib.ivlm_argmax_f32_bump(logits.data_ptr(), rows, cols, out.data_ptr(), bump.data_ptr(), _stream()), "argmax_bump")
        return out
    check(lib.ivlm_argmax_f32(logits.data_ptr(), rows, cols, out.data_ptr(), _stream()), "argmax")
    return out


# --------------------------------------------------------------------------------------------
# data movement
# --------------------------------------------------------------------------------------------
def im2col_nchw(x, ks, stride, kpad=None):
    lib = _lib.load()
    x = _req(x, BF16, "x")
    B, C, H, W = x.shape
    K = C * ks * ks
    kpad = kpad or ((K + 63) // 64) * 64
    gh, gw = (H - ks) // stride + 1, (W - ks) // stride + 1
    out = torch.empty(B * gh * gw, kpad, dtype=BF16, device=x.device)
    check(lib.ivlm_im2col_nchw(x.data_ptr(), out.data_ptr(), B, C, H, W, ks, stride, kpad, _stream()), "im2col_nchw")
    return out


def im2col3x3_nhwc(x):
    lib = _lib.load()
    x = _req(x, BF16, "x")
    B, H, W, C = x.shape
    out = torch.empty(B * H * W, 9 * C, dtype=BF16, device=x.device)
    check(lib.ivlm_im2col3x3_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, C, _stream()), "im2col3x3")
    return out


def im2col3x3_nhwc_split(x_split, B, H, W, C):
    """x_split bf16 [B*H*W, 2C] = [hi | lo] rows -> [B*H*W, 2 * 9C]: both halves unfolded (the a_split operand of the 3x3 conv)."""
    lib = _lib.load()
    x_split = _req(x_split, BF16, "x_split")
    assert x_split.shape == (B * H * W, 2 * C)
    out = torch.empty(B * H * W, 18 * C, dtype=BF16, device=x_split.device)
    for half in range(2):
        check(lib.ivlm_im2col3x3_nhwc_strided(x_split.data_ptr() + half * C * 2, 2 * C, out.data_ptr() + half * 9 * C * 2,
                                              18 * C, B, H, W, C, _stream()), "im2col3x3_strided")
    return out


_KIND = {"bf16": IVLM_BF16, "f32": IVLM_F32, "split": 2, "fp8": 3}


def gather_rows(src, idx=None, add=None, out=None, out_kind=None, scale=None):
    """out[r] = src[idx[r]] (zeros where idx < 0; idx None: out[r] = src[r]) (+ add[r]); src [R,C] / add [n,C] rows may be
    strided, bf16 or fp32.  out_kind 'bf16' | 'f32' | 'split' ([hi | lo] bf16 rows of width 2C, see split_rows); default: the
    dtype of ``out`` if given, else of ``add`` if given, else of ``src``."""
    lib = _lib.load()
    assert src.dtype in (BF16, F32) and src.stride(-1) == 1
    cols = src.shape[-1]
    if idx is not None:
        assert idx.dtype == torch.int32 and idx.is_contiguous()
        rows = idx.numel()
    else:
        src = src.reshape(-1, cols)
        rows = src.shape[0]
    if out_kind is None:
        ref = out if out is not None else (add if add is not None else src)
        out_kind = "f32" if ref.dtype == F32 else "bf16"
    if out is None:
        out = (torch.empty(rows, 2 * cols, dtype=BF16, device=src.device) if out_kind == "split"
               else torch.empty(rows, cols, dtype={"f32": F32, "fp8": torch.uint8}.get(out_kind, BF16), device=src.device))
    assert out.stride(-1) == 1 and out.dtype == {"f32": F32, "fp8": torch.uint8}.get(out_kind, BF16)
    assert (out_kind == "fp8") == (scale is not None)
    lda = 0
    if add is not None:
        assert add.dtype in (BF16, F32) and add.stride(-1) == 1
        lda = add.stride(0)
    check(lib.ivlm_gather_rows(out.data_ptr(), _KIND[out_kind], out.stride(0), src.data_ptr(), _dtc(src), src.stride(0),
                               _p(idx), _p(add), _dtc(add) if add is not None else 0, lda, rows, cols, _p(scale), _stream()),
          "gather_rows")
    return out


def fill_rows(dst, idx, row):
    """dst[idx[r]] = row (bf16 [C] broadcast to the listed rows of dst [R, C])."""
    lib = _lib.load()
    assert dst.dtype in (BF16, F16) and row.dtype == dst.dtype and dst.stride(-1) == 1 and row.is_contiguous()  # (a 16-bit copy)
    assert idx.dtype == torch.int32 and idx.is_contiguous() and row.numel() == dst.shape[-1]
    check(lib.ivlm_fill_rows(dst.data_ptr(), dst.stride(0), idx.data_ptr(), idx.numel(), row.data_ptr(), dst.shape[-1],
                             _stream()), "fill_rows")
    return dst


def split_rows(x):
    """fp32 [..., C] -> bf16 [..., 2C] = [hi | lo] with x = hi + lo to 2^-17: the A operand of an fp32-activation GEMM on the
    bf16 matrix cores, to be multiplied with ``torch.cat([W, W], 1)`` (K' = 2K)."""
    C = x.shape[-1]
    return gather_rows(x.reshape(-1, C), out_kind="split").view(x.shape[:-1] + (2 * C,))


def add_rows(a, b, out=None, op="add", out_kind=None):
    """a [R,C] (+|*) b [r,C] broadcast with row modulo (R % r == 0 not required); a / b bf16 or fp32 -> bf16, fp32 or split."""
    lib = _lib.load()
    a = _req(a, None, "a")
    b = _req(b, None, "b")
    cols = a.shape[-1]
    if out_kind is None:
        out_kind = "f32" if (out if out is not None else a).dtype == F32 else "bf16"
    if out is None:
        out = (torch.empty(a.shape[:-1] + (2 * cols,), dtype=BF16, device=a.device) if out_kind == "split"
               else torch.empty(a.shape, dtype=F32 if out_kind == "f32" else BF16, device=a.device))
    check(lib.ivlm_add_rows(out.data_ptr(), _KIND[out_kind], a.data_ptr(), _dtc(a), b.data_ptr(), _dtc(b),
                            a.numel() // cols, cols, b.numel() // cols, 1 if op == "mul" else 0, _stream()), "add_rows")
    return out


def uncertainty_mlp(emb, w1, b1, w2, b2, w3, b3):
    """UncertaintyModule's per-pixel MLP (components.py:55-78) with the bf16 module's rounding points: emb fp32 [..., 256]
    (channels last) -> fp32 [...] holding bf16 values."""
    emb = _req(emb, F32, "emb")
    assert emb.shape[-1] == 256 and tuple(w1.shape) == (64, 256) and tuple(w2.shape) == (16, 64) and w3.numel() == 16
    ws = [_req(t, BF16, "w") for t in (w1, b1, w2, b2, w3, b3)]
    out = torch.empty(emb.shape[:-1], dtype=F32, device=emb.device)
    check(_lib.load().ivlm_uncertainty_mlp(emb.data_ptr(), out.numel(), *[t.data_ptr() for t in ws], out.data_ptr(), _stream()),
          "uncertainty_mlp")
    return out


def resize_bilinear(src, size, dtype=torch.float32):
    """F.interpolate(src, size, mode='bilinear', align_corners=False): src fp32 [..., h, w] -> [..., oh, ow] fp32 or bf16."""
    src = _req(src, F32, "src")
    h, w = src.shape[-2:]
    oh, ow = int(size[0]), int(size[1])
    out = torch.empty(src.shape[:-2] + (oh, ow), dtype=dtype, device=src.device)
    n = src.numel() // (h * w)
    check(_lib.load().ivlm_resize_bilinear(src.data_ptr(), n, h, w, out.data_ptr(), _dtc(out), oh, ow, _stream()), "resize_bilinear")
    return out


def dense_pe(gauss, h, w, dtype=torch.float32):
    lib = _lib.load()
    gauss = _req(gauss, torch.float32, "gauss")
    nf = gauss.shape[1]
    pe = torch.empty(h * w, 2 * nf, dtype=dtype, device=gauss.device)
    check(lib.ivlm_dense_pe(gauss.data_ptr(), pe.data_ptr(), _dtc(pe), h, w, nf, _stream()), "dense_pe")
    return pe


def rope_kv(qkv, H, D, pos0, theta, kcache=None, vcache=None, table=None):
    """qkv [T, 3*H*D] (in place). table = (cos, sin) fp32 [Tmax, D/2] from rope_table()."""
    lib = _lib.load()
    assert qkv.dtype in (BF16, F16) and qkv.stride(-1) == 1 and (kcache is None or (kcache.dtype == qkv.dtype == vcache.dtype))
    T = qkv.shape[0]
    check((lib.ivlm_rope_kv_f16 if qkv.dtype == F16 else lib.ivlm_rope_kv)(qkv.data_ptr(), qkv.stride(0), T, H, D, int(pos0), float(theta), _p(kcache), _p(vcache),
                                                                           _p(table[0]) if table else 0, _p(table[1]) if table else 0, _stream()), "rope_kv")
    return qkv


def rope_kv_split(qkv, H, D, pos0, caches, table):
    """"Parity" precision of rope_kv: qkv bf16 [T, 2*3*H*D] = [hi | lo] rows (in place); caches = (k, k_lo, v, v_lo) each
    [Tmax, H, D] or None; table = (cos, sin) from rope_table()."""
    lib = _lib.load()
    assert qkv.dtype == BF16 and qkv.stride(-1) == 1 and qkv.shape[-1] == 6 * H * D
    T = qkv.shape[0]
    kc = caches if caches is not None else (None,) * 4
    check(lib.ivlm_rope_kv_split(qkv.data_ptr(), qkv.stride(0), T, H, D, int(pos0), _p(kc[0]), _p(kc[1]), _p(kc[2]), _p(kc[3]),
                                 table[0].data_ptr(), table[1].data_ptr(), _stream()), "rope_kv_split")
    return qkv


def rope_table(T, D, theta, device):
    lib = _lib.load()
    c = torch.empty(T, D // 2, dtype=torch.float32, device=device)
    s = torch.empty_like(c)
    check(lib.ivlm_rope_table(c.data_ptr(), s.data_ptr(), T, D, float(theta), _stream()), "rope_table")
    return c, s


def mask_dot(up, hyper, B, gh, gw):
    lib = _lib.load()
    up = _req(up, None, "up")
    hyper = _req(hyper, up.dtype, "hyper")
    C = hyper.shape[-1]
    low = torch.empty(B, 4 * gh, 4 * gw, dtype=F32, device=up.device)
    check(lib.ivlm_mask_dot(up.data_ptr(), hyper.data_ptr(), _dtc(up), low.data_ptr(), B, gh, gw, C, _stream()), "mask_dot")
    return low


def llama_decode_attn_parts(qkv, kcache, vcache, H, D, pos, theta, scale, parts, table=None):
    """Split-KV decode attention WITHOUT the merge: RoPE + cache append + four key ranges per head -> parts fp32 [H, 4, D + 4] (o
    unnormalised | max | sum | pad) for ``linear_bf12(..., parts=)`` (the o_proj merges them while it stages its activation row)."""
    lib = _lib.load()
    assert qkv.dtype == F32 and qkv.is_contiguous() and kcache.is_contiguous() and vcache.is_contiguous()
    assert parts.dtype == F32 and parts.is_contiguous() and parts.numel() >= H * 4 * (D + 4) and kcache.dtype in (BF16, F16)
    dev_pos = isinstance(pos, torch.Tensor)
    check(lib.ivlm_llama_decode_attn_parts(qkv.data_ptr(), 4 if kcache.dtype == F16 else IVLM_BF16, kcache.data_ptr(), vcache.data_ptr(),
                                           kcache.shape[0], parts.data_ptr(), H, D, 0 if dev_pos else int(pos),
                                           pos.data_ptr() if dev_pos else 0, float(theta), float(scale),
                                           _p(table[0]) if table else 0, _p(table[1]) if table else 0, _stream()),
          "llama_decode_attn_parts")
    return parts


def decode_attn_scratch(H, D, device):
    """Zeroed scratch of the split-KV decode attention (per-head arrival counters + range partials): one per model / stream."""
    n = int(_lib.load().ivlm_llama_decode_attn_splitkv_scratch_bytes(H, D))
    return torch.zeros(n, dtype=torch.uint8, device=device)


def llama_decode_attn(qkv, kcache, vcache, H, D, pos, theta, scale, out=None, table=None, lo=None, scratch=None):
    """qkv bf16 | fp32 [1, 3*H*D] of the newest token -> o (same dtype) [1, H*D]; RoPE + cache append fused.
    pos: python int, or an int32 device tensor [1] (read by the kernel: HIP-graph friendly).  kcache [Tmax, H, D].
    lo = (kcache_lo, vcache_lo): "parity" precision, K / V cached as hi + lo planes (fp32 qkv only).
    scratch (decode_attn_scratch; fp32 qkv, no lo planes): the split-KV kernel - H x S blocks instead of H."""
    lib = _lib.load()
    assert qkv.dtype in (BF16, F32) and qkv.is_contiguous() and kcache.is_contiguous() and vcache.is_contiguous()
    if out is None:
        out = torch.empty(1, H * D, dtype=qkv.dtype, device=qkv.device)
    dev_pos = isinstance(pos, torch.Tensor)
    if dev_pos:
        assert pos.dtype == torch.int32 and pos.is_cuda
    if scratch is not None and lo is None and qkv.dtype == F32:
        assert kcache.dtype in (BF16, F16) and vcache.dtype == kcache.dtype
        check(lib.ivlm_llama_decode_attn_splitkv(qkv.data_ptr(), 4 if kcache.dtype == F16 else IVLM_BF16,
                                                 kcache.data_ptr(), vcache.data_ptr(), kcache.shape[0], out.data_ptr(), H, D,
                                                 0 if dev_pos else int(pos), pos.data_ptr() if dev_pos else 0, float(theta),
                                                 float(scale), _p(table[0]) if table else 0, _p(table[1]) if table else 0,
                                                 scratch.data_ptr(), scratch.numel(), _stream()), "llama_decode_attn_splitkv")
        return out
    if lo is not None:
        assert qkv.dtype == F32 and lo[0].is_contiguous() and lo[1].is_contiguous() and lo[0].shape == kcache.shape
        check(lib.ivlm_llama_decode_attn_split(qkv.data_ptr(), kcache.data_ptr(), lo[0].data_ptr(), vcache.data_ptr(),
                                               lo[1].data_ptr(), kcache.shape[0], out.data_ptr(), H, D,
                                               0 if dev_pos else int(pos), pos.data_ptr() if dev_pos else 0, float(theta),
                                               float(scale), _p(table[0]) if table else 0, _p(table[1]) if table else 0,
                                               _stream()), "llama_decode_attn_split")
        return out
    if kcache.dtype == F16:  # fp16 cache (the fp16-operand prefill): fp32 qkv / o
        assert qkv.dtype == F32 and vcache.dtype == F16
        check(lib.ivlm_llama_decode_attn_f16(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), kcache.shape[0], out.data_ptr(),
                                             H, D, 0 if dev_pos else int(pos), pos.data_ptr() if dev_pos else 0, float(theta),
                                             float(scale), _p(table[0]) if table else 0, _p(table[1]) if table else 0, _stream()),
              "llama_decode_attn_f16")
        return out
    check(lib.ivlm_llama_decode_attn(qkv.data_ptr(), _dtc(qkv), kcache.data_ptr(), vcache.data_ptr(), kcache.shape[0],
                                     out.data_ptr(), H, D, 0 if dev_pos else int(pos), pos.data_ptr() if dev_pos else 0,
                                     float(theta), float(scale), _p(table[0]) if table else 0,
                                     _p(table[1]) if table else 0, _stream()), "llama_decode_attn")
    return out


def llama_decode_attn_batch(qkv, kcache, vcache, H, D, pos_dev, theta, scale, table=None, out=None, lo=None):
    """One decode step of B sequences: qkv bf16 | fp32 [B, 3*H*D], kcache/vcache bf16 [B, Tmax, H, D] (one slab per
    sequence), pos_dev int32 [B] on the device -> o [B, H*D].  A sequence whose position has reached Tmax is skipped: its
    output row is zero (written by the kernel)."""
    lib = _lib.load()
    B = qkv.shape[0]
    assert qkv.dtype in (BF16, F32) and qkv.stride(1) == 1 and kcache.dim() == 4 and kcache.shape[0] == B
    assert kcache[0].is_contiguous() and vcache[0].is_contiguous() and kcache.stride(0) == vcache.stride(0)
    assert pos_dev.dtype == torch.int32 and pos_dev.is_cuda and pos_dev.numel() == B and pos_dev.is_contiguous()
    if out is None:
        out = torch.empty(B, H * D, dtype=qkv.dtype, device=qkv.device)
    if lo is not None:  # "parity" precision: hi + lo cache planes
        assert qkv.dtype == F32 and lo[0].stride() == kcache.stride() and lo[1].stride() == vcache.stride()
        check(lib.ivlm_llama_decode_attn_batch_split(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), lo[0].data_ptr(),
                                                     vcache.data_ptr(), lo[1].data_ptr(), kcache.stride(0), kcache.shape[1],
                                                     out.data_ptr(), out.stride(0), B, H, D, pos_dev.data_ptr(), float(theta),
                                                     float(scale), _p(table[0]) if table else 0, _p(table[1]) if table else 0,
                                                     _stream()), "llama_decode_attn_batch_split")
        return out
    if kcache.dtype == F16:
        assert qkv.dtype == F32 and vcache.dtype == F16
        check(lib.ivlm_llama_decode_attn_batch_f16(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                                   kcache.stride(0), kcache.shape[1], out.data_ptr(), out.stride(0), B, H, D,
                                                   pos_dev.data_ptr(), float(theta), float(scale), _p(table[0]) if table else 0,
                                                   _p(table[1]) if table else 0, _stream()), "llama_decode_attn_batch_f16")
        return out
    check(lib.ivlm_llama_decode_attn_batch(qkv.data_ptr(), _dtc(qkv), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                           kcache.stride(0), kcache.shape[1], out.data_ptr(), out.stride(0), B, H, D,
                                           pos_dev.data_ptr(), float(theta), float(scale), _p(table[0]) if table else 0,
                                           _p(table[1]) if table else 0, _stream()), "llama_decode_attn_batch")
    return out


def llama_attn_oproj(qkv, kcache, vcache, wo, x, H, D, pos_dev, step_dev, counter, status, theta, scale, table, scratch):
    """One launch: single-token attention of every head + o_proj GEMV + residual -> x_out fp32 [1, H*D] (qkv, x fp32)."""
    lib = _lib.load()
    assert qkv.dtype == F32 and x.dtype == F32 and scratch.dtype == F32
    out = torch.empty(1, H * D, dtype=F32, device=qkv.device)
    check(lib.ivlm_llama_attn_oproj(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), kcache.shape[0], scratch.data_ptr(),
                                    wo.data_ptr(), x.data_ptr(), out.data_ptr(), H, D, float(theta), float(scale),
                                    table[0].data_ptr(), table[1].data_ptr(), pos_dev.data_ptr(), step_dev.data_ptr(),
                                    counter.data_ptr(), status.data_ptr(), _stream()), "llama_attn_oproj")
    return out


# --------------------------------------------------------------------------------------------
# right after the path: metrics, SMPL -> SMPL-X transfer
# --------------------------------------------------------------------------------------------
def contact_prf(gt, pred, threshold=0.5):
    """get_h_contact_metrics (utils/eval_utils.py:63-94) -> f32 [B,3] = (f1, precision, recall) per sample."""
    lib = _lib.load()
    gt = _req(gt, torch.float32, "gt")
    pred = _req(pred, torch.float32, "pred")
    B, n = pred.shape
    out = torch.empty(B, 3, dtype=torch.float32, device=pred.device)
    check(lib.ivlm_contact_prf(gt.data_ptr(), pred.data_ptr(), B, n, float(threshold), out.data_ptr(), _stream()),
          "contact_prf")
    return out


def h_geo_metric(pred, gt, dist):
    """get_h_geo_metric (utils/eval_utils.py:129-151) on the device: pred / gt f32 [B,n], dist f32 [n,n] ->
    (fp_dist_avg, fn_dist_avg) python floats (batch means) and the per-sample f32 [B,2] tensor."""
    lib = _lib.load()
    pred = _req(pred, torch.float32, "pred")
    gt = _req(gt, torch.float32, "gt")
    dist = _req(dist, torch.float32, "dist")
    B, n = pred.shape
    assert gt.shape == pred.shape and dist.shape == (n, n)
    nbytes = lib.ivlm_h_geo_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
    out = torch.empty(B, 2, dtype=torch.float32, device=pred.device)
    check(lib.ivlm_h_geo_metric(dist.data_ptr(), pred.data_ptr(), gt.data_ptr(), B, n, out.data_ptr(), ws.data_ptr(), nbytes,
                                _stream()), "h_geo_metric")
    m = out.mean(0)
    return float(m[0]), float(m[1]), out


class SparseRows:
    """CSR copy of a (mostly empty) dense matrix, e.g. the SMPL->SMPL-X transfer matrix [10475, 6890]."""

    def __init__(self, dense: torch.Tensor, device):
        d = dense.detach().float().cpu()
        self.rows, self.cols = d.shape
        nz = d != 0
        self.row_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), nz.sum(1).cumsum(0)]).to(torch.int32).to(device)
        r, c = nz.nonzero(as_tuple=True)
        self.col = c.to(torch.int32).to(device)
        self.val = d[r, c].contiguous().to(device)

    def matvec(self, x):
        """x f32 [B, cols] -> f32 [B, rows] (convert_contacts, utils/utils.py:428-443)."""
        lib = _lib.load()
        x = _req(x, torch.float32, "x")
        y = torch.empty(x.shape[0], self.rows, dtype=torch.float32, device=x.device)
        check(lib.ivlm_spmv_csr(self.row_ptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr(), x.data_ptr(),
                                x.shape[0], self.rows, self.cols, y.data_ptr(), _stream()), "spmv_csr")
        return y
