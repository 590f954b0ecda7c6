"""Thin torch-tensor wrappers over the C ABI (``include/ivlm_hip.h``).

torch is plumbing here: device memory, the current HIP stream and nothing else.  Every function
requires CUDA(HIP) tensors and raises if the extension is missing — there is no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import IvlmError, check

IVLM_F32, IVLM_BF16 = 0, 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of individual kernel launches (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream).  Off by default: zero overhead."""

    def __init__(self):
        self.enabled = False
        self.records = {}  # name -> list[(start_event, end_event, work)]

    def start(self):
        self.enabled, self.records = True, {}

    def stop(self):
        self.enabled = False

    def time(self, name, work, fn, tag=None):
        if not self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.records.setdefault(name, []).append((a, b, work, tag))
        return r

    def by_tag(self, name):
        """per-tag (e.g. GEMM shape) totals of one record family: tag -> {launches, total_s, work}."""
        torch.cuda.synchronize()
        out = {}
        for a, b, w, tag in self.records.get(name, []):
            d = out.setdefault(tag, {"launches": 0, "total_s": 0.0, "work": 0.0})
            d["launches"] += 1
            d["total_s"] += a.elapsed_time(b) * 1e-3
            d["work"] += w
        return out

    def event_pair_overhead_ms(self, n=64):
        """Elapsed time an EMPTY event pair reports on the current stream (median).  Reported next to the timings for
        information only: it varies run to run (5-12 us measured) and over-corrects when subtracted - rocprofv3 durations
        (profiles/) are 2.5-6.5 us below the raw event brackets for the 15-25 us kernels."""
        torch.cuda.synchronize()
        pairs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in pairs)
        return v[len(v) // 2]

    def summary(self, subtract_event_overhead=False):
        torch.cuda.synchronize()
        ovh = self.event_pair_overhead_ms() if subtract_event_overhead else 0.0
        out = {}
        for name, rec in self.records.items():
            raw = [a.elapsed_time(b) for a, b, _, _ in rec]
            ms = [max(x - ovh, 0.2 * x) for x in raw]
            out[name] = {"launches": len(rec), "total_s": sum(ms) * 1e-3, "avg_us": sum(ms) / len(ms) * 1e3,
                         "avg_us_events_raw": sum(raw) / len(raw) * 1e3, "event_pair_overhead_us": ovh * 1e3,
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


def postprocess_masks(low_res, input_size, original_size, img_size: int = 1024, apply_sigmoid: bool = False):
    """low_res f32|bf16 [...,h,w] -> f32 [...,oh,ow] (Sam.postprocess_masks)."""
    lib = _lib.load()
    low_res = _req(low_res, None, "low_res")
    lead = tuple(low_res.shape[:-2])
    h, w = low_res.shape[-2:]
    n = 1
    for s in lead:
        n *= s
    oh, ow = int(original_size[0]), int(original_size[1])
    out = torch.empty(lead + (oh, ow), dtype=torch.float32, device=low_res.device)
    check(lib.ivlm_postprocess_masks(low_res.data_ptr(), _dt(low_res), n, h, w, int(img_size), int(input_size[0]),
                                     int(input_size[1]), oh, ow, 1 if apply_sigmoid else 0, out.data_ptr(),
                                     _stream()), "postprocess_masks")
    return out


# --------------------------------------------------------------------------------------------
# dense building blocks
# --------------------------------------------------------------------------------------------
ACT = {"none": 0, "gelu": 1, "quick_gelu": 2, "relu": 3, "silu": 4, "swiglu": 5, "sigmoid": 6}
BF16 = torch.bfloat16


SPLITK = True  # small-M GEMMs (prefill, CLIP): cut K so that >= ~3 tiles per CU are in flight


def _splitk_choice(M, N, K, act, rms):
    """number of K slices (1 = plain kernel): only where the 128x64 tiling leaves most of the 256 CUs idle."""
    if not SPLITK or M <= 8 or M > 1024 or act == "swiglu" or rms is not None or N % 4 or K % 64:
        return 1
    if M <= 16 and N >= 1024 and K >= 1024:
        return 1  # the skinny split-K MFMA kernel (csrc/gemv_mfma.hip) takes these
    tiles = ((M + 127) // 128) * ((N + 63) // 64)
    if tiles >= 256:
        return 1
    best, k64 = 1, K // 64
    for sp in range(2, min(8, 1024 // tiles) + 1):
        if k64 % sp == 0 and K // sp >= 512:
            best = sp
    return best


# opt-in: +12 % on SAM mlp2 in the warm micro-benchmark (244 -> 216 us), but 2 ms SLOWER end to end - three launches and
# 67 MB of fp32 partials per GEMM next to a second stream that wants the same caches
TAILSPLIT = False


def _tail_choice(M, N, K, act, rms):
    """K slices for the under-filled last round of a big GEMM on 256 x 256 tiles (0 = not applicable): the tile count is
    above one round of the 256 CUs, the remainder at most half a round, and K long enough to be worth slicing."""
    if not TAILSPLIT or act == "swiglu" or rms is not None or M < 4096 or N % 4 or K < 2048 or K % 64:
        return 0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    tail = tiles % 256
    if tiles <= 256 or tail == 0 or tail > 128:
        return 0
    k64 = K // 64
    for sp in range(min(256 // tail, 8), 1, -1):
        if k64 % sp == 0:
            return sp
    return 0


def linear(x, weight, bias=None, act="none", residual=None, res_mod=0, out=None, out_f32=False, rms=None):
    """act(x @ weight.T + bias) + residual.  x [..., K] bf16 (last dim contiguous, uniform row stride),
    weight [N, K] bf16."""
    lib = _lib.load()
    K = x.shape[-1]
    N = weight.shape[0]
    assert weight.shape[1] == K and x.dtype == BF16 and weight.dtype == BF16
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    n_out = N // 2 if act == "swiglu" else N
    if out is None:
        out = torch.empty(x.shape[:-1] + (n_out,), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    o2 = out.reshape(-1, n_out)
    assert o2.stride(-1) == 1 and weight.stride(-1) == 1
    r2, ldr = None, 0
    if residual is not None:
        r2 = residual.reshape(-1, N)
        assert r2.dtype == BF16 and r2.stride(-1) == 1
        ldr = r2.stride(0)
    if bias is not None:
        assert bias.dtype == BF16 and bias.is_contiguous()
    call = lambda: check(lib.ivlm_gemm_bf16(
        x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0), o2.data_ptr(), o2.stride(0), _p(bias),
        _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out.dtype == torch.float32 else 0, 1, 0, 0, 0, 0,
        _p(rms[0]) if rms else 0, float(rms[1]) if rms else 0.0, _stream()), "gemm_bf16")
    splits = _splitk_choice(M, N, K, act, rms)
    if splits > 1 and o2.stride(0) % 4 == 0:
        ws = torch.empty(splits * M * N, dtype=torch.float32, device=x.device)  # caching allocator: stream-safe
        call = lambda: check(lib.ivlm_gemm_bf16_splitk(
            x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0), o2.data_ptr(), o2.stride(0), _p(bias),
            _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out_f32 else 0, splits, ws.data_ptr(),
            ws.numel() * 4, _stream()), "gemm_bf16_splitk")
    tsp = _tail_choice(M, N, K, act, rms) if splits <= 1 else 0
    if tsp > 1 and o2.stride(0) % 4 == 0:
        ws = torch.empty(tsp * M * N, dtype=torch.float32, device=x.device)  # only the tail tiles' region is touched
        call = lambda: check(lib.ivlm_gemm_bf16_tailsplit(
            x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0), o2.data_ptr(), o2.stride(0), _p(bias),
            _p(r2), ldr, int(res_mod), M, N, K, ACT[act], 1 if out_f32 else 0, tsp, ws.data_ptr(),
            ws.numel() * 4, _stream()), "gemm_bf16_tailsplit")
    if TIMER.enabled:  # work = algorithmic FLOPs (MFMA path) or weight bytes (GEMV path)
        if M > 8:
            TIMER.time("gemm_bf16_mfma", 2.0 * M * N * K, call, tag=(M, N, K, act))
        else:
            TIMER.time("gemv_bf16", 2.0 * N * K, call, tag=(M, N, K, act))
    else:
        call()
    return out


def layernorm(x, weight, bias, eps=1e-5, gelu=False, out=None):
    lib = _lib.load()
    x = _req(x, BF16, "x")
    y = torch.empty_like(x) if out is None else out
    cols = x.shape[-1]
    check(lib.ivlm_layernorm_bf16(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), x.numel() // cols,
                                  cols, float(eps), 1 if gelu else 0, _stream()), "layernorm")
    return y


def rmsnorm(x, weight, eps=1e-5):
    lib = _lib.load()
    x = _req(x, BF16, "x")
    y = torch.empty_like(x)
    cols = x.shape[-1]
    check(lib.ivlm_rmsnorm_bf16(x.data_ptr(), weight.data_ptr(), y.data_ptr(), x.numel() // cols, cols, float(eps),
                                _stream()), "rmsnorm")
    return y


def attention(q, k, v, scale, causal=False, q_pos0=0, rel=None, out=None, prescale_q=False):
    """q [B,H,Sq,D], k/v [Bk,H,Sk,D] (arbitrary strides, last dim contiguous; B % Bk == 0: K/V of batch
    b // (B//Bk)) -> o [B,H,Sq,D] as a view of a [B,Sq,H,D] buffer (so o.transpose(1,2) is contiguous).
    rel = (rel_h f32 [B*H,Sq,KH], rel_w f32 [B*H,Sq,KW]) adds SAM's decomposed rel-pos bias."""
    import ctypes

    lib = _lib.load()
    B, H, Sq, D = q.shape
    Bk, Sk = k.shape[0], k.shape[2]
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and B % Bk == 0
    if out is None:
        out = torch.empty(B, Sq, H, D, dtype=BF16, device=q.device).permute(0, 2, 1, 3)
    assert out.stride(3) == 1
    st = (ctypes.c_int64 * 12)(q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                               v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2))
    rel_h = rel_w = None
    kh = kw = 0
    if rel is not None:
        rel_h, rel_w = rel
        assert rel_h.dtype == torch.float32 and rel_h.is_contiguous() and rel_w.is_contiguous()
        kh, kw = rel_h.shape[-1], rel_w.shape[-1]
    check(lib.ivlm_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                  ctypes.cast(st, ctypes.c_void_p), B, H, Sq, Sk, D, float(scale), 1 if causal else 0,
                                  int(q_pos0), _p(rel_h), _p(rel_w), kh, kw, B // Bk,
                                  1 if (prescale_q or rel is not None) else 0, _stream()), "attention")
    return out


def relpos_tables_cat(tab_h, tab_w):
    """[rel_pos_h ; rel_pos_w] padded with zero rows to a multiple of 8: the weight matrix of the GEMM formulation."""
    n = tab_h.shape[0] + tab_w.shape[0]
    cat = torch.zeros((n + 7) // 8 * 8, tab_h.shape[1], dtype=BF16, device=tab_h.device)
    cat[: tab_h.shape[0]] = tab_h
    cat[tab_h.shape[0]: n] = tab_w
    return cat


RELPOS_GEMM = True  # rel-pos operands through one batched MFMA GEMM + gather instead of the VALU dot-product kernel


def relpos_bias(q, tab_h, tab_w, SH, SW, cat=None):
    """q [B,H,S=SH*SW,D] -> (rel_h f32 [B*H,S,SH], rel_w f32 [B*H,S,SW]).  cat = relpos_tables_cat(tab_h, tab_w)
    (pre-built once per block) selects the GEMM formulation when the q rows of all (b, s) are uniformly strided."""
    lib = _lib.load()
    B, H, S, D = q.shape
    assert S == SH * SW and q.stride(3) == 1 and tab_h.is_contiguous() and tab_w.is_contiguous()
    rel_h = torch.empty(B * H, S, SH, dtype=torch.float32, device=q.device)
    rel_w = torch.empty(B * H, S, SW, dtype=torch.float32, device=q.device)
    # (measured, SAM ViT-H: global 64x64 grid 242 -> 139 us; 14x14 windows 54 -> 62 us - the GEMM's N = 54 wastes half a tile and
    #  the gather moves as many bytes as the dot kernel writes - so only grids of 32x32 and up take this path)
    if (RELPOS_GEMM and cat is not None and min(SH, SW) >= 32 and q.dtype == BF16 and q.stride(0) == S * q.stride(2) and q.stride(2) % 8 == 0
            and q.stride(1) % 8 == 0 and D % 8 == 0 and q.data_ptr() % 16 == 0):
        npad, M = cat.shape[0], B * S
        G = torch.empty(H, M, npad, dtype=BF16, device=q.device)
        call = lambda: check(lib.ivlm_gemm_bf16(q.data_ptr(), q.stride(2), cat.data_ptr(), D, G.data_ptr(), npad, 0, 0, 0, 0,
                                                M, npad, D, 0, 0, H, q.stride(1), 0, M * npad, 0, 0, 0.0, _stream()),
                             "relpos gemm")
        if TIMER.enabled:
            TIMER.time("gemm_bf16_mfma", 2.0 * H * M * npad * D, call, tag=("relpos", M, npad, D))
        else:
            call()
        check(lib.ivlm_relpos_gather(G.data_ptr(), M * npad, npad, B, H, SH, SW, rel_h.data_ptr(), rel_w.data_ptr(),
                                     _stream()), "relpos_gather")
        return rel_h, rel_w
    check(lib.ivlm_relpos_bias(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), tab_h.data_ptr(), tab_w.data_ptr(),
                               B, H, SH, SW, D, rel_h.data_ptr(), rel_w.data_ptr(), _stream()), "relpos_bias")
    return rel_h, rel_w


def argmax(logits):
    lib = _lib.load()
    logits = _req(logits, torch.float32, "logits")
    rows, cols = logits.shape
    out = torch.empty(rows, dtype=torch.int32, device=logits.device)
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


def gather_rows(src, idx, add=None, out=None):
    """out[r] = src[idx[r]] (zeros where idx < 0) (+ add[r]); src [R,C] / add [n,C] rows may be strided."""
    lib = _lib.load()
    assert src.dtype == BF16 and src.stride(-1) == 1 and idx.dtype == torch.int32 and idx.is_contiguous()
    rows, cols = idx.numel(), src.shape[-1]
    if out is None:
        out = torch.empty(rows, cols, dtype=BF16, device=src.device)
    assert out.stride(-1) == 1
    lda = 0
    if add is not None:
        assert add.dtype == BF16 and add.stride(-1) == 1
        lda = add.stride(0)
    check(lib.ivlm_gather_rows(out.data_ptr(), out.stride(0), src.data_ptr(), src.stride(0), idx.data_ptr(), _p(add),
                               lda, rows, cols, _stream()), "gather_rows")
    return out


def add_rows(a, b, out=None, op="add"):
    """a [R,C] (+|*) b [r,C] broadcast with row modulo (R % r == 0 not required)."""
    lib = _lib.load()
    a = _req(a, BF16, "a")
    b = _req(b, BF16, "b")
    cols = a.shape[-1]
    out = torch.empty_like(a) if out is None else out
    check(lib.ivlm_add_rows(out.data_ptr(), a.data_ptr(), b.data_ptr(), a.numel() // cols, cols, b.numel() // cols,
                            1 if op == "mul" else 0, _stream()), "add_rows")
    return out


def dense_pe(gauss, h, w):
    lib = _lib.load()
    gauss = _req(gauss, torch.float32, "gauss")
    F = gauss.shape[1]
    pe = torch.empty(h * w, 2 * F, dtype=BF16, device=gauss.device)
    check(lib.ivlm_dense_pe(gauss.data_ptr(), pe.data_ptr(), h, w, F, _stream()), "dense_pe")
    return pe


def rope_kv(qkv, H, D, pos0, theta, kcache=None, vcache=None, table=None):
    """qkv [T, 3*H*D] (in place). table = (cos, sin) fp32 [Tmax, D/2] from rope_table()."""
    lib = _lib.load()
    assert qkv.dtype == BF16 and qkv.stride(-1) == 1
    T = qkv.shape[0]
    check(lib.ivlm_rope_kv(qkv.data_ptr(), qkv.stride(0), T, H, D, int(pos0), float(theta), _p(kcache), _p(vcache),
                           _p(table[0]) if table else 0, _p(table[1]) if table else 0, _stream()), "rope_kv")
    return qkv


def rope_table(T, D, theta, device):
    lib = _lib.load()
    c = torch.empty(T, D // 2, dtype=torch.float32, device=device)
    s = torch.empty_like(c)
    check(lib.ivlm_rope_table(c.data_ptr(), s.data_ptr(), T, D, float(theta), _stream()), "rope_table")
    return c, s


def mask_dot(up, hyper, B, gh, gw):
    lib = _lib.load()
    up = _req(up, BF16, "up")
    hyper = _req(hyper, BF16, "hyper")
    C = hyper.shape[-1]
    low = torch.empty(B, 4 * gh, 4 * gw, dtype=torch.float32, device=up.device)
    check(lib.ivlm_mask_dot(up.data_ptr(), hyper.data_ptr(), low.data_ptr(), B, gh, gw, C, _stream()), "mask_dot")
    return low


def llama_decode_attn(qkv, kcache, vcache, H, D, pos, theta, scale, out=None, table=None):
    """qkv bf16 [1, 3*H*D] of the newest token -> o bf16 [1, H*D]; RoPE + cache append fused.
    pos: python int, or an int32 device tensor [1] (read by the kernel: HIP-graph friendly)."""
    lib = _lib.load()
    assert qkv.dtype == BF16 and qkv.is_contiguous() and kcache.is_contiguous() and vcache.is_contiguous()
    if out is None:
        out = torch.empty(1, H * D, dtype=BF16, device=qkv.device)
    if isinstance(pos, torch.Tensor):
        assert pos.dtype == torch.int32 and pos.is_cuda
        check(lib.ivlm_llama_decode_attn_devpos(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), H, D,
                                                pos.data_ptr(), float(theta), float(scale),
                                                _p(table[0]) if table else 0, _p(table[1]) if table else 0, _stream()),
              "llama_decode_attn_devpos")
        return out
    check(lib.ivlm_llama_decode_attn(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), H, D,
                                     int(pos), float(theta), float(scale), _p(table[0]) if table else 0,
                                     _p(table[1]) if table else 0, _stream()), "llama_decode_attn")
    return out


def llama_decode_attn_batch(qkv, kcache, vcache, H, D, pos_dev, theta, scale, table=None, out=None):
    """One decode step of B sequences: qkv bf16 [B, 3*H*D], kcache/vcache bf16 [B, Tmax, H, D] (one slab per sequence),
    pos_dev int32 [B] on the device -> o bf16 [B, H*D]."""
    lib = _lib.load()
    B = qkv.shape[0]
    assert qkv.dtype == BF16 and qkv.stride(1) == 1 and kcache.dim() == 4 and kcache.shape[0] == B
    assert kcache[0].is_contiguous() and vcache[0].is_contiguous() and kcache.stride(0) == vcache.stride(0)
    assert pos_dev.dtype == torch.int32 and pos_dev.is_cuda and pos_dev.numel() == B and pos_dev.is_contiguous()
    if out is None:
        out = torch.empty(B, H * D, dtype=BF16, device=qkv.device)
    check(lib.ivlm_llama_decode_attn_batch(qkv.data_ptr(), qkv.stride(0), kcache.data_ptr(), vcache.data_ptr(),
                                           kcache.stride(0), out.data_ptr(), out.stride(0), B, H, D, pos_dev.data_ptr(),
                                           float(theta), float(scale), _p(table[0]) if table else 0,
                                           _p(table[1]) if table else 0, _stream()), "llama_decode_attn_batch")
    return out


def llama_attn_oproj(qkv, kcache, vcache, wo, x, H, D, pos_dev, step_dev, counter, status, theta, scale, table, scratch):
    """One launch: single-token attention of every head + o_proj GEMV + residual -> x_out bf16 [1, H*D]."""
    lib = _lib.load()
    out = torch.empty(1, H * D, dtype=BF16, device=qkv.device)
    check(lib.ivlm_llama_attn_oproj(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), scratch.data_ptr(), wo.data_ptr(),
                                    x.data_ptr(), out.data_ptr(), H, D, float(theta), float(scale), table[0].data_ptr(),
                                    table[1].data_ptr(), pos_dev.data_ptr(), step_dev.data_ptr(), counter.data_ptr(),
                                    status.data_ptr(), _stream()), "llama_attn_oproj")
    return out


def llama_gateup_down(x2, ln_w, eps, wgu, wdown, step_dev, counter, status, scratch):
    """One launch: x_out = x2 + W_down . SwiGLU(W_gu . RMSNorm(x2)) for one decode token, bf16 [1, hidden]."""
    lib = _lib.load()
    hidden, inter = wdown.shape[0], wdown.shape[1]
    out = torch.empty(1, hidden, dtype=BF16, device=x2.device)
    check(lib.ivlm_llama_gateup_down(x2.data_ptr(), ln_w.data_ptr(), float(eps), wgu.data_ptr(), wdown.data_ptr(),
                                     scratch.data_ptr(), out.data_ptr(), hidden, inter, step_dev.data_ptr(), counter.data_ptr(),
                                     status.data_ptr(), _stream()), "llama_gateup_down")
    return out


def llama_decode_layers(layer_ptrs, L, H, D, hidden, inter, eps, theta, rope, kcache, vcache, x0, pos_dev, step_dev, ws):
    """All decoder layers of one token in one dataflow launch -> residual stream after the last layer, bf16 [1, hidden]."""
    lib = _lib.load()
    out = torch.empty(1, hidden, dtype=BF16, device=x0.device)
    check(lib.ivlm_llama_decode_layers(layer_ptrs.data_ptr(), L, H, D, hidden, inter, float(eps), float(theta), float(D) ** -0.5,
                                       rope[0].data_ptr(), rope[1].data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                       kcache.stride(0), x0.data_ptr(), out.data_ptr(), pos_dev.data_ptr(), step_dev.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _stream()), "llama_decode_layers")
    return out


def llama_generate(layer_ptrs, L, H, D, hidden, inter, vocab, eps, scale, rope, kcache, vcache, max_len, embed, final_norm,
                   lm_head, hidden_out, pos0, n_max, eos, forced=None):
    """Whole greedy generation after the prefill in one persistent launch (ivlm_llama_generate).
    Returns (new_ids i32 [n_max], argmax_ids i32 [n_max], status i32 [2]) device tensors; status = (n generated, error)."""
    lib = _lib.load()
    dev = hidden_out.device
    new_ids = torch.zeros(n_max, dtype=torch.int32, device=dev)
    arg_ids = torch.zeros(n_max, dtype=torch.int32, device=dev)
    nbytes = lib.ivlm_llama_generate_workspace_bytes(hidden, inter)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    assert kcache.is_contiguous() and vcache.is_contiguous() and hidden_out.is_contiguous()
    if forced is not None:
        assert forced.dtype == torch.int32 and forced.numel() >= n_max
    call = lambda: check(lib.ivlm_llama_generate(
        layer_ptrs.data_ptr(), L, H, D, hidden, inter, vocab, float(eps), float(scale), rope[0].data_ptr(),
        rope[1].data_ptr(), kcache.data_ptr(), vcache.data_ptr(), kcache.stride(0), int(max_len), embed.data_ptr(),
        final_norm.data_ptr(), lm_head.data_ptr(), hidden_out.data_ptr(), int(pos0), int(n_max), int(eos),
        _p(forced), new_ids.data_ptr(), arg_ids.data_ptr(), ws.data_ptr(), nbytes, _stream()), "llama_generate")
    if TIMER.enabled:  # work = upper bound of the weight bytes streamed (n_max tokens; fewer if EOS comes early)
        per_tok = 2.0 * (L * (4.0 * hidden * hidden + 3.0 * hidden * inter) + float(vocab) * hidden)
        TIMER.time("llama_generate", per_tok * n_max - 2.0 * L * (4.0 * hidden * hidden + 3.0 * hidden * inter), call)
    else:
        call()
    llama_generate.last_workspace = ws  # debugging (IVLM_GEN_TRACE=1: timestamps behind the scratch vectors)
    return new_ids, arg_ids, ws[:8].view(torch.int32)


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
