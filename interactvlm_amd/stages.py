"""ctypes mirror of the stage-level C entry points (``ivlm_llama_prefill`` / ``ivlm_llama_decode_step``, include/ivlm_hip.h):
what a non-Python caller binds.  ``interactvlm_amd.llava.Llama`` sequences the same kernels from Python; this module exists to
exercise the C sequencers (tests/test_stages_gpu.py) and as the binding example of INTEGRATION.md."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


class LlamaCfg(C.Structure):
    _fields_ = [("layers", C.c_int), ("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("max_len", C.c_int),
                ("eps", C.c_float), ("theta", C.c_float)]


class LlamaLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1", "qkv", "o", "ln2", "gu", "down")]


class LlamaStages:
    """Wraps a ``llava.Llama`` instance's weights / caches / rope tables as the C structs and calls the C sequencers."""

    def __init__(self, llm):
        self.llm = llm
        c = llm.cfg
        self.cfg = LlamaCfg(c.layers, c.hidden, c.heads, c.inter, llm.max_len, c.eps, c.theta)
        self.layers = (LlamaLayer * c.layers)(*[LlamaLayer(*[L[k].data_ptr() for k in ("ln1", "qkv", "o", "ln2", "gu", "down")])
                                                for L in llm.layers])
        lib = _lib.load()
        self._dws = torch.zeros(lib.ivlm_llama_decode_workspace_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=llm.device)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def prefill(self, x, pos0=0):
        """x fp32 [T, hidden] -> final-norm hidden fp32 [T, hidden]; appends to the instance's KV cache."""
        lib = _lib.load()
        llm = self.llm
        T = x.shape[0]
        x = x.contiguous()
        out = torch.empty_like(x)
        nbytes = lib.ivlm_llama_prefill_workspace_bytes(C.byref(self.cfg), T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib.ivlm_llama_prefill(C.byref(self.cfg), self.layers, llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                     llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(), T,
                                     int(pos0), out.data_ptr(), ws.data_ptr(), nbytes, self._stream()), "llama_prefill")
        return out

    def start_generation(self):
        self._dws.zero_()

    def decode_step(self, x, pos_dev, advance=True):
        """x fp32 [1, hidden], pos_dev int32 [1] on the device -> hidden fp32 [1, hidden]."""
        lib = _lib.load()
        llm = self.llm
        out = torch.empty_like(x)
        check(lib.ivlm_llama_decode_step(C.byref(self.cfg), self.layers, llm.norm.data_ptr(), llm.kcache.data_ptr(),
                                         llm.vcache.data_ptr(), llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), x.data_ptr(),
                                         pos_dev.data_ptr(), 1 if advance else 0, out.data_ptr(), self._dws.data_ptr(),
                                         self._dws.numel(), self._stream()), "llama_decode_step")
        return out
