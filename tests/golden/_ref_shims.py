"""Import shims that let the *reference* (``/root/reference``, read-only, Python) run in this
container so that golden vectors can be generated from it.  Used ONLY by
``tests/golden/make_golden.py`` in the build container; nothing here (nor the reference) is
needed — or present — on the GPU box.  Recipe follows SURVEY.md §8(c).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("IVLM_REFERENCE_ROOT", "/root/reference")


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(full_model: bool = False):
    """Make ``import model.components`` etc. resolve to the reference.

    full_model=True additionally prepares the LLaVA/InteractVLM facade import
    (transformers-5 registration collisions, MPT stub, ``.cuda()`` no-ops).
    """
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")

    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before any stubbing)
    from transformers import (AutoConfig, AutoModelForCausalLM, CLIPVisionModel,  # noqa: F401
                              LlamaForCausalLM, LlamaModel)

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        ops = _stub("torchvision.ops")
        boxes = _stub("torchvision.ops.boxes", batched_nms=None, box_area=None)
        ops.boxes = boxes
        tv.ops = ops
        tr = _stub("torchvision.transforms")
        trf = _stub("torchvision.transforms.functional", resize=None, to_pil_image=None)
        tr.functional = trf
        tv.transforms = tr
    if "wandb" not in sys.modules:
        _stub("wandb")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    if full_model:
        # llava_llama.py registers model_type "llava", which transformers 5 already owns.
        AutoConfig.register = staticmethod(lambda *a, **k: None)
        AutoModelForCausalLM.register = staticmethod(lambda *a, **k: None)

        class _Dummy:  # the MPT backbone is never instantiated by InteractVLM
            pass

        _stub("model.llava.model.language_model.llava_mpt",
              LlavaMPTConfig=_Dummy, LlavaMPTForCausalLM=_Dummy)
        # the reference hard-codes .cuda(); on this CPU-only box they become no-ops
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.cuda.empty_cache = lambda: None
