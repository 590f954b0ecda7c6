"""interactvlm_amd — MI355X (gfx950) implementation of InteractVLM's contact-inference hot path.

Python here is host plumbing that mirrors the reference's operator interfaces
(``model/InteractVLM.py``, ``model/components.py``); the arithmetic lives in hand-written HIP
behind the C ABI of ``include/ivlm_hip.h`` (``libivlm_hip.so``, built by ``interactvlm_amd.build``).
"""
__version__ = "0.1.0"
