"""CPU oracle for the InteractVLM contact-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
call, link or execute it — and there only as the checker, never as the thing measured or
shipped.  ``interactvlm_amd`` never imports this package.

Pinning status:
  * lift (K14/K15/K16), SAM prompt-encoder / mask-decoder / postprocess, SAM ViT blocks,
    cam-pose encoders, [SEG] selection and the model_forward(inference=True) wiring (an hcontact sample and an
    oafford sample with the object predictors enabled) are PINNED against golden vectors produced by importing
    the reference itself
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * LLaMA / CLIP arithmetic lives in un-vendored ``transformers==4.31.0`` (absent here).
    The stand-in used to produce goldens is transformers 5.15 (same published architecture);
    against the pinned 4.31 wheel itself this part is "parity unpinned".
  * pytorch3d rasteriser (``pytorch3d@stable``, absent): restated from its published
    conventions, self-consistency tests only -> "parity unpinned".
"""
