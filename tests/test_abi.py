"""The C-ABI library must build for gfx950, load without a GPU, and export every symbol that
include/ivlm_hip.h declares (no compute is launched here)."""
import ctypes
import os

from interactvlm_amd import _lib, build


def test_builds_and_exports_every_declared_symbol():
    import torch  # noqa: F401  (the library binds to torch's HIP runtime)

    path = build.build(verbose=False)
    assert os.path.exists(path)
    protos = _lib.header_prototypes()
    assert len(protos) >= 8
    lib = ctypes.CDLL(path)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, f"declared in ivlm_hip.h but not exported: {missing}"


def test_identity_and_error_strings():
    lib = _lib.load()
    assert lib.ivlm_abi_version() == 5  # (bumped whenever a struct / buffer contract of include/ivlm_hip.h changes: see core.hip)
    assert lib.ivlm_build_arch() == b"gfx950"
    assert lib.ivlm_error_string(0) == b"ok"
    assert b"workspace" in lib.ivlm_error_string(-2)


def test_argument_validation_without_gpu():
    # invalid arguments are rejected before anything touches the device
    lib = _lib.load()
    assert lib.ivlm_lift_mesh_plan(None, None, None, None, 1, 4, 16, 8, 0, 20.0, None, None, None) == -1
    assert lib.ivlm_postprocess_masks(None, 0, 1, 4, 4, 16, 16, 16, 16, 16, 0, None, None) == -1
    assert lib.ivlm_lift_mesh_dense_workspace_bytes(2, 4, 6890) == 2 * 4 * 6890 * 2 * 4 * 128  # one [2][Nv] slab per block: 128 per (image, view)
    assert lib.ivlm_lift_mesh_dense_workspace_bytes(0, 4, 6890) == 0


def test_ops_fail_loudly_on_cpu_tensors():
    import pytest
    import torch

    from interactvlm_amd import ops

    with pytest.raises(_lib.IvlmError):
        ops.postprocess_masks(torch.zeros(1, 1, 8, 8), (32, 32), (32, 32), img_size=32)
