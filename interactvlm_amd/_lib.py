"""ctypes loader for libivlm_hip.so (the C-ABI of include/ivlm_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (IVLM_LIB_PATH: load another build of the same library - kernel experiments; the default is the in-tree build)
LIB_PATH = os.environ.get("IVLM_LIB_PATH") or os.path.join(_HERE, "libivlm_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ivlm_hip.h")

_lib = None

_CTYPES = {
    "int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "size_t": C.c_size_t, "float": C.c_float,
    "ivlm_stream_t": C.c_void_p, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "uint8_t": C.c_uint8,
}


class IvlmError(RuntimeError):
    pass


def header_prototypes(path: str = HEADER_PATH):
    """Parse ``include/ivlm_hip.h`` -> {name: (restype_str, [argtype_str, ...])}."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(ivlm_\w+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        argl = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, argl)
    return protos


def _to_ctype(decl: str):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        base = decl.split("*")[0].strip().split()[0]
        return C.c_char_p if base == "char" else C.c_void_p
    base = decl.split()[0]
    return _CTYPES[base]


def load():
    """Load (once) and type-annotate the library. Raises IvlmError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IvlmError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m interactvlm_amd.build` (hipcc --offload-arch=gfx950).")
    import torch  # noqa: F401  -- first: libivlm_hip.so must bind to torch's HIP runtime instance
    lib = C.CDLL(LIB_PATH)
    for name, (ret, args) in header_prototypes().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # header/library drift is a build error, not a soft failure
            # (an experiment library built from an older tree - tools/experiments, timing only - may lack newer symbols: allowed only
            #  with BOTH an explicit library path and IVLM_ALLOW_MISSING_SYMBOLS=1, and never silently)
            if os.environ.get("IVLM_LIB_PATH") and os.environ.get("IVLM_ALLOW_MISSING_SYMBOLS") == "1":
                import warnings

                warnings.warn(f"{LIB_PATH} does not export {name} (declared in ivlm_hip.h): skipped (IVLM_ALLOW_MISSING_SYMBOLS=1)")
                continue
            raise IvlmError(f"libivlm_hip.so does not export {name} declared in ivlm_hip.h") from e
        fn.restype = _to_ctype(ret) if ret != "void" else None
        fn.argtypes = [_to_ctype(a) for a in args]
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        lib = load()
        msg = lib.ivlm_error_string(rc)
        detail = lib.ivlm_last_hip_error() if rc == -3 else b""
        raise IvlmError(f"{what or 'ivlm call'} failed: {msg.decode() if msg else rc} ({rc})"
                        + (f" [{detail.decode()}]" if detail else ""))
