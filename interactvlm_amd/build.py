"""Build libivlm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m interactvlm_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libivlm_hip.so")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# per-file extra flags (lift/postprocess follow the reference's mul-then-add arithmetic)
EXTRA = {
    "lift.hip": ["-ffp-contract=off"],
    "postprocess.hip": ["-ffp-contract=off"],
    "heads.hip": ["-ffp-contract=off"],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _torch_lib_dir() -> str:
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        raise RuntimeError("torch not found: libivlm_hip.so links against torch's HIP runtime")
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    if not os.path.exists(os.path.join(d, "libamdhip64.so")):
        raise RuntimeError(f"{d}/libamdhip64.so missing (not a ROCm build of torch?)")
    return d


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hpp")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src: str, force: bool, hdr_m: float) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(sp)
            and os.path.getmtime(obj) >= hdr_m):
        return obj
    cmd = [_hipcc(), *COMMON, *EXTRA.get(src, []), "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    hdr_m = _deps_mtime()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        # Link against the SAME HIP runtime torch uses (its bundled libamdhip64.so), not hipcc's
        # default /opt/rocm copy: two runtimes in one process do not share streams/events/contexts.
        tl = _torch_lib_dir()
        cmd = ["g++", "-shared", "-fPIC", "-o", LIB, *objs, "-L" + tl, "-lamdhip64",
               "-Wl,-rpath," + tl, "-Wl,--no-undefined", "-Wl,-z,defs"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[ivlm] built {LIB} from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
