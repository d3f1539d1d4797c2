"""Build libragfin_hip.so (gfx950) in-tree with hipcc.

`python -m rag_fin_amd.build` or `rag_fin_amd.build.build_lib()`.  hipcc
cross-compiles without a GPU, so this runs in the CPU-only container; the built
.so travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libragfin_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ["index.hip", "scan.hip", "scan_fused.hip", "scan_wide.hip", "merge.hip", "api.hip", "encoder.hip",
           "tokenizer.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build " + LIB_NAME)
    return exe


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "ragfin.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [os.path.splitext(s)[0] + ".o" for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and not _stale(obj, [src] + headers):
            return
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
