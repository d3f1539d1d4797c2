"""Build libragfin_hip.so (gfx950) in-tree with hipcc.

`python -m rag_fin_amd.build` or `rag_fin_amd.build.build_lib()`.  hipcc
cross-compiles without a GPU, so this runs in the CPU-only container; the built
.so travels to the GPU box with the source snapshot.

Two flavours share the sources:
  libragfin_hip.so      the product: no run-time tuning surface (knobs are compile-time
                        constants), only the kernels the product dispatches to
  libragfin_hip_exp.so  `--experiments` (-DRF_EXPERIMENTS): adds rf_set_tuning, the diagnostic
                        hooks and the ablation / alternative kernel instantiations that
                        tools/ A/B against each other; loaded through RAGFIN_LIB by tools only

The library carries a digest of its sources (rf_build_id); _lib.load_library() compares it
with source_digest() so that a stale .so is rebuilt or refused, never silently loaded.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libragfin_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
EXP_LIB_PATH = os.path.join(CSRC, "libragfin_hip_exp.so")
SOURCES = ["index.hip", "scan.hip", "scan_wide.hip", "merge.hip", "api.hip", "comm.hip", "encoder.hip", "encoder_post.hip", "tokenizer.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build " + LIB_NAME)
    return exe


def have_hipcc() -> bool:
    return os.path.exists(shutil.which("hipcc") or "/opt/rocm/bin/hipcc")


def _headers() -> list[str]:
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hs.append(os.path.normpath(os.path.join(HERE, "..", "include", "ragfin.h")))
    return hs


def source_digest(experiments: bool = False) -> str:
    """sha256 over the compile flags and every source / header the library is built from."""
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + (" -DRF_EXPERIMENTS" if experiments else "")).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES + ["build_id.cpp"]] + _headers():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _object_digest(src: str, headers: list[str], flags: list[str]) -> str:
    """What an object file was compiled from: the flags, the source and every header (content, not mtime: a
    tree synced with preserved mtimes, or an edit of FLAGS, must recompile too)."""
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for path in [src] + headers:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_lib(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    # one builder at a time per tree: under torch.distributed.run every rank that finds a stale library would
    # otherwise write the same .o / .so files concurrently
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_lib_locked(force, verbose, experiments)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_lib_locked(force: bool, verbose: bool, experiments: bool) -> str:
    hipcc = _hipcc()
    headers = _headers()
    suffix = ".exp.o" if experiments else ".o"
    flags = FLAGS + (["-DRF_EXPERIMENTS"] if experiments else [])
    out = EXP_LIB_PATH if experiments else LIB_PATH
    if not force and os.path.exists(out) and built_digest(out) == source_digest(experiments):
        return out   # another process built it while this one waited for the lock
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = [os.path.splitext(s)[0] + suffix for s in srcs]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd[:1] + cmd[-3:])} failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    rebuilt = []

    def compile_one(pair):
        src, obj = pair
        want = _object_digest(src, headers, flags)
        side = obj + ".id"
        have = open(side).read().strip() if os.path.exists(side) else ""
        if force or not os.path.exists(obj) or have != want:
            run([hipcc, *flags, "-c", src, "-o", obj])
            with open(side, "w") as f:
                f.write(want)
            rebuilt.append(obj)

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    # the digest TU: recompiled whenever the digest it carries differs from the sources'
    digest = source_digest(experiments)
    id_obj = os.path.join(CSRC, "build_id" + suffix)
    id_txt = id_obj + ".id"
    have = open(id_txt).read().strip() if os.path.exists(id_txt) else ""
    relink = force or bool(rebuilt) or _stale(out, objs)
    if have != digest or not os.path.exists(id_obj):
        run([hipcc, *flags, f'-DRF_BUILD_ID="{digest}"', "-c", os.path.join(CSRC, "build_id.cpp"), "-o", id_obj])
        with open(id_txt, "w") as f:
            f.write(digest)
        relink = True
    if relink:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, id_obj, "-ldl"])
    with open(out + ".id", "w") as f:   # sidecar: lets the loader decide without dlopen-ing a stale library
        f.write(digest)
    return out


def built_digest(path: str) -> str:
    try:
        with open(path + ".id") as f:
            return f.read().strip()
    except OSError:
        return ""


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
