"""ctypes binding of libragfin_hip.so (include/ragfin.h).

There is no CPU fallback: if the library cannot be loaded, or a call returns a
non-zero status, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint32, c_void_p)

from . import build as _build

RF_OK = 0
RF_MAX_K = 64
RF_QCHUNK = 64
RF_FLAG_CAND_OVERFLOW = 1
RF_FLAG_TIE_OVERFLOW = 2


class RagfinError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libragfin_hip error {code}: {msg}")
        self.code = code


class EncoderConfig(Structure):
    _fields_ = [("vocab_size", c_int32), ("hidden", c_int32), ("layers", c_int32),
                ("heads", c_int32), ("intermediate", c_int32), ("max_position", c_int32),
                ("type_vocab", c_int32), ("ln_eps", c_float)]


ENCODER_WEIGHT_FIELDS = ["word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "qkv_w",
                         "qkv_b", "ao_w", "ao_b", "ln1_g", "ln1_b", "ff1_w", "ff1_b", "ff2_w",
                         "ff2_b", "ln2_g", "ln2_b"]


class EncoderWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ENCODER_WEIGHT_FIELDS]


# name -> (restype, argtypes); every symbol include/ragfin.h declares
SIGNATURES = {
    "rf_version": (c_int, []),
    "rf_build_id": (c_char_p, []),
    "rf_last_error": (c_char_p, []),
    "rf_device_check": (c_int, [c_int]),
    "rf_index_storage_bytes": (c_size_t, [c_int, c_int64]),
    "rf_index_create": (c_int, [POINTER(c_void_p), c_int, c_int64, c_void_p, c_size_t, c_int]),
    "rf_index_destroy": (c_int, [c_void_p]),
    "rf_index_add_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rf_index_size": (c_int64, [c_void_p]),
    "rf_index_dim": (c_int, [c_void_p]),
    "rf_index_reset": (c_int, [c_void_p, c_void_p]),
    "rf_index_get_rows_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "rf_normalize_f32_to_f16": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "rf_search_workspace_bytes": (c_size_t, [c_void_p]),
    "rf_search": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "rf_search_profile": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                  POINTER(c_float)]),
    "rf_search_exhaustive": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "rf_search_exhaustive_after": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rf_merge_shards": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p]),
    "rf_packed_shard_words": (c_size_t, [c_int, c_int]),
    "rf_merge_shards_packed": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rf_map_ids": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "rf_comm_unique_id": (c_int, [c_void_p]),
    "rf_comm_init": (c_int, [c_int, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    "rf_comm_destroy": (c_int, [c_void_p]),
    "rf_comm_rank": (c_int, [c_void_p]),
    "rf_comm_world": (c_int, [c_void_p]),
    "rf_search_sharded_scratch_words": (c_size_t, [c_void_p, c_int, c_int]),
    "rf_search_sharded": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "rf_tokenizer_create": (c_int, [POINTER(c_void_p), c_char_p, c_size_t, c_int, c_int]),
    "rf_tokenizer_destroy": (c_int, [c_void_p]),
    "rf_tokenizer_set_punctuation": (c_int, [c_void_p, c_void_p, c_int]),
    "rf_tokenizer_special_ids": (c_int, [c_void_p, c_void_p]),
    "rf_tokenize_batch": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
    "rf_utf8_offsets": (c_int, [c_char_p, c_int64, c_void_p, c_int, c_void_p]),
    "rf_debug_scores": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "rf_encoder_storage_bytes": (c_size_t, [POINTER(EncoderConfig)]),
    "rf_encoder_create": (c_int, [POINTER(c_void_p), POINTER(EncoderConfig),
                                  POINTER(EncoderWeights), c_void_p, c_size_t, c_int, c_void_p]),
    "rf_encoder_destroy": (c_int, [c_void_p]),
    "rf_encode_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "rf_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                          c_size_t, c_void_p]),
}

# present only in the experiments build (libragfin_hip_exp.so, tools/): bound when exported
EXPERIMENT_SIGNATURES = {
    "rf_set_tuning": (c_int, [c_char_p, c_int]),
    "rf_debug_workspace_offset": (c_size_t, [c_char_p]),
    "rf_debug_set_buffer": (c_int, [c_void_p]),
}

_lib = None


def library_path() -> str:
    # RAGFIN_LIB=exp: the experiments build (tools/); any other value: an explicit path
    v = os.environ.get("RAGFIN_LIB")
    if v == "exp":
        return _build.EXP_LIB_PATH
    return v or _build.LIB_PATH


def load_library() -> ctypes.CDLL:
    """Load the library, making sure it was built from the sources on disk: a missing or stale
    .so is rebuilt when hipcc is present and refused when it is not (no silent old kernels)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    in_tree = path in (_build.LIB_PATH, _build.EXP_LIB_PATH)
    exp = path == _build.EXP_LIB_PATH
    if in_tree and (not os.path.exists(path) or _build.built_digest(path) != _build.source_digest(exp)):
        # missing, or built from other sources than those on disk: rebuild where hipcc exists, refuse
        # where it does not -- no fallback.  A current library is loaded as it is: no compiler runs at import.
        if not _build.have_hipcc():
            raise RuntimeError(f"{path} is missing or was built from other sources (build id "
                               f"{_build.built_digest(path) or 'none'}, sources {_build.source_digest(exp)}) and hipcc is "
                               "not available to rebuild it; run `python -m rag_fin_amd.build` where the ROCm toolchain is")
        _build.build_lib(experiments=exp)
    elif not os.path.exists(path):
        raise RuntimeError(f"{path} does not exist")
    # torch ships its own libamdhip64; load it FIRST so that this library binds to the
    # same HIP runtime instance (loaded the other way round, the two runtimes disagree
    # about device visibility: rf_device_check saw "no HIP device" on a GPU box)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if in_tree:
        have = (lib.rf_build_id() or b"").decode()
        want = _build.source_digest(exp)
        if have != want:
            raise RuntimeError(f"{path} was built from other sources (build id {have}, sources {want}) and "
                               "hipcc is not available to rebuild it; run `python -m rag_fin_amd.build` "
                               "where the ROCm toolchain is")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != RF_OK:
        msg = load_library().rf_last_error()
        raise RagfinError(code, msg.decode() if msg else "")


def current_stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
