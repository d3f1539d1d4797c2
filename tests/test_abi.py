"""The C-ABI library builds for gfx950, loads on a CPU-only host and exports every
symbol include/ragfin.h declares (no compute calls here)."""
import os
import re

from rag_fin_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ragfin.h")).read()
    text = re.sub(r"#ifdef RF_EXPERIMENTS.*?#endif", "", text, flags=re.S)   # not in the shipped library
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_loads():
    path = build.build_lib()
    assert os.path.exists(path)
    lib = _lib.load_library()
    assert lib.rf_version() >= 100


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ragfin.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound in _lib.py but not declared in ragfin.h"


def test_host_side_argument_checks_need_no_gpu():
    lib = _lib.load_library()
    assert lib.rf_index_storage_bytes(384, 1000) == 32 * 24 * 1024 + 256
    assert lib.rf_index_storage_bytes(383, 1000) == 0      # dim must be a multiple of 16
    assert lib.rf_index_storage_bytes(384, 0) == 0
    assert lib.rf_search_workspace_bytes(None) > 4 * 1024 * 1024
    assert lib.rf_index_size(None) == -1
    # the sharded step's entry points check their arguments before they look for RCCL
    import ctypes
    comm = ctypes.c_void_p()
    assert lib.rf_comm_init(2, 2, ctypes.create_string_buffer(128), 0, ctypes.byref(comm)) == -1   # rank >= world
    assert lib.rf_comm_init(0, 1, None, 0, ctypes.byref(comm)) == -1 and not comm.value
    assert lib.rf_comm_rank(None) == -1 and lib.rf_comm_world(None) == 0 and lib.rf_comm_destroy(None) == 0
    assert lib.rf_search_sharded_scratch_words(None, 64, 10) == 0
    assert lib.rf_search_sharded(None, None, None, 64, 10, 0, None, 0, None, None, None, None, 0, None, 0, None) == -1


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    from rag_fin_amd import store
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        store.GpuIndex(384, 1024)


def test_shipped_library_has_no_tuning_surface_and_carries_its_build_id():
    lib = _lib.load_library()
    for name in _lib.EXPERIMENT_SIGNATURES:
        assert not hasattr(lib, name), f"{name} must only exist in the experiments build"
    assert lib.rf_build_id().decode() == build.source_digest()
    assert lib.rf_version() >= 200


def test_stale_library_is_detected(tmp_path, monkeypatch):
    """A library built from other sources must not load silently: _lib compares rf_build_id()
    with the digest of the sources on disk (and, with hipcc present, rebuilds instead)."""
    import pytest
    monkeypatch.setattr(build, "have_hipcc", lambda: False)
    monkeypatch.setattr(build, "source_digest", lambda experiments=False: "0" * 16)
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib.load_library()
