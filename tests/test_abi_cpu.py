"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, the ctypes table covers exactly those, host-only entry points work, and creating an
engine without a GPU fails loudly (no CPU fallback)."""
import os
import re
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "voitta_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from voitta_rag_amd import _lib

    lib = _lib.load_library()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/voitta_engine.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    assert lib.vr_abi_version() == 1


def test_struct_layouts_match_header(tmp_path):
    """sizeof of every ABI struct, as gcc sees the header, equals the ctypes mirror."""
    import ctypes as C
    import subprocess

    from voitta_rag_amd import _lib

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "voitta_engine.h"\nint main(void){printf("%zu %zu %zu\\n",'
                   'sizeof(vr_config),sizeof(vr_bert_desc),sizeof(vr_filter));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(_lib.VrConfig), C.sizeof(_lib.VrBertDesc), C.sizeof(_lib.VrFilter)]


def test_engine_creation_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from voitta_rag_amd import Engine, EngineError

    with pytest.raises(EngineError, match="no CPU fallback|no HIP device|hipGetDeviceCount"):
        Engine(64)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from voitta_rag_amd import _lib

    monkeypatch.setenv("VOITTA_ENGINE_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.EngineError, match="not found"):
        _lib.load_library()
    monkeypatch.delenv("VOITTA_ENGINE_LIB")
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load_library()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "voitta_rag_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle_core" not in src, f


def test_install_swaps_reference_singletons():
    for name in ("voitta", "voitta.services"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from voitta_rag_amd import install, vector_store

    install.install()
    assert sys.modules["voitta.services.vector_store"].VectorStoreService is vector_store.VectorStoreService
    assert sys.modules["voitta.services.sparse_embedding"].SPARSE_VECTOR_NAME == "bm25"
    assert callable(sys.modules["voitta.services.embedding"].get_embedding_service)
    for name in list(sys.modules):
        if name == "voitta" or name.startswith("voitta."):
            del sys.modules[name]


def test_sparse_query_service_is_host_only():
    from oracle import bm25 as obm
    from voitta_rag_amd.sparse_embedding import SparseEmbeddingService

    s = SparseEmbeddingService()
    assert s.embed_query("the of and") == ([], [])
    ids, vals = s.embed_query("Quick foxes, quick!")
    assert (ids, vals) == obm.query_embed("Quick foxes, quick!")
    assert s.embed_texts([]) == []
