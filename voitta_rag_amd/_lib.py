"""ctypes binding of libvoitta_engine.so — the exact stub a maintainer of the reference would
add. Fails loudly when the library has not been built; nothing here falls
back to a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libvoitta_engine.so"
_lib = None


class EngineError(RuntimeError):
    """Raised for every non-zero return of the C-ABI (message from vr_last_error())."""


class VrConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("device", C.c_int32),
        ("dim", C.c_int32),
        ("flags", C.c_int32),
        ("initial_rows", C.c_int64),
    ]


class VrFilter(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_must_folder_sets", C.c_int32),
        ("must_folder_ids", C.POINTER(C.c_int32)),
        ("must_folder_off", C.POINTER(C.c_int32)),
        ("not_folder_ids", C.POINTER(C.c_int32)),
        ("n_not_folder", C.c_int32),
        ("n_not_index_folder", C.c_int32),
        ("not_index_folder_ids", C.POINTER(C.c_int32)),
        ("has_date_start", C.c_int32),
        ("has_date_end", C.c_int32),
        ("date_start", C.c_int64),
        ("date_end", C.c_int64),
        ("date_field", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class VrBertDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("layers", C.c_int32),
        ("hidden", C.c_int32),
        ("heads", C.c_int32),
        ("intermediate", C.c_int32),
        ("vocab", C.c_int32),
        ("max_pos", C.c_int32),
        ("type_vocab", C.c_int32),
        ("pooling", C.c_int32),
        ("normalize", C.c_int32),
        ("eps", C.c_float),
        ("precision", C.c_int32),
    ]


VR_PRECISION_F32 = 0
VR_PRECISION_F16X3 = 1
VR_PRECISION_F16 = 2
VR_POOL_MEAN = 0
VR_POOL_CLS = 1
VR_MEM_HOST = 0
VR_MEM_DEVICE = 1
VR_TS_ABSENT = -(2**63)
VR_FUSION_MINMAX = 0
VR_FUSION_RRF = 1

_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/voitta_engine.h declares
SIGNATURES = {
    "vr_abi_version": (C.c_int, []),
    "vr_last_error": (C.c_char_p, []),
    "vr_engine_create": (C.c_int, [C.POINTER(VrConfig), C.POINTER(_vp)]),
    "vr_engine_destroy": (None, [_vp]),
    "vr_sync": (C.c_int, [_vp]),
    "vr_stream": (_vp, [_vp]),
    "vr_set_stream": (C.c_int, [_vp, _vp]),
    "vr_encoder_load": (C.c_int, [_vp, C.POINTER(VrBertDesc), C.POINTER(_vp), C.c_int32, C.c_int]),
    "vr_encode": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int, _vp, C.c_int]),
    "vr_wordpiece_create": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(_vp)]),
    "vr_wordpiece_destroy": (None, [_vp]),
    "vr_wordpiece_encode": (C.c_int, [_vp, C.POINTER(C.c_char_p), _i64p, C.c_int64, C.c_int32, _i64p, _i32p, C.c_int64,
                                     _i64p]),
    "vr_chunk_texts": (C.c_int, [C.POINTER(C.c_char_p), _i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(_vp)]),
    "vr_chunks_view": (C.c_int, [_vp, _i64p, C.POINTER(_i64p), C.POINTER(_i64p), C.POINTER(_i64p),
                                C.POINTER(C.POINTER(C.c_char))]),
    "vr_chunks_free": (None, [_vp]),
    "vr_bm25_tf": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp]),
    "vr_bm25_tokenize": (C.c_int, [C.POINTER(C.c_char_p), _i64p, C.c_int64, _i64p, _i32p, C.c_int64, _i64p]),
    "vr_porter2_stem": (C.c_int, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]),
    "vr_index_batch": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                 _i32p, _i32p, _i64p, _i64p, _i64p]),
    "vr_upsert": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, _i32p, _i32p, _i64p, _i64p, _i64p]),
    "vr_delete_rows": (C.c_int, [_vp, _i64p, C.c_int64]),
    "vr_stats": (C.c_int, [_vp, C.c_int32, _i64p]),
    "vr_count": (C.c_int, [_vp, _i64p, _i64p]),
    "vr_get_dense": (C.c_int, [_vp, _i64p, C.c_int64, _fp]),
    "vr_sparse_stats": (C.c_int, [_vp, _i32p, C.c_int32, _i32p, _i64p]),
    "vr_search_dense": (C.c_int, [_vp, _vp, C.c_int32, C.c_int, C.c_int32, C.POINTER(VrFilter), _i64p, _fp, _i32p]),
    "vr_search_dense_keys": (C.c_int, [_vp, _vp, C.c_int32, C.c_int, C.c_int32, C.POINTER(VrFilter), _vp, C.c_int]),
    "vr_search_sparse": (C.c_int, [_vp, _i32p, _fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(VrFilter), _i64p, _fp,
                                   _i32p]),
    "vr_idf": (C.c_float, [C.c_int64, C.c_int32]),
    "vr_profile": (C.c_int, [_vp, C.c_int]),
    "vr_profile_read": (C.c_int, [_vp, C.c_int, _dp, _i64p, _dp]),
    "vr_search_hybrid": (C.c_int, [_vp, _vp, C.c_int, _i32p, _fp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                   C.POINTER(VrFilter), _i64p, _dp, _i32p, _i32p]),
    "vr_compact": (C.c_int, [_vp, _i64p, _i64p]),
    "vr_save": (C.c_int, [_vp, C.c_char_p]),
    "vr_load": (C.c_int, [_vp, C.c_char_p]),
    "vr_fuse_minmax": (C.c_int, [_i64p, _fp, C.c_int32, _i64p, _fp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                 _i64p, _dp, _i32p, _i32p]),
    "vr_fuse_rrf": (C.c_int, [_i64p, C.c_int32, _i64p, C.c_int32, C.c_int32, _i64p, _dp, _i32p, _i32p]),
    "vr_search_sparse_batch": (C.c_int, [_vp, _i64p, _i32p, _fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(VrFilter),
                                         _i64p, _fp, _i32p]),
    "vr_search_hybrid_batch": (C.c_int, [_vp, _vp, C.c_int32, C.c_int, _i64p, _i32p, _fp, C.c_int32, C.c_double,
                                         C.c_int32, C.POINTER(VrFilter), _i64p, _dp, _i32p, _i32p]),
    "vr_search_hybrid_keys": (C.c_int, [_vp, _vp, C.c_int32, C.c_int, _i64p, _i32p, _fp, C.c_int32, C.c_int32,
                                        C.POINTER(VrFilter), _vp, C.c_int]),
    "vr_merge_keys": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int, _i64p, _fp, _i32p]),
    "vr_fuse_batch": (C.c_int, [_i64p, _fp, _i32p, _i64p, _fp, _i32p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                C.c_int32, C.c_int32, _i64p, _dp, _i32p, _i32p]),
    "vr_sparse_row_ids": (C.c_int, [_vp, _i64p, C.c_int64, _vp, C.c_int64, C.c_int, _i32p, _i64p]),
    "vr_df_apply": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int64, C.c_int32]),
    "vr_query_text": (C.c_int, [_vp, _vp, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                C.c_int32, C.POINTER(VrFilter), _i64p, _dp, _i32p, _i32p, _i32p]),
}


def library_path() -> str:
    return os.environ.get("VOITTA_ENGINE_LIB", os.path.join(_HERE, _LIB_NAME))


def load_library():
    """Load libvoitta_engine.so and declare every prototype. Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EngineError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C voitta_rag_amd/csrc`). There is no CPU fallback."
        )
    lib = C.CDLL(path)
    # VOITTA_ENGINE_HOST_ONLY=1: a library that carries only the host-only entry points (the sanitizer build of the
    # tokenizers, the chunker and the fusion — `make asan`); everything that needs the GPU is simply absent from it
    host_only = os.environ.get("VOITTA_ENGINE_HOST_ONLY") == "1"
    for name, (restype, argtypes) in SIGNATURES.items():
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    abi = lib.vr_abi_version()
    if abi != 1:
        raise EngineError(f"libvoitta_engine ABI {abi}, binding expects 1")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().vr_last_error()
        raise EngineError(msg.decode("utf-8", "replace") if msg else f"libvoitta_engine error {rc}")
