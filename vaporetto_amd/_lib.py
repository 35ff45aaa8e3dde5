"""ctypes binding of libvaporetto_hip.so (the C ABI declared in include/vaporetto_hip.h).

The library is the product: there is no Python or CPU fallback.  If it has not been built, loading fails
loudly; build it with `python -m vaporetto_amd.build` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libvaporetto_hip.so")

VPT_OK, VPT_INVALID_MODEL, VPT_INVALID_ARGUMENT, VPT_RUNTIME_ERROR = 0, 1, 2, 3
VPT_FLAG_KYTEA_FULLWIDTH = 1
VPT_FLAG_SPLIT_LINEBREAKS = 1 << 7


def VPT_FLAG_WSCONST(char_type: int) -> int:
    return 1 << int(char_type)


class ModelInfo(C.Structure):
    _fields_ = [
        ("n_char_ngrams", C.c_uint32), ("n_type_ngrams", C.c_uint32), ("n_dict_words", C.c_uint32),
        ("n_tag_models", C.c_uint32), ("bias", C.c_int32), ("char_window", C.c_uint32), ("type_window", C.c_uint32),
        ("max_pattern_chars", C.c_uint32), ("n_short_entries", C.c_uint32), ("n_long_nodes", C.c_uint32),
        ("type_kind", C.c_uint32), ("device_table_bytes", C.c_uint64), ("hot_table_bytes", C.c_uint64),
        ("packed", C.c_uint32), ("n_displaced", C.c_uint32), ("type_rows", C.c_uint32), ("n_overflow_children", C.c_uint32),
        ("predict_tags", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/vaporetto_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "vpt_last_error": (C.c_char_p, []),
    "vpt_version": (C.c_char_p, []),
    "vpt_predictor_create": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.POINTER(_P)]),
    "vpt_predictor_destroy": (None, [_P]),
    "vpt_predictor_save": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vpt_predictor_load": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "vpt_predictor_clone_to_device": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "vpt_predictor_describe": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "vpt_predictor_adopt_device": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "vpt_count_boundaries": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "vpt_predict_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P]),
    "vpt_predict_batch_flags": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, C.c_uint]),
    "vpt_predict_one": (C.c_int, [_P, _P, C.c_size_t, _P, _P, C.POINTER(C.c_size_t)]),
    "vpt_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "vpt_host_free": (None, [_P]),
    "vpt_predict_batch_sharded": (C.c_int, [_P, C.c_size_t, _P, _P, C.c_size_t, _P, _P, _P, C.c_uint]),
    "vpt_shard_bounds": (C.c_int, [_P, C.c_size_t, C.c_size_t, _P]),
    "vpt_char_types_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_uint, _P]),
    "vpt_char_types_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, _P]),
    "vpt_predictor_n_tags": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "vpt_fill_tags_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P]),
    "vpt_batch_last_plan": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vpt_predictor_tag_score_stride": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "vpt_fill_tags_scores_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, C.c_uint, _P, _P, _P]),
    "vpt_fill_tags_scores_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, _P, _P, _P, _P]),
    "vpt_fill_tags_batch_flags": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, C.c_uint]),
    "vpt_batch_set_flags": (C.c_int, [_P, C.c_uint]),
    "vpt_write_tokenized_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, C.c_uint64, _P]),
    "vpt_predict_write_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, _P, _P, _P, C.c_uint64, _P, _P]),
    "vpt_write_tokenized_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, _P, C.c_uint64, _P, _P]),
    "vpt_predictor_max_tag_suffix": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "vpt_count_boundaries_device": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, _P]),
    "vpt_tokenize_batch": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_uint, C.c_int, _P, C.c_uint64, _P]),
    "vpt_write_tagged_batch": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, C.c_uint, _P, C.c_uint64, _P]),
    "vpt_write_tagged_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, _P, _P, C.c_uint64, _P, _P]),
    "vpt_fill_tags_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, _P, _P]),
    "vpt_expand_tags_batch_device": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, _P, _P]),
    "vpt_batch_create": (C.c_int, [_P, C.POINTER(_P)]),
    "vpt_batch_destroy": (None, [_P]),
    "vpt_predict_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, _P, _P, _P]),
    "vpt_batch_sync": (C.c_int, [_P]),
    "vpt_batch_set_max_sentence_chars": (C.c_int, [_P, C.c_uint64]),
    "vpt_batch_set_timing": (C.c_int, [_P, C.c_int]),
    "vpt_batch_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "vpt_batch_kernel_times": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vpt_batch_phase_cycles": (C.c_int, [_P, C.POINTER(C.c_uint64 * 8)]),
    "vpt_batch_node_reads": (C.c_int, [_P, C.POINTER(C.c_uint64 * 8)]),
    "vpt_model_inspect": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(ModelInfo)]),
    "vpt_model_read_len": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vpt_predictor_info": (C.c_int, [_P, C.POINTER(ModelInfo)]),
}

_lib = None


def _init_torch_hip_first():
    """PyTorch wheels bundle their own libamdhip64/libhsa-runtime64 (different soname from /opt/rocm's, which
    this library links).  Two HIP runtimes can share a process only if torch's initialises first, so when torch
    is importable and sees a GPU, let it.  Torch is never used for compute here."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


def load():
    """Loads the shared library; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        _init_torch_hip_first()
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: build it with `python -m vaporetto_amd.build` "
                          "(there is no CPU or Python fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return load().vpt_last_error().decode("utf-8", "replace")
