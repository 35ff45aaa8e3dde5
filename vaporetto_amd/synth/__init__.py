"""Synthetic models and sentences in the real Vaporetto formats (bench / test tooling).

Not on the product path.  `build_synth()` compiles libvpt_synth.so with g++ in-tree."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvpt_synth.so")
_SOURCES = [os.path.join(HERE, "synth.cpp"), os.path.join(HERE, "..", "csrc", "model.cpp")]

M1_BCCWJ_LIKE, M2_KYTEA_LIKE, M3_TAGS = 1, 2, 3
M1_NONBMP = 6   # M1 + 40 unigrams, 100 n-grams and 100 dictionary words with kanji outside the BMP
SEED_BASE = 0x5EED0000  # + config number


def build_synth(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in _SOURCES):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB_PATH] + _SOURCES)
    return LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build_synth()
        L = C.CDLL(LIB_PATH)
        L.vpt_synth_model.argtypes = [C.c_int, C.c_uint64, C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.vpt_synth_model_ex.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_uint32, C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.vpt_synth_sentences.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.vpt_synth_sentences_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32, C.c_double,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.vpt_synth_sentences_nb.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.vpt_synth_blocks.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.vpt_synth_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def synth_model(kind: int = M1_BCCWJ_LIKE, seed: int = SEED_BASE + 2, scale: float = 1.0, vocab: int = 0, dup_share: float = 0.0) -> bytes:
    """Model file bytes ("VaporettoTokenizer 0.5.0\\n" + bincode) of the documented shape.  vocab / dup_share: the model-shape sweep's
    knobs (alphabet size; share of dictionary words that repeat a char n-gram, i.e. of rows outside the packed fields)."""
    L = _load()
    out, n = C.c_void_p(), C.c_size_t()
    st = L.vpt_synth_model_ex(kind, seed, scale, vocab, dup_share, C.byref(out), C.byref(n))
    if st != 0:
        raise RuntimeError("vpt_synth_model failed: %d" % st)
    try:
        return C.string_at(out, n.value)
    finally:
        L.vpt_synth_free(out)


def synth_sentences(model_bytes: bytes, n_sentences: int, min_len: int = 64, max_len: int = 64,
                    seed: int = SEED_BASE + 2, hit_share: float = 0.7, nonbmp_share: float = 0.0):
    """(utf8 uint8[bytes], byte_offsets uint64[S+1]) -- `hit_share` (70 %) of a sentence's items are model patterns
    (Zipf), the others alphabet-A characters; `nonbmp_share` of the items (in front of that draw) are chars outside the BMP, half of
    them as one of the model's patterns that hold such a char."""
    L = _load()
    text, nbytes, boff = C.c_void_p(), C.c_size_t(), C.c_void_p()
    if nonbmp_share:
        st = L.vpt_synth_sentences_nb(model_bytes, len(model_bytes), seed, n_sentences, min_len, max_len, C.c_double(hit_share), C.c_double(nonbmp_share),
                                      C.byref(text), C.byref(nbytes), C.byref(boff))
    else:
        st = L.vpt_synth_sentences_ex(model_bytes, len(model_bytes), seed, n_sentences, min_len, max_len, C.c_double(hit_share),
                                      C.byref(text), C.byref(nbytes), C.byref(boff))
    if st != 0:
        raise RuntimeError("vpt_synth_sentences failed: %d" % st)
    try:
        utf8 = np.frombuffer(C.string_at(text, nbytes.value), dtype=np.uint8).copy()
        offs = np.frombuffer(C.string_at(boff, 8 * (n_sentences + 1)), dtype=np.uint64).copy()
    finally:
        L.vpt_synth_free(text)
        L.vpt_synth_free(boff)
    return utf8, offs


def synth_blocks(model_bytes: bytes, first_block: int, n_blocks: int, block_sentences: int = 100000, min_len: int = 64, max_len: int = 64,
                 seed: int = SEED_BASE + 3, nthreads: int = 0):
    """A run of blocks of a big batch (block j = `block_sentences` sentences from the stream seeded seed + 7919 j), generated in
    parallel: (utf8, byte_offsets) of blocks first_block .. first_block + n_blocks - 1, offsets starting at 0."""
    L = _load()
    text, nbytes, boff = C.c_void_p(), C.c_size_t(), C.c_void_p()
    st = L.vpt_synth_blocks(model_bytes, len(model_bytes), seed, first_block, n_blocks, block_sentences, min_len, max_len,
                            nthreads or min(32, os.cpu_count() or 1), C.byref(text), C.byref(nbytes), C.byref(boff))
    if st != 0:
        raise RuntimeError("vpt_synth_blocks failed: %d" % st)
    try:
        utf8 = np.ctypeslib.as_array(C.cast(text, C.POINTER(C.c_uint8)), shape=(nbytes.value,)).copy()
        offs = np.ctypeslib.as_array(C.cast(boff, C.POINTER(C.c_uint64)), shape=(n_blocks * block_sentences + 1,)).copy()
    finally:
        L.vpt_synth_free(text)
        L.vpt_synth_free(boff)
    return utf8, offs
