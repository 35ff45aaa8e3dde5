"""ctypes binding of oracle/libvaporetto_oracle.so (ORACLE = test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvaporetto_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "vaporetto_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libvaporetto_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.vo_predictor_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.vo_predictor_destroy.argtypes = [C.c_void_p]
        L.vo_predict.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        L.vo_count_boundaries.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vo_predict_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.vo_predict_batch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
        L.vo_baseline_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.vo_predict_tags.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.vo_n_tags.argtypes = [C.c_void_p]
        L.vo_n_tags.restype = C.c_uint32
        L.vo_tag_score_stride.argtypes = [C.c_void_p]
        L.vo_tag_score_stride.restype = C.c_uint32
        L.vo_fill_tags_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_void_p, C.c_int]
        L.vo_write_tokenized_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.vo_tag_scores_probe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32]
        L.vo_char_pattern_count.argtypes = [C.c_void_p]
        L.vo_char_pattern_count.restype = C.c_uint32
        L.vo_char_pattern_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                          C.c_void_p, C.c_uint32]
        L.vo_test_posw_add_assign.argtypes = [C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32,
                                              C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, status, msg):
        super().__init__("status %d: %s" % (status, msg))
        self.status = status
        self.msg = msg


class OraclePredictor:
    """CPU oracle predictor (reference algorithm).  `predict_tags` as in Predictor::new."""

    def __init__(self, model_bytes: bytes, predict_tags: bool = False):
        self._h = C.c_void_p()
        err = C.create_string_buffer(256)
        st = lib().vo_predictor_create(model_bytes, len(model_bytes), int(predict_tags), C.byref(self._h), err, 256)
        if st != 0:
            raise OracleError(st, err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_predictor_destroy(self._h)
            self._h = None

    def predict(self, text: str):
        raw = text.encode("utf-8")
        scores = np.zeros(max(len(raw), 1), dtype=np.int32)
        labels = np.zeros(max(len(raw), 1), dtype=np.uint8)
        nb = C.c_size_t()
        st = lib().vo_predict(self._h, raw, len(raw), scores.ctypes.data, labels.ctypes.data, C.byref(nb))
        if st != 0:
            raise OracleError(st, "predict failed")
        return scores[:nb.value].tolist(), labels[:nb.value].tolist()

    def predict_batch(self, utf8: np.ndarray, byte_offsets: np.ndarray, nthreads: int = 1, out=None, pin: bool = False, double_array: bool = False):
        """utf8: uint8 array, byte_offsets: uint64[S+1].  Returns (scores, labels, out_offsets, A_char bytes).
        `out` = the (scores, labels, out_offsets) of an earlier call on the same batch: written in place -- a timed repeat then pays
        no page faults of fresh arrays; `pin` = worker t stays on the t-th CPU of the process (timed baseline runs); `double_array` = the char scorer's automaton walked
        as a double array (what the reference's daachorse matcher is; the baseline leg of bench.py -- same scores by construction, checked in
        tests/test_oracle_c_kat.py)."""
        S = len(byte_offsets) - 1
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        if out is None:
            ooff = np.zeros(S + 1, dtype=np.uint64)
            st = lib().vo_count_boundaries(utf8.ctypes.data, byte_offsets.ctypes.data, S, ooff.ctypes.data)
            if st != 0:
                raise OracleError(st, "count_boundaries failed")
            nb = int(ooff[S])
            scores = np.zeros(nb, dtype=np.int32)
            labels = np.zeros(nb, dtype=np.uint8)
        else:
            scores, labels, ooff = out
        ab = C.c_uint64()
        st = lib().vo_predict_batch_ex(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, scores.ctypes.data,
                                       labels.ctypes.data, ooff.ctypes.data, nthreads, C.byref(ab), (1 if pin else 0) | (2 if double_array else 0))
        if st != 0:
            raise OracleError(st, "predict_batch failed")
        return scores, labels, ooff, ab.value

    def baseline_timed(self, utf8: np.ndarray, byte_offsets: np.ndarray, out, nthreads: int = 1, reps: int = 3, double_array: bool = True, replicate: bool = True,
                       huge_pages: bool = False):
        """bench.py's cpu_baseline leg (vo_baseline_timed): `reps` passes of the reference's loop over the batch on a pool of `nthreads` pinned workers
        that lives for the whole call -- the passes are timed between barriers, thread start-up is in none -- with the data every char walks
        replicated per NUMA node (`huge_pages`: those copies on 2 MB pages).  `out` = (scores, labels, out_offsets) of an earlier call on this batch (written in place).  Returns
        (seconds per pass, A_char bytes, NUMA nodes used)."""
        S = len(byte_offsets) - 1
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        scores, labels, ooff = out
        secs = np.zeros(max(reps, 1), dtype=np.float64)
        ab, nodes = C.c_uint64(), C.c_int()
        st = lib().vo_baseline_timed(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, scores.ctypes.data, labels.ctypes.data, ooff.ctypes.data, nthreads,
                                     (2 if double_array else 0) | (4 if replicate else 0) | (8 if huge_pages else 0), reps, secs.ctypes.data, C.byref(ab), C.byref(nodes))
        if st != 0:
            raise OracleError(st, "baseline_timed failed")
        return secs.tolist(), ab.value, nodes.value

    def predict_tags(self, text: str, labels=None):
        raw = text.encode("utf-8")
        n = len(text)
        nt = C.c_uint32()
        out = np.full(max(n, 1) * 64, -1, dtype=np.int32)
        lab = None
        if labels is not None:
            lab = np.ascontiguousarray(labels, dtype=np.uint8)
        st = lib().vo_predict_tags(self._h, raw, len(raw), lab.ctypes.data if lab is not None else None,
                                   out.ctypes.data, C.byref(nt))
        if st != 0:
            raise OracleError(st, "predict_tags failed")
        return out[:n * nt.value].reshape(n, nt.value) if nt.value else out[:0].reshape(n, 0), nt.value

    def n_tags(self) -> int:
        return int(lib().vo_n_tags(self._h))

    def tag_score_stride(self) -> int:
        return int(lib().vo_tag_score_stride(self._h))

    def fill_tags_batch(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, labels: np.ndarray, nthreads: int = 1,
                        want_scores: bool = True):
        """Sentence::fill_tags over a packed batch on the caller's labels (uint8 per boundary, 0/1/2).  Returns (tags [chars, n_tags],
        scores [chars, stride] or None, models [chars]): what Predictor::store_tag_scores(true) keeps per token (predictor.rs:599-601)."""
        S = len(byte_offsets) - 1
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        nt, stride = self.n_tags(), self.tag_score_stride()
        rows = int(out_offsets[S]) + S
        tags = np.full((rows, max(nt, 1)), -1, dtype=np.int32)
        scores = np.zeros((rows, max(stride, 1)), dtype=np.int32) if want_scores else None
        models = np.full(rows, -1, dtype=np.int32)
        lab = labels if len(labels) else np.zeros(1, dtype=np.uint8)
        st = lib().vo_fill_tags_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data, lab.ctypes.data,
                                      tags.ctypes.data, scores.ctypes.data if want_scores else None, stride, models.ctypes.data, nthreads)
        if st != 0:
            raise OracleError(st, "fill_tags_batch failed")
        return tags[:, :nt], (scores[:, :stride] if want_scores else None), models

    def write_tokenized_batch(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, labels: np.ndarray,
                              tags: np.ndarray = None, models: np.ndarray = None, nthreads: int = 1):
        """Sentence::write_tokenized_text (sentence.rs:850-886) for every sentence of a packed batch; with `tags` / `models` (as
        fill_tags_batch returned them) the "/tag" suffixes too.  Returns (text uint8, text offsets uint64[S+1])."""
        S = len(byte_offsets) - 1
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        lab = labels if len(labels) else np.zeros(1, dtype=np.uint8)
        toff = np.zeros(S + 1, dtype=np.uint64)
        if tags is not None:
            tags = np.ascontiguousarray(tags, dtype=np.int32)
            models = np.ascontiguousarray(models, dtype=np.int32)
            assert tags.shape[1] == self.n_tags()
        # the size first (a capacity of 0 fails after the offsets are complete), then the text
        lib().vo_write_tokenized_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data, lab.ctypes.data,
                                       tags.ctypes.data if tags is not None else None, models.ctypes.data if tags is not None else None,
                                       None, 0, toff.ctypes.data, nthreads)
        out = np.zeros(int(toff[S]) + 1, dtype=np.uint8)
        st = lib().vo_write_tokenized_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data, lab.ctypes.data,
                                            tags.ctypes.data if tags is not None else None, models.ctypes.data if tags is not None else None,
                                            out.ctypes.data, int(toff[S]), toff.ctypes.data, nthreads)
        if st != 0:
            raise OracleError(st, "write_tokenized_batch failed")
        return out[:int(toff[S])], toff

    def tag_scores_probe(self, text: str, which: int, token_id: int, pos: int, init):
        raw = text.encode("utf-8")
        z = np.array(init, dtype=np.int32)
        st = lib().vo_tag_scores_probe(self._h, raw, len(raw), which, token_id, pos, z.ctypes.data, len(z))
        if st != 0:
            raise OracleError(st, "tag probe failed")
        return z.tolist()

    def char_patterns(self):
        out = []
        for i in range(lib().vo_char_pattern_count(self._h)):
            off, ln = C.c_int32(), C.c_uint32()
            w = np.zeros(64, dtype=np.int32)
            lib().vo_char_pattern_get(self._h, i, C.byref(off), C.byref(ln), w.ctypes.data, 64)
            out.append((off.value, w[:ln.value].tolist()))
        return out


def posw_add_assign(off_a, wa, off_b, wb):
    a = np.array(wa, dtype=np.int32)
    b = np.array(wb, dtype=np.int32)
    out = np.zeros(64, dtype=np.int32)
    off, ln = C.c_int32(), C.c_uint32()
    lib().vo_test_posw_add_assign(off_a, a.ctypes.data, len(a), off_b, b.ctypes.data, len(b), C.byref(off),
                                  out.ctypes.data, C.byref(ln), 64)
    return off.value, out[:ln.value].tolist()
