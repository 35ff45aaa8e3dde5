"""Host-side mirror of the reference's public API for the predict path (vaporetto/src/lib.rs:82-91):
`Model`, `Predictor`, `Sentence`, `CharacterBoundary`, `CharacterType`, `VaporettoError`.

Same names, argument meaning and error behaviour as the Rust crate, over the C ABI of libvaporetto_hip.so.
All scoring happens in the HIP kernels; this module only parses text and moves buffers.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, modelfmt


class VaporettoError(Exception):
    """errors.rs:15-38.  `kind` is "InvalidModel", "InvalidArgument" or "Runtime"."""

    def __init__(self, kind: str, message: str):
        super().__init__(message)
        self.kind = kind


def _raise(status: int):
    msg = _lib.last_error()
    kind = {_lib.VPT_INVALID_MODEL: "InvalidModel", _lib.VPT_INVALID_ARGUMENT: "InvalidArgument"}.get(status, "Runtime")
    raise VaporettoError(kind, msg)


class CharacterType(enum.IntEnum):  # sentence.rs:11-29
    Digit = 1
    Roman = 2
    Hiragana = 3
    Katakana = 4
    Kanji = 5
    Other = 6

    @staticmethod
    def get_type(c: str) -> "CharacterType":  # sentence.rs:50-67
        return CharacterType(int(_types_of(np.array([ord(c)], dtype=np.uint32))[0]))


class CharacterBoundary(enum.IntEnum):  # sentence.rs:70-82
    NotWordBoundary = 0
    WordBoundary = 1
    Unknown = 2


_RANGES = [  # (lo, hi, type) -- sentence.rs:52-65
    (0x30, 0x39, 1), (0xFF10, 0xFF19, 1),
    (0x41, 0x5A, 2), (0x61, 0x7A, 2), (0xFF21, 0xFF3A, 2), (0xFF41, 0xFF5A, 2),
    (0x3040, 0x3096, 3),
    (0x30A0, 0x30FA, 4), (0x30FC, 0x30FF, 4), (0xFF66, 0xFF9F, 4),
    (0x3400, 0x4DBF, 5), (0x4E00, 0x9FFF, 5), (0xF900, 0xFAFF, 5), (0x20000, 0x2A6DF, 5), (0x2A700, 0x2B73F, 5),
    (0x2B740, 0x2B81F, 5), (0x2B820, 0x2CEAF, 5), (0x2F800, 0x2FA1F, 5),
]


def _types_of(cps: np.ndarray) -> np.ndarray:
    out = np.full(cps.shape, 6, dtype=np.uint8)
    for lo, hi, t in _RANGES:
        out[(cps >= lo) & (cps <= hi)] = t
    return out


class KyteaFullwidthFilter:
    """vaporetto_rules/src/string_filters/kytea_fullwidth.rs:13-117 (host-side statement of the map the kernels apply
    when `fullwidth=True` is passed; useful to build the normalised text that tokens are NOT taken from)."""

    _PAIRS = None

    @classmethod
    def table(cls) -> dict:
        if cls._PAIRS is None:
            t = {}
            for i in range(26):
                t[ord("a") + i] = 0xFF41 + i
                t[ord("A") + i] = 0xFF21 + i
            for i in range(10):
                t[ord("0") + i] = 0xFF10 + i
            t.update({ord("("): 0xFF08, ord(")"): 0xFF09, ord("{"): 0xFF5B, ord("}"): 0xFF5D, ord("<"): 0xFF1C, ord(">"): 0xFF1E,
                      0xFF62: 0x300C, 0xFF63: 0x300D, ord("["): 0xFF3B, ord("]"): 0xFF3D, ord("-"): 0x2212, 0xFF5E: 0x301C,
                      ord("."): 0x3002, 0xFF0D: 0x30FC, ord("/"): 0xFF0F, ord("_"): 0xFF3F, ord(","): 0xFF0C, ord("%"): 0xFF05,
                      ord("?"): 0xFF1F, 0xFF64: 0x3001, 0x2015: 0x30FC, ord('"'): 0x201D, ord("'"): 0x2019, 0xFF65: 0x30FB,
                      0x2500: 0x30FC, ord("+"): 0xFF0B, ord(":"): 0xFF1A, 0x2013: 0x30FC, ord("!"): 0xFF01, 0xFF61: 0x3002,
                      ord("&"): 0xFF06, ord("*"): 0xFF0A, ord("@"): 0xFF20, ord("="): 0xFF1D})
            cls._PAIRS = t
        return cls._PAIRS

    def filter(self, string: str) -> str:
        return string.translate(self.table())


class ConcatGraphemeClustersFilter:
    """vaporetto_rules/src/sentence_filters/concat_grapheme_clusters.rs:10-36 (the CLI's `--wsconst G`, predict/src/main.rs:101-104): a
    sentence filter that runs on the HOST between predict and fill_tags -- every boundary inside an extended grapheme cluster (UAX #29)
    becomes NotWordBoundary, so a ZWJ sequence or a base + modifier is never cut.  The reference segments with the `unicode-segmentation`
    crate (1.12.0); here the clusters come from the `regex` module's \\X, whichever Unicode version that module carries (the crate's and
    the module's rules agree on everything the reference's tests hold: concat_grapheme_clusters.rs:43-88).  It only CLEARS boundaries, so it
    commutes with KyteaWsConstFilter (which clears too) -- but not with SplitLinebreaksFilter, which SETS the boundary between "\r" and "\n"
    while CR LF is one cluster: the reference applies its filters in the caller's order (predict/src/main.rs:130-134);
    `Predictor.tokenize(wsconst=("G", ..), split_linebreaks=True)` has ONE fixed order -- the device's label flags (wsconst types, then
    split_linebreaks), then "G" on the host -- so "\r\n" stays joined there.  A caller that wants the other order runs this filter itself
    between `predict_packed(.., wsconst=.., split_linebreaks=False)` and its own SplitLinebreaksFilter pass."""

    _X = None

    @classmethod
    def cluster_lengths(cls, text: str) -> List[int]:
        """Chars of every extended grapheme cluster of `text`, in order."""
        if cls._X is None:
            try:
                import regex
            except ImportError as e:   # no silent "one char per cluster"
                raise ImportError("ConcatGraphemeClustersFilter needs the `regex` module (UAX #29 segmentation)") from e
            cls._X = regex.compile(r"\X")
        return [len(m.group()) for m in cls._X.finditer(text)]

    def filter(self, sentence: "Sentence") -> None:
        b = sentence.boundaries_mut()
        start = 0
        for n in self.cluster_lengths(sentence.as_raw_text()):
            b[start:start + n - 1] = CharacterBoundary.NotWordBoundary
            start += n

    def filter_packed(self, texts: Sequence[str], out_offsets: np.ndarray, labels: np.ndarray) -> None:
        """The same on a packed batch's labels (sentence i's boundaries are labels[out_offsets[i]:out_offsets[i + 1]]), in place."""
        for i, t in enumerate(texts):
            if t.isascii():      # a cluster of several ASCII chars is CR LF only
                if "\r\n" not in t:
                    continue
            start = int(out_offsets[i])
            for n in self.cluster_lengths(t):
                if n > 1:
                    labels[start:start + n - 1] = CharacterBoundary.NotWordBoundary
                start += n


class Model:
    """model.rs:55-169."""

    def __init__(self, data: Optional[modelfmt.ModelData], raw: Optional[bytes] = None):
        self._data_ = data      # decoded records; None until somebody asks for them (a 34 MB model takes Python 20 s)
        self._raw = raw
        assert data is not None or raw is not None

    @property
    def _data(self) -> modelfmt.ModelData:
        if self._data_ is None:
            try:
                self._data_ = modelfmt.decode_model(self._raw)[0]
            except modelfmt.ModelFormatError as e:   # cannot happen after read_slice's validation; still the crate's error type
                raise VaporettoError("InvalidModel", "InvalidModelError: %s" % e)
        return self._data_

    @staticmethod
    def read(rdr) -> "Model":  # model.rs:138-153
        return Model.read_slice(rdr.read())[0]

    @staticmethod
    def read_slice(buf: bytes) -> Tuple["Model", bytes]:  # model.rs:127-135
        """The bytes are validated by the library's decoder (the same one vpt_predictor_create uses); the Python records
        behind dictionary() / tag_models() are decoded on first use."""
        buf = bytes(buf)
        try:
            L = _lib.load()
        except OSError:
            # Reading, inspecting or converting a model needs no GPU and no built library (model tooling on a host without
            # ROCm): the Python codec decodes -- and so validates -- the records right away instead.  COMPUTE never takes this
            # route: Predictor() loads the library and fails loudly without it.
            try:
                data, n_used = modelfmt.decode_model(buf)
            except modelfmt.ModelFormatError as e:
                raise VaporettoError("InvalidModel", "InvalidModelError: %s" % e)
            return Model(data, buf[:n_used]), buf[n_used:]
        used = C.c_size_t(0)
        st = L.vpt_model_read_len(buf, len(buf), C.byref(used))
        if st != _lib.VPT_OK:
            raise VaporettoError("InvalidModel" if st == _lib.VPT_INVALID_MODEL else "Runtime", L.vpt_last_error().decode("utf-8"))
        return Model(None, buf[:used.value]), buf[used.value:]

    def to_vec(self) -> bytes:  # model.rs:99-104
        if self._raw is None:
            self._raw = modelfmt.encode_model(self._data)
        return self._raw

    def write(self, wtr) -> None:  # model.rs:112-121
        wtr.write(self.to_vec())

    def dictionary(self):  # model.rs:155-158
        return self._data.dict_model

    def replace_dictionary(self, dict_records) -> None:  # model.rs:160-163
        self._data.dict_model = list(dict_records)
        self._raw = None

    def tag_models(self):  # model.rs:165-168
        return self._data.tag_models


class Token:
    """sentence.rs:1195-1263: a token of a Sentence, chars [start, end)."""

    def __init__(self, sentence: "Sentence", start: int, end: int):
        self._s, self._start, self._end = sentence, start, end

    def surface(self) -> str:
        return self._s._text[self._start:self._end]

    def tags(self) -> List[Optional[str]]:  # sentence.rs:1210-1214
        nt = self._s._n_tags
        return list(self._s._tags[(self._end - 1) * nt:self._end * nt])

    def tag_candidates(self) -> List[List[Tuple[str, int]]]:  # sentence.rs:1216-1250
        """Per tag slot the candidates with their scores (score 0 for the only candidate of a slot).  Like the reference this
        requires Predictor.store_tag_scores(True) before fill_tags."""
        if not self._s._tag_scores:
            raise AssertionError("Predictor::store_tag_scores() must be set to true to use this function.")
        entry = self._s._tag_scores[self._end - 1]
        results = []
        if entry is not None:
            tags, scores = entry
            i = 0
            for cands in tags:
                if len(cands) == 1:
                    results.append([(cands[0], 0)])
                else:
                    results.append([(c, scores[i + k]) for k, c in enumerate(cands)])
                    i += len(cands)
        return results

    def start(self) -> int:
        return self._start

    def end(self) -> int:
        return self._end


class Sentence:
    """sentence.rs:85-101 (raw-text path).  Holds the text, its character types, and after `predict`
    the boundary scores and labels."""

    def __init__(self):
        self._set_default()

    def _set_default(self):  # sentence.rs:140-158: a single space
        self._text = " "
        self._utf8 = b" "
        self._char_types = np.array([6], dtype=np.uint8)
        self._boundaries = np.zeros(0, dtype=np.uint8)
        self._scores = np.zeros(0, dtype=np.int32)
        self._has_scores = False
        self._predictor = None
        self._tags = []
        self._n_tags = 0
        self._tag_scores = []

    @staticmethod
    def default() -> "Sentence":
        return Sentence()

    def _parse_raw(self, text: str):  # sentence.rs:160-196
        if "\0" in text:
            raise VaporettoError("InvalidArgument", "InvalidArgumentError: text: must not contain NULL")
        if len(text) == 0:
            raise VaporettoError("InvalidArgument", "InvalidArgumentError: text: must contain at least one character")
        cps = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32)
        self._text = text
        self._utf8 = text.encode("utf-8")
        self._char_types = _types_of(cps)
        self._boundaries = np.full(len(cps) - 1, CharacterBoundary.Unknown, dtype=np.uint8)
        self._scores = np.zeros(0, dtype=np.int32)
        self._has_scores = False
        self._predictor = None
        self._tags = []
        self._n_tags = 0
        self._tag_scores = []

    @staticmethod
    def from_raw(text: str) -> "Sentence":  # sentence.rs:217-245
        s = Sentence()
        s._parse_raw(text)
        return s

    def update_raw(self, text: str) -> None:  # sentence.rs:264-283: on error the sentence becomes " "
        try:
            self._parse_raw(text)
        except VaporettoError:
            self._set_default()
            raise

    def as_raw_text(self) -> str:  # sentence.rs:782
        return self._text

    def __len__(self) -> int:
        return len(self._char_types)

    def char_types(self) -> np.ndarray:  # sentence.rs:1034
        return self._char_types

    def boundaries(self) -> np.ndarray:  # sentence.rs:993
        return self._boundaries

    def boundaries_mut(self) -> np.ndarray:  # sentence.rs:1016
        return self._boundaries

    def boundary_scores(self) -> np.ndarray:  # sentence.rs:1040-1046
        return self._scores if self._has_scores else np.zeros(0, dtype=np.int32)

    def char_to_str_pos(self) -> List[int]:
        pos, out = 0, [0]
        for ch in self._text:
            pos += len(ch.encode("utf-8"))
            out.append(pos)
        return out

    def n_tags(self) -> int:  # sentence.rs:1161
        return self._n_tags

    def tags(self) -> List[Optional[str]]:  # sentence.rs:1068: len() * n_tags entries, set on a token's LAST char
        return self._tags

    def fill_tags(self) -> None:
        """sentence.rs:1144-1148: tags for the current boundaries, with the predictor that last predicted this
        sentence (which must have been created with predict_tags = True, predictor.rs:548-551)."""
        if self._predictor is None:
            raise VaporettoError("InvalidArgument", "InvalidArgumentError: sentence: predict() has not been called")
        self._predictor.fill_tags_batch([self])

    def _token_ranges(self):
        """(start, end) char ranges of the tokens; tokens adjacent to an Unknown boundary are skipped
        (TokenIterator, sentence.rs:1265-1309)."""
        n = len(self)
        start, valid = 0, True
        for e in range(n):
            b = CharacterBoundary.WordBoundary if e == n - 1 else int(self._boundaries[e])
            if b == CharacterBoundary.Unknown:
                valid = False
            elif b == CharacterBoundary.WordBoundary:
                if valid:
                    yield start, e
                start, valid = e + 1, True

    def iter_tokens(self) -> Iterable[str]:  # sentence.rs:819 (surfaces only)
        for a, e in self._token_ranges():
            yield self._text[a:e + 1]

    def tokens(self) -> List["Token"]:  # sentence.rs:819: the Token objects of iter_tokens (surface, tags, tag_candidates, start, end)
        return [Token(self, a, e + 1) for a, e in self._token_ranges()]

    def write_tokenized_text(self) -> str:  # sentence.rs:850-886
        def esc(tok):
            return "".join("\\" + c if c in " \\/" else c for c in tok)
        out = []
        for a, e in self._token_ranges():
            parts = [esc(self._text[a:e + 1])]
            if self._n_tags:
                row = list(self._tags[e * self._n_tags:(e + 1) * self._n_tags])
                while row and row[-1] is None:   # up to the last Some (sentence.rs:868)
                    row.pop()
                parts += [esc(t) if t is not None else "" for t in row]
            out.append("/".join(parts))
        return " ".join(out)


class Predictor:
    """predictor.rs:433-665 (boundary prediction)."""

    def __init__(self, model: Model, predict_tags: bool = False, device: int = 0):  # Predictor::new, predictor.rs:450
        raw = model.to_vec()
        self._model = model
        self._predict_tags = bool(predict_tags)
        self._tag_models = None
        self._store_tag_scores = False
        self._h = C.c_void_p()
        st = _lib.load().vpt_predictor_create(raw, len(raw), int(predict_tags), device, C.byref(self._h))
        if st != _lib.VPT_OK:
            self._h = None
            _raise(st)
        self.device = device

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.load().vpt_predictor_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    @classmethod
    def _adopt(cls, handle, model: Optional[Model], device: int) -> "Predictor":
        self = cls.__new__(cls)
        self._h = handle
        self._model = model
        self._tag_models = None
        self._store_tag_scores = False
        self.device = device
        self._predict_tags = bool(self.info()["predict_tags"])
        return self

    def store_tag_scores(self, flag: bool) -> None:  # predictor.rs:510-514
        """Stores tag scores if `flag` is true: fill_tags then keeps every token's score vector for Token.tag_candidates."""
        self._store_tag_scores = bool(flag)

    def tag_score_stride(self) -> int:
        """vpt_predictor_tag_score_stride: the longest score vector (TagPredictor bias) over the tag models."""
        v = C.c_uint32()
        st = _lib.load().vpt_predictor_tag_score_stride(self._h, C.byref(v))
        if st != _lib.VPT_OK:
            _raise(st)
        return v.value

    def fill_tags_scores_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, labels: np.ndarray,
                                fullwidth: bool = False):
        """fill_tags_packed plus what Predictor::store_tag_scores(true) keeps (predictor.rs:599-601): returns (tags, scores, models)
        -- tags int32 [chars, n_tags]; scores int32 [chars, stride]: at a token's last char its score vector in entries
        [0, bias.len()), zeros elsewhere; models int32 [chars]: index into Model.tag_models() of the token's tag model or -1."""
        L = _lib.load()
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        S = len(byte_offsets) - 1
        nt = self.n_tags() if self._predict_tags else 0
        stride = self.tag_score_stride() if self._predict_tags else 0
        total_c = int(out_offsets[S]) + S
        tags = np.full((total_c, max(nt, 1)), -1, dtype=np.int32)
        scores = np.zeros((total_c, max(stride, 1)), dtype=np.int32)
        models = np.full(total_c, -1, dtype=np.int32)
        lab = labels if len(labels) else np.zeros(1, dtype=np.uint8)
        st = L.vpt_fill_tags_scores_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data, lab.ctypes.data,
                                          _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0, tags.ctypes.data, scores.ctypes.data, models.ctypes.data)
        if st != _lib.VPT_OK:
            _raise(st)
        return tags[:, :nt], scores[:, :stride], models

    def save_compiled(self) -> bytes:
        """Predictor::serialize_to_vec (predictor.rs:640-651), in this library's own format: the device tables + a header."""
        L = _lib.load()
        need = C.c_size_t(0)
        st = L.vpt_predictor_save(self._h, None, 0, C.byref(need))
        if st != _lib.VPT_OK:
            _raise(st)
        buf = np.empty(need.value, dtype=np.uint8)
        st = L.vpt_predictor_save(self._h, buf.ctypes.data, buf.nbytes, C.byref(need))
        if st != _lib.VPT_OK:
            _raise(st)
        return buf.tobytes()

    @classmethod
    def load_compiled(cls, blob, device: int = 0, model: Optional[Model] = None) -> "Predictor":
        """Predictor::deserialize_from_slice_unchecked (predictor.rs:653-664).  `model` is only needed to turn tag indices
        into tag strings (Sentence.tags); scoring does not use it."""
        arr = np.frombuffer(blob, dtype=np.uint8)
        h = C.c_void_p()
        st = _lib.load().vpt_predictor_load(arr.ctypes.data, arr.nbytes, device, C.byref(h))
        if st != _lib.VPT_OK:
            _raise(st)
        return cls._adopt(h, model, device)

    def clone_to_device(self, device: int) -> "Predictor":
        """The same predictor on another GPU of the node: the compiled tables go device to device (xGMI), nothing is
        compiled again."""
        h = C.c_void_p()
        st = _lib.load().vpt_predictor_clone_to_device(self._h, device, C.byref(h))
        if st != _lib.VPT_OK:
            _raise(st)
        return Predictor._adopt(h, self._model, device)

    def info(self) -> dict:
        mi = _lib.ModelInfo()
        st = _lib.load().vpt_predictor_info(self._h, C.byref(mi))
        if st != _lib.VPT_OK:
            _raise(st)
        return mi.as_dict()

    def max_tag_suffix(self) -> int:
        """vpt_predictor_max_tag_suffix: the most bytes "/tag/tag.." can add per char (sizes write_tagged's output)."""
        sfx = C.c_uint32(0)
        st = _lib.load().vpt_predictor_max_tag_suffix(self._h, C.byref(sfx))
        if st != _lib.VPT_OK:
            _raise(st)
        return int(sfx.value)

    def predict(self, sentence: Sentence) -> None:  # predictor.rs:518-543
        """One sentence through vpt_predict_one: two kernel launches and a synchronisation, tens of microseconds for work the
        reference does in one -- it exists for API parity and tests.  Anything with more than a handful of sentences belongs in
        predict_batch / predict_packed (one launch for all of them); a loop over this method is the slowest way to use the library."""
        n = len(sentence)
        scores = np.zeros(max(n - 1, 1), dtype=np.int32)
        labels = np.zeros(max(n - 1, 1), dtype=np.uint8)
        nb = C.c_size_t()
        raw = sentence._utf8
        st = _lib.load().vpt_predict_one(self._h, raw, len(raw), scores.ctypes.data, labels.ctypes.data, C.byref(nb))
        if st != _lib.VPT_OK:
            _raise(st)
        sentence._scores = scores[:nb.value]
        sentence._boundaries = labels[:nb.value]
        sentence._has_scores = True
        sentence._predictor = self   # predictor.rs:542: enables a later fill_tags

    def predict_batch(self, sentences: Sequence[Sentence], fullwidth: bool = False) -> None:
        """Predictor::predict for many sentences in one launch."""
        if not sentences:
            return
        utf8, boff = pack_texts([s._utf8 for s in sentences])
        scores, labels, ooff = self.predict_packed(utf8, boff, fullwidth=fullwidth)
        for i, s in enumerate(sentences):
            a, b = int(ooff[i]), int(ooff[i + 1])
            s._scores = scores[a:b]
            s._boundaries = labels[a:b]
            s._has_scores = True
            s._predictor = self

    def n_tags(self) -> int:
        v = C.c_uint32()
        st = _lib.load().vpt_predictor_n_tags(self._h, C.byref(v))
        if st != _lib.VPT_OK:
            _raise(st)
        return v.value

    def fill_tags_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, labels: np.ndarray,
                         fullwidth: bool = False) -> np.ndarray:
        """Predictor::predict_tags over a packed batch (predictor.rs:546-637).  labels: uint8 per boundary (0/1/2).
        Returns int32 [total chars, n_tags]: candidate index per slot, -1 = None."""
        L = _lib.load()
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        S = len(byte_offsets) - 1
        nt = self.n_tags() if self._predict_tags else 0
        total_c = int(out_offsets[S]) + S
        tags = np.full((total_c, max(nt, 1)), -1, dtype=np.int32)
        lab = labels if len(labels) else np.zeros(1, dtype=np.uint8)
        st = L.vpt_fill_tags_batch_flags(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data,
                                         lab.ctypes.data, tags.ctypes.data, _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0)
        if st != _lib.VPT_OK:
            _raise(st)
        return tags[:, :nt]

    def tokenize_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, tagged: bool = False, flags: int = 0,
                        text_out: Optional[np.ndarray] = None, offsets_out: Optional[np.ndarray] = None):
        """vpt_tokenize_batch over a packed batch: (uint8 tokenized text, uint64 [S+1] offsets).  `text_out` / `offsets_out` may be
        preallocated (pinned) arrays; text_out needs 3 x the text bytes (+ text bytes x max_tag_suffix() when tagged)."""
        L = _lib.load()
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        S = len(byte_offsets) - 1
        cap = 3 * len(utf8) + (len(utf8) * self.max_tag_suffix() if tagged else 0)
        if text_out is None:
            text_out = np.zeros(max(cap, 1), dtype=np.uint8)
        if offsets_out is None:
            offsets_out = np.zeros(S + 1, dtype=np.uint64)
        st = L.vpt_tokenize_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, flags, int(tagged), text_out.ctypes.data,
                                  min(cap, text_out.nbytes), offsets_out.ctypes.data)
        if st != _lib.VPT_OK:
            _raise(st)
        return text_out[:int(offsets_out[S])], offsets_out

    def tokenize(self, texts: Sequence[str], tagged: bool = False, fullwidth: bool = False, wsconst: Sequence = (),
                 split_linebreaks: bool = False) -> List[str]:
        """Lines in, tokenized lines out (vpt_tokenize_batch): the CLI's loop (predict/src/main.rs:122-176) for a batch,
        with char counting, scoring, post-filters, tagging and the writer on the device.  `wsconst`: CharacterType values (the
        KyteaWsConstFilter of that type, on the device) and / or "G" (ConcatGraphemeClustersFilter, predict/src/main.rs:101-104: on the
        host -- the batch then takes three calls, predict / the filter on the labels / fill_tags + writer, instead of one)."""
        if not texts:
            return []
        utf8, boff = pack_texts([t.encode("utf-8") for t in texts])
        S = len(texts)
        graphemes = any(isinstance(t, str) and t == "G" for t in wsconst)
        types = [int(t) for t in wsconst if not (isinstance(t, str) and t == "G")]
        if graphemes:
            _, labels, ooff = self.predict_packed(utf8, boff, fullwidth=fullwidth, wsconst=tuple(types), split_linebreaks=split_linebreaks)
            norm = KyteaFullwidthFilter()
            ConcatGraphemeClustersFilter().filter_packed([norm.filter(t) for t in texts] if fullwidth else texts, ooff, labels)
            text, toff = self.write_tokenized_packed(utf8, boff, ooff, labels, tagged=tagged, fullwidth=fullwidth)
        else:
            flags = _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0
            for t in types:
                flags |= _lib.VPT_FLAG_WSCONST(t)
            if split_linebreaks:
                flags |= _lib.VPT_FLAG_SPLIT_LINEBREAKS
            text, toff = self.tokenize_packed(utf8, boff, tagged=tagged, flags=flags)
        raw = bytes(text)
        return [raw[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(S)]

    def write_tokenized_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray,
                               labels: np.ndarray, tagged: bool = False, fullwidth: bool = False):
        """Sentence::write_tokenized_text (sentence.rs:850-886) over a packed batch on the device; with `tagged`
        (predictors created with predict_tags) fill_tags runs first and every token gets its "/tag" suffixes.
        Returns (uint8 text, uint64 [S+1] offsets): sentence i is text[offsets[i]:offsets[i+1]]."""
        L = _lib.load()
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        S = len(byte_offsets) - 1
        nbytes = int(byte_offsets[S] - byte_offsets[0]) if S else 0
        nchars = int(out_offsets[S] - out_offsets[0]) + S if S else 0
        cap = 2 * nbytes + nchars
        if tagged:
            sfx = C.c_uint32(0)
            if L.vpt_predictor_max_tag_suffix(self._h, C.byref(sfx)) != _lib.VPT_OK:
                _raise(_lib.VPT_INVALID_ARGUMENT)
            cap += nchars * int(sfx.value)
        text = np.zeros(max(cap, 1), dtype=np.uint8)
        toff = np.zeros(S + 1, dtype=np.uint64)
        lab = labels if len(labels) else np.zeros(1, dtype=np.uint8)
        if tagged:
            st = L.vpt_write_tagged_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data, lab.ctypes.data,
                                          _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0, text.ctypes.data, cap, toff.ctypes.data)
        else:
            st = L.vpt_write_tokenized_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data,
                                             lab.ctypes.data, text.ctypes.data, cap, toff.ctypes.data)
        if st != _lib.VPT_OK:
            _raise(st)
        return text[:int(toff[S])], toff

    def write_tokenized_batch(self, sentences: Sequence["Sentence"], tagged: bool = False) -> List[str]:
        """write_tokenized_text for many sentences in one launch.  tagged: fill_tags + "/tag" suffixes on the device
        (the sentences' own tags are not touched); otherwise sentences that carry tags take the host writer."""
        if not sentences:
            return []
        if tagged:
            utf8, boff = pack_texts([s._utf8 for s in sentences])
            ooff = np.zeros(len(sentences) + 1, dtype=np.uint64)
            ooff[1:] = np.cumsum([len(s) - 1 for s in sentences])
            labels = np.concatenate([np.asarray(s._boundaries, dtype=np.uint8) for s in sentences]) if int(ooff[-1]) else np.zeros(0, np.uint8)
            text, toff = self.write_tokenized_packed(utf8, boff, ooff, labels, tagged=True)
            raw = bytes(text)
            return [raw[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(len(sentences))]
        utf8, boff = pack_texts([s._utf8 for s in sentences])
        ooff = np.zeros(len(sentences) + 1, dtype=np.uint64)
        ooff[1:] = np.cumsum([len(s) - 1 for s in sentences])
        labels = np.concatenate([np.asarray(s._boundaries, dtype=np.uint8) for s in sentences]) if int(ooff[-1]) else np.zeros(0, np.uint8)
        text, toff = self.write_tokenized_packed(utf8, boff, ooff, labels)
        raw = bytes(text)
        out = []
        for i, s in enumerate(sentences):
            t = raw[int(toff[i]):int(toff[i + 1])].decode("utf-8")
            if s._n_tags:   # "/tag" suffixes: splice them in token by token (rare path, host strings)
                t = s.write_tokenized_text()
            out.append(t)
        return out

    def fill_tags_batch(self, sentences: Sequence["Sentence"]) -> None:
        """Sentence::fill_tags for many sentences in one launch; candidate indices are mapped to the tag strings of
        the tag model whose token equals the token's surface (the LAST one of a repeated token, predictor.rs:466-478)."""
        if not sentences:
            return
        utf8, boff = pack_texts([s._utf8 for s in sentences])
        ooff = np.zeros(len(sentences) + 1, dtype=np.uint64)
        ooff[1:] = np.cumsum([len(s) - 1 for s in sentences])
        labels = np.concatenate([np.asarray(s._boundaries, dtype=np.uint8) for s in sentences]) if int(ooff[-1]) else np.zeros(0, np.uint8)
        scores = models = None
        if self._store_tag_scores:
            tags, scores, models = self.fill_tags_scores_packed(utf8, boff, ooff, labels)
        else:
            tags = self.fill_tags_packed(utf8, boff, ooff, labels)
        nt = tags.shape[1]
        tag_model_list = self._model.tag_models() if scores is not None else None
        if self._tag_models is None:
            self._tag_models = {tm.token: tm for tm in self._model.tag_models()}
        for i, s in enumerate(sentences):
            g0 = int(ooff[i]) + i
            n = len(s)
            s._n_tags = nt
            s._tags = [None] * (n * nt)
            s._tag_scores = []   # sentence.rs:96: Vec<Option<(&[Vec<String>], Vec<i32>)>>, empty unless store_tag_scores
            if scores is not None and nt:
                s._tag_scores = [None] * n
                for e in range(n):
                    mi = int(models[g0 + e])
                    if mi >= 0:
                        tm_ = tag_model_list[mi]
                        s._tag_scores[e] = (tm_.tags, scores[g0 + e, :len(tm_.bias)].tolist())
            if nt == 0:
                continue
            start, valid = 0, True
            for e in range(n):
                b = CharacterBoundary.WordBoundary if e == n - 1 else int(s._boundaries[e])
                if b == CharacterBoundary.Unknown:
                    valid = False
                    continue
                if b != CharacterBoundary.WordBoundary:
                    continue
                if valid:
                    tm = self._tag_models.get(s._text[start:e + 1])
                    if tm is not None:
                        for j in range(min(nt, len(tm.tags))):
                            idx = int(tags[g0 + e, j])
                            if idx >= 0:
                                s._tags[e * nt + j] = tm.tags[j][idx]
                start, valid = e + 1, True

    def predict_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, fullwidth: bool = False,
                       wsconst: Sequence[int] = (), split_linebreaks: bool = False):
        """utf8: uint8[total bytes]; byte_offsets: uint64[S+1].  Returns (scores, labels, out_offsets).
        fullwidth: score the text as KyteaFullwidthFilter would rewrite it (the CLI's default normalisation).
        wsconst: CharacterTypes for KyteaWsConstFilter; split_linebreaks: SplitLinebreaksFilter (labels only)."""
        flags = _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0
        for t in wsconst:
            flags |= _lib.VPT_FLAG_WSCONST(int(t))
        if split_linebreaks:
            flags |= _lib.VPT_FLAG_SPLIT_LINEBREAKS
        L = _lib.load()
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        S = len(byte_offsets) - 1
        ooff = np.zeros(S + 1, dtype=np.uint64)
        st = L.vpt_count_boundaries(utf8.ctypes.data, byte_offsets.ctypes.data, S, ooff.ctypes.data)
        if st != _lib.VPT_OK:
            _raise(st)
        nb = int(ooff[S])
        scores = np.zeros(max(nb, 1), dtype=np.int32)
        labels = np.zeros(max(nb, 1), dtype=np.uint8)
        st = L.vpt_predict_batch_flags(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, scores.ctypes.data,
                                       labels.ctypes.data, ooff.ctypes.data, flags)
        if st != _lib.VPT_OK:
            _raise(st)
        return scores[:nb], labels[:nb], ooff


    def char_types_packed(self, utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, fullwidth: bool = False) -> np.ndarray:
        """Sentence::char_types for a packed batch, from the device: uint8 per char (char c of sentence i at out_offsets[i] + i + c)."""
        utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        S = len(byte_offsets) - 1
        types = np.zeros(int(out_offsets[S]) + S + 1, dtype=np.uint8)
        st = _lib.load().vpt_char_types_batch(self._h, utf8.ctypes.data, byte_offsets.ctypes.data, S, out_offsets.ctypes.data,
                                              _lib.VPT_FLAG_KYTEA_FULLWIDTH if fullwidth else 0, types.ctypes.data)
        if st != _lib.VPT_OK:
            _raise(st)
        return types[:int(out_offsets[S]) + S]


def predict_packed_sharded(predictors: Sequence[Predictor], utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: Optional[np.ndarray] = None,
                           scores: Optional[np.ndarray] = None, labels: Optional[np.ndarray] = None, flags: int = 0, want_scores: bool = True):
    """vpt_predict_batch_sharded: one batch over several predictors (one per GPU), contiguous shards balanced by chars.
    `scores` / `labels` may be preallocated (pinned) arrays.  Returns (scores, labels, out_offsets); with want_scores=False the
    scores stay on the device (NULL scores_out: a tokenizer only needs the labels, and 4 of the 5 bytes per boundary that
    would cross the link back are scores) and None is returned for them."""
    L = _lib.load()
    utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
    byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
    S = len(byte_offsets) - 1
    if out_offsets is None:
        out_offsets = count_boundaries(utf8, byte_offsets)
    out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
    nb = int(out_offsets[S])
    if scores is None and want_scores:
        scores = np.zeros(max(nb, 1), dtype=np.int32)
    if labels is None:
        labels = np.zeros(max(nb, 1), dtype=np.uint8)
    handles = (C.c_void_p * len(predictors))(*[p.handle for p in predictors])
    st = L.vpt_predict_batch_sharded(handles, len(predictors), utf8.ctypes.data, byte_offsets.ctypes.data, S,
                                     scores.ctypes.data if want_scores else None, labels.ctypes.data, out_offsets.ctypes.data, flags)
    if st != _lib.VPT_OK:
        _raise(st)
    return (scores[:nb] if want_scores else None), labels[:nb], out_offsets


def shard_bounds(out_offsets: np.ndarray, n_shards: int) -> np.ndarray:
    """vpt_shard_bounds: contiguous sentence ranges balanced by characters."""
    out_offsets = np.ascontiguousarray(out_offsets, dtype=np.uint64)
    bounds = np.zeros(n_shards + 1, dtype=np.uint64)
    st = _lib.load().vpt_shard_bounds(out_offsets.ctypes.data, len(out_offsets) - 1, n_shards, bounds.ctypes.data)
    if st != _lib.VPT_OK:
        _raise(st)
    return bounds


class PinnedArray:
    """A numpy array over page-locked host memory from vpt_host_alloc (hipHostMalloc): copies to and from the device are
    DMA transfers at the PCIe rate.  Keep the object alive as long as `.array` is in use."""

    def __init__(self, shape, dtype):
        self.array = None
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = C.c_void_p()
        st = _lib.load().vpt_host_alloc(max(n, 1), C.byref(self._ptr))
        if st != _lib.VPT_OK:
            self._ptr = None
            _raise(st)
        buf = (C.c_uint8 * max(n, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        if getattr(self, "_ptr", None):
            self.array = None
            _lib.load().vpt_host_free(self._ptr)
            self._ptr = None


class DeviceBatch:
    """Per-caller workspace for the device-resident entry point (vpt_batch)."""

    def __init__(self, predictor: Predictor, timing: bool = False):
        self._p = predictor
        self._h = C.c_void_p()
        L = _lib.load()
        st = L.vpt_batch_create(predictor.handle, C.byref(self._h))
        if st != _lib.VPT_OK:
            self._h = None
            _raise(st)
        if timing:
            st = L.vpt_batch_set_timing(self._h, 1)
            if st != _lib.VPT_OK:
                _raise(st)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.load().vpt_batch_destroy(self._h)
            self._h = None

    def predict(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int,
                max_sentence_bytes: int, d_scores: int, d_labels: int, stream: int = 0) -> None:
        """All pointers are raw device addresses (e.g. torch.Tensor.data_ptr()); enqueues and returns."""
        st = _lib.load().vpt_predict_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences,
                                                  total_boundaries, max_sentence_bytes, d_scores, d_labels, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def fill_tags(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int, d_labels: int,
                  d_tags: int, stream: int = 0) -> None:
        """Device-resident Sentence::fill_tags for the batch (vpt_fill_tags_batch_device); enqueues and returns.  d_tags = 0 (NULL): the
        tags stay in the workspace as one record per token that has a tag model (what write_tagged reads; expand_tags makes the dense array)."""
        st = _lib.load().vpt_fill_tags_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences,
                                                    total_boundaries, d_labels, d_tags or None, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def expand_tags(self, n_sentences: int, total_boundaries: int, d_tags: int, stream: int = 0) -> None:
        """Sentence::tags() of the batch of the last fill_tags call on this workspace as the dense int32 [chars, n_tags] array
        (vpt_expand_tags_batch_device): None (-1) everywhere but at the last char of a token that has a tag model."""
        st = _lib.load().vpt_expand_tags_batch_device(self._p.handle, self._h, n_sentences, total_boundaries, d_tags, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def fill_tags_scores(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int, d_labels: int,
                         d_tags: int, d_tag_scores: int, d_tag_models: int, stream: int = 0) -> None:
        """fill_tags plus Predictor::store_tag_scores' score vectors and every token's tag model (vpt_fill_tags_scores_batch_device);
        d_tag_scores / d_tag_models may be 0 (NULL)."""
        st = _lib.load().vpt_fill_tags_scores_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences, total_boundaries,
                                                           d_labels, d_tags, d_tag_scores or None, d_tag_models or None, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def write_tokenized(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int, d_labels: int,
                        d_text_out: int, text_capacity: int, d_text_offsets: int, stream: int = 0) -> None:
        """Device-resident write_tokenized_text for the batch (vpt_write_tokenized_batch_device); enqueues and returns."""
        st = _lib.load().vpt_write_tokenized_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences,
                                                          total_boundaries, d_labels, d_text_out, text_capacity, d_text_offsets, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def predict_write(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int, max_sentence_bytes: int,
                      d_scores: int, d_labels: int, d_text_out: int, text_capacity: int, d_text_offsets: int, stream: int = 0) -> None:
        """Predictor::predict and write_tokenized_text (no tags) in ONE scoring launch (vpt_predict_write_batch_device): the tiles of the
        specialised kernel write their tokenized text themselves.  d_scores / d_labels may be 0; enqueues and returns."""
        st = _lib.load().vpt_predict_write_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences, total_boundaries,
                                                        max_sentence_bytes, d_scores or None, d_labels or None, d_text_out, text_capacity,
                                                        d_text_offsets, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def write_tagged(self, d_utf8: int, d_boff: int, d_ooff: int, n_sentences: int, total_boundaries: int, d_labels: int,
                     d_tags: int, d_text_out: int, text_capacity: int, d_text_offsets: int, stream: int = 0) -> None:
        """write_tokenized_text with "/tag" suffixes from the records the fill_tags call on this workspace left for this batch
        (vpt_write_tagged_batch_device; d_tags is not read any more and may be 0); enqueues and returns."""
        st = _lib.load().vpt_write_tagged_batch_device(self._p.handle, self._h, d_utf8, d_boff, d_ooff, n_sentences, total_boundaries,
                                                       d_labels, d_tags or None, d_text_out, text_capacity, d_text_offsets, stream)
        if st != _lib.VPT_OK:
            _raise(st)

    def set_fullwidth(self, enabled: bool) -> None:
        """Score the text as KyteaFullwidthFilter would rewrite it (vpt_batch_set_flags)."""
        st = _lib.load().vpt_batch_set_flags(self._h, _lib.VPT_FLAG_KYTEA_FULLWIDTH if enabled else 0)
        if st != _lib.VPT_OK:
            _raise(st)

    def set_flags(self, flags: int) -> None:
        """VPT_FLAG_* for the calls that follow on this workspace (vpt_batch_set_flags): KyteaFullwidthFilter on the text, KyteaWsConstFilter /
        SplitLinebreaksFilter on the labels."""
        st = _lib.load().vpt_batch_set_flags(self._h, int(flags))
        if st != _lib.VPT_OK:
            _raise(st)

    def set_max_sentence_chars(self, max_chars: int) -> None:
        """Optional tighter bound than max_sentence_bytes (fuller tiles); 0 = unknown."""
        st = _lib.load().vpt_batch_set_max_sentence_chars(self._h, int(max_chars))
        if st != _lib.VPT_OK:
            _raise(st)

    def sync(self) -> None:
        st = _lib.load().vpt_batch_sync(self._h)
        if st != _lib.VPT_OK:
            _raise(st)

    def kernel_ms(self) -> Tuple[float, int]:
        ms, nt = C.c_float(), C.c_uint32()
        st = _lib.load().vpt_batch_kernel_ms(self._h, C.byref(ms), C.byref(nt))
        if st != _lib.VPT_OK:
            _raise(st)
        return ms.value, nt.value


    def last_plan(self) -> dict:
        """How the last predict call cut its batch (vpt_batch_last_plan)."""
        n, tf, kind = C.c_uint32(), C.c_uint32(), C.c_uint32()
        st = _lib.load().vpt_batch_last_plan(self._h, C.byref(n), C.byref(tf), C.byref(kind))
        if st != _lib.VPT_OK:
            _raise(st)
        return {"tiles": n.value, "tile_flat": tf.value, "kind": ("general kernels", "whole-sentence tiles", "cut tiles")[min(kind.value, 2)]}

    def kernel_times(self) -> np.ndarray:
        """Durations (ms) of the timed scoring-kernel launches since the last kernel_ms(), oldest first (at most 256)."""
        out = np.zeros(256, dtype=np.float32)
        n = C.c_size_t(0)
        st = _lib.load().vpt_batch_kernel_times(self._h, out.ctypes.data, len(out), C.byref(n))
        if st != _lib.VPT_OK:
            _raise(st)
        return out[:n.value].copy()

    def phase_cycles(self):
        """Per-phase shader cycles of the specialised kernel (needs VPT_PROFILE_PHASES set before creation)."""
        arr = (C.c_uint64 * 8)()
        st = _lib.load().vpt_batch_phase_cycles(self._h, C.byref(arr))
        if st != _lib.VPT_OK:
            _raise(st)
        return list(arr)

    def node_reads(self):
        """Node reads the specialised kernel issued since the last call (vpt_batch_node_reads; needs VPT_PROFILE_PHASES)."""
        arr = (C.c_uint64 * 8)()
        st = _lib.load().vpt_batch_node_reads(self._h, C.byref(arr))
        if st != _lib.VPT_OK:
            _raise(st)
        return dict(zip(["unigram_nodes", "bigram_nodes", "trigram_nodes", "deep_entries", "deep_rows", "global_type_rows"], [int(x) for x in arr][:6]))


def pack_texts(raws: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    boff = np.zeros(len(raws) + 1, dtype=np.uint64)
    boff[1:] = np.cumsum([len(r) for r in raws], dtype=np.uint64)
    return np.frombuffer(b"".join(raws), dtype=np.uint8), boff


def count_boundaries(utf8: np.ndarray, byte_offsets: np.ndarray) -> np.ndarray:
    utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
    byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
    S = len(byte_offsets) - 1
    ooff = np.zeros(S + 1, dtype=np.uint64)
    st = _lib.load().vpt_count_boundaries(utf8.ctypes.data, byte_offsets.ctypes.data, S, ooff.ctypes.data)
    if st != _lib.VPT_OK:
        _raise(st)
    return ooff


def model_inspect(model_bytes: bytes, predict_tags: bool = False) -> dict:
    mi = _lib.ModelInfo()
    st = _lib.load().vpt_model_inspect(model_bytes, len(model_bytes), int(predict_tags), C.byref(mi))
    if st != _lib.VPT_OK:
        _raise(st)
    return mi.as_dict()
