"""Vaporetto model file codec (host side, pure Python).

Mirrors `Model::read / read_slice / write / to_vec` of the reference
(/root/reference/vaporetto/src/model.rs:99-153): a fixed magic line followed by the
bincode-2 "standard" configuration encoding of `ModelData` (model.rs:58-70), whose
records are declared in ngram_model.rs:6-27, dict_model.rs:18-22 and model.rs:41-47.

bincode 2 standard config (crate `bincode 2.0.1`, not vendored in the reference):
  * u8            -> one raw byte
  * u16/u32/u64   -> varint: v < 251 -> 1 byte; 251 + LE u16; 252 + LE u32; 253 + LE u64
  * i32           -> zig-zag ((v << 1) ^ (v >> 31)) then the unsigned varint
  * Vec<T>/String -> u64 varint length, then the items / UTF-8 bytes
  * struct        -> fields in declaration order, newtypes add nothing

The predictor itself never uses this module: the C++ loader in csrc/model.cpp parses the
same bytes natively.  This codec exists so that Python callers get `Model.read/write`
and so that tests can build the in-code models of the reference's unit tests.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Tuple

MODEL_MAGIC = b"VaporettoTokenizer 0.5.0\n"  # model.rs:15


class ModelFormatError(ValueError):
    """Raised for a malformed model file (VaporettoError::InvalidModel / DecodeError)."""


# --------------------------------------------------------------------------- records
@dataclass
class NgramData:  # ngram_model.rs:6-9
    ngram: object  # str for char n-grams, bytes for type n-grams
    weights: List[int]


@dataclass
class WordWeightRecord:  # dict_model.rs:18-22
    word: str
    weights: List[int]
    comment: str = ""


@dataclass
class TagWeight:  # ngram_model.rs:15-18
    rel_position: int
    weights: List[int]


@dataclass
class TagNgramData:  # ngram_model.rs:21-24
    ngram: object
    weights: List[TagWeight]


@dataclass
class TagModel:  # model.rs:41-47
    token: str
    tags: List[List[str]]
    char_ngram_model: List[TagNgramData] = field(default_factory=list)
    type_ngram_model: List[TagNgramData] = field(default_factory=list)
    bias: List[int] = field(default_factory=list)


@dataclass
class ModelData:  # model.rs:61-70
    char_ngram_model: List[NgramData] = field(default_factory=list)
    type_ngram_model: List[NgramData] = field(default_factory=list)
    dict_model: List[WordWeightRecord] = field(default_factory=list)
    bias: int = 0
    char_window_size: int = 0
    type_window_size: int = 0
    tag_models: List[TagModel] = field(default_factory=list)


# --------------------------------------------------------------------------- encoder
class _Writer:
    def __init__(self) -> None:
        self.parts: List[bytes] = []

    def u8(self, v: int) -> None:
        if not 0 <= v <= 0xFF:
            raise ValueError("u8 out of range")
        self.parts.append(bytes((v,)))

    def uvar(self, v: int) -> None:
        if v < 0:
            raise ValueError("negative length")
        if v < 251:
            self.parts.append(bytes((v,)))
        elif v < 1 << 16:
            self.parts.append(b"\xfb" + struct.pack("<H", v))
        elif v < 1 << 32:
            self.parts.append(b"\xfc" + struct.pack("<I", v))
        else:
            self.parts.append(b"\xfd" + struct.pack("<Q", v))

    def i32(self, v: int) -> None:
        if not -(1 << 31) <= v < (1 << 31):
            raise ValueError("i32 out of range")
        self.uvar(((v << 1) ^ (v >> 31)) & 0xFFFFFFFF)

    def raw(self, b: bytes) -> None:
        self.uvar(len(b))
        self.parts.append(bytes(b))

    def string(self, s: str) -> None:
        self.raw(s.encode("utf-8"))

    def weights(self, ws) -> None:
        self.uvar(len(ws))
        for w in ws:
            self.i32(int(w))


def _enc_ngrams(w: _Writer, model, is_char: bool) -> None:
    w.uvar(len(model))
    for d in model:
        if is_char:
            w.string(d.ngram)
        else:
            w.raw(bytes(d.ngram))
        w.weights(d.weights)


def _enc_tag_ngrams(w: _Writer, model, is_char: bool) -> None:
    w.uvar(len(model))
    for d in model:
        if is_char:
            w.string(d.ngram)
        else:
            w.raw(bytes(d.ngram))
        w.uvar(len(d.weights))
        for tw in d.weights:
            w.u8(tw.rel_position)
            w.weights(tw.weights)


def encode_model(m: ModelData) -> bytes:
    """`Model::to_vec` (model.rs:99-104)."""
    w = _Writer()
    _enc_ngrams(w, m.char_ngram_model, True)
    _enc_ngrams(w, m.type_ngram_model, False)
    w.uvar(len(m.dict_model))
    for r in m.dict_model:
        w.string(r.word)
        w.weights(r.weights)
        w.string(r.comment)
    w.i32(m.bias)
    w.u8(m.char_window_size)
    w.u8(m.type_window_size)
    w.uvar(len(m.tag_models))
    for t in m.tag_models:
        w.string(t.token)
        w.uvar(len(t.tags))
        for cands in t.tags:
            w.uvar(len(cands))
            for c in cands:
                w.string(c)
        _enc_tag_ngrams(w, t.char_ngram_model, True)
        _enc_tag_ngrams(w, t.type_ngram_model, False)
        w.weights(t.bias)
    return MODEL_MAGIC + b"".join(w.parts)


# --------------------------------------------------------------------------- decoder
class _Reader:
    def __init__(self, buf: bytes, pos: int) -> None:
        self.buf = buf
        self.pos = pos

    def _take(self, n: int) -> bytes:
        if self.pos + n > len(self.buf):
            raise ModelFormatError("unexpected end of model data")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def u8(self) -> int:
        return self._take(1)[0]

    def uvar(self) -> int:
        t = self.u8()
        if t < 251:
            return t
        if t == 251:
            return struct.unpack("<H", self._take(2))[0]
        if t == 252:
            return struct.unpack("<I", self._take(4))[0]
        if t == 253:
            return struct.unpack("<Q", self._take(8))[0]
        raise ModelFormatError("unsupported varint tag %d" % t)

    def i32(self) -> int:
        u = self.uvar()
        if u >= 1 << 32:
            raise ModelFormatError("i32 out of range")
        return (u >> 1) ^ -(u & 1)

    def length(self) -> int:
        n = self.uvar()
        if n > len(self.buf) - self.pos:  # every item takes >= 1 byte
            raise ModelFormatError("length exceeds remaining data")
        return n

    def raw(self) -> bytes:
        return self._take(self.length())

    def string(self) -> str:
        try:
            return self.raw().decode("utf-8")
        except UnicodeDecodeError as e:
            raise ModelFormatError("invalid UTF-8 in model") from e

    def weights(self) -> List[int]:
        return [self.i32() for _ in range(self.length())]


def _dec_ngrams(r: _Reader, is_char: bool) -> List[NgramData]:
    out = []
    for _ in range(r.length()):
        g = r.string() if is_char else r.raw()
        out.append(NgramData(g, r.weights()))
    return out


def _dec_tag_ngrams(r: _Reader, is_char: bool) -> List[TagNgramData]:
    out = []
    for _ in range(r.length()):
        g = r.string() if is_char else r.raw()
        ws = []
        for _ in range(r.length()):
            rel = r.u8()
            ws.append(TagWeight(rel, r.weights()))
        out.append(TagNgramData(g, ws))
    return out


def decode_model(buf: bytes) -> Tuple[ModelData, int]:
    """`Model::read_slice` (model.rs:127-135): returns the model and the bytes consumed."""
    buf = bytes(buf)
    if buf[:len(MODEL_MAGIC)] != MODEL_MAGIC:
        raise ModelFormatError("model version mismatch")
    r = _Reader(buf, len(MODEL_MAGIC))
    m = ModelData()
    m.char_ngram_model = _dec_ngrams(r, True)
    m.type_ngram_model = _dec_ngrams(r, False)
    for _ in range(r.length()):
        word = r.string()
        ws = r.weights()
        m.dict_model.append(WordWeightRecord(word, ws, r.string()))
    m.bias = r.i32()
    m.char_window_size = r.u8()
    m.type_window_size = r.u8()
    for _ in range(r.length()):
        token = r.string()
        tags = []
        for _ in range(r.length()):
            tags.append([r.string() for _ in range(r.length())])
        cm = _dec_tag_ngrams(r, True)
        tm = _dec_tag_ngrams(r, False)
        m.tag_models.append(TagModel(token, tags, cm, tm, r.weights()))
    return m, r.pos
