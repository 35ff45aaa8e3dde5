"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the header
declares, the model loader/table compiler accepts the fixtures and rejects what the reference rejects, and
the Sentence mirror parses like sentence.rs.  No compute entry point is called (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from tests import kat
from vaporetto_amd import _lib, api, build
from vaporetto_amd.modelfmt import ModelData, NgramData, WordWeightRecord, decode_model, encode_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_hip()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "vaporetto_hip.h"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(vpt_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.vpt_version()
    # ... and the Rust binding a maintainer would add names every one of them
    doc = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    assert not [n for n in declared if n not in doc]


def test_model_codec_roundtrip_fixtures():
    for name in ("model.bin", "tantivy_model.bin"):
        raw, m = kat.load_fixture(name)
        assert encode_model(m) == raw  # byte-exact re-encoding pins the bincode layout
        m2, rest = api.Model.read_slice(raw + b"xyz")
        assert rest == b"xyz" and m2.to_vec() == raw


def test_model_inspect_fixtures():
    raw, m = kat.load_fixture("model.bin")
    info = api.model_inspect(raw)
    assert info["n_char_ngrams"] == 2 and info["n_type_ngrams"] == 5 and info["n_dict_words"] == 0
    assert info["n_tag_models"] == 8 and info["bias"] == 0
    assert info["char_window"] == 3 and info["type_window"] == 3
    assert info["type_kind"] == 1  # window table (cache variant): W <= 3, no tags
    # with tag models the reference switches to the BoundaryTag automaton; the sums are the same function of the
    # type window, so W <= 3 keeps the window table / type rows here
    assert api.model_inspect(raw, predict_tags=True)["type_kind"] == 1
    info = api.model_inspect(encode_model(kat.BOUNDARY_KATS[3][2]))  # long dict words
    assert info["max_pattern_chars"] == 5 and info["n_long_nodes"] >= 2


def test_model_errors():
    with pytest.raises(api.VaporettoError) as e:
        api.model_inspect(b"VaporettoTokenizer 0.4.0\n\0\0\0\0\0\0\0")
    assert e.value.kind == "InvalidModel" and "model version mismatch" in str(e.value)
    with pytest.raises(api.VaporettoError) as e:
        api.model_inspect(kat.load_fixture("model.bin")[0][:100])
    assert e.value.kind == "InvalidModel"
    dup = ModelData(type_ngram_model=[NgramData(bytes([3]), [1]), NgramData(bytes([3]), [2])], type_window_size=3)
    with pytest.raises(api.VaporettoError, match="invalid character type n-grams"):  # boundary_scorer_cache.rs:23-24
        api.model_inspect(encode_model(dup))
    empty = ModelData(char_ngram_model=[NgramData("", [1])], char_window_size=3)
    with pytest.raises(api.VaporettoError, match="failed to build the automaton"):  # boundary_scorer.rs:83-87
        api.model_inspect(encode_model(empty))
    long_w = ModelData(char_ngram_model=[NgramData("あ", [1] * 7)], char_window_size=3)
    with pytest.raises(api.VaporettoError, match="longer than"):
        api.model_inspect(encode_model(long_w))
    long_d = ModelData(dict_model=[WordWeightRecord("あい", [1, 2, 3, 4])], char_window_size=3)
    with pytest.raises(api.VaporettoError, match="longer than"):
        api.model_inspect(encode_model(long_d))
    # scorers that the reference drops (None): still a valid model
    assert api.model_inspect(encode_model(ModelData(bias=7)))["type_kind"] == 0


def test_count_boundaries():
    texts = ["まぁ社長は火星猫だ", "a", "🤌🏿x"]
    utf8, boff = api.pack_texts([t.encode() for t in texts])
    assert api.count_boundaries(utf8, boff).tolist() == [0, 8, 8, 10]
    utf8, boff = api.pack_texts([b"ab", b""])
    with pytest.raises(api.VaporettoError, match="must contain at least one character"):
        api.count_boundaries(utf8, boff)
    utf8, boff = api.pack_texts(["A1あ\0ア亜".encode()])
    with pytest.raises(api.VaporettoError, match="must not contain NULL"):
        api.count_boundaries(utf8, boff)


def test_sentence_from_raw():
    """sentence.rs:1311-1477."""
    with pytest.raises(api.VaporettoError) as e:
        api.Sentence.from_raw("")
    assert str(e.value) == "InvalidArgumentError: text: must contain at least one character"
    with pytest.raises(api.VaporettoError) as e:
        api.Sentence.from_raw("A1あ\0ア亜")
    assert str(e.value) == "InvalidArgumentError: text: must not contain NULL"
    s = api.Sentence.from_raw("12345")
    with pytest.raises(api.VaporettoError):
        s.update_raw("")
    assert s.as_raw_text() == " " and s.char_types().tolist() == [6] and len(s.boundaries()) == 0
    assert len(s.boundary_scores()) == 0 and s.char_to_str_pos() == [0, 1]
    s = api.Sentence.from_raw("あ")
    assert s.char_types().tolist() == [3] and len(s.boundaries()) == 0 and s.char_to_str_pos() == [0, 3]
    s.update_raw(kat.PARSE_TEXT)
    assert s.as_raw_text() == kat.PARSE_TEXT
    assert s.char_types().tolist() == kat.PARSE_TYPES
    assert s.char_to_str_pos() == kat.PARSE_CHAR_TO_STR
    assert s.boundaries().tolist() == [2] * 17 and len(s.boundary_scores()) == 0
    assert api.CharacterType.get_type("A") == api.CharacterType.Roman  # sentence.rs:44-49


def test_no_gpu_fails_loudly():
    """There is no CPU fallback: without a device creating a predictor must fail, not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    raw, _ = kat.load_fixture("model.bin")
    with pytest.raises(api.VaporettoError) as e:
        api.Predictor(api.Model.read_slice(raw)[0])
    assert e.value.kind == "Runtime" and "no CPU fallback" in str(e.value)


def test_entry_points_reject_null_handles_without_a_device():
    """Argument errors are host-side: they are reported before any HIP call (so also on a box without a GPU)."""
    L = _lib.load()
    toff = np.zeros(2, np.uint64)
    for call in (lambda: L.vpt_write_tokenized_batch(None, None, None, 1, None, None, None, 0, toff.ctypes.data),
                 lambda: L.vpt_write_tagged_batch(None, None, None, 1, None, None, 0, None, 0, toff.ctypes.data),
                 lambda: L.vpt_fill_tags_batch(None, None, None, 1, None, None, None),
                 lambda: L.vpt_predict_batch(None, None, None, 1, None, None, None),
                 lambda: L.vpt_predictor_max_tag_suffix(None, None)):
        assert call() == _lib.VPT_INVALID_ARGUMENT and "InvalidArgumentError" in _lib.last_error()
