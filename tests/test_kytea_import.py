"""KyTea model import (vaporetto_amd/kytea.py) against the reference's fixture resources/kytea-model.bin:
the decoded records and the doc-test of kytea_model.rs:401-422."""
import os

import pytest

from oracle import cbind, spec
from vaporetto_amd import kytea
from vaporetto_amd.modelfmt import decode_model, encode_model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kytea-model.bin")


def test_fixture_is_consumed_and_decodes_like_the_reference():
    raw = open(GOLDEN, "rb").read()
    km = kytea.KyteaModel.read(raw)
    # KyteaModel::read stops after the subword dictionary; the fixture carries 8 more zero bytes it never reads
    assert km.consumed == len(raw) - 8 and raw[km.consumed:] == bytes(8)
    c = km.config
    assert c.model_tag.startswith("KyTea 0.4.0 B utf8")
    assert (c.do_ws, c.do_tags, c.n_tags) == (True, True, 2)
    assert (c.char_w, c.char_n, c.type_w, c.type_n, c.dict_n) == (3, 3, 3, 3, 4)
    assert len(c.char_map) == 224
    m = km.to_model_data()
    assert (m.bias, m.char_window_size, m.type_window_size) == (0, 3, 3)
    chars = {d.ngram: d.weights for d in m.char_ngram_model}
    assert chars == {"ぁ": [0, 0, 0, 3343, 0, 0], "ま": [0, 0, 0, 0, 3745, 0]}
    types = {bytes(d.ngram): d.weights for d in m.type_ngram_model}
    assert types == {bytes([3]): [0, 0, -6862, 0, 0, 0], bytes([3, 3]): [3945, 0, -13920, 0, -535],   # truncated to 2W-n+1
                     bytes([3, 5]): [0, 0, 11259, -21250, 0], bytes([5]): [0, 0, 0, 0, 32767, 0]}
    words = {d.word: d.weights for d in m.dict_model}
    assert sorted(words) == sorted(["だ", "だろう", "は", "まぁ", "火星", "猫", "社長", "良い"])
    assert all(w == [0] * (len(k) + 1) for k, w in words.items())     # in_dict = 0 for every word of the fixture


def test_converted_model_tokenises_like_the_doc_test():
    raw = kytea.convert(open(GOLDEN, "rb").read())
    m, used = decode_model(raw)
    assert used == len(raw) and encode_model(m) == raw
    text = "まぁ社長は火星猫だ"
    for labels in (cbind.OraclePredictor(raw).predict(text)[1],
                   [1 if s > 0 else 0 for s in spec.predict_scores(m, text)] if hasattr(spec, "predict_scores") else None):
        if labels is None:
            continue
        assert spec.tokens(text, labels) == ["まぁ", "社長", "は", "火星", "猫", "だ"]   # kytea_model.rs:418-421


def test_truncated_models_are_rejected():
    raw = open(GOLDEN, "rb").read()
    for cut in (10, 300, 1600, len(raw) - 9):
        with pytest.raises(kytea.KyteaFormatError):
            kytea.KyteaModel.read(raw[:cut]).to_model_data()
