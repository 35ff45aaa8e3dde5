"""Known-answer vectors of the reference's own unit tests, restated as data.

Every case cites the reference test it comes from (paths under /root/reference/vaporetto/src/).
The models are the in-code models those tests construct; expected vectors are the literals the
tests assert.  Cases whose reference test calls a scorer directly (char_scorer.rs / type_scorer.rs
tests) initialise the score buffer with a constant: that constant is the model `bias` here, and
the other scorer is left empty, which gives the same arithmetic through `Predictor::predict`.
"""
from __future__ import annotations

import os

from vaporetto_amd.modelfmt import (ModelData, NgramData, TagModel, TagNgramData, TagWeight,
                                    WordWeightRecord, decode_model)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# CharacterType discriminants (sentence.rs:11-29)
D, R, H, T, K, O = 1, 2, 3, 4, 5, 6


def _n(g, w):
    return NgramData(g, list(w))


def _t(*ts):
    return bytes(ts)


def predictor_test_model() -> ModelData:
    """create_test_model(), predictor.rs:749-838."""
    return ModelData(
        char_ngram_model=[_n("この人", [1, -2, 3, 4]), _n("人だ", [-5, 6, 7, 8, 9])],
        type_ngram_model=[_n(_t(H, H, K), [10, -11, 12, 13]), _n(_t(K, H), [-14, 15, 16, 17, -18])],
        dict_model=[WordWeightRecord("人", [19, 20]), WordWeightRecord("地球", [21, -22, 23])],
        bias=5, char_window_size=3, type_window_size=3,
        tag_models=[
            TagModel(token="人", tags=[["名詞", "接尾辞"], ["ジン", "ヒト"]],
                     char_ngram_model=[TagNgramData("は地球人", [TagWeight(0, [-32, 33, 34, -35])])],
                     type_ngram_model=[TagNgramData(_t(H, K, H), [TagWeight(1, [36, -37, -38, 39])])],
                     bias=[40, 41, 42, 43]),
            TagModel(token="地球", tags=[["名詞"], ["マンホーム", "チキュー"]],
                     char_ngram_model=[TagNgramData("は地球人", [TagWeight(1, [-44, 45])])],
                     type_ngram_model=[], bias=[46, 47]),
        ])


_CHAR_NGRAMS_W3 = [_n("我ら", [1, 2, 3, 4, 5]), _n("全世界", [6, 7, 8, 9]), _n("国民", [10, 11, 12, 13, 14]),
                   _n("世界", [15, 16, 17, 18, 19]), _n("界", [20, 21, 22, 23, 24, 25])]
_DICT_3 = [WordWeightRecord("全世界", [26, 27, 28, 29]), WordWeightRecord("世界", [30, 31, 32]),
           WordWeightRecord("世", [33, 34])]

# (name, reference citation, model, text, expected scores)
BOUNDARY_KATS = [
    ("predict_boundaries", "predictor.rs:840-859", predictor_test_model(), "この人は地球人だ",
     [-22, 54, 58, 43, -54, 68, 48]),
    ("char_add_scores_1", "char_scorer.rs:187-252",
     ModelData(char_ngram_model=_CHAR_NGRAMS_W3, dict_model=_DICT_3, bias=1, char_window_size=3),
     "我らは全世界の国民", [4, 5, 73, 135, 141, 122, 55, 38]),
    ("char_add_scores_2", "char_scorer.rs:254-320",
     ModelData(char_ngram_model=[_n("我ら", [1, 2, 3]), _n("全世界", [4, 5]), _n("国民", [6, 7, 8]),
                                 _n("世界", [9, 10, 11]), _n("界", [12, 13, 14, 15])],
               dict_model=[WordWeightRecord("全世界", [16, 17, 18, 19]), WordWeightRecord("世界", [20, 21, 22]),
                           WordWeightRecord("世", [23, 24])],
               bias=2, char_window_size=2),
     "我らは全世界の国民", [4, 5, 18, 87, 93, 68, 23, 9]),
    ("char_add_scores_3_long_dict", "char_scorer.rs:322-401",
     ModelData(char_ngram_model=_CHAR_NGRAMS_W3,
               dict_model=_DICT_3 + [WordWeightRecord("世界の国民", [35, 36, 37, 38, 39, 40]),
                                     WordWeightRecord("は全世界", [41, 42, 43, 44, 45])],
               bias=3, char_window_size=3),
     "我らは全世界の国民", [6, 48, 117, 215, 223, 206, 95, 79]),
    ("type_add_scores_automaton_w4", "type_scorer.rs:210-256",
     ModelData(type_ngram_model=[_n(_t(K, H), [1, 2, 3, 4, 5, 6, 7]), _n(_t(K, K, K), [8, 9, 10, 11, 12, 13]),
                                 _n(_t(K, K), [14, 15, 16, 17, 18, 19, 20]),
                                 _n(_t(K), [21, 22, 23, 24, 25, 26, 27, 28])],
               bias=1, type_window_size=4),
     "我らは全世界の国民", [87, 135, 144, 174, 182, 192, 202, 148]),
    ("type_add_scores_cache_w3", "type_scorer.rs:258-309",
     ModelData(type_ngram_model=[_n(_t(K, H), [1, 2, 3, 4, 5]), _n(_t(K, K, K), [6, 7, 8, 9]),
                                 _n(_t(K, K), [10, 11, 12, 13, 14]), _n(_t(K), [15, 16, 17, 18, 19, 20])],
               bias=2, type_window_size=3),
     "我らは全世界の国民", [38, 66, 102, 84, 106, 139, 103, 74]),
    ("type_add_scores_cache_w2", "type_scorer.rs:311-363",
     ModelData(type_ngram_model=[_n(_t(K, H), [1, 2, 3]), _n(_t(K, K, K), [4, 5]), _n(_t(K, K), [6, 7, 8]),
                                 _n(_t(K), [9, 10, 11, 12])],
               bias=3, type_window_size=2),
     "我らは全世界の国民", [16, 27, 28, 50, 57, 45, 43, 31]),
]

PREDICT_BOUNDARIES_LABELS = [0, 1, 1, 1, 0, 1, 1]  # predictor.rs:847-858


def char_tag_test_model() -> ModelData:
    """CharScorerBoundaryTag::new(...) of char_scorer.rs:403-502 wrapped in a Model (bias 1)."""
    return ModelData(
        char_ngram_model=[_n("この人", [1, 2, 3, 4]), _n("人だ", [5, 6, 7, 8, 9])],
        dict_model=[WordWeightRecord("人", [10, 11]), WordWeightRecord("火星", [12, 13, 14])],
        bias=1, char_window_size=3,
        tag_models=[
            TagModel("t0", [["a", "b", "c"]], char_ngram_model=[
                TagNgramData("の人", [TagWeight(0, [15, 16, 17]), TagWeight(1, [18, 19, 20])]),
                TagNgramData("人は", [TagWeight(1, [21, 22, 23]), TagWeight(3, [24, 25, 26])]),
                TagNgramData("火星人", [TagWeight(0, [27, 28, 29])])], bias=[1, 1, 1]),
            TagModel("t1", [["a"]], bias=[]),
            TagModel("t2", [["a", "b"]], char_ngram_model=[
                TagNgramData("人は", [TagWeight(0, [27, 28]), TagWeight(3, [29, 30])]),
                TagNgramData("は火星人", [TagWeight(3, [31, 32])])], bias=[1, 1]),
        ])


# char_scorer.rs:503-525: boundary scores with tag models present, then add_tag_scores(token_id, pos)
CHAR_TAG_TEXT = "この人は火星人だ"
CHAR_TAG_BOUNDARY_SCORES = [3, 14, 16, 13, 19, 31, 19]
CHAR_TAG_SCORES = [(0, 2, [37, 39, 41]), (0, 6, [28, 29, 30]), (2, 3, [59, 61])]


def type_tag_test_model() -> ModelData:
    """TypeScorerBoundaryTag::new(...) of type_scorer.rs:365-452 wrapped in a Model (bias 1)."""
    return ModelData(
        type_ngram_model=[_n(_t(H, H, K), [1, 2, 3, 4]), _n(_t(K, H), [5, 6, 7, 8, 9])],
        bias=1, type_window_size=3,
        tag_models=[
            TagModel("t0", [["a", "b", "c"]], type_ngram_model=[
                TagNgramData(_t(H, K), [TagWeight(0, [10, 11, 12]), TagWeight(1, [13, 14, 15])]),
                TagNgramData(_t(K, H), [TagWeight(1, [16, 17, 18]), TagWeight(3, [19, 20, 21])]),
                TagNgramData(_t(K, K, K), [TagWeight(0, [22, 23, 24])])], bias=[1, 1, 1]),
            TagModel("t1", [["a"]], bias=[]),
            TagModel("t2", [["a", "b"]], type_ngram_model=[
                TagNgramData(_t(K, H), [TagWeight(0, [25, 26]), TagWeight(3, [27, 28])]),
                TagNgramData(_t(H, K, K, K), [TagWeight(3, [29, 30])])], bias=[1, 1]),
        ])


# type_scorer.rs:453-473
TYPE_TAG_TEXT = "この人は火星人だ"
TYPE_TAG_BOUNDARY_SCORES = [8, 10, 12, 9, 15, 7, 8]
TYPE_TAG_SCORES = [(0, 2, [27, 29, 31]), (0, 6, [39, 41, 43]), (2, 3, [55, 57])]

def tag_score_kat_through_the_public_path(model: ModelData):
    """The scorer-level vectors above call add_tag_scores(token_id, pos) directly; through the public path (predict, the caller's
    labels, fill_tags with store_tag_scores) a token with that tag model must END at char `pos`.  The tag models are renamed to
    surfaces of the KAT text -- t0 -> "人" (chars 2 and 6), t2 -> "は" (char 3), t1 -> "だ" -- and the labels cut exactly those tokens;
    scores start from the bias [1, ..], the `[1; 8]` the reference's test starts from, so the stored vectors are the KAT values.
    Returns (model, labels, {pos: (model index, scores)})."""
    import copy
    m = copy.deepcopy(model)
    for tm, tok in zip(m.tag_models, ["人", "だ", "は"]):
        tm.token = tok
    #        こ の 人 は 火 星 人 だ      boundaries after chars 0..6
    labels = [0, 1, 1, 1, 0, 1, 1]
    return m, labels


# predictor.rs:861-903: tags of "この人は地球人だ" with create_test_model(), n_tags = 2
PREDICT_TAGS_EXPECTED = [None, None, None, None, "名詞", "ヒト", None, None, None, None,
                         "名詞", "チキュー", "接尾辞", "ジン", None, None]


def load_fixture(name: str):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        raw = f.read()
    model, used = decode_model(raw)
    assert used == len(raw), "model fixture must be consumed to the last byte"
    return raw, model


# Fixture models: expected token splits (+tags) from doc-tests and resources/docs.tok
FIXTURE_SPLITS = [
    # (fixture, text, expected tokens, citation)
    ("model.bin", "まぁ社長は火星猫だ", ["まぁ", "社長", "は", "火星", "猫", "だ"], "predictor.rs:392-401"),
    ("model.bin", "まぁ良いだろう", ["まぁ", "良い", "だろう"], "lib.rs:34-41"),
    ("tantivy_model.bin", "東京特許許可局", ["東京", "特許", "許可", "局"], "vaporetto_tantivy/src/lib.rs:263-296"),
    # KyteaFullwidthFilter already applied (vaporetto_tantivy/src/lib.rs:161-176), 9 tokens expected
    ("tantivy_model.bin", "１２３４５６円🤌🏿", ["１", "２", "３", "４", "５", "６", "円", "🤌", "🏿"],
     "vaporetto_tantivy/src/lib.rs:298-364"),
]

# lib.rs:25-41 / resources/docs.tok
FIXTURE_TAGGED = [
    ("model.bin", "まぁ社長は火星猫だ", "まぁ/名詞/マー 社長/名詞/シャチョー は/助詞/ワ 火星/名詞/カセー 猫/名詞/ネコ だ/助動詞/ダ"),
    ("model.bin", "まぁ良いだろう", "まぁ/副詞/マー 良い/形容詞/ヨイ だろう/助動詞/ダロー"),
]

# SURVEY.md Appendix A (survey-computed with the section-0 spec; regression aid, not reference-pinned)
APPENDIX_SCORES = [
    ("model.bin", "まぁ社長は火星猫だ", [-20845, 18525, -22231, 26247, 41050, -21407, 32767, 26247]),
    ("model.bin", "まぁ良いだろう", [-20845, 22513, -24763, 15910, -20845, -21669]),
    ("tantivy_model.bin", "東京特許許可局", [-21212, 21234, -21211, 21234, -21211, 32767]),
    ("tantivy_model.bin", "１２３４５６円🤌🏿", [36480, 36480, 40155, 40155, 40155, 40155, 36442, 36442]),
]

# sentence.rs:1311-1477 (raw parsing): char types of "Rustで良いプログラミング体験を！"
PARSE_TEXT = "Rustで良いプログラミング体験を！"
PARSE_TYPES = [R, R, R, R, H, K, H, T, T, T, T, T, T, T, K, K, H, O]
PARSE_CHAR_TO_STR = [0, 1, 2, 3, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 34, 37, 40, 43, 46]
PARSE_STR_TO_CHAR = [0, 1, 2, 3, 4, 0, 0, 5, 0, 0, 6, 0, 0, 7, 0, 0, 8, 0, 0, 9, 0, 0, 10, 0, 0, 11, 0,
                     0, 12, 0, 0, 13, 0, 0, 14, 0, 0, 15, 0, 0, 16, 0, 0, 17, 0, 0, 18]
