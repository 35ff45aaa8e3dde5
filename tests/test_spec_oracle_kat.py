"""The brute-force spec oracle (oracle/spec.py) against every known-answer vector the reference's
own tests hold for the predict path."""
import pytest

from oracle import spec
from tests import kat


@pytest.mark.parametrize("name,cite,model,text,expected", kat.BOUNDARY_KATS, ids=[k[0] for k in kat.BOUNDARY_KATS])
def test_boundary_kats(name, cite, model, text, expected):
    assert spec.boundary_scores(model, text) == expected, cite


def test_predict_labels():
    m = kat.predictor_test_model()
    ys = spec.boundary_scores(m, "この人は地球人だ")
    assert spec.boundaries(ys) == kat.PREDICT_BOUNDARIES_LABELS  # predictor.rs:847-858
    # with tag prediction on, the type scorer switches to the automaton variant: same scores
    assert spec.boundary_scores(m, "この人は地球人だ", predict_tags=True) == ys  # predictor.rs:869


def test_predict_tags():
    m = kat.predictor_test_model()
    text = "この人は地球人だ"
    b = spec.boundaries(spec.boundary_scores(m, text, predict_tags=True))
    assert spec.fill_tags(m, text, b) == kat.PREDICT_TAGS_EXPECTED  # predictor.rs:882-902


def test_char_tag_scores():
    m = kat.char_tag_test_model()
    assert spec.boundary_scores(m, kat.CHAR_TAG_TEXT, predict_tags=True) == kat.CHAR_TAG_BOUNDARY_SCORES
    for token_id, pos, expected in kat.CHAR_TAG_SCORES:  # char_scorer.rs:507-524
        z = spec.tag_scores_for_token(m, kat.CHAR_TAG_TEXT, m.tag_models[token_id], pos)
        assert z == expected


def test_type_tag_scores():
    m = kat.type_tag_test_model()
    assert spec.boundary_scores(m, kat.TYPE_TAG_TEXT, predict_tags=True) == kat.TYPE_TAG_BOUNDARY_SCORES
    for token_id, pos, expected in kat.TYPE_TAG_SCORES:  # type_scorer.rs:456-472
        z = spec.tag_scores_for_token(m, kat.TYPE_TAG_TEXT, m.tag_models[token_id], pos)
        assert z == expected


@pytest.mark.parametrize("fixture,text,expected,cite", kat.FIXTURE_SPLITS)
def test_fixture_splits(fixture, text, expected, cite):
    _, m = kat.load_fixture(fixture)
    ys = spec.boundary_scores(m, text)
    assert spec.tokens(text, spec.boundaries(ys)) == expected, cite


@pytest.mark.parametrize("fixture,text,expected", kat.FIXTURE_TAGGED)
def test_fixture_tags(fixture, text, expected):
    _, m = kat.load_fixture(fixture)
    b = spec.boundaries(spec.boundary_scores(m, text, predict_tags=True))
    tags = spec.fill_tags(m, text, b)
    nt = spec.n_tags(m)
    out, start = [], 0
    for i, bb in enumerate(b + [1]):
        if bb == 1:
            tok = text[start:i + 1]
            out.append("/".join([tok] + [t for t in tags[i * nt:(i + 1) * nt] if t is not None]))
            start = i + 1
    assert " ".join(out) == expected


@pytest.mark.parametrize("fixture,text,expected", kat.APPENDIX_SCORES)
def test_appendix_scores(fixture, text, expected):
    _, m = kat.load_fixture(fixture)
    assert spec.boundary_scores(m, text) == expected


def test_char_types():
    assert spec.char_types(kat.PARSE_TEXT) == kat.PARSE_TYPES  # sentence.rs:1394-1436


def test_text_errors():
    with pytest.raises(ValueError, match="must contain at least one character"):
        spec.boundary_scores(kat.predictor_test_model(), "")
    with pytest.raises(ValueError, match="must not contain NULL"):
        spec.boundary_scores(kat.predictor_test_model(), "A1あ\0ア亜")
