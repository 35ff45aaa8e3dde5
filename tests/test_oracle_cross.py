"""Two independent oracles must agree: the brute-force "sum over all matches" spec (oracle/spec.py) and the
C restatement of the reference's algorithm (suffix-merged weights + longest match + type table)."""
import pytest

from oracle import cbind, spec
from tests import randmodel
from tests.test_oracle_c_kat import tag_strings
from vaporetto_amd.modelfmt import encode_model


@pytest.mark.parametrize("seed", range(40))
def test_random_models_boundaries(seed):
    alphabet = ["mixed", "tiny", "kana"][seed % 3]
    m = randmodel.rand_model(seed, alphabet=alphabet, max_n=3 + seed % 2, big=(seed % 7 == 0))
    p = cbind.OraclePredictor(encode_model(m))
    for text in randmodel.rand_sentences(seed, m, 25, alphabet=alphabet):
        expected = spec.boundary_scores(m, text)
        scores, labels = p.predict(text)
        assert scores == expected, (seed, text)
        assert labels == spec.boundaries(expected)


@pytest.mark.parametrize("seed", range(20))
def test_random_models_tags(seed):
    alphabet = ["tiny", "mixed"][seed % 2]
    m = randmodel.rand_model(1000 + seed, alphabet=alphabet, n_tag_models=12, max_word=4)
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    for text in randmodel.rand_sentences(seed, m, 15, alphabet=alphabet, max_len=20):
        expected = spec.boundary_scores(m, text, predict_tags=True)
        scores, labels = p.predict(text)
        assert scores == expected, (seed, text)
        tags, nt = p.predict_tags(text)
        assert nt == spec.n_tags(m)
        assert tag_strings(m, text, labels, tags, nt) == spec.fill_tags(m, text, labels), (seed, text)
