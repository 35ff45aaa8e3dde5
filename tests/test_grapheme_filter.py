"""ConcatGraphemeClustersFilter on the host (the CLI's `--wsconst G`): the reference's own tests
(/root/reference/vaporetto_rules/src/sentence_filters/concat_grapheme_clusters.rs:43-88) restated on the Python mirror, and the
packed form against the per-sentence one.  No GPU: the filter only edits boundaries."""
import numpy as np
import pytest

from vaporetto_amd import api

B = api.CharacterBoundary


def _from_tokenized(tokenized: str) -> api.Sentence:
    """Sentence::from_tokenized for plain tokens (no escapes, no tags): a space is a WordBoundary, anything else NotWordBoundary."""
    toks = tokenized.split(" ")
    s = api.Sentence.from_raw("".join(toks))
    b = s.boundaries_mut()
    b[:] = B.NotWordBoundary
    pos = 0
    for t in toks[:-1]:
        pos += len(t)
        b[pos - 1] = B.WordBoundary
    return s


@pytest.mark.parametrize("tokenized,expected", [
    ("‍", "‍"),                                                                                   # rs:44-52
    ("\U0001f468 ‍ \U0001f469 ‍ \U0001f466", "\U0001f468‍\U0001f469‍\U0001f466"),      # rs:54-63  ZWJ sequence
    ("\U0001f44f \U0001f3fd", "\U0001f44f\U0001f3fd"),                                                      # rs:65-73  skin tone modifier
    ("これ は 手 \U0001f44f \U0001f3fd で す", "これ は 手 \U0001f44f\U0001f3fd で す"),                      # rs:75-87
], ids=["no_boundary", "zwj", "color", "combined"])
def test_reference_known_answers(tokenized, expected):
    s = _from_tokenized(tokenized)
    api.ConcatGraphemeClustersFilter().filter(s)
    assert s.write_tokenized_text() == expected


def test_only_boundaries_inside_clusters_change():
    text = "éa\r\nb\U0001f1ef\U0001f1f5がc"          # e + combining acute | a | CR LF | b | regional-indicator pair | か + dakuten | c
    s = api.Sentence.from_raw(text)
    s.boundaries_mut()[:] = B.WordBoundary
    s.boundaries_mut()[2] = B.Unknown                               # (between a and CR: stays)
    api.ConcatGraphemeClustersFilter().filter(s)
    want = [B.NotWordBoundary, B.WordBoundary, B.Unknown, B.NotWordBoundary, B.WordBoundary, B.WordBoundary, B.NotWordBoundary, B.WordBoundary, B.NotWordBoundary, B.WordBoundary]
    assert s.boundaries().tolist() == [int(x) for x in want]
    assert api.ConcatGraphemeClustersFilter.cluster_lengths(text) == [2, 1, 2, 1, 2, 2, 1]


def test_packed_form_equals_per_sentence_form():
    rng = np.random.default_rng(3)
    alphabet = list("あいう漢字abc \r\n") + ["́", "‍", "\U0001f468", "\U0001f469", "\U0001f3fd", "\U0001f44f", "゙", "\U0001f1ef", "\U0001f1f5"]
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in rng.integers(1, 40, 300)] + ["a", "‍", "abc def", "x\r\ny"]
    ooff = np.concatenate([[0], np.cumsum([len(t) - 1 for t in texts])]).astype(np.uint64)
    labels = rng.integers(0, 3, int(ooff[-1])).astype(np.uint8)
    packed = labels.copy()
    f = api.ConcatGraphemeClustersFilter()
    f.filter_packed(texts, ooff, packed)
    for i, t in enumerate(texts):
        s = api.Sentence.from_raw(t)
        s.boundaries_mut()[:] = labels[int(ooff[i]):int(ooff[i + 1])]
        f.filter(s)
        assert np.array_equal(s.boundaries(), packed[int(ooff[i]):int(ooff[i + 1])]), repr(t)
    changed = packed != labels
    assert (packed[changed] == B.NotWordBoundary).all()
