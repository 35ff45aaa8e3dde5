"""ConcatGraphemeClustersFilter on the host (the CLI's `--wsconst G`): the reference's own tests
(/root/reference/vaporetto_rules/src/sentence_filters/concat_grapheme_clusters.rs:43-88) restated on the Python mirror, and the
packed form against the per-sentence one.  No GPU: the filter only edits boundaries."""
import os
import subprocess

import numpy as np
import pytest

from vaporetto_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_generated_tables_name_the_regex_module_they_came_from():
    """the C++ mirror's class tables are a snapshot of the `regex` module's Unicode data, the Python path uses the installed module's \\X
    directly -- the two agree only while the snapshot is of the installed version.  The generated file names its source; a newer `regex` (another
    Unicode version) fails here with what to do, instead of in the random-string comparison below with a code point nobody recognises.  (The
    reference pins unicode-segmentation 1.12.0; neither side can be checked against that crate here -- no GraphemeBreakTest.txt in this image.)"""
    import re
    import regex
    head = open(os.path.join(ROOT, "include", "vaporetto_grapheme_tables.inc"), encoding="utf-8").readline()
    m = re.search(r"`regex` module (\S+) --", head)
    assert m, head
    assert m.group(1) == regex.__version__, ("include/vaporetto_grapheme_tables.inc was generated from regex %s, the installed module is %s: "
                                             "rerun tools/gen_grapheme_tables.py and commit the file" % (m.group(1), regex.__version__))


def test_cpp_mirror_segments_like_the_regex_module(tmp_path):
    """include/vaporetto_grapheme.hpp (UAX #29's rules over generated class tables) against \\X of the `regex` module: the reference's four
    cases, hand-picked rule cases (Hangul jamo, Indic conjuncts, prepend, flags, ZWJ sequences, CR LF, controls) and 4 000 random strings over
    code points of every class."""
    import regex
    exe = str(tmp_path / "grapheme_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "grapheme_test.cpp")])
    rng = np.random.default_rng(29)
    pool = []
    props = ["GCB=CR", "GCB=LF", "GCB=Control", "GCB=Extend", "GCB=ZWJ", "GCB=Regional_Indicator", "GCB=Prepend", "GCB=SpacingMark", "GCB=L", "GCB=V",
             "GCB=T", "GCB=LV", "GCB=LVT", "Extended_Pictographic", "InCB=Consonant", "InCB=Linker", "InCB=Extend"]
    every = "".join(chr(c) for c in range(0x110000) if not 0xD800 <= c <= 0xDFFF and c != 0)
    for pr in props:
        members = [m.group() for m in regex.finditer(r"\p{%s}" % pr, every)]
        pool += [members[int(k)] for k in rng.integers(0, len(members), 12)] + members[:2] + members[-2:]
    pool += list("aあ漢 1") + ["\U0001f468", "\U0001f469", "\U0001f466", "\U0001f3fd", "\u200d", "\u0915", "\u094d", "\u0937", "\u0600", "\u1100", "\u1161", "\u11a8", "\uac00", "\uac01"]
    cases = ["\u200d", "\U0001f468\u200d\U0001f469\u200d\U0001f466", "\U0001f44f\U0001f3fd", "これは手\U0001f44f\U0001f3fdです",
             "\u0915\u094d\u0937", "\u0915\u094d\u200d\u0937", "\u0915\u0937", "\u0600a", "a\u0600", "\U0001f1ef\U0001f1f5\U0001f1fa\U0001f1f8\U0001f1ef",
             "\u1100\u1161\u11a8", "\uac00\u11a8\u1100", "\uac01\u11a8", "a\r\nb\n\rc", "\r\u0301", "a\u0301\u0301b", "\U0001f468\u0301\u200d\U0001f469", "\U0001f468\u200d\u200d\U0001f469", "a\u200d\U0001f469"]
    cases += ["".join(pool[int(k)] for k in rng.integers(0, len(pool), int(n))) for n in rng.integers(1, 24, 4000)]
    esc = lambda t: t.replace("\\", "\\\\").replace("\r", "\\r").replace("\n", "\\n")
    out = subprocess.run([exe], input="\n".join(esc(t) for t in cases).encode("utf-8") + b"\n", stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    for t, line in zip(cases, out):
        want = api.ConcatGraphemeClustersFilter.cluster_lengths(t)
        assert [int(x) for x in line.split()] == want, [hex(ord(c)) for c in t]
