"""The C oracle (oracle/vaporetto_oracle.c, the reference's own algorithm) against every known-answer
vector the reference's tests hold for the predict path, and against the brute-force spec oracle."""
import numpy as np
import pytest

from oracle import cbind, spec
from tests import kat
from vaporetto_amd.modelfmt import ModelData, NgramData, WordWeightRecord, encode_model


@pytest.fixture(scope="module", autouse=True)
def _build():
    cbind.build()


@pytest.mark.parametrize("name,cite,model,text,expected", kat.BOUNDARY_KATS, ids=[k[0] for k in kat.BOUNDARY_KATS])
def test_boundary_kats(name, cite, model, text, expected):
    p = cbind.OraclePredictor(encode_model(model))
    scores, labels = p.predict(text)
    assert scores == expected, cite
    assert labels == [1 if s > 0 else 0 for s in expected]


def tag_strings(model, text, labels, tags, nt):
    """candidate indices -> the flat Option<str> array of Sentence::tags()."""
    flat = []
    bounds = list(labels) + [1]
    for i in range(len(text)):
        start = max([k + 1 for k in range(i) if bounds[k] == 1], default=0)
        for j in range(nt):
            idx = tags[i, j]
            if idx < 0:
                flat.append(None)
            else:
                tm = [t for t in model.tag_models if t.token == text[start:i + 1]][-1]
                flat.append(tm.tags[j][idx])
    return flat


def test_predict_labels_and_tags():
    m = kat.predictor_test_model()
    text = "この人は地球人だ"
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    scores, labels = p.predict(text)
    assert scores == [-22, 54, 58, 43, -54, 68, 48]  # predictor.rs:869
    assert labels == kat.PREDICT_BOUNDARIES_LABELS
    tags, nt = p.predict_tags(text)
    assert nt == 2
    assert tag_strings(m, text, labels, tags, nt) == kat.PREDICT_TAGS_EXPECTED  # predictor.rs:882-902


def test_fill_tags_requires_predict_tags():
    p = cbind.OraclePredictor(encode_model(kat.predictor_test_model()), predict_tags=False)
    with pytest.raises(cbind.OracleError):  # predictor.rs:974-983 (#[should_panic])
        p.predict_tags("この人は地球人だ")


def test_char_tag_scores():
    m = kat.char_tag_test_model()
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    assert p.predict(kat.CHAR_TAG_TEXT)[0] == kat.CHAR_TAG_BOUNDARY_SCORES
    for token_id, pos, expected in kat.CHAR_TAG_SCORES:  # char_scorer.rs:507-524
        z = p.tag_scores_probe(kat.CHAR_TAG_TEXT, 0, token_id, pos, [1] * 8)
        assert z == expected + [1] * (8 - len(expected))


def test_type_tag_scores():
    m = kat.type_tag_test_model()
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    assert p.predict(kat.TYPE_TAG_TEXT)[0] == kat.TYPE_TAG_BOUNDARY_SCORES
    for token_id, pos, expected in kat.TYPE_TAG_SCORES:  # type_scorer.rs:456-472
        z = p.tag_scores_probe(kat.TYPE_TAG_TEXT, 1, token_id, pos, [1] * 8)
        assert z == expected + [1] * (8 - len(expected))


def test_char_weight_merger():
    """char_scorer.rs:169-185, expressed through a model: n-grams get offset -W (=-3), dict words -len."""
    m = ModelData(char_ngram_model=[NgramData("東京都", [1, 2, 3, 4]), NgramData("京都", [2, 4, 6, 8, 10])],
                  dict_model=[WordWeightRecord("京都", [3, 6, 9]), WordWeightRecord("大阪", [4, 8, 12])],
                  char_window_size=3)
    p = cbind.OraclePredictor(encode_model(m))
    assert p.char_patterns() == [(-3, [2, 7, 12, 17, 10]), (-2, [4, 8, 12]), (-3, [3, 9, 15, 21, 10])]


@pytest.mark.parametrize("off_b,expected", [
    (4, (-2, [1, 2, 3, 4, 0, 0, 2, 4, 8])), (2, (-2, [1, 2, 3, 4, 2, 4, 8])), (0, (-2, [1, 2, 5, 8, 8])),
    (-1, (-2, [1, 4, 7, 12])), (-2, (-2, [3, 6, 11, 4])), (-4, (-4, [2, 4, 9, 2, 3, 4])),
    (-5, (-5, [2, 4, 8, 1, 2, 3, 4])), (-7, (-7, [2, 4, 8, 0, 0, 1, 2, 3, 4]))])
def test_positional_weight_add_assign(off_b, expected):
    """predictor.rs:677-747."""
    assert cbind.posw_add_assign(-2, [1, 2, 3, 4], off_b, [2, 4, 8]) == expected


@pytest.mark.parametrize("fixture,text,expected,cite", kat.FIXTURE_SPLITS)
def test_fixture_splits(fixture, text, expected, cite):
    raw, _ = kat.load_fixture(fixture)
    _, labels = cbind.OraclePredictor(raw).predict(text)
    assert spec.tokens(text, labels) == expected, cite


@pytest.mark.parametrize("fixture,text,expected", kat.FIXTURE_TAGGED)
def test_fixture_tags(fixture, text, expected):
    raw, m = kat.load_fixture(fixture)
    p = cbind.OraclePredictor(raw, predict_tags=True)
    _, labels = p.predict(text)
    tags, nt = p.predict_tags(text)
    out, start = [], 0
    for i, b in enumerate(labels + [1]):
        if b == 1:
            tok = text[start:i + 1]
            tm = [t for t in m.tag_models if t.token == tok]
            parts = [tok]
            for j in range(nt):
                if tags[i, j] >= 0:
                    parts.append(tm[-1].tags[j][tags[i, j]])
            out.append("/".join(parts))
            start = i + 1
    assert " ".join(out) == expected


@pytest.mark.parametrize("which,model_fn,text,expected", [(0, kat.char_tag_test_model, kat.CHAR_TAG_TEXT, kat.CHAR_TAG_SCORES),
                                                         (1, kat.type_tag_test_model, kat.TYPE_TAG_TEXT, kat.TYPE_TAG_SCORES)])
def test_stored_tag_scores_reproduce_the_scorer_kats(which, model_fn, text, expected):
    """Predictor::store_tag_scores (predictor.rs:510-514,599-601) through the batch entry point: the vectors of
    char_scorer.rs:508-524 / type_scorer.rs:456-472 as the STORED scores of tokens that end at those chars."""
    m, labels = kat.tag_score_kat_through_the_public_path(model_fn())
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    utf8, boff = np.frombuffer(text.encode("utf-8"), dtype=np.uint8), np.array([0, len(text.encode("utf-8"))], dtype=np.uint64)
    ooff = np.array([0, len(text) - 1], dtype=np.uint64)
    tags, scores, models = p.fill_tags_batch(utf8, boff, ooff, np.array(labels, dtype=np.uint8))
    assert p.tag_score_stride() == 3
    for token_id, pos, want in expected:
        assert models[pos] == token_id and scores[pos, :len(want)].tolist() == want, (token_id, pos)
    assert models.tolist() == [-1, -1, 0, 2, -1, -1, 0, 1]      # "だ" carries t1: one candidate, no scores
    assert tags[7].tolist() == [0] and scores[7].tolist() == [0, 0, 0]


def test_fill_tags_batch_equals_sentence_by_sentence():
    from tests import randmodel
    m = randmodel.rand_model(4, alphabet="kana", wc=3, wt=3, n_tag_models=40, max_word=3, n_char=80, n_dict=80)
    raw = encode_model(m)
    p = cbind.OraclePredictor(raw, predict_tags=True)
    texts = randmodel.rand_sentences(5, m, 300, alphabet="kana", max_len=40) + [t.token * 3 for t in m.tag_models]
    utf8 = np.frombuffer("".join(texts).encode("utf-8"), dtype=np.uint8)
    boff = np.concatenate([[0], np.cumsum([len(t.encode("utf-8")) for t in texts])]).astype(np.uint64)
    _, labels, ooff, _ = p.predict_batch(utf8, boff)
    tags, scores, models = p.fill_tags_batch(utf8, boff, ooff, labels, nthreads=3)
    assert (models >= 0).sum() > 50
    for i, t in enumerate(texts):
        a = int(ooff[i])
        want, nt = p.predict_tags(t, labels=labels[a:a + len(t) - 1])
        assert np.array_equal(tags[a + i:a + i + len(t)], want)
    # the models named are the LAST of a repeated token, and rows without a model hold no scores
    toks = [tm.token for tm in m.tag_models]
    for r in np.nonzero(models >= 0)[0][:200]:
        assert toks.index(toks[models[r]], models[r]) == len(toks) - 1 - toks[::-1].index(toks[models[r]])
    assert not scores[models < 0].any()


@pytest.mark.parametrize("fixture,text,expected", kat.FIXTURE_TAGGED)
def test_tokenized_text_writer_on_the_fixtures(fixture, text, expected):
    """Sentence::write_tokenized_text (sentence.rs:850-886) restated: lib.rs:25-41 / resources/docs.tok."""
    raw, _ = kat.load_fixture(fixture)
    p = cbind.OraclePredictor(raw, predict_tags=True)
    utf8, boff = np.frombuffer(text.encode("utf-8"), dtype=np.uint8), np.array([0, len(text.encode("utf-8"))], dtype=np.uint64)
    _, labels, ooff, _ = p.predict_batch(utf8, boff)
    tags, _, models = p.fill_tags_batch(utf8, boff, ooff, labels)
    out, toff = p.write_tokenized_batch(utf8, boff, ooff, labels, tags, models)
    assert bytes(out).decode("utf-8") == expected and toff.tolist() == [0, len(expected.encode("utf-8"))]
    plain, _ = p.write_tokenized_batch(utf8, boff, ooff, labels)
    assert bytes(plain).decode("utf-8") == " ".join(tok.split("/")[0] for tok in expected.split(" "))


def test_tokenized_text_writer_escapes_and_skips_like_the_reference():
    """sentence.rs:828-848 (doc-test): tokens adjacent to an Unknown boundary are skipped; ' ', '\\' and '/' get a '\\' in surfaces and tags;
    a None between two Some tags is an empty field, trailing Nones are dropped (sentence.rs:866-881)."""
    from vaporetto_amd.modelfmt import ModelData, TagModel
    m = ModelData(bias=5, char_window_size=3, type_window_size=3)
    m.char_ngram_model.append(kat._n("あ", [0, 0, 0, 1, 0, 0]))
    m.tag_models = [TagModel("a/b", [["x y", "q"], [], ["z\\w"]], bias=[3, 1]), TagModel("c", [[], ["only"]], bias=[]), TagModel("d e", [[]], bias=[])]
    p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
    text = "a/bcd ef"
    utf8, boff = np.frombuffer(text.encode("utf-8"), dtype=np.uint8), np.array([0, len(text)], dtype=np.uint64)
    ooff = np.array([0, len(text) - 1], dtype=np.uint64)
    #         a / b | c | d ' ' e | f
    labels = np.array([0, 0, 1, 1, 0, 0, 1], dtype=np.uint8)
    tags, _, models = p.fill_tags_batch(utf8, boff, ooff, labels)
    out, _ = p.write_tokenized_batch(utf8, boff, ooff, labels, tags, models)
    assert bytes(out).decode("utf-8") == "a\\/b/x\\ y//z\\\\w c//only d\\ e f"
    labels[3] = 2     # Unknown between "c" and "d e": both tokens are skipped
    tags, _, models = p.fill_tags_batch(utf8, boff, ooff, labels)
    out, _ = p.write_tokenized_batch(utf8, boff, ooff, labels, tags, models)
    assert bytes(out).decode("utf-8") == "a\\/b/x\\ y//z\\\\w f"


@pytest.mark.parametrize("fixture,text,expected", kat.APPENDIX_SCORES)
def test_appendix_scores(fixture, text, expected):
    raw, _ = kat.load_fixture(fixture)
    assert cbind.OraclePredictor(raw).predict(text)[0] == expected


def test_text_errors():
    p = cbind.OraclePredictor(encode_model(kat.predictor_test_model()))
    for bad in ["", "A1あ\0ア亜"]:  # sentence.rs:1311-1364
        with pytest.raises(cbind.OracleError) as e:
            p.predict(bad)
        assert e.value.status == 2


def test_model_errors():
    with pytest.raises(cbind.OracleError) as e:
        cbind.OraclePredictor(b"VaporettoTokenizer 0.4.0\n" + b"\0" * 8)
    assert e.value.status == 1 and "model version mismatch" in e.value.msg
    dup = ModelData(type_ngram_model=[NgramData(bytes([3]), [1]), NgramData(bytes([3]), [2])], type_window_size=3)
    with pytest.raises(cbind.OracleError) as e:  # boundary_scorer_cache.rs:23-24
        cbind.OraclePredictor(encode_model(dup))
    assert "invalid character type n-grams" in e.value.msg


def test_batch_matches_single():
    raw, _ = kat.load_fixture("model.bin")
    p = cbind.OraclePredictor(raw)
    texts = ["まぁ社長は火星猫だ", "まぁ良いだろう", "あ", "火星猫"]
    enc = [t.encode() for t in texts]
    utf8 = np.frombuffer(b"".join(enc), dtype=np.uint8)
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.uint64)
    for nthreads in (1, 3):
        scores, labels, ooff, _ = p.predict_batch(utf8, offs, nthreads)
        for i, t in enumerate(texts):
            s, l = p.predict(t)
            assert scores[int(ooff[i]):int(ooff[i + 1])].tolist() == s
            assert labels[int(ooff[i]):int(ooff[i + 1])].tolist() == l


def test_double_array_automaton_of_the_baseline_leg_equals_the_checker():
    """bench.py's cpu_baseline walks the char scorer's automaton as a DOUBLE ARRAY (what daachorse is: char_scorer/boundary_scorer.rs:76-99); the
    checker keeps its hash-table automaton.  The double array is built from the checker's automaton (same states, failure links, outputs), and here
    the two are held against each other: the reference's own boundary vectors (predictor.rs:749-859 through the fixture model), random models with
    nested suffix chains and chars outside every pattern (incl. non-BMP ones), one and several threads."""
    import numpy as np
    from tests import kat, randmodel
    from vaporetto_amd import api
    from vaporetto_amd.modelfmt import encode_model
    raw, _ = kat.load_fixture("model.bin")
    texts = ["まぁ社長は火星猫だ", "まぁ良いだろう", "火星猫", "あ", "𠮷野家でまぁ", "abc まぁ"] * 20
    models = [(raw, texts)]
    for seed, alphabet in ((1, "kana"), (2, "mixed"), (3, "tiny"), (4, "mixed")):
        m = randmodel.rand_model(7700 + seed, alphabet=alphabet, wc=3, wt=3, n_char=400, n_dict=600, max_word=9)
        models.append((encode_model(m), randmodel.rand_sentences(seed, m, 700, alphabet="mixed" if seed % 2 == 0 else alphabet, max_len=90)))
    for rawm, tx in models:
        orc = cbind.OraclePredictor(rawm)
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in tx])
        want = orc.predict_batch(utf8, boff)
        for nthreads in (1, 3):
            got = orc.predict_batch(utf8, boff, nthreads=nthreads, double_array=True)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[3] == want[3]


def test_timed_baseline_pool_equals_the_plain_batch():
    """bench.py's cpu_baseline leg (vo_baseline_timed): a pool of pinned workers that lives for all the passes, the tables
    replicated per NUMA node or shared -- the same scores, labels and algorithmic bytes as the plain batch call, a positive time per pass."""
    from tests import randmodel
    from vaporetto_amd import api
    from vaporetto_amd.modelfmt import encode_model
    m = randmodel.rand_model(55, alphabet="kana", wc=3, wt=3, n_char=200, n_dict=200, max_word=8)
    raw = encode_model(m)
    texts = randmodel.rand_sentences(8, m, 3000, alphabet="kana", max_len=80)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o = cbind.OraclePredictor(raw)
    s0, l0, ooff, ab0 = o.predict_batch(utf8, boff)
    for nthreads, replicate, da in ((1, False, True), (3, True, True), (4, False, False), (64, True, True)):
        s1, l1 = np.zeros_like(s0), np.zeros_like(l0)
        secs, ab, nodes = o.baseline_timed(utf8, boff, (s1, l1, ooff), nthreads=nthreads, reps=3, double_array=da, replicate=replicate)
        assert np.array_equal(s0, s1) and np.array_equal(l0, l1) and ab == ab0, (nthreads, replicate, da)
        assert len(secs) == 3 and all(x > 0 for x in secs) and nodes >= 1
