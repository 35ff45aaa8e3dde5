"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, must reproduce
bit-exactly (i32 scores, u8 labels) the reference's known-answer vectors and the CPU oracle."""
import os

import numpy as np
import pytest

from oracle import cbind, spec
from tests import devmem, kat, randmodel
from vaporetto_amd import api
from vaporetto_amd.modelfmt import ModelData, NgramData, WordWeightRecord, encode_model

pytestmark = pytest.mark.gpu


def make_predictor(model_data, predict_tags=False):
    raw = encode_model(model_data) if not isinstance(model_data, (bytes, bytearray)) else bytes(model_data)
    model, _ = api.Model.read_slice(raw)
    return api.Predictor(model, predict_tags), cbind.OraclePredictor(raw, predict_tags)


def check_batch(pred, orc, texts):
    """GPU batch vs oracle batch on the same packed input; returns the scores."""
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff, nthreads=4)
    assert ooff.tolist() == o_ooff.tolist()
    if not np.array_equal(scores, o_scores):
        bad = int(np.nonzero(scores != o_scores)[0][0])
        sent = int(np.searchsorted(ooff, bad, side="right") - 1)
        raise AssertionError("first mismatch at boundary %d (sentence %d %r): gpu %d oracle %d"
                             % (bad, sent, texts[sent][:40], scores[bad], o_scores[bad]))
    assert np.array_equal(labels, o_labels)
    return scores, labels, ooff


# ------------------------------------------------------------------------------------------------ KATs
@pytest.mark.parametrize("name,cite,model,text,expected", kat.BOUNDARY_KATS, ids=[k[0] for k in kat.BOUNDARY_KATS])
def test_boundary_kats(name, cite, model, text, expected):
    pred, _ = make_predictor(model)
    s = api.Sentence.from_raw(text)
    pred.predict(s)
    assert s.boundary_scores().tolist() == expected, cite
    assert s.boundaries().tolist() == [1 if v > 0 else 0 for v in expected]


def test_predict_boundaries_like_reference():
    """predictor.rs:840-859, written like the reference test."""
    model, _ = api.Model.read_slice(encode_model(kat.predictor_test_model()))
    predictor = api.Predictor(model, False)
    sentence = api.Sentence.from_raw("この人は地球人だ")
    predictor.predict(sentence)
    assert sentence.boundary_scores().tolist() == [-22, 54, 58, 43, -54, 68, 48]
    B = api.CharacterBoundary
    assert [B(b) for b in sentence.boundaries()] == [B.NotWordBoundary, B.WordBoundary, B.WordBoundary, B.WordBoundary,
                                                     B.NotWordBoundary, B.WordBoundary, B.WordBoundary]
    # predict_tags = true switches the type scorer variant; the scores must not change (predictor.rs:869)
    predictor = api.Predictor(model, True)
    predictor.predict(sentence)
    assert sentence.boundary_scores().tolist() == [-22, 54, 58, 43, -54, 68, 48]


@pytest.mark.parametrize("fixture,text,expected,cite", kat.FIXTURE_SPLITS)
def test_fixture_splits(fixture, text, expected, cite):
    raw, _ = kat.load_fixture(fixture)
    predictor = api.Predictor(api.Model.read_slice(raw)[0], False)
    s = api.Sentence.default()
    s.update_raw(text)
    predictor.predict(s)
    assert list(s.iter_tokens()) == expected, cite
    assert s.write_tokenized_text() == " ".join(expected)


@pytest.mark.parametrize("fixture,text,expected", kat.APPENDIX_SCORES)
def test_appendix_scores(fixture, text, expected):
    raw, _ = kat.load_fixture(fixture)
    predictor = api.Predictor(api.Model.read_slice(raw)[0], False)
    s = api.Sentence.from_raw(text)
    predictor.predict(s)
    assert s.boundary_scores().tolist() == expected


def test_sentence_reuse_and_overwrite():
    """`update_raw` + `predict` on a reused sentence overwrites the previous scores (predict/src/main.rs:122-130)."""
    raw, _ = kat.load_fixture("model.bin")
    predictor = api.Predictor(api.Model.read_slice(raw)[0], False)
    s = api.Sentence.default()
    for text, expected in [(t, e) for _, t, e in kat.APPENDIX_SCORES[:2]] * 2:
        s.update_raw(text)
        predictor.predict(s)
        assert s.boundary_scores().tolist() == expected
    s.update_raw("あ")
    predictor.predict(s)
    assert len(s.boundary_scores()) == 0 and len(s.boundaries()) == 0


# ------------------------------------------------------------------------------------------------ random models
@pytest.mark.parametrize("seed", range(24))
def test_random_models_vs_oracle(seed):
    alphabet = ["mixed", "tiny", "kana"][seed % 3]
    m = randmodel.rand_model(seed, alphabet=alphabet, max_n=3 + seed % 2, big=(seed % 7 == 0),
                             max_word=7 + (seed % 4) * 3)
    pred, orc = make_predictor(m)
    texts = randmodel.rand_sentences(seed, m, 300, alphabet=alphabet, max_len=60)
    check_batch(pred, orc, texts)


@pytest.mark.parametrize("wc,wt", [(1, 1), (2, 2), (3, 3), (4, 4), (1, 4), (4, 1), (5, 2), (8, 8), (3, 0), (0, 3), (6, 3), (7, 7), (3, 5), (5, 5), (8, 1), (2, 8)])
def test_window_sizes(wc, wt):
    m = randmodel.rand_model(100 + wc * 10 + wt, alphabet="tiny", wc=wc, wt=wt, max_n=4, n_char=40, n_type=30)
    pred, orc = make_predictor(m)
    info = pred.info()
    assert info["char_window"] == wc
    # every window up to 8 (train/src/main.rs:33-51 lets --charw / --typew be anything) runs on the packed tables and the specialised
    # kernel: windows 1 .. 3 in the rows of window 3, like the distributed models, wider ones in rows of their own; the type n-grams
    # (up to 4 symbols here) as type rows -- in LDS when none is longer than 3, in global memory otherwise
    assert info["packed"] == (1 if wc >= 1 else 0)
    if wc >= 1 and wt >= 1 and m.type_ngram_model:
        assert info["type_rows"] == (1 if max(len(d.ngram) for d in m.type_ngram_model) <= 3 else 2)
    texts = randmodel.rand_sentences(wc * 10 + wt, m, 200, alphabet="tiny", max_len=30)
    texts += randmodel.rand_sentences(wc * 10 + wt + 1, m, 6, alphabet="tiny", min_len=1500, max_len=4000)
    check_batch(pred, orc, texts)
    if wc >= 1:
        for n_first, want in ((50, "whole-sentence tiles"), (len(texts), "cut tiles")):   # i.e. the specialised kernel, both kinds of tile
            batch = api.DeviceBatch(pred)
            utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts[:n_first]])
            ooff = api.count_boundaries(utf8, boff)
            d = [devmem.put(np.concatenate([utf8, np.zeros(32, np.uint8)])), devmem.put(boff), devmem.put(ooff), devmem.zeros(int(ooff[-1]) + 1, np.int32), devmem.zeros(int(ooff[-1]) + 1, np.uint8)]
            batch.predict(d[0].ptr, d[1].ptr, d[2].ptr, n_first, int(ooff[-1]), int(np.max(np.diff(boff.astype(np.int64)))), d[3].ptr, d[4].ptr, devmem.stream())
            batch.sync()
            assert batch.last_plan()["kind"] == want


@pytest.mark.parametrize("seed", range(6))
def test_predict_tags_variant_same_scores(seed):
    """With tag models and predict_tags = true the reference uses the BoundaryTag scorers (automaton instead of
    the type window table); boundary scores must be identical to the oracle in that mode too."""
    m = randmodel.rand_model(500 + seed, alphabet="tiny", n_tag_models=8, max_word=4)
    pred, orc = make_predictor(m, predict_tags=True)
    texts = randmodel.rand_sentences(seed, m, 200, alphabet="tiny", max_len=25)
    check_batch(pred, orc, texts)


# ------------------------------------------------------------------------------------------------ shapes
def test_ragged_and_edge_lengths():
    m = randmodel.rand_model(7, alphabet="mixed", wc=3, wt=3, max_word=12)
    pred, orc = make_predictor(m)
    rng = np.random.RandomState(3)
    alpha = randmodel.ALPHABETS["mixed"]
    texts = []
    for n in [1, 1, 1, 2, 3, 1, 64, 1, 1000, 1, 1, 1017, 1018, 1019, 1020, 1021, 1022, 1023, 1024, 1025, 2, 2047, 2048, 2049,
              5000, 1, 1, 3, 20000, 1] + [1] * 700 + list(rng.randint(1, 200, size=300)):
        texts.append("".join(alpha[i] for i in rng.randint(0, len(alpha), size=n)))
    check_batch(pred, orc, texts)


def test_ascii_and_four_byte_text():
    m = ModelData(char_ngram_model=[NgramData("ab", [1, 2, 3, 4, 5]), NgramData("b", [6, 7, 8, 9, 10, 11]),
                                    NgramData("🤌🏿", [100, 200, 300, 400, 500]), NgramData("🏿", [7, 7, 7, 7, 7, 7])],
                  type_ngram_model=[NgramData(bytes([2, 2]), [1, 1, 1, 1, 1]), NgramData(bytes([6]), [3, 2, 1, 1, 2, 3])],
                  dict_model=[WordWeightRecord("abab", [9, 8, 7, 6, 5]), WordWeightRecord("𠮷🤌🏿𠮷x", [1, 2, 3, 4, 5, 6])],
                  bias=-3, char_window_size=3, type_window_size=3)
    pred, orc = make_predictor(m)
    texts = ["ab" * 40, "b", "abababab", "🤌🏿" * 30, "𠮷🤌🏿𠮷x" * 7, "a🤌b🏿" * 300, "x" * 3000, "🏿" * 2100]
    check_batch(pred, orc, texts)


def test_long_dictionary_words_cross_tile_sized_sentences():
    words = ["あいうえおかきくけこ" * k for k in (1, 2, 5)] + ["漢字" * 9]
    m = ModelData(char_ngram_model=[NgramData("あい", [1, 2, 3, 4, 5])],
                  dict_model=[WordWeightRecord(w, list(range(1, len(w) + 2))) for w in words],
                  bias=1, char_window_size=3)
    pred, orc = make_predictor(m)
    texts = ["あいうえおかきくけこ" * 30, "漢字" * 700, "あいうえおかきくけこ" * 250, "あいうえおかきくけ"]
    check_batch(pred, orc, texts)


# ------------------------------------------------------------------------------------------------ cut tiles
def _documents(seed, m, n_docs, lo, hi, alphabet):
    """Sentences of lo..hi chars whose bytes per char vary 1..4 (so tiles are cut at every kind of byte position), with the
    model's words sprinkled in so that dictionary words and n-grams straddle the cuts."""
    import random
    rng = random.Random(seed)
    words = [w.word for w in m.dict_model] + [g.ngram for g in m.char_ngram_model]
    filler = list(alphabet) + list("ab 1/") + ["🤌", "𠮷"]
    out = []
    for _ in range(n_docs):
        n, parts, have = rng.randint(lo, hi), [], 0
        while have < n:
            w = rng.choice(words) if rng.random() < 0.6 else rng.choice(filler)
            parts.append(w); have += len(w)
        out.append("".join(parts)[:n])
    return out


@pytest.mark.parametrize("force_cut,tile_flat", [(False, None), (True, None), (False, "256"), (True, "61")])
def test_sentences_of_any_length_are_cut_across_tiles(force_cut, tile_flat, monkeypatch):
    """the reference scores a sentence of any length in one loop (char_scorer/boundary_scorer.rs:93-113; its
    tantivy adapter passes a whole document as ONE Sentence, vaporetto_tantivy/src/lib.rs:171-176).  The specialised kernel cuts
    such a batch at fixed flat positions: a tile scores a range of positions and reads a halo of the longest pattern on either
    side -- dictionary words, n-grams and type windows that straddle a cut, cuts inside multi-byte chars' neighbourhood, sentence
    breaks at and next to a cut, 1-char sentences.  VPT_FORCE_CUT_TILES=1 sends batches of SHORT sentences the same way."""
    if force_cut:
        monkeypatch.setenv("VPT_FORCE_CUT_TILES", "1")
    if tile_flat:
        monkeypatch.setenv("VPT_TILE_FLAT", tile_flat)     # small tiles: cuts at many more places (a full GPU cuts small batches finely too)
    alpha = randmodel.ALPHABETS["kana"][:14]
    m = randmodel.rand_model(9100, alphabet=alpha, wc=3, wt=3, max_word=14, n_char=300, n_dict=300, n_type=80)
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    small = devmem.EMULATED and tile_flat == "61"   # (the emulator runs a tile of 61 positions no faster than one of 1 093: a few documents do)
    docs = _documents(1, m, 3 if small else 12, 2000, 9000, alpha) + _documents(2, m, 0 if small else 3, 20000, 30000, alpha)
    shorts = randmodel.rand_sentences(3, m, 900, alphabet=alpha, max_len=70) + ["あ"] * 40 + _documents(4, m, 200, 1, 12, alpha)
    rng = np.random.RandomState(5)
    for texts in ([docs[i] for i in rng.permutation(len(docs))],
                  [(docs + shorts)[i] for i in rng.permutation(len(docs) + len(shorts))],
                  shorts):
        check_batch(pred, orc, texts)
    # every tile boundary in turn next to a sentence break: sentences whose lengths walk over the tile size
    check_batch(pred, orc, [docs[0][:n] for n in range(700, 760)] + [docs[1][:n] for n in range(1690, 1720)])


def test_cut_tiles_with_tags_filters_unaligned_text_and_errors(monkeypatch):
    """Cut tiles carry everything whole-sentence tiles do: the chars left for fill_tags, KyteaFullwidthFilter, the label
    post-filters, a text pointer that is not 16-byte aligned, and the device-side error flags."""
    monkeypatch.setenv("VPT_FORCE_CUT_TILES", "1")
    alpha = randmodel.ALPHABETS["kana"][:10] + list("Ａ１ア")
    m = randmodel.rand_model(9200, alphabet=alpha, wc=3, wt=3, max_word=6, n_char=200, n_dict=150, n_type=60, n_tag_models=30)
    raw = encode_model(m)
    pred, orc = api.Predictor(api.Model.read_slice(raw)[0], True), cbind.OraclePredictor(raw, True)
    texts = _documents(7, m, 6, 1500, 5000, alpha) + randmodel.rand_sentences(8, m, 300, alphabet=alpha, max_len=50) + [t.token * 2 for t in m.tag_models]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    o_scores, o_labels, _, _ = orc.predict_batch(utf8, boff, nthreads=4)
    assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
    # device-resident, the text 5 bytes into its allocation: predict -> fill_tags (takes the chars predict left)
    nb, S, nt = int(ooff[-1]), len(texts), pred.n_tags()
    d_text = devmem.put(np.concatenate([np.full(5, 0x41, np.uint8), utf8, np.zeros(32, np.uint8)]))
    d_boff, d_ooff = devmem.put(boff.astype(np.uint64)), devmem.put(ooff.astype(np.uint64))
    d_scores, d_labels, d_tags = devmem.zeros(nb + 1, np.int32), devmem.zeros(nb + 1, np.uint8), devmem.zeros((nb + S) * nt + 1, np.int32)
    batch = api.DeviceBatch(pred)
    mb = int(np.max(np.diff(boff.astype(np.int64))))
    batch.predict(d_text.ptr + 5, d_boff.ptr, d_ooff.ptr, S, nb, mb, d_scores.ptr, d_labels.ptr, devmem.stream())
    assert batch.last_plan()["kind"] == "cut tiles"
    batch.fill_tags(d_text.ptr + 5, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    assert np.array_equal(d_scores.get(nb), o_scores)
    o_tags, _, _ = orc.fill_tags_batch(utf8, boff, ooff, o_labels, nthreads=2, want_scores=False)
    assert np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), o_tags)
    # the filters: fullwidth on the way in, wsconst / linebreaks on the labels
    plain = api.Predictor(api.Model.read_slice(raw)[0], False)
    fw = api.KyteaFullwidthFilter()
    half = [t.replace("Ａ", "A").replace("１", "1") + "\n" + "ｱ1A" for t in texts[:40]]
    u2, b2 = api.pack_texts([t.encode("utf-8") for t in half])
    got, glab, _ = plain.predict_packed(u2, b2, fullwidth=True, wsconst=[api.CharacterType.Hiragana], split_linebreaks=True)
    u3, b3 = api.pack_texts([fw.filter(t).encode("utf-8") for t in half])
    want, wlab, wooff, _ = cbind.OraclePredictor(raw, False).predict_batch(u3, b3, nthreads=2)
    assert np.array_equal(got, want)
    for i, t in enumerate(half):
        f = fw.filter(t)
        types = api.Sentence.from_raw(f).char_types()
        for k in range(len(f) - 1):
            lab = wlab[int(wooff[i]) + k]
            if types[k] == types[k + 1] == api.CharacterType.Hiragana:
                lab = 0
            if f[k] in "\n\r" or f[k + 1] in "\n\r":
                lab = 1
            assert glab[int(wooff[i]) + k] == lab, (i, k)
    # errors: offsets that promise a different number of chars, a NUL deep inside a long sentence
    bad = ooff.copy(); bad[3:] += 1
    d_bad = devmem.put(bad.astype(np.uint64))
    d_s2, d_l2 = devmem.zeros(int(bad[-1]) + 1, np.int32), devmem.zeros(int(bad[-1]) + 1, np.uint8)
    batch.predict(d_text.ptr + 5, d_boff.ptr, d_bad.ptr, S, int(bad[-1]), mb, d_s2.ptr, d_l2.ptr, devmem.stream())
    with pytest.raises(api.VaporettoError, match="do not match the text"):
        batch.sync()
    nul = list(texts); nul[2] = nul[2][:1234] + "\0" + nul[2][1235:]
    with pytest.raises(api.VaporettoError, match="must not contain NULL"):
        plain.predict_packed(*api.pack_texts([t.encode("utf-8") for t in nul]))
    assert np.array_equal(plain.predict_packed(utf8, boff)[0], o_scores)      # and the workspace is clean again


def test_many_batches_through_one_predictor():
    raw, _ = kat.load_fixture("tantivy_model.bin")
    pred, orc = make_predictor(raw)
    for k in range(5):
        texts = ["東京特許許可局" * (1 + (i + k) % 5) for i in range(50 + 100 * k)]
        check_batch(pred, orc, texts)


# ------------------------------------------------------------------------------------------------ errors
def test_batch_errors():
    raw, _ = kat.load_fixture("model.bin")
    pred, _ = make_predictor(raw)
    utf8, boff = api.pack_texts([b"abc", b"", b"de"])
    with pytest.raises(api.VaporettoError, match="must contain at least one character") as e:
        pred.predict_packed(utf8, boff)
    assert e.value.kind == "InvalidArgument"
    utf8, boff = api.pack_texts(["A1あ\0ア亜".encode()])
    with pytest.raises(api.VaporettoError, match="must not contain NULL"):
        pred.predict_packed(utf8, boff)
    # the predictor stays usable after an error
    s = api.Sentence.from_raw("まぁ良いだろう")
    pred.predict(s)
    assert s.boundary_scores().tolist() == kat.APPENDIX_SCORES[1][2]


def test_device_side_error_flags():
    """Device-resident entry point: the caller's offsets are trusted, so NUL / empty sentences / inconsistent
    offsets are detected by the kernel and reported at sync."""
    raw, _ = kat.load_fixture("model.bin")
    pred, _ = make_predictor(raw)
    batch = api.DeviceBatch(pred)

    def run(raws, ooff=None):
        utf8, boff = api.pack_texts(raws)
        if ooff is None:
            ooff = np.cumsum([0] + [max(len(r.decode()) - 1, 0) for r in raws]).astype(np.uint64)
        d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
        d_boff = devmem.put(boff.astype(np.uint64))
        d_ooff = devmem.put(np.asarray(ooff).astype(np.uint64))
        nb = int(ooff[-1])
        d_scores = devmem.zeros(nb + 1, np.int32)
        d_labels = devmem.zeros(nb + 1, np.uint8)
        batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, len(raws), nb, max(len(r) for r in raws), d_scores.ptr,
                      d_labels.ptr, devmem.stream())
        batch.sync()
        return d_scores.get(nb)

    assert run(["まぁ良いだろう".encode()]).tolist() == kat.APPENDIX_SCORES[1][2]
    with pytest.raises(api.VaporettoError, match="must not contain NULL"):
        run([b"ab\0cd"])
    with pytest.raises(api.VaporettoError, match="must contain at least one character"):
        run([b"ab", b"", b"cd"], ooff=[0, 1, 1, 2])
    with pytest.raises(api.VaporettoError, match="do not match the text"):
        run(["まぁ良いだろう".encode()], ooff=[0, 4])
    assert run(["まぁ良いだろう".encode()]).tolist() == kat.APPENDIX_SCORES[1][2]


def test_concurrent_host_threads_share_a_predictor():
    """`&self` semantics: one predictor, many caller threads (vaporetto_tantivy/src/lib.rs:62-67)."""
    import threading
    raw, _ = kat.load_fixture("tantivy_model.bin")
    pred, orc = make_predictor(raw)
    errors = []

    def work(k):
        try:
            for r in range(5):
                texts = ["東京特許許可局" * (1 + (i + k + r) % 7) for i in range(200)]
                check_batch(pred, orc, texts)
        except Exception as e:  # noqa
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[0]


# ------------------------------------------------------------------------------------------------ both kernels
@pytest.mark.parametrize("force_generic", [False, True])
def test_fast_and_general_kernels_agree_with_oracle(force_generic, monkeypatch):
    """W = 3 models run on the specialised kernel; VPT_FORCE_GENERIC routes the same model through the general
    kernel.  Both must match the oracle on the same ragged batch (long words, many tiny sentences, long ones)."""
    if force_generic:
        monkeypatch.setenv("VPT_FORCE_GENERIC", "1")
    m = randmodel.rand_model(4242, alphabet="kana", wc=3, wt=3, max_word=14, n_char=200, n_dict=300, n_type=60)
    pred, orc = make_predictor(m)
    rng = np.random.RandomState(11)
    texts = randmodel.rand_sentences(5, m, 1500, alphabet="kana", max_len=90)
    texts += randmodel.rand_sentences(6, m, 40, alphabet="kana", min_len=400, max_len=1600)
    texts += randmodel.rand_sentences(7, m, 800, alphabet="kana", max_len=2)
    order = rng.permutation(len(texts))
    check_batch(pred, orc, [texts[i] for i in order])


# ------------------------------------------------------------------------------------------------ packed tables
def test_packed_path_is_used_and_handles_wide_rows():
    """W = 3 BMP models run on the packed tables; rows with values outside their fields take the kPkWide escape."""
    m = ModelData(bias=-7, char_window_size=3, type_window_size=3)
    m.char_ngram_model.append(NgramData("あ", [0, 0, 40000, -5, 1, 2]))
    m.char_ngram_model.append(NgramData("あい", [1, 2, 30000, 4, 5]))
    m.dict_model.append(WordWeightRecord("あい", [7, 30000, -9], ""))
    m.char_ngram_model.append(NgramData("いうえ", [32767, 32767, 3, 4]))
    m.dict_model.append(WordWeightRecord("いうえ", [1, 32767, 5, 32767], ""))
    m.dict_model.append(WordWeightRecord("いうえお", [1, 2, 3, 4, 5], ""))
    m.char_ngram_model.append(NgramData("う", [-1048577, 0, 1048576, -5, 1, 2]))          # just outside 21 bits
    m.char_ngram_model.append(NgramData("え", [-1048576, 1048575, 0, -5, 1, 2]))          # just inside
    m.char_ngram_model.append(NgramData("いう", [5, 2097152, -2097153, 1, 2]))            # just outside 22 bits
    m.char_ngram_model.append(NgramData("うえ", [2097151, -2097152, 2097151, -1, -2097152]))  # just inside
    m.dict_model.append(WordWeightRecord("あいうえおか", [100000, -100000, 3, 4, 5, 6, 2000000000], ""))
    m.dict_model.append(WordWeightRecord("あいうえおかきくけこ", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -70000], ""))
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [5, -6, 7, 8, 9]))
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    texts = ["あいうえおかきくけこ", "ああいいうえお", "いうえ", "あ", "んあいうえおかん", "えおかきあいうえおかきくけこあい"] * 50
    check_batch(pred, orc, texts)


def test_packed_text_with_non_bmp_and_noncharacters():
    m = randmodel.rand_model(5, alphabet="kana", wc=3, wt=2, n_char=200, n_dict=200, max_word=8)
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    pats = [d.ngram for d in m.char_ngram_model] + [r.word for r in m.dict_model]
    texts = []
    for i, p in enumerate(pats):
        for filler in ("𠮷", "￿", "🤌", "￾"):
            texts.append(p[: len(p) // 2] + filler + p + filler + pats[(i + 1) % len(pats)])
    check_batch(pred, orc, texts)


def test_non_bmp_pattern_models_use_the_general_tables():
    """(The name is history: until round 5 one pattern char outside the BMP sent a model to the general kernels.)  Any `String` is a
    pattern to the reference (char_scorer/boundary_scorer.rs:56-89, dict_model.rs:18-50; vaporetto_tantivy/src/lib.rs:298-364 feeds
    non-BMP text): such chars get ids like every other char of the alphabet, found through the side table `xcid`."""
    m = randmodel.rand_model(21, alphabet="mixed", wc=3, wt=3, n_char=120, n_dict=120, max_word=6)
    assert any(ord(c) >= 0x10000 for d in m.char_ngram_model for c in d.ngram) and any(ord(c) >= 0x10000 for r in m.dict_model for c in r.word)
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    check_batch(pred, orc, randmodel.rand_sentences(3, m, 600, alphabet="mixed", max_len=70))


def test_deep_arena_limit_falls_to_the_general_kernels(monkeypatch):
    """a deep arena past the 22 bits the trigram node's child filter leaves a mini-table base is no InvalidModel error: the
    general tables and kernels score the model (the reference accepts any dictionary, dict_model.rs:18-50)."""
    m = randmodel.rand_model(5, alphabet="kana", wc=3, wt=3, n_char=60, n_dict=200, max_word=9)
    monkeypatch.setenv("VPT_DEBUG_KIDS_MAX_BASE", "1")
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 0
    check_batch(pred, orc, randmodel.rand_sentences(8, m, 500, alphabet="kana", max_len=80))
    monkeypatch.delenv("VPT_DEBUG_KIDS_MAX_BASE")
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1


def test_non_bmp_and_ffff_alphabets_on_the_packed_path(monkeypatch):
    """Patterns made of chars outside the BMP only, U+FFFF / U+FFFE as pattern chars, a wide (outside-its-fields) unigram, bigram and
    trigram row that start with a non-BMP char (the replay from the general tables, which keep such a unigram with the short strings),
    long words of them below depth 3, with and without KyteaFullwidthFilter; text with non-BMP chars no pattern holds."""
    m = ModelData(bias=11, char_window_size=3, type_window_size=3)
    m.char_ngram_model.append(NgramData("𠮷", [1, -2, 200000, 4, -5, 6]))                 # wide unigram outside the BMP
    m.char_ngram_model.append(NgramData("𠮷野", [1, 300000, 3, -4, 5]))                   # wide bigram
    m.char_ngram_model.append(NgramData("𠮷野家", [40000, 2, -3, 4]))                     # wide trigram
    m.char_ngram_model.append(NgramData("𩸽", [7, 8, 9, 10, 11, 12]))
    m.char_ngram_model.append(NgramData("𩸽𠮟", [1, 2, 3, 4, 5]))
    m.char_ngram_model.append(NgramData("a𩸽b", [9, 8, 7, 6]))
    m.char_ngram_model.append(NgramData("\uffff", [3, 1, 4, 1, 5, 9]))
    m.char_ngram_model.append(NgramData("\uffff\ufffe", [2, 7, 1, 8, 2]))
    m.dict_model.append(WordWeightRecord("𠮷野家の𩸽", [1, 2, 3, 4, 5, 6], ""))
    m.dict_model.append(WordWeightRecord("𠮟る", [-1, -2, -3], ""))
    m.dict_model.append(WordWeightRecord("🤌🏿🤌🏿🤌🏿🤌", [1, 2, 3, 4, 5, 6, 7, 8], ""))
    m.dict_model.append(WordWeightRecord("𩸽\uffffＡ１", [5, 4, 3, 2, 1], ""))
    m.type_ngram_model.append(NgramData(bytes([5, 5]), [5, -6, 7, 8, 9]))
    m.type_ngram_model.append(NgramData(bytes([6]), [1, 2, 3, 4, 5, 6]))
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    texts = ["𠮷野家の𩸽", "𠮷", "𠮷野", "𠮷野家", "あ𠮷野家の𩸽を𠮟る", "a𩸽b𩸽𠮟𩸽", "\uffff\ufffe\uffff", "🤌🏿🤌🏿🤌🏿🤌🏿🤌", "𩸽\uffffA1𩸽\uffffＡ１",
             "𠀋𠮷𪚲野家", "😀𩸽😀", "𠮟", "る𠮟る𠮟"] * 40
    check_batch(pred, orc, texts)
    fw = api.KyteaFullwidthFilter()
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    n_utf8, n_boff = api.pack_texts([fw.filter(t).encode("utf-8") for t in texts])
    scores, labels, _ = pred.predict_packed(utf8, boff, fullwidth=True)
    o_scores, o_labels, _, _ = orc.predict_batch(n_utf8, n_boff)
    assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
    # the general kernels score the same model the same way
    monkeypatch.setenv("VPT_FORCE_GENERIC", "1")
    check_batch(api.Predictor(api.Model.read_slice(encode_model(m))[0], False), orc, texts[:60])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_dense_packed_tables(seed):
    """A 16-char alphabet with 9 000 patterns: every char continues every char (dense double-array rows, saturated
    child filters), deep tries behind most trigrams, and -- seeds 2, 3 -- text with chars outside the alphabet, whose
    lookups land on other parents' nodes."""
    alpha = [chr(c) for c in range(0x3041, 0x3051)]
    m = randmodel.rand_model(70 + seed, alphabet=alpha, wc=3, wt=3, n_char=3000, n_dict=6000, max_word=12)
    pred, orc = make_predictor(m)
    info = pred.info()
    assert info["packed"] == 1 and info["n_displaced"] == 0
    text_alpha = alpha if seed < 2 else alpha + list("漢字カA9、んー")
    texts = randmodel.rand_sentences(seed, m, 3000, alphabet=text_alpha, max_len=120)
    check_batch(pred, orc, texts)


def test_sparse_double_array_rows_interleave():
    """600 kanji with few patterns each: the rows of different parents share the double arrays slot by slot, so absent
    bigrams and trigrams land on FOREIGN nodes; the key / parent checks must reject every one of them."""
    alpha = [chr(c) for c in range(0x4E00, 0x4E00 + 600)]
    m = randmodel.rand_model(79, alphabet=alpha, wc=3, wt=3, n_char=2500, n_dict=2500, max_word=6)
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    pats = [d.ngram for d in m.char_ngram_model] + [r.word for r in m.dict_model]
    texts = randmodel.rand_sentences(9, m, 2500, alphabet=alpha, max_len=90)
    texts += [p[:2] + q[2:] for p, q in zip(pats[:800], pats[800:1600]) if len(p) >= 2 and len(q) >= 3]
    check_batch(pred, orc, texts)


@pytest.mark.parametrize("wt,force_window", [(3, False), (3, True), (2, False), (1, False), (2, True)])
def test_type_rows_and_window_table_agree_with_oracle(wt, force_window, monkeypatch):
    """Type n-grams of <= 3 symbols are scored from 512 LDS type rows; VPT_FORCE_WINDOW_TABLE routes the same model
    through the 8^(2W) window table (the reference's cache variant).  Both must match the oracle."""
    if force_window:
        monkeypatch.setenv("VPT_FORCE_WINDOW_TABLE", "1")
    m = randmodel.rand_model(500 + wt, alphabet="kana", wc=3, wt=wt, n_char=150, n_dict=150, n_type=120, max_word=8)
    pred, orc = make_predictor(m)
    info = pred.info()
    assert info["packed"] == 1 and info["type_rows"] == 1
    mixed = randmodel.ALPHABETS["mixed"] + randmodel.ALPHABETS["kana"][:8]
    texts = randmodel.rand_sentences(8, m, 2500, alphabet=mixed, max_len=80)
    check_batch(pred, orc, texts)


@pytest.mark.parametrize("wt", [1, 2, 3])
@pytest.mark.parametrize("predict_tags", [False, True])
def test_type_weights_no_window_reads_and_padding_ngrams(wt, predict_tags, monkeypatch):
    """boundary_scorer_cache.rs:41-46 only ever reads w[2W - end]: a type n-gram weight with an index above 2W - n is in
    no window and must add nothing -- on the type rows as on the window table (the rows used to add it).  A
    type n-gram holding code 0 matches the padding in the window table (it is a pattern like any other there), which a
    start-position row cannot express: such a model must be scored from the window table.  With tag models the
    reference matches types with an automaton over char_types (1..6), where a 0 never matches."""
    from vaporetto_amd.modelfmt import TagModel
    for case in ("overlong", "padding"):
        m = randmodel.rand_model(900 + wt, alphabet="kana", wc=3, wt=wt, n_char=60, n_dict=60, n_type=0, max_word=6,
                                 n_tag_models=3 if predict_tags else 0)
        m.type_ngram_model = []
        if case == "overlong" and not predict_tags:   # (with tag models an over-long vector is InvalidModel: DESIGN.md, model contract)
            m.type_ngram_model.append(NgramData(bytes([3]), [1, 10, 100, 1000, 10000, 100000][:min(6, 2 * wt + 3)]))
            m.type_ngram_model.append(NgramData(bytes([3, 5]), [7, 70, 700, 7000, 70000][:min(5, 2 * wt + 2)]))
        elif case == "overlong":
            m.type_ngram_model.append(NgramData(bytes([3]), [1, 10, 100, 1000, 10000, 100000][:2 * wt]))
        else:
            m.type_ngram_model.append(NgramData(bytes([0, 3]), [5, -50, 500, -5000, 50000][:2 * wt - 1]))
            m.type_ngram_model.append(NgramData(bytes([3, 0]), [3, -30, 300, -3000, 30000][:2 * wt - 1]))
            m.type_ngram_model.append(NgramData(bytes([5]), [2, -20, 200, -2000, 20000, -200000][:2 * wt]))
        pred, orc = make_predictor(m, predict_tags)
        info = pred.info()
        assert info["packed"] == 1
        if case == "padding" and not predict_tags:
            assert info["type_rows"] == 0   # the window table, which knows about the padding
        mixed = randmodel.ALPHABETS["kana"][:6] + list("漢字カA9、")
        texts = randmodel.rand_sentences(3, m, 800, alphabet=mixed, max_len=30) + ["あ", "あ漢", "漢あ", "あああ", "カあ"]
        check_batch(pred, orc, texts)
        if case == "overlong" and not predict_tags:
            monkeypatch.setenv("VPT_FORCE_WINDOW_TABLE", "1")
            check_batch(make_predictor(m, predict_tags)[0], orc, texts)
            monkeypatch.delenv("VPT_FORCE_WINDOW_TABLE")


def test_long_type_ngrams_use_global_type_rows(monkeypatch):
    """Type n-grams of 4 .. 6 symbols: type rows in global memory (layout.h, "TYPE ROWS"); with the window table forced, that table."""
    m = randmodel.rand_model(77, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=80, n_type=40, max_word=6)
    m.type_ngram_model.append(NgramData(bytes([3, 3, 3, 3]), [5, -6, 7]))
    m.type_ngram_model.append(NgramData(bytes([3, 5, 3, 3, 6]), [11, -12]))
    m.type_ngram_model.append(NgramData(bytes([3, 3, 5, 3, 3, 6]), [-70000]))
    mixed = randmodel.ALPHABETS["mixed"] + randmodel.ALPHABETS["kana"][:8]
    texts = randmodel.rand_sentences(2, m, 1500, alphabet=mixed, max_len=60)
    pred, orc = make_predictor(m)
    assert pred.info()["type_rows"] == 2 and pred.info()["packed"] == 1
    check_batch(pred, orc, texts)
    monkeypatch.setenv("VPT_FORCE_WINDOW_TABLE", "1")
    pred, orc = make_predictor(m)
    check_batch(pred, orc, texts)


def test_very_long_words_and_compressed_chains():
    import random
    rng = random.Random(5)
    alpha = [chr(c) for c in range(0x3041, 0x3049)]
    m = ModelData(bias=11, char_window_size=3, type_window_size=3)
    words = set()
    base = "".join(rng.choice(alpha) for _ in range(40))
    for n in (4, 5, 9, 12, 13, 14, 15, 21, 22, 30, 40):
        words.add(base[:n])
    for n in (6, 13, 17, 25):
        words.add(base[:n - 1] + "ん")
    for _ in range(60):
        words.add("".join(rng.choice(alpha) for _ in range(rng.randint(4, 28))))
    for w in sorted(words):
        m.dict_model.append(WordWeightRecord(w, [rng.randint(-3000, 3000) for _ in range(len(w) + 1)], ""))
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [5, -6, 7, 8, 9]))
    pred, orc = make_predictor(m)
    assert pred.info()["packed"] == 1
    texts = [base, base[:29] + "ん" + base, "あ" + base[:13] + base[:24] + "ん", base[3:] + base]
    texts += ["".join(rng.choice(sorted(words)) for _ in range(3)) for _ in range(400)]
    check_batch(pred, orc, texts)


# ------------------------------------------------------------------------------------------------ BASELINE configs
@pytest.mark.parametrize("kind,scale,min_len,max_len,n", [
    (1, 0.05, 64, 64, 20000),     # configs[1] shape (bccwj-suw+unidic-like), scaled model
    (2, 0.05, 64, 64, 20000),     # configs[3] shape (jp-0.4.7-5-like: dictionary-heavy)
    (1, 0.05, 8, 512, 6000),      # configs[4] text shape: mixed 8..512-char sentences
    (2, 0.02, 1, 40, 30000),      # many short sentences
])
def test_synthetic_configs_match_oracle(kind, scale, min_len, max_len, n):
    from vaporetto_amd import synth
    raw = synth.synth_model(kind, synth.SEED_BASE + kind, scale)
    utf8, boff = synth.synth_sentences(raw, n, min_len, max_len, seed=synth.SEED_BASE + 7 * kind)
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    assert pred.info()["packed"] == 1
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    o_scores, o_labels, o_ooff, _ = cbind.OraclePredictor(raw).predict_batch(utf8, boff, nthreads=8)
    assert np.array_equal(ooff, o_ooff) and np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)


def test_batch_properties_at_full_config_size():
    """Size-independent properties on a configs[1]-sized batch (100 K x 64 chars): sentences are independent, so a
    permuted batch gives the permuted scores, a batch scored in two halves gives the same scores, and every label is
    the sign of its score."""
    from vaporetto_amd import synth
    raw = synth.synth_model(1, synth.SEED_BASE + 2, 0.1)
    n = 100000
    utf8, boff = synth.synth_sentences(raw, n, 64, 64, seed=synth.SEED_BASE + 2)
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    assert len(scores) == 63 * n and np.array_equal(labels, (scores > 0).astype(np.uint8))
    # two halves
    h = n // 2
    cut = int(boff[h])
    s1, l1, _ = pred.predict_packed(utf8[:cut], boff[:h + 1])
    s2, l2, _ = pred.predict_packed(utf8[cut:], boff[h:] - boff[h])
    assert np.array_equal(np.concatenate([s1, s2]), scores) and np.array_equal(np.concatenate([l1, l2]), labels)
    # a permutation of the sentences
    perm = np.random.RandomState(3).permutation(n)
    b = boff.astype(np.int64)
    text = utf8.reshape(n, 192)            # every synthetic char is 3 bytes, 64 chars per sentence
    assert np.all(np.diff(b) == 192)
    sp, lp, _ = pred.predict_packed(np.ascontiguousarray(text[perm]).reshape(-1), boff)
    assert np.array_equal(sp.reshape(n, 63), scores.reshape(n, 63)[perm])
    # checksum of checksums against the oracle on a 5 % sample of sentences
    idx = np.sort(perm[: n // 20])
    sub = np.ascontiguousarray(text[idx]).reshape(-1)
    sub_boff = (np.arange(len(idx) + 1, dtype=np.uint64) * 192)
    o_scores, _, _, _ = cbind.OraclePredictor(raw).predict_batch(sub, sub_boff, nthreads=8)
    assert int(o_scores.astype(np.int64).sum()) == int(scores.reshape(n, 63)[idx].astype(np.int64).sum())
    assert np.array_equal(o_scores.reshape(-1, 63), scores.reshape(n, 63)[idx])


def test_understated_length_bounds_are_reported():
    """Device entry point: a sentence longer than the caller's max_sentence_bytes / max_sentence_chars is an error
    at sync (never a silent gap in the outputs)."""
    raw, _ = kat.load_fixture("model.bin")
    pred, orc = make_predictor(raw)
    texts = ["まぁ良いだろう" * 300, "まぁ社長は火星猫だ"] * 3      # 2100-char sentences need the long-sentence path
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    nb = int(ooff[-1])
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
    d_boff = devmem.put(boff.astype(np.uint64))
    d_ooff = devmem.put(ooff.astype(np.uint64))
    d_scores = devmem.zeros(nb + 1, np.int32)
    d_labels = devmem.zeros(nb + 1, np.uint8)

    def run(batch, max_bytes):
        batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, len(texts), nb, max_bytes, d_scores.ptr, d_labels.ptr, devmem.stream())
        batch.sync()
        return d_scores.get(nb)

    want = orc.predict_batch(utf8, boff)[0]
    true_bytes = int(np.max(np.diff(boff.astype(np.int64))))
    batch = api.DeviceBatch(pred)
    assert np.array_equal(run(batch, true_bytes), want)
    batch.set_max_sentence_chars(2100)
    assert np.array_equal(run(batch, true_bytes), want)
    batch.set_max_sentence_chars(100)          # understated: the 2100-char sentences do not fit the tiles cut for 100
    with pytest.raises(api.VaporettoError, match="smaller than the longest sentence"):
        run(batch, true_bytes)
    batch.set_max_sentence_chars(0)
    with pytest.raises(api.VaporettoError, match="smaller than the longest sentence"):
        run(batch, 64)                         # understated bytes
    assert np.array_equal(run(batch, true_bytes), want)


def test_tag_enabled_model_with_duplicate_type_ngrams_runs_on_the_packed_path():
    m = randmodel.rand_model(41, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=80, n_type=50, max_word=6, n_tag_models=4)
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [100, -200, 300, 400, -500]))
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [7, 8, 9]))
    pred, orc = make_predictor(m, predict_tags=True)
    info = pred.info()
    assert info["packed"] == 1 and info["type_kind"] == 1
    mixed = randmodel.ALPHABETS["kana"][:10] + list("漢字AZ09、")
    check_batch(pred, orc, randmodel.rand_sentences(4, m, 1500, alphabet=mixed, max_len=60))


# ------------------------------------------------------------------------------------------------ tags
def _tag_strings(m, text, labels, tags, nt):
    """tags[(char, slot)] candidate indices -> the reference's flat Option<str> list (Sentence::tags())."""
    flat, start = [None] * (len(text) * nt), 0
    for i, b in enumerate(list(labels) + [1]):
        if b == 1:
            tms = [t for t in m.tag_models if t.token == text[start:i + 1]]
            if tms:
                for j in range(min(nt, len(tms[-1].tags))):
                    if tags[i, j] >= 0:
                        flat[i * nt + j] = tms[-1].tags[j][tags[i, j]]
            start = i + 1
    return flat


def test_predict_tags_like_reference():
    """predictor.rs:861-903 through the C ABI: same scores, labels and the 16-slot tag array."""
    m = kat.predictor_test_model()
    text = "この人は地球人だ"
    model, _ = api.Model.read_slice(encode_model(m))
    pred = api.Predictor(model, True)
    s = api.Sentence.from_raw(text)
    pred.predict(s)
    assert s.boundary_scores().tolist() == [-22, 54, 58, 43, -54, 68, 48]
    s.fill_tags()
    assert s.n_tags() == 2 and s.tags() == kat.PREDICT_TAGS_EXPECTED


def test_fill_tags_requires_predict_tags_gpu():
    model, _ = api.Model.read_slice(encode_model(kat.predictor_test_model()))
    pred = api.Predictor(model, False)
    s = api.Sentence.from_raw("この人は地球人だ")
    pred.predict(s)
    with pytest.raises(api.VaporettoError, match="predict_tags = false"):   # predictor.rs:974-983 (#[should_panic])
        s.fill_tags()


@pytest.mark.parametrize("fixture,text,expected", kat.FIXTURE_TAGGED)
def test_fixture_tags_gpu(fixture, text, expected):
    raw, m = kat.load_fixture(fixture)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    s = api.Sentence.from_raw(text)
    pred.predict(s)
    s.fill_tags()
    assert s.write_tokenized_text() == expected   # lib.rs:25-41 / resources/docs.tok


@pytest.mark.parametrize("seed", range(10))
def test_random_tag_models_match_oracle(seed):
    """Tag candidate indices for predicted AND caller-edited boundaries (incl. Unknown) equal the oracle's."""
    m = randmodel.rand_model(800 + seed, alphabet="tiny" if seed % 2 else "kana", n_tag_models=25, max_word=4, n_char=40, n_dict=30)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    nt = pred.n_tags()
    rng = np.random.RandomState(seed)
    texts = randmodel.rand_sentences(seed, m, 150, alphabet="tiny" if seed % 2 else "kana", max_len=40)
    texts += [t.token * 3 for t in m.tag_models[:10]] + [t.token for t in m.tag_models]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    for edit in (False, True):
        lab = labels.copy()
        if edit and len(lab):
            k = rng.randint(0, len(lab), size=max(1, len(lab) // 7))
            lab[k] = rng.randint(0, 3, size=len(k))          # NotWordBoundary / WordBoundary / Unknown
        got = pred.fill_tags_packed(utf8, boff, ooff, lab)
        assert got.shape == (int(ooff[-1]) + len(texts), nt)
        for i, t in enumerate(texts):
            a, b = int(ooff[i]), int(ooff[i + 1])
            want, ont = orc.predict_tags(t, labels=lab[a:b])
            assert ont == nt
            g0 = a + i
            assert np.array_equal(got[g0:g0 + len(t)], want), (t, lab[a:b].tolist())


@pytest.mark.parametrize("model_fn,text,expected", [(kat.char_tag_test_model, kat.CHAR_TAG_TEXT, kat.CHAR_TAG_SCORES),
                                                   (kat.type_tag_test_model, kat.TYPE_TAG_TEXT, kat.TYPE_TAG_SCORES)], ids=["char_scorer.rs:508-524", "type_scorer.rs:456-472"])
def test_stored_tag_scores_reproduce_the_scorer_kats(model_fn, text, expected):
    """Predictor::store_tag_scores(true) (predictor.rs:510-514) on the HIP path: the reference's scorer-level tag vectors
    ([37,39,41], [28,29,30], [59,61] / [27,29,31], [39,41,43], [55,57]) are the STORED scores of tokens ending at those chars,
    and Token::tag_candidates (sentence.rs:1218-1250) pairs them with the candidates."""
    m, labels = kat.tag_score_kat_through_the_public_path(model_fn())
    model, _ = api.Model.read_slice(encode_model(m))
    pred = api.Predictor(model, True)
    assert pred.tag_score_stride() == 3
    utf8, boff = api.pack_texts([text.encode("utf-8")])
    ooff = api.count_boundaries(utf8, boff)
    tags, scores, models = pred.fill_tags_scores_packed(utf8, boff, ooff, np.array(labels, dtype=np.uint8))
    for token_id, pos, want in expected:
        assert models[pos] == token_id and scores[pos, :len(want)].tolist() == want, (token_id, pos)
    assert models.tolist() == [-1, -1, 0, 2, -1, -1, 0, 1]
    assert not scores[[0, 1, 4, 5, 7]].any() and tags[7].tolist() == [0]
    # the crate's surface: store_tag_scores -> predict -> (edit boundaries) -> fill_tags -> Token::tag_candidates
    pred.store_tag_scores(True)
    s = api.Sentence.from_raw(text)
    pred.predict(s)
    s.boundaries_mut()[:] = labels
    s.fill_tags()
    by_end = {t.end() - 1: t for t in s.tokens()}
    for token_id, pos, want in expected:
        cands = m.tag_models[token_id].tags[0]
        assert by_end[pos].tag_candidates() == [list(zip(cands, want))], pos
        assert by_end[pos].tags() == [cands[int(np.argmax(want))]]
    assert by_end[7].tag_candidates() == [[("a", 0)]] and by_end[7].surface() == "だ"      # one candidate: score 0 (sentence.rs:1233-1235)
    assert by_end[1].tag_candidates() == [] and by_end[1].tags() == [None]
    pred.store_tag_scores(False)
    s.fill_tags()
    with pytest.raises(AssertionError, match="store_tag_scores"):
        s.tokens()[0].tag_candidates()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_stored_tag_scores_match_oracle_on_random_models(seed):
    """Tags, the stored score vectors and the tag model of every token against the oracle's batch fill_tags -- predicted and edited
    labels (Unknown included), models inside and outside the kernel's record form (more than 16 scores), device-resident too."""
    from vaporetto_amd.modelfmt import TagModel
    import random
    m = randmodel.rand_model(870 + seed, alphabet="tiny" if seed % 2 else "kana", n_tag_models=30, max_word=4, n_char=50, n_dict=40)
    rng = random.Random(seed)
    big = m.tag_models[3]
    big.tags = [["t%d" % k for k in range(11)], ["u%d" % k for k in range(9)], ["solo"]]      # 20 scores: the whole-wave routine
    big.bias = [rng.randint(-500, 500) for _ in range(20)]
    for ng in big.char_ngram_model + big.type_ngram_model:
        for tw in ng.weights:
            tw.weights = [rng.randint(-900, 900) for _ in range(20)]
    m.tag_models.append(TagModel(m.tag_models[5].token, [["late", "later"]], bias=[7, -7]))   # a repeated token: the last model wins
    raw = encode_model(m)
    pred, orc = api.Predictor(api.Model.read_slice(raw)[0], True), cbind.OraclePredictor(raw, True)
    assert pred.tag_score_stride() == orc.tag_score_stride() == 20
    texts = randmodel.rand_sentences(seed, m, 250, alphabet="tiny" if seed % 2 else "kana", max_len=60)
    texts += [t.token * 3 for t in m.tag_models] + [big.token + m.tag_models[5].token + big.token] + [t.token for t in m.tag_models]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    _, labels, ooff = pred.predict_packed(utf8, boff)
    nrng = np.random.RandomState(seed)
    for edit in (0, 1, 2):
        lab = labels.copy()
        if edit == 1:
            k = nrng.randint(0, len(lab), size=len(lab) // 9)
            lab[k] = nrng.randint(0, 3, size=len(k))
        elif edit == 2:
            lab[:] = 0        # every sentence one token: the sentences that ARE a model's token get that model
        tags, scores, models = pred.fill_tags_scores_packed(utf8, boff, ooff, lab)
        o_tags, o_scores, o_models = orc.fill_tags_batch(utf8, boff, ooff, lab, nthreads=2)
        assert np.array_equal(models, o_models) and np.array_equal(tags, o_tags)
        bad = np.nonzero((scores != o_scores).any(axis=1))[0]
        assert len(bad) == 0, (int(bad[0]), int(models[bad[0]]), scores[bad[0]].tolist(), o_scores[bad[0]].tolist())
        assert not (models == 5).any()
        if edit == 2:
            assert (models == 3).any() and (models == len(m.tag_models) - 1).any()
    # device-resident: predict -> fill_tags with scores on one stream; rows without a model are left untouched
    nb, S, nt, stride = int(ooff[-1]), len(texts), pred.n_tags(), pred.tag_score_stride()
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.put(ooff.astype(np.uint64))
    d_scores, d_labels, d_tags = devmem.zeros(nb + 1, np.int32), devmem.zeros(nb + 1, np.uint8), devmem.zeros((nb + S) * nt + 1, np.int32)
    d_ts, d_tm = devmem.put(np.full((nb + S) * stride + 1, 12345, np.int32)), devmem.zeros(nb + S + 1, np.int32)
    batch = api.DeviceBatch(pred)
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr, d_labels.ptr, devmem.stream())
    batch.fill_tags_scores(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, d_ts.ptr, d_tm.ptr, devmem.stream())
    batch.sync()
    o_tags, o_scores, o_models = orc.fill_tags_batch(utf8, boff, ooff, labels)
    got_m, got_s = d_tm.get(nb + S), d_ts.get((nb + S) * stride).reshape(nb + S, stride)
    assert np.array_equal(got_m, o_models) and np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), o_tags)
    lens = np.array([len(t.bias) for t in m.tag_models])
    for r in np.nonzero(o_models >= 0)[0]:
        n = lens[o_models[r]]
        assert got_s[r, :n].tolist() == o_scores[r, :n].tolist() and (got_s[r, n:] == 12345).all()
    assert (got_s[o_models < 0] == 12345).all()


def test_fill_tags_as_two_launches():
    """(The name is history.)  fill_tags is the front end (flat over the batch's chars: a wave takes a run of sentences) that leaves the tokens
    that have a tag model in a queue in HBM, a chained scan over the runs' counts, and a launch of passes over the queue -- for every batch
    size since round 6 (the one-launch kernel, the by-sentence front end and the queue that could overflow are gone: the queue holds a token
    per char).  What it leaves is a RECORD per token with a tag model; the dense array is a scatter of them over a memset.  Both forms against
    the oracle: the dense array asked for with the call, and the one expanded from the records of a call that asked for none."""
    m = randmodel.rand_model(9102, alphabet="tiny", n_tag_models=40, max_word=4, n_char=40, n_dict=30)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    texts = randmodel.rand_sentences(77, m, 700, alphabet="tiny", max_len=90)
    toks = [t.token for t in m.tag_models]
    texts += ["".join(toks[(k + j) % len(toks)] for j in range(30)) for k in range(20)]      # tokens with tag models, back to back: a record per token
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o_scores, o_labels, ooff, _ = orc.predict_batch(utf8, boff)
    o_tags, _, _ = orc.fill_tags_batch(utf8, boff, ooff, o_labels, want_scores=False)
    nb, S, nt = int(ooff[-1]), len(texts), pred.n_tags()
    assert (o_tags >= 0).any()
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.put(ooff.astype(np.uint64))
    d_labels = devmem.put(np.concatenate([o_labels, np.zeros(1, np.uint8)]))
    d_dense = devmem.put(np.full((nb + S) * nt + 1, 77, np.int32)); d_expanded = devmem.put(np.full((nb + S) * nt + 1, 55, np.int32))
    batch = api.DeviceBatch(pred)
    with pytest.raises(api.VaporettoError, match="vpt_fill_tags_batch_device on this workspace"):
        batch.expand_tags(S, nb, d_expanded.ptr, devmem.stream())
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_dense.ptr, devmem.stream())        # the dense array with the call
    batch.sync()
    assert np.array_equal(d_dense.get((nb + S) * nt).reshape(nb + S, nt), o_tags) and d_dense.get()[-1] == 77
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, 0, devmem.stream())                  # records only ...
    batch.expand_tags(S, nb, d_expanded.ptr, devmem.stream())                                                      # ... expanded afterwards
    batch.sync()
    assert np.array_equal(d_expanded.get((nb + S) * nt).reshape(nb + S, nt), o_tags) and d_expanded.get()[-1] == 55
    with pytest.raises(api.VaporettoError, match="vpt_fill_tags_batch_device on this workspace"):                   # another batch's records are nobody's tags
        batch.expand_tags(S - 1, nb, d_expanded.ptr, devmem.stream())
    test_predict_tags_like_reference()
    test_tag_models_inside_and_outside_the_record_form()
    for seed in ((3,) if devmem.EMULATED else (0, 3, 7)):   # (the emulator takes its time)
        test_random_tag_models_match_oracle(seed)
    test_stored_tag_scores_match_oracle_on_random_models(1)


def test_tag_front_end_over_runs_of_sentences():
    """The front-end launch of fill_tags walks RUNS of sentences as consecutive chars, 128 per step: runs of hundreds of one- and two-char
    sentences (more sentence starts in a step than the 64 offsets the lanes hold), sentences that end exactly at a half-step's or a step's
    last char, sentences of several steps, tokens that begin steps before they end, Unknown labels anywhere, several runs per wave --
    tags of every char against the oracle's, for predicted and for edited labels."""
    m = randmodel.rand_model(9100, alphabet="tiny", n_tag_models=30, max_word=4, n_char=40, n_dict=30)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    rng = np.random.RandomState(5)
    alpha = randmodel.ALPHABETS["tiny"]
    def text(n):
        return "".join(alpha[k] for k in rng.randint(0, len(alpha), size=n))
    toks = [t.token for t in m.tag_models]
    lens = [1] * 300 + [2] * 150 + [1, 63, 1, 64, 1, 65, 127, 128, 129, 1, 1, 255, 256, 257, 700, 3, 62, 2, 64, 64, 64, 128, 128, 1]
    lens += list(rng.randint(1, 6, size=400)) + list(rng.randint(1, 300, size=120)) + [1] * 70
    texts = [text(int(n)) for n in lens]
    texts += [toks[k % len(toks)] * int(1 + k % 5) for k in range(60)]      # tokens with tag models, back to back
    texts += ["".join(toks[(k + j) % len(toks)] for j in range(40)) for k in range(6)]
    order = rng.permutation(len(texts))
    texts = [texts[k] for k in order]
    # (a run's 64th sentence starting exactly at the next step's first char: 66 + 62 = 128)
    texts = [text(66)] + [text(1) for _ in range(62)] + [text(40), text(30)] + [text(1) for _ in range(333)] + texts + [toks[k % len(toks)][:1] for k in range(200)] + [text(2) for _ in range(150)]   # blocks of them, too
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    nt = pred.n_tags()
    for edit in (0, 1, 2):
        lab = labels.copy()
        if edit == 1:
            k = rng.randint(0, len(lab), size=len(lab) // 5)
            lab[k] = rng.randint(0, 3, size=len(k))          # NotWordBoundary / WordBoundary / Unknown
        if edit == 2:
            lab[:] = 0                                        # whole sentences as tokens: they begin many steps before they end
            lab[rng.randint(0, len(lab), size=len(lab) // 40)] = 1
        got = pred.fill_tags_packed(utf8, boff, ooff, lab)
        assert got.shape == (int(ooff[-1]) + len(texts), nt)
        for i, t in enumerate(texts):
            a, b = int(ooff[i]), int(ooff[i + 1])
            want, _ = orc.predict_tags(t, labels=lab[a:b])
            g0 = a + i
            assert np.array_equal(got[g0:g0 + len(t)], want), (edit, i, len(t))


@pytest.mark.parametrize("shape", ["only one-char sentences", "one-char sentences at the end", "one sentence of many steps, then one-char ones"])
def test_fill_tags_front_end_where_the_labels_run_out(shape, monkeypatch):
    """A one-char sentence has no label, so a batch (or its tail) of them has none: the front end's runs there have nothing to read in `labels`
    (an empty array for the first shape) and every char ends a token -- the densest the records get: one per char.  Tags against the oracle."""
    m = randmodel.rand_model(9410, alphabet="tiny", n_tag_models=30, max_word=4, n_char=40, n_dict=30)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    rng = np.random.RandomState(11)
    alpha = randmodel.ALPHABETS["tiny"]
    ones = [t.token[:1] for t in m.tag_models] + [alpha[k] for k in rng.randint(0, len(alpha), size=40)]
    def text(n):
        return "".join(alpha[k] for k in rng.randint(0, len(alpha), size=n))
    singles = [ones[k % len(ones)] for k in range(700)]
    if shape == "only one-char sentences":
        texts = singles
    elif shape == "one-char sentences at the end":
        texts = [text(int(n)) for n in rng.randint(1, 90, size=60)] + singles
    else:
        texts = [text(1000)] + singles[:300]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    _, labels, ooff = pred.predict_packed(utf8, boff)
    got = pred.fill_tags_packed(utf8, boff, ooff, labels)
    assert got.shape == (int(ooff[-1]) + len(texts), pred.n_tags())
    for i, t in enumerate(texts):
        a, b = int(ooff[i]), int(ooff[i + 1])
        want, _ = orc.predict_tags(t, labels=labels[a:b])
        assert np.array_equal(got[a + i:a + i + len(t)], want), (i, len(t))


@pytest.mark.parametrize("n_slots", [1, 2, 3, 4, 5])
def test_fill_tags_front_end_stores_for_any_tag_count_and_alignment(n_slots, monkeypatch):
    """(The name is history: until round 6 the front end stored a char's None entries, one instance per tag count; the dense array is a memset
    and a scatter of the records now, for any count and alignment.)  Models with
    exactly `n_slots` tag slots, sentences of many steps: tags against the oracle through the host path; through the device path with the array
    16-byte aligned and 4 bytes off, the words around it untouched; with the models' indices asked for (vpt_fill_tags_scores_batch_device)."""
    import random
    from vaporetto_amd.modelfmt import TagModel, TagNgramData, TagWeight
    m = randmodel.rand_model(9300 + n_slots, alphabet="tiny", n_tag_models=0, max_word=4, n_char=40, n_dict=30)
    rng = random.Random(77 + n_slots)
    alpha = randmodel.ALPHABETS["tiny"]
    seen = set()
    while len(m.tag_models) < 24:
        tok = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 3)))
        if tok in seen:
            continue
        seen.add(tok)
        k = n_slots if len(m.tag_models) < 6 else rng.randint(1, n_slots)
        slots = [["t%d" % j for j in range(rng.randint(0, 4))] for _ in range(k)]
        zlen = sum(len(x) for x in slots if len(x) >= 2)
        tm = TagModel(tok, slots, bias=randmodel.rand_weights(rng, zlen))
        for _ in range(rng.randint(0, 3)):
            g = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 2))) + tok + "".join(rng.choice(alpha) for _ in range(rng.randint(0, 2)))
            tm.char_ngram_model.append(TagNgramData(g, [TagWeight(r, randmodel.rand_weights(rng, zlen)) for r in sorted({rng.randint(0, m.char_window_size) for _ in range(2)})]))
        m.tag_models.append(tm)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    nt = pred.n_tags()
    assert nt == n_slots
    toks = [t.token for t in m.tag_models]
    def text(n):
        return "".join(rng.choice(toks) if rng.random() < 0.3 else rng.choice(alpha) for _ in range(n))
    texts = [text(n) for n in [1, 2, 63, 64, 65, 128, 129, 300, 700, 1500] + [rng.randint(1, 200) for _ in range(120)]]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    _, labels, ooff = pred.predict_packed(utf8, boff)
    got = pred.fill_tags_packed(utf8, boff, ooff, labels)
    nb, S = int(ooff[-1]), len(texts)
    for i, t in enumerate(texts):
        a, b = int(ooff[i]), int(ooff[i + 1])
        want, _ = orc.predict_tags(t, labels=labels[a:b])
        assert np.array_equal(got[a + i:a + i + (b - a + 1)], want), (i, len(t))
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
    d_boff, d_ooff = devmem.put(boff.astype(np.uint64)), devmem.put(ooff.astype(np.uint64))
    d_labels = devmem.put(np.concatenate([labels, np.zeros(1, np.uint8)]))
    n_words = (nb + S) * nt
    for off_words in (4, 1, 5):   # 16 bytes into the allocation (aligned), 4 and 20 bytes (not)
        d_tags = devmem.put(np.full(n_words + 16, 777, np.int32))
        batch = api.DeviceBatch(pred)
        batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr + 4 * off_words, devmem.stream())
        batch.sync()
        w = d_tags.get(n_words + 16)
        assert np.array_equal(w[off_words:off_words + n_words].reshape(nb + S, nt), got), off_words
        assert np.all(w[:off_words] == 777) and np.all(w[off_words + n_words:] == 777), off_words


@pytest.mark.parametrize("wc,wt", [(3, 6), (5, 5), (8, 8), (4, 2), (2, 7)])
def test_tag_models_under_wide_windows(wc, wt):
    """Tag n-grams end up to `window` chars past their token (tag_trainer.rs:79-103).  The fast pass keeps p - 11 .. p + 4 of a token's context:
    a model with a TYPE tag n-gram further out (type windows above 4) must take the whole-wave routine -- round 4's fuzz over every window found
    such models tagged from a window that did not hold the n-gram.  Boundaries, tags and tag scores against the oracle."""
    for seed in range(4):
        m = randmodel.rand_model(7300 + 10 * wc + wt + 100 * seed, alphabet=["kana", "tiny"][seed % 2], wc=wc, wt=wt, max_n=4, n_char=60, n_dict=60, n_type=40,
                                 n_tag_models=12, max_word=4)
        raw = encode_model(m)
        pred = api.Predictor(api.Model.read_slice(raw)[0], True)
        orc = cbind.OraclePredictor(raw, True)
        texts = randmodel.rand_sentences(seed, m, 150, alphabet=["kana", "tiny"][seed % 2], max_len=40)
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
        scores, labels, ooff = pred.predict_packed(utf8, boff)
        o_scores, o_labels, _, _ = orc.predict_batch(utf8, boff)
        assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
        got = pred.fill_tags_packed(utf8, boff, ooff, labels)
        for i, t in enumerate(texts):
            a, b = int(ooff[i]), int(ooff[i + 1])
            want, _ = orc.predict_tags(t, labels=labels[a:b])
            assert np.array_equal(got[a + i:a + i + len(t)], want), (seed, t)


def test_tag_models_inside_and_outside_the_record_form():
    """The tag kernel's fast path checks whole n-grams from 32-byte records (<= 12 BMP symbols, <= 16 scores per model);
    models outside that form -- an n-gram of 14 chars, a non-BMP n-gram, 24 scores -- take the whole-wave routine.  Both kinds
    mixed in the same sentences, many tokens with models per 64-char step, long sentences (several steps, tokens that
    cross a step), predicted and edited labels."""
    from vaporetto_amd.modelfmt import TagModel, TagNgramData, TagWeight
    import random
    rng = random.Random(7)
    alpha = randmodel.ALPHABETS["kana"][:6]
    m = randmodel.rand_model(950, alphabet=alpha, wc=3, wt=3, n_char=60, n_dict=40, max_word=3, n_tag_models=0)
    toks = sorted({"".join(rng.choice(alpha) for _ in range(rng.randint(1, 2))) for _ in range(30)})

    def w(n):
        return [rng.randint(-3000, 3000) for _ in range(n)]
    for i, tok in enumerate(toks):
        kind = i % 5
        slots = [["a", "b", "c"], ["x", "y"]] if kind != 3 else [["t%d" % k for k in range(8)] for _ in range(3)]   # 5 or 24 scores
        zlen = sum(len(s_) for s_ in slots if len(s_) >= 2)
        tm = TagModel(tok, slots, bias=w(zlen))
        for _ in range(rng.randint(2, 8)):
            left = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 2)))
            right = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 2)))
            tm.char_ngram_model.append(TagNgramData(left + tok + right, [TagWeight(r, w(zlen)) for r in sorted({rng.randint(0, 3), len(right)})]))
        for _ in range(rng.randint(0, 4)):
            tm.type_ngram_model.append(TagNgramData(bytes(rng.choice([3, 3, 5, 4]) for _ in range(rng.randint(1, 4))),
                                                    [TagWeight(rng.randint(0, 3), w(zlen))]))
        if kind == 1:   # 14 chars: more than a record holds
            tm.char_ngram_model.append(TagNgramData("".join(rng.choice(alpha) for _ in range(13 - len(tok))) + tok + alpha[0], [TagWeight(1, w(zlen))]))
        if kind == 2:   # a non-BMP symbol
            tm.char_ngram_model.append(TagNgramData(tok + "𠮷", [TagWeight(1, w(zlen))]))
        m.tag_models.append(tm)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    nt = pred.n_tags()
    assert nt == 3
    texts = ["".join(rng.choice(alpha + ["𠮷"] if rng.random() < 0.05 else alpha) for _ in range(rng.choice([3, 20, 64, 65, 130, 300]))) for _ in range(120)]
    texts += ["".join(rng.choice(toks) for _ in range(40)) for _ in range(20)]          # dense in tokens with models
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    nrng = np.random.RandomState(3)
    for edit in (0, 1, 2):
        lab = labels.copy()
        if edit == 1:
            lab[:] = 1                                   # every char a token: 64 tokens per step
        if edit == 2:
            k = nrng.randint(0, len(lab), size=len(lab) // 5)
            lab[k] = nrng.randint(0, 3, size=len(k))
        got = pred.fill_tags_packed(utf8, boff, ooff, lab)
        for i, t in enumerate(texts):
            a, b = int(ooff[i]), int(ooff[i + 1])
            want, _ = orc.predict_tags(t, labels=lab[a:b])
            assert np.array_equal(got[a + i:a + i + len(t)], want), (edit, i, t[:30])


def test_tag_token_table_keys_and_queue():
    """The token table is keyed by the length and the low 16 bits of the first four chars: surfaces that share both
    (same 4-char prefix and length; a non-BMP char whose low 16 bits are another model's BMP char) must be told apart, a
    token that began more than a ring (64 chars) before its last char is read from the batch's chars, and tokens with
    models are queued across steps AND sentences (16 per pass): short sentences with one token each, sentences with
    more than 16 per step."""
    from vaporetto_amd.modelfmt import TagModel, TagNgramData, TagWeight
    import random
    rng = random.Random(11)
    alpha = randmodel.ALPHABETS["kana"][:5]
    m = randmodel.rand_model(951, alphabet=alpha, wc=3, wt=2, n_char=40, n_dict=20, max_word=3, n_tag_models=0)
    a = alpha
    long_tok = "".join(rng.choice(a) for _ in range(75))
    toks = [a[0] + a[1] + a[2] + a[3] + a[4], a[0] + a[1] + a[2] + a[3] + a[0], a[0] + a[1] + a[2] + a[3], a[0] + a[1] + a[2] + a[3] + a[4] + a[0],
            "\u0bb7", "𠮷", "\u0bb7" + a[0], a[0], a[1], a[2] + a[3], long_tok, long_tok[:74] + ("x" if long_tok[74] != "x" else "y")]

    def w(n):
        return [rng.randint(-3000, 3000) for _ in range(n)]
    for i, tok in enumerate(toks):
        slots = [["a%d" % i, "b", "c"], ["x", "y%d" % i]]
        tm = TagModel(tok, slots, bias=w(5))
        for _ in range(3):
            left = "".join(rng.choice(a) for _ in range(rng.randint(0, 2)))
            right = "".join(rng.choice(a) for _ in range(rng.randint(0, 2)))
            tm.char_ngram_model.append(TagNgramData((left + tok + right)[-12:] if len(left + tok + right) > 12 and not right else (tok[-3:] + right),
                                                    [TagWeight(len(right), w(5))]))
        tm.type_ngram_model.append(TagNgramData(bytes([3]), [TagWeight(0, w(5))]))
        m.tag_models.append(tm)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    orc = cbind.OraclePredictor(raw, True)
    nt = pred.n_tags()
    texts, labs = [], []

    def add(tokens):
        t = "".join(tokens)
        lab = []
        for tk in tokens:
            lab += [0] * (len(tk) - 1) + [1]
        texts.append(t); labs.append(lab[:-1])
    for _ in range(60):
        add([rng.choice(toks + ["𠮷" + a[0], "\u0bb7" + a[1], a[4] * 3]) for _ in range(rng.randint(1, 12))])
    for tk in toks:
        add([tk])                                         # one token per sentence: the queue fills across sentences
    for _ in range(10):
        add([rng.choice([a[0], a[1], a[2] + a[3]]) for _ in range(200)])   # > 16 tokens with models per step
    add([a[1]] * 30 + [long_tok] + [a[0]] + [toks[-1]] + [a[1]])
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    lab = np.array([x for l in labs for x in l], dtype=np.uint8)
    assert len(lab) == int(ooff[-1])
    got = pred.fill_tags_packed(utf8, boff, ooff, lab)
    for i, t in enumerate(texts):
        s, e = int(ooff[i]), int(ooff[i + 1])
        want, _ = orc.predict_tags(t, labels=lab[s:e])
        assert np.array_equal(got[s + i:s + i + len(t)], want), (i, t[:30])
    assert (got >= 0).any()


@pytest.mark.parametrize("chunk_bytes", ["700"])
def test_tokenize_batch_in_chunks(chunk_bytes, monkeypatch):
    """vpt_tokenize_batch runs chunk by chunk over several lanes with every slice of its device buffers placed by an upper
    bound and no number read back on the way; with VPT_TOKENIZE_CHUNK_BYTES tiny the same batches go through dozens of chunks
    (tagged and plain, filters on, a sentence longer than a chunk, errors still reported)."""
    monkeypatch.setenv("VPT_TOKENIZE_CHUNK_BYTES", chunk_bytes)
    test_tokenize_batch_is_the_whole_pipeline()
    m = randmodel.rand_model(852, alphabet="kana", wc=3, wt=3, n_char=60, n_dict=60)
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], False)
    texts = randmodel.rand_sentences(8, m, 300, alphabet="kana", max_len=40)
    texts[150] = "a\x00b"
    with pytest.raises(api.VaporettoError) as e:
        pred.tokenize(texts)
    assert "NULL" in str(e.value)
    assert pred.tokenize(texts[:150]) == pred.tokenize(texts[:100]) + pred.tokenize(texts[100:150])   # and the workspaces are clean again


def _fused_write(pred, texts, flags=0, want_scores=True, cap=None):
    """vpt_predict_write_batch_device over `texts`: (scores, labels, tokenized lines, plan)."""
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    S, nb = len(texts), int(ooff[-1])
    d = [devmem.put(np.concatenate([utf8, np.zeros(32, np.uint8)])), devmem.put(boff), devmem.put(ooff)]
    d_scores, d_labels = devmem.zeros(nb + 1, np.int32), devmem.zeros(nb + 1, np.uint8)
    cap = 3 * len(utf8) + 16 if cap is None else cap
    d_out, d_toff = devmem.zeros(cap + 16, np.uint8), devmem.zeros(S + 1, np.uint64)
    batch = api.DeviceBatch(pred)
    batch.set_flags(flags)
    batch.predict_write(d[0].ptr, d[1].ptr, d[2].ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr if want_scores else 0,
                        d_labels.ptr if want_scores else 0, d_out.ptr, cap, d_toff.ptr, devmem.stream())
    batch.sync()
    toff = d_toff.get()
    raw = bytes(d_out.get()[:int(toff[S])])
    lines = [raw[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(S)]
    return d_scores.get()[:nb], d_labels.get()[:nb], lines, batch.last_plan(), (utf8, boff, ooff)


@pytest.mark.parametrize("case", ["rows", "window-table", "no-types", "window-5", "window-8", "general"])
def test_predict_and_write_in_one_launch(case, monkeypatch):
    """(The name is history: rounds 4 - 5 fused the writer into the scoring kernel; since round 6 the call is the scoring launch and the
    writer's, back to back.)  vpt_predict_write_batch_device: the tokenized text is byte for byte what the oracle's
    Sentence::write_tokenized_text (sentence.rs:850-886) makes of the oracle's labels, for whole-sentence and cut tiles, every UTF-8 length,
    the escaped bytes, 1-char sentences, sentences longer than a tile, with and without the score / label outputs (without: the labels stay
    in the workspace)."""
    wc, wt = {"window-5": (5, 2), "window-8": (8, 8)}.get(case, (3, 3))
    if case == "window-table":
        monkeypatch.setenv("VPT_FORCE_WINDOW_TABLE", "1")
    m = randmodel.rand_model(4100 + wc, alphabet="mixed", wc=wc, wt=wt, n_char=120, n_dict=120, n_type=0 if case == "no-types" else 60, max_word=9)
    if case == "general":
        monkeypatch.setenv("VPT_FORCE_GENERIC", "1")                        # the general kernels, then the writer's own launch
    pred, orc = make_predictor(m)
    rng = np.random.default_rng(23)
    alphabet = randmodel.ALPHABETS["mixed"] + list("/\\  /") + ["\n"]
    lens = list(rng.integers(1, 60, 400)) + [1, 1, 1, 2, 63, 64, 65] + [1] * 40      # (at most 4 * 65 bytes: whole-sentence tiles)
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in lens]
    texts += randmodel.rand_sentences(5, m, 120, alphabet="mixed", max_len=50)
    # tiles of more than 4 KB of text (4-byte chars): more 16-byte chunks than the block has threads, the writer numbers the chars again
    wide = [c for c in alphabet if len(c.encode("utf-8")) == 4] + ["/", "あ"]
    texts += ["".join(rng.choice(wide, size=60)) for _ in range(60)]
    for long_ones in (False, True):
        batch_texts = texts + (["".join(rng.choice(alphabet, size=int(n))) for n in (300, 319, 320, 321, 1500, 4000, 2, 2600)]
                               + ["".join(rng.choice(wide, size=3000))] if long_ones else [])
        for want_scores in (True, False):
            scores, labels, lines, plan, (utf8, boff, ooff) = _fused_write(pred, batch_texts, want_scores=want_scores)
            o_scores, o_labels, _, _ = orc.predict_batch(utf8, boff)
            if want_scores:
                assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
            o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, o_labels, None, None)
            want = [bytes(o_text[int(o_toff[i]):int(o_toff[i + 1])]).decode("utf-8") for i in range(len(batch_texts))]
            assert lines == want, next((i, batch_texts[i], lines[i], want[i]) for i in range(len(want)) if lines[i] != want[i])
            if case == "general":
                assert plan["kind"] == "general kernels"
            else:
                assert plan["kind"] == ("cut tiles" if long_ones else "whole-sentence tiles")
    # the label filters act before the writer; small tiles cut sentences at many places
    flags = api._lib.VPT_FLAG_SPLIT_LINEBREAKS | (1 << 3)
    for tile_flat in ("", "61", "256"):
        if tile_flat:
            monkeypatch.setenv("VPT_TILE_FLAT", tile_flat)
            monkeypatch.setenv("VPT_FORCE_CUT_TILES", "1")
        _, labels, lines, _, (utf8, boff, ooff) = _fused_write(pred, texts, flags=flags)
        o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, labels, None, None)
        assert lines == [bytes(o_text[int(o_toff[i]):int(o_toff[i + 1])]).decode("utf-8") for i in range(len(texts))]
        assert np.array_equal(labels, pred.predict_packed(utf8, boff, wsconst=(3,), split_linebreaks=True)[1])
    monkeypatch.delenv("VPT_TILE_FLAT", raising=False)
    monkeypatch.delenv("VPT_FORCE_CUT_TILES", raising=False)
    # too small a buffer is an error, not a write outside it
    with pytest.raises(api.VaporettoError, match="text_capacity"):
        _fused_write(pred, texts, cap=100)


@pytest.mark.parametrize("chunk_bytes", ["", "900", "20000", "20000:small-runs", "300"])
def test_tokenize_batch_into_pinned_buffers(chunk_bytes, monkeypatch):
    """vpt_tokenize_batch without tags: chunk after chunk -- scoring launch, then the writer's -- into one contiguous text (the writers hand
    the position on through device words), copied out chunk by chunk while the next one is scored and the one after it has its chars counted
    and its tiles found on another stream; pinned (vpt_host_alloc) and pageable caller buffers.  Too small a buffer is an error."""
    if chunk_bytes.endswith(":small-runs"):   # many runs per chunk for the writer's look-back (run numbers that are multiples of 64 among them), cut tiles
        chunk_bytes = chunk_bytes.split(":")[0]
        monkeypatch.setenv("VPT_EMIT_PER_BLOCK", "1")
        monkeypatch.setenv("VPT_TILE_FLAT", "64")
        monkeypatch.setenv("VPT_FORCE_CUT_TILES", "1")
    if chunk_bytes:
        monkeypatch.setenv("VPT_TOKENIZE_CHUNK_BYTES", chunk_bytes)
    m = randmodel.rand_model(853, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=80, max_word=6)
    pred, orc = make_predictor(m)
    rng = np.random.default_rng(4)
    alphabet = randmodel.ALPHABETS["kana"][:12] + list("漢字 /\\aé🤌")
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in list(rng.integers(1, 70, 500)) + [1, 1, 900, 2, 3000, 1]]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    S = len(texts)
    o_scores, o_labels, ooff, _ = orc.predict_batch(utf8, boff)
    o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, o_labels, None, None)
    cap = 3 * len(utf8)
    pin_text, pin_off = api.PinnedArray((cap,), np.uint8), api.PinnedArray((S + 1,), np.uint64)
    pin_text.array[:] = 0xEE
    for _ in range(2):   # the workspace, its chain words and the ticket are reused
        pin_text.array[:] = 0xEE   # (poisoned every time: a run written to another run's place must not find the right bytes there from the call before)
        text, toff = pred.tokenize_packed(utf8, boff, text_out=pin_text.array, offsets_out=pin_off.array)
        assert np.array_equal(toff, o_toff) and np.array_equal(text, o_text)
        assert (pin_text.array[int(o_toff[-1]):] == 0xEE).all()          # nothing behind the text was touched
    text, toff = pred.tokenize_packed(utf8, boff)                          # pageable buffers
    assert np.array_equal(toff, o_toff) and np.array_equal(text, o_text)
    L = api._lib.load()
    for out in (pin_text.array, np.zeros(cap, np.uint8)):
        st = L.vpt_tokenize_batch(pred.handle, utf8.ctypes.data, boff.ctypes.data, S, 0, 0, out.ctypes.data, 64, pin_off.array.ctypes.data)
        assert st == api._lib.VPT_INVALID_ARGUMENT and "text_capacity" in api._lib.last_error()
    for _ in range(4):
        pin_text.array[:] = 0xEE
        text, toff = pred.tokenize_packed(utf8, boff, text_out=pin_text.array, offsets_out=pin_off.array)   # and the workspace is clean again
        assert np.array_equal(toff, o_toff) and np.array_equal(text, o_text)


def test_converted_kytea_fixture_on_gpu():
    """resources/kytea-model.bin converted by vaporetto_amd/kytea.py (kytea_model.rs:401-422): same tokens on the GPU."""
    from vaporetto_amd import kytea
    raw = kytea.convert(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kytea-model.bin"), "rb").read())
    pred, orc = make_predictor(raw)
    s = api.Sentence.from_raw("まぁ社長は火星猫だ")
    pred.predict(s)
    assert s.write_tokenized_text() == "まぁ 社長 は 火星 猫 だ"
    check_batch(pred, orc, ["まぁ社長は火星猫だ", "まぁ良いだろう", "火星猫"] * 20)


def test_fullwidth_filter_on_device():
    """VPT_FLAG_KYTEA_FULLWIDTH scores the text as KyteaFullwidthFilter rewrites it.  Pinned by the tantivy adapter's
    test (vaporetto_tantivy/src/lib.rs:298-364): "123456円" + emoji with the filter splits into single chars; and
    checked against the oracle on the host-filtered text for both kernels, boundary scores and tags."""
    raw, _ = kat.load_fixture("tantivy_model.bin")
    pred, orc = make_predictor(raw)
    utf8, boff = api.pack_texts(["123456円🤌🏿".encode("utf-8")])
    scores, labels, _ = pred.predict_packed(utf8, boff, fullwidth=True)
    assert scores.tolist() == [36480, 36480, 40155, 40155, 40155, 40155, 36442, 36442] and labels.tolist() == [1] * 8
    f = api.KyteaFullwidthFilter()
    for model_seed, alphabet in ((3, "mixed"), (4, "kana")):
        extra = list("abcXYZ019-.,!?()[]/_+:&*@=%<>{}\"'") + ["｢", "｣", "～", "－", "､", "―", "･", "─", "–", "｡"]
        alpha = (randmodel.ALPHABETS[alphabet] if alphabet == "mixed" else randmodel.ALPHABETS["kana"][:12]) + extra + [f.filter(c) for c in extra]
        m = randmodel.rand_model(600 + model_seed, alphabet=alpha, wc=3, wt=3, n_char=300, n_dict=300, n_type=60, max_word=6, n_tag_models=12)
        rawm = encode_model(m)
        texts = randmodel.rand_sentences(model_seed, m, 800, alphabet=alpha, max_len=50)
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
        n_utf8, n_boff = api.pack_texts([f.filter(t).encode("utf-8") for t in texts])
        for predict_tags in (False, True):
            p = api.Predictor(api.Model.read_slice(rawm)[0], predict_tags)
            o = cbind.OraclePredictor(rawm, predict_tags)
            scores, labels, ooff = p.predict_packed(utf8, boff, fullwidth=True)
            o_scores, o_labels, o_ooff, _ = o.predict_batch(n_utf8, n_boff)
            assert np.array_equal(ooff, o_ooff) and np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
            if predict_tags:
                got = p.fill_tags_packed(utf8, boff, ooff, labels, fullwidth=True)
                for i, t in enumerate(texts[:200]):
                    a, b = int(ooff[i]), int(ooff[i + 1])
                    want, _ = o.predict_tags(f.filter(t), labels=labels[a:b])
                    assert np.array_equal(got[a + i:a + i + len(t)], want), t


@pytest.mark.parametrize("alphabet_name", ["kana", "mixed"])
def test_label_post_filters_on_device(alphabet_name):
    """KyteaWsConstFilter (kytea_wsconst.rs:26-43) and SplitLinebreaksFilter (split_linebreaks.rs:9-36) as label masks:
    scores stay those of the oracle, labels equal the oracle's labels with the filters applied on the host; pinned
    by the tantivy adapter's expectation for wsconst "D" (vaporetto_tantivy/src/lib.rs:366-420: "123456円" keeps the
    digit run together)."""
    raw, _ = kat.load_fixture("tantivy_model.bin")
    pred, _ = make_predictor(raw)
    utf8, boff = api.pack_texts(["123456円🤌🏿".encode("utf-8")])
    _, labels, _ = pred.predict_packed(utf8, boff, fullwidth=True, wsconst=[api.CharacterType.Digit])
    assert labels.tolist() == [0, 0, 0, 0, 0, 1, 1, 1]      # １２３４５６ | 円 | 🤌 | 🏿

    extra = list("0123456789abcXYZ ") + [chr(10), chr(13)]
    base = randmodel.ALPHABETS["mixed"] if alphabet_name == "mixed" else randmodel.ALPHABETS["kana"][:12]
    alpha = base + extra
    m = randmodel.rand_model(640, alphabet=alpha, wc=3, wt=3, n_char=300, n_dict=300, n_type=60, max_word=6)
    pred, orc = make_predictor(m)
    texts = randmodel.rand_sentences(9, m, 1200, alphabet=alpha, max_len=60)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff)
    ws = [api.CharacterType.Digit, api.CharacterType.Roman, api.CharacterType.Kanji]
    scores, labels, ooff = pred.predict_packed(utf8, boff, wsconst=ws, split_linebreaks=True)
    assert np.array_equal(scores, o_scores)
    want = o_labels.copy()
    for i, t in enumerate(texts):
        a = int(o_ooff[i])
        types = api._types_of(np.frombuffer(t.encode("utf-32-le"), dtype=np.uint32))
        for k in range(len(t) - 1):
            if types[k] == types[k + 1] and int(types[k]) in [int(x) for x in ws]:
                want[a + k] = 0
        for k in range(len(t) - 1):
            if t[k] in (chr(10), chr(13)) or t[k + 1] in (chr(10), chr(13)):
                want[a + k] = 1
    assert np.array_equal(labels, want)


def test_device_resident_predict_then_fill_tags():
    """The whole config-5 pipeline without leaving HBM: vpt_predict_batch_device -> vpt_fill_tags_batch_device on the
    labels it wrote, one stream; equals the host-buffer path."""
    m = randmodel.rand_model(830, alphabet="kana", wc=3, wt=3, n_tag_models=30, max_word=4, n_char=60, n_dict=60)
    raw = encode_model(m)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    nt = pred.n_tags()
    texts = randmodel.rand_sentences(2, m, 3000, alphabet="kana", max_len=45) + [t.token * 2 for t in m.tag_models]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff)
    want = pred.fill_tags_packed(utf8, boff, ooff, labels)
    nb, S = int(ooff[-1]), len(texts)
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
    d_boff = devmem.put(boff.astype(np.uint64))
    d_ooff = devmem.put(ooff.astype(np.uint64))
    d_scores = devmem.zeros(nb + 1, np.int32)
    d_labels = devmem.zeros(nb + 1, np.uint8)
    d_tags = devmem.zeros((nb + S) * nt + 1, np.int32)
    batch = api.DeviceBatch(pred)
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr,
                  d_labels.ptr, devmem.stream())
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    assert np.array_equal(d_scores.get(nb), scores) and np.array_equal(d_labels.get(nb), labels)
    assert np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), want)
    # fill_tags after predict on the same workspace reuses the chars predict decoded; a fill_tags on its own (another
    # workspace: it decodes itself), through KyteaFullwidthFilter too, with mixed-width text, must give the same tags
    mixed = ["ab1" + t + "Ｚ９" for t in texts[:500]] + texts[500:1500]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in mixed])
    for fw in (False, True):
        _, labels, ooff = pred.predict_packed(utf8, boff, fullwidth=fw)
        want = pred.fill_tags_packed(utf8, boff, ooff, labels, fullwidth=fw)
        nb, S = int(ooff[-1]), len(mixed)
        d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
        d_boff, d_ooff = devmem.put(boff.astype(np.uint64)), devmem.put(ooff.astype(np.uint64))
        d_scores, d_labels = devmem.zeros(nb + 1, np.int32), devmem.zeros(nb + 1, np.uint8)
        got = []
        for reuse in (True, False):
            d_tags = devmem.zeros((nb + S) * nt + 1, np.int32)
            b1, b2 = api.DeviceBatch(pred), api.DeviceBatch(pred)
            b1.set_fullwidth(fw); b2.set_fullwidth(fw)
            b1.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr, d_labels.ptr, devmem.stream())
            (b1 if reuse else b2).fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
            b1.sync(); b2.sync()
            got.append(d_tags.get((nb + S) * nt).reshape(nb + S, nt))
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want), fw


def test_chars_left_by_predict_are_never_another_batchs():
    """the chars a predict call leaves decoded for the fill_tags call that follows it (predictor.rs:542) were
    remembered by buffer address and shape, and the host entry points reuse their staging buffers: predict(A) followed by
    fill_tags(B) with as many sentences and chars tagged B with A's chars.  The chars are now good for the NEXT fill_tags call
    only, and forgotten whenever a workspace is taken from the pool or staged into."""
    m = randmodel.rand_model(831, alphabet="kana", wc=3, wt=3, n_tag_models=40, max_word=3, n_char=60, n_dict=60)
    raw = encode_model(m)
    fresh = api.Predictor(api.Model.read_slice(raw)[0], True)
    pred = api.Predictor(api.Model.read_slice(raw)[0], True)
    toks = [t.token for t in m.tag_models]
    a_texts = [(toks[i % len(toks)] + toks[(i + 3) % len(toks)]) * 3 for i in range(200)]
    b_texts = [(toks[(i + 7) % len(toks)] + toks[(i + 11) % len(toks)]) * 3 for i in range(200)]
    b_texts = [b[:len(a)].ljust(len(a), "あ") for a, b in zip(a_texts, b_texts)]      # the same shape, char for char
    ua, ba = api.pack_texts([t.encode("utf-8") for t in a_texts])
    ub, bb = api.pack_texts([t.encode("utf-8") for t in b_texts])
    assert len(ua) == len(ub) and np.array_equal(ba, bb)
    _, lab_b, ooff = fresh.predict_packed(ub, bb)
    want = fresh.fill_tags_packed(ub, bb, ooff, lab_b)
    pred.predict_packed(ua, ba)                                   # leaves A's chars in the pooled workspace
    assert np.array_equal(pred.fill_tags_packed(ub, bb, ooff, lab_b), want)
    # the device entry points: the chars are taken once, by the call that follows; a later fill_tags on rewritten text decodes
    nb, S, nt = int(ooff[-1]), len(a_texts), pred.n_tags()
    d_text = devmem.put(np.concatenate([ua, np.zeros(16, np.uint8)]))
    d_boff, d_ooff = devmem.put(ba.astype(np.uint64)), devmem.put(ooff.astype(np.uint64))
    d_scores, d_labels, d_tags = devmem.zeros(nb + 1, np.int32), devmem.zeros(nb + 1, np.uint8), devmem.zeros((nb + S) * nt + 1, np.int32)
    batch = api.DeviceBatch(pred)
    mb = int(np.max(np.diff(ba.astype(np.int64))))
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, mb, d_scores.ptr, d_labels.ptr, devmem.stream())
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    d_text.set(np.concatenate([ub, np.zeros(16, np.uint8)]))
    d_labels.set(np.concatenate([lab_b, np.zeros(1, np.uint8)]))
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    assert np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), want)
    # predict(A), a sync, the SAME buffer rewritten with B, fill_tags(B) -- a sync ends the chars' validity
    d_text.set(np.concatenate([ua, np.zeros(16, np.uint8)]))
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, mb, d_scores.ptr, d_labels.ptr, devmem.stream())
    batch.sync()
    d_text.set(np.concatenate([ub, np.zeros(16, np.uint8)]))
    d_labels.set(np.concatenate([lab_b, np.zeros(1, np.uint8)]))
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    assert np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), want)


def test_fill_tags_with_offsets_that_do_not_match_the_text():
    """The tag entry points trust the caller's out_offsets as little as predict does: offsets that promise fewer (or
    more) chars than the text holds are an error, never a write outside the batch's arrays."""
    m = randmodel.rand_model(831, alphabet="kana", wc=3, wt=3, n_tag_models=10, max_word=4, n_char=40, n_dict=40)
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], True)
    texts = randmodel.rand_sentences(3, m, 40, alphabet="kana", max_len=30)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    _, labels, ooff = pred.predict_packed(utf8, boff)
    want = pred.fill_tags_packed(utf8, boff, ooff, labels)
    S, nb, nt = len(texts), int(ooff[-1]), pred.n_tags()
    short = ooff.copy(); short[5:] -= 3              # sentence 4 is promised 3 chars fewer than it has
    long_ = ooff.copy(); long_[7:] += 2              # sentence 6 two more
    back = ooff.copy(); back[9] = back[8] - 1        # decreasing
    for bad in (short, long_, back):
        nlab = max(int(bad[-1]), nb)
        lab = np.zeros(nlab + 1, np.uint8); lab[:nb] = labels
        with pytest.raises(api.VaporettoError, match="do not match the text"):
            pred.fill_tags_packed(utf8, boff, bad, lab)
    # device-resident entry point: reported at sync, once; the workspace stays usable
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)]))
    d_boff = devmem.put(boff.astype(np.uint64))
    d_labels = devmem.put(np.concatenate([labels, np.zeros(16, np.uint8)]))
    d_tags = devmem.zeros((nb + S) * nt + 1, np.int32)
    batch = api.DeviceBatch(pred)
    d_bad = devmem.put(short.astype(np.uint64))
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_bad.ptr, S, int(short[-1]), d_labels.ptr, d_tags.ptr, devmem.stream())
    with pytest.raises(api.VaporettoError, match="do not match the text"):
        batch.sync()
    d_ooff = devmem.put(ooff.astype(np.uint64))
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.sync()
    assert np.array_equal(d_tags.get((nb + S) * nt).reshape(nb + S, nt), want)


# ------------------------------------------------------------------------------------------------ token emission
def _tokenized_reference(text, labels):
    """sentence.rs:850-886 restated: runs between WordBoundary labels, ' ' between them, '\\' before ' ', '\\', '/'."""
    out, tok = [], []
    for i, c in enumerate(text):
        tok.append("\\" + c if c in " \\/" else c)
        if i == len(text) - 1 or labels[i] == 1:
            out.append("".join(tok))
            tok = []
    return " ".join(out)


def test_write_tokenized_text_on_device():
    """vpt_write_tokenized_batch = Sentence::write_tokenized_text (boundary part) for every sentence of a batch."""
    raw, _ = kat.load_fixture("model.bin")
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    for text, expected, _cite in [(t, e, c) for _f, t, e, c in kat.FIXTURE_SPLITS if _f == "model.bin"]:
        s = api.Sentence.from_raw(text)
        pred.predict(s)
        assert pred.write_tokenized_batch([s]) == [" ".join(expected)] == [s.write_tokenized_text()]
    rng = np.random.default_rng(11)
    alphabet = list("あいう漢字ab /\\\\/ .🤌é") + ["\n"]
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in list(rng.integers(1, 40, 300)) + [1, 1, 63, 64, 65, 127, 128, 129, 700]]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    labels = (rng.random(int(ooff[-1])) < 0.4).astype(np.uint8)
    text, toff = pred.write_tokenized_packed(utf8, boff, ooff, labels)
    got = bytes(text)
    for i, t in enumerate(texts):
        want = _tokenized_reference(t, labels[int(ooff[i]):int(ooff[i + 1])])
        assert got[int(toff[i]):int(toff[i + 1])].decode("utf-8") == want, (i, t)
    # the mirror's own writer agrees (it is the oracle of the reference's doc tests, tests/test_host_cabi.py)
    sents = []
    for i, t in enumerate(texts[:50]):
        s = api.Sentence.from_raw(t)
        s._boundaries = labels[int(ooff[i]):int(ooff[i + 1])].copy()
        sents.append(s)
    assert pred.write_tokenized_batch(sents) == [s.write_tokenized_text() for s in sents]
    # errors: an Unknown label, offsets that do not match the text, too small a buffer
    bad = labels.copy(); bad[3] = 2
    with pytest.raises(api.VaporettoError, match="can be written as tokenized text"):
        pred.write_tokenized_packed(utf8, boff, ooff, bad)
    short = ooff.copy(); short[5:] -= 1
    with pytest.raises(api.VaporettoError, match="do not match the text"):
        pred.write_tokenized_packed(utf8, boff, short, labels)
    L = api._lib.load()
    tiny = np.zeros(8, np.uint8); toff2 = np.zeros(len(texts) + 1, np.uint64)
    st = L.vpt_write_tokenized_batch(pred.handle, utf8.ctypes.data, boff.ctypes.data, len(texts), ooff.ctypes.data, labels.ctypes.data,
                                     tiny.ctypes.data, 8, toff2.ctypes.data)
    assert st == api._lib.VPT_INVALID_ARGUMENT and "text_capacity" in api._lib.last_error()
    # device-resident: predict -> write on one stream
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.put(ooff.astype(np.uint64))
    nb, S = int(ooff[-1]), len(texts)
    d_scores = devmem.zeros(nb + 1, np.int32); d_labels = devmem.zeros(nb + 1, np.uint8)
    cap = 2 * len(utf8) + nb + S
    d_out = devmem.zeros(cap + 1, np.uint8); d_toff = devmem.zeros(S + 1, np.uint64)
    batch = api.DeviceBatch(pred)
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr, d_labels.ptr, devmem.stream())
    batch.write_tokenized(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_out.ptr, cap, d_toff.ptr, devmem.stream())
    batch.sync()
    _, lab, _ = pred.predict_packed(utf8, boff)
    want_text, want_off = pred.write_tokenized_packed(utf8, boff, ooff, lab)
    assert np.array_equal(d_toff.get(S + 1), want_off) and np.array_equal(d_out.get(int(want_off[-1])), want_text)


def test_writer_every_byte_value_at_every_alignment():
    """The writer's masks (which bytes begin a char, which are escaped: sentence.rs:850-886 escapes ' ', '\\' and '/') come from byte flags gathered
    sixteen at a time (device_common.h, flag_bytes_to_mask16 / esc_flags, round 6): every byte value a text can hold -- every ASCII char but NUL, lead
    and continuation bytes of two-, three- and four-byte chars -- at every offset of a 16-byte chunk, with and without a space in front; waves whose
    chunks hold no escaped byte take another loop than those that do, and a run's ragged ends a third."""
    raw, _ = kat.load_fixture("model.bin")
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    rng = np.random.default_rng(23)
    chars = [chr(c) for c in range(1, 128)] + ["\u00e9", "\u07ff", "\u3042", "\u6f22", "\uffee", "\U0001f90c", "\U0010ffff"]
    texts = []
    for shift in range(16):
        for c in chars:
            texts.append("a" * shift + c + "\u3042" * 3)                      # the byte at offset `shift` of its chunk
    texts += ["\u6f22\u5b57" * 700, "x" * 5000, "/ \\" * 900]                # long runs without and with escaped bytes (whole waves of either kind)
    texts += ["".join(rng.choice(chars, size=int(n))) for n in rng.integers(1, 200, 200)]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    for p_one in (0.0, 0.35, 1.0):
        labels = (rng.random(int(ooff[-1])) < p_one).astype(np.uint8)
        text, toff = pred.write_tokenized_packed(utf8, boff, ooff, labels)
        got = bytes(text)
        for i, t in enumerate(texts):
            want = _tokenized_reference(t, labels[int(ooff[i]):int(ooff[i + 1])])
            assert got[int(toff[i]):int(toff[i + 1])].decode("utf-8") == want, (p_one, i, t[:40])


def test_write_tagged_text_on_device():
    """vpt_write_tagged_batch = fill_tags + write_tokenized_text with "/tag" suffixes (sentence.rs:850-886) on the device."""
    # the reference's own tagged outputs (resources/docs.tok lines, kat.FIXTURE_TAGGED)
    for fixture, text, expected in kat.FIXTURE_TAGGED:
        raw, _ = kat.load_fixture(fixture)
        pred = api.Predictor(api.Model.read_slice(raw)[0], True)
        s = api.Sentence.from_raw(text)
        pred.predict(s)
        assert pred.write_tokenized_batch([s], tagged=True) == [expected]
    # random tag models (tags with ' ', '/', '\\' in them, None slots in the middle) against the host writer
    m = randmodel.rand_model(840, alphabet="kana", wc=3, wt=3, n_tag_models=40, max_word=4, n_char=60, n_dict=60)
    for k, tm in enumerate(m.tag_models):
        tm.tags = [[t + (" /\\"[k % 3]) * (k % 2) for t in cands] for cands in tm.tags]
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], True)
    texts = randmodel.rand_sentences(4, m, 400, alphabet="kana", max_len=40) + [t.token * 3 for t in m.tag_models] + ["あ", "い/う え\\"]
    sents = [api.Sentence.from_raw(t) for t in texts]
    pred.predict_batch(sents)
    got = pred.write_tokenized_batch(sents, tagged=True)
    pred.fill_tags_batch(sents)
    want = [s.write_tokenized_text() for s in sents]
    assert got == want
    assert any("/" in w.replace("\\/", "") for w in want)   # some token did get a tag
    # a predictor without predict_tags refuses
    plain = api.Predictor(api.Model.read_slice(encode_model(m))[0], False)
    with pytest.raises(api.VaporettoError, match="predict_tags = false"):
        plain.write_tokenized_batch(sents[:3], tagged=True)
    # device-resident: predict -> fill_tags -> write_tagged on one stream
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    nb, S, nt = int(ooff[-1]), len(texts), pred.n_tags()
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.put(ooff.astype(np.uint64))
    d_scores = devmem.zeros(nb + 1, np.int32); d_labels = devmem.zeros(nb + 1, np.uint8); d_tags = devmem.zeros((nb + S) * nt + 1, np.int32)
    cap = 2 * len(utf8) + (nb + S) * 64
    d_out = devmem.zeros(cap + 1, np.uint8); d_toff = devmem.zeros(S + 1, np.uint64)
    batch = api.DeviceBatch(pred)
    with pytest.raises(api.VaporettoError, match="vpt_fill_tags_batch_device on this workspace"):
        batch.write_tagged(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, d_out.ptr, cap, d_toff.ptr, devmem.stream())
    batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_scores.ptr, d_labels.ptr, devmem.stream())
    batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, devmem.stream())
    batch.write_tagged(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d_tags.ptr, d_out.ptr, cap, d_toff.ptr, devmem.stream())
    batch.sync()
    toff = d_toff.get(S + 1)
    out = bytes(d_out.get(int(toff[-1])))
    assert [out[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(S)] == want
    # the dense array is not what the writer reads (round 6: the workspace's records are): NULL, or anything else, changes nothing
    d_none = devmem.put(np.full((nb + S) * nt + 1, -1, np.int32))
    for d in (0, d_none.ptr):
        d_out.set(np.zeros(cap + 1, np.uint8))
        batch.write_tagged(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, d, d_out.ptr, cap, d_toff.ptr, devmem.stream())
        batch.sync()
        toff = d_toff.get(S + 1)
        out = bytes(d_out.get(int(toff[-1])))
        assert [out[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(S)] == want
    # labels changed after fill_tags (a token the records have tags for ends nowhere any more): reported, nothing written out of place
    labels = d_labels.get(nb + 1).copy()
    labels[:nb] = 0
    d_labels.set(labels)
    batch.write_tagged(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, nb, d_labels.ptr, 0, d_out.ptr, cap, d_toff.ptr, devmem.stream())
    with pytest.raises(api.VaporettoError, match="do not match the text"):
        batch.sync()


@pytest.mark.parametrize("per_block", [1, 3, 64, 256, 300, 512])
def test_writer_blocks_of_any_size(per_block, monkeypatch):
    """The writer takes runs of consecutive sentences, a workgroup each (kernels_emit.hip, emit_flat_kernel, with and without tags); the run
    size comes from the mean sentence length (with tags: a whole multiple of fill_tags' runs, so that a workgroup's tag records are one slice
    named by two prefix words) -- here it is forced (VPT_EMIT_PER_BLOCK, read when a workspace is made) to sizes that are NO such multiple:
    one sentence per run, a few, 64, 256, more than a thread each (300, 512: two sentences a thread); sentences of 1 .. 13 000 chars (several 4 KB pieces of a workgroup's walk) with escapes, 1- to
    4-byte chars, every alignment of text, labels and output."""
    monkeypatch.setenv("VPT_EMIT_PER_BLOCK", str(per_block))
    m = randmodel.rand_model(843, alphabet="kana", wc=3, wt=3, n_tag_models=30, max_word=3, n_char=60, n_dict=60)
    for k, tm in enumerate(m.tag_models):
        tm.tags = [[t + (" /\\"[k % 3]) * (k % 2) for t in cands] for cands in tm.tags]
    raw = encode_model(m)
    rng = np.random.default_rng(13 + per_block)
    alphabet = list("あいう漢字ab /\\/ .🤌é") + ["\n"]
    lens = list(rng.integers(1, 40, 500)) + [1, 1, 63, 64, 65, 127, 128, 129, 341, 342, 343, 700, 900, 1365, 1366, 4095, 4096, 4097, 13000] + [1] * 300
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in lens]
    texts += ["a b/c\\" * 700, "ab" * 2048, "\\" * 4096 + "あ", "🤌" * 1100]   # pieces of one-byte chars only, every byte escaped, four-byte chars across pieces
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    for trial in range(3):   # the same workspace again: its state words alternate between two arrays
        pred = api.Predictor(api.Model.read_slice(raw)[0], False) if trial == 0 else pred
        labels = (rng.random(int(ooff[-1])) < (0.1, 0.5, 1.0)[trial]).astype(np.uint8)
        text, toff = pred.write_tokenized_packed(utf8, boff, ooff, labels)
        got = bytes(text)
        for i, t in enumerate(texts):
            assert got[int(toff[i]):int(toff[i + 1])].decode("utf-8") == _tokenized_reference(t, labels[int(ooff[i]):int(ooff[i + 1])]), (trial, i, t)
    # with tags: against the mirror's writer
    tagged = api.Predictor(api.Model.read_slice(raw)[0], True)
    ttexts = randmodel.rand_sentences(6, m, 300, alphabet="kana", max_len=40) + [t.token * 3 for t in m.tag_models] + ["あ", "い/う え\\"]
    sents = [api.Sentence.from_raw(t) for t in ttexts]
    tagged.predict_batch(sents)
    got = tagged.write_tokenized_batch(sents, tagged=True)
    tagged.fill_tags_batch(sents)
    assert got == [s.write_tokenized_text() for s in sents]
    # one-byte chars whose one- and two-char tokens nearly all have tags: many suffixes in front of a lane's sixteen bytes
    m2 = randmodel.rand_model(844, alphabet=list("abc"), wc=3, wt=3, n_tag_models=12, max_word=2, n_char=20, n_dict=10)
    dense = api.Predictor(api.Model.read_slice(encode_model(m2))[0], True)
    sents = [api.Sentence.from_raw("".join(rng.choice(list("abc"), size=int(n)))) for n in list(rng.integers(1, 90, 120)) + [400]]
    dense.predict_batch(sents)
    for s in sents:
        s._boundaries = (rng.random(len(s._boundaries)) < 0.7).astype(np.uint8)
    got = dense.write_tokenized_batch(sents, tagged=True)
    dense.fill_tags_batch(sents)
    assert got == [s.write_tokenized_text() for s in sents]
    assert sum(g.count("/") for g in got) > 1000


WRITER_TEST_SENTENCES = 9000   # (tests/test_kernel_emu.py runs the same test on fewer)


def test_writer_long_tags_many_sentences_and_long_sentences():
    """The writer assembles a step's output (1 KB of text and what is inserted) in LDS; tag strings of hundreds of bytes do not
    fit there and go out byte by byte; a block's position comes from the look-back over the earlier blocks' sizes, 64 per trip:
    thousands of blocks, sentences of thousands of bytes (a block of its own, many steps), every alignment."""
    m = randmodel.rand_model(841, alphabet="kana", wc=3, wt=3, n_tag_models=30, max_word=3, n_char=60, n_dict=60)
    for k, tm in enumerate(m.tag_models):
        tm.tags = [[(t + "/x " * 3) * (40 + 25 * (k % 5)) if k % 2 else t for t in cands] for cands in tm.tags]   # up to ~1.7 KB per tag
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], True)
    texts = randmodel.rand_sentences(5, m, 300, alphabet="kana", max_len=30) + [t.token * 5 for t in m.tag_models]
    sents = [api.Sentence.from_raw(t) for t in texts]
    pred.predict_batch(sents)
    got = pred.write_tokenized_batch(sents, tagged=True)
    pred.fill_tags_batch(sents)
    assert got == [s.write_tokenized_text() for s in sents]
    assert max(len(g) for g in got) > 2000
    # 9 000 short sentences (three scan workgroups) and a few of 2 000 .. 6 000 chars, ASCII with escapes among them
    import random
    rng = random.Random(5)
    alpha = randmodel.ALPHABETS["kana"][:20] + list("ab /\\9")
    texts = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(WRITER_TEST_SENTENCES)]
    for at, n in ((17, 2000), (4095, 6000), (4096, 3001), (WRITER_TEST_SENTENCES - 1, 2500)):
        texts[at] = "".join(rng.choice(alpha) for _ in range(n))
    plain = api.Predictor(api.Model.read_slice(encode_model(m))[0], False)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = plain.predict_packed(utf8, boff)
    ttext, toff = plain.write_tokenized_packed(utf8, boff, ooff, labels)
    tb = bytes(ttext)
    for i, t in enumerate(texts):
        lab = labels[int(ooff[i]):int(ooff[i + 1])]
        out, tok = [], []
        for k, c in enumerate(t):
            tok.append("\\" + c if c in " \\/" else c)
            if k == len(t) - 1 or lab[k] == 1:
                out.append("".join(tok)); tok = []
        assert tb[int(toff[i]):int(toff[i + 1])].decode("utf-8") == " ".join(out), i
    assert np.array_equal(api.count_boundaries(utf8, boff), ooff)
    # the whole pipeline on the device (vpt_count_boundaries_device's prefix sum over the same 9 000 sentences)
    lines = plain.tokenize(texts)
    assert lines == [tb[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(len(texts))]


def test_device_calls_accept_an_upper_bound_of_the_boundaries():
    """A caller that counted the chars on the device and does not wait for the total passes an upper bound of it (text bytes - S)
    to predict / fill_tags / write_tagged: same scores, tags and text (tiles that start past the real total are empty)."""
    m = randmodel.rand_model(843, alphabet="kana", wc=3, wt=3, n_tag_models=30, max_word=4, n_char=80, n_dict=80)
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], True)
    texts = randmodel.rand_sentences(7, m, 300, alphabet="kana", max_len=60) + ["abc de", "x" * 2000, "あ"]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    nb, S, nt = int(ooff[-1]), len(texts), pred.n_tags()
    max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
    outs = []
    for bound in (nb, len(utf8) - S):
        assert bound >= nb
        d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.put(ooff.astype(np.uint64))
        d_scores = devmem.zeros(bound + 1, np.int32); d_labels = devmem.zeros(bound + 1, np.uint8); d_tags = devmem.zeros((bound + S) * nt + 1, np.int32)
        cap = 3 * len(utf8) + len(utf8) * pred.max_tag_suffix()
        d_out = devmem.zeros(cap + 1, np.uint8); d_toff = devmem.zeros(S + 1, np.uint64)
        batch = api.DeviceBatch(pred)
        batch.predict(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, bound, max_bytes, d_scores.ptr, d_labels.ptr, devmem.stream())
        batch.fill_tags(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, bound, d_labels.ptr, d_tags.ptr, devmem.stream())
        batch.write_tagged(d_text.ptr, d_boff.ptr, d_ooff.ptr, S, bound, d_labels.ptr, d_tags.ptr, d_out.ptr, cap, d_toff.ptr, devmem.stream())
        batch.sync()
        toff = d_toff.get(S + 1)
        outs.append((d_scores.get(nb).copy(), d_labels.get(nb).copy(), d_tags.get((nb + S) * nt).copy(), bytes(d_out.get(int(toff[-1]))), toff.copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
    o_scores, o_labels, _, _ = cbind.OraclePredictor(encode_model(m), True).predict_batch(utf8, boff)
    assert np.array_equal(outs[1][0], o_scores) and np.array_equal(outs[1][1], o_labels)


def test_tokenize_batch_is_the_whole_pipeline():
    """vpt_tokenize_batch (lines in, tokenized lines out) = from_raw + predict (+ filters) (+ fill_tags) + write_tokenized_text,
    and vpt_count_boundaries_device = vpt_count_boundaries."""
    m = randmodel.rand_model(850, alphabet="mixed", wc=3, wt=3, n_tag_models=25, max_word=5, n_char=120, n_dict=120)
    raw = encode_model(m)
    texts = randmodel.rand_sentences(6, m, 500, alphabet="mixed", max_len=50) + ["a", "あ", "12 ab/c\\d", "x" * 300, "\n改行\r\n", "🤌🏿"]
    for tagged in (False, True):
        pred = api.Predictor(api.Model.read_slice(raw)[0], tagged)
        for kw in ({}, {"fullwidth": True}, {"wsconst": [1, 2], "split_linebreaks": True}):
            sents = [api.Sentence.from_raw(t) for t in texts]
            pred.predict_batch(sents, fullwidth=kw.get("fullwidth", False))
            if "wsconst" in kw:   # the label flags of predict_packed, applied through the packed entry point
                utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
                _, labels, ooff = pred.predict_packed(utf8, boff, wsconst=kw["wsconst"], split_linebreaks=True)
                for i, s in enumerate(sents):
                    s._boundaries = labels[int(ooff[i]):int(ooff[i + 1])].copy()
            if tagged:
                want = pred.write_tokenized_batch(sents, tagged=True) if not kw.get("fullwidth") else None
            else:
                want = pred.write_tokenized_batch(sents)
            got = pred.tokenize(texts, tagged=tagged, **kw)
            if want is not None:
                assert got == want, kw
            else:   # tags on the normalised text: the same tokens, checked against the packed tagged writer with the flag
                utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
                ooff = api.count_boundaries(utf8, boff)
                labels = np.concatenate([np.asarray(s._boundaries, np.uint8) for s in sents])
                t2, o2 = pred.write_tokenized_packed(utf8, boff, ooff, labels, tagged=True, fullwidth=True)
                assert got == [bytes(t2[int(o2[i]):int(o2[i + 1])]).decode("utf-8") for i in range(len(texts))]
    # errors of Sentence::from_raw, found on the device
    pred = api.Predictor(api.Model.read_slice(raw)[0], False)
    with pytest.raises(api.VaporettoError, match="must not contain NULL"):
        pred.tokenize(["ab", "c\0d"])
    # device-resident counting
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    d_text = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64))
    d_ooff = devmem.zeros(len(texts) + 1, np.uint64)
    batch = api.DeviceBatch(pred)
    st = api._lib.load().vpt_count_boundaries_device(pred.handle, batch._h, d_text.ptr, d_boff.ptr, len(texts), d_ooff.ptr, devmem.stream())
    assert st == api._lib.VPT_OK
    batch.sync()
    assert np.array_equal(d_ooff.get(len(texts) + 1), api.count_boundaries(utf8, boff))


def test_count_boundaries_on_the_device_flat_kernel():
    """vpt_count_boundaries_device (round 4: flat over the text, a workgroup per run of sentences, 16 KB pieces): sentences of 1 char and of
    tens of thousands (several pieces, piece edges inside chars), every UTF-8 length, unaligned text; NUL and empty sentences are reported."""
    m = randmodel.rand_model(860, alphabet="kana", wc=3, wt=3, n_char=20, n_dict=20)
    pred = api.Predictor(api.Model.read_slice(encode_model(m))[0], False)
    rng = np.random.default_rng(9)
    alphabet = list("あ漢aé🤌 ｱ/") + ["\n"]
    lens = list(rng.integers(1, 90, 1500)) + [1, 1, 5461, 5462, 16384, 16385, 40000, 1, 2, 70001, 3] + list(rng.integers(1, 5, 700))
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in lens]
    L = api._lib.load()
    for lead_in in (0, 1, 7):   # the text need not be aligned
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
        buf = devmem.put(np.concatenate([np.zeros(lead_in, np.uint8), utf8, np.zeros(16, np.uint8)]))
        d_boff = devmem.put((boff + np.uint64(lead_in)).astype(np.uint64)); d_ooff = devmem.zeros(len(texts) + 1, np.uint64)
        batch = api.DeviceBatch(pred)
        assert L.vpt_count_boundaries_device(pred.handle, batch._h, buf.ptr, d_boff.ptr, len(texts), d_ooff.ptr, devmem.stream()) == api._lib.VPT_OK
        batch.sync()
        assert np.array_equal(d_ooff.get(len(texts) + 1), api.count_boundaries(utf8, boff))
    for bad, msg in ((["ab", "c\0d", "e"], "must not contain NULL"), (["ab", "", "e"], "at least one character")):
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in bad])
        buf = devmem.put(np.concatenate([utf8, np.zeros(16, np.uint8)])); d_boff = devmem.put(boff.astype(np.uint64)); d_ooff = devmem.zeros(4, np.uint64)
        batch = api.DeviceBatch(pred)
        assert L.vpt_count_boundaries_device(pred.handle, batch._h, buf.ptr, d_boff.ptr, 3, d_ooff.ptr, devmem.stream()) == api._lib.VPT_OK
        with pytest.raises(api.VaporettoError, match=msg):
            batch.sync()


# ------------------------------------------------------------------------------------------------ compiled form, clones
def test_compiled_predictor_round_trip_and_clone():
    """vpt_predictor_save -> vpt_predictor_load and vpt_predictor_clone_to_device give predictors that score and tag
    exactly like the one compiled from the model (packed path, with tag models; and a model on the general path)."""
    m = randmodel.rand_model(611, alphabet="kana", wc=3, wt=3, n_char=300, n_dict=300, max_word=9, n_tag_models=6)
    raw = encode_model(m)
    model = api.Model.read_slice(raw)[0]
    pred = api.Predictor(model, True)
    orc = cbind.OraclePredictor(raw, True)
    assert pred.info()["packed"] == 1 and pred.info()["predict_tags"] == 1
    blob = pred.save_compiled()
    loaded = api.Predictor.load_compiled(blob, model=model)
    clone = pred.clone_to_device(0)
    assert loaded.info() == pred.info() == clone.info()
    mixed = randmodel.ALPHABETS["kana"][:12] + list("漢字A9、")
    texts = randmodel.rand_sentences(4, m, 1200, alphabet=mixed, max_len=70)
    base = check_batch(pred, orc, texts)
    for other in (loaded, clone):
        got = check_batch(other, orc, texts)
        assert all(np.array_equal(a, b) for a, b in zip(base, got))
        utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
        t_a = pred.fill_tags_packed(utf8, boff, base[2], base[1])
        t_b = other.fill_tags_packed(utf8, boff, base[2], base[1])
        assert np.array_equal(t_a, t_b)
    del pred   # the copies own their tables
    check_batch(loaded, orc, texts[:100])
    check_batch(clone, orc, texts[:100])
    # the general path (an alphabet of more chars than the packed tables have ids for): no packed tables in the compiled form
    m2 = randmodel.rand_model(612, alphabet="mixed", wc=4, wt=4, n_char=100, n_dict=100, max_word=6)
    for cp in range(0x20000, 0x20000 + 65600):
        m2.char_ngram_model.append(NgramData(chr(cp), [cp & 7, 0, -(cp & 3)]))
    raw2 = encode_model(m2)
    p2 = api.Predictor(api.Model.read_slice(raw2)[0], False)
    l2 = api.Predictor.load_compiled(p2.save_compiled())
    assert l2.info()["packed"] == 0
    check_batch(l2, cbind.OraclePredictor(raw2), randmodel.rand_sentences(5, m2, 400, alphabet="mixed", max_len=50))


def test_compiled_predictor_rejects_damaged_blobs():
    m = randmodel.rand_model(613, alphabet="kana", wc=3, wt=3, n_char=50, n_dict=50, max_word=5)
    blob = bytearray(api.Predictor(api.Model.read_slice(encode_model(m))[0], False).save_compiled())

    def refuse(b, what):
        with pytest.raises(api.VaporettoError) as e:
            api.Predictor.load_compiled(bytes(b))
        assert e.value.kind == "InvalidModel" and what in str(e.value), str(e.value)

    refuse(blob[:100], "too short")
    refuse(b"VaporettoTokenizer 0.5.0\n" + bytes(blob[25:]), "not a compiled predictor")
    refuse(blob[:-256], "truncated")
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x40
    refuse(bad, "checksum")
    bad = bytearray(blob); bad[16] ^= 0x01            # the version word
    refuse(bad, "version mismatch")
    # the description is covered too (a flipped bit in bias / geometry / section sizes passed the arena's checksum)
    import struct
    meta_bytes = struct.unpack_from("<I", blob, 20)[0]
    for at in (48, meta_bytes - 200, meta_bytes - 100):
        bad = bytearray(blob); bad[at] ^= 0x04
        refuse(bad, "checksum")
    assert api.Predictor.load_compiled(bytes(blob)).info()["n_char_ngrams"] == len(m.char_ngram_model)


# ------------------------------------------------------------------------------------------------ host-buffer pipeline, shards
@pytest.mark.parametrize("chunk_chars,pinned,lanes", [("700", False, "0"), ("2500", True, "0"), ("700", True, "4"), ("2500", False, "3"), ("1500", True, "8")])
def test_pipelined_host_path_matches_oracle(chunk_chars, pinned, lanes, monkeypatch):
    """vpt_predict_batch cuts a large batch into chunks and overlaps copy in / kernels / copy out -- on three streams with events
    (VPT_PIPE_LANES=0: what batches over 16 M chars take) or chunk by chunk over several independent lanes (what smaller ones
    take); with VPT_CHUNK_CHARS tiny, a 1 000-sentence batch goes through dozens of chunks, ragged sizes, every buffer set, pinned
    and pageable caller buffers -- scores identical to the oracle's, device-side errors still reported."""
    monkeypatch.setenv("VPT_CHUNK_CHARS", chunk_chars)
    monkeypatch.setenv("VPT_PIPE_LANES", lanes)
    m = randmodel.rand_model(777, alphabet="kana", wc=3, wt=3, n_char=200, n_dict=200, max_word=8)
    raw = encode_model(m)
    pred, orc = make_predictor(raw)
    mixed = randmodel.ALPHABETS["kana"][:12] + list("漢字A9、")
    texts = randmodel.rand_sentences(3, m, 1000, alphabet=mixed, max_len=90)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff, nthreads=4)
    if pinned:
        keep = [api.PinnedArray(utf8.shape, np.uint8), api.PinnedArray(o_scores.shape, np.int32), api.PinnedArray(o_labels.shape, np.uint8)]
        keep[0].array[:] = utf8
        keep[1].array[:] = 0
        keep[2].array[:] = 9
        scores, labels, ooff = api.predict_packed_sharded([pred], keep[0].array, boff, scores=keep[1].array, labels=keep[2].array)
    else:
        scores, labels, ooff = pred.predict_packed(utf8, boff)
    assert np.array_equal(ooff, o_ooff) and np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)
    # a NUL char in the middle of the batch: reported by the chunk that holds it
    texts2 = list(texts)
    texts2[500] = "a\x00b"
    u2, b2 = api.pack_texts([t.encode("utf-8") for t in texts2])
    with pytest.raises(api.VaporettoError) as e:
        api.predict_packed_sharded([pred], u2, b2, out_offsets=np.concatenate([[0], np.cumsum([len(t) - 1 for t in texts2])]).astype(np.uint64))
    assert "NULL" in str(e.value)


def test_tokenize_with_the_grapheme_cluster_filter():
    """`--wsconst G` of the CLI (predict/src/main.rs:101-104): Predictor.tokenize(.., wsconst=("G", ..)) = predict, ConcatGraphemeClustersFilter on
    the host, [fill_tags,] writer -- line for line what the per-sentence mirror gives, with and without the char-type filter, tags and
    the fullwidth normalisation beside it; and a ZWJ sequence the model wants to cut stays whole."""
    m = randmodel.rand_model(9300, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=60, max_word=5, n_tag_models=8)
    m.bias = 5000                                  # a model that cuts everywhere it can
    raw = encode_model(m)
    rng = np.random.default_rng(11)
    alphabet = randmodel.ALPHABETS["kana"][:10] + list("ab12 ") + ["\u0301", "\u200d", "\U0001f468", "\U0001f469", "\U0001f3fd", "\U0001f44f", "\u3099"]
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in rng.integers(1, 50, 200)] + ["\U0001f468\u200d\U0001f469\u200d\U0001f466", "a", "\u200d"]
    f = api.ConcatGraphemeClustersFilter()
    for tagged in (False, True):
        pred = api.Predictor(api.Model.read_slice(raw)[0], tagged)
        for fullwidth, types in ((False, ()), (True, (int(api.CharacterType.Roman),)), (False, (int(api.CharacterType.Digit), int(api.CharacterType.Kanji)))):
            got = pred.tokenize(texts, tagged=tagged, fullwidth=fullwidth, wsconst=("G",) + types)
            plain = pred.tokenize(texts, tagged=tagged, fullwidth=fullwidth, wsconst=types)
            norm = api.KyteaFullwidthFilter()
            sents = [api.Sentence.from_raw(norm.filter(t) if fullwidth else t) for t in texts]
            pred.predict_batch(sents)
            for s in sents:
                for ty in types:      # KyteaWsConstFilter (vaporetto_rules/src/sentence_filters/kytea_wsconst.rs): no boundary between two chars of the type
                    ct = s.char_types()
                    b = s.boundaries_mut()
                    b[(ct[:-1] == ty) & (ct[1:] == ty)] = api.CharacterBoundary.NotWordBoundary
                f.filter(s)
            if tagged:
                pred.fill_tags_batch(sents)
            want = []
            for t, s in zip(texts, sents):      # the CLI writes the ORIGINAL line with the normalised sentence's boundaries and tags (main.rs:158-163)
                o = api.Sentence.from_raw(t)
                o.boundaries_mut()[:] = s.boundaries()
                o._tags, o._n_tags = s._tags, s._n_tags
                want.append(o.write_tokenized_text())
            assert got == want
            assert any(a != b for a, b in zip(got, plain))      # the filter did something
        assert " " not in pred.tokenize(["\U0001f468\u200d\U0001f469\u200d\U0001f466"], wsconst=("G",))[0]


def test_labels_only_and_packed_tokenize_through_the_host_path(monkeypatch):
    """scores_out = NULL (a tokenizer only needs the labels; 4 of the 5 bytes per boundary stay on the device) gives the same
    labels one-shot, through the lanes and through the event pipeline; tokenize_packed returns what tokenize returns."""
    m = randmodel.rand_model(779, alphabet="kana", wc=3, wt=3, n_char=150, n_dict=150, max_word=6)
    raw = encode_model(m)
    pred, orc = make_predictor(raw)
    texts = randmodel.rand_sentences(12, m, 700, alphabet="kana", max_len=70)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff, nthreads=4)
    for chunk, lanes in ((None, None), ("900", "4"), ("900", "0")):
        if chunk:
            monkeypatch.setenv("VPT_CHUNK_CHARS", chunk)
            monkeypatch.setenv("VPT_PIPE_LANES", lanes)
            pred, _ = make_predictor(raw)      # the library reads its knobs when a predictor is made, never on the launch path
        scores, labels, ooff = api.predict_packed_sharded([pred], utf8, boff, want_scores=False)
        assert scores is None and np.array_equal(labels, o_labels) and np.array_equal(ooff, o_ooff)
    text, toff = pred.tokenize_packed(utf8, boff)
    lines = pred.tokenize(texts)
    tb = bytes(text)
    assert [tb[int(toff[i]):int(toff[i + 1])].decode("utf-8") for i in range(len(texts))] == lines
    assert "".join(lines[5].split(" ")) == texts[5] or any(c in texts[5] for c in " /\\")


def test_sharded_predict_over_clones_equals_unsharded(monkeypatch):
    """take shards -> score each on its own predictor (clones of one, as the ranks of a multi-GPU job hold) -> the
    concatenation is the unsharded result; shard bounds balance CHARACTERS, not sentences or bytes."""
    m = randmodel.rand_model(778, alphabet="kana", wc=3, wt=3, n_char=200, n_dict=200, max_word=8)
    raw = encode_model(m)
    pred, orc = make_predictor(raw)
    import random
    rng = random.Random(11)
    # mixed ASCII / JA text: bytes per char vary 1..3, lengths vary 1..200
    texts = []
    for i in range(600):
        alpha = list("abcdefgh 0123") if i % 3 == 0 else randmodel.ALPHABETS["kana"][:20] + list("漢字")
        texts.append("".join(rng.choice(alpha) for _ in range(rng.randint(1, 200 if i % 7 == 0 else 30))))
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff, nthreads=4)
    for n in (1, 2, 3, 5):
        bounds = api.shard_bounds(o_ooff, n)
        assert bounds[0] == 0 and bounds[n] == len(texts) and all(bounds[r] <= bounds[r + 1] for r in range(n))
        chars = [sum(len(t) for t in texts[int(bounds[r]):int(bounds[r + 1])]) for r in range(n)]
        assert max(chars) - min(chars) <= 2 * 200            # every cut is within one (longest) sentence of its target
        clones = [pred] + [pred.clone_to_device(0) for _ in range(n - 1)]
        scores, labels, ooff = api.predict_packed_sharded(clones, utf8, boff)
        assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels) and np.array_equal(ooff, o_ooff)
    monkeypatch.setenv("VPT_CHUNK_CHARS", "300")            # shards that are themselves pipelined (knobs are read when a predictor is made)
    clones = [pred.clone_to_device(0), pred.clone_to_device(0)]
    scores, labels, _ = api.predict_packed_sharded(clones, utf8, boff)
    assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels)


def test_char_types_from_the_device():
    """vpt_char_types_batch = Sentence::char_types (sentence.rs:1016) for a batch, plain and through KyteaFullwidthFilter."""
    pred, _ = make_predictor(kat.predictor_test_model())
    texts = ["Ab1あア漢、𠮷", "x", "ｱＡ１", "0９a"] + randmodel.rand_sentences(2, kat.predictor_test_model(), 200, alphabet="mixed", max_len=40)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    got = pred.char_types_packed(utf8, boff, ooff)
    want = np.concatenate([api.Sentence.from_raw(t).char_types() for t in texts])
    assert np.array_equal(got, want)
    got_fw = pred.char_types_packed(utf8, boff, ooff, fullwidth=True)
    fw = api.KyteaFullwidthFilter()
    want_fw = np.concatenate([api.Sentence.from_raw(fw.filter(t)).char_types() for t in texts])
    assert np.array_equal(got_fw, want_fw)
    # every scalar value of the BMP (and a few planes above): the decode kernel computes a char's type itself and asks the char table only where
    # KyteaFullwidthFilter can rewrite it (round 6) -- both against CharacterType::get_type (sentence.rs:50-67) of the plain / the filtered char
    cps = [c for c in range(1, 0x10000) if not 0xD800 <= c <= 0xDFFF] + [0x10000, 0x1F600, 0x20000, 0x2A6DF, 0x2A6E0, 0x2B740, 0x2CEAF, 0x2F800, 0x2FA1F, 0x2FA20, 0x10FFFF]
    texts = ["".join(chr(c) for c in cps[i:i + 500]) for i in range(0, len(cps), 500)]
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    ooff = api.count_boundaries(utf8, boff)
    want = np.array([int(api.CharacterType.get_type(chr(c))) for c in cps], dtype=np.uint8)
    assert np.array_equal(pred.char_types_packed(utf8, boff, ooff), want)
    table = fw.table()
    want_fw = np.array([int(api.CharacterType.get_type(chr(table.get(c, c)))) for c in cps], dtype=np.uint8)
    assert np.array_equal(pred.char_types_packed(utf8, boff, ooff, fullwidth=True), want_fw)
