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


# ------------------------------------------------------------------------------------------------------------------------------------
# The C restatement is what every large-scale parity claim rests on (bench.py checks hundreds of millions of GPU scores against it), and
# both oracles were written from one reading of the reference.  What keeps a shared misreading out is the reference's own vectors
# (tests/kat.py); what keeps the two implementations from drifting apart in the corners those vectors do not reach is this sweep:
# 2 400 models built to hit them -- windows 0 .. 8 on either side, n-grams up to 2 W symbols, the same string as n-gram AND dictionary
# word AND twice over, chains of patterns that are suffixes of each other (what CharWeightMerger::merge folds, char_scorer.rs:50-78),
# words longer than the sentence and overhanging both of its ends (char_scorer.rs:322-401), weight vectors shorter than allowed,
# 1-char sentences, tag models whose n-grams reach past the sentence.
import random

from vaporetto_amd.modelfmt import ModelData, NgramData, TagModel, TagNgramData, TagWeight, WordWeightRecord


def adversarial_model(seed: int, tags: bool):
    rng = random.Random(0xC0FFEE + seed)
    alpha = list(rng.choice(["あい", "あい漢", "あいカ漢A9", "あいうえおカキ漢字AZ09、。"]))
    wc, wt = rng.randint(0, 8), rng.randint(0, 8)
    if tags:
        wc, wt = max(wc, 1), max(wt, 1)
    m = ModelData(bias=rng.randint(-40000, 40000), char_window_size=wc, type_window_size=wt)

    def weights(full, window=0):
        # shorter vectors than the bound are legal -- except under window 8, where a vector of <= 8 weights is a Fixed([i32; 8]) that the
        # reference adds at end + 6 - W, in front of its 7 padding slots for a match at the sentence start: it panics (predictor.rs:176-213,
        # :519), the C restatement reports that as VO_INTERNAL, and no trainer emits it (trainer.rs:416,433: always the full length)
        n = rng.randint(1, full) if rng.random() < 0.4 and window < 8 else full
        return [rng.randint(-32767, 32767) if rng.random() < 0.85 else 0 for _ in range(n)]
    chars = []
    if wc:
        for _ in range(rng.randint(1, 14)):
            n = rng.randint(1, min(2 * wc, 6))
            g = "".join(rng.choice(alpha) for _ in range(n))
            chars.append(g)
            for k in range(1, n):                                          # every suffix of it, sometimes: the merger's input
                if rng.random() < 0.5:
                    chars.append(g[k:])
        for g in list(chars):
            if rng.random() < 0.15:
                chars.append(g)                                            # the same n-gram twice: summed (CharWeightMerger::add)
        for g in chars:
            full = 2 * wc - len(g) + 1
            if full > 0:
                m.char_ngram_model.append(NgramData(g, weights(full, wc)))
    words = set()
    for _ in range(rng.randint(0, 10)):
        n = rng.choice([1, 1, 2, 2, 3, 4, 5, 7, 12, 20])
        words.add("".join(rng.choice(alpha) for _ in range(n)))
    for g in chars:
        if rng.random() < 0.3:
            words.add(g)                                                   # an n-gram that is a dictionary word too
    for w in sorted(words):
        for k in range(1, len(w)):
            if rng.random() < 0.2:
                words.add(w[k:])
    for w in sorted(words):
        m.dict_model.append(WordWeightRecord(w, weights(len(w) + 1), ""))
    seen = set()
    if wt:
        for _ in range(rng.randint(0, 12)):
            n = rng.randint(1, min(2 * wt, 6))
            g = bytes(rng.choice([2, 3, 3, 4, 5, 5, 6, 1]) for _ in range(n))
            dup_ok = tags or wt > 3                                        # only the automaton variants merge duplicates (type_scorer.rs:46-56)
            if g in seen and not dup_ok:
                continue
            seen.add(g)
            m.type_ngram_model.append(NgramData(g, weights(2 * wt - n + 1, wt)))
    if tags:
        toks = set()
        for _ in range(rng.randint(1, 8)):
            toks.add("".join(rng.choice(alpha) for _ in range(rng.randint(1, 3))))
        for tok in sorted(toks):
            slots = [["t%d" % k for k in range(rng.randint(0, 4))] for _ in range(rng.randint(1, 3))]
            zlen = sum(len(s) for s in slots if len(s) >= 2)
            tm = TagModel(tok, slots, bias=[rng.randint(-999, 999) for _ in range(zlen)])
            for _ in range(rng.randint(0, 4)):
                left, extra = rng.randint(0, 3), rng.randint(0, min(wc, 3))
                g = "".join(rng.choice(alpha) for _ in range(left)) + tok + "".join(rng.choice(alpha) for _ in range(extra))
                ws = [TagWeight(r, [rng.randint(-999, 999) for _ in range(zlen)]) for r in sorted(set(rng.randint(0, wc) for _ in range(2)))]
                tm.char_ngram_model.append(TagNgramData(g, ws))
            for _ in range(rng.randint(0, 3)):
                g = bytes(rng.choice([3, 4, 5, 6, 2, 1]) for _ in range(rng.randint(1, 4)))
                ws = [TagWeight(r, [rng.randint(-999, 999) for _ in range(zlen)]) for r in sorted(set(rng.randint(0, wt) for _ in range(2)))]
                tm.type_ngram_model.append(TagNgramData(g, ws))
            m.tag_models.append(tm)
    pats = chars + sorted(words) + [t.token for t in m.tag_models]
    texts = []
    for _ in range(10):
        kind = rng.random()
        if kind < 0.2 and pats:                                            # a piece out of the middle of a pattern: it overhangs both ends
            p = rng.choice(pats)
            a = rng.randint(0, len(p) - 1)
            texts.append(p[a:rng.randint(a + 1, len(p))])
        elif kind < 0.3:
            texts.append(rng.choice(alpha))                                # one char: no boundary at all
        else:
            out = []
            for _ in range(rng.randint(1, 6)):
                out.append(rng.choice(pats) if pats and rng.random() < 0.7 else rng.choice(alpha))
            texts.append("".join(out)[:rng.randint(1, 40)])
    return m, texts


@pytest.mark.parametrize("chunk", range(16))
def test_adversarial_models_boundaries(chunk):
    for seed in range(chunk * 100, chunk * 100 + 100):
        m, texts = adversarial_model(seed, tags=False)
        p = cbind.OraclePredictor(encode_model(m))
        for text in texts:
            expected = spec.boundary_scores(m, text)
            scores, labels = p.predict(text)
            assert scores == expected, (seed, text)
            assert labels == spec.boundaries(expected)


@pytest.mark.parametrize("chunk", range(8))
def test_adversarial_models_tags(chunk):
    for seed in range(100_000 + chunk * 100, 100_000 + chunk * 100 + 100):
        m, texts = adversarial_model(seed, tags=True)
        p = cbind.OraclePredictor(encode_model(m), predict_tags=True)
        for text in texts:
            expected = spec.boundary_scores(m, text, predict_tags=True)
            scores, labels = p.predict(text)
            assert scores == expected, (seed, text)
            tags, nt = p.predict_tags(text)
            assert nt == spec.n_tags(m)
            assert tag_strings(m, text, labels, tags, nt) == spec.fill_tags(m, text, labels), (seed, text)
