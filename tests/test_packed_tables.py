"""CPU check of the PACKED tables (vaporetto_amd/csrc/tables.cpp, layout.h): a small g++-built walker
(tests/native/tablecheck.cpp, test infrastructure) follows the specialised kernel's lookup protocol over the tables
the host-side compiler emits; its char-pattern scores must equal the oracle's for models without type n-grams."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import cbind
from tests import randmodel
from vaporetto_amd.modelfmt import ModelData, NgramData, WordWeightRecord, encode_model

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "vaporetto_amd", "csrc")
LIB = os.path.join(HERE, "native", "libvpt_tablecheck.so")
SRCS = [os.path.join(HERE, "native", "tablecheck.cpp"), os.path.join(CSRC, "tables.cpp"), os.path.join(CSRC, "model.cpp")]
DEPS = SRCS + [os.path.join(CSRC, h) for h in ("layout.h", "tables.hpp", "model.hpp", "patset.hpp")]


@pytest.fixture(scope="module")
def tc():
    import fcntl
    with open(LIB + ".lock", "w") as lock:   # pytest-xdist workers build it once, not at the same time
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in DEPS):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB + ".tmp"] + SRCS)
            os.replace(LIB + ".tmp", LIB)
    L = C.CDLL(LIB)
    L.tc_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.tc_create_tags.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.tc_destroy.argtypes = [C.c_void_p]
    L.tc_packed_present.argtypes = [C.c_void_p]
    L.tc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 8)]
    L.tc_trow_present.argtypes = [C.c_void_p]
    L.tc_row_window.argtypes = [C.c_void_p]
    L.tc_score.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64 * 4)]
    return L


class Walker:
    def __init__(self, L, raw, predict_tags=False):
        self.L = L
        self.h = C.c_void_p()
        assert L.tc_create_tags(raw, len(raw), int(predict_tags), C.byref(self.h)) == 0

    def __del__(self):
        self.L.tc_destroy(self.h)

    @property
    def packed(self):
        return bool(self.L.tc_packed_present(self.h))

    def stats(self):
        a = (C.c_uint32 * 8)()
        self.L.tc_stats(self.h, C.byref(a))
        return dict(zip(["n_bi", "n_tri", "bi_slots", "n_deep", "tri_slots", "bi_shift", "n_wide", "n_alpha"], list(a)))

    @property
    def trow(self):
        """0 = no type rows, 1 = LDS rows (n-grams of <= 3 symbols, 18-bit totals), 2 = global rows (up to 6 symbols, i32)"""
        return int(self.L.tc_trow_present(self.h))

    @property
    def row_window(self):
        return int(self.L.tc_row_window(self.h))

    def score(self, text, probes=None, want=None):
        """Scores from the packed tables; `want` = 1 (types not included) or 2 (type rows included)."""
        cps = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32).copy()
        y = np.zeros(max(len(cps) - 1, 1), dtype=np.int32)
        pr = (C.c_uint64 * 4)()
        rc = self.L.tc_score(self.h, cps.ctypes.data, len(cps), y.ctypes.data, C.byref(pr))
        assert rc in (1, 2), rc
        if want is not None:
            assert rc == want
        if probes is not None:
            for i in range(4):
                probes[i] += pr[i]
        return y[:len(cps) - 1].tolist()


def strip_types(m: ModelData) -> ModelData:
    m.type_ngram_model = []
    return m


@pytest.mark.parametrize("seed", range(12))
def test_packed_walk_matches_oracle_random_models(tc, seed):
    alphabet = ["kana", "tiny", [chr(c) for c in range(0x3041, 0x3049)] + list("漢字AZ09")][seed % 3]
    m = strip_types(randmodel.rand_model(900 + seed, alphabet=alphabet, wc=3, wt=3, n_char=400, n_dict=500, max_word=11))
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed
    orc = cbind.OraclePredictor(raw)
    probes = [0, 0, 0, 0]
    for t in randmodel.rand_sentences(seed, m, 300, alphabet=alphabet, max_len=60):
        assert w.score(t, probes, want=1) == orc.predict(t)[0], t
    st = w.stats()
    assert st["n_bi"] > 0 and st["n_tri"] > 0 and st["n_deep"] > 0


def test_packed_walk_dense_tables(tc):
    """Tiny alphabets make dense double-array rows (every char continues every char), saturated child filters and deep
    tries: every present key must be found by one node read per level, every absent one must be recognised."""
    alpha = [chr(c) for c in range(0x3041, 0x3051)]
    m = strip_types(randmodel.rand_model(77, alphabet=alpha, wc=3, wt=3, n_char=3000, n_dict=6000, max_word=9))
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed
    st = w.stats()
    assert st["n_alpha"] == 16 and st["n_bi"] == 256 and st["n_tri"] > 2000 and st["tri_slots"] < 2 * st["n_tri"]
    orc = cbind.OraclePredictor(raw)
    probes = [0, 0, 0, 0]
    mixed = alpha + list("漢字カA9、")   # chars outside the alphabet too: kNoId lookups
    for t in randmodel.rand_sentences(5, m, 400, alphabet=mixed, max_len=80):
        assert w.score(t, probes) == orc.predict(t)[0], t
    assert probes[0] > 0 and probes[1] > 0 and probes[2] > 0   # bigram, trigram and deep levels were all walked


def test_packed_walk_sparse_rows_and_filters(tc):
    """A wide alphabet with few patterns: sparse rows interleave in the double arrays (absent keys land on other
    parents' nodes, which the key / parent checks must reject) and the child filters reject most third chars."""
    alpha = [chr(c) for c in range(0x4E00, 0x4E00 + 600)]
    m = strip_types(randmodel.rand_model(78, alphabet=alpha, wc=3, wt=3, n_char=2500, n_dict=2500, max_word=6))
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed
    st = w.stats()
    assert st["bi_slots"] < 80000 + 65536   # far fewer than one 600-slot span per first char: the rows interleave
    orc = cbind.OraclePredictor(raw)
    probes = [0, 0, 0, 0]
    pats = [d.ngram for d in m.char_ngram_model] + [r.word for r in m.dict_model]
    texts = randmodel.rand_sentences(6, m, 500, alphabet=alpha, max_len=60)
    texts += [p[:2] + q[2:] for p, q in zip(pats[:300], pats[300:600]) if len(p) >= 2 and len(q) >= 3]   # right prefix, foreign third char
    for t in texts:
        assert w.score(t, probes) == orc.predict(t)[0], t
    assert probes[3] > 0 and probes[1] > 0          # the filters rejected lookups, and admitted some


def test_a_deep_arena_past_the_kid_filters_reach_is_refused_not_an_error(tc, monkeypatch):
    """the 8-bit child filter leaves a mini-table base 22 bits; a model whose deep arena outgrows them (a NEologd-size dictionary
    of long words) used to be an InvalidModel error -- the reference loads any such model (dict_model.rs:18-50).  It now falls to the
    general tables (present = false) like the other limits of the format.  VPT_DEBUG_KIDS_MAX_BASE lowers the limit for the test."""
    m = randmodel.rand_model(5, alphabet="kana", wc=3, wt=3, n_char=60, n_dict=200, max_word=9)
    raw = encode_model(m)
    assert Walker(tc, raw).packed and Walker(tc, raw).stats()["n_deep"] > 4
    monkeypatch.setenv("VPT_DEBUG_KIDS_MAX_BASE", "1")
    wk = Walker(tc, raw)            # no error ...
    assert not wk.packed            # ... the general tables serve it (GPU parity: test_gpu_parity.py::test_deep_arena_limit_falls_to_the_general_kernels)


def test_packed_eligibility(tc):
    base = dict(bias=3, char_window_size=3, type_window_size=3)
    # chars outside the BMP, U+FFFF and U+FFFE are pattern chars like any other (round 5): ids through the side table `xcid`
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("𠮷", [1, 2, 3, 4, 5, 6]))
    m.char_ngram_model.append(NgramData("𠮷野", [1, 2, 3, 4, 5]))
    m.dict_model.append(WordWeightRecord("a￿", [1, 2, 3], ""))
    m.dict_model.append(WordWeightRecord("𠮷野家の𩸽", [1, 2, 3, 4, 5, 6], ""))
    wk = Walker(tc, encode_model(m))
    assert wk.packed
    orc = cbind.OraclePredictor(encode_model(m))
    for t in ("𠮷", "𠮷野家の𩸽", "a￿a￿", "あ𠮷野𠮷", "𩸽𠮷野家の𩸽𠀋"):
        assert wk.score(t, [0, 0, 0, 0]) == orc.predict(t)[0], t
    # a pattern that holds U+0000 can match no text (sentence.rs:174-179): it is left out, the model stays eligible
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("a\0b", [1, 2, 3, 4]))
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    wk = Walker(tc, encode_model(m))
    assert wk.packed
    assert wk.score("あaあ", [0, 0, 0, 0]) == cbind.OraclePredictor(encode_model(m)).predict("あaあ")[0]
    # every char window up to 8 is eligible: 1 and 2 are laid out in the rows of window 3, wider ones have rows of their own
    m = ModelData(bias=3, char_window_size=4, type_window_size=3)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6, 7, 8]))
    wk = Walker(tc, encode_model(m))
    assert wk.packed and wk.row_window == 4
    for wc, w in ((2, [1, 2, 3, 4]), (1, [5, 6])):
        m = ModelData(bias=3, char_window_size=wc, type_window_size=3)
        m.char_ngram_model.append(NgramData("あ", w))
        m.char_ngram_model.append(NgramData("あい", w[:2 * wc - 1]))
        m.dict_model.append(WordWeightRecord("あいう", [9, 8, 7, 6], ""))
        wk = Walker(tc, encode_model(m))
        assert wk.packed
        orc = cbind.OraclePredictor(encode_model(m))
        for t in ("あ", "あい", "ああいういあ", "いあいうあ"):
            assert wk.score(t, [0, 0, 0, 0]) == orc.predict(t)[0], (wc, t)
    # and the plain eligible case
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    assert Walker(tc, encode_model(m)).packed


def test_packed_wide_rows(tc):
    """Values outside a row's fields (18 bits in a unigram node, 19 in a bigram node, i16 below: a large weight, or an
    n-gram and a word with the same string summing past the field) keep the model on the packed path through the
    wide-row escape; values that only the wider unigram / bigram fields hold are scored in place."""
    m = ModelData(bias=-7, char_window_size=3, type_window_size=3)
    m.char_ngram_model.append(NgramData("あ", [0, 0, 40000, -5, 1, 2]))
    m.char_ngram_model.append(NgramData("あい", [1, 2, 30000, 4, 5]))
    m.dict_model.append(WordWeightRecord("あい", [7, 30000, -9], ""))
    m.char_ngram_model.append(NgramData("いうえ", [32767, 32767, 3, 4]))
    m.dict_model.append(WordWeightRecord("いうえ", [1, 32767, 5, 32767], ""))
    m.dict_model.append(WordWeightRecord("いうえお", [1, 2, 3, 4, 5], ""))
    m.char_ngram_model.append(NgramData("う", [-131073, 0, 131072, -5, 1, 2]))            # just outside 18 bits
    m.char_ngram_model.append(NgramData("え", [-131072, 131071, 0, -5, 1, 2]))            # just inside
    m.char_ngram_model.append(NgramData("いう", [5, 262144, -262145, 1, 2]))              # just outside 19 bits
    m.char_ngram_model.append(NgramData("うえ", [262143, -262144, 262143, -1, -262144]))  # just inside
    m.dict_model.append(WordWeightRecord("あいうえおか", [100000, -100000, 3, 4, 5, 6, 2000000000], ""))
    m.dict_model.append(WordWeightRecord("あいうえおかきくけこ", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -70000], ""))
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed and w.stats()["n_wide"] == 5
    orc = cbind.OraclePredictor(raw)
    for t in ["あいうえおかきくけこ", "ああいいうえお", "いうえ", "あ", "んあいうえおかん", "えおかきあいうえおかきくけこあい"]:
        assert w.score(t) == orc.predict(t)[0], t


def test_packed_text_with_non_bmp_and_ffff_chars(tc):
    m = strip_types(randmodel.rand_model(5, alphabet="kana", wc=3, wt=3, n_char=200, n_dict=200, max_word=8))
    raw = encode_model(m)
    w = Walker(tc, raw)
    orc = cbind.OraclePredictor(raw)
    pats = [d.ngram for d in m.char_ngram_model] + [r.word for r in m.dict_model]
    for i, p in enumerate(pats[:60]):
        for filler in ("𠮷", "￿", "🤌"):
            t = p[: len(p) // 2] + filler + p + filler + pats[(i + 1) % len(pats)]
            assert w.score(t) == orc.predict(t)[0], t


@pytest.mark.parametrize("seed", range(8))
def test_type_rows_match_oracle(tc, seed):
    """Models whose type n-grams have <= 3 symbols get LDS type rows; the walker then reproduces the FULL score."""
    wt = [3, 2, 1, 3][seed % 4]
    m = randmodel.rand_model(300 + seed, alphabet="mixed" if seed % 2 else "kana", wc=3, wt=wt, n_char=60, n_dict=60, n_type=80, max_word=7)
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed and w.trow == 1
    orc = cbind.OraclePredictor(raw)
    for t in randmodel.rand_sentences(seed, m, 300, alphabet="mixed", max_len=50):
        assert w.score(t, want=2) == orc.predict(t)[0], t


def test_type_rows_homes_for_long_or_large_type_ngrams(tc):
    base = dict(bias=1, char_window_size=3, type_window_size=3)
    texts = ["あ", "ああカあ", "あああカあああ漢あ", "カあああ", "AあああカZ"]

    def check(m, want_mode):
        raw = encode_model(m)
        w = Walker(tc, raw)
        assert w.trow == want_mode
        if want_mode:
            orc = cbind.OraclePredictor(raw)
            for t in texts:
                assert w.score(t, want=2) == orc.predict(t)[0], t
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3, 3, 4, 3]), [1, 2, 3]))          # 4 symbols: global rows
    check(m, 2)
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3]), [0, 0, 100000]))
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [0, 100000]))               # merged 200000 > 18 bits: global rows (i32)
    check(m, 2)
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3, 4]), [1, 2, 3, 4, 5]))
    check(m, 1)
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3, 3, 4, 3, 3, 3]), [7]))           # 6 symbols = 2 W: still rows
    m.type_ngram_model.append(NgramData(bytes([3, 3, 4]), [1, 2, 3, 4]))
    check(m, 2)
    m = ModelData(**base)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3, 0]), [1, 2, 3, 4, 5]))           # code 0: only the window table can say "outside"
    check(m, 0)
    m = ModelData(bias=1, char_window_size=3, type_window_size=4)
    m.char_ngram_model.append(NgramData("あ", [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3, 3, 4, 3, 3, 3, 3]), [7, 8]))     # 7 symbols: no rows, the general tables score it
    check(m, 0)


@pytest.mark.parametrize("wc,wt", [(4, 4), (5, 3), (3, 5), (6, 6), (7, 2), (8, 8), (2, 7), (4, 1), (1, 4)])
def test_packed_walk_other_windows(tc, wc, wt):
    """Row windows 4 .. 8 (layout.h, "ROW WINDOW"): wider nodes, the same protocol; type n-grams of up to 4 symbols; the walker
    reproduces the FULL score."""
    for seed in range(3):
        alphabet = ["kana", "tiny", [chr(c) for c in range(0x3041, 0x3049)] + list("漢字AZ09")][seed % 3]
        m = randmodel.rand_model(7000 + 100 * wc + 10 * wt + seed, alphabet=alphabet, wc=wc, wt=wt, max_n=4, n_char=300, n_dict=300, n_type=60, max_word=16)
        raw = encode_model(m)
        w = Walker(tc, raw)
        assert w.packed and w.row_window == max(3, wc, wt) and w.trow in (1, 2)
        orc = cbind.OraclePredictor(raw)
        probes = [0, 0, 0, 0]
        for t in randmodel.rand_sentences(seed, m, 200, alphabet="mixed" if seed == 2 else alphabet, max_len=60):
            assert w.score(t, probes, want=2) == orc.predict(t)[0], t
        assert probes[0] > 0 and probes[1] > 0 and probes[2] > 0


def test_packed_wide_rows_other_windows(tc):
    """Values outside the fields of the wider nodes, and rows past the 14 inline weights of a deep entry."""
    for wc in (4, 6, 8):
        m = ModelData(bias=-7, char_window_size=wc, type_window_size=2)
        m.char_ngram_model.append(NgramData("あ", [0, 0, 40000, -5, 1, 2] + [9] * (2 * wc - 6)))
        m.char_ngram_model.append(NgramData("う", [-131073, 0, 131072, -5, 1, 2] + [-3] * (2 * wc - 6)))          # outside 18 bits
        m.char_ngram_model.append(NgramData("いう", [5, 262144, -262145, 1, 2] + [4] * (2 * wc - 6)))            # outside 19 bits
        m.char_ngram_model.append(NgramData("いうえ", [32767, 32767, 3, 4] + [-32768] * (2 * wc - 6)))
        m.dict_model.append(WordWeightRecord("いうえ", [1, 32767, 5, 32767], ""))                               # sums past i16
        m.char_ngram_model.append(NgramData("いうえお", [11] * (2 * wc - 3)))
        m.dict_model.append(WordWeightRecord("あいうえおか", [100000, -100000, 3, 4, 5, 6, 2000000000], ""))
        m.dict_model.append(WordWeightRecord("あいうえおかきくけこさしすせ", list(range(1, 16)), ""))               # 15 weights: external row
        m.type_ngram_model.append(NgramData(bytes([3, 3]), [1, 2, 3]))
        raw = encode_model(m)
        w = Walker(tc, raw)
        assert w.packed and w.row_window == wc and w.stats()["n_wide"] >= 3
        orc = cbind.OraclePredictor(raw)
        for t in ["あいうえおかきくけこさしすせそ", "ああいいうえお", "いうえ", "あ", "んあいうえおかん", "えおかきあいうえおかきくけこあい", "いうえおいうえお"]:
            assert w.score(t, want=2) == orc.predict(t)[0], (wc, t)


def test_long_words_compressed_chains(tc):
    """Dictionary words far longer than one compressed chain (3 + 1 + 8 chars) and than an inline row (14 weights),
    nested prefixes that own rows in the middle of a chain, and branches inside long words."""
    import random
    rng = random.Random(5)
    alpha = [chr(c) for c in range(0x3041, 0x3049)]
    m = ModelData(bias=11, char_window_size=3, type_window_size=3)
    words = set()
    base = "".join(rng.choice(alpha) for _ in range(40))
    for n in (4, 5, 9, 12, 13, 14, 15, 21, 22, 30, 40):
        words.add(base[:n])                      # nested prefixes of one long string
    for n in (6, 13, 17, 25):
        words.add(base[:n - 1] + "ん")            # branches off the long string
    for _ in range(60):
        words.add("".join(rng.choice(alpha) for _ in range(rng.randint(4, 28))))
    for w in sorted(words):
        m.dict_model.append(WordWeightRecord(w, [rng.randint(-3000, 3000) for _ in range(len(w) + 1)], ""))
    raw = encode_model(m)
    w = Walker(tc, raw)
    assert w.packed
    orc = cbind.OraclePredictor(raw)
    texts = [base, base[:29] + "ん" + base, "あ" + base[:13] + base[:24] + "ん", base[3:] + base]
    texts += ["".join(rng.choice(sorted(words)) for _ in range(3)) for _ in range(80)]
    for t in texts:
        assert w.score(t) == orc.predict(t)[0], t


def test_tag_enabled_models_merge_duplicate_type_ngrams(tc):
    """predict_tags = true with tag models: the reference merges identical type n-grams (TypeWeightMerger) instead of
    rejecting them; the type rows are built from the merged list and must match the oracle in that mode."""
    from vaporetto_amd.modelfmt import TagModel
    m = randmodel.rand_model(41, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=80, n_type=50, max_word=6, n_tag_models=4)
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [100, -200, 300, 400, -500]))
    m.type_ngram_model.append(NgramData(bytes([3, 3]), [7, 8, 9]))                      # duplicate, shorter vector
    m.type_ngram_model.append(NgramData(bytes([3]), [1, 2, 3, 4, 5, 6]))
    m.type_ngram_model.append(NgramData(bytes([3]), [-1, -2, -3, -4, -5, -6]))
    raw = encode_model(m)
    w = Walker(tc, raw, predict_tags=True)
    assert w.packed and w.trow == 1
    orc = cbind.OraclePredictor(raw, True)
    mixed = randmodel.ALPHABETS["kana"][:10] + list("漢字AZ09、")
    for t in randmodel.rand_sentences(4, m, 300, alphabet=mixed, max_len=40):
        assert w.score(t, want=2) == orc.predict(t)[0], t


def test_cpp_fullwidth_map_equals_reference_pairs(tc):
    import ctypes as C
    tc.tc_fullwidth.argtypes = [C.c_uint32]
    tc.tc_fullwidth.restype = C.c_uint32
    golden = {int(a, 16): int(b, 16) for a, b in
              (line.split() for line in open(os.path.join(HERE, "golden", "kytea_fullwidth_pairs.txt"), encoding="utf-8"))}
    for cp in range(0x10000):
        assert tc.tc_fullwidth(cp) == golden.get(cp, cp), hex(cp)


@pytest.mark.parametrize("kind", [1, 2])
def test_packed_walk_on_scaled_synthetic_models(tc, kind):
    """The bench models at 1/20 scale (tens of thousands of patterns: a perfect hash over ~40 K prefixes, overflow
    mini-tables, compressed chains): the table walk reproduces the oracle's FULL scores (type rows included)."""
    from vaporetto_amd import synth
    raw = synth.synth_model(kind, synth.SEED_BASE + kind, 0.05)
    w = Walker(tc, raw)
    assert w.packed and w.trow == 1
    st = w.stats()
    assert st["n_bi"] > 0 and st["n_tri"] > 0 and st["n_deep"] > 0
    utf8, boff = synth.synth_sentences(raw, 400, 8, 96, seed=synth.SEED_BASE + 11 * kind)
    orc = cbind.OraclePredictor(raw)
    text = bytes(utf8)
    for i in range(400):
        t = text[int(boff[i]):int(boff[i + 1])].decode("utf-8")
        assert w.score(t, want=2) == orc.predict(t)[0], t
