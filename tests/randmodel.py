"""Seeded random small models and sentences for oracle-vs-oracle and GPU-vs-oracle parity tests.

Small alphabets make patterns hit often, overlap, nest as suffixes of each other and overhang the
sentence edges -- the situations the reference's tests probe (char_scorer.rs:322-401)."""
from __future__ import annotations

import random

from vaporetto_amd.modelfmt import ModelData, NgramData, TagModel, TagNgramData, TagWeight, WordWeightRecord

ALPHABETS = {
    # every CharacterType occurs; 1-, 2-, 3- and 4-byte UTF-8; BMP and non-BMP
    "mixed": list("あいうえカキ漢字人世界09AZaz、。 -") + ["𠮷", "🤌", "é", "ß", "１", "Ａ", "ｱ"],
    "tiny": list("あい漢"),
    "kana": [chr(c) for c in range(0x3041, 0x3061)],
}


def rand_weights(rng, n, big=False):
    hi = 2_000_000_000 if big else 32767
    return [rng.randint(-hi, hi) if rng.random() < 0.8 else 0 for _ in range(n)]


def rand_model(seed: int, alphabet="mixed", wc=None, wt=None, max_n=3, n_char=30, n_type=20, n_dict=25,
               max_word=7, n_tag_models=0, big=False) -> ModelData:
    rng = random.Random(seed)
    alpha = ALPHABETS[alphabet] if isinstance(alphabet, str) else alphabet
    wc = rng.randint(1, 4) if wc is None else wc
    wt = rng.randint(1, 4) if wt is None else wt
    m = ModelData(bias=rng.randint(-50000, 50000), char_window_size=wc, type_window_size=wt)
    seen = set()
    for _ in range(n_char):
        n = rng.randint(1, max_n)
        g = "".join(rng.choice(alpha) for _ in range(n))
        full = 2 * wc - n + 1
        if g in seen or full <= 0:
            continue
        seen.add(g)
        m.char_ngram_model.append(NgramData(g, rand_weights(rng, rng.randint(max(1, full - 2), full), big)))
    seen = set()
    for _ in range(n_type):
        n = rng.randint(1, max_n)
        g = bytes(rng.randint(1, 6) for _ in range(n))
        full = 2 * wt - n + 1
        if g in seen or full <= 0:
            continue
        seen.add(g)
        m.type_ngram_model.append(NgramData(g, rand_weights(rng, rng.randint(max(1, full - 2), full), big)))
    seen = set()
    for _ in range(n_dict):
        n = rng.randint(1, max_word)
        g = "".join(rng.choice(alpha) for _ in range(n))
        if g in seen:
            continue
        seen.add(g)
        m.dict_model.append(WordWeightRecord(g, rand_weights(rng, n + 1, big), "c"))
    seen = set()
    for _ in range(n_tag_models):
        n = rng.randint(1, 3)
        tok = "".join(rng.choice(alpha) for _ in range(n))
        if tok in seen:
            continue
        seen.add(tok)
        slots = []
        for _ in range(rng.randint(1, 3)):
            slots.append(["t%d" % k for k in range(rng.randint(0, 4))])
        zlen = sum(len(s) for s in slots if len(s) >= 2)
        tm = TagModel(tok, slots, bias=rand_weights(rng, zlen))
        for _ in range(rng.randint(0, 4)):
            extra = rng.randint(0, 2)
            left = rng.randint(0, 2)
            g = "".join(rng.choice(alpha) for _ in range(left)) + tok + "".join(rng.choice(alpha) for _ in range(extra))
            ws = [TagWeight(r, rand_weights(rng, zlen)) for r in sorted(set(rng.randint(0, wc) for _ in range(2)))]
            tm.char_ngram_model.append(TagNgramData(g, ws))
        for _ in range(rng.randint(0, 4)):
            g = bytes(rng.randint(1, 6) for _ in range(rng.randint(1, 4)))
            ws = [TagWeight(r, rand_weights(rng, zlen)) for r in sorted(set(rng.randint(0, wt) for _ in range(2)))]
            tm.type_ngram_model.append(TagNgramData(g, ws))
        m.tag_models.append(tm)
    return m


def rand_text(rng, alpha, min_len=1, max_len=40, patterns=()):
    n = rng.randint(min_len, max_len)
    out = []
    while len(out) < n:
        if patterns and rng.random() < 0.5:
            out.extend(rng.choice(patterns))
        else:
            out.append(rng.choice(alpha))
    return "".join(out[:n])


def rand_sentences(seed: int, model: ModelData, count: int, alphabet="mixed", min_len=1, max_len=40):
    rng = random.Random(seed ^ 0x5EED)
    alpha = ALPHABETS[alphabet] if isinstance(alphabet, str) else alphabet
    pats = [d.ngram for d in model.char_ngram_model] + [r.word for r in model.dict_model]
    return [rand_text(rng, alpha, min_len, max_len, pats) for _ in range(count)]
