"""ORACLE (test infrastructure, never shipped, never on the product path).

Brute-force "sum over all matches" statement of what `vaporetto::Predictor::predict`
computes.  Pure-Python loops: only for small cases.  It is the
independent second opinion next to oracle/vaporetto_oracle.c, which follows the
reference's own algorithm (suffix-merged weights + longest-match automaton).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference lines followed (all under /root/reference/vaporetto/src/):
  * get_type                  sentence.rs:50-67
  * sentence validation       sentence.rs:160-196
  * bias fill, sign threshold predictor.rs:518-543
  * positional add            predictor.rs:176-213  (ys[end + offset + k] += w[k], clipped)
  * char n-gram offset = -W, dict word offset = -len(word in chars)
                              char_scorer/boundary_scorer.rs:56-113
  * type n-grams, automaton   type_scorer/boundary_scorer.rs:45-80
  * type n-grams, cache table type_scorer/boundary_scorer_cache.rs:22-110
  * variant choice            type_scorer.rs:104-144, char_scorer.rs:92-124
  * tag scores + argmax       predictor.rs:264-305,546-637,
                              char_scorer/boundary_tag_scorer.rs:62-174,
                              type_scorer/boundary_tag_scorer.rs:51-143
Pinned by the reference's known-answer vectors in tests/test_oracle_kat.py.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

I32_MASK = 0xFFFFFFFF
CACHE_MAX_WINDOW_SIZE = 3  # type_scorer.rs:35


def wrap_i32(v: int) -> int:
    """--release builds wrap on overflow (README.md:188)."""
    v &= I32_MASK
    return v - (1 << 32) if v & 0x80000000 else v


def get_type(cp: int) -> int:
    """CharacterType::get_type (sentence.rs:50-67). 1..6 = Digit, Roman, Hiragana, Katakana, Kanji, Other."""
    if 0x30 <= cp <= 0x39 or 0xFF10 <= cp <= 0xFF19:
        return 1
    if 0x41 <= cp <= 0x5A or 0x61 <= cp <= 0x7A or 0xFF21 <= cp <= 0xFF3A or 0xFF41 <= cp <= 0xFF5A:
        return 2
    if 0x3040 <= cp <= 0x3096:
        return 3
    if 0x30A0 <= cp <= 0x30FA or 0x30FC <= cp <= 0x30FF or 0xFF66 <= cp <= 0xFF9F:
        return 4
    if (0x3400 <= cp <= 0x4DBF or 0x4E00 <= cp <= 0x9FFF or 0xF900 <= cp <= 0xFAFF
            or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0x2F800 <= cp <= 0x2FA1F):
        return 5
    return 6


def char_types(text: str) -> List[int]:
    return [get_type(ord(c)) for c in text]


def check_text(text: str) -> None:
    """Sentence::parse_raw errors (sentence.rs:172-186)."""
    if "\0" in text:
        raise ValueError("InvalidArgument: text: must not contain NULL")
    if len(text) == 0:
        raise ValueError("InvalidArgument: text: must contain at least one character")


def _add(ys: List[int], start: int, w: Sequence[int]) -> None:
    """PositionalWeight::add_score with clipping to the real boundaries (predictor.rs:176-213;
    the +-7 padding of predictor.rs:519-524 is what absorbs the out-of-range part there)."""
    for k, x in enumerate(w):
        b = start + k
        if 0 <= b < len(ys):
            ys[b] = wrap_i32(ys[b] + x)


def _type_uses_cache(model, predict_tags: bool) -> bool:
    has_tags = predict_tags and len(model.tag_models) > 0
    return (not has_tags) and model.type_window_size <= CACHE_MAX_WINDOW_SIZE


def boundary_scores(model, text: str, predict_tags: bool = False) -> List[int]:
    """i32 score of every boundary of `text` (len(text)-1 values)."""
    check_text(text)
    n = len(text)
    ys = [wrap_i32(model.bias)] * (n - 1)

    # ---- character n-grams and dictionary words (char_scorer/boundary_scorer.rs:56-113)
    wc = model.char_window_size
    if wc != 0 and (model.char_ngram_model or model.dict_model):  # char_scorer.rs:98-100
        for d in model.char_ngram_model:
            g = d.ngram
            for e in range(len(g), n + 1):
                if text[e - len(g):e] == g:
                    _add(ys, e - 1 - wc, d.weights)
        for r in model.dict_model:
            g = r.word
            for e in range(len(g), n + 1):
                if text[e - len(g):e] == g:
                    _add(ys, e - 1 - len(g), r.weights)

    # ---- character-type n-grams
    wt = model.type_window_size
    if wt != 0 and model.type_ngram_model:  # type_scorer.rs:109-111
        t = bytes(char_types(text))
        if _type_uses_cache(model, predict_tags):
            # boundary_scorer_cache.rs:30-49 + 59-81: the window t[b-W+1 .. b+W] is padded
            # with 0 outside the sentence and every pattern fully inside it contributes
            # w[2W - end] when that index exists.
            for b in range(n - 1):
                win = bytes(t[i] if 0 <= i < n else 0 for i in range(b - wt + 1, b + wt + 1))
                for d in model.type_ngram_model:
                    g = bytes(d.ngram)
                    for end in range(len(g), 2 * wt + 1):
                        if win[end - len(g):end] == g:
                            k = 2 * wt - end
                            if k < len(d.weights):
                                ys[b] = wrap_i32(ys[b] + d.weights[k])
        else:
            for d in model.type_ngram_model:
                g = bytes(d.ngram)
                for e in range(len(g), n + 1):
                    if t[e - len(g):e] == g:
                        _add(ys, e - 1 - wt, d.weights)
    return ys


def boundaries(scores: Sequence[int]) -> List[int]:
    """predictor.rs:531-541: 1 = WordBoundary iff score > 0 else 0 = NotWordBoundary."""
    return [1 if s > 0 else 0 for s in scores]


def tokens(text: str, bounds: Sequence[int]) -> List[str]:
    out, start = [], 0
    for i, b in enumerate(bounds):
        if b == 1:
            out.append(text[start:i + 1])
            start = i + 1
    out.append(text[start:])
    return out


def n_tags(model) -> int:
    """predictor.rs:466."""
    return max((len(t.tags) for t in model.tag_models), default=0)


def tag_scores_for_token(model, text: str, tag_model, p: int) -> List[int]:
    """Scores of one token whose last char is text[p] (predictor.rs:573-594 with
    char_scorer/boundary_tag_scorer.rs:154-174 and type_scorer/boundary_tag_scorer.rs:123-143).
    A tag n-gram with rel_position r counts when it ends at char index p + r (inclusive);
    only r in 0..=window are ever looked at, and positions past the sentence end do not exist."""
    n = len(text)
    t = bytes(char_types(text))
    z = list(tag_model.bias)

    def add(ws):
        for i, x in enumerate(ws):
            if i < len(z):
                z[i] = wrap_i32(z[i] + x)

    if model.char_window_size != 0 and (model.char_ngram_model or model.dict_model):
        # CharScorer::new returns None without boundary n-grams/dict words (char_scorer.rs:98-100),
        # which drops the tag char n-grams as well.
        for d in tag_model.char_ngram_model:
            for tw in d.weights:
                r = tw.rel_position
                e = p + r + 1
                if r <= model.char_window_size and e <= n and e >= len(d.ngram) \
                        and text[e - len(d.ngram):e] == d.ngram:
                    add(tw.weights)
    if model.type_window_size != 0 and model.type_ngram_model:
        # TypeScorer::new returns None when there are no boundary type n-grams
        # (type_scorer.rs:109-111), which drops the tag type n-grams as well.
        for d in tag_model.type_ngram_model:
            g = bytes(d.ngram)
            for tw in d.weights:
                r = tw.rel_position
                e = p + r + 1
                if r <= model.type_window_size and e <= n and e >= len(g) and t[e - len(g):e] == g:
                    add(tw.weights)
    return z


def pick_tags(tag_model, z: Sequence[int], slots: int) -> List[Optional[str]]:
    """TagPredictor::predict (predictor.rs:286-304): first maximum wins."""
    out: List[Optional[str]] = [None] * slots
    off = 0
    for j, cands in enumerate(tag_model.tags):
        if j >= slots:
            break
        if len(cands) >= 2:
            idx, best = 0, -(1 << 31)
            for i in range(len(cands)):
                s = z[off + i]
                if s > best:
                    idx, best = i, s
            out[j] = cands[idx]
            off += len(cands)
        else:
            out[j] = cands[0] if cands else None
    return out


def fill_tags(model, text: str, bounds: Sequence[int]) -> List[Optional[str]]:
    """Predictor::predict_tags (predictor.rs:546-637); `bounds` uses 0/1/2 (2 = Unknown)."""
    nt = n_tags(model)
    n = len(text)
    tags: List[Optional[str]] = [None] * (n * nt)
    if nt == 0:
        return tags
    by_token = {}
    for tm in model.tag_models:
        by_token[tm.token] = tm  # HashMap::insert: a repeated token keeps the last model (predictor.rs:468)
    start: Optional[int] = 0
    ends = list(bounds) + [1]
    for i, b in enumerate(ends):
        if b == 2:
            start = None
        elif b == 1:
            if start is not None:
                tm = by_token.get(text[start:i + 1])
                if tm is not None:
                    z = tag_scores_for_token(model, text, tm, i)
                    tags[i * nt:(i + 1) * nt] = pick_tags(tm, z, nt)
            start = i + 1
    return tags
