/*
 * ORACLE -- test infrastructure.  NOT part of the product, never linked into it.
 *
 * A plain-C, CPU restatement of the REFERENCE'S OWN ALGORITHM for the hot path
 * `vaporetto::Predictor::predict` (paths below are under /root/reference/vaporetto/src/):
 *
 *   model decoding        model.rs:15,58-70,127-153; ngram_model.rs:6-27; dict_model.rs:18-22
 *                         (bincode 2.0.1 "standard" config; crate not vendored -- layout pinned by
 *                          the fixture files tests/golden/ *.bin being consumed to the last byte)
 *   PositionalWeight +=   predictor.rs:149-165
 *   add_score             predictor.rs:176-213 (Fixed [i32;8] when len <= 8, else Variable, padding 7)
 *   weight merger         char_scorer.rs:33-78, type_scorer.rs:42-88 (suffix weights pre-added)
 *   char scorer           char_scorer/boundary_scorer.rs:56-113 (n-gram offset -W, dict offset -len)
 *   char scorer (tags)    char_scorer/boundary_tag_scorer.rs:62-147
 *   type scorer variants  type_scorer.rs:104-144
 *   type cache table      type_scorer/boundary_scorer_cache.rs:22-110
 *   type automaton        type_scorer/boundary_scorer.rs:45-80, boundary_tag_scorer.rs:51-116
 *   predict               predictor.rs:518-543
 *   tag prediction        predictor.rs:264-305,546-637; boundary_tag_scorer.rs add_tag_scores; the stored scores of
 *                         Predictor::store_tag_scores (predictor.rs:510-514,599-601) / Token::tag_candidates (sentence.rs:1218-1250)
 *   tokenized text        Sentence::write_tokenized_text, sentence.rs:850-886 (tokens, escaping, "/tag" up to the last Some)
 *   char classes          sentence.rs:50-67; sentence checks sentence.rs:160-196
 *
 * The pattern matcher `daachorse 1.0.0` (Cargo.toml:17) is a third-party crate that is not in
 * /root/reference.  Its contract at the call sites is restated here as a plain Aho-Corasick
 * automaton: `find_overlapping_no_suffix_iter` reports, for every haystack position where at
 * least one pattern ends, exactly the LONGEST such pattern; `new` fails on an empty or duplicate
 * pattern.  Parity is pinned by the reference's known-answer tests (tests/test_oracle_c_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#define _GNU_SOURCE   /* pthread_setaffinity_np, sched_getaffinity (the timed baseline pins its workers) */
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>

#define VO_OK 0
#define VO_INVALID_MODEL 1
#define VO_INVALID_ARGUMENT 2
#define VO_INTERNAL 3

#define WEIGHT_FIXED_LEN 8 /* predictor.rs:32 */
#define NONE_ID 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t m) {
    void *p = calloc(n ? n : 1, m ? m : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void set_err(char *err, size_t errlen, const char *msg) {
    if (err && errlen) { strncpy(err, msg, errlen - 1); err[errlen - 1] = 0; }
}

/* ------------------------------------------------------------------------------------------ */
/* bincode reader (model.rs:127-153)                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t *p; size_t n, pos; int bad; } rd_t;

static uint8_t rd_u8(rd_t *r) {
    if (r->pos + 1 > r->n) { r->bad = 1; return 0; }
    return r->p[r->pos++];
}
static uint64_t rd_uvar(rd_t *r) {
    uint8_t t = rd_u8(r);
    int nb;
    if (t < 251) return t;
    if (t == 251) nb = 2; else if (t == 252) nb = 4; else if (t == 253) nb = 8; else { r->bad = 1; return 0; }
    if (r->pos + (size_t)nb > r->n) { r->bad = 1; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < nb; i++) v |= (uint64_t)r->p[r->pos + i] << (8 * i);
    r->pos += nb;
    return v;
}
static int32_t rd_i32(rd_t *r) {
    uint64_t u = rd_uvar(r);
    if (u > 0xFFFFFFFFull) { r->bad = 1; return 0; }
    return (int32_t)((uint32_t)(u >> 1) ^ (uint32_t)(-(int32_t)(u & 1)));
}
static size_t rd_len(rd_t *r) {
    uint64_t v = rd_uvar(r);
    if (v > r->n - r->pos) { r->bad = 1; return 0; }
    return (size_t)v;
}

typedef struct { int32_t *w; uint32_t len; } wvec;
static wvec rd_weights(rd_t *r) {
    wvec v; v.len = (uint32_t)rd_len(r); v.w = (int32_t *)xmalloc(sizeof(int32_t) * v.len);
    for (uint32_t i = 0; i < v.len && !r->bad; i++) v.w[i] = rd_i32(r);
    return v;
}

/* decodes UTF-8 bytes into code points; returns count or -1 when malformed */
static long utf8_decode(const uint8_t *s, size_t n, uint32_t *out) {
    size_t i = 0; long k = 0;
    while (i < n) {
        uint32_t c = s[i], cp; int extra;
        if (c < 0x80) { cp = c; extra = 0; }
        else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
        else return -1;
        if (extra > 0 && i + (size_t)extra > n - 1) return -1;
        for (int j = 1; j <= extra; j++) {
            if ((s[i + j] & 0xC0) != 0x80) return -1;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        if (out) out[k] = cp;
        k++; i += (size_t)extra + 1;
    }
    return k;
}

/* a symbol string: code points for char patterns, type ids for type patterns */
typedef struct { uint32_t *s; uint32_t len; } symstr;

static symstr rd_string_cp(rd_t *r) { /* String -> code points */
    symstr v = {0, 0};
    size_t n = rd_len(r);
    if (r->bad) return v;
    v.s = (uint32_t *)xmalloc(sizeof(uint32_t) * (n + 1));
    long k = utf8_decode(r->p + r->pos, n, v.s);
    if (k < 0) { r->bad = 1; k = 0; }
    v.len = (uint32_t)k; r->pos += n;
    return v;
}
static symstr rd_bytes_sym(rd_t *r) { /* Vec<u8> -> symbols */
    symstr v = {0, 0};
    size_t n = rd_len(r);
    if (r->bad) return v;
    v.s = (uint32_t *)xmalloc(sizeof(uint32_t) * (n + 1));
    for (size_t i = 0; i < n; i++) v.s[i] = r->p[r->pos + i];
    v.len = (uint32_t)n; r->pos += n;
    return v;
}
static void rd_skip_string(rd_t *r) { size_t n = rd_len(r); r->pos += n; }
static uint8_t *rd_string_raw(rd_t *r, uint32_t *len_out) { /* String -> its UTF-8 bytes (copied) */
    size_t n = rd_len(r);
    if (r->bad) { *len_out = 0; return NULL; }
    uint8_t *q = (uint8_t *)xmalloc(n + 1);
    memcpy(q, r->p + r->pos, n);
    r->pos += n;
    *len_out = (uint32_t)n;
    return q;
}

/* ------------------------------------------------------------------------------------------ */
/* decoded model (model.rs:58-70)                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct { symstr g; wvec w; } ngram_rec;
typedef struct { symstr g; uint32_t nw; uint8_t *rel; wvec *w; } tag_ngram_rec;
typedef struct {
    symstr token;
    uint32_t n_slots; uint32_t *n_cands; /* only the candidate counts matter for scoring */
    uint32_t *cand_first;                /* per slot: index of its first candidate in cand_str / cand_len (the writer needs the strings) */
    uint8_t **cand_str; uint32_t *cand_len; uint32_t n_cand_total;
    uint32_t n_char; tag_ngram_rec *chr;
    uint32_t n_type; tag_ngram_rec *typ;
    wvec bias;
} tag_model_rec;
typedef struct {
    uint32_t n_char; ngram_rec *chr;
    uint32_t n_type; ngram_rec *typ;
    uint32_t n_dict; ngram_rec *dict;
    int32_t bias; uint8_t char_w, type_w;
    uint32_t n_tag; tag_model_rec *tag;
} model_t;

static const char MODEL_MAGIC[] = "VaporettoTokenizer 0.5.0\n";

static tag_ngram_rec *rd_tag_ngrams(rd_t *r, uint32_t *n, int is_char) {
    *n = (uint32_t)rd_len(r);
    tag_ngram_rec *a = (tag_ngram_rec *)xcalloc(*n, sizeof(tag_ngram_rec));
    for (uint32_t i = 0; i < *n && !r->bad; i++) {
        a[i].g = is_char ? rd_string_cp(r) : rd_bytes_sym(r);
        a[i].nw = (uint32_t)rd_len(r);
        a[i].rel = (uint8_t *)xcalloc(a[i].nw, 1);
        a[i].w = (wvec *)xcalloc(a[i].nw, sizeof(wvec));
        for (uint32_t j = 0; j < a[i].nw && !r->bad; j++) { a[i].rel[j] = rd_u8(r); a[i].w[j] = rd_weights(r); }
    }
    return a;
}

static int model_decode(const uint8_t *buf, size_t len, model_t *m) {
    size_t ml = sizeof(MODEL_MAGIC) - 1;
    memset(m, 0, sizeof(*m));
    if (len < ml || memcmp(buf, MODEL_MAGIC, ml) != 0) return -1; /* "model version mismatch" */
    rd_t r = {buf, len, ml, 0};
    m->n_char = (uint32_t)rd_len(&r);
    m->chr = (ngram_rec *)xcalloc(m->n_char, sizeof(ngram_rec));
    for (uint32_t i = 0; i < m->n_char && !r.bad; i++) { m->chr[i].g = rd_string_cp(&r); m->chr[i].w = rd_weights(&r); }
    m->n_type = (uint32_t)rd_len(&r);
    m->typ = (ngram_rec *)xcalloc(m->n_type, sizeof(ngram_rec));
    for (uint32_t i = 0; i < m->n_type && !r.bad; i++) { m->typ[i].g = rd_bytes_sym(&r); m->typ[i].w = rd_weights(&r); }
    m->n_dict = (uint32_t)rd_len(&r);
    m->dict = (ngram_rec *)xcalloc(m->n_dict, sizeof(ngram_rec));
    for (uint32_t i = 0; i < m->n_dict && !r.bad; i++) {
        m->dict[i].g = rd_string_cp(&r); m->dict[i].w = rd_weights(&r); rd_skip_string(&r); /* comment */
    }
    m->bias = rd_i32(&r); m->char_w = rd_u8(&r); m->type_w = rd_u8(&r);
    m->n_tag = (uint32_t)rd_len(&r);
    m->tag = (tag_model_rec *)xcalloc(m->n_tag, sizeof(tag_model_rec));
    for (uint32_t i = 0; i < m->n_tag && !r.bad; i++) {
        tag_model_rec *t = &m->tag[i];
        t->token = rd_string_cp(&r);
        t->n_slots = (uint32_t)rd_len(&r);
        t->n_cands = (uint32_t *)xcalloc(t->n_slots, sizeof(uint32_t));
        t->cand_first = (uint32_t *)xcalloc(t->n_slots + 1, sizeof(uint32_t));
        for (uint32_t j = 0; j < t->n_slots && !r.bad; j++) {
            t->n_cands[j] = (uint32_t)rd_len(&r);
            t->cand_first[j] = t->n_cand_total;
            t->cand_str = (uint8_t **)xrealloc(t->cand_str, sizeof(uint8_t *) * (t->n_cand_total + t->n_cands[j] + 1));
            t->cand_len = (uint32_t *)xrealloc(t->cand_len, sizeof(uint32_t) * (t->n_cand_total + t->n_cands[j] + 1));
            for (uint32_t k = 0; k < t->n_cands[j] && !r.bad; k++) {
                t->cand_str[t->n_cand_total] = rd_string_raw(&r, &t->cand_len[t->n_cand_total]);
                t->n_cand_total++;
            }
        }
        t->chr = rd_tag_ngrams(&r, &t->n_char, 1);
        t->typ = rd_tag_ngrams(&r, &t->n_type, 0);
        t->bias = rd_weights(&r);
    }
    return r.bad ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* PositionalWeight (predictor.rs:137-165) and its tag-carrying variant (predictor.rs:215-262) */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t offset; int32_t *w; uint32_t len; } posw;

/* self += other  (predictor.rs:149-165) */
static void posw_add_assign(posw *a, const posw *b) {
    int32_t new_off = a->offset < b->offset ? a->offset : b->offset;
    uint32_t shift = (uint32_t)(a->offset - new_off);
    uint32_t oshift = (uint32_t)(b->offset - new_off);
    uint32_t new_size = shift + a->len;
    if (oshift + b->len > new_size) new_size = oshift + b->len;
    int32_t *w = (int32_t *)xcalloc(new_size, sizeof(int32_t));
    for (uint32_t i = 0; i < a->len; i++) w[shift + i] = a->w[i];
    for (uint32_t i = 0; i < b->len; i++) w[oshift + i] = (int32_t)((uint32_t)w[oshift + i] + (uint32_t)b->w[i]);
    free(a->w);
    a->w = w; a->len = new_size; a->offset = new_off;
}
static posw posw_clone(const posw *a) {
    posw c = *a; c.w = (int32_t *)xmalloc(sizeof(int32_t) * a->len);
    memcpy(c.w, a->w, sizeof(int32_t) * a->len);
    return c;
}

typedef struct { uint32_t token_id; uint8_t rel; int32_t *w; uint32_t len; } taginfo;

/* one BTreeMap entry of the merger: key -> (weight, merged flag)                               */
typedef struct {
    symstr key;
    int has_w; posw pw;                   /* PositionalWeightWithTag::weight (None for tag-only patterns) */
    taginfo *tags; uint32_t n_tags;       /* PositionalWeightWithTag::tag_info                             */
    uint64_t abytes;                      /* accounting only: sum of 4*len(w) over the un-merged records  */
    int merged;
    uint32_t seq;                         /* insertion order, keeps the sort stable                       */
} ment;

static int sym_cmp(const symstr *a, const symstr *b) {
    uint32_t n = a->len < b->len ? a->len : b->len;
    for (uint32_t i = 0; i < n; i++) if (a->s[i] != b->s[i]) return a->s[i] < b->s[i] ? -1 : 1;
    return a->len < b->len ? -1 : (a->len > b->len ? 1 : 0);
}
static int ment_cmp(const void *x, const void *y) {
    const ment *a = (const ment *)x, *b = (const ment *)y;
    int c = sym_cmp(&a->key, &b->key);
    if (c) return c;
    return a->seq < b->seq ? -1 : (a->seq > b->seq ? 1 : 0);
}

/* PositionalWeightWithTag += (predictor.rs:242-262) */
static void ment_add_assign(ment *a, const ment *b) {
    if (a->has_w) { if (b->has_w) posw_add_assign(&a->pw, &b->pw); }
    else if (b->has_w) { a->pw = posw_clone(&b->pw); a->has_w = 1; }
    for (uint32_t i = 0; i < b->n_tags; i++) {
        const taginfo *t = &b->tags[i];
        uint32_t j;
        for (j = 0; j < a->n_tags; j++) if (a->tags[j].token_id == t->token_id && a->tags[j].rel == t->rel) break;
        if (j < a->n_tags) { /* and_modify: zip-add */
            uint32_t n = a->tags[j].len < t->len ? a->tags[j].len : t->len;
            for (uint32_t k = 0; k < n; k++) a->tags[j].w[k] = (int32_t)((uint32_t)a->tags[j].w[k] + (uint32_t)t->w[k]);
        } else {
            a->tags = (taginfo *)xrealloc(a->tags, sizeof(taginfo) * (a->n_tags + 1));
            taginfo c = *t; c.w = (int32_t *)xmalloc(sizeof(int32_t) * t->len);
            memcpy(c.w, t->w, sizeof(int32_t) * t->len);
            a->tags[a->n_tags++] = c;
        }
    }
    a->abytes += b->abytes;
}

typedef struct { ment *e; uint32_t n, cap; } merger;

static void merger_push(merger *mg, const symstr *key, int has_w, int32_t offset, const wvec *w,
                        int has_tag, uint32_t token_id, uint8_t rel, int count_bytes) {
    if (mg->n == mg->cap) { mg->cap = mg->cap ? mg->cap * 2 : 1024; mg->e = (ment *)xrealloc(mg->e, sizeof(ment) * mg->cap); }
    ment *m = &mg->e[mg->n];
    memset(m, 0, sizeof(*m));
    m->key = *key; m->seq = mg->n;
    if (has_w) {
        m->has_w = 1; m->pw.offset = offset; m->pw.len = w->len;
        m->pw.w = (int32_t *)xmalloc(sizeof(int32_t) * w->len);
        memcpy(m->pw.w, w->w, sizeof(int32_t) * w->len);
        if (count_bytes) m->abytes = 4ull * w->len;
    }
    if (has_tag) {
        m->tags = (taginfo *)xmalloc(sizeof(taginfo)); m->n_tags = 1;
        m->tags[0].token_id = token_id; m->tags[0].rel = rel; m->tags[0].len = w->len;
        m->tags[0].w = (int32_t *)xmalloc(sizeof(int32_t) * w->len);
        memcpy(m->tags[0].w, w->w, sizeof(int32_t) * w->len);
    }
    mg->n++;
}

/* `add` for every record (char_scorer.rs:37-47): equal keys are summed; then sorted like a BTreeMap */
static void merger_unique(merger *mg) {
    qsort(mg->e, mg->n, sizeof(ment), ment_cmp);
    uint32_t o = 0;
    for (uint32_t i = 0; i < mg->n; i++) {
        if (o > 0 && sym_cmp(&mg->e[o - 1].key, &mg->e[i].key) == 0) ment_add_assign(&mg->e[o - 1], &mg->e[i]);
        else mg->e[o++] = mg->e[i];
    }
    mg->n = o;
}
static long merger_find(const merger *mg, const uint32_t *s, uint32_t len) {
    symstr k; k.s = (uint32_t *)s; k.len = len;
    long lo = 0, hi = (long)mg->n - 1;
    while (lo <= hi) {
        long mid = (lo + hi) / 2;
        int c = sym_cmp(&mg->e[mid].key, &k);
        if (c == 0) return mid;
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
/* `merge` (char_scorer.rs:50-78 / type_scorer.rs:59-87): walk each pattern's proper suffixes, longest first,
 * stop after the first already-merged one, then fold the shorter into the longer. */
static void merger_merge(merger *mg) {
    uint32_t *stack = (uint32_t *)xmalloc(sizeof(uint32_t) * 16); uint32_t scap = 16;
    for (uint32_t i = 0; i < mg->n; i++) {
        ment *e = &mg->e[i];
        if (e->merged) continue;
        uint32_t sp = 0;
        stack[sp++] = i;
        for (uint32_t j = 1; j < e->key.len; j++) {
            long f = merger_find(mg, e->key.s + j, e->key.len - j);
            if (f >= 0) {
                if (sp == scap) { scap *= 2; stack = (uint32_t *)xrealloc(stack, sizeof(uint32_t) * scap); }
                stack[sp++] = (uint32_t)f;
                if (mg->e[f].merged) break;
            }
        }
        uint32_t from = stack[--sp];
        mg->e[from].merged = 1;
        while (sp > 0) {
            uint32_t to = stack[--sp];
            mg->e[to].merged = 1;
            ment_add_assign(&mg->e[to], &mg->e[from]);
            from = to;
        }
    }
    free(stack);
}

/* ------------------------------------------------------------------------------------------ */
/* pattern matcher: Aho-Corasick restating daachorse's find_overlapping_no_suffix_iter contract */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; uint32_t next; uint32_t used; } tr_slot;
typedef struct {
    tr_slot *tab; uint64_t mask; int shift;
    uint32_t *fail, *output; uint32_t n_states;
} pma_t;

static inline uint64_t tr_hash(uint64_t k, int shift) { return (k * 0x9E3779B97F4A7C15ull) >> shift; }
static inline uint32_t pma_goto(const pma_t *a, uint32_t state, uint32_t sym) {
    uint64_t key = ((uint64_t)state << 32) | sym;
    uint64_t h = tr_hash(key, a->shift);
    for (;;) {
        const tr_slot *s = &a->tab[h];
        if (!s->used) return NONE_ID;
        if (s->key == key) return s->next;
        h = (h + 1) & a->mask;
    }
}
static void pma_put(pma_t *a, uint32_t state, uint32_t sym, uint32_t next) {
    uint64_t key = ((uint64_t)state << 32) | sym;
    uint64_t h = tr_hash(key, a->shift);
    while (a->tab[h].used) h = (h + 1) & a->mask;
    a->tab[h].key = key; a->tab[h].next = next; a->tab[h].used = 1;
}

/* patterns must be unique and non-empty (daachorse `new` rejects otherwise); value = index */
static int pma_build(pma_t *a, const symstr *pats, uint32_t n) {
    uint64_t total = 1;
    for (uint32_t i = 0; i < n; i++) { if (pats[i].len == 0) return -1; total += pats[i].len; }
    uint64_t cap = 16; int bits = 4;
    while (cap < total * 2) { cap <<= 1; bits++; }
    memset(a, 0, sizeof(*a));
    a->tab = (tr_slot *)xcalloc(cap, sizeof(tr_slot)); a->mask = cap - 1; a->shift = 64 - bits;
    uint32_t *own = (uint32_t *)xmalloc(sizeof(uint32_t) * total);
    uint32_t *parent = (uint32_t *)xmalloc(sizeof(uint32_t) * total);
    uint32_t *psym = (uint32_t *)xmalloc(sizeof(uint32_t) * total);
    uint32_t *depth = (uint32_t *)xmalloc(sizeof(uint32_t) * total);
    uint32_t ns = 1, maxd = 0; own[0] = NONE_ID; parent[0] = 0; psym[0] = 0; depth[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t st = 0;
        for (uint32_t j = 0; j < pats[i].len; j++) {
            uint32_t nx = pma_goto(a, st, pats[i].s[j]);
            if (nx == NONE_ID) {
                nx = ns++; own[nx] = NONE_ID; parent[nx] = st; psym[nx] = pats[i].s[j]; depth[nx] = depth[st] + 1;
                if (depth[nx] > maxd) maxd = depth[nx];
                pma_put(a, st, pats[i].s[j], nx);
            }
            st = nx;
        }
        if (own[st] != NONE_ID) return -1; /* duplicate pattern */
        own[st] = i;
    }
    a->n_states = ns;
    a->fail = (uint32_t *)xcalloc(ns, sizeof(uint32_t));
    a->output = (uint32_t *)xmalloc(sizeof(uint32_t) * ns);
    /* breadth-first order by counting sort on depth */
    uint32_t *cnt = (uint32_t *)xcalloc(maxd + 2, sizeof(uint32_t));
    for (uint32_t s = 0; s < ns; s++) cnt[depth[s] + 1]++;
    for (uint32_t d = 0; d <= maxd; d++) cnt[d + 1] += cnt[d];
    uint32_t *order = (uint32_t *)xmalloc(sizeof(uint32_t) * ns);
    for (uint32_t s = 0; s < ns; s++) order[cnt[depth[s]]++] = s;
    a->output[0] = NONE_ID;
    for (uint32_t k = 1; k < ns; k++) {
        uint32_t v = order[k], f = 0;
        if (parent[v] != 0) {
            f = a->fail[parent[v]];
            for (;;) {
                uint32_t g = pma_goto(a, f, psym[v]);
                if (g != NONE_ID) { f = g; break; }
                if (f == 0) break;
                f = a->fail[f];
            }
        }
        a->fail[v] = f;
        /* longest pattern that is a suffix of this state's string */
        a->output[v] = own[v] != NONE_ID ? own[v] : a->output[f];
    }
    free(own); free(parent); free(psym); free(depth); free(cnt); free(order);
    return 0;
}
static inline uint32_t pma_step(const pma_t *a, uint32_t state, uint32_t sym) {
    for (;;) {
        uint32_t g = pma_goto(a, state, sym);
        if (g != NONE_ID) return g;
        if (state == 0) return 0;
        state = a->fail[state];
    }
}
static void pma_free(pma_t *a) { free(a->tab); free(a->fail); free(a->output); memset(a, 0, sizeof(*a)); }

/* ------------------------------------------------------------------------------------------ */
/* The same automaton as a DOUBLE ARRAY -- what the reference's matcher is (daachorse's CharwiseDoubleArrayAhoCorasick behind
 * char_scorer/boundary_scorer.rs:76-99: 16-byte states {base, check, fail, output}, a char -> code map by frequency, child of s by
 * code c at base[s] ^ c, there if its check names s).  BASELINE LEG ONLY (bench.py's cpu_baseline; flag VO_FLAG_DOUBLE_ARRAY of
 * vo_predict_batch_ex): the checker stays the hash-table automaton above, and tests/test_oracle_c_kat.py holds the two against each
 * other on the reference's known answers and on random text.  Built FROM the automaton above (same states, failure links, outputs),
 * so it is the same function of the text by construction; what changes is the memory it walks: 16 bytes per state, dense, where the
 * hash table spends 32+ per transition -- at 10 M sentences on 256 threads that is the difference VERDICT r4 (weak 5) asked about. */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t base, check, fail, output; } da_state;
typedef struct {
    da_state *st; uint32_t n;          /* positions in use: [0, n) */
    uint32_t *code_of; uint32_t n_codes_map;   /* scalar value -> code (0: no pattern holds the char), direct for [0, n_codes_map) */
} da_t;
static void da_free(da_t *d) { if (!d) return; free(d->st); free(d->code_of); free(d); }

typedef struct { uint32_t state, sym, next; } da_edge;
static int da_edge_cmp(const void *x, const void *y) {
    const da_edge *a = (const da_edge *)x, *b = (const da_edge *)y;
    if (a->state != b->state) return a->state < b->state ? -1 : 1;
    return a->sym < b->sym ? -1 : (a->sym > b->sym ? 1 : 0);
}
static da_t *da_build(const pma_t *a) {
    const uint32_t ns = a->n_states;
    /* the edges, by state */
    uint64_t n_edges = 0;
    for (uint64_t h = 0; h <= a->mask; h++) n_edges += a->tab[h].used ? 1 : 0;
    da_edge *e = (da_edge *)xmalloc(sizeof(da_edge) * (n_edges ? n_edges : 1));
    uint64_t k = 0; uint32_t max_sym = 0;
    for (uint64_t h = 0; h <= a->mask; h++) if (a->tab[h].used) {
        e[k].state = (uint32_t)(a->tab[h].key >> 32); e[k].sym = (uint32_t)a->tab[h].key; e[k].next = a->tab[h].next;
        if (e[k].sym > max_sym) max_sym = e[k].sym;
        k++;
    }
    qsort(e, n_edges, sizeof(da_edge), da_edge_cmp);
    /* codes by how many edges carry the symbol (most first: the children of a state then sit close to its base) */
    da_t *d = (da_t *)xcalloc(1, sizeof(da_t));
    d->n_codes_map = max_sym + 1;
    d->code_of = (uint32_t *)xcalloc(d->n_codes_map, sizeof(uint32_t));
    uint32_t *cnt = (uint32_t *)xcalloc(d->n_codes_map, sizeof(uint32_t));
    for (uint64_t i = 0; i < n_edges; i++) cnt[e[i].sym]++;
    uint32_t n_codes = 0;
    for (uint32_t c = 0; c <= max_sym; c++) if (cnt[c]) n_codes++;
    uint32_t *by = (uint32_t *)xmalloc(sizeof(uint32_t) * (n_codes ? n_codes : 1));
    { uint32_t j = 0; for (uint32_t c = 0; c <= max_sym; c++) if (cnt[c]) by[j++] = c; }
    /* (insertion-free: sort the symbols by count, descending, ties by value) */
    for (uint32_t gap = n_codes / 2; gap > 0; gap /= 2)
        for (uint32_t i = gap; i < n_codes; i++) {
            uint32_t v = by[i]; uint32_t j = i;
            while (j >= gap && (cnt[by[j - gap]] < cnt[v] || (cnt[by[j - gap]] == cnt[v] && by[j - gap] > v))) { by[j] = by[j - gap]; j -= gap; }
            by[j] = v;
        }
    for (uint32_t j = 0; j < n_codes; j++) d->code_of[by[j]] = j + 1;
    free(by); free(cnt);
    /* first edge of every state */
    uint64_t *first = (uint64_t *)xmalloc(sizeof(uint64_t) * ((uint64_t)ns + 1));
    { uint64_t i = 0; for (uint32_t s = 0; s <= ns; s++) { while (i < n_edges && e[i].state < s) i++; first[s] = i; } }
    /* placement: states in the order they were made (parents before children), children at base ^ code; first fit from the lowest
     * free position, in blocks of `blk` positions (base ^ code stays inside the block of `base` when code < blk) */
    uint32_t blk = 1; while (blk <= n_codes) blk <<= 1;
    uint64_t cap = ((uint64_t)ns * 5 / 4 + 2 * (uint64_t)blk + 63) & ~(uint64_t)(blk - 1);
    uint8_t *used = (uint8_t *)xcalloc(cap, 1);
    uint32_t *pos = (uint32_t *)xmalloc(sizeof(uint32_t) * ns);
    d->st = (da_state *)xcalloc(cap, sizeof(da_state));
    for (uint64_t i = 0; i < cap; i++) d->st[i].check = NONE_ID;
    pos[0] = 0; used[0] = 1;
    uint64_t low = 1, top = 1;
    uint32_t *codes = (uint32_t *)xmalloc(sizeof(uint32_t) * (n_codes ? n_codes : 1));
    /* a state with several children does not start its search further back than a window behind where the last state of its size class
     * went: what did not take that one will hardly take this one, and the holes left behind are filled by the one-child states, which fit
     * anywhere -- keeps the placement linear in the number of states (the same device as vaporetto_amd/csrc/tables.cpp's Placer) */
    uint64_t hint[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t lookback = 4096;
    for (uint32_t s = 0; s < ns; s++) {       /* (state ids grow along every path: a parent is placed before its children) */
        const uint64_t e0 = first[s], e1 = first[s + 1];
        if (e0 == e1) continue;
        const uint32_t nk = (uint32_t)(e1 - e0);
        for (uint32_t j = 0; j < nk; j++) codes[j] = d->code_of[e[e0 + j].sym];
        while (low < cap && used[low]) low++;
        uint64_t base = 0; int found = 0;
        const uint32_t cls = nk < 8 ? nk : 8;
        uint64_t from = low;
        if (nk > 1 && hint[cls] > lookback && hint[cls] - lookback > from) from = hint[cls] - lookback;
        /* candidate bases: those that put the first child on a free position, from `from` on */
        for (uint64_t q = from; !found; q++) {
            if (q + blk >= cap) {   /* grow */
                uint64_t ncap = cap * 2;
                used = (uint8_t *)xrealloc(used, ncap); memset(used + cap, 0, ncap - cap);
                d->st = (da_state *)xrealloc(d->st, sizeof(da_state) * ncap);
                for (uint64_t i = cap; i < ncap; i++) { d->st[i].base = 0; d->st[i].check = NONE_ID; d->st[i].fail = 0; d->st[i].output = 0; }
                cap = ncap;
            }
            if (used[q]) continue;
            const uint64_t b = q ^ codes[0];
            int ok = 1;
            for (uint32_t j = 1; j < nk && ok; j++) ok = !used[b ^ codes[j]];
            if (ok) { base = b; found = 1; hint[cls] = q; }
        }
        d->st[pos[s]].base = (uint32_t)base;
        for (uint32_t j = 0; j < nk; j++) {
            const uint64_t t = base ^ codes[j];
            used[t] = 1; pos[e[e0 + j].next] = (uint32_t)t;
            d->st[t].check = pos[s];
            if (t + 1 > top) top = t + 1;
        }
    }
    for (uint32_t s = 0; s < ns; s++) { d->st[pos[s]].fail = pos[a->fail[s]]; d->st[pos[s]].output = a->output[s]; }
    d->n = (uint32_t)top;
    free(codes); free(used); free(pos); free(first); free(e);
    return d;
}
static inline uint32_t da_step(const da_t *d, uint32_t st, uint32_t sym) {
    const uint32_t c = sym < d->n_codes_map ? d->code_of[sym] : 0;
    if (c == 0) return 0;               /* a char of no pattern: every state fails down to the root, which has no such child */
    for (;;) {
        const uint32_t t = d->st[st].base ^ c;
        if (d->st[t].check == st) return t;
        if (st == 0) return 0;
        st = d->st[st].fail;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* scorer = automaton + PositionalWeight<WeightVector> per pattern                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int present;          /* Option<weight> (tag-only patterns have none) */
    int fixed;            /* WeightVector::Fixed when len <= 8 (predictor.rs:118-135) */
    int32_t offset; uint32_t len; uint64_t woff; /* into scorer->wdata */
    uint64_t abytes;
} pw_rec;
typedef struct { uint32_t pattern; uint32_t token_id; uint8_t rel; uint32_t len; uint64_t woff; } tagw_rec;
typedef struct {
    pma_t pma;
    pw_rec *pw; uint32_t n_pat;
    int32_t *wdata; uint64_t n_wdata;
    tagw_rec *tagw; uint32_t n_tagw;  /* sorted by (token_id, rel, pattern): tag_weight[token][rel].get(pattern) */
    int record_states;                /* BoundaryTag variants store the per-position pattern ids */
    uint32_t window;
    da_t *da;                         /* the automaton as a double array: made on first use by the baseline leg (VO_FLAG_DOUBLE_ARRAY), else NULL */
} scorer_t;

static int tagw_cmp(const void *x, const void *y) {
    const tagw_rec *a = (const tagw_rec *)x, *b = (const tagw_rec *)y;
    if (a->token_id != b->token_id) return a->token_id < b->token_id ? -1 : 1;
    if (a->rel != b->rel) return a->rel < b->rel ? -1 : 1;
    if (a->pattern != b->pattern) return a->pattern < b->pattern ? -1 : 1;
    return 0;
}

static int scorer_from_merger(scorer_t *sc, merger *mg, uint32_t window, int record_states) {
    memset(sc, 0, sizeof(*sc));
    sc->window = window; sc->record_states = record_states;
    merger_unique(mg);
    merger_merge(mg);
    symstr *pats = (symstr *)xmalloc(sizeof(symstr) * mg->n);
    uint64_t nw = 0; uint32_t nt = 0;
    for (uint32_t i = 0; i < mg->n; i++) {
        pats[i] = mg->e[i].key;
        if (mg->e[i].has_w) nw += mg->e[i].pw.len <= WEIGHT_FIXED_LEN ? WEIGHT_FIXED_LEN : mg->e[i].pw.len;
        for (uint32_t j = 0; j < mg->e[i].n_tags; j++) { nw += mg->e[i].tags[j].len; nt++; }
    }
    if (pma_build(&sc->pma, pats, mg->n) != 0) { free(pats); return -1; }
    free(pats);
    sc->n_pat = mg->n;
    sc->pw = (pw_rec *)xcalloc(mg->n, sizeof(pw_rec));
    sc->wdata = (int32_t *)xcalloc(nw, sizeof(int32_t));
    sc->tagw = (tagw_rec *)xcalloc(nt, sizeof(tagw_rec));
    uint64_t wo = 0; uint32_t to = 0;
    for (uint32_t i = 0; i < mg->n; i++) {
        ment *e = &mg->e[i];
        pw_rec *p = &sc->pw[i];
        p->abytes = e->abytes;
        if (e->has_w) {
            p->present = 1; p->offset = e->pw.offset; p->len = e->pw.len; p->woff = wo;
            p->fixed = e->pw.len <= WEIGHT_FIXED_LEN;
            memcpy(sc->wdata + wo, e->pw.w, sizeof(int32_t) * e->pw.len);
            wo += p->fixed ? WEIGHT_FIXED_LEN : e->pw.len;
        }
        for (uint32_t j = 0; j < e->n_tags; j++) {
            tagw_rec *t = &sc->tagw[to++];
            t->pattern = i; t->token_id = e->tags[j].token_id; t->rel = e->tags[j].rel; t->len = e->tags[j].len; t->woff = wo;
            memcpy(sc->wdata + wo, e->tags[j].w, sizeof(int32_t) * e->tags[j].len);
            wo += e->tags[j].len;
        }
    }
    sc->n_wdata = wo; sc->n_tagw = to;
    qsort(sc->tagw, sc->n_tagw, sizeof(tagw_rec), tagw_cmp);
    return 0;
}
static void scorer_free(scorer_t *sc) {
    da_free(sc->da);
    pma_free(&sc->pma); free(sc->pw); free(sc->wdata); free(sc->tagw); memset(sc, 0, sizeof(*sc));
}
static void merger_free(merger *mg) {
    for (uint32_t i = 0; i < mg->n; i++) {
        if (mg->e[i].has_w) free(mg->e[i].pw.w);
        for (uint32_t j = 0; j < mg->e[i].n_tags; j++) free(mg->e[i].tags[j].w);
        free(mg->e[i].tags);
    }
    free(mg->e); memset(mg, 0, sizeof(*mg));
}

/* PositionalWeight<WeightVector>::add_score (predictor.rs:176-213); ys has `ylen` entries.
 * Returns -1 where the reference would panic (Fixed slice out of range). */
static inline int add_score(const scorer_t *sc, const pw_rec *p, long end, int32_t *ys, long ylen) {
    long pos = end + p->offset;
    const int32_t *w = sc->wdata + p->woff;
    if (p->fixed) {
        if (pos < 0 || pos + WEIGHT_FIXED_LEN > ylen) return -1;
        for (int k = 0; k < WEIGHT_FIXED_LEN; k++) ys[pos + k] = (int32_t)((uint32_t)ys[pos + k] + (uint32_t)w[k]);
    } else if (pos >= 0) {
        for (long k = 0; k < (long)p->len && pos + k < ylen; k++) ys[pos + k] = (int32_t)((uint32_t)ys[pos + k] + (uint32_t)w[k]);
    } else if ((uint64_t)(-pos) <= p->len) {
        long skip = -pos;
        for (long k = 0; skip + k < (long)p->len && k < ylen; k++) ys[k] = (int32_t)((uint32_t)ys[k] + (uint32_t)w[skip + k]);
    }
    return 0;
}

/* CharScorerBoundary(Tag)::add_scores / TypeScorerBoundary(Tag)::add_scores:
 * one automaton step per symbol, the longest pattern ending there adds its merged weights. */
static int scorer_add_scores(const scorer_t *sc, const uint32_t *syms, long n, int32_t *ys, long ylen, long padding,
                             uint32_t *states, uint64_t *abytes) {
    uint32_t st = 0;
    int rc = 0;
    uint64_t ab = 0;
    if (states) for (long i = 0; i < n; i++) states[i] = NONE_ID;
    for (long i = 0; i < n; i++) {
        st = pma_step(&sc->pma, st, syms[i]);
        uint32_t m = sc->pma.output[st];
        if (m != NONE_ID) {
            const pw_rec *p = &sc->pw[m];
            if (p->present && add_score(sc, p, (i + 1) + padding - 1, ys, ylen) != 0) rc = -1;
            ab += p->abytes;
            if (states) states[i] = m;
        }
    }
    if (abytes) *abytes += ab;
    return rc;
}

/* ... over the double array (baseline leg): the same loop, the other representation of the same automaton */
static int scorer_add_scores_da(const scorer_t *sc, const uint32_t *syms, long n, int32_t *ys, long ylen, long padding,
                                uint32_t *states, uint64_t *abytes) {
    const da_t *d = sc->da;
    uint32_t st = 0;
    int rc = 0;
    uint64_t ab = 0;
    if (states) for (long i = 0; i < n; i++) states[i] = NONE_ID;
    for (long i = 0; i < n; i++) {
        st = da_step(d, st, syms[i]);
        uint32_t m = d->st[st].output;
        if (m != NONE_ID) {
            const pw_rec *p = &sc->pw[m];
            if (p->present && add_score(sc, p, (i + 1) + padding - 1, ys, ylen) != 0) rc = -1;
            ab += p->abytes;
            if (states) states[i] = m;
        }
    }
    if (abytes) *abytes += ab;
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* type score cache (type_scorer/boundary_scorer_cache.rs)                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t *scores; uint32_t window; uint64_t mask; } tcache_t;

static int tcache_build(tcache_t *tc, const model_t *m) {
    uint32_t W = m->type_w, L = 2 * W;
    /* DoubleArrayAhoCorasick::new over the UNMERGED n-grams: duplicate or empty patterns are an error
     * (boundary_scorer_cache.rs:23-24) */
    for (uint32_t i = 0; i < m->n_type; i++) {
        if (m->typ[i].g.len == 0) return -1;
        for (uint32_t j = 0; j < i; j++) if (sym_cmp(&m->typ[i].g, &m->typ[j].g) == 0) return -1;
    }
    uint64_t all = 1ull << (3 * L);
    tc->scores = (int32_t *)xcalloc(all, sizeof(int32_t));
    tc->window = W; tc->mask = all - 1;
    /* scores[seq] = sum over every pattern occurrence inside the 2W window of w[2W - end] (rs:30-49).
     * Equivalent enumeration: for each pattern and each placement, visit the windows that contain it;
     * windows holding the invalid code 7 keep score 0, free positions range over 0..6. */
    for (uint32_t i = 0; i < m->n_type; i++) {
        const symstr *g = &m->typ[i].g; const wvec *w = &m->typ[i].w;
        int ok = 1;
        for (uint32_t j = 0; j < g->len; j++) if (g->s[j] > 6) ok = 0; /* never occurs in a window */
        if (!ok || g->len > L) continue;
        for (uint32_t end = g->len; end <= L; end++) {
            uint32_t k = L - end;
            if (k >= w->len) continue;
            int32_t x = w->w[k];
            uint32_t nfree = L - g->len;
            uint64_t combos = 1;
            for (uint32_t f = 0; f < nfree; f++) combos *= 7;
            for (uint64_t c = 0; c < combos; c++) {
                uint64_t rest = c, seq = 0;
                for (uint32_t pos = 0; pos < L; pos++) { /* pos 0 = leftmost = most significant */
                    uint32_t sym;
                    if (pos >= end - g->len && pos < end) sym = g->s[pos - (end - g->len)];
                    else { sym = (uint32_t)(rest % 7); rest /= 7; }
                    seq = (seq << 3) | sym;
                }
                tc->scores[seq] = (int32_t)((uint32_t)tc->scores[seq] + (uint32_t)x);
            }
        }
    }
    return 0;
}
/* add_scores (boundary_scorer_cache.rs:59-81) */
static void tcache_add_scores(const tcache_t *tc, const uint32_t *types, long n, int32_t *ys, long padding) {
    uint64_t seqid = 0;
    for (uint32_t i = 0; i < tc->window; i++) {
        uint64_t t = (long)i < n ? types[i] : 0;
        seqid = ((seqid << 3) | t) & tc->mask;
    }
    for (long b = 0; b < n - 1; b++) {
        long j = b + tc->window;
        uint64_t t = j < n ? types[j] : 0;
        seqid = ((seqid << 3) | t) & tc->mask;
        ys[padding + b] = (int32_t)((uint32_t)ys[padding + b] + (uint32_t)tc->scores[seqid]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* predictor (predictor.rs:307-316, 450-543)                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct vo_predictor {
    model_t model;
    int has_char; scorer_t chr;
    int type_kind; /* 0 none, 1 cache, 2 automaton */
    scorer_t typ; tcache_t tcache;
    int32_t bias;
    int predict_tags; uint32_t n_tags;
    /* tag_predictor: HashMap<String, (token id, TagPredictor)> (predictor.rs:466-478): open addressing over the token's
     * code points; an insert of an equal token replaces the model index, so the LAST model of a repeated token wins */
    uint32_t *tokmap; uint32_t tokmask; uint32_t max_zlen;
} vo_predictor;

static uint8_t get_type(uint32_t c) { /* sentence.rs:50-67 */
    if ((c >= 0x30 && c <= 0x39) || (c >= 0xFF10 && c <= 0xFF19)) return 1;
    if ((c >= 0x41 && c <= 0x5A) || (c >= 0x61 && c <= 0x7A) || (c >= 0xFF21 && c <= 0xFF3A) || (c >= 0xFF41 && c <= 0xFF5A)) return 2;
    if (c >= 0x3040 && c <= 0x3096) return 3;
    if ((c >= 0x30A0 && c <= 0x30FA) || (c >= 0x30FC && c <= 0x30FF) || (c >= 0xFF66 && c <= 0xFF9F)) return 4;
    if ((c >= 0x3400 && c <= 0x4DBF) || (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0xF900 && c <= 0xFAFF) ||
        (c >= 0x20000 && c <= 0x2A6DF) || (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) ||
        (c >= 0x2B820 && c <= 0x2CEAF) || (c >= 0x2F800 && c <= 0x2FA1F)) return 5;
    return 6;
}

void vo_predictor_destroy(vo_predictor *p);

static uint32_t tok_hash(const uint32_t *s, uint32_t n) {
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; i++) { h ^= s[i]; h *= 16777619u; h ^= h >> 15; }
    return h;
}
/* tag_predictor.get(token): index of the tag model whose token is s[0..n), or -1 */
static long tok_find(const vo_predictor *p, const uint32_t *s, long n) {
    if (!p->tokmap) return -1;
    for (uint32_t h = tok_hash(s, (uint32_t)n) & p->tokmask;; h = (h + 1) & p->tokmask) {
        uint32_t e = p->tokmap[h];
        if (e == 0) return -1;
        const symstr *tk = &p->model.tag[e - 1].token;
        if ((long)tk->len == n && memcmp(tk->s, s, sizeof(uint32_t) * tk->len) == 0) return (long)e - 1;
    }
}

int vo_predictor_create(const uint8_t *bytes, size_t len, int predict_tags, vo_predictor **out, char *err, size_t errlen) {
    vo_predictor *p = (vo_predictor *)xcalloc(1, sizeof(vo_predictor));
    *out = NULL;
    int rc = model_decode(bytes, len, &p->model);
    if (rc == -1) { set_err(err, errlen, "InvalidModelError: model version mismatch"); vo_predictor_destroy(p); return VO_INVALID_MODEL; }
    if (rc != 0) { set_err(err, errlen, "DecodeError: malformed model data"); vo_predictor_destroy(p); return VO_INVALID_MODEL; }
    const model_t *m = &p->model;
    p->bias = m->bias; p->predict_tags = predict_tags;
    /* predictor.rs:463-479: tag models only take part when predict_tags is set */
    uint32_t n_tagm = predict_tags ? m->n_tag : 0;
    for (uint32_t i = 0; i < n_tagm; i++) if (m->tag[i].n_slots > p->n_tags) p->n_tags = m->tag[i].n_slots;

    /* CharScorer::new (char_scorer.rs:92-124) */
    if (!((m->n_char == 0 && m->n_dict == 0) || m->char_w == 0)) {
        merger mg; memset(&mg, 0, sizeof(mg));
        for (uint32_t i = 0; i < m->n_char; i++)
            merger_push(&mg, &m->chr[i].g, 1, -(int32_t)m->char_w, &m->chr[i].w, 0, 0, 0, 1);
        for (uint32_t i = 0; i < m->n_dict; i++) {
            if (m->dict[i].g.len > 32767) {
                set_err(err, errlen, "InvalidModelError: words must be shorter than or equal to 32767 characters");
                merger_free(&mg); vo_predictor_destroy(p); return VO_INVALID_MODEL;
            }
            merger_push(&mg, &m->dict[i].g, 1, -(int32_t)m->dict[i].g.len, &m->dict[i].w, 0, 0, 0, 1);
        }
        for (uint32_t t = 0; t < n_tagm; t++)
            for (uint32_t i = 0; i < m->tag[t].n_char; i++)
                for (uint32_t j = 0; j < m->tag[t].chr[i].nw; j++) {
                    if (m->tag[t].chr[i].rel[j] > m->char_w) { /* index out of bounds in the reference (boundary_tag_scorer.rs:100-103) */
                        set_err(err, errlen, "reference panics: tag rel_position exceeds the window"); merger_free(&mg); vo_predictor_destroy(p); return VO_INVALID_MODEL;
                    }
                    merger_push(&mg, &m->tag[t].chr[i].g, 0, 0, &m->tag[t].chr[i].w[j], 1, t, m->tag[t].chr[i].rel[j], 0);
                }
        rc = scorer_from_merger(&p->chr, &mg, m->char_w, n_tagm > 0);
        merger_free(&mg);
        if (rc != 0) { set_err(err, errlen, "InvalidModelError: failed to build the automaton"); vo_predictor_destroy(p); return VO_INVALID_MODEL; }
        p->has_char = 1;
    }
    /* TypeScorer::new (type_scorer.rs:104-144) */
    if (!(m->n_type == 0 || m->type_w == 0)) {
        if (n_tagm == 0 && m->type_w <= 3) {
            if (tcache_build(&p->tcache, m) != 0) { set_err(err, errlen, "InvalidModelError: invalid character type n-grams"); vo_predictor_destroy(p); return VO_INVALID_MODEL; }
            p->type_kind = 1;
        } else {
            merger mg; memset(&mg, 0, sizeof(mg));
            for (uint32_t i = 0; i < m->n_type; i++)
                merger_push(&mg, &m->typ[i].g, 1, -(int32_t)m->type_w, &m->typ[i].w, 0, 0, 0, 1);
            for (uint32_t t = 0; t < n_tagm; t++)
                for (uint32_t i = 0; i < m->tag[t].n_type; i++)
                    for (uint32_t j = 0; j < m->tag[t].typ[i].nw; j++) {
                        if (m->tag[t].typ[i].rel[j] > m->type_w) {
                            set_err(err, errlen, "reference panics: tag rel_position exceeds the window"); merger_free(&mg); vo_predictor_destroy(p); return VO_INVALID_MODEL;
                        }
                        merger_push(&mg, &m->tag[t].typ[i].g, 0, 0, &m->tag[t].typ[i].w[j], 1, t, m->tag[t].typ[i].rel[j], 0);
                    }
            rc = scorer_from_merger(&p->typ, &mg, m->type_w, n_tagm > 0);
            merger_free(&mg);
            if (rc != 0) { set_err(err, errlen, "InvalidModelError: failed to build the automaton"); vo_predictor_destroy(p); return VO_INVALID_MODEL; }
            p->type_kind = 2;
        }
    }
    if (n_tagm) {
        uint32_t cap = 16;
        while (cap < 2 * n_tagm) cap <<= 1;
        p->tokmap = (uint32_t *)xcalloc(cap, sizeof(uint32_t));
        p->tokmask = cap - 1;
        for (uint32_t t = 0; t < n_tagm; t++) {
            const symstr *tk = &m->tag[t].token;
            if (m->tag[t].bias.len > p->max_zlen) p->max_zlen = m->tag[t].bias.len;
            uint32_t h = tok_hash(tk->s, tk->len) & p->tokmask;
            for (;; h = (h + 1) & p->tokmask) {
                uint32_t e = p->tokmap[h];
                if (e == 0 || (m->tag[e - 1].token.len == tk->len && memcmp(m->tag[e - 1].token.s, tk->s, sizeof(uint32_t) * tk->len) == 0)) { p->tokmap[h] = t + 1; break; }
            }
        }
    }
    *out = p;
    return VO_OK;
}

static void free_symstr(symstr *s) { free(s->s); }
static void free_tag_ngrams(tag_ngram_rec *a, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) { free_symstr(&a[i].g); for (uint32_t j = 0; j < a[i].nw; j++) free(a[i].w[j].w); free(a[i].w); free(a[i].rel); }
    free(a);
}
void vo_predictor_destroy(vo_predictor *p) {
    if (!p) return;
    model_t *m = &p->model;
    for (uint32_t i = 0; i < m->n_char; i++) { free_symstr(&m->chr[i].g); free(m->chr[i].w.w); }
    for (uint32_t i = 0; i < m->n_type; i++) { free_symstr(&m->typ[i].g); free(m->typ[i].w.w); }
    for (uint32_t i = 0; i < m->n_dict; i++) { free_symstr(&m->dict[i].g); free(m->dict[i].w.w); }
    free(m->chr); free(m->typ); free(m->dict);
    for (uint32_t i = 0; i < m->n_tag; i++) {
        free_symstr(&m->tag[i].token); free(m->tag[i].n_cands); free(m->tag[i].cand_first);
        for (uint32_t k = 0; k < m->tag[i].n_cand_total; k++) free(m->tag[i].cand_str[k]);
        free(m->tag[i].cand_str); free(m->tag[i].cand_len);
        free_tag_ngrams(m->tag[i].chr, m->tag[i].n_char); free_tag_ngrams(m->tag[i].typ, m->tag[i].n_type);
        free(m->tag[i].bias.w);
    }
    free(m->tag);
    if (p->has_char) scorer_free(&p->chr);
    if (p->type_kind == 2) scorer_free(&p->typ);
    free(p->tcache.scores);
    free(p->tokmap);
    free(p);
}

/* per-thread scratch: what a reused `Sentence` owns (sentence.rs:85-101) */
typedef struct {
    uint32_t *cps, *types, *cstates, *tstates; int32_t *ys; long cap;
} scratch_t;
static void scratch_reserve(scratch_t *s, long n) {
    if (n <= s->cap) return;
    long c = s->cap ? s->cap : 256;
    while (c < n) c *= 2;
    s->cps = (uint32_t *)xrealloc(s->cps, sizeof(uint32_t) * c);
    s->types = (uint32_t *)xrealloc(s->types, sizeof(uint32_t) * c);
    s->cstates = (uint32_t *)xrealloc(s->cstates, sizeof(uint32_t) * c);
    s->tstates = (uint32_t *)xrealloc(s->tstates, sizeof(uint32_t) * c);
    s->ys = (int32_t *)xrealloc(s->ys, sizeof(int32_t) * (c + 2 * (WEIGHT_FIXED_LEN - 1)));
    s->cap = c;
}
static void scratch_free(scratch_t *s) { free(s->cps); free(s->types); free(s->cstates); free(s->tstates); free(s->ys); memset(s, 0, sizeof(*s)); }

/* Sentence::parse_raw + Predictor::predict for one sentence.  Returns the number of chars (>=1) or a
 * negative status.  scores/labels receive n-1 entries. */
static long predict_one_ex(const vo_predictor *p, const uint8_t *utf8, size_t len, scratch_t *s,
                           int32_t *scores, uint8_t *labels, uint64_t *abytes, int use_da) {
    if (len == 0) return -VO_INVALID_ARGUMENT;                  /* "must contain at least one character" */
    scratch_reserve(s, (long)len);
    long n = utf8_decode(utf8, len, s->cps);
    if (n <= 0) return -VO_INVALID_ARGUMENT;
    for (long i = 0; i < n; i++) {
        if (s->cps[i] == 0) return -VO_INVALID_ARGUMENT;        /* "must not contain NULL" */
        s->types[i] = get_type(s->cps[i]);
    }
    const long pad = WEIGHT_FIXED_LEN - 1;                      /* predictor.rs:519 */
    long ylen = 2 * pad + n - 1;
    for (long i = 0; i < ylen; i++) s->ys[i] = p->bias;         /* predictor.rs:520-524 */
    int rc = 0;
    if (p->has_char) rc |= (use_da && p->chr.da ? scorer_add_scores_da : scorer_add_scores)(&p->chr, s->cps, n, s->ys, ylen, pad, p->chr.record_states ? s->cstates : NULL, abytes);
    if (p->type_kind == 1) tcache_add_scores(&p->tcache, s->types, n, s->ys, pad);
    else if (p->type_kind == 2) rc |= scorer_add_scores(&p->typ, s->types, n, s->ys, ylen, pad, p->typ.record_states ? s->tstates : NULL, NULL);
    if (rc != 0) return -VO_INTERNAL;                           /* the reference would have panicked */
    for (long b = 0; b < n - 1; b++) {                          /* predictor.rs:531-541 */
        int32_t y = s->ys[pad + b];
        if (scores) scores[b] = y;
        if (labels) labels[b] = y > 0 ? 1 : 0;
    }
    return n;
}

static long predict_one(const vo_predictor *p, const uint8_t *utf8, size_t len, scratch_t *s,
                        int32_t *scores, uint8_t *labels, uint64_t *abytes) {
    return predict_one_ex(p, utf8, len, s, scores, labels, abytes, 0);
}

int vo_predict(const vo_predictor *p, const uint8_t *utf8, size_t len, int32_t *scores, uint8_t *labels, size_t *n_boundaries) {
    scratch_t s; memset(&s, 0, sizeof(s));
    long n = predict_one(p, utf8, len, &s, scores, labels, NULL);
    scratch_free(&s);
    if (n < 0) return (int)-n;
    if (n_boundaries) *n_boundaries = (size_t)(n - 1);
    return VO_OK;
}

/* number of chars of every sentence -> out_offsets[i] = sum_{j<i} (n_j - 1) */
int vo_count_boundaries(const uint8_t *utf8, const uint64_t *byte_offsets, size_t S, uint64_t *out_offsets) {
    uint64_t acc = 0;
    for (size_t i = 0; i < S; i++) {
        out_offsets[i] = acc;
        uint64_t n = 0;
        for (uint64_t b = byte_offsets[i]; b < byte_offsets[i + 1]; b++) n += (utf8[b] & 0xC0) != 0x80;
        if (n == 0) return VO_INVALID_ARGUMENT;
        acc += n - 1;
    }
    out_offsets[S] = acc;
    return VO_OK;
}

typedef struct {
    const vo_predictor *p; const uint8_t *utf8; const uint64_t *boff, *ooff;
    size_t lo, hi; int32_t *scores; uint8_t *labels; uint64_t abytes; int status;
    int cpu;   /* >= 0: the worker pins itself to this CPU (timing runs: no migration, first-touch locality) */
    int use_da; /* the char scorer's automaton as a double array (baseline leg) */
} job_t;
static void *job_run(void *arg) {
    job_t *j = (job_t *)arg;
    if (j->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(j->cpu, &set); (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    scratch_t s; memset(&s, 0, sizeof(s));
    for (size_t i = j->lo; i < j->hi; i++) {
        long n = predict_one_ex(j->p, j->utf8 + j->boff[i], (size_t)(j->boff[i + 1] - j->boff[i]), &s,
                                j->scores ? j->scores + j->ooff[i] : NULL, j->labels ? j->labels + j->ooff[i] : NULL, &j->abytes, j->use_da);
        if (n < 0) { j->status = (int)-n; break; }
        if ((uint64_t)(n - 1) != j->ooff[i + 1] - j->ooff[i]) { j->status = VO_INVALID_ARGUMENT; break; }
    }
    scratch_free(&s);
    return NULL;
}

/* The reference's `for line in stdin { update_raw; predict }` loop (predict/src/main.rs:126-148) over a batch,
 * optionally on `nthreads` host threads over contiguous sentence shards (the reference itself is serial).
 * `char_bytes_out` (may be NULL) receives A_char of BASELINE.md section 4: the sum over every un-merged
 * char n-gram / dict word occurrence of 4*len(w). */
/* flags: bit 0 = pin worker t to the t-th CPU this process may run on (timed baseline runs); bit 1 (VO_FLAG_DOUBLE_ARRAY) = the char
 * scorer walks its automaton as a double array (made here on first use, before any worker starts; not thread-safe against a concurrent
 * first use from another caller -- bench.py and the tests are single callers) */
int vo_predict_batch_ex(const vo_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets, size_t S,
                        int32_t *scores, uint8_t *labels, const uint64_t *out_offsets, int nthreads, uint64_t *char_bytes_out, int flags) {
    if ((flags & 2) && p->has_char && !p->chr.da) ((vo_predictor *)p)->chr.da = da_build(&p->chr.pma);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > S) nthreads = S ? (int)S : 1;
    job_t *jobs = (job_t *)xcalloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)xcalloc((size_t)nthreads, sizeof(pthread_t));
    int *cpus = NULL; int ncpus = 0;
    if ((flags & 1) && nthreads > 1) {
        cpu_set_t allowed; CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
            cpus = (int *)xcalloc(CPU_SETSIZE, sizeof(int));
            for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpus++] = c;
        }
    }
    /* shards balanced by byte count */
    uint64_t total = S ? byte_offsets[S] - byte_offsets[0] : 0;
    size_t lo = 0;
    for (int t = 0; t < nthreads; t++) {
        size_t hi = lo;
        uint64_t target = byte_offsets[0] + total * (uint64_t)(t + 1) / (uint64_t)nthreads;
        if (t == nthreads - 1) hi = S; else while (hi < S && byte_offsets[hi + 1] <= target) hi++;
        jobs[t].p = p; jobs[t].utf8 = utf8; jobs[t].boff = byte_offsets; jobs[t].ooff = out_offsets;
        jobs[t].lo = lo; jobs[t].hi = hi; jobs[t].scores = scores; jobs[t].labels = labels;
        jobs[t].cpu = (ncpus >= nthreads) ? cpus[t] : -1;
        jobs[t].use_da = (flags & 2) ? 1 : 0;
        lo = hi;
    }
    if (nthreads == 1) job_run(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, job_run, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    int status = VO_OK; uint64_t ab = 0;
    for (int t = 0; t < nthreads; t++) { if (jobs[t].status && !status) status = jobs[t].status; ab += jobs[t].abytes; }
    if (char_bytes_out) *char_bytes_out = ab;
    free(jobs); free(th); free(cpus);
    return status;
}
/* ---- the TIMED baseline (bench.py's cpu_baseline leg; VERDICT r5 item 9: "a CPU baseline that scales").
 * The reference's loop (predict/src/main.rs:126-148) on a POOL of pinned workers that lives for the whole run -- `reps` passes over the batch
 * with a barrier on either side of each, timed by the calling thread between the barriers, so that thread start-up is in no pass -- and with
 * the read-only data every char walks (the double-array automaton, the chars' codes, the weight records and vectors, the type window table)
 * REPLICATED per NUMA node: the first worker of a node copies them (first touch: the pages land on its node), the node's workers read that
 * copy.  (One copy made by the main thread serves both sockets of the box from one node's memory: 256 threads were 10.7 x one.)
 * flags: bit 1 = double array (as vo_predict_batch_ex), bit 2 = replicate per node, bit 3 = the replicas on 2 MB pages (transparent huge
 * pages by madvise, where the host allows them).  seconds_out[reps]; *nodes_out = NUMA nodes used. */
typedef struct {
    const vo_predictor *p; vo_predictor *view;   /* what this worker reads: p, or its node's replica */
    const uint8_t *utf8; const uint64_t *boff, *ooff; size_t lo, hi; int32_t *scores; uint8_t *labels;
    int cpu, node, first_of_node, use_da, reps, status; uint64_t abytes;
    pthread_barrier_t *bar; vo_predictor **replicas; int replicate;
    double *t_start, *t_end;   /* [reps]: this worker's own clock readings around its share of a pass (the caller may not run while the workers do) */
} pool_job;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int node_of_cpu(int cpu) {   /* /sys/devices/system/node/nodeN/cpulist; 0 when the machine does not say */
    for (int n = 0; n < 64; n++) {
        char path[96]; snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", n);
        FILE *f = fopen(path, "r");
        if (!f) { if (n == 0) return 0; break; }
        char buf[4096]; size_t len = fread(buf, 1, sizeof(buf) - 1, f); buf[len] = 0; fclose(f);
        const char *q = buf;
        while (*q) {
            char *e; long a = strtol(q, &e, 10), b = a;
            if (e == q) break;
            if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
            if (cpu >= a && cpu <= b) return n;
            q = (*e == ',') ? e + 1 : e;
            if (*q == '\n') break;
        }
    }
    return 0;
}
static int g_dup_huge = 0;   /* the copies on 2 MB pages (madvise: a random walk over 200 MB of tables otherwise misses the TLB at every step) */
static void *dup_bytes(const void *src, size_t n) {
    if (!src || !n) return NULL;
    void *d = NULL;
    if (g_dup_huge && n >= ((size_t)1 << 21)) {
        const size_t al = (size_t)1 << 21, len = (n + al - 1) & ~(al - 1);
        if (posix_memalign(&d, al, len) != 0) d = NULL;
        else (void)madvise(d, len, MADV_HUGEPAGE);
    }
    if (!d) d = xmalloc(n);
    memcpy(d, src, n);
    return d;
}
static vo_predictor *replicate_hot(const vo_predictor *p) {   /* a view of p whose hot read-only arrays are fresh copies (made by the calling thread) */
    vo_predictor *v = (vo_predictor *)xmalloc(sizeof(*v));
    *v = *p;
    if (p->has_char) {
        v->chr.pw = (pw_rec *)dup_bytes(p->chr.pw, sizeof(pw_rec) * p->chr.n_pat);
        v->chr.wdata = (int32_t *)dup_bytes(p->chr.wdata, sizeof(int32_t) * p->chr.n_wdata);
        if (p->chr.da) {
            da_t *d = (da_t *)xmalloc(sizeof(*d));
            *d = *p->chr.da;
            d->st = (da_state *)dup_bytes(p->chr.da->st, sizeof(da_state) * p->chr.da->n);
            d->code_of = (uint32_t *)dup_bytes(p->chr.da->code_of, sizeof(uint32_t) * p->chr.da->n_codes_map);
            v->chr.da = d;
        }
    }
    if (p->type_kind == 1 && p->tcache.scores) v->tcache.scores = (int32_t *)dup_bytes(p->tcache.scores, sizeof(int32_t) * (size_t)(p->tcache.mask + 1));
    return v;
}
static void replica_free(const vo_predictor *p, vo_predictor *v) {
    if (!v) return;
    if (p->has_char) { free(v->chr.pw); free(v->chr.wdata); if (p->chr.da) { free(v->chr.da->st); free(v->chr.da->code_of); free(v->chr.da); } }
    if (p->type_kind == 1 && p->tcache.scores) free(v->tcache.scores);
    free(v);
}
static void *pool_run(void *arg) {
    pool_job *j = (pool_job *)arg;
    if (j->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(j->cpu, &set); (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    if (j->replicate && j->first_of_node) j->replicas[j->node] = replicate_hot(j->p);
    pthread_barrier_wait(j->bar);                                   /* every worker pinned, every node's copy made */
    const vo_predictor *view = (j->replicate && j->replicas[j->node]) ? j->replicas[j->node] : j->p;
    scratch_t s; memset(&s, 0, sizeof(s));
    for (int r = 0; r < j->reps; r++) {
        pthread_barrier_wait(j->bar);                               /* the pass starts */
        j->t_start[r] = now_s();
        uint64_t ab = 0;
        for (size_t i = j->lo; i < j->hi && !j->status; i++) {
            long n = predict_one_ex(view, j->utf8 + j->boff[i], (size_t)(j->boff[i + 1] - j->boff[i]), &s,
                                    j->scores ? j->scores + j->ooff[i] : NULL, j->labels ? j->labels + j->ooff[i] : NULL, &ab, j->use_da);
            if (n < 0) j->status = (int)-n;
            else if ((uint64_t)(n - 1) != j->ooff[i + 1] - j->ooff[i]) j->status = VO_INVALID_ARGUMENT;
        }
        j->abytes = ab;
        j->t_end[r] = now_s();
        pthread_barrier_wait(j->bar);                               /* the pass is over */
    }
    scratch_free(&s);
    return NULL;
}
int vo_baseline_timed(const vo_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets, size_t S, int32_t *scores, uint8_t *labels,
                      const uint64_t *out_offsets, int nthreads, int flags, int reps, double *seconds_out, uint64_t *char_bytes_out, int *nodes_out) {
    if ((flags & 2) && p->has_char && !p->chr.da) ((vo_predictor *)p)->chr.da = da_build(&p->chr.pma);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > S) nthreads = S ? (int)S : 1;
    if (reps < 1) reps = 1;
    g_dup_huge = (flags & 8) ? 1 : 0;
    int *cpus = (int *)xcalloc(CPU_SETSIZE, sizeof(int)); int ncpus = 0;
    cpu_set_t allowed; CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpus++] = c;
    pool_job *jobs = (pool_job *)xcalloc((size_t)nthreads, sizeof(pool_job));
    pthread_t *th = (pthread_t *)xcalloc((size_t)nthreads, sizeof(pthread_t));
    vo_predictor *replicas[64]; memset(replicas, 0, sizeof(replicas));
    int seen[64]; memset(seen, 0, sizeof(seen));
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
    uint64_t total = S ? byte_offsets[S] - byte_offsets[0] : 0;
    size_t lo = 0; int n_nodes = 0;
    for (int t = 0; t < nthreads; t++) {
        size_t hi = lo;
        uint64_t target = byte_offsets[0] + total * (uint64_t)(t + 1) / (uint64_t)nthreads;
        if (t == nthreads - 1) hi = S; else while (hi < S && byte_offsets[hi + 1] <= target) hi++;
        pool_job *j = &jobs[t];
        j->p = p; j->utf8 = utf8; j->boff = byte_offsets; j->ooff = out_offsets; j->lo = lo; j->hi = hi; j->scores = scores; j->labels = labels;
        j->cpu = (ncpus >= nthreads && nthreads > 1) ? cpus[t] : -1;   /* (one worker: wherever the scheduler puts it, like the reference's one thread) */
        j->node = j->cpu >= 0 ? node_of_cpu(j->cpu) & 63 : 0;
        j->first_of_node = !seen[j->node]; if (!seen[j->node]) { seen[j->node] = 1; n_nodes++; }
        j->use_da = (flags & 2) ? 1 : 0; j->reps = reps; j->bar = &bar; j->replicas = replicas; j->replicate = (flags & 4) ? 1 : 0;
        j->t_start = (double *)xcalloc((size_t)reps, sizeof(double)); j->t_end = (double *)xcalloc((size_t)reps, sizeof(double));
        lo = hi;
    }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, pool_run, &jobs[t]);
    pthread_barrier_wait(&bar);
    for (int r = 0; r < reps; r++) { pthread_barrier_wait(&bar); pthread_barrier_wait(&bar); }
    int status = VO_OK; uint64_t ab = 0;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); if (jobs[t].status && !status) status = jobs[t].status; ab += jobs[t].abytes; }
    for (int r = 0; r < reps && seconds_out; r++) {   /* a pass: from the first worker's start to the last one's end */
        double a = jobs[0].t_start[r], b = jobs[0].t_end[r];
        for (int t = 1; t < nthreads; t++) { if (jobs[t].t_start[r] < a) a = jobs[t].t_start[r]; if (jobs[t].t_end[r] > b) b = jobs[t].t_end[r]; }
        seconds_out[r] = b - a;
    }
    for (int t = 0; t < nthreads; t++) { free(jobs[t].t_start); free(jobs[t].t_end); }
    for (int n = 0; n < 64; n++) replica_free(p, replicas[n]);
    pthread_barrier_destroy(&bar);
    if (char_bytes_out) *char_bytes_out = ab;
    if (nodes_out) *nodes_out = n_nodes;
    free(jobs); free(th); free(cpus);
    return status;
}
int vo_predict_batch(const vo_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets, size_t S,
                     int32_t *scores, uint8_t *labels, const uint64_t *out_offsets, int nthreads, uint64_t *char_bytes_out) {
    return vo_predict_batch_ex(p, utf8, byte_offsets, S, scores, labels, out_offsets, nthreads, char_bytes_out, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* tag prediction (predictor.rs:546-637)                                                      */
/* ------------------------------------------------------------------------------------------ */
static const tagw_rec *tagw_find(const scorer_t *sc, uint32_t token_id, uint8_t rel, uint32_t pattern) {
    long lo = 0, hi = (long)sc->n_tagw - 1;
    tagw_rec k; k.token_id = token_id; k.rel = rel; k.pattern = pattern;
    while (lo <= hi) {
        long mid = (lo + hi) / 2;
        int c = tagw_cmp(&sc->tagw[mid], &k);
        if (c == 0) return &sc->tagw[mid];
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}
/* Char/TypeScorerBoundaryTag::add_tag_scores: states[pos..] zipped with tag_weight[token] (window+1 maps) */
static void add_tag_scores(const scorer_t *sc, uint32_t token_id, long pos, const uint32_t *states, long n, int32_t *z, uint32_t zlen) {
    for (uint32_t r = 0; r <= sc->window && pos + r < n; r++) {
        uint32_t st = states[pos + r];
        if (st == NONE_ID) continue;
        const tagw_rec *t = tagw_find(sc, token_id, (uint8_t)r, st);
        if (!t) continue;
        for (uint32_t i = 0; i < t->len && i < zlen; i++) z[i] = (int32_t)((uint32_t)z[i] + (uint32_t)sc->wdata[t->woff + i]);
    }
}

/* Predictor::predict_tags (predictor.rs:546-637) on a sentence whose pattern states `s` holds (predict_one ran): tags for
 * the boundary labels `lab` (n - 1 of them; 0/1/2 = Unknown).  tags_out: n * nt candidate indices or -1 (None).  With
 * store_tag_scores (predictor.rs:510-514): scores_out (n * stride, optional) receives at the LAST char of every token that
 * has a tag model the score vector the reference keeps in sentence.tag_scores[i] (predictor.rs:599-601,632-634), entries
 * [0, bias.len()); models_out (n, optional) the index of that tag model in Model::tag_models order, -1 elsewhere. */
static void fill_tags_one(const vo_predictor *p, const scratch_t *s, long n, const uint8_t *lab, int32_t *tags_out,
                          int32_t *scores_out, uint32_t stride, int32_t *models_out, int32_t **zbuf, uint32_t *zcap) {
    const model_t *m = &p->model;
    const uint32_t nt = p->n_tags;
    for (long i = 0; i < n * (long)nt; i++) tags_out[i] = -1;
    if (models_out) for (long i = 0; i < n; i++) models_out[i] = -1;
    if (nt == 0) return;
    long start = 0; int have_start = 1;
    for (long i = 0; i < n; i++) {
        int b = i < n - 1 ? lab[i] : 1;
        if (b == 2) { have_start = 0; continue; }
        if (b != 1) continue;
        if (have_start) {
            long tm = tok_find(p, s->cps + start, i + 1 - start);   /* tag_predictor.get(token) */
            if (tm >= 0) {
                const tag_model_rec *t = &m->tag[tm];
                uint32_t zlen = t->bias.len;
                if (zlen + 1 > *zcap) { *zcap = 2 * (zlen + 1); *zbuf = (int32_t *)xrealloc(*zbuf, sizeof(int32_t) * *zcap); }
                int32_t *z = *zbuf;
                memcpy(z, t->bias.w, sizeof(int32_t) * zlen);
                if (p->has_char) add_tag_scores(&p->chr, (uint32_t)tm, i, s->cstates, n, z, zlen);
                if (p->type_kind == 2) add_tag_scores(&p->typ, (uint32_t)tm, i, s->tstates, n, z, zlen);
                uint32_t off = 0; /* TagPredictor::predict, predictor.rs:286-304 */
                for (uint32_t j = 0; j < t->n_slots && j < nt; j++) {
                    uint32_t nc = t->n_cands[j];
                    if (nc >= 2) {
                        uint32_t idx = 0; int32_t best = INT32_MIN;
                        for (uint32_t c = 0; c < nc && off + c < zlen; c++) if (z[off + c] > best) { idx = c; best = z[off + c]; }
                        tags_out[i * nt + j] = (int32_t)idx; off += nc;
                    } else tags_out[i * nt + j] = nc == 1 ? 0 : -1;
                }
                if (scores_out) for (uint32_t k = 0; k < zlen && k < stride; k++) scores_out[i * (long)stride + k] = z[k];
                if (models_out) models_out[i] = (int32_t)tm;
            }
        }
        start = i + 1; have_start = 1;
    }
}

/* Scores boundaries, then fills tags for the given boundary labels (0/1/2 = Unknown) exactly like
 * `predict` + (caller edits boundaries) + `fill_tags`.  `labels_in` may be NULL: the predicted ones are used.
 * tags_out: n * n_tags int32 entries, candidate index per slot or -1 (None). */
int vo_predict_tags(const vo_predictor *p, const uint8_t *utf8, size_t len, const uint8_t *labels_in,
                    int32_t *tags_out, uint32_t *n_tags_out) {
    if (!p->predict_tags) return VO_INVALID_ARGUMENT; /* "this predictor is created with predict_tags = false" */
    scratch_t s; memset(&s, 0, sizeof(s));
    scratch_reserve(&s, (long)len + 1);
    uint8_t *lab = (uint8_t *)xmalloc(len + 1);
    long n = predict_one(p, utf8, len, &s, NULL, lab, NULL);
    if (n < 0) { free(lab); scratch_free(&s); return (int)-n; }
    if (labels_in) memcpy(lab, labels_in, (size_t)(n - 1));
    if (n_tags_out) *n_tags_out = p->n_tags;
    int32_t *z = NULL; uint32_t zcap = 0;
    fill_tags_one(p, &s, n, lab, tags_out, NULL, 0, NULL, &z, &zcap);
    free(z); free(lab); scratch_free(&s);
    return VO_OK;
}

uint32_t vo_n_tags(const vo_predictor *p) { return p->n_tags; }
uint32_t vo_tag_score_stride(const vo_predictor *p) { return p->max_zlen; }   /* the longest TagPredictor bias */

/* Sentence::fill_tags over a batch (the labels are the caller's: `predict`, possibly post-filtered, then `fill_tags`,
 * predict/src/main.rs:130-170), on `nthreads` host threads over contiguous sentence shards.  Layouts as the product's:
 * char c of sentence i is row out_offsets[i] + i + c of tags_out [rows * n_tags], scores_out [rows * stride] (optional),
 * models_out [rows] (optional). */
typedef struct {
    const vo_predictor *p; const uint8_t *utf8; const uint64_t *boff, *ooff; const uint8_t *labels;
    size_t lo, hi; int32_t *tags, *scores, *models; uint32_t stride; int status;
} tjob_t;
static void *tjob_run(void *arg) {
    tjob_t *j = (tjob_t *)arg;
    scratch_t s; memset(&s, 0, sizeof(s));
    int32_t *z = NULL; uint32_t zcap = 0;
    const uint32_t nt = j->p->n_tags;
    for (size_t i = j->lo; i < j->hi; i++) {
        long n = predict_one(j->p, j->utf8 + j->boff[i], (size_t)(j->boff[i + 1] - j->boff[i]), &s, NULL, NULL, NULL);
        if (n < 0) { j->status = (int)-n; break; }
        if ((uint64_t)(n - 1) != j->ooff[i + 1] - j->ooff[i]) { j->status = VO_INVALID_ARGUMENT; break; }
        const uint64_t row = j->ooff[i] + i;
        fill_tags_one(j->p, &s, n, j->labels + j->ooff[i], j->tags + row * nt, j->scores ? j->scores + row * j->stride : NULL, j->stride,
                      j->models ? j->models + row : NULL, &z, &zcap);
    }
    free(z); scratch_free(&s);
    return NULL;
}
int vo_fill_tags_batch(const vo_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets, size_t S, const uint64_t *out_offsets,
                       const uint8_t *labels, int32_t *tags_out, int32_t *scores_out, uint32_t stride, int32_t *models_out, int nthreads) {
    if (!p->predict_tags) return VO_INVALID_ARGUMENT;
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > S) nthreads = S ? (int)S : 1;
    tjob_t *jobs = (tjob_t *)xcalloc((size_t)nthreads, sizeof(tjob_t));
    pthread_t *th = (pthread_t *)xcalloc((size_t)nthreads, sizeof(pthread_t));
    uint64_t total = S ? byte_offsets[S] - byte_offsets[0] : 0;
    size_t lo = 0;
    for (int t = 0; t < nthreads; t++) {
        size_t hi = lo;
        uint64_t target = byte_offsets[0] + total * (uint64_t)(t + 1) / (uint64_t)nthreads;
        if (t == nthreads - 1) hi = S; else while (hi < S && byte_offsets[hi + 1] <= target) hi++;
        tjob_t *j = &jobs[t];
        j->p = p; j->utf8 = utf8; j->boff = byte_offsets; j->ooff = out_offsets; j->labels = labels; j->lo = lo; j->hi = hi;
        j->tags = tags_out; j->scores = scores_out; j->models = models_out; j->stride = stride;
        lo = hi;
    }
    if (nthreads == 1) tjob_run(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, tjob_run, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    int status = VO_OK;
    for (int t = 0; t < nthreads; t++) if (jobs[t].status && !status) status = jobs[t].status;
    free(jobs); free(th);
    return status;
}

/* Sentence::write_tokenized_text (sentence.rs:850-886) for sentence i of a batch: tokens = the runs between WordBoundary
 * labels, those adjacent to an Unknown label skipped (TokenIterator, sentence.rs:1265-1309); ' ' between tokens; '\\' in
 * front of ' ', '\\', '/' in surfaces and tags; "/tag" for the token's slots up to the last Some, empty for a None in
 * between.  tags (may be NULL) / models: rows as vo_fill_tags_batch wrote them.  dst NULL: only the size. */
static uint64_t put_escaped(uint8_t *dst, uint64_t at, const uint8_t *src, size_t n) {
    for (size_t k = 0; k < n; k++) {
        uint8_t b = src[k];
        if (b == ' ' || b == '\\' || b == '/') { if (dst) dst[at] = '\\'; at++; }
        if (dst) dst[at] = b;
        at++;
    }
    return at;
}
static uint64_t write_one(const vo_predictor *p, const uint8_t *text, size_t len, long n, const uint8_t *lab, const int32_t *tags,
                          const int32_t *models, uint8_t *dst) {
    const uint32_t nt = tags ? p->n_tags : 0;
    uint64_t at = 0;
    size_t tok_byte = 0, pos = 0;          /* byte where the open token starts; byte of char c */
    int have_start = 1, first = 1;
    for (long c = 0; c < n; c++) {
        size_t nxt = pos + 1;               /* byte after char c */
        while (nxt < len && (text[nxt] & 0xC0) == 0x80) nxt++;
        int b = c < n - 1 ? lab[c] : 1;
        if (b == 2) have_start = 0;
        else if (b == 1) {
            if (have_start) {
                if (!first) { if (dst) dst[at] = ' '; at++; }
                first = 0;
                at = put_escaped(dst, at, text + tok_byte, nxt - tok_byte);
                if (nt && models && models[c] >= 0) {
                    const tag_model_rec *t = &p->model.tag[models[c]];
                    long last = -1;
                    for (uint32_t j = 0; j < nt; j++) if (tags[c * (long)nt + j] >= 0) last = j;
                    for (long j = 0; j <= last; j++) {
                        if (dst) dst[at] = '/';
                        at++;
                        int32_t idx = tags[c * (long)nt + j];
                        if (idx >= 0 && (uint32_t)j < t->n_slots && (uint32_t)idx < t->n_cands[j]) {
                            uint32_t k = t->cand_first[j] + (uint32_t)idx;
                            at = put_escaped(dst, at, t->cand_str[k], t->cand_len[k]);
                        }
                    }
                }
            }
            tok_byte = nxt; have_start = 1;
        }
        pos = nxt;
    }
    return at;
}
typedef struct {
    const vo_predictor *p; const uint8_t *utf8; const uint64_t *boff, *ooff; const uint8_t *labels; const int32_t *tags, *models;
    size_t lo, hi; uint8_t *out; uint64_t *toff; int pass;
} wjob_t;
static void *wjob_run(void *arg) {
    wjob_t *j = (wjob_t *)arg;
    const uint32_t nt = j->p->n_tags;
    for (size_t i = j->lo; i < j->hi; i++) {
        const uint64_t row = j->ooff[i] + i;
        const long n = (long)(j->ooff[i + 1] - j->ooff[i]) + 1;
        const uint64_t sz = write_one(j->p, j->utf8 + j->boff[i], (size_t)(j->boff[i + 1] - j->boff[i]), n, j->labels + j->ooff[i],
                                      j->tags ? j->tags + row * nt : NULL, j->models ? j->models + row : NULL, j->pass ? j->out + j->toff[i] : NULL);
        if (!j->pass) j->toff[i + 1] = sz;
    }
    return NULL;
}
/* text_offsets_out [S + 1]; returns VO_INVALID_ARGUMENT when `cap` is too small (text_offsets_out is complete even then) */
int vo_write_tokenized_batch(const vo_predictor *p, const uint8_t *utf8, const uint64_t *byte_offsets, size_t S, const uint64_t *out_offsets,
                             const uint8_t *labels, const int32_t *tags, const int32_t *models, uint8_t *out, uint64_t cap,
                             uint64_t *text_offsets_out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > S) nthreads = S ? (int)S : 1;
    wjob_t *jobs = (wjob_t *)xcalloc((size_t)nthreads, sizeof(wjob_t));
    pthread_t *th = (pthread_t *)xcalloc((size_t)nthreads, sizeof(pthread_t));
    text_offsets_out[0] = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int t = 0; t < nthreads; t++) {
            wjob_t *j = &jobs[t];
            j->p = p; j->utf8 = utf8; j->boff = byte_offsets; j->ooff = out_offsets; j->labels = labels; j->tags = tags; j->models = models;
            j->lo = S * (size_t)t / (size_t)nthreads; j->hi = S * (size_t)(t + 1) / (size_t)nthreads; j->out = out; j->toff = text_offsets_out; j->pass = pass;
        }
        if (nthreads == 1) wjob_run(&jobs[0]);
        else {
            for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, wjob_run, &jobs[t]);
            for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        }
        if (pass == 0) {
            for (size_t i = 0; i < S; i++) text_offsets_out[i + 1] += text_offsets_out[i];
            if (text_offsets_out[S] > cap) { free(jobs); free(th); return VO_INVALID_ARGUMENT; }
        }
    }
    free(jobs); free(th);
    return VO_OK;
}

/* scorer-level probe for the KATs of char_scorer.rs:503-525 / type_scorer.rs:453-473:
 * z (zlen entries, pre-filled by the caller) += tag scores of (token_id, pos) from one scorer. */
int vo_tag_scores_probe(const vo_predictor *p, const uint8_t *utf8, size_t len, int which /*0 char, 1 type*/,
                        uint32_t token_id, uint32_t pos, int32_t *z, uint32_t zlen) {
    scratch_t s; memset(&s, 0, sizeof(s));
    long n = predict_one(p, utf8, len, &s, NULL, NULL, NULL);
    if (n < 0) { scratch_free(&s); return (int)-n; }
    if (which == 0 && p->has_char && p->chr.record_states) add_tag_scores(&p->chr, token_id, pos, s.cstates, n, z, zlen);
    if (which == 1 && p->type_kind == 2 && p->typ.record_states) add_tag_scores(&p->typ, token_id, pos, s.tstates, n, z, zlen);
    scratch_free(&s);
    return VO_OK;
}

/* test hook for the PositionalWeight += KATs (predictor.rs:677-747) */
int vo_test_posw_add_assign(int32_t off_a, const int32_t *wa, uint32_t la, int32_t off_b, const int32_t *wb, uint32_t lb,
                            int32_t *off_out, int32_t *w_out, uint32_t *len_out, uint32_t cap) {
    posw a, b;
    a.offset = off_a; a.len = la; a.w = (int32_t *)xmalloc(sizeof(int32_t) * la); memcpy(a.w, wa, sizeof(int32_t) * la);
    b.offset = off_b; b.len = lb; b.w = (int32_t *)wb;
    posw_add_assign(&a, &b);
    *off_out = a.offset; *len_out = a.len;
    for (uint32_t i = 0; i < a.len && i < cap; i++) w_out[i] = a.w[i];
    free(a.w);
    return VO_OK;
}

/* merged pattern table of the char scorer, for the merger KAT (char_scorer.rs:169-185):
 * returns number of patterns; fills offset/len of pattern `idx` and copies its weights. */
uint32_t vo_char_pattern_count(const vo_predictor *p) { return p->has_char ? p->chr.n_pat : 0; }
int vo_char_pattern_get(const vo_predictor *p, uint32_t idx, int32_t *offset, uint32_t *len, int32_t *w, uint32_t wcap) {
    if (!p->has_char || idx >= p->chr.n_pat) return VO_INVALID_ARGUMENT;
    const pw_rec *r = &p->chr.pw[idx];
    *offset = r->offset; *len = r->len;
    for (uint32_t i = 0; i < r->len && i < wcap; i++) w[i] = p->chr.wdata[r->woff + i];
    return VO_OK;
}
