// Synthetic models and sentences (bench / test tooling; NOT on the product path).
//
// The real bccwj-suw+unidic and jp-0.4.7-5 model files are not available offline, so the benchmark follows
// models of the documented shape (W = 3, n <= 3, KyTea-style dictionary) written in
// the real Vaporetto model format, and sentences whose pattern hit rates resemble real text.  Deterministic:
// splitmix64 streams keyed by the seed.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../csrc/model.hpp"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return uint32_t(uniform() * n); }
    // Zipf(s = 1) rank in [0, n): continuous inverse-CDF approximation r = (n+1)^u
    uint32_t zipf(uint32_t n) {
        uint32_t r = uint32_t(std::exp(uniform() * std::log(double(n) + 1.0))) - 1;
        return r < n ? r : n - 1;
    }
    int32_t weight(double p_nonzero) {
        if (uniform() >= p_nonzero) return 0;
        return int32_t(below(65535)) - 32767;
    }
};

struct Writer {  // bincode-2 standard encoding
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void uvar(uint64_t v) {
        if (v < 251) b.push_back(uint8_t(v));
        else if (v < (1u << 16)) { b.push_back(251); for (int i = 0; i < 2; ++i) b.push_back(uint8_t(v >> (8 * i))); }
        else if (v < (1ull << 32)) { b.push_back(252); for (int i = 0; i < 4; ++i) b.push_back(uint8_t(v >> (8 * i))); }
        else { b.push_back(253); for (int i = 0; i < 8; ++i) b.push_back(uint8_t(v >> (8 * i))); }
    }
    void i32(int32_t v) { uvar((uint32_t(v) << 1) ^ uint32_t(v >> 31)); }
    void utf8(const std::vector<uint32_t>& s) {
        std::string o;
        for (uint32_t c : s) {
            if (c < 0x80) o.push_back(char(c));
            else if (c < 0x800) { o.push_back(char(0xC0 | (c >> 6))); o.push_back(char(0x80 | (c & 0x3F))); }
            else if (c < 0x10000) { o.push_back(char(0xE0 | (c >> 12))); o.push_back(char(0x80 | ((c >> 6) & 0x3F))); o.push_back(char(0x80 | (c & 0x3F))); }
            else { o.push_back(char(0xF0 | (c >> 18))); o.push_back(char(0x80 | ((c >> 12) & 0x3F))); o.push_back(char(0x80 | ((c >> 6) & 0x3F))); o.push_back(char(0x80 | (c & 0x3F))); }
        }
        uvar(o.size());
        b.insert(b.end(), o.begin(), o.end());
    }
    void str(const std::string& s) { uvar(s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void bytes(const std::vector<uint8_t>& s) { uvar(s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void weights(const std::vector<int32_t>& w) { uvar(w.size()); for (int32_t x : w) i32(x); }
};

// 4000-character vocabulary, most frequent first: kana, punctuation, full-width digits/roman, kanji
std::vector<uint32_t> make_vocab(uint32_t size) {
    std::vector<uint32_t> v;
    for (uint32_t c = 0x3041; c <= 0x3093; ++c) v.push_back(c);  // hiragana
    for (uint32_t c : {0x3001u, 0x3002u, 0x300Cu, 0x300Du, 0x30FBu, 0x30FCu}) v.push_back(c);
    for (uint32_t c = 0x30A1; c <= 0x30F6; ++c) v.push_back(c);  // katakana
    for (uint32_t c = 0xFF10; c <= 0xFF19; ++c) v.push_back(c);
    for (uint32_t c = 0xFF21; c <= 0xFF3A; ++c) v.push_back(c);
    for (uint32_t c = 0xFF41; c <= 0xFF5A; ++c) v.push_back(c);
    for (uint32_t c = 0x4E00; v.size() < size; ++c) v.push_back(c);
    v.resize(size);
    return v;
}

// 40 kanji of CJK Extension B (kind 6): the three UniDic is known for, the others spread over the block
std::vector<uint32_t> nonbmp_chars() {
    std::vector<uint32_t> v = {0x20B9Fu, 0x20BB7u, 0x29E3Du};
    for (uint32_t k = 0; v.size() < 40; ++k) v.push_back(0x20000u + 1031u * k + 7u);
    return v;
}

std::vector<int32_t> rand_weights(Rng& r, size_t n, double p) {
    std::vector<int32_t> w(n);
    for (auto& x : w) x = r.weight(p);
    return w;
}

uint32_t dict_len(Rng& r, int kind) {
    double u = r.uniform();
    if (kind == 2) {  // jp-0.4.7-5-like: longer words, mean ~3.5, max 16
        static const double cdf[16] = {0.02, 0.24, 0.50, 0.72, 0.84, 0.90, 0.935, 0.955, 0.97, 0.98, 0.986, 0.991, 0.994, 0.996, 0.998, 1.0};
        for (uint32_t i = 0; i < 16; ++i) if (u < cdf[i]) return i + 1;
        return 16;
    }
    // bccwj-suw+unidic-like: {1:3, 2:35, 3:27, 4:20, 5:8, 6:4, 7-12:3} %
    if (u < 0.03) return 1;
    if (u < 0.38) return 2;
    if (u < 0.65) return 3;
    if (u < 0.85) return 4;
    if (u < 0.93) return 5;
    if (u < 0.97) return 6;
    return 7 + r.below(6);
}

}  // namespace

extern "C" {

void vpt_synth_free(void* p) { std::free(p); }

// kind: 1 = M1 (bccwj-suw+unidic-like), 2 = M2 (jp-0.4.7-5-like), 3 = M3 (M1 + tag models), 4 / 5 = M1 trained with
// --charw 2 --typew 2 / --charw 4 --typew 4 (train/src/main.rs:33-51: the windows are free parameters), 6 = M1 + 100 char n-grams and
// 100 dictionary words that hold kanji OUTSIDE the BMP (UniDic has such entries: U+20B9F, U+20BB7, U+29E3D ..).  scale multiplies
// every count (1.0 = full size).  Returns 0 and a malloc'ed model file.
int vpt_synth_model_ex(int kind, uint64_t seed, double scale, uint32_t vocab, double dup_share, uint8_t** out, size_t* out_len);
int vpt_synth_model(int kind, uint64_t seed, double scale, uint8_t** out, size_t* out_len) {
    return vpt_synth_model_ex(kind, seed, scale, 0, 0.0, out, out_len);
}
// vocab (0: 4000 at full scale): size of the alphabet -- a unidic-sized model has about twice M1's; dup_share: that share of the
// dictionary words are copies of the model's own 2- and 3-char n-grams, whose merged rows then overflow the packed tables' 16-bit
// fields (the kernel's wide-row path); the model-shape sweep of tools/model_sweep.py
int vpt_synth_model_ex(int kind, uint64_t seed, double scale, uint32_t vocab_size, double dup_share, uint8_t** out, size_t* out_len) {
    if (kind < 1 || kind > 6 || !(scale > 0) || !out || !out_len || vocab_size > 20000 || dup_share < 0 || dup_share > 1) return 2;
    Rng r(seed);
    const uint32_t V = vocab_size ? std::max<uint32_t>(300, vocab_size) : std::max<uint32_t>(300, uint32_t(4000 * std::min(1.0, std::sqrt(scale))));
    const std::vector<uint32_t> vocab = make_vocab(V);
    const size_t n_bi = size_t(400000 * scale), n_tri = size_t(600000 * scale);
    const size_t n_dict = size_t((kind == 2 ? 1000000 : 700000) * scale);
    const size_t n_tag = kind == 3 ? size_t(50000 * scale) : 0;
    const double p_nz = 0.35;
    const int W = kind == 4 ? 2 : kind == 5 ? 4 : 3;

    Writer w;
    const char magic[] = "VaporettoTokenizer 0.5.0\n";
    w.b.insert(w.b.end(), magic, magic + sizeof(magic) - 1);

    // ---- char n-grams: all unigrams, Zipf-composed bigrams and trigrams
    std::unordered_set<uint64_t> seen;
    std::vector<std::vector<uint32_t>> grams;
    for (uint32_t i = 0; i < V; ++i) grams.push_back({vocab[i]});
    auto draw = [&](size_t count, int n) {
        size_t made = 0, tries = 0;
        while (made < count && tries < count * 50) {
            ++tries;
            uint64_t key = uint64_t(n);
            std::vector<uint32_t> g(static_cast<size_t>(n));
            for (int k = 0; k < n; ++k) { uint32_t id = r.zipf(V); g[size_t(k)] = vocab[id]; key = key * 4099 + id; }
            if (!seen.insert(key).second) continue;
            grams.push_back(std::move(g));
            ++made;
        }
    };
    draw(n_bi, 2);
    draw(n_tri, 3);
    // (kind 6) the patterns with chars outside the BMP come from a stream of their own: everything else is M1, value for value
    Rng rx(seed ^ 0x6E6F6E2D424D50ull);
    const std::vector<uint32_t> outside = nonbmp_chars();
    auto nonbmp_string = [&](uint32_t n) {
        std::vector<uint32_t> g(n);
        for (uint32_t k = 0; k < n; ++k) g[k] = vocab[rx.zipf(V)];
        g[rx.below(n)] = outside[rx.below(uint32_t(outside.size()))];
        if (n > 2 && rx.uniform() < 0.3) g[rx.below(n)] = outside[rx.below(uint32_t(outside.size()))];
        return g;
    };
    const size_t n_base_grams = grams.size();
    if (kind == 6) {
        for (uint32_t c : outside) grams.push_back({c});
        std::unordered_set<std::string> have;
        while (grams.size() < n_base_grams + outside.size() + 100) {
            std::vector<uint32_t> g = nonbmp_string(2 + rx.below(2));
            std::string key(reinterpret_cast<const char*>(g.data()), g.size() * 4);
            if (have.insert(key).second) grams.push_back(std::move(g));
        }
    }
    w.uvar(grams.size());
    for (size_t gi = 0; gi < grams.size(); ++gi) {
        const auto& g = grams[gi];
        w.utf8(g);
        w.weights(rand_weights(gi < n_base_grams ? r : rx, size_t(2 * W - int(g.size()) + 1), p_nz));
    }
    // ---- all 258 type n-grams
    std::vector<std::vector<uint8_t>> tgrams;
    for (uint8_t a = 1; a <= 6; ++a) {
        tgrams.push_back({a});
        for (uint8_t b = 1; b <= 6; ++b) {
            tgrams.push_back({a, b});
            for (uint8_t c = 1; c <= 6; ++c) tgrams.push_back({a, b, c});
        }
    }
    w.uvar(tgrams.size());
    for (const auto& g : tgrams) {
        w.bytes(g);
        w.weights(rand_weights(r, size_t(2 * W - int(g.size()) + 1), 0.8));
    }
    // ---- dictionary
    seen.clear();
    std::vector<std::vector<uint32_t>> words;
    {
        size_t tries = 0;
        while (words.size() < n_dict && tries < n_dict * 50) {
            ++tries;
            if (dup_share > 0 && r.uniform() < dup_share && grams.size() > V) {   // a copy of one of the 2- / 3-char n-grams
                const std::vector<uint32_t>& src = grams[V + r.below(uint32_t(grams.size() - V))];
                uint64_t key = 0x9E3779B97F4A7C15ull + src.size();
                for (uint32_t c : src) key = key * 0x100000001B3ull + c;
                if (!seen.insert(key).second) continue;
                words.push_back(src);
                continue;
            }
            uint32_t L = dict_len(r, kind);
            std::vector<uint32_t> g(L);
            uint64_t key = L;
            for (uint32_t k = 0; k < L; ++k) { uint32_t id = r.zipf(V); g[k] = vocab[id]; key = key * 0x100000001B3ull + id + 1; }
            if (!seen.insert(key).second) continue;
            words.push_back(std::move(g));
        }
    }
    const size_t n_base_words = words.size();
    if (kind == 6) {
        std::unordered_set<std::string> have;
        while (words.size() < n_base_words + 100) {
            std::vector<uint32_t> g = nonbmp_string(1 + rx.below(6));
            std::string key(reinterpret_cast<const char*>(g.data()), g.size() * 4);
            if (have.insert(key).second) words.push_back(std::move(g));
        }
    }
    w.uvar(words.size());
    for (size_t wi = 0; wi < words.size(); ++wi) {
        const auto& g = words[wi];
        w.utf8(g);
        w.weights(rand_weights(wi < n_base_words ? r : rx, g.size() + 1, 0.6));
        w.str("");
    }
    w.i32(int32_t(r.below(20001)) - 10000);  // bias
    w.u8(uint8_t(W));
    w.u8(uint8_t(W));
    // ---- tag models (M3): tokens are dictionary words; 2 slots (<=10 POS, <=6 pron candidates)
    const size_t nt = std::min(n_tag, words.size());
    w.uvar(nt);
    for (size_t t = 0; t < nt; ++t) {
        const auto& tok = words[t * (words.size() / std::max<size_t>(nt, 1))];
        w.utf8(tok);
        uint32_t k1 = 1 + r.below(10), k2 = 1 + r.below(6);
        w.uvar(2);
        w.uvar(k1); for (uint32_t i = 0; i < k1; ++i) w.str("P" + std::to_string(i));
        w.uvar(k2); for (uint32_t i = 0; i < k2; ++i) w.str("R" + std::to_string(i));
        const size_t zlen = (k1 >= 2 ? k1 : 0) + (k2 >= 2 ? k2 : 0);
        const uint32_t nc = 8 + r.below(9);
        w.uvar(nc);
        for (uint32_t i = 0; i < nc; ++i) {  // left context + token + rel extra chars
            uint32_t left = r.below(3), rel = r.below(4);
            std::vector<uint32_t> g;
            for (uint32_t k = 0; k < left; ++k) g.push_back(vocab[r.zipf(V)]);
            g.insert(g.end(), tok.begin(), tok.end());
            for (uint32_t k = 0; k < rel; ++k) g.push_back(vocab[r.zipf(V)]);
            w.utf8(g);
            w.uvar(1); w.u8(uint8_t(rel)); w.weights(rand_weights(r, zlen, 0.7));
        }
        const uint32_t ntp = 4 + r.below(5);
        w.uvar(ntp);
        for (uint32_t i = 0; i < ntp; ++i) {
            uint32_t n = 1 + r.below(4), rel = r.below(4);
            std::vector<uint8_t> g(n);
            for (auto& x : g) x = uint8_t(1 + r.below(6));
            w.bytes(g);
            w.uvar(1); w.u8(uint8_t(rel)); w.weights(rand_weights(r, zlen, 0.7));
        }
        w.weights(rand_weights(r, zlen, 0.9));
    }
    *out_len = w.b.size();
    *out = static_cast<uint8_t*>(std::malloc(w.b.size() ? w.b.size() : 1));
    if (!*out) return 3;
    std::memcpy(*out, w.b.data(), w.b.size());
    return 0;
}

}  // extern "C"

// Sentences: items drawn 70 % from the model's own pattern list (Zipf over code-point order, which is the
// BTreeMap order of the reference's merger) and 30 % single characters from alphabet A, cut to a length drawn
// log-uniformly in [min_len, max_len].  Every character is 3-byte UTF-8.
namespace {
struct PatternList {
    vpt::ModelData m;
    std::vector<const vpt::SymString*> pats;
    bool load(const uint8_t* model, size_t model_len) {
        try { m = vpt::parse_model(model, model_len, nullptr); } catch (const vpt::ModelError&) { return false; }
        for (const auto& d : m.char_ngrams) pats.push_back(&d.ngram);
        for (const auto& d : m.dict) pats.push_back(&d.word);
        std::sort(pats.begin(), pats.end(), [](const vpt::SymString* a, const vpt::SymString* b) { return *a < *b; });
        pats.erase(std::unique(pats.begin(), pats.end(), [](const vpt::SymString* a, const vpt::SymString* b) { return *a == *b; }), pats.end());
        return true;
    }
};
// one block of sentences from its own splitmix64 stream: text bytes appended to `text`, byte length of every sentence to `lens`
void gen_block(const PatternList& P, uint64_t seed, size_t n_sent, uint32_t min_len, uint32_t max_len, std::vector<uint8_t>& text,
               std::vector<uint32_t>& lens, double hit_share = 0.7, double nonbmp_share = 0.0) {
    const std::vector<const vpt::SymString*>& pats = P.pats;
    Rng r(seed);
    // nonbmp_share > 0: that share of a sentence's ITEMS is a char outside the BMP -- half of them one of the model's patterns that hold
    // such a char (sorted last, a Zipf draw over the list never reaches them), half any kanji of CJK Extension B
    std::vector<const vpt::SymString*> nb_pats;
    if (nonbmp_share > 0)
        for (const vpt::SymString* p : pats)
            if (std::any_of(p->begin(), p->end(), [](uint32_t c) { return c >= 0x10000u; })) nb_pats.push_back(p);
    auto alphabet_a = [&]() -> uint32_t {
        double u = r.uniform();
        if (u < 0.50) return 0x3041 + r.below(0x3093 - 0x3041 + 1);
        if (u < 0.65) return 0x30A1 + r.below(0x30F6 - 0x30A1 + 1);
        if (u < 0.90) return 0x4E00 + r.below(0x9FA5 - 0x4E00 + 1);
        if (u < 0.94) return 0xFF10 + r.below(10);
        if (u < 0.98) return (r.below(2) ? 0xFF21 : 0xFF41) + r.below(26);
        static const uint32_t punct[6] = {0x3001, 0x3002, 0x300C, 0x300D, 0x30FB, 0x30FC};
        return punct[r.below(6)];
    };
    text.reserve(text.size() + n_sent * size_t(max_len + min_len) / 2 * 3 + 64);
    std::vector<uint32_t> sent;
    for (size_t i = 0; i < n_sent; ++i) {
        const size_t before = text.size();
        uint32_t L = min_len;
        if (max_len > min_len) {
            L = uint32_t(std::exp(std::log(double(min_len)) + r.uniform() * (std::log(double(max_len) + 1.0) - std::log(double(min_len)))));
            L = std::min(std::max(L, min_len), max_len);
        }
        sent.clear();
        while (sent.size() < L) {
            if (nonbmp_share > 0 && r.uniform() < nonbmp_share) {
                if (!nb_pats.empty() && r.below(2)) { const vpt::SymString& p = *nb_pats[r.below(uint32_t(nb_pats.size()))]; sent.insert(sent.end(), p.begin(), p.end()); }
                else sent.push_back(0x20000u + r.below(0xA6E0u));
                continue;
            }
            if (!pats.empty() && r.uniform() < hit_share) {
                const vpt::SymString& p = *pats[r.zipf(uint32_t(pats.size()))];
                sent.insert(sent.end(), p.begin(), p.end());
            } else sent.push_back(alphabet_a());
        }
        sent.resize(L);
        for (uint32_t c : sent) {
            if (c < 0x80) text.push_back(uint8_t(c));
            else if (c < 0x800) { text.push_back(uint8_t(0xC0 | (c >> 6))); text.push_back(uint8_t(0x80 | (c & 0x3F))); }
            else if (c < 0x10000) { text.push_back(uint8_t(0xE0 | (c >> 12))); text.push_back(uint8_t(0x80 | ((c >> 6) & 0x3F))); text.push_back(uint8_t(0x80 | (c & 0x3F))); }
            else { text.push_back(uint8_t(0xF0 | (c >> 18))); text.push_back(uint8_t(0x80 | ((c >> 12) & 0x3F))); text.push_back(uint8_t(0x80 | ((c >> 6) & 0x3F))); text.push_back(uint8_t(0x80 | (c & 0x3F))); }
        }
        lens.push_back(uint32_t(text.size() - before));
    }
}
int emit(const std::vector<std::vector<uint8_t>>& texts, const std::vector<std::vector<uint32_t>>& lens, uint8_t** utf8_out, size_t* nbytes_out,
         uint64_t** boff_out) {
    size_t total = 0, n_sent = 0;
    for (size_t b = 0; b < texts.size(); ++b) { total += texts[b].size(); n_sent += lens[b].size(); }
    uint64_t* boff = static_cast<uint64_t*>(std::malloc(sizeof(uint64_t) * (n_sent + 1)));
    uint8_t* out = static_cast<uint8_t*>(std::malloc(total + 64));
    if (!boff || !out) { std::free(boff); std::free(out); return 3; }
    size_t at = 0, si = 0;
    for (size_t b = 0; b < texts.size(); ++b) {
        std::memcpy(out + at, texts[b].data(), texts[b].size());
        size_t o = at;
        for (uint32_t l : lens[b]) { boff[si++] = o; o += l; }
        at += texts[b].size();
    }
    boff[n_sent] = total;
    std::memset(out + total, 0, 64);
    *utf8_out = out; *nbytes_out = total; *boff_out = boff;
    return 0;
}
}  // namespace

extern "C" {

// hit_share: the share of a sentence's ITEMS drawn from the model's pattern list (0.7); the rest are single chars of
// alphabet A, most of which no pattern continues -- the knob of the text-realism sweep (tools/hit_share_sweep.py)
int vpt_synth_sentences_ex(const uint8_t* model, size_t model_len, uint64_t seed, size_t n_sent, uint32_t min_len,
                           uint32_t max_len, double hit_share, uint8_t** utf8_out, size_t* nbytes_out, uint64_t** boff_out) {
    if (!model || !utf8_out || !nbytes_out || !boff_out || min_len < 1 || max_len < min_len || !(hit_share >= 0 && hit_share <= 1)) return 2;
    PatternList P;
    if (!P.load(model, model_len)) return 1;
    std::vector<std::vector<uint8_t>> texts(1);
    std::vector<std::vector<uint32_t>> lens(1);
    gen_block(P, seed, n_sent, min_len, max_len, texts[0], lens[0], hit_share);
    return emit(texts, lens, utf8_out, nbytes_out, boff_out);
}
// ... and `nonbmp_share` of the items chars outside the BMP (gen_block)
int vpt_synth_sentences_nb(const uint8_t* model, size_t model_len, uint64_t seed, size_t n_sent, uint32_t min_len,
                           uint32_t max_len, double hit_share, double nonbmp_share, uint8_t** utf8_out, size_t* nbytes_out, uint64_t** boff_out) {
    if (!model || !utf8_out || !nbytes_out || !boff_out || min_len < 1 || max_len < min_len || !(hit_share >= 0 && hit_share <= 1) || !(nonbmp_share >= 0 && nonbmp_share <= 1)) return 2;
    PatternList P;
    if (!P.load(model, model_len)) return 1;
    std::vector<std::vector<uint8_t>> texts(1);
    std::vector<std::vector<uint32_t>> lens(1);
    gen_block(P, seed, n_sent, min_len, max_len, texts[0], lens[0], hit_share, nonbmp_share);
    return emit(texts, lens, utf8_out, nbytes_out, boff_out);
}
int vpt_synth_sentences(const uint8_t* model, size_t model_len, uint64_t seed, size_t n_sent, uint32_t min_len,
                        uint32_t max_len, uint8_t** utf8_out, size_t* nbytes_out, uint64_t** boff_out) {
    return vpt_synth_sentences_ex(model, model_len, seed, n_sent, min_len, max_len, 0.7, utf8_out, nbytes_out, boff_out);
}

// A big batch as BLOCKS of `block_sentences` sentences, block j from the stream seeded seed0 + 7919 j: any contiguous run of
// blocks can be generated on its own (a rank's shard of the 10 M-sentence batch), on `nthreads` threads.
int vpt_synth_blocks(const uint8_t* model, size_t model_len, uint64_t seed0, size_t first_block, size_t n_blocks, size_t block_sentences,
                     uint32_t min_len, uint32_t max_len, uint32_t nthreads, uint8_t** utf8_out, size_t* nbytes_out, uint64_t** boff_out) {
    if (!model || !utf8_out || !nbytes_out || !boff_out || min_len < 1 || max_len < min_len) return 2;
    PatternList P;
    if (!P.load(model, model_len)) return 1;
    std::vector<std::vector<uint8_t>> texts(n_blocks);
    std::vector<std::vector<uint32_t>> lens(n_blocks);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t b; (b = next.fetch_add(1)) < n_blocks;)
            gen_block(P, seed0 + 7919ull * (first_block + b), block_sentences, min_len, max_len, texts[b], lens[b]);
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < std::max<uint32_t>(1, nthreads); ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
    return emit(texts, lens, utf8_out, nbytes_out, boff_out);
}

}  // extern "C"
