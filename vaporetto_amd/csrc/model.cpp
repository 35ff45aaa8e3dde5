// Model file decoder (model.rs:127-135).  bincode 2.0.1 "standard" configuration:
//   u8 raw; unsigned ints varint (<251 one byte, 251/252/253 + LE u16/u32/u64); signed ints zig-zag first;
//   Vec/String = varint length + items; structs = fields in order.
#include "model.hpp"

#include <cstring>

namespace vpt {
namespace {

constexpr char kMagic[] = "VaporettoTokenizer 0.5.0\n";  // model.rs:15
constexpr size_t kMagicLen = sizeof(kMagic) - 1;

class Cursor {
public:
    Cursor(const uint8_t* p, size_t n, size_t pos) : p_(p), n_(n), pos_(pos) {}
    size_t pos() const { return pos_; }

    uint8_t u8() {
        need(1);
        return p_[pos_++];
    }
    uint64_t uvar() {
        uint8_t tag = u8();
        if (tag < 251) return tag;
        int nb = tag == 251 ? 2 : tag == 252 ? 4 : tag == 253 ? 8 : 0;
        if (!nb) throw ModelError("DecodeError: unsupported integer width in model data");
        need(nb);
        uint64_t v = 0;
        for (int i = 0; i < nb; ++i) v |= uint64_t(p_[pos_ + i]) << (8 * i);
        pos_ += nb;
        return v;
    }
    int32_t i32() {
        uint64_t u = uvar();
        if (u > 0xFFFFFFFFull) throw ModelError("DecodeError: i32 out of range in model data");
        uint32_t z = uint32_t(u);
        return int32_t((z >> 1) ^ (~(z & 1) + 1));
    }
    size_t length() {
        uint64_t v = uvar();
        if (v > n_ - pos_) throw ModelError("DecodeError: length exceeds the remaining model data");
        return size_t(v);
    }
    const uint8_t* take(size_t n) {
        need(n);
        const uint8_t* q = p_ + pos_;
        pos_ += n;
        return q;
    }
    std::vector<int32_t> weights() {
        size_t n = length();
        std::vector<int32_t> w(n);
        for (auto& x : w) x = i32();
        return w;
    }
    SymString string_syms() {  // String -> scalar values
        size_t n = length();
        SymString s;
        if (!decode_utf8(take(n), n, s)) throw ModelError("DecodeError: invalid UTF-8 in model data");
        return s;
    }
    std::string string_raw() {
        size_t n = length();
        const uint8_t* q = take(n);
        return std::string(reinterpret_cast<const char*>(q), n);
    }
    SymString bytes_syms() {  // Vec<u8> -> one symbol per byte
        size_t n = length();
        const uint8_t* q = take(n);
        return SymString(q, q + n);
    }

private:
    void need(size_t n) const {
        if (n > n_ - pos_) throw ModelError("DecodeError: unexpected end of model data");
    }
    const uint8_t* p_;
    size_t n_, pos_;
};

std::vector<NgramRecord> read_ngrams(Cursor& c, bool is_char) {
    std::vector<NgramRecord> v(c.length());
    for (auto& r : v) {
        r.ngram = is_char ? c.string_syms() : c.bytes_syms();
        r.weights = c.weights();
    }
    return v;
}

std::vector<TagNgramRecord> read_tag_ngrams(Cursor& c, bool is_char) {
    std::vector<TagNgramRecord> v(c.length());
    for (auto& r : v) {
        r.ngram = is_char ? c.string_syms() : c.bytes_syms();
        r.weights.resize(c.length());
        for (auto& w : r.weights) {
            w.rel_position = c.u8();
            w.weights = c.weights();
        }
    }
    return v;
}

}  // namespace

bool decode_utf8(const uint8_t* s, size_t n, SymString& out) {
    out.clear();
    out.reserve(n / 2 + 1);
    size_t i = 0;
    while (i < n) {
        uint32_t b = s[i], cp;
        size_t extra;
        if (b < 0x80) { cp = b; extra = 0; }
        else if ((b & 0xE0) == 0xC0) { cp = b & 0x1F; extra = 1; }
        else if ((b & 0xF0) == 0xE0) { cp = b & 0x0F; extra = 2; }
        else if ((b & 0xF8) == 0xF0) { cp = b & 0x07; extra = 3; }
        else return false;
        if (extra > n - i - 1) return false;
        for (size_t j = 1; j <= extra; ++j) {
            if ((s[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        // a Rust String holds well-formed UTF-8 only (bincode's String decode fails otherwise): no overlong forms,
        // no surrogates, nothing above U+10FFFF -- the same verdicts as vaporetto_amd/modelfmt.py
        static const uint32_t kMinCp[4] = {0u, 0x80u, 0x800u, 0x10000u};
        if (cp < kMinCp[extra] || cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) return false;
        out.push_back(cp);
        i += extra + 1;
    }
    return true;
}

ModelData parse_model(const uint8_t* bytes, size_t len, size_t* consumed) {
    if (len < kMagicLen || std::memcmp(bytes, kMagic, kMagicLen) != 0)
        throw ModelError("InvalidModelError: model version mismatch");  // model.rs:128-130
    Cursor c(bytes, len, kMagicLen);
    ModelData m;
    m.char_ngrams = read_ngrams(c, true);
    m.type_ngrams = read_ngrams(c, false);
    m.dict.resize(c.length());
    for (auto& r : m.dict) {
        r.word = c.string_syms();
        r.weights = c.weights();
        (void)c.string_raw();  // comment
    }
    m.bias = c.i32();
    m.char_window = c.u8();
    m.type_window = c.u8();
    m.tag_models.resize(c.length());
    for (auto& t : m.tag_models) {
        t.token = c.string_syms();
        t.tags.resize(c.length());
        for (auto& cands : t.tags) {
            cands.resize(c.length());
            for (auto& s : cands) s = c.string_raw();
        }
        t.char_ngrams = read_tag_ngrams(c, true);
        t.type_ngrams = read_tag_ngrams(c, false);
        t.bias = c.weights();
    }
    if (consumed) *consumed = c.pos();
    return m;
}

}  // namespace vpt
