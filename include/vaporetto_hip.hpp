// vaporetto_hip.hpp -- C++ mirror of the `vaporetto` crate's boundary-scoring API on top of the C ABI (vaporetto_hip.h).
//
// The reference is compiled code (Rust) and has no FFI; INTEGRATION.md shows the Rust binding a maintainer would add.  This
// header is the same host side in C++, for callers that are not Rust: the names, argument meaning and error behaviour of
// vaporetto::{Model, Predictor, Sentence, CharacterBoundary, CharacterType, VaporettoError} (lib.rs:82-91) for the path this
// library accelerates -- Predictor::predict -- and the calls around it (fill_tags, iter_tokens, write_tokenized_text).
// Header only; link with -lvaporetto_hip.  There is no CPU path: every predict goes to the device.
//
//   auto model = vaporetto_hip::Model::read_slice(bytes.data(), bytes.size()).first;     // model.rs:127-135
//   vaporetto_hip::Predictor predictor(model, /*predict_tags=*/true);                     // predictor.rs:450
//   auto s = vaporetto_hip::Sentence::from_raw("まぁ社長は火星猫だ");                        // sentence.rs:217
//   predictor.predict(s);                                                                 // predictor.rs:518
//   s.fill_tags();                                                                        // sentence.rs:1144
//   std::string out = s.write_tokenized_text();                                           // sentence.rs:850
//
// One sentence per call costs two kernel launches and a synchronisation: anything with more than a handful of sentences
// belongs in Predictor::predict_batch (one launch for all of them) or Predictor::tokenize (lines in, tokenized lines out).
#ifndef VAPORETTO_HIP_HPP
#define VAPORETTO_HIP_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vaporetto_hip.h"

namespace vaporetto_hip {

// errors.rs:15-38.  `what()` is the reference's Display text where it defines one.
class VaporettoError : public std::runtime_error {
public:
    enum Kind { InvalidModel, InvalidArgument, Device };
    VaporettoError(Kind kind, const std::string& msg) : std::runtime_error(msg), kind_(kind) {}
    Kind kind() const { return kind_; }

private:
    Kind kind_;
};

namespace detail {
inline void check(vpt_status st) {
    if (st == VPT_OK) return;
    const std::string msg = vpt_last_error();
    throw VaporettoError(st == VPT_INVALID_MODEL ? VaporettoError::InvalidModel
                                                 : st == VPT_INVALID_ARGUMENT ? VaporettoError::InvalidArgument : VaporettoError::Device,
                         msg);
}
// What a Predictor and the sentences it predicted share: the library handle and the store_tag_scores flag (predictor.rs:436).
struct Shared {
    vpt_predictor* raw = nullptr;
    bool store_tag_scores = false;
    ~Shared() { if (raw) vpt_predictor_destroy(raw); }
};
}  // namespace detail

// sentence.rs:70-82
enum class CharacterBoundary : uint8_t { NotWordBoundary = VPT_NOT_WORD_BOUNDARY, WordBoundary = VPT_WORD_BOUNDARY, Unknown = VPT_BOUNDARY_UNKNOWN };

// sentence.rs:24-48 (the discriminants the models are trained with)
enum class CharacterType : uint8_t { Digit = 1, Roman = 2, Hiragana = 3, Katakana = 4, Kanji = 5, Other = 6 };

// CharacterType::get_type, sentence.rs:50-67
inline CharacterType get_type(char32_t c) {
    if ((c >= 0x30 && c <= 0x39) || (c >= 0xFF10 && c <= 0xFF19)) return CharacterType::Digit;
    if ((c >= 0x41 && c <= 0x5A) || (c >= 0x61 && c <= 0x7A) || (c >= 0xFF21 && c <= 0xFF3A) || (c >= 0xFF41 && c <= 0xFF5A)) return CharacterType::Roman;
    if (c >= 0x3040 && c <= 0x3096) return CharacterType::Hiragana;
    if ((c >= 0x30A0 && c <= 0x30FA) || (c >= 0x30FC && c <= 0x30FF) || (c >= 0xFF66 && c <= 0xFF9F)) return CharacterType::Katakana;
    if ((c >= 0x3400 && c <= 0x4DBF) || (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x20000 && c <= 0x2A6DF) ||
        (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF) || (c >= 0x2F800 && c <= 0x2FA1F))
        return CharacterType::Kanji;
    return CharacterType::Other;
}

// model.rs:58-70: the bytes of a model ("VaporettoTokenizer 0.5.0\n" + bincode), validated by the library's decoder.
class Model {
public:
    // Model::read_slice (model.rs:127-135): the model and how many bytes it took (the rest belongs to the caller).
    static std::pair<Model, size_t> read_slice(const uint8_t* data, size_t len) {
        size_t used = 0;
        detail::check(vpt_model_read_len(data, len, &used));
        return {Model(std::vector<uint8_t>(data, data + used)), used};
    }
    const std::vector<uint8_t>& to_vec() const { return bytes_; }   // model.rs:99-104

private:
    explicit Model(std::vector<uint8_t> b) : bytes_(std::move(b)) {}
    std::vector<uint8_t> bytes_;
};

class Predictor;

// sentence.rs:85-101, raw-text path: the text, its character types, and after `predict` the boundary scores and labels.
class Sentence {
public:
    Sentence() { set_default(); }                                   // Sentence::default: a single space (sentence.rs:116)
    static Sentence from_raw(const std::string& text) {             // sentence.rs:217-245
        Sentence s;
        s.parse_raw(text);
        return s;
    }
    void update_raw(const std::string& text) {                      // sentence.rs:264-283: on error the sentence becomes " "
        try {
            parse_raw(text);
        } catch (const VaporettoError&) {
            set_default();
            throw;
        }
    }
    const std::string& as_raw_text() const { return text_; }                         // sentence.rs:782
    size_t len() const { return char_types_.size(); }                                // chars
    const std::vector<uint8_t>& char_types() const { return char_types_; }           // sentence.rs:1034
    const std::vector<uint8_t>& boundaries() const { return boundaries_; }           // sentence.rs:993 (CharacterBoundary values)
    std::vector<uint8_t>& boundaries_mut() { return boundaries_; }                   // sentence.rs:1016
    const std::vector<int32_t>& boundary_scores() const { return scores_; }          // sentence.rs:1040-1046 (empty before predict)
    const std::vector<size_t>& char_to_str_pos() const { return char_pos_; }         // len() + 1 byte positions
    uint32_t n_tags() const { return n_tags_; }                                      // sentence.rs:1161
    // Candidate indices per (char, slot), -1 = None; an entry is set on a token's LAST char (sentence.rs:1068 holds the strings:
    // the C ABI hands out indices, write_tokenized_text the strings).
    const std::vector<int32_t>& tag_indices() const { return tags_; }
    // Token::tag_candidates' numbers (sentence.rs:1218-1250) for the token that ENDS at char `c`: the tag model it matched (index
    // in Model::tag_models order, -1: none) and its score vector, slot after slot over the slots with >= 2 candidates.  Like the
    // reference this needs Predictor::store_tag_scores(true) before fill_tags (it throws otherwise, where the reference panics).
    int32_t tag_model(size_t c) const { need_scores(); return tag_models_.at(c); }
    std::vector<int32_t> tag_scores(size_t c) const {
        need_scores();
        return std::vector<int32_t>(tag_scores_.begin() + c * score_stride_, tag_scores_.begin() + (c + 1) * score_stride_);
    }
    inline void fill_tags();                                                         // sentence.rs:1144-1148
    inline std::string write_tokenized_text() const;                                 // sentence.rs:850-886

    // TokenIterator (sentence.rs:1265-1309), surfaces only: tokens next to an Unknown boundary are skipped.
    std::vector<std::string> iter_tokens() const {
        std::vector<std::string> out;
        const size_t n = len();
        size_t start = 0;
        bool valid = true;
        for (size_t e = 0; e < n; ++e) {
            const uint8_t b = e == n - 1 ? uint8_t(VPT_WORD_BOUNDARY) : boundaries_[e];
            if (b == VPT_BOUNDARY_UNKNOWN) valid = false;
            else if (b == VPT_WORD_BOUNDARY) {
                if (valid) out.push_back(text_.substr(char_pos_[start], char_pos_[e + 1] - char_pos_[start]));
                start = e + 1;
                valid = true;
            }
        }
        return out;
    }

private:
    friend class Predictor;
    void need_scores() const {
        if (tag_models_.empty()) throw VaporettoError(VaporettoError::InvalidArgument, "Predictor::store_tag_scores() must be set to true to use this function.");
    }
    void set_default() {                                            // sentence.rs:140-158
        text_ = " ";
        char_types_.assign(1, uint8_t(CharacterType::Other));
        char_pos_ = {0, 1};
        boundaries_.clear(); scores_.clear(); tags_.clear(); tag_scores_.clear(); tag_models_.clear();
        n_tags_ = 0; predictor_.reset();
    }
    void parse_raw(const std::string& text) {                       // sentence.rs:160-196
        if (text.find('\0') != std::string::npos) throw VaporettoError(VaporettoError::InvalidArgument, "InvalidArgumentError: text: must not contain NULL");
        if (text.empty()) throw VaporettoError(VaporettoError::InvalidArgument, "InvalidArgumentError: text: must contain at least one character");
        std::vector<uint8_t> types;
        std::vector<size_t> pos;
        for (size_t i = 0; i < text.size();) {                      // (a Rust &str is valid UTF-8; here: lead bytes delimit the chars)
            const unsigned char b0 = static_cast<unsigned char>(text[i]);
            size_t n = b0 < 0x80 ? 1 : b0 < 0xE0 ? 2 : b0 < 0xF0 ? 3 : 4;
            if (i + n > text.size()) n = text.size() - i;
            char32_t c = n == 1 ? b0 : n == 2 ? (b0 & 0x1F) : n == 3 ? (b0 & 0x0F) : (b0 & 0x07);
            for (size_t k = 1; k < n; ++k) c = (c << 6) | (static_cast<unsigned char>(text[i + k]) & 0x3F);
            pos.push_back(i);
            types.push_back(uint8_t(get_type(c)));
            i += n;
        }
        pos.push_back(text.size());
        text_ = text;
        char_types_ = std::move(types);
        char_pos_ = std::move(pos);
        boundaries_.assign(char_types_.size() - 1, uint8_t(VPT_BOUNDARY_UNKNOWN));
        scores_.clear(); tags_.clear(); tag_scores_.clear(); tag_models_.clear();
        n_tags_ = 0; predictor_.reset();
    }

    std::string text_;
    std::vector<uint8_t> char_types_;
    std::vector<size_t> char_pos_;
    std::vector<uint8_t> boundaries_;
    std::vector<int32_t> scores_;
    std::vector<int32_t> tags_;
    std::vector<int32_t> tag_scores_, tag_models_;   // only with store_tag_scores (sentence.rs:96)
    uint32_t n_tags_ = 0, score_stride_ = 0;
    // predictor.rs:542: the predictor that last predicted this sentence.  The reference ties the two with a lifetime
    // (`Sentence<'_, 'a>` borrows `&'a Predictor`); here the sentence SHARES the library handle, so fill_tags() and
    // write_tokenized_text() stay valid when the Predictor object has been moved from or destroyed in the meantime.
    std::shared_ptr<detail::Shared> predictor_;
};

// predictor.rs:433-665.  Immutable after construction; `predict` takes `&self`, any number of threads may share one.
class Predictor {
public:
    Predictor(const Model& model, bool predict_tags, int device_id = 0) : predict_tags_(predict_tags) {   // Predictor::new, predictor.rs:450-508
        raw_ = std::make_shared<detail::Shared>();
        detail::check(vpt_predictor_create(model.to_vec().data(), model.to_vec().size(), predict_tags ? 1 : 0, device_id, &raw_->raw));
    }
    // the handle is reference-counted (sentences that were predicted hold it too): moving is cheap, copying is not offered
    // -- the reference's Predictor is not Clone either
    Predictor(const Predictor&) = delete;
    Predictor& operator=(const Predictor&) = delete;
    Predictor(Predictor&&) noexcept = default;
    Predictor& operator=(Predictor&&) noexcept = default;

    const vpt_predictor* raw() const { return raw_->raw; }
    uint32_t n_tags() const { uint32_t n = 0; detail::check(vpt_predictor_n_tags(raw_->raw, &n)); return n; }
    void store_tag_scores(bool flag) { raw_->store_tag_scores = flag; }   // predictor.rs:510-514

    // Predictor::predict (predictor.rs:518-543): scores and labels of one sentence.
    void predict(Sentence& s) const {
        const size_t n = s.len();
        std::vector<int32_t> scores(n > 1 ? n - 1 : 1);
        std::vector<uint8_t> labels(n > 1 ? n - 1 : 1);
        size_t nb = 0;
        detail::check(vpt_predict_one(raw_->raw, reinterpret_cast<const uint8_t*>(s.text_.data()), s.text_.size(), scores.data(), labels.data(), &nb));
        scores.resize(nb); labels.resize(nb);
        s.scores_ = std::move(scores); s.boundaries_ = std::move(labels);
        s.tags_.clear(); s.tag_scores_.clear(); s.tag_models_.clear(); s.n_tags_ = 0; s.predictor_ = raw_;   // tags and their scores go together (sentence.rs:96, predictor.rs:599-601)
    }

    // The same for many sentences with one launch (flags: VPT_FLAG_*).
    void predict_batch(std::vector<Sentence>& sentences, unsigned flags = 0) const {
        if (sentences.empty()) return;
        std::string text;
        std::vector<uint64_t> boff(1, 0), ooff(1, 0);
        for (const Sentence& s : sentences) {
            text += s.text_;
            boff.push_back(text.size());
            ooff.push_back(ooff.back() + (s.len() - 1));
        }
        std::vector<int32_t> scores(size_t(ooff.back()) + 1);
        std::vector<uint8_t> labels(size_t(ooff.back()) + 1);
        detail::check(vpt_predict_batch_flags(raw_->raw, reinterpret_cast<const uint8_t*>(text.data()), boff.data(), sentences.size(), scores.data(),
                                              labels.data(), ooff.data(), flags));
        for (size_t i = 0; i < sentences.size(); ++i) {
            Sentence& s = sentences[i];
            s.scores_.assign(scores.begin() + ooff[i], scores.begin() + ooff[i + 1]);
            s.boundaries_.assign(labels.begin() + ooff[i], labels.begin() + ooff[i + 1]);
            s.tags_.clear(); s.tag_scores_.clear(); s.tag_models_.clear(); s.n_tags_ = 0; s.predictor_ = raw_;   // tags and their scores go together (sentence.rs:96, predictor.rs:599-601)
        }
    }

    // Lines in, tokenized lines out: the CLI's loop (predict/src/main.rs:122-176) for a batch, everything on the device.
    std::vector<std::string> tokenize(const std::vector<std::string>& lines, bool tagged = false, unsigned flags = 0) const {
        std::vector<std::string> out;
        if (lines.empty()) return out;
        std::string text;
        std::vector<uint64_t> boff(1, 0);
        for (const std::string& l : lines) { text += l; boff.push_back(text.size()); }
        uint32_t sfx = 0;
        if (tagged) detail::check(vpt_predictor_max_tag_suffix(raw_->raw, &sfx));
        std::vector<uint8_t> buf(3 * text.size() + text.size() * sfx + 16);
        std::vector<uint64_t> toff(lines.size() + 1);
        detail::check(vpt_tokenize_batch(raw_->raw, reinterpret_cast<const uint8_t*>(text.data()), boff.data(), lines.size(), flags, tagged ? 1 : 0, buf.data(),
                                         buf.size(), toff.data()));
        for (size_t i = 0; i < lines.size(); ++i) out.emplace_back(buf.begin() + toff[i], buf.begin() + toff[i + 1]);
        return out;
    }

private:
    friend class Sentence;
    std::shared_ptr<detail::Shared> raw_;
    bool predict_tags_;
};

inline void Sentence::fill_tags() {
    if (!predictor_) throw VaporettoError(VaporettoError::InvalidArgument, "InvalidArgumentError: sentence: predict() has not been called");
    const uint64_t boff[2] = {0, text_.size()}, ooff[2] = {0, len() - 1};
    uint32_t nt = 0, stride = 0;
    detail::check(vpt_predictor_n_tags(predictor_->raw, &nt));
    std::vector<int32_t> tags(len() * size_t(nt) + 1);
    tag_scores_.clear(); tag_models_.clear();
    if (predictor_->store_tag_scores && nt != 0) {   // predictor.rs:563-566
        detail::check(vpt_predictor_tag_score_stride(predictor_->raw, &stride));
        std::vector<int32_t> sc(len() * size_t(stride) + 1), md(len(), -1);
        detail::check(vpt_fill_tags_scores_batch(predictor_->raw, reinterpret_cast<const uint8_t*>(text_.data()), boff, 1, ooff, boundaries_.data(), 0u,
                                                 tags.data(), sc.data(), md.data()));
        sc.resize(len() * size_t(stride));
        tag_scores_ = std::move(sc); tag_models_ = std::move(md); score_stride_ = stride;
    } else {
        detail::check(vpt_fill_tags_batch(predictor_->raw, reinterpret_cast<const uint8_t*>(text_.data()), boff, 1, ooff, boundaries_.data(), tags.data()));
    }
    tags.resize(len() * size_t(nt));
    tags_ = std::move(tags);
    n_tags_ = nt;
}

inline std::string Sentence::write_tokenized_text() const {
    // The writer is the device's (the same bytes Sentence::write_tokenized_text produces): with tags when fill_tags has been
    // called, which is when the reference's Sentence holds any.
    const uint64_t boff[2] = {0, text_.size()}, ooff[2] = {0, len() - 1};
    uint64_t toff[2] = {0, 0};
    const bool tagged = n_tags_ != 0 && predictor_ != nullptr;
    uint32_t sfx = 0;
    if (tagged) detail::check(vpt_predictor_max_tag_suffix(predictor_->raw, &sfx));
    std::vector<uint8_t> buf(3 * text_.size() + text_.size() * sfx + 16);
    if (tagged) {
        detail::check(vpt_write_tagged_batch(predictor_->raw, reinterpret_cast<const uint8_t*>(text_.data()), boff, 1, ooff, boundaries_.data(), 0u,
                                             buf.data(), buf.size(), toff));
    } else {
        if (!predictor_) throw VaporettoError(VaporettoError::InvalidArgument, "InvalidArgumentError: sentence: predict() has not been called");
        detail::check(vpt_write_tokenized_batch(predictor_->raw, reinterpret_cast<const uint8_t*>(text_.data()), boff, 1, ooff, boundaries_.data(),
                                                buf.data(), buf.size(), toff));
    }
    return std::string(buf.begin(), buf.begin() + toff[1]);
}

}  // namespace vaporetto_hip

#endif  // VAPORETTO_HIP_HPP
