// Host-side model loader: Model::read_slice of the reference (model.rs:127-135), restated in C++.
// Records mirror ngram_model.rs:6-27, dict_model.rs:18-22, model.rs:41-47,61-70.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace vpt {

using Sym = uint32_t;               // a Unicode scalar value, or a character-type id for type n-grams
using SymString = std::vector<Sym>;

struct ModelError : std::runtime_error {  // VaporettoError::InvalidModel / bincode DecodeError
    using std::runtime_error::runtime_error;
};

struct NgramRecord {        // NgramData<T>           (ngram_model.rs:6-9)
    SymString ngram;
    std::vector<int32_t> weights;
};
struct DictRecord {         // WordWeightRecord       (dict_model.rs:18-22); the comment is not needed to score
    SymString word;
    std::vector<int32_t> weights;
};
struct TagWeightRecord {    // TagWeight              (ngram_model.rs:15-18)
    uint8_t rel_position;
    std::vector<int32_t> weights;
};
struct TagNgramRecord {     // TagNgramData<T>        (ngram_model.rs:21-24)
    SymString ngram;
    std::vector<TagWeightRecord> weights;
};
struct TagModelRecord {     // TagModel               (model.rs:41-47)
    SymString token;
    std::vector<std::vector<std::string>> tags;
    std::vector<TagNgramRecord> char_ngrams, type_ngrams;
    std::vector<int32_t> bias;
};
struct ModelData {          // ModelData              (model.rs:61-70)
    std::vector<NgramRecord> char_ngrams, type_ngrams;
    std::vector<DictRecord> dict;
    int32_t bias = 0;
    uint8_t char_window = 0, type_window = 0;
    std::vector<TagModelRecord> tag_models;
};

// Decodes "VaporettoTokenizer 0.5.0\n" + bincode-2 standard encoding.  Throws ModelError.
// `consumed` (optional) receives the number of bytes read (read_slice returns the remainder).
ModelData parse_model(const uint8_t* bytes, size_t len, size_t* consumed = nullptr);

// UTF-8 -> scalar values.  Returns false on malformed input.
bool decode_utf8(const uint8_t* s, size_t n, SymString& out);

}  // namespace vpt
