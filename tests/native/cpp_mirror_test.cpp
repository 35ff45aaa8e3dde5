// Driver for include/vaporetto_hip.hpp (the C++ mirror of the crate's Predictor / Sentence on this path), built by
// tests/test_cpp_mirror.py against libvaporetto_hip.so (GPU) or the emulated build of the same sources (CPU tests).
//   cpp_mirror_test model.bin tags|plain < lines      prints, per line: tokens, scores, labels, tokenized text, char types
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "vaporetto_grapheme.hpp"
#include "vaporetto_hip.hpp"

using namespace vaporetto_hip;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const bool tags = std::string(argv[2]) == "tags";
    try {
        bytes.push_back('x');   // read_slice returns what it did not consume
        auto mr = Model::read_slice(bytes.data(), bytes.size());
        std::cout << "consumed " << mr.second << " of " << bytes.size() << "\n";
        Predictor predictor(mr.first, tags);
        std::vector<std::string> lines;
        for (std::string l; std::getline(std::cin, l);) lines.push_back(l);
        std::vector<Sentence> batch;
        for (const std::string& l : lines) {
            Sentence s = Sentence::from_raw(l);
            predictor.predict(s);
            std::cout << "tokens";
            for (const std::string& t : s.iter_tokens()) std::cout << " [" << t << "]";
            std::cout << "\nscores";
            for (int32_t v : s.boundary_scores()) std::cout << " " << v;
            std::cout << "\nlabels";
            for (uint8_t v : s.boundaries()) std::cout << " " << int(v);
            std::cout << "\ntypes";
            for (uint8_t v : s.char_types()) std::cout << " " << int(v);
            if (tags) {
                predictor.store_tag_scores(true);       // predictor.rs:510-514: fill_tags keeps every token's score vector
                s.fill_tags();
                std::cout << "\nstored";
                for (size_t c = 0; c < s.len(); ++c) {
                    if (s.tag_model(c) < 0) continue;
                    std::cout << " " << c << ":" << s.tag_model(c) << ":";
                    for (int32_t v : s.tag_scores(c)) std::cout << v << ",";
                }
                predictor.store_tag_scores(false);
            }
            std::cout << "\ntext " << s.write_tokenized_text() << "\n";
            {   // --wsconst G: the grapheme-cluster filter between predict and the writer (predict/src/main.rs:101-104, 130-134)
                Sentence g = Sentence::from_raw(l);
                predictor.predict(g);
                vaporetto_hip::ConcatGraphemeClustersFilter().filter(g);
                std::cout << "graphemes " << g.write_tokenized_text() << "\n";
            }
            batch.push_back(Sentence::from_raw(l));
        }
        predictor.predict_batch(batch);   // one launch: the same scores
        for (size_t i = 0; i < batch.size(); ++i) {
            Sentence one = Sentence::from_raw(lines[i]);
            predictor.predict(one);
            if (one.boundary_scores() != batch[i].boundary_scores() || one.boundaries() != batch[i].boundaries()) { std::cout << "BATCH MISMATCH " << i << "\n"; return 1; }
        }
        const auto toks = predictor.tokenize(lines, tags);
        for (const std::string& t : toks) std::cout << "tokenize " << t << "\n";
        // errors: the reference's messages, and update_raw leaves " " behind (sentence.rs:264-283)
        Sentence s = Sentence::from_raw("abc");
        try { s.update_raw(""); } catch (const VaporettoError& e) { std::cout << "error " << e.kind() << " " << e.what() << " -> [" << s.as_raw_text() << "]\n"; }
        try { s.update_raw(std::string("a\0b", 3)); } catch (const VaporettoError& e) { std::cout << "error " << e.kind() << " " << e.what() << "\n"; }
        try { Sentence::from_raw("x").fill_tags(); } catch (const VaporettoError& e) { std::cout << "error " << e.kind() << " " << e.what() << "\n"; }
        {   // a predicted sentence shares the library handle (the reference: a borrow with the predictor's lifetime, predictor.rs:542):
            // it stays usable after the Predictor object was moved from and after the last Predictor object is gone
            Sentence kept = Sentence::from_raw(lines.empty() ? std::string("abc") : lines[0]);
            std::string before;
            {
                Predictor p2(mr.first, tags);
                p2.predict(kept);
                if (tags) kept.fill_tags();
                before = kept.write_tokenized_text();
                Predictor p3(std::move(p2));
                if (kept.write_tokenized_text() != before) { std::cout << "LIFETIME MISMATCH (moved)\n"; return 1; }
            }
            if (tags) kept.fill_tags();
            if (kept.write_tokenized_text() != before) { std::cout << "LIFETIME MISMATCH (destroyed)\n"; return 1; }
            std::cout << "lifetime ok\n";
        }
        std::vector<uint8_t> junk(bytes.begin(), bytes.begin() + 30);
        try { Model::read_slice(junk.data(), junk.size()); } catch (const VaporettoError& e) { std::cout << "error " << e.kind() << " model\n"; }
    } catch (const VaporettoError& e) {
        std::cout << "FAILED " << e.kind() << " " << e.what() << "\n";
        return 1;
    }
    return 0;
}
