// The reference's README example (README.md:96-133: Model::read, Predictor::new, Sentence::from_raw, predict, fill_tags,
// write_tokenized_text) through the C++ mirror include/vaporetto_hip.hpp:
//   g++ -O2 -std=c++17 -Iinclude -o predict_sentence examples/predict_sentence.cpp -Lvaporetto_amd/lib -lvaporetto_hip -Wl,-rpath,'$ORIGIN/../vaporetto_amd/lib'
//   ./predict_sentence model.bin "まぁ社長は火星猫だ"          (model.bin: an un-zstd'ed Vaporetto model)
#include <fstream>
#include <iostream>
#include <iterator>
#include <vector>

#include "vaporetto_hip.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: " << argv[0] << " model.bin TEXT\n"; return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    const std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        const vaporetto_hip::Model model = vaporetto_hip::Model::read_slice(bytes.data(), bytes.size()).first;
        const vaporetto_hip::Predictor predictor(model, /*predict_tags=*/true);
        vaporetto_hip::Sentence s = vaporetto_hip::Sentence::from_raw(argv[2]);
        predictor.predict(s);
        s.fill_tags();
        std::cout << s.write_tokenized_text() << "\n";
        for (const std::string& token : s.iter_tokens()) std::cout << token << "\n";
    } catch (const vaporetto_hip::VaporettoError& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
