// Example / hardware smoke test of the C ABI (include/vaporetto_hip.h): lines in, tokenized lines out.
//   g++ -O2 -std=c++17 -Iinclude -o tokenize_lines examples/tokenize_lines.cpp -Lvaporetto_amd/lib -lvaporetto_hip -Wl,-rpath,'$ORIGIN/../vaporetto_amd/lib'
//   ./tokenize_lines model.bin [--tags] < lines.txt          (model.bin: an un-zstd'ed Vaporetto model)
//   ./tokenize_lines model.bin --self-test                   (the reference's own tagged outputs for resources/model.bin)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "vaporetto_hip.h"

static int tokenize(const vpt_predictor* p, const std::vector<std::string>& lines, bool tagged, std::vector<std::string>* out) {
    std::string text;
    std::vector<uint64_t> off(1, 0);
    for (const std::string& l : lines) { text += l; off.push_back(text.size()); }
    uint32_t sfx = 0;
    if (tagged && vpt_predictor_max_tag_suffix(p, &sfx) != VPT_OK) return 1;
    std::vector<uint8_t> buf(3 * text.size() + text.size() * sfx + 16);
    std::vector<uint64_t> toff(lines.size() + 1);
    const vpt_status st = vpt_tokenize_batch(p, reinterpret_cast<const uint8_t*>(text.data()), off.data(), lines.size(),
                                             VPT_FLAG_KYTEA_FULLWIDTH, tagged ? 1 : 0, buf.data(), buf.size(), toff.data());
    if (st != VPT_OK) { std::fprintf(stderr, "error: %s\n", vpt_last_error()); return 1; }
    for (size_t i = 0; i < lines.size(); ++i) out->emplace_back(buf.begin() + toff[i], buf.begin() + toff[i + 1]);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s model.bin [--tags | --self-test]\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> model((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const bool self_test = argc > 2 && !std::strcmp(argv[2], "--self-test");
    const bool tagged = self_test || (argc > 2 && !std::strcmp(argv[2], "--tags"));
    vpt_predictor* p = nullptr;
    if (vpt_predictor_create(model.data(), model.size(), tagged ? 1 : 0, 0, &p) != VPT_OK) {
        std::fprintf(stderr, "error: %s\n", vpt_last_error());
        return 1;
    }
    int rc = 0;
    if (self_test) {   // resources/docs.tok lines of the reference, and a surface that needs escaping
        const std::vector<std::string> lines = {"まぁ社長は火星猫だ", "まぁ良いだろう", "abc/d e"};
        const std::vector<std::string> want = {"まぁ/名詞/マー 社長/名詞/シャチョー は/助詞/ワ 火星/名詞/カセー 猫/名詞/ネコ だ/助動詞/ダ",
                                               "まぁ/副詞/マー 良い/形容詞/ヨイ だろう/助動詞/ダロー", "abc\\/d\\ e"};
        std::vector<std::string> got;
        rc = tokenize(p, lines, true, &got);
        for (size_t i = 0; rc == 0 && i < want.size(); ++i)
            if (got[i] != want[i]) { std::printf("MISMATCH line %zu: %s\n", i, got[i].c_str()); rc = 1; }
        std::printf(rc == 0 ? "self-test ok: %s\n" : "self-test FAILED (%s)\n", vpt_version());
    } else {
        std::vector<std::string> lines, out;
        for (std::string l; std::getline(std::cin, l);)
            if (!l.empty()) lines.push_back(l);
        rc = tokenize(p, lines, tagged, &out);
        for (const std::string& s : out) std::puts(s.c_str());
    }
    vpt_predictor_destroy(p);
    return rc;
}
