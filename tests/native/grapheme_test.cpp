// TEST DRIVER: lines of UTF-8 on stdin -> the lengths of their extended grapheme clusters (include/vaporetto_grapheme.hpp), one line each.
#include <cstdio>
#include <iostream>
#include <string>

#include "vaporetto_grapheme.hpp"

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {   // (the test escapes CR / LF as \r / \n: a line is a sentence)
        std::string s;
        for (size_t i = 0; i < line.size(); ++i) {
            if (line[i] == '\\' && i + 1 < line.size() && (line[i + 1] == 'r' || line[i + 1] == 'n' || line[i + 1] == '\\')) {
                s += line[i + 1] == 'r' ? '\r' : line[i + 1] == 'n' ? '\n' : '\\';
                ++i;
            } else s += line[i];
        }
        bool first = true;
        for (uint32_t n : vaporetto_hip::grapheme_cluster_lengths(s)) { std::printf(first ? "%u" : " %u", n); first = false; }
        std::printf("\n");
    }
    return 0;
}
