// Flat storage for the patterns of one scorer (char n-grams + dictionary words, or type n-grams) while the table
// compiler works on them: all symbols in one array, all rows in another, 32-byte descriptors that sort fast.
// Sorting puts every set of patterns with a common prefix next to each other, which is what lets tables.cpp build its
// tries and prefix groups by one linear scan with a stack instead of hash maps (the 1.7 M patterns of a
// bccwj-suw+unidic-sized model compile in a fraction of the time).
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "model.hpp"

namespace vpt {

struct PatRef {
    uint64_t key3;        // order-preserving key of the first three symbols: s0 << 42 | s1 << 21 | s2 (absent = 0; a symbol is never 0)
    uint32_t soff, roff;  // first symbol / first row value in PatSet::syms / PatSet::rows
    uint32_t n, rlen;     // symbols, row values
};

struct PatSet {
    std::vector<Sym> syms;
    std::vector<int32_t> rows;
    std::vector<PatRef> p;   // after finish(): sorted by string, identical strings merged (rows summed, wrapping)

    void reserve(size_t n_pats, size_t n_syms, size_t n_rows) { p.reserve(n_pats); syms.reserve(n_syms); rows.reserve(n_rows); }
    // a new pattern with a zeroed row of `rlen` values; returns the row for the caller to fill
    int32_t* add(const SymString& g, size_t rlen) {
        PatRef r;
        r.soff = uint32_t(syms.size()); r.roff = uint32_t(rows.size());
        r.n = uint32_t(g.size()); r.rlen = uint32_t(rlen);
        r.key3 = (uint64_t(g[0]) << 42) | (g.size() > 1 ? uint64_t(g[1]) << 21 : 0) | (g.size() > 2 ? uint64_t(g[2]) : 0);
        syms.insert(syms.end(), g.begin(), g.end());
        rows.resize(rows.size() + rlen, 0);
        p.push_back(r);
        return rows.data() + r.roff;
    }
    const Sym* s(const PatRef& r) const { return syms.data() + r.soff; }
    const int32_t* row(const PatRef& r) const { return rows.data() + r.roff; }
    int32_t* row(const PatRef& r) { return rows.data() + r.roff; }

    // -1 / 0 / +1 like strcmp over whole strings
    int compare(const PatRef& a, const PatRef& b) const {
        if (a.key3 != b.key3) return a.key3 < b.key3 ? -1 : 1;
        const uint32_t m = std::min(a.n, b.n);
        const Sym *x = s(a), *y = s(b);
        for (uint32_t i = 3; i < m; ++i)
            if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
        return a.n == b.n ? 0 : (a.n < b.n ? -1 : 1);
    }
    // length of the common prefix of two patterns
    uint32_t lcp(const PatRef& a, const PatRef& b) const {
        const uint32_t m = std::min(a.n, b.n);
        const Sym *x = s(a), *y = s(b);
        uint32_t i = 0;
        while (i < m && x[i] == y[i]) ++i;
        return i;
    }
    // identical strings are summed (CharWeightMerger::add, char_scorer.rs:37-47; TypeWeightMerger::add, type_scorer.rs:46-56)
    void finish() {
        auto less = [this](const PatRef& a, const PatRef& b) { return compare(a, b) < 0; };
        if (p.size() < (size_t(1) << 16)) std::sort(p.begin(), p.end(), less);
        else {   // four sorted quarters on four threads, then two merges and a last one
            const size_t q = p.size() / 4;
            PatRef* b = p.data();
            PatRef* cut[5] = {b, b + q, b + 2 * q, b + 3 * q, b + p.size()};
            std::thread t1([&] { std::sort(cut[0], cut[1], less); }), t2([&] { std::sort(cut[1], cut[2], less); }), t3([&] { std::sort(cut[2], cut[3], less); });
            std::sort(cut[3], cut[4], less);
            t1.join(); t2.join(); t3.join();
            std::thread m1([&] { std::inplace_merge(cut[0], cut[1], cut[2], less); });
            std::inplace_merge(cut[2], cut[3], cut[4], less);
            m1.join();
            std::inplace_merge(cut[0], cut[2], cut[4], less);
        }
        size_t o = 0;
        for (size_t i = 0; i < p.size(); ++i) {
            if (o > 0 && compare(p[o - 1], p[i]) == 0) {
                int32_t* d = row(p[o - 1]);
                const int32_t* e = row(p[i]);   // same string => same row geometry
                for (uint32_t k = 0; k < p[i].rlen; ++k) d[k] = int32_t(uint32_t(d[k]) + uint32_t(e[k]));
            } else {
                p[o++] = p[i];
            }
        }
        p.resize(o);
    }
};

}  // namespace vpt
