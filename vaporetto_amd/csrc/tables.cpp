// Table compiler.  Reference behaviour mirrored here:
//   CharScorer::new / CharScorerBoundary::new     char_scorer.rs:92-124, char_scorer/boundary_scorer.rs:56-89
//   TypeScorer::new (variant choice)              type_scorer.rs:104-144
//   TypeScorerBoundaryCache::new                  type_scorer/boundary_scorer_cache.rs:22-57
//   TypeScorerBoundary::new                       type_scorer/boundary_scorer.rs:45-62
// See layout.h for why all-matches tables give the same sums as the reference's merged automaton.
#include "tables.hpp"
#include "patset.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>
#include <unordered_map>

namespace vpt {
namespace {

// VPT_DEBUG_TIMING=1: stage times of the table compiler on stderr
struct StageTimer {
    const bool on = std::getenv("VPT_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vpt compile] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

inline int32_t wadd(int32_t a, int32_t b) { return int32_t(uint32_t(a) + uint32_t(b)); }  // release builds wrap

bool has_zero(const SymString& s) {
    for (Sym c : s)
        if (c == 0) return true;
    return false;
}

// n-gram weights: w[k] -> boundary start + n-1-W + k   (offset -W from the END position, boundary_scorer.rs:63).
// WL >= W is the window the ROW is laid out for (row_lo / row_len): a model trained with a char window of 1 or 2 is stored in
// the rows of window 3 -- its weights at the boundaries they belong to, zeros around them -- and so runs on the tables and the
// kernel of the window every distributed model has (train/src/main.rs:33-51 lets --charw be anything).
void add_ngram(PatSet& out, const SymString& g, const std::vector<int32_t>& w, int W, bool is_char, int WL = 0) {
    if (WL < W) WL = W;
    const int n = int(g.size());
    if (n == 0) throw ModelError("InvalidModelError: failed to build the automaton");  // daachorse rejects ""
    const int cap = std::max(0, 2 * W - n + 1);
    if (int(w.size()) > cap)
        throw ModelError(std::string("InvalidModelError: ") + (is_char ? "character" : "character type") +
                         " n-gram weight vector is longer than 2*window_size-n+1");
    if (w.empty() || has_zero(g)) return;  // contributes nothing / can never match a sentence
    int32_t* row = out.add(g, size_t(row_len(n, WL)));
    const int base = (n - 1 - W) - row_lo(n, WL);   // >= 0: row_lo(n, WL) <= n - 1 - WL <= n - 1 - W
    for (size_t k = 0; k < w.size(); ++k) row[size_t(base) + k] = w[k];
}

// dictionary word weights: w[k] -> boundary start - 1 + k   (offset -len from the END position, rs:67-74)
void add_word(PatSet& out, const SymString& g, const std::vector<int32_t>& w, int W) {
    const size_t n = g.size();
    if (n == 0) throw ModelError("InvalidModelError: failed to build the automaton");
    if (n > 32767)
        throw ModelError("InvalidModelError: words must be shorter than or equal to 32767 characters");
    if (w.size() > n + 1)
        throw ModelError("InvalidModelError: dictionary weight vector is longer than the word length + 1");
    if (w.empty() || has_zero(g)) return;
    int32_t* row = out.add(g, size_t(row_len(int(n), W)));
    const int base = -1 - row_lo(int(n), W);
    for (size_t k = 0; k < w.size(); ++k) row[size_t(base) + k] = w[k];
}

uint32_t bits_for(size_t count) {  // capacity 2^bits >= 2*count, at least 16
    uint32_t bits = 4;
    while ((size_t(1) << bits) < count * 2) ++bits;
    return bits;
}

// General tables (layout.h, top) from the sorted, merged pattern set.  Sorted order puts the patterns that share a
// prefix next to each other, so the long trie is built by one scan with the current path on a stack.
HostPatternTable build_table(const PatSet& S, int W, uint32_t uni_n) {
    StageTimer tm;
    HostPatternTable t;
    t.present = true;
    t.window = W;
    t.uni_n = uni_n;
    for (int n = 1; n <= 3; ++n) {
        t.lo[n - 1] = row_lo(n, W);
        t.len[n - 1] = row_len(n, W);
    }

    // ---- long trie (strings of more than 3 symbols)
    struct Edge { uint32_t parent; Sym sym; uint32_t child; };
    struct Root { uint64_t key3; uint32_t node; };
    std::vector<Edge> edges;
    std::vector<Root> roots;                        // one per 3-symbol prefix of a long pattern, in key order
    std::vector<uint32_t> node_woff(1, kNoRow);     // node 0 is unused (0 = "no continuation")
    std::vector<uint32_t> node_kids(1, 0);          // number of outgoing edges per node
    {
        std::vector<uint32_t> path;                 // path[i] = node of depth 3 + i of the previous long pattern
        const PatRef* prev = nullptr;
        auto new_node = [&]() { node_woff.push_back(kNoRow); node_kids.push_back(0); return uint32_t(node_woff.size() - 1); };
        for (const PatRef& p : S.p) {
            t.max_pattern = std::max<uint32_t>(t.max_pattern, p.n);
            if (p.n <= 3) continue;
            t.has_long = true;
            uint32_t keep = (prev && prev->key3 == p.key3) ? S.lcp(*prev, p) : 0;   // symbols already on the path (>= 3 then)
            if (keep < 3) {
                path.clear();
                path.push_back(new_node());
                roots.push_back({p.key3, path.back()});
                keep = 3;
            } else path.resize(keep - 2);
            const Sym* s = S.s(p);
            for (uint32_t i = keep; i < p.n; ++i) {
                const uint32_t child = new_node();
                edges.push_back({path.back(), s[i], child});
                ++node_kids[path.back()];
                path.push_back(child);
            }
            node_woff[path.back()] = uint32_t(t.wdata.size());
            t.wdata.insert(t.wdata.end(), S.row(p), S.row(p) + p.rlen);
            prev = &p;
        }
    }
    t.n_long_nodes = uint32_t(node_woff.size() - 1);
    if (t.wdata.empty()) t.wdata.push_back(0);
    tm.mark("general: long trie");

    // ---- short entries (<= 3 symbols) and their slot count
    t.slots = uint32_t(std::max(std::max(t.len[0], t.len[1]), t.len[2] + 1));
    t.ext_slot = t.slots - 1;
    t.stride_dw = (2 + t.slots + 3) & ~3u;
    t.uni_dw = (t.slots + 3) & ~3u;
    t.uni.assign(size_t(uni_n) * t.uni_dw, 0);

    struct ShortEnt { uint64_t key; const PatRef* pat; uint32_t ext; };
    std::vector<ShortEnt> ents;
    {
        auto key_of = [](uint64_t key3) { return short_key(uint32_t(key3 >> 42), uint32_t(key3 >> 21) & 0x1FFFFFu, uint32_t(key3) & 0x1FFFFFu); };
        size_t ri = 0;   // the 3-symbol prefixes of long patterns (prefix closure at level 3 only) merge in by key order
        for (const PatRef& p : S.p) {
            if (p.n > 3) continue;
            while (ri < roots.size() && roots[ri].key3 < p.key3) { ents.push_back({key_of(roots[ri].key3), nullptr, roots[ri].node}); ++ri; }
            const Sym* s = S.s(p);
            if (p.n == 1 && s[0] < uni_n) {
                for (uint32_t k = 0; k < p.rlen; ++k) t.uni[size_t(s[0]) * t.uni_dw + k] = uint32_t(S.row(p)[k]);
                ++t.n_short;
                continue;
            }
            uint32_t ext = 0;
            if (ri < roots.size() && roots[ri].key3 == p.key3) ext = roots[ri++].node;   // only a 3-symbol pattern can tie
            ents.push_back({key_of(p.key3), &p, ext});
        }
        for (; ri < roots.size(); ++ri) ents.push_back({key_of(roots[ri].key3), nullptr, roots[ri].node});
    }
    t.n_short += uint32_t(ents.size());
    // buckets of kShortBucket (2) entries = 64 bytes when the entry is 32 bytes: a lookup reads its whole home
    // bucket at once; only keys that overflow a bucket are displaced to the following buckets
    t.short_bits = bits_for(ents.size());
    const uint32_t sb_bits = t.short_bits - 1, sb_mask = (1u << sb_bits) - 1;
    t.short_tab.assign((size_t(1) << t.short_bits) * t.stride_dw, 0);
    auto slot_used = [&](size_t slot) { return (t.short_tab[slot * t.stride_dw] | t.short_tab[slot * t.stride_dw + 1]) != 0; };
    for (const ShortEnt& e : ents) {
        const uint32_t home = hash_slot(e.key, 32 - sb_bits);
        uint32_t b = home, probes = 1;
        size_t slot;
        for (;;) {
            slot = size_t(b) * kShortBucket;
            if (!slot_used(slot)) break;
            if (!slot_used(slot + 1)) { ++slot; break; }
            b = (b + 1) & sb_mask;
            ++probes;
        }
        t.max_probe_short = std::max(t.max_probe_short, probes);
        // a key that overflowed its home bucket marks that bucket (kDisplacedBit of the first entry's key_hi):
        // a lookup that finds neither the key nor the mark in the home bucket knows the key is absent
        if (b != home) { t.short_tab[size_t(home) * kShortBucket * t.stride_dw + 1] |= kDisplacedBit; ++t.n_displaced_short; }
        uint32_t* d = &t.short_tab[slot * t.stride_dw];
        d[0] = uint32_t(e.key);
        d[1] |= uint32_t(e.key >> 32);
        if (e.pat)
            for (uint32_t k = 0; k < e.pat->rlen; ++k) d[2 + k] = uint32_t(S.row(*e.pat)[k]);
        if (e.ext) d[2 + t.ext_slot] = e.ext;
    }
    tm.mark("general: short table");

    // ---- edge table: buckets of kEdgeBucket (4) edges = 64 bytes
    t.edge_bits = std::max<uint32_t>(bits_for(edges.size()), 4);
    const uint32_t eb_bits = t.edge_bits - 2, eb_mask = (1u << eb_bits) - 1;
    t.edges.assign((size_t(1) << t.edge_bits) * 4, 0);
    for (const Edge& e : edges) {
        const uint64_t ek = edge_key(e.parent, e.sym);
        const uint32_t home = hash_slot(ek, 32 - eb_bits);
        uint32_t b = home, probes = 1;
        size_t slot = 0;
        for (bool placed = false; !placed;) {
            for (uint32_t k = 0; k < kEdgeBucket; ++k) {
                slot = size_t(b) * kEdgeBucket + k;
                if ((t.edges[slot * 4] | t.edges[slot * 4 + 1]) == 0) { placed = true; break; }
            }
            if (!placed) { b = (b + 1) & eb_mask; ++probes; }
        }
        t.max_probe_edge = std::max(t.max_probe_edge, probes);
        if (b != home) t.edges[size_t(home) * kEdgeBucket * 4 + 1] |= kDisplacedBit;
        uint32_t* d = &t.edges[slot * 4];
        d[0] = uint32_t(ek);
        d[1] |= uint32_t(ek >> 32);
        d[2] = e.child | (node_kids[e.child] ? kHasKidsBit : 0u);  // node ids stay below 2^31
        d[3] = node_woff[e.child];
    }
    tm.mark("general: edge table");
    return t;
}

// ---- packed tables (layout.h, "PACKED TABLES").  `S_in` must be sorted and merged.
// Not eligible (present = false) only when the format cannot hold the model (an alphabet of 65 534 chars or more, bigram bases past
// their 19 bits, more than 2^28 nodes of a level): the general tables/kernel handle such a model.  A merged row with a value outside its fields keeps its node with zero weights and a wide flag (patterns
// of <= 3 chars: the kernel takes the row from the general tables) or goes to `xrows` as i32 (longer patterns).
inline bool fits_i16(int32_t v) { return v >= -32768 && v <= 32767; }
inline uint32_t pack16(int32_t lo, int32_t hi) { return (uint32_t(lo) & 0xFFFFu) | (uint32_t(hi) << 16); }
// n signed `bits`-wide fields, little-endian from bit 0 of d[0..] (OR-ed in: other bits are left alone)
inline void pack_fields(uint32_t* d, const int32_t* v, int n, int bits) {
    for (int j = 0; j < n; ++j) {
        const uint64_t f = uint64_t(uint32_t(v[j])) & ((uint64_t(1) << bits) - 1);
        const int bit = bits * j, q = bit >> 5, r = bit & 31;
        d[q] |= uint32_t(f << r);
        if (r + bits > 32) d[q + 1] |= uint32_t(f >> (32 - r));
    }
}
inline bool row_fits(const int32_t* row, uint32_t n, int bits) {
    for (uint32_t k = 0; k < n; ++k)
        if (!fits_field(row[k], bits)) return false;
    return true;
}

// mini-table (layout.h): `size` consecutive entries of `dw` dwords each; returns the ref, entries zeroed
uint32_t mini_alloc(std::vector<uint32_t>& arena, uint32_t dw, size_t count) {
    uint32_t lg = 0;
    const size_t want = count <= 2 ? count : count + count / 3 + 1;
    while ((size_t(1) << lg) < want) ++lg;
    const size_t base = arena.size() / dw;
    if (base >= (size_t(1) << 27)) throw ModelError("InvalidModelError: too many patterns for the packed tables");
    arena.resize(arena.size() + (size_t(dw) << lg), 0);
    return uint32_t(base << 5) | lg;
}
uint32_t* mini_insert(std::vector<uint32_t>& arena, uint32_t dw, uint32_t ref, uint32_t sym) {
    const uint32_t size = 1u << (ref & 31u), base = ref >> 5;
    uint32_t i = packed_mini_slot(sym, ref);
    while (arena[(size_t(base) + i) * dw] != 0) i = (i + 1) & (size - 1);   // count <= size: a free entry exists
    return &arena[(size_t(base) + i) * dw];
}

// First-fit placement of child rows into a double array (Tarjan-Yao row displacement): `cols` = the sorted child symbols
// of one parent; returns the smallest base d >= from, a multiple of `align`, with every slot d + cols[i] free, and takes
// the slots.  A failed try jumps past the run of taken slots that blocked it instead of stepping by one.
struct Occupancy {
    std::vector<uint64_t> w;
    bool test(size_t i) const { return (i >> 6) < w.size() && ((w[i >> 6] >> (i & 63)) & 1u); }
    void set(size_t i) {
        if ((i >> 6) >= w.size()) w.resize(std::max((i >> 6) + 1, w.size() * 2), 0);
        w[i >> 6] |= uint64_t(1) << (i & 63);
    }
    size_t next_zero(size_t i) const {
        size_t q = i >> 6;
        if (q >= w.size()) return i;
        uint64_t v = ~w[q] & (~uint64_t(0) << (i & 63));
        while (v == 0) {
            if (++q >= w.size()) return q << 6;
            v = ~w[q];
        }
        return (q << 6) + size_t(__builtin_ctzll(v));
    }
    // `first` = the slot the row's first child takes; the base is first - cols[0], which may be negative when `align` is 1
    // (the kernel adds modulo 2^32); otherwise the base is >= 0 and a multiple of `align`
    size_t place(const uint32_t* cols, size_t n, size_t align, size_t from_first) {
        const size_t c0 = cols[0];
        auto legal = [&](size_t first) {   // the first legal position at or after `first`
            if (align == 1) return first;
            if (first < c0) first = c0;
            return c0 + (first - c0 + align - 1) / align * align;
        };
        size_t first = legal(from_first);
        for (;;) {
            size_t i = 0;
            for (; i < n; ++i)
                if (test(first + (cols[i] - c0))) break;
            if (i == n) break;
            const size_t free_at = next_zero(first + (cols[i] - c0));   // the first slot this child could take
            first = legal(free_at - (cols[i] - c0));                    // > first
        }
        for (size_t i = 0; i < n; ++i) set(first + (cols[i] - c0));
        return first;
    }
};
// Rows arrive largest first.  A row starts its search at the lowest free slot, except that a row of two or more children
// does not go back further than a look-back window behind where the last row of its size class went: what did not take
// that row will hardly take this one, and the holes left behind are filled by the one-child rows, which come last and
// fit anywhere.  Keeps the whole placement linear in the table size.
struct Placer {
    Occupancy occ;
    size_t low = 0, lookback, lookback_small;
    size_t hint[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Placer(size_t lb, size_t lb_small) : lookback(lb), lookback_small(lb_small) {}
    // returns the slot of the row's first child (the row's base = that - cols[0])
    size_t put(const uint32_t* cols, size_t n, size_t align) {
        low = occ.next_zero(low);
        size_t from = low;
        const size_t cls = n < 8 ? n : 8;
        const size_t lb = n < 8 ? lookback_small : lookback;
        if (n > 1 && hint[cls] > lb) from = std::max(from, hint[cls] - lb);
        const size_t first = occ.place(cols, n, align, from);
        hint[cls] = first;
        return first;
    }
};

HostPackedTable build_packed(const PatSet& S_in, int wl) {
    StageTimer tm;
    HostPackedTable t;
    t.wl = wl;
    const size_t udw = size_t(pk_uni_dw(wl)), bdw = size_t(pk_bi_dw(wl)), tdw = size_t(pk_tri_dw(wl));
    // (tests lower the limit to reach the refusal without a 256 MB deep arena)
    const char* kmb = std::getenv("VPT_DEBUG_KIDS_MAX_BASE");
    const uint32_t kids_max_base = kmb ? std::min<uint32_t>(kTriKidsMaxBase, uint32_t(std::strtoul(kmb, nullptr, 10))) : kTriKidsMaxBase;
    // A pattern that holds U+0000 matches no text (a sentence with a NUL is an error, sentence.rs:174-179): it takes no part here (0 is
    // the tables' "outside the sentence").  `S` = the patterns that can match; the copy is made for such a model only.
    PatSet filtered;
    {
        bool nul = false;
        for (const PatRef& p : S_in.p)
            for (uint32_t i = 0; i < p.n && !nul; ++i) nul = S_in.s(p)[i] == 0;
        if (nul) {
            filtered.syms = S_in.syms; filtered.rows = S_in.rows;
            for (const PatRef& p : S_in.p) {
                bool has = false;
                for (uint32_t i = 0; i < p.n; ++i) has = has || S_in.s(p)[i] == 0;
                if (!has) filtered.p.push_back(p);
            }
        }
    }
    const PatSet& S = filtered.syms.empty() ? S_in : filtered;
    // ---- the alphabet: ids by how many pattern symbols the char is (most first; ties in code-point order).  A char of the BMP finds
    // its id in the kernel's 65536-word table, any other in `xcid` (layout.h)
    {
        constexpr uint32_t kScalars = 0x110000u;
        std::vector<uint8_t> seen(kScalars, 0);
        t.hot.assign(kScalars, 0);
        for (const PatRef& p : S.p)
            for (uint32_t i = 0; i < p.n; ++i) {
                const Sym c = S.s(p)[i];
                if (c >= kScalars) return t;   // no char: nothing a decoded model holds
                seen[c] = 1;
                ++t.hot[c];
            }
        t.id_of.assign(65536, uint16_t(kNoId));
        t.cpid.push_back(0);
        std::vector<uint32_t> order;
        for (uint32_t cp = 1; cp < kScalars; ++cp)
            if (seen[cp]) order.push_back(cp);
        if (order.size() >= size_t(kNoId) - 1) return t;   // ids are 16 bits wide, 0 and kNoId are taken
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t.hot[a] > t.hot[b]; });
        std::vector<std::pair<uint32_t, uint32_t>> outside;
        for (uint32_t cp : order) {
            const uint32_t id = uint32_t(t.cpid.size());
            if (cp < 0x10000u) t.id_of[cp] = uint16_t(id); else outside.push_back({cp, id});
            t.cpid.push_back(cp);
        }
        t.n_alpha = uint32_t(t.cpid.size() - 1);
        t.cpid.push_back(0);   // the id every char outside the alphabet reads its (zero) unigram node at
        if (!outside.empty()) {
            uint32_t bits = 1;
            while ((size_t(1) << bits) < 2 * outside.size()) ++bits;
            t.xcid.assign(2 + (size_t(2) << bits), 0);
            t.xcid[0] = bits;
            const uint32_t mask = (1u << bits) - 1u;
            for (const auto& e : outside) {
                uint32_t i = xcid_slot(e.first, bits);
                while (t.xcid[2 + 2 * size_t(i)] != 0) i = (i + 1) & mask;
                t.xcid[2 + 2 * size_t(i)] = e.first; t.xcid[3 + 2 * size_t(i)] = e.second;
            }
        }
    }
    auto id = [&](Sym c) { return t.id_for(c); };
    auto wide16 = [&](const PatRef& p) {
        for (uint32_t k = 0; k < p.rlen; ++k)
            if (!fits_i16(S.row(p)[k])) return true;
        return false;
    };
    t.uni.assign(size_t(t.n_alpha + 2) * udw, 0);

    // ---- trie over the patterns of >= 2 chars by one scan: prefixes (depth 2) own nodes (depth >= 3)
    struct Node { uint32_t sym, depth; const PatRef* pat; uint32_t parent; uint32_t ref; };   // parent: prefix index at depth 3, node index below
    struct Prefix { uint32_t key; const PatRef* pat; uint32_t slot; };
    std::vector<Node> nodes;
    std::vector<Prefix> prefixes;
    std::vector<const PatRef*> uni_pat(size_t(t.n_alpha) + 2, nullptr);
    {
        std::vector<uint32_t> path;   // path[i] = node of depth 3 + i of the previous pattern
        const PatRef* prev = nullptr;
        for (const PatRef& p : S.p) {
            const Sym* s = S.s(p);
            if (p.n == 1) { uni_pat[id(s[0])] = &p; continue; }
            const bool same_prefix = prev && (prev->key3 >> 21) == (p.key3 >> 21);
            if (!same_prefix) {
                prefixes.push_back({id(s[0]) | (id(s[1]) << 16), nullptr, 0});
                path.clear();
            }
            if (p.n == 2) { prefixes.back().pat = &p; prev = &p; continue; }   // the first pattern of its prefix
            const uint32_t keep = same_prefix ? S.lcp(*prev, p) : 2;              // >= 2, < p.n, <= prev->n
            path.resize(keep - 2);
            for (uint32_t i = keep; i < p.n; ++i) {
                const uint32_t ni = uint32_t(nodes.size());
                nodes.push_back({id(s[i]), i + 1, nullptr, i == 2 ? uint32_t(prefixes.size() - 1) : path.back(), 0});
                path.push_back(ni);
            }
            nodes[path.back()].pat = &p;
            prev = &p;
        }
    }
    // children lists (CSR, in code-point order: a parent's children are created in that order; the placements sort them by id)
    std::vector<uint32_t> nkid_off(nodes.size() + 1, 0), pkid_off(prefixes.size() + 1, 0), nkid(0), pkid(0);
    for (const Node& nd : nodes) ++(nd.depth == 3 ? pkid_off : nkid_off)[nd.parent + 1];
    for (size_t i = 0; i < nodes.size(); ++i) nkid_off[i + 1] += nkid_off[i];
    for (size_t i = 0; i < prefixes.size(); ++i) pkid_off[i + 1] += pkid_off[i];
    nkid.resize(nkid_off.back()); pkid.resize(pkid_off.back());
    {
        std::vector<uint32_t> nfill(nkid_off.begin(), nkid_off.end() - 1), pfill(pkid_off.begin(), pkid_off.end() - 1);
        for (uint32_t i = 0; i < nodes.size(); ++i) {
            const Node& nd = nodes[i];
            if (nd.depth == 3) pkid[pfill[nd.parent]++] = i; else nkid[nfill[nd.parent]++] = i;
        }
    }
    auto n_kids = [&](uint32_t ni) { return nkid_off[ni + 1] - nkid_off[ni]; };
    auto kid = [&](uint32_t ni, uint32_t j) { return nkid[nkid_off[ni] + j]; };
    tm.mark("packed: trie");

    // (on a thread of its own: the placements below do not depend on it; joined before the nodes are written)
    // ---- deep arena: the children mini-table of every node that owns one, deepest owners first (an entry names
    // the table of the node it ends at).  Chains of nodes that carry no row and have a single child are COMPRESSED
    // into the entry of their first node (up to kPackedMaxSkip further symbols), so that a dictionary word of any
    // ordinary length costs one trie step past its third char.
    std::exception_ptr deep_err;
    std::thread deep_thread([&] { try {
    t.deep.assign(16, 0);   // entry 0 unused: ref 0 = none
    auto chain_end = [&](uint32_t ki, uint32_t* skipped, uint32_t* n_skipped) {
        uint32_t cur = ki, steps = 0;
        while (nodes[cur].pat == nullptr && n_kids(cur) == 1 && steps < kPackedMaxSkip) {
            cur = kid(cur, 0);
            if (skipped) skipped[steps] = nodes[cur].sym;
            ++steps;
        }
        if (n_skipped) *n_skipped = steps;
        return cur;
    };
    std::vector<uint32_t> owners;           // nodes that own a mini-table: depth-3 nodes and chain ends, with children
    {
        std::vector<uint32_t> work;
        for (uint32_t i = 0; i < nodes.size(); ++i)
            if (nodes[i].depth == 3 && n_kids(i) != 0) work.push_back(i);
        while (!work.empty()) {
            const uint32_t ni = work.back();
            work.pop_back();
            owners.push_back(ni);
            for (uint32_t j = 0; j < n_kids(ni); ++j) {
                const uint32_t e = chain_end(kid(ni, j), nullptr, nullptr);
                if (n_kids(e) != 0) work.push_back(e);
            }
        }
    }
    std::stable_sort(owners.begin(), owners.end(), [&](uint32_t x, uint32_t y) { return nodes[x].depth > nodes[y].depth; });
    for (uint32_t ni : owners) {
        nodes[ni].ref = mini_alloc(t.deep, 16, n_kids(ni));
        for (uint32_t j = 0; j < n_kids(ni); ++j) {
            const uint32_t ki = kid(ni, j);
            uint32_t skipped[kPackedMaxSkip], n_skipped = 0;
            const Node& k = nodes[chain_end(ki, skipped, &n_skipped)];      // the node this entry ends at
            uint32_t* e = mini_insert(t.deep, 16, nodes[ni].ref, nodes[ki].sym);
            uint32_t fl = 0;
            if (k.pat) {
                const int32_t* r = S.row(*k.pat);   // row_len(depth, wl) values from boundary s + row_lo(depth, wl) (wl = 3: depth + 1 from s - 1)
                const uint32_t rl = k.pat->rlen;
                if (rl <= kPackedInlineRow && !wide16(*k.pat)) {
                    fl |= kPkHasRow;
                    for (uint32_t q = 0; q < rl; q += 2) e[8 + q / 2] = pack16(r[q], q + 1 < rl ? r[q + 1] : 0);
                } else {
                    fl |= kPkExtRow;
                    if (wide16(*k.pat)) ++t.n_wide;
                    e[8] = uint32_t(t.xrows.size());
                    t.xrows.insert(t.xrows.end(), r, r + rl);
                }
            }
            e[0] = nodes[ki].sym | (fl << 16) | (n_skipped << 24);
            e[1] = k.ref;
            for (uint32_t q = 0; q < n_skipped; ++q) e[2 + q / 2] |= skipped[q] << (16 * (q & 1));
            ++t.n_deep;
        }
    }
    if (t.xrows.empty()) t.xrows.push_back(0);
    } catch (...) { deep_err = std::current_exception(); } });
    struct Joiner { std::thread& th; ~Joiner() { if (th.joinable()) th.join(); } } deep_join{deep_thread};

    // ---- bigram level: the bigram nodes of a first char id1 sit at (B1[id1] << bi_shift) + id2.  Rows with the most
    // children are placed first; the base must fit the 19 bits a unigram node has for it.
    std::vector<uint32_t> b1(size_t(t.n_alpha) + 2, 0);
    size_t bi_slots = 0;
    {
        struct Row { uint32_t id1, first, count; };
        std::vector<Row> rows;
        for (uint32_t pi = 0; pi < prefixes.size();) {
            uint32_t e = pi;
            while (e < prefixes.size() && (prefixes[e].key & 0xFFFFu) == (prefixes[pi].key & 0xFFFFu)) ++e;
            rows.push_back({prefixes[pi].key & 0xFFFFu, pi, e - pi});
            pi = e;
        }
        struct Col { uint32_t id2, pi; };
        std::vector<std::vector<Col>> rest(rows.size());
        for (size_t ri = 0; ri < rows.size(); ++ri) {
            const Row& r = rows[ri];
            for (uint32_t j = 0; j < r.count; ++j) rest[ri].push_back({prefixes[r.first + j].key >> 16, r.first + j});
            std::sort(rest[ri].begin(), rest[ri].end(), [](const Col& a, const Col& b) { return a.id2 < b.id2; });   // ids are not in code-point order
        }
        std::vector<uint32_t> by_size(rows.size());
        for (uint32_t i = 0; i < rows.size(); ++i) by_size[i] = i;
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return rest[x].size() > rest[y].size(); });
        std::vector<uint32_t> cols;
        // a first guess at the alignment the 19-bit base needs (no interleaving at all would take one span per row; half of
        // that is typical), raised if the placement does not fit
        size_t spans = 0;
        for (const auto& r : rest) if (!r.empty()) spans += r.back().id2 - r.front().id2 + 1;
        t.bi_shift = 2;
        while (t.bi_shift < 8 && (size_t(kUniBaseMask) << t.bi_shift) < spans / 16) ++t.bi_shift;
        for (;; ++t.bi_shift) {
            if (t.bi_shift > 8) {   // not placeable within the format: the general tables serve the model
                if (tm.on) std::fprintf(stderr, "[vpt compile] packed tables refused: bigram bases do not fit %u bits\n", 19u);
                return t;
            }
            Placer pl(8192, 8192);
            bool ok = true;
            size_t top = 0;
            for (uint32_t ri : by_size) {
                if (rest[ri].empty()) continue;
                cols.clear();
                for (const Col& c : rest[ri]) cols.push_back(c.id2);   // ascending
                const size_t d = pl.put(cols.data(), cols.size(), size_t(1) << t.bi_shift) - cols[0];
                if ((d >> t.bi_shift) > kUniBaseMask) { ok = false; break; }
                b1[rows[ri].id1] = uint32_t(d >> t.bi_shift);
                for (const Col& c : rest[ri]) prefixes[c.pi].slot = uint32_t(d + c.id2);
                top = std::max(top, d + cols.back() + 1);
            }
            if (ok) { bi_slots = top; break; }
        }
    }
    // any id2 may be asked of any base: keep 65536 nodes of slack behind the highest base
    {
        size_t max_base = 0;
        for (uint32_t v : b1) max_base = std::max<size_t>(max_base, size_t(v) << t.bi_shift);
        bi_slots = std::max(bi_slots, max_base + 65536);
    }
    if (bi_slots >= kTriParentMask) {
        if (tm.on) std::fprintf(stderr, "[vpt compile] packed tables refused: %zu bigram slots (shift %u) for %zu nodes\n", bi_slots, t.bi_shift, prefixes.size());
        return t;
    }
    tm.mark("packed: bigram placement");

    // ---- trigram level: the children of bigram node p sit at B2[p] + id3
    std::vector<uint32_t> b2(prefixes.size(), 0);
    std::vector<uint32_t> tri_slot(nodes.size(), 0);   // depth-3 nodes only
    size_t tri_slots = 1;
    {
        std::vector<uint32_t> order;
        for (uint32_t pi = 0; pi < prefixes.size(); ++pi)
            if (pkid_off[pi + 1] != pkid_off[pi]) order.push_back(pi);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return pkid_off[x + 1] - pkid_off[x] > pkid_off[y + 1] - pkid_off[y]; });
        Placer pl(4096, 256);
        pl.occ.set(0);   // slot 0 stays free: the base of a node without children is 0
        std::vector<uint32_t> cols;
        for (uint32_t pi : order) {
            cols.clear();
            for (uint32_t j = pkid_off[pi]; j < pkid_off[pi + 1]; ++j) cols.push_back(nodes[pkid[j]].sym);
            std::sort(cols.begin(), cols.end());   // ascending (ids are not in code-point order)
            const size_t first = pl.put(cols.data(), cols.size(), 1);
            b2[pi] = uint32_t(first) - cols[0];   // modulo 2^32: a row of high ids may start below slot cols[0]
            for (uint32_t j = pkid_off[pi]; j < pkid_off[pi + 1]; ++j) tri_slot[pkid[j]] = uint32_t(first + (nodes[pkid[j]].sym - cols[0]));
            tri_slots = std::max(tri_slots, first + (cols.back() - cols[0]) + 1);
        }
    }
    if (tri_slots >= (size_t(1) << 28)) return t;
    tm.mark("packed: trigram placement");

    // ---- emit
    deep_thread.join();
    if (deep_err) std::rethrow_exception(deep_err);
    tm.mark("packed: deep arena (joined)");
    const int nu = pk_uni_fields(wl), nb = pk_bi_fields(wl), nt = pk_tri_fields(wl);
    auto set_bits = [](uint32_t* d, int bit, int n, uint32_t v) {   // n <= 32 bits of v at bit `bit` of the dword string (OR-ed in)
        const uint64_t f = uint64_t(v) & ((uint64_t(1) << n) - 1);
        const int q = bit >> 5, r = bit & 31;
        d[q] |= uint32_t(f << r);
        if (r + n > 32) d[q + 1] |= uint32_t(f >> (32 - r));
    };
    for (uint32_t i = 1; i <= t.n_alpha; ++i) {
        uint32_t* d = &t.uni[size_t(i) * udw];
        if (const PatRef* p = uni_pat[i]) {   // row_len(1, wl) = 2 wl values from boundary s - wl
            if (!row_fits(S.row(*p), uint32_t(nu), kUniFieldBits)) { set_bits(d, pk_uni_base_bit(wl) + kUniBaseBits, 1, 1u); ++t.n_wide; }
            else pack_fields(d, S.row(*p), nu, kUniFieldBits);
        }
        set_bits(d, pk_uni_base_bit(wl), kUniBaseBits, b1[i]);
    }
    t.bi.assign(bi_slots * bdw, 0);
    t.tri.assign(tri_slots * tdw, 0);
    for (uint32_t pi = 0; pi < prefixes.size(); ++pi) {
        const Prefix& pf = prefixes[pi];
        uint32_t* r = &t.bi[size_t(pf.slot) * bdw];
        r[pk_bi_key_dw(wl)] = pf.key;
        uint32_t* rrow = r + pk_bi_row_dw(wl);   // row_len(2, wl) = 2 wl - 1 values from boundary s - wl + 1
        if (pf.pat && !row_fits(S.row(*pf.pat), uint32_t(nb), kBiFieldBits)) { set_bits(rrow, pk_bi_wide_bit(wl), 1, 1u); ++t.n_wide; }
        else if (pf.pat) pack_fields(rrow, S.row(*pf.pat), nb, kBiFieldBits);
        uint64_t mask = 0;
        for (uint32_t j = pkid_off[pi]; j < pkid_off[pi + 1]; ++j) {
            const Node& k = nodes[pkid[j]];
            mask |= uint64_t(1) << packed_filter_bit(k.sym);
            uint32_t* e = &t.tri[size_t(tri_slot[pkid[j]]) * tdw];
            uint32_t fl = 0;
            if (k.pat && wide16(*k.pat)) { fl |= kPkWide; ++t.n_wide; }
            else if (k.pat) {   // row_len(3, wl) = 2 wl - 2 values from boundary s - wl + 2
                const int32_t* w = S.row(*k.pat);
                for (int q = 0; q < nt; q += 2) e[pk_tri_w_dw(wl) + q / 2] = pack16(w[q], w[q + 1]);
            }
            // the child filter (layout.h, "tri"): a mini-table entry is looked up by the FIRST symbol of a compressed chain, the child's own
            uint32_t filt = 0;
            for (uint32_t q = 0; q < n_kids(pkid[j]); ++q) filt |= 1u << packed_kid_filter_bit(nodes[kid(pkid[j], q)].sym);
            if ((k.ref >> 5) >= kids_max_base) {   // the deep arena outgrew the 22 bits the child filter leaves its bases: the general tables serve the model
                if (tm.on) std::fprintf(stderr, "[vpt compile] packed tables refused: a mini-table base past %u deep entries\n", kids_max_base);
                HostPackedTable none;
                none.wl = wl;
                return none;
            }
            e[0] = (pf.slot + 1) | (fl << kTriFlagShift) | ((filt >> 5) << 28);
            e[pk_tri_kids_dw(wl)] = k.ref | ((filt & 31u) << 27);
            ++t.n_tri;
        }
        r[pk_bi_base_dw(wl)] = b2[pi]; r[pk_bi_filter_dw(wl)] = uint32_t(mask); r[pk_bi_filter_dw(wl) + 1] = uint32_t(mask >> 32);
        if (wl <= 3) r[7] = pkid_off[pi + 1] - pkid_off[pi];
    }
    t.n_bi = uint32_t(prefixes.size());
    tm.mark("packed: nodes");
    if (tm.on) std::fprintf(stderr, "[vpt compile] alphabet %u, bigram nodes %u in %zu slots (shift %u), trigram nodes %u in %zu slots, deep entries %u, wide rows %u\n",
                            t.n_alpha, t.n_bi, bi_slots, t.bi_shift, t.n_tri, tri_slots, t.n_deep, t.n_wide);
    t.hot = std::vector<uint32_t>();
    t.present = true;
    return t;
}

// Type rows (layout.h, "TYPE ROWS") for the row window `wl` (>= W); mode kTypeRowsNone when the n-grams fit neither form.
// `cache_semantics`: the n-grams are scored by the reference's window table (boundary_scorer_cache.rs:30-57), which matches code 0
// against the padding -- a start-position row cannot say "outside", so such a model has no rows; otherwise (the automaton variants,
// type_scorer/boundary_scorer.rs:45-62) an n-gram with a 0 in it matches nothing (char_types holds 1..6) and is dropped.
// `ngrams` already passed the validity checks of the table builders.
struct TypeRows { std::vector<uint32_t> data; uint32_t mode = kTypeRowsNone, levels = 0; };
TypeRows build_type_rows(const std::vector<NgramRecord>& ngrams, int W, int wl, bool cache_semantics) {
    TypeRows out;
    const int nf = 2 * wl;
    std::unordered_map<uint64_t, std::vector<int64_t>> rows;   // (n << 32 | codes, 3 bits each, first = lowest) -> the n-gram's row
    int levels = 3;
    for (const NgramRecord& d : ngrams) {
        const int n = int(d.ngram.size());
        bool usable = true, pads = false;
        for (Sym s : d.ngram) {
            if (s > 6) usable = false;   // can never match: char types are 1..6
            if (s == 0) pads = true;
        }
        if (!usable || d.weights.empty() || n == 0 || n > 2 * W) continue;
        if (pads) { if (cache_semantics) return out; continue; }
        if (n > kMaxTypeRowLevels) return out;
        levels = std::max(levels, n);
        uint64_t key = uint64_t(n) << 32;
        for (int i = 0; i < n; ++i) key |= uint64_t(d.ngram[size_t(i)]) << (3 * i);
        std::vector<int64_t>& row = rows[key];
        row.resize(size_t(nf), 0);
        // the window table only ever reads w[2W - end], end = n .. 2W (boundary_scorer_cache.rs:41-46): a weight with an
        // index above 2W - n exists in no window and is ignored there, so it is ignored here (the automaton variants reject it)
        const size_t used = std::min(d.weights.size(), size_t(2 * W - n + 1));
        for (size_t k = 0; k < used; ++k) {
            const int slot = (n - 1 - W + int(k)) + wl;   // boundary start + n-1-W+k  (type_scorer/boundary_scorer.rs:48); row slot 0 = start - wl
            row[size_t(slot)] += d.weights[k];             // in [0, 2 wl): W <= wl
        }
    }
    const uint32_t count = type_row_count(levels);
    std::vector<int64_t> all(size_t(count) * size_t(nf), 0);
    bool fits18 = true;
    for (uint32_t idx = 0; idx < count; ++idx) {
        uint32_t t[kMaxTypeRowLevels];
        uint32_t q = idx;
        t[0] = q % 6u + 1u; q /= 6u;
        for (int i = 1; i < levels; ++i) { t[i] = q % 7u; q /= 7u; }
        int64_t* r = &all[size_t(idx) * size_t(nf)];
        uint64_t key = 0;
        for (int n = 1; n <= levels; ++n) {
            if (t[n - 1] == 0) break;   // outside the sentence: only the shorter n-grams start here
            key |= uint64_t(t[n - 1]) << (3 * (n - 1));
            auto it = rows.find((uint64_t(n) << 32) | key);
            if (it == rows.end()) continue;
            for (int j = 0; j < nf; ++j) r[j] += it->second[size_t(j)];
        }
        for (int j = 0; j < nf; ++j) fits18 = fits18 && r[j] >= -131072 && r[j] <= 131071;
    }
    out.levels = uint32_t(levels);
    if (levels == 3 && fits18) {
        out.mode = kTypeRowsLds;
        const size_t dw = size_t(pk_trow_dw(wl));
        out.data.assign(size_t(count) * dw, 0);
        for (uint32_t idx = 0; idx < count; ++idx) {
            int32_t v[2 * kMaxWindow];
            for (int j = 0; j < nf; ++j) v[j] = int32_t(all[size_t(idx) * size_t(nf) + size_t(j)]);
            pack_fields(&out.data[size_t(idx) * dw], v, nf, kUniFieldBits);
        }
    } else {
        out.mode = kTypeRowsGlobal;
        const size_t dw = size_t(pk_trow_global_dw(wl));
        out.data.assign(size_t(count) * dw, 0);
        for (uint32_t idx = 0; idx < count; ++idx)
            for (int j = 0; j < nf; ++j) out.data[size_t(idx) * dw + size_t(j)] = uint32_t(uint64_t(all[size_t(idx) * size_t(nf) + size_t(j)]));   // wraps like the reference's i32 sums
    }
    return out;
}

// TypeScorerBoundaryCache::new (boundary_scorer_cache.rs:22-57): scores[seq] for every window of 2W type codes
// (3 bits each, leftmost = most significant, 0 = outside the sentence, 7 = invalid -> score 0) is the sum over
// every pattern occurrence inside the window of w[2W - end] when that index exists.
std::vector<int32_t> build_type_window_table(const std::vector<NgramRecord>& ngrams, int W) {
    const int L = 2 * W;
    std::unordered_map<uint64_t, const std::vector<int32_t>*> by_key;  // (n << 32 | packed codes) -> weights
    for (const NgramRecord& d : ngrams) {
        // the reference feeds the UNMERGED list to the automaton builder: empty or duplicate patterns fail
        if (d.ngram.empty()) throw ModelError("InvalidModelError: invalid character type n-grams");
        bool usable = int(d.ngram.size()) <= L;
        uint64_t packed = 0;
        for (Sym s : d.ngram) {
            if (s > 6) usable = false;  // windows only hold codes 0..6
            packed = (packed << 3) | (s & 7);
        }
        if (!usable) continue;
        uint64_t key = (uint64_t(d.ngram.size()) << 32) | packed;
        if (!by_key.emplace(key, &d.weights).second)
            throw ModelError("InvalidModelError: invalid character type n-grams");
    }
    {  // duplicate check for the patterns skipped above (longer than the window or with codes > 6)
        std::vector<const SymString*> all;
        for (const NgramRecord& d : ngrams) all.push_back(&d.ngram);
        std::sort(all.begin(), all.end(), [](const SymString* a, const SymString* b) { return *a < *b; });
        for (size_t i = 1; i < all.size(); ++i)
            if (*all[i] == *all[i - 1]) throw ModelError("InvalidModelError: invalid character type n-grams");
    }
    std::vector<int32_t> table(size_t(1) << (3 * L), 0);
    std::vector<uint32_t> codes(size_t(L), 0);
    const size_t total = table.size();
    for (size_t seq = 0; seq < total; ++seq) {
        bool valid = true;
        for (int i = 0; i < L; ++i) {
            uint32_t c = uint32_t(seq >> (3 * (L - 1 - i))) & 7;
            if (c == 7) { valid = false; break; }
            codes[size_t(i)] = c;
        }
        if (!valid) continue;
        int32_t y = 0;
        for (int end = 1; end <= L; ++end) {
            uint64_t packed = 0;
            for (int n = 1; n <= end; ++n) {  // pattern = codes[end-n .. end)
                packed |= uint64_t(codes[size_t(end - n)]) << (3 * (n - 1));
                auto it = by_key.find((uint64_t(n) << 32) | packed);
                if (it == by_key.end()) continue;
                size_t k = size_t(L - end);
                if (k < it->second->size()) y = wadd(y, (*it->second)[k]);
            }
        }
        table[seq] = y;
    }
    return table;
}

}  // namespace

uint32_t tag_token_hash(const uint32_t* cps, size_t n) {
    uint32_t k[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n && i < 4; ++i) k[i] = cps[i] & 0xFFFFu;
    return tag_token_hash_key(k[0] | (k[1] << 16), k[2] | (k[3] << 16), uint32_t(n));
}

namespace {
HostTagTables build_tag_tables(const ModelData& m, bool use_char, bool use_type) {
    HostTagTables t;
    t.present = true;
    t.use_char = use_char; t.use_type = use_type;
    if (m.tag_models.size() >= kTokModelMask) throw ModelError("InvalidModelError: too many tag models");
    t.n_models = uint32_t(m.tag_models.size());
    t.tok_bits = bits_for(m.tag_models.size()) + 1;   // a quarter full: a lane's probe sequence is the wave's when it is the longest
    // ... followed by the FILTER: 32 bits per slot, bit (hash >> (32 - tok_bits - 5)) set for every token of the table -- one
    // token in thirty has a tag model (BASELINE's configs[4]); the others learn it from one 4-byte read (layout.h, kTagFilterLog2)
    t.tok_tab.assign((size_t(4) << t.tok_bits) + (size_t(1) << t.tok_bits), 0);
    const uint32_t mask = (1u << t.tok_bits) - 1;
    for (uint32_t mi = 0; mi < m.tag_models.size(); ++mi) {
        const TagModelRecord& tm = m.tag_models[mi];
        t.n_tags = std::max<uint32_t>(t.n_tags, uint32_t(tm.tags.size()));   // predictor.rs:466
        uint32_t rec[12] = {0};
        rec[0] = uint32_t(t.syms.size()); rec[1] = uint32_t(tm.token.size());
        t.syms.insert(t.syms.end(), tm.token.begin(), tm.token.end());
        // The entries of a model: (n-gram, rel_position) pairs, char ones first -- ordered by rel_position, so that the kernel
        // can skip a whole group whose filter does not hold the text char the n-grams of the group must END with -- then type ones.
        bool all_compact = true;
        uint32_t rel_count[4] = {0, 0, 0, 0};
        uint64_t rel_filter[4] = {0, 0, 0, 0};
        bool rel_ok = true;
        auto add_ngrams = [&](const std::vector<TagNgramRecord>& list, uint32_t kind, uint32_t* first, uint32_t* count) {
            *first = uint32_t(t.ngrams.size() / 4);
            struct Ent { uint32_t rel; const TagNgramRecord* d; const TagWeightRecord* w; uint32_t so; };
            std::vector<Ent> ents;
            for (const TagNgramRecord& d : list) {
                const uint32_t so = uint32_t(t.syms.size());
                t.syms.insert(t.syms.end(), d.ngram.begin(), d.ngram.end());
                if (d.ngram.size() >= (size_t(1) << 24)) throw ModelError("InvalidModelError: tag n-gram too long");
                for (const TagWeightRecord& w : d.weights) ents.push_back({w.rel_position, &d, &w, so});
            }
            if (kind == 0) std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.rel < b.rel; });
            for (const Ent& e : ents) {
                const TagNgramRecord& d = *e.d;
                const TagWeightRecord& w = *e.w;
                bool compact = d.ngram.size() <= kTagFastSyms;
                for (Sym c : d.ngram) compact = compact && c < 0xFFFFu;
                t.ngrams.push_back(e.so);
                t.ngrams.push_back(uint32_t(d.ngram.size()) | (uint32_t(w.rel_position) << 24));
                t.ngrams.push_back(uint32_t(t.weights.size()));
                t.ngrams.push_back(uint32_t(w.weights.size()));
                uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                r[0] = (compact ? uint32_t(d.ngram.size()) : 0u) | (uint32_t(w.rel_position) << 8) | (kind << 16) | (compact ? 1u << 17 : 0u) |
                       (uint32_t(std::min<size_t>(w.weights.size(), 255)) << 24);
                r[1] = uint32_t(t.weights.size());
                if (compact)
                    for (size_t j = 0; j < d.ngram.size(); ++j) r[2 + j / 2] |= d.ngram[j] << (16 * (j & 1));
                t.nrec.insert(t.nrec.end(), r, r + 8);
                all_compact = all_compact && compact;
                if (kind == 0) {
                    if (w.rel_position > 3 || d.ngram.empty()) rel_ok = false;
                    else { ++rel_count[w.rel_position]; rel_filter[w.rel_position] |= uint64_t(1) << packed_filter_bit(d.ngram.back()); }
                } else if (w.rel_position > kTagFastMaxRel) {
                    rel_ok = false;   // a type tag n-gram that ends further past the token than the pass's context window holds (type windows above 4:
                }                     // found by the round-4 fuzz over every window) -- the whole-wave routine scores such a model
                t.weights.insert(t.weights.end(), w.weights.begin(), w.weights.end());
            }
            *count = uint32_t(t.ngrams.size() / 4) - *first;
        };
        add_ngrams(tm.char_ngrams, 0u, &rec[2], &rec[3]);
        add_ngrams(tm.type_ngrams, 1u, &rec[4], &rec[5]);
        // the fast path walks char and type entries as one run of records
        const bool fast_model = all_compact && rel_ok && rec[4] == rec[2] + rec[3] && tm.bias.size() <= kTagFastZ && tm.tags.size() <= 3 &&
                                rel_count[0] < 256 && rel_count[1] < 256 && rel_count[2] < 256 && rel_count[3] < 256 && rec[5] < 256;
        rec[6] = uint32_t(t.weights.size()); rec[7] = uint32_t(tm.bias.size());
        t.weights.insert(t.weights.end(), tm.bias.begin(), tm.bias.end());
        if (tm.bias.size() > kTagMaxZ) throw ModelError("InvalidModelError: more than 1024 tag scores per token are not supported");
        t.max_zlen = std::max<uint32_t>(t.max_zlen, uint32_t(tm.bias.size()));
        rec[8] = uint32_t(t.slots.size() / 2); rec[9] = uint32_t(tm.tags.size());
        uint32_t off = 0;
        for (const auto& cands : tm.tags) {   // TagPredictor::predict, predictor.rs:286-304
            t.slots.push_back(uint32_t(cands.size()));
            t.slots.push_back(off);
            if (cands.size() >= 2) off += uint32_t(cands.size());
            t.slot_str.push_back(uint32_t(t.str_off.size()));
            for (const std::string& tag : cands) {
                t.str_off.push_back(uint32_t(t.str_bytes.size()));
                for (unsigned char ch : tag) {
                    if (ch == ' ' || ch == '\\' || ch == '/') t.str_bytes.push_back('\\');
                    t.str_bytes.push_back(ch);
                }
            }
        }
        if (fast_model) {
            rec[10] = 1u;
            for (size_t j = 0; j < tm.tags.size(); ++j)
                rec[11] |= (t.slots[size_t(rec[8] + j) * 2] | (t.slots[size_t(rec[8] + j) * 2 + 1] << 5)) << (9 * j);   // <= 16 candidates, offset <= 15
        }
        t.models.insert(t.models.end(), rec, rec + 12);
        {   // mfilt: everything the fast path needs of a model in one record
            uint32_t f[kTagFiltStride] = {0};
            for (int r = 0; r < 4; ++r) { f[2 * r] = uint32_t(rel_filter[r]); f[2 * r + 1] = uint32_t(rel_filter[r] >> 32); }
            f[8] = rel_count[0] | (rel_count[1] << 8) | (rel_count[2] << 16) | (rel_count[3] << 24);
            if (fast_model) {
                f[9] = rec[2];
                f[10] = rec[5] | (uint32_t(tm.bias.size()) << 8) | (uint32_t(tm.tags.size()) << 16);
                f[11] = rec[11];
                for (size_t i = 0; i < tm.bias.size(); ++i) f[12 + i] = uint32_t(tm.bias[i]);
                for (size_t j = 0; j < tm.tags.size(); ++j) f[28 + j] = t.slot_str[size_t(rec[8]) + j];   // (<= 3 slots)
            }
            t.mfilt.insert(t.mfilt.end(), f, f + kTagFiltStride);
        }
        // token table: a repeated token keeps its slot and takes the later model
        const uint32_t th = tag_token_hash(tm.token.data(), tm.token.size());
        {
            const uint32_t fbit = th >> (32 - t.tok_bits - kTagFilterLog2);
            t.tok_tab[(size_t(4) << t.tok_bits) + (fbit >> 5)] |= 1u << (fbit & 31u);
        }
        uint32_t b = th >> (32 - t.tok_bits);
        for (;;) {
            uint32_t* e = &t.tok_tab[size_t(b) * 4];
            if (e[0] != 0) {
                const uint32_t* cr = &t.models[size_t(e[0] - 1) * 12];
                if (!(cr[1] == tm.token.size() && std::equal(tm.token.begin(), tm.token.end(), t.syms.begin() + cr[0]))) { b = (b + 1) & mask; continue; }
            }
            e[0] = mi + 1;
            bool inl = tm.token.size() <= 4;
            for (Sym c : tm.token) inl = inl && c < 0xFFFFu;
            if (tm.token.size() > kTagTokLenMask) throw ModelError("InvalidModelError: tag token too long");
            e[1] = uint32_t(tm.token.size()) | (inl ? kTagTokInline : 0u) | (fast_model ? kTagTokFast : 0u);
            e[2] = e[3] = 0;   // the hashed key: low 16 bits of the first four chars (the whole surface when inline)
            for (size_t j = 0; j < tm.token.size() && j < 4; ++j) e[2 + j / 2] |= (tm.token[j] & 0xFFFFu) << (16 * (j & 1));
            break;
        }
    }
    if (t.syms.empty()) t.syms.push_back(0);
    if (t.weights.empty()) t.weights.push_back(0);
    if (t.ngrams.empty()) t.ngrams.assign(4, 0);
    if (t.nrec.empty()) t.nrec.assign(8, 0);
    if (t.mfilt.empty()) t.mfilt.assign(kTagFiltStride, 0);
    if (t.slots.empty()) t.slots.assign(2, 0);
    if (t.slot_str.empty()) t.slot_str.push_back(0);
    t.str_off.push_back(uint32_t(t.str_bytes.size()));   // the end of the last string
    if (t.str_bytes.empty()) t.str_bytes.push_back(0);
    return t;
}
}  // namespace

uint32_t kytea_fullwidth_host(uint32_t c) {
    if (c >= 'a' && c <= 'z') return 0xFF41 + (c - 'a');   // kytea_fullwidth.rs:17-42
    if (c >= 'A' && c <= 'Z') return 0xFF21 + (c - 'A');   // :43-68
    if (c >= '0' && c <= '9') return 0xFF10 + (c - '0');   // :69-78
    switch (c) {                                           // :79-113
        case '(': return 0xFF08; case ')': return 0xFF09; case '{': return 0xFF5B; case '}': return 0xFF5D;
        case '<': return 0xFF1C; case '>': return 0xFF1E; case 0xFF62: return 0x300C; case 0xFF63: return 0x300D;
        case '[': return 0xFF3B; case ']': return 0xFF3D; case '-': return 0x2212; case 0xFF5E: return 0x301C;
        case '.': return 0x3002; case 0xFF0D: return 0x30FC; case '/': return 0xFF0F; case '_': return 0xFF3F;
        case ',': return 0xFF0C; case '%': return 0xFF05; case '?': return 0xFF1F; case 0xFF64: return 0x3001;
        case 0x2015: return 0x30FC; case '"': return 0x201D; case '\'': return 0x2019; case 0xFF65: return 0x30FB;
        case 0x2500: return 0x30FC; case '+': return 0xFF0B; case ':': return 0xFF1A; case 0x2013: return 0x30FC;
        case '!': return 0xFF01; case 0xFF61: return 0x3002; case '&': return 0xFF06; case '*': return 0xFF0A;
        case '@': return 0xFF20; case '=': return 0xFF1D;
        default: return c;
    }
}

uint8_t char_type_host(uint32_t c) {
    auto in = [c](uint32_t lo, uint32_t hi) { return c >= lo && c <= hi; };
    if (in(0x30, 0x39) || in(0xFF10, 0xFF19)) return 1;
    if (in(0x41, 0x5A) || in(0x61, 0x7A) || in(0xFF21, 0xFF3A) || in(0xFF41, 0xFF5A)) return 2;
    if (in(0x3040, 0x3096)) return 3;
    if (in(0x30A0, 0x30FA) || in(0x30FC, 0x30FF) || in(0xFF66, 0xFF9F)) return 4;
    if (in(0x3400, 0x4DBF) || in(0x4E00, 0x9FFF) || in(0xF900, 0xFAFF) || in(0x20000, 0x2A6DF) || in(0x2A700, 0x2B73F) ||
        in(0x2B740, 0x2B81F) || in(0x2B820, 0x2CEAF) || in(0x2F800, 0x2FA1F))
        return 5;
    return 6;
}

CompiledModel compile_model(const ModelData& m, bool predict_tags) {
    CompiledModel c;
    c.bias = m.bias;
    c.predict_tags = predict_tags;
    c.n_char_ngrams = uint32_t(m.char_ngrams.size());
    c.n_type_ngrams = uint32_t(m.type_ngrams.size());
    c.n_dict_words = uint32_t(m.dict.size());
    c.n_tag_models = uint32_t(m.tag_models.size());
    const bool tags_on = predict_tags && !m.tag_models.empty();  // predictor.rs:463-479
    if (tags_on) {
        // With tag models the reference adds every tag n-gram to the automata (boundary_tag_scorer.rs:87-104):
        // an empty one fails the build, a rel_position beyond the window indexes out of bounds (panic).
        for (const auto& t : m.tag_models) {
            for (const auto& d : t.char_ngrams) {
                if (d.ngram.empty() && m.char_window != 0 && !(m.char_ngrams.empty() && m.dict.empty()))
                    throw ModelError("InvalidModelError: failed to build the automaton");
                for (const auto& w : d.weights)
                    if (w.rel_position > m.char_window)
                        throw ModelError("InvalidModelError: tag n-gram rel_position exceeds char_window_size");
            }
            for (const auto& d : t.type_ngrams) {
                if (d.ngram.empty() && m.type_window != 0 && !m.type_ngrams.empty())
                    throw ModelError("InvalidModelError: failed to build the automaton");
                for (const auto& w : d.weights)
                    if (w.rel_position > m.type_window)
                        throw ModelError("InvalidModelError: tag n-gram rel_position exceeds type_window_size");
            }
        }
    }

    // CharScorer::new: None when there is nothing to match or the window is 0 (char_scorer.rs:98-100)
    const int wc = m.char_window;
    const int wt = m.type_window;
    const bool chars_on = !((m.char_ngrams.empty() && m.dict.empty()) || wc == 0);
    const bool types_on = !(m.type_ngrams.empty() || wt == 0);
    // The row window of the packed tables (layout.h, "ROW WINDOW") covers the type window too when the type scores can take the
    // start-position form (type rows, added together with a position's unigram row); whether they can is found out first -- this
    // pass checks nothing and throws nothing, the builders below do, in the reference's order.
    TypeRows type_rows;
    if (chars_on && types_on && wc <= kMaxWindow && wt <= kMaxWindow)
        type_rows = build_type_rows(m.type_ngrams, wt, pk_row_window(wc, wt), !tags_on && wt <= 3);
    const int wl = type_rows.mode != kTypeRowsNone ? pk_row_window(wc, wt) : pk_row_window(wc, 0);
    if (chars_on) {
        if (wc > kMaxWindow) throw ModelError("InvalidModelError: char_window_size above 8 is not supported");
        StageTimer tm;
        PatSet pats;
        {
            size_t n_syms = 0, n_rows = 0;
            for (const auto& d : m.char_ngrams) { n_syms += d.ngram.size(); n_rows += size_t(row_len(int(d.ngram.size()), wc)); }
            for (const auto& d : m.dict) { n_syms += d.word.size(); n_rows += size_t(row_len(int(d.word.size()), wc)); }
            pats.reserve(m.char_ngrams.size() + m.dict.size(), n_syms, n_rows);
        }
        // narrower windows are laid out in the rows of `wl` (add_ngram): one set of tables, one kernel instance per row window
        c.char_window = wc;
        for (const auto& d : m.char_ngrams) add_ngram(pats, d.ngram, d.weights, wc, true, wl);
        for (const auto& d : m.dict) add_word(pats, d.word, d.weights, wl);
        tm.mark("pattern rows");
        pats.finish();
        tm.mark("sort + merge");
        // the general and the packed tables are independent of each other: two threads
        std::exception_ptr packed_err;
        std::thread packed_thread([&] {
            try { c.packed = build_packed(pats, wl); } catch (...) { packed_err = std::current_exception(); }
        });
        try { c.chars = build_table(pats, wl, kUniDirectChars); } catch (...) { packed_thread.join(); throw; }
        packed_thread.join();
        if (packed_err) std::rethrow_exception(packed_err);
    }

    // TypeScorer::new: None without n-grams or window (type_scorer.rs:109-111); the window table when
    // window <= 3 and no tag models take part, pattern matching otherwise (type_scorer.rs:113-131)
    if (types_on) {
        c.type_window = wt;
        if (!tags_on && wt <= 3) {
            c.type_kind = kTypeWindowTable;
            c.type_table = build_type_window_table(m.type_ngrams, wt);
        } else if (wt <= 3) {
            // With tag models the reference scores types through TypeScorerBoundaryTag (type_scorer.rs:113-131): an
            // automaton over the MERGED n-grams (identical n-grams sum, TypeWeightMerger::add, type_scorer.rs:46-56).
            // The sums per boundary are the same function of the 2W-type window, so the same table / type rows serve.
            std::vector<NgramRecord> merged;
            std::vector<size_t> order(m.type_ngrams.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return m.type_ngrams[x].ngram < m.type_ngrams[y].ngram; });
            for (size_t i : order) {
                const NgramRecord& d = m.type_ngrams[i];
                if (d.ngram.empty()) throw ModelError("InvalidModelError: failed to build the automaton");
                if (int(d.weights.size()) > std::max(0, 2 * wt - int(d.ngram.size()) + 1))
                    throw ModelError("InvalidModelError: character type n-gram weight vector is longer than 2*window_size-n+1");
                // TypeScorerBoundaryTag matches the n-grams against char_types, whose bytes are 1..6 (sentence.rs:50-67): an
                // n-gram with a 0 in it matches nothing there, while the window table below would match it against the padding
                if (has_zero(d.ngram)) continue;
                if (!merged.empty() && merged.back().ngram == d.ngram) {
                    std::vector<int32_t>& w = merged.back().weights;
                    if (w.size() < d.weights.size()) w.resize(d.weights.size(), 0);
                    for (size_t k = 0; k < d.weights.size(); ++k) w[k] = wadd(w[k], d.weights[k]);
                } else merged.push_back(d);
            }
            c.type_kind = kTypeWindowTable;
            c.type_table = build_type_window_table(merged, wt);
        } else {
            if (wt > kMaxWindow) throw ModelError("InvalidModelError: type_window_size above 8 is not supported");
            c.type_kind = kTypePatternTable;
            PatSet pats;
            for (const auto& d : m.type_ngrams) add_ngram(pats, d.ngram, d.weights, wt, false);
            pats.finish();
            c.types = build_table(pats, wt, kUniDirectTypes);
        }
        if (c.packed.present && type_rows.mode != kTypeRowsNone) {
            c.packed.trow = std::move(type_rows.data); c.packed.trow_mode = type_rows.mode; c.packed.trow_levels = type_rows.levels;
        }
    }
    c.pad = std::max(1, std::max(c.chars.present ? c.chars.window : 0, c.type_kind != kTypeNone ? wt : 0));   // (the window the rows are laid out for)
    if (tags_on) c.tags = build_tag_tables(m, c.chars.present, c.type_kind != kTypeNone);
    return c;
}

}  // namespace vpt
