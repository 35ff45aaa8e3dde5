// Table compiler.  Reference behaviour mirrored here:
//   CharScorer::new / CharScorerBoundary::new     char_scorer.rs:92-124, char_scorer/boundary_scorer.rs:56-89
//   TypeScorer::new (variant choice)              type_scorer.rs:104-144
//   TypeScorerBoundaryCache::new                  type_scorer/boundary_scorer_cache.rs:22-57
//   TypeScorerBoundary::new                       type_scorer/boundary_scorer.rs:45-62
// See layout.h for why all-matches tables give the same sums as the reference's merged automaton.
#include "tables.hpp"

#include <algorithm>
#include <string>
#include <unordered_map>

namespace vpt {
namespace {

inline int32_t wadd(int32_t a, int32_t b) { return int32_t(uint32_t(a) + uint32_t(b)); }  // release builds wrap

struct Pat {
    SymString s;
    std::vector<int32_t> row;  // row_len(n, W) totals, first entry = boundary (start + row_lo(n, W))
};

bool has_zero(const SymString& s) {
    for (Sym c : s)
        if (c == 0) return true;
    return false;
}

// n-gram weights: w[k] -> boundary start + n-1-W + k   (offset -W from the END position, boundary_scorer.rs:63)
void add_ngram(std::vector<Pat>& out, const SymString& g, const std::vector<int32_t>& w, int W, bool is_char) {
    const int n = int(g.size());
    if (n == 0) throw ModelError("InvalidModelError: failed to build the automaton");  // daachorse rejects ""
    const int cap = std::max(0, 2 * W - n + 1);
    if (int(w.size()) > cap)
        throw ModelError(std::string("InvalidModelError: ") + (is_char ? "character" : "character type") +
                         " n-gram weight vector is longer than 2*window_size-n+1");
    if (w.empty() || has_zero(g)) return;  // contributes nothing / can never match a sentence
    Pat p;
    p.s = g;
    p.row.assign(size_t(row_len(n, W)), 0);
    const int base = (n - 1 - W) - row_lo(n, W);
    for (size_t k = 0; k < w.size(); ++k) p.row[size_t(base) + k] = w[k];
    out.push_back(std::move(p));
}

// dictionary word weights: w[k] -> boundary start - 1 + k   (offset -len from the END position, rs:67-74)
void add_word(std::vector<Pat>& out, const SymString& g, const std::vector<int32_t>& w, int W) {
    const size_t n = g.size();
    if (n == 0) throw ModelError("InvalidModelError: failed to build the automaton");
    if (n > 32767)
        throw ModelError("InvalidModelError: words must be shorter than or equal to 32767 characters");
    if (w.size() > n + 1)
        throw ModelError("InvalidModelError: dictionary weight vector is longer than the word length + 1");
    if (w.empty() || has_zero(g)) return;
    Pat p;
    p.s = g;
    p.row.assign(size_t(row_len(int(n), W)), 0);
    const int base = -1 - row_lo(int(n), W);
    for (size_t k = 0; k < w.size(); ++k) p.row[size_t(base) + k] = w[k];
    out.push_back(std::move(p));
}

uint32_t bits_for(size_t count) {  // capacity 2^bits >= 2*count, at least 16
    uint32_t bits = 4;
    while ((size_t(1) << bits) < count * 2) ++bits;
    return bits;
}

HostPatternTable build_table(std::vector<Pat>& pats, int W, uint32_t uni_n) {
    HostPatternTable t;
    t.present = true;
    t.window = W;
    t.uni_n = uni_n;
    for (int n = 1; n <= 3; ++n) {
        t.lo[n - 1] = row_lo(n, W);
        t.len[n - 1] = row_len(n, W);
    }
    // identical strings are summed (CharWeightMerger::add, char_scorer.rs:37-47)
    std::sort(pats.begin(), pats.end(), [](const Pat& a, const Pat& b) { return a.s < b.s; });
    size_t o = 0;
    for (size_t i = 0; i < pats.size(); ++i) {
        if (o > 0 && pats[o - 1].s == pats[i].s) {
            for (size_t k = 0; k < pats[i].row.size(); ++k) pats[o - 1].row[k] = wadd(pats[o - 1].row[k], pats[i].row[k]);
        } else {
            if (o != i) pats[o] = std::move(pats[i]);
            ++o;
        }
    }
    pats.resize(o);

    // ---- long trie (strings of more than 3 symbols)
    struct Root { uint32_t node; };
    std::unordered_map<uint64_t, uint32_t> root3;   // short_key(3-prefix) -> node id
    std::unordered_map<uint64_t, uint32_t> edge;    // edge_key(parent, sym) -> child
    std::vector<uint32_t> node_woff(1, kNoRow);     // node 0 is unused (0 = "no continuation")
    std::vector<uint32_t> node_kids(1, 0);          // number of outgoing edges per node
    for (const Pat& p : pats) {
        const size_t n = p.s.size();
        t.max_pattern = std::max<uint32_t>(t.max_pattern, uint32_t(n));
        if (n <= 3) continue;
        t.has_long = true;
        uint64_t k3 = short_key(p.s[0], p.s[1], p.s[2]);
        auto it = root3.find(k3);
        uint32_t node;
        if (it == root3.end()) {
            node = uint32_t(node_woff.size());
            node_woff.push_back(kNoRow);
            node_kids.push_back(0);
            root3.emplace(k3, node);
        } else node = it->second;
        for (size_t i = 3; i < n; ++i) {
            uint64_t ek = edge_key(node, p.s[i]);
            auto e = edge.find(ek);
            if (e == edge.end()) {
                uint32_t child = uint32_t(node_woff.size());
                node_woff.push_back(kNoRow);
                node_kids.push_back(0);
                edge.emplace(ek, child);
                ++node_kids[node];
                node = child;
            } else node = e->second;
        }
        node_woff[node] = uint32_t(t.wdata.size());
        t.wdata.insert(t.wdata.end(), p.row.begin(), p.row.end());
    }
    t.n_long_nodes = uint32_t(node_woff.size() - 1);
    if (t.wdata.empty()) t.wdata.push_back(0);

    // ---- short entries (<= 3 symbols) and their slot count
    t.slots = uint32_t(std::max(std::max(t.len[0], t.len[1]), t.len[2] + 1));
    t.ext_slot = t.slots - 1;
    t.stride_dw = (2 + t.slots + 3) & ~3u;
    t.uni_dw = (t.slots + 3) & ~3u;
    t.uni.assign(size_t(uni_n) * t.uni_dw, 0);

    struct ShortEnt { uint64_t key; const std::vector<int32_t>* row; uint32_t ext; };
    std::vector<ShortEnt> ents;
    std::unordered_map<uint64_t, size_t> ent_of;  // only for 3-symbol keys that need a continuation
    for (const Pat& p : pats) {
        const size_t n = p.s.size();
        if (n > 3) continue;
        if (n == 1 && p.s[0] < uni_n) {
            for (size_t k = 0; k < p.row.size(); ++k) t.uni[size_t(p.s[0]) * t.uni_dw + k] = uint32_t(p.row[k]);
            ++t.n_short;
            continue;
        }
        uint64_t key = short_key(p.s[0], n > 1 ? p.s[1] : 0, n > 2 ? p.s[2] : 0);
        if (n == 3) ent_of.emplace(key, ents.size());
        ents.push_back({key, &p.row, 0});
    }
    for (const auto& r : root3) {  // prefix closure at level 3 only
        auto it = ent_of.find(r.first);
        if (it == ent_of.end()) ents.push_back({r.first, nullptr, r.second});
        else ents[it->second].ext = r.second;
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
        if (e.row)
            for (size_t k = 0; k < e.row->size(); ++k) d[2 + k] = uint32_t((*e.row)[k]);
        if (e.ext) d[2 + t.ext_slot] = e.ext;
    }

    // ---- edge table: buckets of kEdgeBucket (4) edges = 64 bytes
    t.edge_bits = std::max<uint32_t>(bits_for(edge.size()), 4);
    const uint32_t eb_bits = t.edge_bits - 2, eb_mask = (1u << eb_bits) - 1;
    t.edges.assign((size_t(1) << t.edge_bits) * 4, 0);
    for (const auto& e : edge) {
        const uint32_t home = hash_slot(e.first, 32 - eb_bits);
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
        d[0] = uint32_t(e.first);
        d[1] |= uint32_t(e.first >> 32);
        d[2] = e.second | (node_kids[e.second] ? kHasKidsBit : 0u);  // node ids stay below 2^31
        d[3] = node_woff[e.second];
    }
    return t;
}

// ---- packed tables (layout.h, "PACKED TABLES").  `pats` must already be sorted and merged (build_table did it).
// Not eligible (present = false) when a pattern symbol is outside [1, 0xFFFE]: the general tables/kernel handle
// such a model.  A merged row with a value outside i16 (the same string as n-gram AND dictionary word can sum past
// 16 bits) keeps its slot with zero weights and kPkWide: the kernel then takes the row from the general tables
// (patterns of <= 3 chars) or reads the row as i32 (longer patterns).
inline bool fits_i16(int32_t v) { return v >= -32768 && v <= 32767; }
inline uint32_t pack16(int32_t lo, int32_t hi) { return (uint32_t(lo) & 0xFFFFu) | (uint32_t(hi) << 16); }

struct PackedInserter {
    std::vector<uint32_t>& tab;
    uint32_t bits, mask;
    int empty_dw;   // the dword whose value 0 marks an empty slot
    int flag_dw;    // flags live in the HIGH half of this dword
    uint32_t n_disp = 0, max_probe = 0;
    PackedInserter(std::vector<uint32_t>& t, size_t count, int empty_dw_, int flag_dw_) : tab(t), empty_dw(empty_dw_), flag_dw(flag_dw_) {
        bits = bits_for(count);
        mask = (1u << bits) - 1;
        tab.assign((size_t(1) << bits) * 4, 0);
    }
    // places {d0..d3} at home or the next free slot; returns the slot
    uint32_t insert(uint32_t home, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
        uint32_t s = home, probes = 1;
        while (tab[size_t(s) * 4 + empty_dw] != 0) { s = (s + 1) & mask; ++probes; }
        max_probe = std::max(max_probe, probes);
        uint32_t* d = &tab[size_t(s) * 4];
        d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3;
        if (s != home) { tab[size_t(home) * 4 + flag_dw] |= kPkDisp << 16; ++n_disp; }
        return s;
    }
};

HostPackedTable build_packed(const std::vector<Pat>& pats) {
    HostPackedTable t;
    for (const Pat& p : pats) {
        for (Sym c : p.s)
            if (c == 0 || c >= kPackedNoMatchSym) return t;
    }
    auto wide = [](const Pat& p) {
        for (int32_t v : p.row)
            if (!fits_i16(v)) return true;
        return false;
    };
    t.uni.assign(size_t(65536) * 4, 0);
    struct Node { uint32_t parent; Sym sym; uint32_t depth; const Pat* pat; uint32_t kids; uint32_t id; };
    // trigram-level keys: patterns of 3 chars and 3-char prefixes of longer ones
    struct Tri { const Pat* pat; uint32_t kids; uint32_t slot; };
    std::unordered_map<uint64_t, uint32_t> tri_of;   // short_key -> index in tris
    std::vector<Tri> tris;
    std::vector<uint64_t> tri_keys;
    std::vector<Node> nodes;                          // trie nodes of depth >= 4
    std::unordered_map<uint64_t, uint32_t> child_of;  // (parent ref << 21 | sym) -> index in nodes; parent ref: tri index or 2^31 | node index
    auto tri_index = [&](const SymString& s) {
        const uint64_t k = short_key(s[0], s[1], s[2]);
        auto it = tri_of.find(k);
        if (it != tri_of.end()) return it->second;
        const uint32_t i = uint32_t(tris.size());
        tris.push_back({nullptr, 0, 0});
        tri_keys.push_back(k);
        tri_of.emplace(k, i);
        return i;
    };
    size_t n_bi = 0;
    for (const Pat& p : pats) {
        const size_t n = p.s.size();
        if (n == 1) {
            uint32_t* d = &t.uni[size_t(p.s[0]) * 4];
            if (wide(p)) { d[3] = kPkWide; ++t.n_wide; }
            else { d[0] = pack16(p.row[0], p.row[1]); d[1] = pack16(p.row[2], p.row[3]); d[2] = pack16(p.row[4], p.row[5]); }
        } else if (n == 2) ++n_bi;
        else if (n == 3) tris[tri_index(p.s)].pat = &p;
        else {
            uint32_t ti = tri_index(p.s);
            uint64_t ref = ti;           // parent reference
            uint32_t* kids = nullptr;
            for (size_t i = 3; i < n; ++i) {
                const uint64_t ck = (ref << 21) | p.s[i];
                auto it = child_of.find(ck);
                uint32_t ni;
                if (it == child_of.end()) {
                    ni = uint32_t(nodes.size());
                    nodes.push_back({uint32_t(ref), p.s[i], uint32_t(i + 1), nullptr, 0, 0});
                    child_of.emplace(ck, ni);
                    if (ref & kPackedEdgeId) ++nodes[ref & ~kPackedEdgeId].kids; else ++tris[ref].kids;
                } else ni = it->second;
                ref = uint64_t(kPackedEdgeId) | ni;
            }
            (void)kids;
            nodes[ref & ~kPackedEdgeId].pat = &p;
        }
    }
    // bigrams
    {
        PackedInserter ins(t.bi, n_bi, 0, 3);
        for (const Pat& p : pats) {
            if (p.s.size() != 2) continue;
            const uint32_t key = p.s[0] | (p.s[1] << 16);
            // flags share dword 3 with w[4]: insert() only ORs into the high half
            if (wide(p)) { ins.insert(packed_hash1(key, 32 - ins.bits), key, 0, 0, kPkWide << 16); ++t.n_wide; }
            else ins.insert(packed_hash1(key, 32 - ins.bits), key, pack16(p.row[0], p.row[1]), pack16(p.row[2], p.row[3]), pack16(p.row[4], 0));
        }
        t.bi_bits = ins.bits; t.n_bi = uint32_t(n_bi); t.n_disp_bi = ins.n_disp; t.max_probe = std::max(t.max_probe, ins.max_probe);
    }
    // trigram level
    {
        PackedInserter ins(t.tri, tris.size(), 0, 1);
        for (size_t i = 0; i < tris.size(); ++i) {
            const uint64_t k = tri_keys[i];
            const uint32_t c1 = uint32_t(k & 0x1FFFFF), c2 = uint32_t((k >> 21) & 0x1FFFFF), c3 = uint32_t(k >> 42);
            const uint32_t klo = c1 | (c2 << 16);
            const Pat* p = tris[i].pat;
            uint32_t fl = tris[i].kids ? kPkHasKids : 0u;
            if (p && wide(*p)) { fl |= kPkWide; p = nullptr; ++t.n_wide; }
            tris[i].slot = ins.insert(packed_hash2(klo, c3, 32 - ins.bits), klo, c3 | (fl << 16),
                                      p ? pack16(p->row[0], p->row[1]) : 0u, p ? pack16(p->row[2], p->row[3]) : 0u);
        }
        t.tri_bits = ins.bits; t.n_tri = uint32_t(tris.size()); t.n_disp_tri = ins.n_disp; t.max_probe = std::max(t.max_probe, ins.max_probe);
    }
    // deeper levels, one depth at a time (a child's key needs its parent's slot)
    {
        PackedInserter ins(t.edge, nodes.size(), 1, 1);
        std::vector<std::vector<uint32_t>> by_depth;
        for (uint32_t i = 0; i < nodes.size(); ++i) {
            if (nodes[i].depth >= by_depth.size()) by_depth.resize(nodes[i].depth + 1);
            by_depth[nodes[i].depth].push_back(i);
        }
        for (const auto& level : by_depth) {
            for (uint32_t i : level) {
                Node& nd = nodes[i];
                const uint32_t parent = (nd.parent & kPackedEdgeId) ? (kPackedEdgeId | nodes[nd.parent & ~kPackedEdgeId].id) : tris[nd.parent].slot;
                uint32_t fl = nd.kids ? kPkHasKids : 0u, woff = 0;
                if (nd.pat) {
                    fl |= kPkHasRow;
                    woff = uint32_t(t.wrows.size() / 4);
                    const std::vector<int32_t>& r = nd.pat->row;   // depth + 1 values, first = boundary s - 1
                    if (wide(*nd.pat)) {
                        fl |= kPkWide;
                        ++t.n_wide;
                        for (int32_t v : r) t.wrows.push_back(uint32_t(v));
                    } else
                        for (size_t j = 0; j < r.size(); j += 2) t.wrows.push_back(pack16(r[j], j + 1 < r.size() ? r[j + 1] : 0));
                    while (t.wrows.size() % 4) t.wrows.push_back(0);
                }
                nd.id = ins.insert(packed_hash2(parent, nd.sym, 32 - ins.bits), parent, nd.sym | (fl << 16), woff, 0);
            }
        }
        t.edge_bits = ins.bits; t.n_edge = uint32_t(nodes.size()); t.n_disp_edge = ins.n_disp; t.max_probe = std::max(t.max_probe, ins.max_probe);
    }
    if (t.wrows.empty()) t.wrows.assign(4, 0);
    t.present = true;
    return t;
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
    if (!((m.char_ngrams.empty() && m.dict.empty()) || wc == 0)) {
        if (wc > kMaxWindow) throw ModelError("InvalidModelError: char_window_size above 8 is not supported");
        std::vector<Pat> pats;
        pats.reserve(m.char_ngrams.size() + m.dict.size());
        for (const auto& d : m.char_ngrams) add_ngram(pats, d.ngram, d.weights, wc, true);
        for (const auto& d : m.dict) add_word(pats, d.word, d.weights, wc);
        c.chars = build_table(pats, wc, kUniDirectChars);
        if (wc == 3) c.packed = build_packed(pats);
    }

    // TypeScorer::new: None without n-grams or window (type_scorer.rs:109-111); the window table when
    // window <= 3 and no tag models take part, pattern matching otherwise (type_scorer.rs:113-131)
    const int wt = m.type_window;
    if (!(m.type_ngrams.empty() || wt == 0)) {
        c.type_window = wt;
        if (!tags_on && wt <= 3) {
            c.type_kind = kTypeWindowTable;
            c.type_table = build_type_window_table(m.type_ngrams, wt);
        } else {
            if (wt > kMaxWindow) throw ModelError("InvalidModelError: type_window_size above 8 is not supported");
            c.type_kind = kTypePatternTable;
            std::vector<Pat> pats;
            for (const auto& d : m.type_ngrams) add_ngram(pats, d.ngram, d.weights, wt, false);
            c.types = build_table(pats, wt, kUniDirectTypes);
        }
    }
    c.pad = std::max(1, std::max(c.chars.present ? wc : 0, c.type_kind != kTypeNone ? wt : 0));
    return c;
}

}  // namespace vpt
