// Table compiler.  Reference behaviour mirrored here:
//   CharScorer::new / CharScorerBoundary::new     char_scorer.rs:92-124, char_scorer/boundary_scorer.rs:56-89
//   TypeScorer::new (variant choice)              type_scorer.rs:104-144
//   TypeScorerBoundaryCache::new                  type_scorer/boundary_scorer_cache.rs:22-57
//   TypeScorerBoundary::new                       type_scorer/boundary_scorer.rs:45-62
// See layout.h for why all-matches tables give the same sums as the reference's merged automaton.
#include "tables.hpp"

#include <algorithm>
#include <cstdlib>
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
// such a model.  A merged row with a value outside i16 keeps its slot with zero weights and kPkWide (patterns of
// <= 3 chars: the kernel takes the row from the general tables) or goes to `xrows` as i32 (longer patterns).
inline bool fits_i16(int32_t v) { return v >= -32768 && v <= 32767; }
inline uint32_t pack16(int32_t lo, int32_t hi) { return (uint32_t(lo) & 0xFFFFu) | (uint32_t(hi) << 16); }
// n signed `bits`-wide fields, little-endian from bit 0 of d[0..3] (OR-ed in: the flag bits are left alone)
inline void pack_fields(uint32_t* d, const int32_t* v, int n, int bits) {
    for (int j = 0; j < n; ++j) {
        const uint64_t f = uint64_t(uint32_t(v[j])) & ((uint64_t(1) << bits) - 1);
        const int bit = bits * j, q = bit >> 5, r = bit & 31;
        d[q] |= uint32_t(f << r);
        if (r + bits > 32) d[q + 1] |= uint32_t(f >> (32 - r));
    }
}
inline bool row_fits(const std::vector<int32_t>& row, int bits) {
    for (int32_t v : row)
        if (!fits_field(v, bits)) return false;
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

HostPackedTable build_packed(const std::vector<Pat>& pats) {
    HostPackedTable t;
    for (const Pat& p : pats)
        for (Sym c : p.s)
            if (c == 0 || c >= kPackedNoMatchSym) return t;
    auto wide = [](const Pat& p) {
        for (int32_t v : p.row)
            if (!fits_i16(v)) return true;
        return false;
    };
    t.uni.assign(size_t(65536) * 4, 0);

    // ---- trie over the patterns of >= 2 chars: prefixes (depth 2) own nodes (depth >= 3)
    struct Node { Sym sym; uint32_t depth; const Pat* pat; std::vector<uint32_t> kids; uint32_t ref; };
    struct Prefix { uint32_t key; const Pat* pat; std::vector<uint32_t> kids; };
    std::vector<Node> nodes;
    std::vector<Prefix> prefixes;
    std::unordered_map<uint32_t, uint32_t> prefix_of;
    std::unordered_map<uint64_t, uint32_t> pchild, nchild;   // (prefix | node index) << 21 | sym -> node index
    uint32_t max_depth = 0;
    for (const Pat& p : pats) {
        const size_t n = p.s.size();
        if (n == 1) {
            uint32_t* d = &t.uni[size_t(p.s[0]) * 4];
            if (!row_fits(p.row, kUniFieldBits)) { d[3] = kUniWideBit; ++t.n_wide; }
            else pack_fields(d, p.row.data(), 6, kUniFieldBits);
            continue;
        }
        const uint32_t key = p.s[0] | (p.s[1] << 16);
        auto it = prefix_of.find(key);
        uint32_t pi;
        if (it == prefix_of.end()) {
            pi = uint32_t(prefixes.size());
            prefixes.push_back({key, nullptr, {}});
            prefix_of.emplace(key, pi);
        } else pi = it->second;
        if (n == 2) { prefixes[pi].pat = &p; continue; }
        uint32_t cur = 0;
        for (size_t i = 2; i < n; ++i) {
            auto& map = (i == 2) ? pchild : nchild;
            const uint64_t ck = (uint64_t(i == 2 ? pi : cur) << 21) | p.s[i];
            auto f = map.find(ck);
            if (f == map.end()) {
                const uint32_t ni = uint32_t(nodes.size());
                nodes.push_back({p.s[i], uint32_t(i + 1), nullptr, {}, 0});
                map.emplace(ck, ni);
                if (i == 2) prefixes[pi].kids.push_back(ni); else nodes[cur].kids.push_back(ni);
                cur = ni;
            } else cur = f->second;
        }
        nodes[cur].pat = &p;
        max_depth = std::max<uint32_t>(max_depth, uint32_t(n));
    }
    t.n_deep = 0;

    // ---- deep arena: the children mini-table of every node that owns one, deepest owners first (an entry names
    // the table of the node it ends at).  Chains of nodes that carry no row and have a single child are COMPRESSED
    // into the entry of their first node (up to kPackedMaxSkip further symbols), so that a dictionary word of any
    // ordinary length costs one trie step past its third char.
    t.deep.assign(16, 0);   // entry 0 unused: ref 0 = none
    t.kids3.assign(4, 0);
    auto chain_end = [&](uint32_t ki, std::vector<Sym>* skipped) {
        uint32_t cur = ki, steps = 0;
        while (nodes[cur].pat == nullptr && nodes[cur].kids.size() == 1 && steps < kPackedMaxSkip) {
            cur = nodes[cur].kids[0];
            if (skipped) skipped->push_back(nodes[cur].sym);
            ++steps;
        }
        return cur;
    };
    std::vector<uint32_t> owners;           // nodes that own a mini-table: depth-3 nodes and chain ends, with children
    {
        std::vector<uint32_t> work;
        for (uint32_t i = 0; i < nodes.size(); ++i)
            if (nodes[i].depth == 3 && !nodes[i].kids.empty()) work.push_back(i);
        while (!work.empty()) {
            const uint32_t ni = work.back();
            work.pop_back();
            owners.push_back(ni);
            for (uint32_t ki : nodes[ni].kids) {
                const uint32_t e = chain_end(ki, nullptr);
                if (!nodes[e].kids.empty()) work.push_back(e);
            }
        }
    }
    std::sort(owners.begin(), owners.end(), [&](uint32_t x, uint32_t y) { return nodes[x].depth > nodes[y].depth; });
    for (uint32_t ni : owners) {
        Node& nd = nodes[ni];
        nd.ref = mini_alloc(t.deep, 16, nd.kids.size());
        for (uint32_t ki : nd.kids) {
            std::vector<Sym> skipped;
            const Node& k = nodes[chain_end(ki, &skipped)];      // the node this entry ends at
            uint32_t* e = mini_insert(t.deep, 16, nd.ref, nodes[ki].sym);
            uint32_t fl = 0;
            if (k.pat) {
                const std::vector<int32_t>& r = k.pat->row;   // depth + 1 values, first = boundary s - 1
                if (r.size() <= kPackedInlineRow && !wide(*k.pat)) {
                    fl |= kPkHasRow;
                    for (size_t j = 0; j < r.size(); j += 2) e[8 + j / 2] = pack16(r[j], j + 1 < r.size() ? r[j + 1] : 0);
                } else {
                    fl |= kPkExtRow;
                    if (wide(*k.pat)) ++t.n_wide;
                    e[8] = uint32_t(t.xrows.size());
                    t.xrows.insert(t.xrows.end(), r.begin(), r.end());
                }
            }
            e[0] = nodes[ki].sym | (fl << 16) | (uint32_t(skipped.size()) << 24);
            e[1] = k.ref;
            for (size_t j = 0; j < skipped.size(); ++j) e[2 + j / 2] |= skipped[j] << (16 * (j & 1));
            ++t.n_deep;
        }
    }
    if (t.xrows.empty()) t.xrows.push_back(0);

    // ---- placement of the trigram-level children (layout.h): a child (a,b,c) sits in a RIGHT slot of record (a,b)
    // or in a LEFT slot of record (b,c) -- the record the NEXT start position fetches anyway -- else in the overflow
    // mini-table of (a,b).  Prefixes with <= 3 children keep them all on the right; the children of bigger prefixes
    // go left first (targets with the fewest takers first), then into the 3 right slots, then overflow.
    struct Placed { uint32_t node; Sym lead; };                  // a left child: trie node + its first char
    std::vector<std::vector<Placed>> lefts(prefixes.size());
    std::vector<std::vector<uint32_t>> rights(prefixes.size()), overflow(prefixes.size());
    {
        struct Cand { uint32_t parent, node, target_key; };
        std::vector<Cand> cands;
        std::unordered_map<uint32_t, uint32_t> ldeg;
        for (uint32_t pi = 0; pi < prefixes.size(); ++pi) {
            const Prefix& pf = prefixes[pi];
            if (pf.kids.size() <= 3) { rights[pi] = pf.kids; continue; }
            for (uint32_t ni : pf.kids) {
                const uint32_t tk = (pf.key >> 16) | (nodes[ni].sym << 16);
                cands.push_back({pi, ni, tk});
                ++ldeg[tk];
            }
        }
        std::stable_sort(cands.begin(), cands.end(), [&](const Cand& x, const Cand& y) { return ldeg[x.target_key] < ldeg[y.target_key]; });
        for (const Cand& c : cands) {
            auto it = prefix_of.find(c.target_key);
            uint32_t ti;
            if (it == prefix_of.end()) {   // the target record exists only to carry left children
                ti = uint32_t(prefixes.size());
                prefixes.push_back({c.target_key, nullptr, {}});
                prefix_of.emplace(c.target_key, ti);
                lefts.emplace_back(); rights.emplace_back(); overflow.emplace_back();
            } else ti = it->second;
            if (lefts[ti].size() < 3) lefts[ti].push_back({c.node, Sym(prefixes[c.parent].key & 0xFFFFu)});
            else if (rights[c.parent].size() < 3) rights[c.parent].push_back(c.node);
            else overflow[c.parent].push_back(c.node);
        }
    }

    // ---- records
    auto child_entry = [&](uint32_t* e, const Node& k, Sym sym) {
        uint32_t fl = 0;
        if (k.pat && wide(*k.pat)) { fl |= kPkWide; ++t.n_wide; }
        else if (k.pat) { e[1] = pack16(k.pat->row[0], k.pat->row[1]); e[2] = pack16(k.pat->row[2], k.pat->row[3]); }
        e[0] = sym | (fl << 16);
        e[3] = k.ref;
    };
    t.rec_bits = bits_for(prefixes.size());
    if (t.rec_bits > 24) return t;   // byte offsets of records stay below 4 GB
    const uint32_t rmask = (1u << t.rec_bits) - 1, rshift = 32 - t.rec_bits;
    t.rec.assign((size_t(1) << t.rec_bits) * 32, 0);
    // perfect hash (layout.h): one seed byte per bucket, buckets placed largest first
    std::vector<uint32_t> slot_of(prefixes.size(), 0);
    {
        t.seed_bits = 4;
        while ((size_t(3) << t.seed_bits) < prefixes.size()) ++t.seed_bits;   // about 3 keys per bucket
        const uint32_t bshift = 32 - t.seed_bits;
        t.seed.assign(size_t(1) << t.seed_bits, 0);
        std::vector<std::vector<uint32_t>> buckets(size_t(1) << t.seed_bits);
        for (uint32_t pi = 0; pi < prefixes.size(); ++pi) buckets[packed_ph_bucket(prefixes[pi].key, bshift)].push_back(pi);
        std::vector<uint32_t> order(buckets.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return buckets[x].size() > buckets[y].size(); });
        std::vector<uint8_t> used(size_t(1) << t.rec_bits, 0);
        // test hook: fewer seeds to try, so that the linear-probing fallback gets exercised
        const char* dbg = std::getenv("VPT_DEBUG_PH_SEEDS");
        const uint32_t max_seed = dbg ? std::min<uint32_t>(255u, uint32_t(std::atoi(dbg))) : 255u;
        std::vector<uint32_t> trial;
        for (uint32_t bi : order) {
            const std::vector<uint32_t>& keys = buckets[bi];
            if (keys.empty()) break;
            uint32_t seed = 0;
            for (; seed < max_seed; ++seed) {
                trial.clear();
                bool ok = true;
                for (uint32_t pi : keys) {
                    const uint32_t sl = packed_ph_slot(prefixes[pi].key, seed, rshift);
                    if (used[sl] || std::find(trial.begin(), trial.end(), sl) != trial.end()) { ok = false; break; }
                    trial.push_back(sl);
                }
                if (ok) break;
            }
            if (seed >= max_seed) seed = 255;
            t.seed[bi] = uint8_t(seed);
            if (seed < 255) {
                for (size_t j = 0; j < keys.size(); ++j) { slot_of[keys[j]] = trial[j]; used[trial[j]] = 1; }
                continue;
            }
            for (uint32_t pi : keys) {   // fallback: linear probing from the seed-255 slot, the home record says where to
                const uint32_t home = packed_ph_slot(prefixes[pi].key, 255, rshift);
                uint32_t b = home, probes = 1;
                while (used[b]) { b = (b + 1) & rmask; ++probes; }
                used[b] = 1;
                slot_of[pi] = b;
                t.max_probe = std::max(t.max_probe, probes);
                if (b != home) {
                    const uint32_t d = (b - home) & rmask;
                    t.rec[size_t(home) * 32 + 3] |= (kPkDisp | (d <= 8 ? 1u << (kPkHopShift + d - 1) : kPkFar)) << 16;
                    ++t.n_disp;
                }
            }
        }
    }
    for (uint32_t pi = 0; pi < prefixes.size(); ++pi) {
        const Prefix& pf = prefixes[pi];
        const uint32_t b = slot_of[pi];
        uint32_t* r = &t.rec[size_t(b) * 32];
        uint32_t fl = 0;
        r[16] = pf.key;
        if (pf.pat && !row_fits(pf.pat->row, kBiFieldBits)) { fl |= kPkWide; ++t.n_wide; }
        else if (pf.pat) pack_fields(r, pf.pat->row.data(), 5, kBiFieldBits);   // bits 0..109; the flags half of dword 3 may already name displaced keys
        for (size_t j = 0; j < rights[pi].size(); ++j) child_entry(r + 4 + 4 * j, nodes[rights[pi][j]], nodes[rights[pi][j]].sym);
        for (size_t j = 0; j < lefts[pi].size(); ++j) child_entry(r + 20 + 4 * j, nodes[lefts[pi][j].node], lefts[pi][j].lead);
        if (!overflow[pi].empty()) {
            fl |= kPkOv;
            const uint32_t ref = mini_alloc(t.kids3, 4, overflow[pi].size());
            uint64_t mask = 0;
            for (uint32_t ni : overflow[pi]) {
                const Node& k = nodes[ni];
                child_entry(mini_insert(t.kids3, 4, ref, k.sym), k, k.sym);
                mask |= uint64_t(1) << packed_filter_bit(k.sym);
            }
            r[17] = ref; r[18] = uint32_t(mask); r[19] = uint32_t(mask >> 32);
            t.n_overflow += uint32_t(overflow[pi].size());
        }
        r[3] |= fl << 16;
        t.n_children += uint32_t(pf.kids.size());
        t.n_left += uint32_t(lefts[pi].size());
    }
    t.n_rec = uint32_t(prefixes.size());
    t.present = true;
    return t;
}

// Type rows (layout.h, "TYPE ROWS"); empty when the n-grams do not fit the form.  `ngrams` already passed the
// window-table builder's validity checks.
std::vector<uint32_t> build_type_rows(const std::vector<NgramRecord>& ngrams, int W) {
    std::vector<int32_t> uni(8 * 6, 0), bi(64 * 6, 0), tri(512 * 6, 0);
    for (const NgramRecord& d : ngrams) {
        const int n = int(d.ngram.size());
        bool usable = true, pads = false;
        for (Sym s : d.ngram) {
            if (s > 6) usable = false;   // can never match: windows only hold codes 0..6
            if (s == 0) pads = true;     // code 0 = outside the sentence: the window table matches it against the padding
        }
        if (!usable || d.weights.empty() || n > 2 * W) continue;
        if (pads) return {};             // a start-position row cannot say "outside": the window table scores this model
        if (n > 3) return {};
        uint32_t idx = 0;
        for (int i = 0; i < n; ++i) idx |= d.ngram[size_t(i)] << (3 * i);
        int32_t* row = n == 1 ? &uni[idx * 6] : n == 2 ? &bi[idx * 6] : &tri[idx * 6];
        // the window table only ever reads w[2W - end], end = n .. 2W (boundary_scorer_cache.rs:41-46): a weight with an
        // index above 2W - n exists in no window and is ignored there, so it is ignored here
        const size_t used = std::min(d.weights.size(), size_t(2 * W - n + 1));
        for (size_t k = 0; k < used; ++k) {
            const int slot = (n - 1 - W + int(k)) + 3;   // boundary start + n-1-W+k  (type_scorer/boundary_scorer.rs:48)
            if (slot < 0 || slot > 5) return {};
            row[slot] = wadd(row[slot], d.weights[k]);
        }
    }
    std::vector<uint32_t> out(size_t(kTypeRowCount) * 4, 0);
    for (uint32_t idx = 0; idx < 512; ++idx) {
        const uint32_t t1 = idx & 7, t2 = (idx >> 3) & 7, t3 = idx >> 6;
        if (t1 == 0 || t1 == 7 || t2 == 7 || t3 == 7) continue;
        const uint32_t row = type_row_index(t1, t2, t3);
        int64_t r[6];
        for (int j = 0; j < 6; ++j) {
            r[j] = uni[t1 * 6 + j];
            if (t2 >= 1 && t2 <= 6) {
                r[j] += bi[(t1 | t2 << 3) * 6 + j];
                if (t3 >= 1 && t3 <= 6) r[j] += tri[idx * 6 + j];
            }
            if (r[j] < -131072 || r[j] > 131071) return {};   // 18-bit fields
        }
        unsigned __int128 bits = 0;
        for (int j = 0; j < 6; ++j) bits |= (unsigned __int128)(uint64_t(r[j]) & 0x3FFFFu) << (18 * j);
        for (int q = 0; q < 4; ++q) out[row * 4 + q] = uint32_t(bits >> (32 * q));
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
    uint32_t h = 0x811C9DC5u;
    for (size_t i = 0; i < n; ++i) h = (h ^ cps[i]) * 0x01000193u;
    h ^= h >> 15;
    return h * kHashMulLo;
}

namespace {
HostTagTables build_tag_tables(const ModelData& m, bool use_char, bool use_type) {
    HostTagTables t;
    t.present = true;
    t.use_char = use_char; t.use_type = use_type;
    t.n_models = uint32_t(m.tag_models.size());
    t.tok_bits = bits_for(m.tag_models.size());
    t.tok_tab.assign(size_t(1) << t.tok_bits, 0);
    const uint32_t mask = (1u << t.tok_bits) - 1;
    for (uint32_t mi = 0; mi < m.tag_models.size(); ++mi) {
        const TagModelRecord& tm = m.tag_models[mi];
        t.n_tags = std::max<uint32_t>(t.n_tags, uint32_t(tm.tags.size()));   // predictor.rs:466
        uint32_t rec[12] = {0};
        rec[0] = uint32_t(t.syms.size()); rec[1] = uint32_t(tm.token.size());
        t.syms.insert(t.syms.end(), tm.token.begin(), tm.token.end());
        auto add_ngrams = [&](const std::vector<TagNgramRecord>& list, uint32_t* first, uint32_t* count) {
            *first = uint32_t(t.ngrams.size() / 4);
            for (const TagNgramRecord& d : list) {
                const uint32_t so = uint32_t(t.syms.size());
                t.syms.insert(t.syms.end(), d.ngram.begin(), d.ngram.end());
                for (const TagWeightRecord& w : d.weights) {
                    if (d.ngram.size() >= (size_t(1) << 24)) throw ModelError("InvalidModelError: tag n-gram too long");
                    t.ngrams.push_back(so);
                    t.ngrams.push_back(uint32_t(d.ngram.size()) | (uint32_t(w.rel_position) << 24));
                    t.ngrams.push_back(uint32_t(t.weights.size()));
                    t.ngrams.push_back(uint32_t(w.weights.size()));
                    t.weights.insert(t.weights.end(), w.weights.begin(), w.weights.end());
                }
            }
            *count = uint32_t(t.ngrams.size() / 4) - *first;
        };
        add_ngrams(tm.char_ngrams, &rec[2], &rec[3]);
        add_ngrams(tm.type_ngrams, &rec[4], &rec[5]);
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
        t.models.insert(t.models.end(), rec, rec + 12);
        // token table: a repeated token keeps its slot and takes the later model
        uint32_t b = tag_token_hash(tm.token.data(), tm.token.size()) >> (32 - t.tok_bits);
        for (;;) {
            const uint32_t cur = t.tok_tab[b];
            if (cur == 0) { t.tok_tab[b] = mi + 1; break; }
            const uint32_t* cr = &t.models[size_t(cur - 1) * 12];
            if (cr[1] == tm.token.size() && std::equal(tm.token.begin(), tm.token.end(), t.syms.begin() + cr[0])) { t.tok_tab[b] = mi + 1; break; }
            b = (b + 1) & mask;
        }
    }
    if (t.syms.empty()) t.syms.push_back(0);
    if (t.weights.empty()) t.weights.push_back(0);
    if (t.ngrams.empty()) t.ngrams.assign(4, 0);
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
            if (c.packed.present) c.packed.trow = build_type_rows(m.type_ngrams, wt);
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
            if (c.packed.present) c.packed.trow = build_type_rows(merged, wt);
        } else {
            if (wt > kMaxWindow) throw ModelError("InvalidModelError: type_window_size above 8 is not supported");
            c.type_kind = kTypePatternTable;
            std::vector<Pat> pats;
            for (const auto& d : m.type_ngrams) add_ngram(pats, d.ngram, d.weights, wt, false);
            c.types = build_table(pats, wt, kUniDirectTypes);
        }
    }
    c.pad = std::max(1, std::max(c.chars.present ? wc : 0, c.type_kind != kTypeNone ? wt : 0));
    if (tags_on) c.tags = build_tag_tables(m, c.chars.present, c.type_kind != kTypeNone);
    return c;
}

}  // namespace vpt
