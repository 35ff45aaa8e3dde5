// Test infrastructure (NOT part of the product library): walks the PACKED tables the host-side table compiler
// emits (vaporetto_amd/csrc/tables.cpp, layout.h "PACKED TABLES") on the CPU with exactly the lookup protocol the
// specialised HIP kernel uses (home slot, kPkDisp continuation, empty-slot stop, trie parent ids), so that the
// table compiler can be checked against the oracle without a GPU.  Built by tests/test_packed_tables.py with g++.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../vaporetto_amd/csrc/layout.h"
#include "../../vaporetto_amd/csrc/model.hpp"
#include "../../vaporetto_amd/csrc/tables.hpp"

using namespace vpt;

namespace {
inline int32_t lo16(uint32_t x) { return int32_t(x << 16) >> 16; }
inline int32_t hi16(uint32_t x) { return int32_t(x) >> 16; }
inline void add(std::vector<int32_t>& y, long b, int32_t v) {
    if (b >= 0 && b < long(y.size())) y[size_t(b)] = int32_t(uint32_t(y[size_t(b)]) + uint32_t(v));
}
// lookup in the GENERAL short table (layout.h: buckets of kShortBucket entries, kDisplacedBit protocol)
const uint32_t* general_find(const HostPatternTable& T, uint64_t key) {
    const uint32_t sb_bits = T.short_bits - 1, sb_mask = (1u << sb_bits) - 1;
    uint32_t b = hash_slot(key, 32 - sb_bits);
    const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
    bool home = true;
    for (;;) {
        bool free_slot = false, displaced = false;
        for (uint32_t j = 0; j < kShortBucket; ++j) {
            const uint32_t* e = &T.short_tab[(size_t(b) * kShortBucket + j) * T.stride_dw];
            if (e[0] == klo && (e[1] & ~kDisplacedBit) == khi) return e;
            if ((e[0] | e[1]) == 0) free_slot = true;
            if (j == 0 && (e[1] & kDisplacedBit)) displaced = true;
        }
        if (free_slot || (home && !displaced)) return nullptr;
        home = false;
        b = (b + 1) & sb_mask;
    }
}
}  // namespace

extern "C" {

struct tc_model {
    CompiledModel c;
};

int tc_create(const uint8_t* bytes, size_t len, tc_model** out) {
    try {
        ModelData m = parse_model(bytes, len, nullptr);
        tc_model* t = new tc_model();
        t->c = compile_model(m, false);
        *out = t;
        return 0;
    } catch (const ModelError&) {
        return 1;
    }
}
void tc_destroy(tc_model* t) { delete t; }
int tc_packed_present(const tc_model* t) { return t->c.packed.present ? 1 : 0; }
void tc_stats(const tc_model* t, uint32_t out[8]) {
    const HostPackedTable& k = t->c.packed;
    out[0] = k.n_bi; out[1] = k.n_tri; out[2] = k.n_edge; out[3] = k.n_disp_bi; out[4] = k.n_disp_tri; out[5] = k.n_disp_edge;
    out[6] = k.max_probe; out[7] = k.n_wide;
}

// char-pattern part of the boundary scores of one sentence (n code points) + bias; y has n-1 entries.
// probes[0..2] count continued bigram / trigram / edge lookups (diagnostics).
int tc_score_chars(const tc_model* t, const uint32_t* cps, size_t n, int32_t* y_out, uint64_t probes[3]) {
    const HostPackedTable& K = t->c.packed;
    if (!K.present) return -1;
    std::vector<int32_t> y(n > 0 ? n - 1 : 0, t->c.bias);
    std::vector<uint32_t> sym(n + 3, 0);
    for (size_t i = 0; i < n; ++i) sym[i] = cps[i] < kPackedNoMatchSym ? cps[i] : kPackedNoMatchSym;
    const uint32_t bi_mask = (1u << K.bi_bits) - 1, tri_mask = (1u << K.tri_bits) - 1, edge_mask = (1u << K.edge_bits) - 1;
    for (size_t s = 0; s < n; ++s) {
        const uint32_t c1 = sym[s], c2 = sym[s + 1], c3 = sym[s + 2];
        const long S = long(s);
        const uint32_t* u = &K.uni[size_t(c1) * 4];
        add(y, S - 3, lo16(u[0])); add(y, S - 2, hi16(u[0])); add(y, S - 1, lo16(u[1]));
        add(y, S, hi16(u[1])); add(y, S + 1, lo16(u[2])); add(y, S + 2, hi16(u[2]));
        const HostPatternTable& G = t->c.chars;
        if (u[3] == kPkWide) {
            const uint32_t* g = &G.uni[size_t(c1) * G.uni_dw];
            for (int j = 0; j < 6; ++j) add(y, S - 3 + j, int32_t(g[j]));
        }
        if (c2 == 0) continue;
        const uint32_t kb = c1 | (c2 << 16);
        {
            uint32_t b = packed_hash1(kb, 32 - K.bi_bits);
            const uint32_t* e = &K.bi[size_t(b) * 4];
            bool hit = e[0] == kb;
            if (!hit && (e[3] & (kPkDisp << 16))) {
                ++probes[0];
                for (;;) {
                    b = (b + 1) & bi_mask;
                    e = &K.bi[size_t(b) * 4];
                    if (e[0] == kb) { hit = true; break; }
                    if (e[0] == 0) break;
                }
            }
            if (hit && (e[3] & (kPkWide << 16))) {
                const uint32_t* g = general_find(G, short_key(c1, c2, 0));
                if (!g) return -2;
                for (int j = 0; j < 5; ++j) add(y, S - 2 + j, int32_t(g[2 + j]));
            } else if (hit) {
                add(y, S - 2, lo16(e[1])); add(y, S - 1, hi16(e[1])); add(y, S, lo16(e[2]));
                add(y, S + 1, hi16(e[2])); add(y, S + 2, lo16(e[3]));
            }
        }
        if (c3 == 0) continue;
        uint32_t b = packed_hash2(kb, c3, 32 - K.tri_bits);
        const uint32_t* e = &K.tri[size_t(b) * 4];
        bool hit = e[0] == kb && (e[1] & 0xFFFFu) == c3;
        if (!hit && (e[1] & (kPkDisp << 16))) {
            ++probes[1];
            for (;;) {
                b = (b + 1) & tri_mask;
                e = &K.tri[size_t(b) * 4];
                if (e[0] == kb && (e[1] & 0xFFFFu) == c3) { hit = true; break; }
                if (e[0] == 0) break;
            }
        }
        if (!hit) continue;
        if (e[1] & (kPkWide << 16)) {
            const uint32_t* g = general_find(G, short_key(c1, c2, c3));
            if (!g) return -2;
            for (int j = 0; j < 4; ++j) add(y, S - 1 + j, int32_t(g[2 + j]));
        } else { add(y, S - 1, lo16(e[2])); add(y, S, hi16(e[2])); add(y, S + 1, lo16(e[3])); add(y, S + 2, hi16(e[3])); }
        if (!(e[1] & (kPkHasKids << 16))) continue;
        uint32_t parent = b, depth = 3;
        for (;;) {
            const uint32_t c = sym[s + depth < n ? s + depth : n];
            if (c == 0) break;
            uint32_t eb = packed_hash2(parent, c, 32 - K.edge_bits);
            const uint32_t* ed = nullptr;
            bool home = true, found = false;
            for (;;) {
                ed = &K.edge[size_t(eb) * 4];
                if (ed[0] == parent && (ed[1] & 0xFFFFu) == c) { found = true; break; }
                if (ed[1] == 0 || (home && !(ed[1] & (kPkDisp << 16)))) break;
                if (home) ++probes[2];
                home = false;
                eb = (eb + 1) & edge_mask;
            }
            if (!found) break;
            const uint32_t m = depth + 1;
            if (ed[1] & (kPkHasRow << 16)) {
                const uint32_t* w = &K.wrows[size_t(ed[2]) * 4];
                for (uint32_t j = 0; j <= m; ++j)
                    add(y, S - 1 + long(j), (ed[1] & (kPkWide << 16)) ? int32_t(w[j]) : (j & 1) ? hi16(w[j >> 1]) : lo16(w[j >> 1]));
            }
            if (!(ed[1] & (kPkHasKids << 16))) break;
            parent = kPackedEdgeId | eb;
            depth = m;
        }
    }
    std::memcpy(y_out, y.data(), y.size() * sizeof(int32_t));
    return 0;
}

}  // extern "C"
