// Test infrastructure (NOT part of the product library): walks the PACKED tables the host-side table compiler
// emits (vaporetto_amd/csrc/tables.cpp, layout.h "PACKED TABLES") on the CPU with exactly the lookup protocol the
// specialised HIP kernel uses (ids, unigram base + id -> bigram node with its key, filter, bigram base + id -> trigram
// node with its parent, mini-tables below), so that the table compiler can be checked against the oracle without a GPU.
// Built by tests/test_packed_tables.py with g++.
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
int tc_create_tags(const uint8_t* bytes, size_t len, int predict_tags, tc_model** out) {
    try {
        ModelData m = parse_model(bytes, len, nullptr);
        tc_model* t = new tc_model();
        t->c = compile_model(m, predict_tags != 0);
        *out = t;
        return 0;
    } catch (const ModelError&) {
        return 1;
    }
}
void tc_destroy(tc_model* t) { delete t; }
uint32_t tc_fullwidth(uint32_t c) { return kytea_fullwidth_host(c); }
int tc_packed_present(const tc_model* t) { return t->c.packed.present ? 1 : 0; }
int tc_trow_present(const tc_model* t) { return t->c.packed.present ? int(t->c.packed.trow_mode) : 0; }
int tc_row_window(const tc_model* t) { return t->c.packed.present ? t->c.packed.wl : 0; }
void tc_stats(const tc_model* t, uint32_t out[8]) {
    const HostPackedTable& k = t->c.packed;
    out[0] = k.n_bi; out[1] = k.n_tri; out[2] = uint32_t(k.bi.size() / size_t(pk_bi_dw(k.wl))); out[3] = k.n_deep; out[4] = uint32_t(k.tri.size() / size_t(pk_tri_dw(k.wl)));
    out[5] = k.bi_shift; out[6] = k.n_wide; out[7] = k.n_alpha;
}

namespace {
// mini-table probe (layout.h): returns the entry or nullptr
const uint32_t* mini_find(const std::vector<uint32_t>& arena, uint32_t dw, uint32_t ref, uint32_t sym, uint64_t* steps) {
    const uint32_t size = 1u << (ref & 31u), base = ref >> 5;
    uint32_t i = packed_mini_slot(sym, ref);
    for (uint32_t n = 0; n < size; ++n) {
        const uint32_t* e = &arena[(size_t(base) + i) * dw];
        ++*steps;
        if ((e[0] & 0xFFFFu) == sym) return e;
        if (e[0] == 0) return nullptr;
        i = (i + 1) & (size - 1);
    }
    return nullptr;
}
}  // namespace

// Boundary scores of one sentence (n code points) from the packed tables: bias + char patterns (+ type rows when
// the model has them in the packed form; returns 2 then, 1 when types are NOT included, < 0 on error).
// probes[0..3] count bigram nodes read / trigram nodes read / deep probes / filter rejections.
int tc_score(const tc_model* t, const uint32_t* cps, size_t n, int32_t* y_out, uint64_t probes[4]) {
    const HostPackedTable& K = t->c.packed;
    const HostPatternTable& G = t->c.chars;
    if (!K.present) return -1;
    const int wl = K.wl;
    const size_t udw = size_t(pk_uni_dw(wl)), bdw = size_t(pk_bi_dw(wl)), tdw = size_t(pk_tri_dw(wl));
    const int nu = pk_uni_fields(wl), nb = pk_bi_fields(wl), nt = pk_tri_fields(wl);
    std::vector<int32_t> y(n > 0 ? n - 1 : 0, t->c.bias);
    std::vector<uint32_t> sym(n + 8, 0), typ(n + 8, 0);
    for (size_t i = 0; i < n; ++i) {
        sym[i] = K.id_for(cps[i]);
        typ[i] = char_type_host(cps[i]);
    }
    const uint32_t uni_last = uint32_t(K.uni.size() / udw) - 1, n_tri = uint32_t(K.tri.size() / tdw);
    auto cp_of = [&](uint32_t id) { return K.cpid[id < uni_last ? id : uni_last]; };
    for (size_t s = 0; s < n; ++s) {
        const uint32_t c1 = sym[s], c2 = sym[s + 1], c3 = sym[s + 2];
        const long S = long(s);
        if (K.trow_mode == kTypeRowsLds) {
            const uint32_t* r = &K.trow[size_t(type_row_index(typ[s], typ[s + 1], typ[s + 2])) * size_t(pk_trow_dw(wl))];
            for (int j = 0; j < nu; ++j) add(y, S - wl + j, bits_signed(r, kUniFieldBits * j, kUniFieldBits));
        } else if (K.trow_mode == kTypeRowsGlobal) {
            uint32_t idx = 0;
            for (int i = int(K.trow_levels) - 1; i >= 1; --i) idx = idx * 7u + typ[s + size_t(i)];
            idx = idx * 6u + (typ[s] - 1u);
            const uint32_t* r = &K.trow[size_t(idx) * size_t(pk_trow_global_dw(wl))];
            for (int j = 0; j < nu; ++j) add(y, S - wl + j, int32_t(r[j]));
        }
        const uint32_t* u = &K.uni[size_t(c1 < uni_last ? c1 : uni_last) * udw];
        for (int j = 0; j < nu; ++j) add(y, S - wl + j, bits_signed(u, kUniFieldBits * j, kUniFieldBits));
        if (bits_unsigned(u, pk_uni_base_bit(wl) + kUniBaseBits, 1)) {
            const uint32_t* g = cp_of(c1) < G.uni_n ? &G.uni[size_t(cp_of(c1)) * G.uni_dw] : general_find(G, short_key(cp_of(c1), 0, 0));
            if (!g) return -2;
            if (cp_of(c1) >= G.uni_n) g += 2;
            for (int j = 0; j < G.len[0]; ++j) add(y, S + G.lo[0] + j, int32_t(g[j]));
        }
        if (c2 == 0 || c1 == kNoId || c2 == kNoId) continue;   // (the kernel issues no load for a char outside the alphabet)
        const uint32_t slot = (bits_unsigned(u, pk_uni_base_bit(wl), kUniBaseBits) << K.bi_shift) + c2;
        if (size_t(slot) * bdw + bdw > K.bi.size()) return -3;   // the table must cover any id behind any base
        const uint32_t* r = &K.bi[size_t(slot) * bdw];
        ++probes[0];
        if (r[pk_bi_key_dw(wl)] != (c1 | (c2 << 16))) continue;
        const uint32_t* rrow = r + pk_bi_row_dw(wl);
        if (bits_unsigned(rrow, pk_bi_wide_bit(wl), 1)) {
            const uint32_t* g = general_find(G, short_key(cp_of(c1), cp_of(c2), 0));
            if (!g) return -2;
            for (int j = 0; j < G.len[1]; ++j) add(y, S + G.lo[1] + j, int32_t(g[2 + j]));
        } else {
            for (int j = 0; j < nb; ++j) add(y, S - wl + 1 + j, bits_signed(rrow, kBiFieldBits * j, kBiFieldBits));
        }
        if (c3 == 0) continue;
        const uint64_t mask = uint64_t(r[pk_bi_filter_dw(wl)]) | (uint64_t(r[pk_bi_filter_dw(wl) + 1]) << 32);
        if (!((mask >> packed_filter_bit(c3)) & 1)) { ++probes[3]; continue; }
        const uint32_t ts = r[pk_bi_base_dw(wl)] + c3;   // modulo 2^32
        if (ts >= n_tri) continue;
        const uint32_t* ch = &K.tri[size_t(ts) * tdw];
        ++probes[1];
        if ((ch[0] & kTriParentMask) != slot + 1) continue;
        if (ch[0] & (kPkWide << kTriFlagShift)) {
            const uint32_t* g = general_find(G, short_key(cp_of(c1), cp_of(c2), cp_of(c3)));
            if (!g) return -2;
            for (int j = 0; j < G.len[2]; ++j) add(y, S + G.lo[2] + j, int32_t(g[2 + j]));
        } else {
            const uint32_t* w = ch + pk_tri_w_dw(wl);
            for (int j = 0; j < nt; ++j) add(y, S - wl + 2 + j, (j & 1) ? hi16(w[j >> 1]) : lo16(w[j >> 1]));
        }
        uint32_t ref = ch[pk_tri_kids_dw(wl)] & kTriKidsRefMask, depth = 3;
        if (ref != 0) {   // the node's child filter (layout.h, "tri"), as the kernel applies it
            const uint32_t c4 = sym[s + 3 < n ? s + 3 : n];
            if (c4 == 0 || c4 == kNoId || !((tri_kid_filter(ch[0], ch[pk_tri_kids_dw(wl)]) >> packed_kid_filter_bit(c4)) & 1u)) ref = 0;
        }
        while (ref != 0) {
            const size_t at = s + depth;
            const uint32_t c = sym[at < n ? at : n];
            if (c == 0) break;
            const uint32_t* e = mini_find(K.deep, 16, ref, c, &probes[2]);
            if (!e) break;
            const uint32_t nskip = (e[0] >> 24) & 15u;
            bool ok = true;
            for (uint32_t j = 0; j < nskip; ++j) {
                const size_t q = at + 1 + j;
                const uint32_t want = (e[2 + j / 2] >> (16 * (j & 1))) & 0xFFFFu;
                if ((q < n ? sym[q] : 0u) != want) ok = false;
            }
            if (!ok) break;
            const uint32_t m = depth + 1 + nskip;
            const long lo = row_lo(int(m), wl);
            const uint32_t rlen = uint32_t(row_len(int(m), wl));
            if (e[0] & (kPkHasRow << 16)) {
                for (uint32_t j = 0; j < rlen; ++j) add(y, S + lo + long(j), (j & 1) ? hi16(e[8 + (j >> 1)]) : lo16(e[8 + (j >> 1)]));
            } else if (e[0] & (kPkExtRow << 16)) {
                for (uint32_t j = 0; j < rlen; ++j) add(y, S + lo + long(j), K.xrows[size_t(e[8]) + j]);
            }
            ref = e[1];
            depth = m;
        }
    }
    std::memcpy(y_out, y.data(), y.size() * sizeof(int32_t));
    return K.trow_mode == kTypeRowsNone ? 1 : 2;
}

}  // extern "C"
