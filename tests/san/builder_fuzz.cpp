// builder_fuzz.cpp -- TEST INFRASTRUCTURE.  The host side of libacmi (csrc/automaton.cpp: trie insertion, BFS failure
// links, state numbering, dense rows, compact goto/failure records, flat output lists, anchor tables) compiled
// with -fsanitize=address,undefined and driven by seeded random pattern sets; the flat automaton is walked here
// exactly as the device's exact path walks it (kdev.hpp exact_step / cold_step / emit_event) and every
// record is compared with the oracle (oracle/ac_oracle.c, itself pinned to the reference's goldens).
//
//   builder_fuzz <cases> <seed>      exit status 0 = all cases identical, no sanitizer report
//
// Shapes on purpose: 1 / 2 / 255 / 256 used byte values (255 once broke the byte-class shortcut), patterns of
// 1024 and 1025 bytes, empty patterns, duplicate storms, several add calls, tiny LDS tiles, compact forced on.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/ac_oracle.h"
#include "../../php_aho_corasick_amd/csrc/automaton.hpp"

using namespace acmi;

namespace {

struct Rng {
    uint64_t s;
    uint64_t next() {  // splitmix64
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return uint32_t(next() % n); }
};

template <typename E>
uint32_t entry(const FlatAutomaton &fa, size_t row, uint32_t col) {
    return reinterpret_cast<const E *>(fa.table.data())[row * fa.n_columns + col];
}

uint32_t dense_entry(const FlatAutomaton &fa, size_t row, uint32_t col) {
    return fa.entry_bytes == 2 ? entry<uint16_t>(fa, row, col) : entry<uint32_t>(fa, row, col);
}

uint32_t tile_entry(const FlatAutomaton &fa, size_t row, uint32_t col) {
    return fa.entry_bytes == 2 ? reinterpret_cast<const uint16_t *>(fa.lds_image.data())[row * fa.n_columns + col]
                               : reinterpret_cast<const uint32_t *>(fa.lds_image.data())[row * fa.n_columns + col];
}

// kdev.hpp cold_step
uint32_t cold_step(const FlatAutomaton &fa, uint32_t s, uint32_t byte) {
    const uint32_t col = fa.class_of_byte[byte];
    if (!fa.compact) return dense_entry(fa, s, col);
    for (int guard = 0; guard < 2048; ++guard) {
        if (s < fa.dense_nf_end) return dense_entry(fa, s, col);
        if (s >= fa.f2 && s < fa.dense_f_end) return dense_entry(fa, fa.dense_nf_end + (s - fa.f2), col);
        const uint32_t x = fa.cold_rec[size_t(s) * 2], y = fa.cold_rec[size_t(s) * 2 + 1];
        const uint32_t code = y >> 24;
        if (code == 1) {
            if ((x & 0xffu) == byte) return x >> 8;
        } else if (code >= 2) {
            const uint32_t *e = fa.cold_edges.data() + x;
            uint32_t n = code;
            if (code == 255) n = *e++;
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((e[mid] & 0xffu) < byte) lo = mid + 1; else hi = mid;
            }
            if (lo < n && (e[lo] & 0xffu) == byte) return e[lo] >> 8;
        }
        s = y & 0xFFFFFFu;
    }
    std::fprintf(stderr, "failure chain does not reach a dense row\n");
    std::abort();
}

bool is_final(const FlatAutomaton &fa, uint32_t s) { return (s >= fa.f1 && s < fa.trap) || s >= fa.f2; }

// one haystack, the device's way: tile first, cold step when the tile says "trap"; records longest pattern first
void walk(const FlatAutomaton &fa, const uint8_t *text, size_t n, int find_all, uint32_t hay, std::vector<uint32_t> &out) {
    uint32_t s = 0;
    for (size_t j = 0; j < n; ++j) {
        uint32_t w = fa.trap;
        if (s < fa.trap) w = tile_entry(fa, s, fa.class_of_byte[text[j]]);
        if (w >= fa.trap) w = cold_step(fa, s, text[j]);
        s = w;
        if (!is_final(fa, s)) continue;
        const uint32_t pos = uint32_t(j + 1);
        uint32_t emitted = 0;
        for (uint32_t t = fa.own_pattern[s] != kNone ? s : fa.dict_link[s]; t != kNone; t = fa.dict_link[t]) {
            const uint32_t pat = fa.own_pattern[t];
            out.insert(out.end(), {hay, pos, pos - fa.pattern_length[pat], pat});
            if (!fa.out_off.empty()) {  // the flat list says the same
                const uint32_t o = fa.out_off[s] + emitted;
                if (fa.out_list[size_t(o) * 2] != pat || fa.out_list[size_t(o) * 2 + 1] != fa.pattern_length[pat]) {
                    std::fprintf(stderr, "flat output list disagrees with the dictionary links\n");
                    std::abort();
                }
            }
            ++emitted;
        }
        if (emitted != fa.out_count[s]) { std::fprintf(stderr, "out_count mismatch\n"); std::abort(); }
        if (!find_all) break;
    }
}

// slot of a window in the fingerprint table (as kdev.hpp verify_candidate finds it), or -1
long anchor_slot(const FlatAutomaton &fa, const uint8_t *window) {
    uint32_t words[4] = {0, 0, 0, 0};
    std::memcpy(words, window, fa.anchor_q);
    const uint32_t h = anchor_hash(words, fa.anchor_q / 4, fa.anchor_mul);
    if (fa.anchor_mode == 2) {
        const uint16_t fp = uint16_t((h >> 16) | 1u);
        const size_t bucket = (h >> fa.anchor_bucket_shift) & ((1u << fa.anchor_bucket_bits) - 1u);
        for (int i = 0; i < 8; ++i) if (fa.anchor_table[bucket * 8 + i] == fp) return long(bucket * 8 + i);
        return -1;
    }
    const uint16_t fp = uint16_t(h >> 16);
    if (fa.anchor_slots == 2) {
        const size_t bucket = (h >> 2) & (2 * kAnchorBuckets - 1);
        for (int i = 0; i < 2; ++i) if (fa.anchor_table[bucket * 2 + i] == fp) return long(bucket * 2 + i);
        return -1;
    }
    const size_t bucket = (h >> 3) & (kAnchorBuckets - 1);
    for (int i = 0; i < 4; ++i) if (fa.anchor_table[bucket * 4 + i] == fp) return long(bucket * 4 + i);
    return -1;
}

bool anchor_probe(const FlatAutomaton &fa, const uint8_t *window) {
    uint32_t words[4] = {0, 0, 0, 0};
    std::memcpy(words, window, fa.anchor_q);
    const uint32_t h = anchor_hash(words, fa.anchor_q / 4, fa.anchor_mul);
    if (fa.anchor_mode == 2) {
        if (!((fa.anchor_bitmap[h >> 17] >> ((h >> 12) & 31u)) & 1u)) return false;
        const uint16_t fp = uint16_t((h >> 16) | 1u);
        const uint16_t *b = &fa.anchor_table[size_t((h >> fa.anchor_bucket_shift) & ((1u << fa.anchor_bucket_bits) - 1u)) * 8];
        for (int i = 0; i < 8; ++i) if (b[i] == fp) return true;
        return false;
    }
    const uint16_t fp = uint16_t(h >> 16);
    if (fa.anchor_slots == 2) {
        const uint16_t *b2 = &fa.anchor_table[size_t((h >> 2) & (2 * kAnchorBuckets - 1)) * 2];
        return b2[0] == fp || b2[1] == fp;
    }
    const uint16_t *b = &fa.anchor_table[size_t((h >> 3) & (kAnchorBuckets - 1)) * 4];
    for (int i = 0; i < 4; ++i) if (b[i] == fp) return true;
    return false;
}

int run_case(uint64_t seed, bool verbose) {
    Rng r{(seed ^ 0x5851F42D4C957F2Dull) * 0xD6E8FEB86659FD93ull + 12345};  // not a multiple of the generator's own increment: seeds k and k+1 would share a stream
    // alphabet
    static const uint32_t sizes[] = {1, 2, 3, 6, 8, 9, 26, 64, 255, 256};
    const uint32_t asz = sizes[r.below(10)];
    std::vector<uint8_t> alpha;
    {
        std::vector<uint8_t> all(256);
        for (int i = 0; i < 256; ++i) all[i] = uint8_t(i);
        for (uint32_t i = 0; i < asz; ++i) { const uint32_t j = i + r.below(256 - i); std::swap(all[i], all[j]); }
        alpha.assign(all.begin(), all.begin() + asz);
    }
    const uint32_t shape = r.below(8);
    const uint32_t np = shape == 7 ? 1 + r.below(4) : 1 + r.below(shape == 6 ? 3000 : 300);
    const uint32_t lmin = shape == 3 ? 7 + r.below(12) : (shape == 4 || shape == 5) ? 4 + r.below(3) : 1 /* shapes 4, 5: 4..6 B shortest, the dense anchor form */, lmax = shape == 0 ? 4 : shape == 1 ? 16 : shape == 2 ? 64 : shape == 3 ? lmin + r.below(40) : shape == 7 ? 1024 : 12;
    const uint32_t calls = 1 + r.below(3);
    // patterns (caller order), with empties / over-long / duplicates mixed in
    std::vector<std::vector<uint8_t>> pats;
    for (uint32_t k = 0; k < np; ++k) {
        const uint32_t kind = r.below(32);
        std::vector<uint8_t> p;
        if (kind == 0) { /* empty */ }
        else if (kind == 1 && !pats.empty()) p = pats[r.below(uint32_t(pats.size()))];           // duplicate
        else if (kind == 2 && !pats.empty()) {                                                     // prefix / suffix / extension of another
            const std::vector<uint8_t> &q = pats[r.below(uint32_t(pats.size()))];
            if (q.size() > 1) { const uint32_t a = r.below(uint32_t(q.size())), b = a + 1 + r.below(uint32_t(q.size()) - a); p.assign(q.begin() + a, q.begin() + b); }
            else { p = q; p.push_back(alpha[r.below(asz)]); }
        } else if (kind == 3 && shape == 7) { p.resize(1025); for (auto &c : p) c = alpha[r.below(asz)]; }   // rejected: > 1024
        else {
            const uint32_t len = shape == 7 ? 1000 + r.below(25) : lmin + r.below(lmax - lmin + 1);
            p.resize(len);
            for (auto &c : p) c = alpha[r.below(asz)];
        }
        pats.push_back(p);
    }
    BuildOptions opt;
    static const uint32_t lds_opts[] = {2048, 4096, 16384, 160 * 1024 - 512};
    opt.lds_bytes = lds_opts[r.below(4)];
    opt.compact = int(r.below(3)) - 1;
    static const uint64_t dense_opts[] = {0, 4096, 65536, 4ull << 20};
    opt.dense_bytes = dense_opts[r.below(4)];
    opt.max_flat_outputs = r.below(4) == 0 ? 0 : (1ull << 25);
    opt.anchor_dense = int(r.below(3)) - 1;
    opt.anchor_slots2 = int(r.below(2));   // the dense anchor form (shortest pattern 4..6 B): automatic, never, whenever possible

    Automaton aut;
    orc_trie *o = orc_create();
    size_t acc_a = 0, acc_o = 0;
    const uint32_t step = (np + calls - 1) / calls;
    for (uint32_t lo = 0; lo < np; lo += step) {
        const uint32_t hi = lo + step < np ? lo + step : np;
        std::vector<const uint8_t *> vals;
        std::vector<uint32_t> lens;
        std::vector<uint8_t> blob;
        std::vector<uint64_t> off{0};
        static const uint8_t nothing = 0;
        for (uint32_t k = lo; k < hi; ++k) {
            vals.push_back(pats[k].empty() ? &nothing : pats[k].data());
            lens.push_back(uint32_t(pats[k].size()));
            blob.insert(blob.end(), pats[k].begin(), pats[k].end());
            off.push_back(blob.size());
        }
        if (blob.empty()) blob.push_back(0);
        acc_a += aut.add_batch(vals.data(), lens.data(), nullptr, nullptr, nullptr, hi - lo);
        acc_o += orc_php_add_patterns(o, blob.data(), off.data(), hi - lo, lo);
    }
    std::string err;
    if (!aut.finalize(opt, &err)) { std::fprintf(stderr, "seed %llu: finalize failed: %s\n", (unsigned long long)seed, err.c_str()); return 1; }
    orc_finalize(o);
    const FlatAutomaton &fa = aut.flat();
    int bad = 0;
    if (acc_a != acc_o || fa.n_states != orc_node_count(o) + 1) {
        std::fprintf(stderr, "seed %llu: accepted %zu vs %zu, states %u vs %zu\n", (unsigned long long)seed, acc_a, acc_o, fa.n_states, orc_node_count(o) + 1);
        bad = 1;
    }
    // anchor filter: no false negatives for any accepted pattern at offsets 0..3
    if (fa.anchor_q) {
        for (const PatternMeta &m : aut.patterns()) {
            if (!m.accepted) continue;
            for (uint32_t a = 0; a < (fa.anchor_dense ? 1u : 4u); ++a) /* dense form: one key per pattern, its first four bytes */ {
                const uint32_t off = fa.anchor_offsets[size_t(&m - aut.patterns().data()) * 4 + a];
                if (off % 4 != a || off + fa.anchor_q > m.length) { std::fprintf(stderr, "seed %llu: bad anchor offset\n", (unsigned long long)seed); bad = 1; continue; }
                if (!anchor_probe(fa, aut.pattern_bytes().data() + m.byte_off + off)) {
                    std::fprintf(stderr, "seed %llu: anchor false negative\n", (unsigned long long)seed);
                    bad = 1;
                }
            }
        }
        // verification pairs: a slot either names exactly the (pattern, offset) whose q-gram it holds, or nothing
        size_t named = 0;
        for (size_t pi = 0; pi < aut.patterns().size() && !fa.anchor_slot_pair.empty(); ++pi) {
            const PatternMeta &m = aut.patterns()[pi];
            if (!m.accepted) continue;
            for (uint32_t a = 0; a < (fa.anchor_dense ? 1u : 4u); ++a) /* dense form: one key per pattern, its first four bytes */ {
                const uint32_t off = fa.anchor_offsets[pi * 4 + a];
                const long slot = anchor_slot(fa, aut.pattern_bytes().data() + m.byte_off + off);
                if (slot < 0) { std::fprintf(stderr, "seed %llu: no slot\n", (unsigned long long)seed); bad = 1; continue; }
                const uint32_t sp = fa.anchor_slot_pair[size_t(slot)];
                if (sp == kNone) continue;
                // the slot's pair list must hold THIS pattern at THIS offset with its length, bytes and final state
                bool found = false;
                for (uint32_t k = 0; k <= (sp >> 28); ++k) {
                    const uint32_t *rec = &fa.anchor_pairs[(size_t(sp & 0x0FFFFFFFu) + k) * 4];
                    const uint32_t st = rec[3];
                    const uint32_t own = st < fa.n_states ? fa.own_pattern[st] : kNone;
                    if (own == kNone || fa.out_count[st] != rec[0] || (rec[1] >> 8) != aut.patterns()[own].length ||
                        rec[2] != aut.patterns()[own].byte_off) {
                        std::fprintf(stderr, "seed %llu: malformed verification pair\n", (unsigned long long)seed);
                        bad = 1;
                    }
                    found = found || (own == pi && (rec[1] & 0xffu) == off);
                }
                if (!found) {
                    std::fprintf(stderr, "seed %llu: verification pairs of pattern %zu offset %u are missing\n", (unsigned long long)seed, pi, off);
                    bad = 1;
                }
                ++named;
            }
        }
        (void)named;
        {
        }
    }
    // haystacks: alphabet text (sometimes with bytes outside it) with planted patterns
    for (uint32_t h = 0; h < 3 && !bad; ++h) {
        const uint32_t n = h == 0 ? r.below(40) : 200 + r.below(shape == 7 ? 6000 : 1500);
        std::vector<uint8_t> text(n);
        const bool outside = r.below(3) == 0;
        for (auto &c : text) c = (outside && r.below(16) == 0) ? uint8_t(r.below(256)) : alpha[r.below(asz)];
        for (uint32_t k = 0; k < 6 && n > 0; ++k) {
            const std::vector<uint8_t> &p = pats[r.below(np)];
            if (p.empty() || p.size() > n) continue;
            const uint32_t at = r.below(uint32_t(n - p.size() + 1));
            std::memcpy(text.data() + at, p.data(), p.size());
        }
        for (int find_all = 1; find_all >= 0; --find_all) {
            std::vector<uint32_t> got, exp(1024);
            walk(fa, text.data(), n, find_all, h, got);
            size_t ne = orc_php_match(o, text.data(), n, find_all, h, exp.data(), exp.size() / 4);
            if (ne * 4 > exp.size()) { exp.resize(ne * 4); ne = orc_php_match(o, text.data(), n, find_all, h, exp.data(), ne); }
            exp.resize(ne * 4);
            if (got != exp) {
                std::fprintf(stderr, "seed %llu hay %u find_all %d: %zu records vs %zu expected (alphabet %u, shape %u, compact %u, lds %u)\n",
                             (unsigned long long)seed, h, find_all, got.size() / 4, ne, asz, shape, fa.compact, opt.lds_bytes);
                bad = 1;
            }
        }
    }
    if (verbose) std::printf("seed %llu: alphabet %u shape %u patterns %u states %u compact %u columns %u anchor_q %u %s\n", (unsigned long long)seed, asz, shape, np,
                             fa.n_states, fa.compact, fa.n_columns, fa.anchor_q, bad ? "MISMATCH" : "ok");
    orc_release(o);
    return bad;
}

}  // namespace

int main(int argc, char **argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 200;
    const uint64_t seed0 = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    const bool verbose = argc > 3;
    int bad = 0;
    for (int c = 0; c < cases; ++c) bad += run_case(seed0 + uint64_t(c), verbose);
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
