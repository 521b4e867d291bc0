// automaton.cpp -- see automaton.hpp.
#include "automaton.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace acmi {

namespace {
inline uint64_t mix(uint64_t k) {
    k *= 0x9E3779B97F4A7C15ull;
    return k ^ (k >> 29);
}

// Sort `v` by a 64-bit key (LSD radix sort of {key, index} pairs, 11 bits a pass, passes in which every key has the same
// digit skipped), elements with equal keys ordered by `tie`: the sorts of a large pattern set's q-gram triples -- 24-byte
// records under a three-level comparison -- were a third of its anchor-table build.
template <typename T, typename Key, typename Tie>
void sort_by_u64(std::vector<T> &v, Key key, Tie tie) {
    const size_t n = v.size();
    if (n < 4096) {
        std::sort(v.begin(), v.end(), [&](const T &a, const T &b) { return key(a) != key(b) ? key(a) < key(b) : tie(a, b); });
        return;
    }
    struct KI { uint64_t k; uint32_t i; };
    std::vector<KI> a(n), b(n);
    uint64_t all_or = 0, all_and = ~0ull;
    for (size_t i = 0; i < n; ++i) {
        a[i] = KI{key(v[i]), uint32_t(i)};
        all_or |= a[i].k;
        all_and &= a[i].k;
    }
    const uint64_t varying = all_or ^ all_and;   // bits that differ between some two keys
    for (uint32_t shift = 0; shift < 64; shift += 11) {
        if (((varying >> shift) & 0x7FFu) == 0) continue;
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; ++i) ++count[((a[i].k >> shift) & 0x7FFu) + 1];
        for (int d = 0; d < 2048; ++d) count[d + 1] += count[d];
        for (size_t i = 0; i < n; ++i) b[count[(a[i].k >> shift) & 0x7FFu]++] = a[i];
        a.swap(b);
    }
    std::vector<T> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = v[a[i].i];
    for (size_t i = 0; i < n;) {
        size_t e = i + 1;
        while (e < n && a[e].k == a[i].k) ++e;
        if (e - i > 1) std::sort(out.begin() + i, out.begin() + e, tie);
        i = e;
    }
    v.swap(out);
}
// Dense rows.  `dense` = how many states, in BFS order, get a row (all of them when not compact); a state's row
// index is its id below dense_nf_end and dense_nf_end + id - f2 for the dense cold final ones.  Trie arrays by BFS
// position (Automaton::finalize): children of position i = positions [cfirst[i], cfirst[i+1]).
template <typename E>
void build_tables(FlatAutomaton &fa, uint32_t dense, const std::vector<uint32_t> &fail_b, const std::vector<uint32_t> &newid,
                  const std::vector<uint32_t> &cfirst, const std::vector<uint8_t> &byte_b) {
    const size_t nc = fa.n_columns;
    fa.table.assign(size_t(fa.n_dense_rows) * nc * sizeof(E), 0);
    fa.table_bytes = fa.table.size();
    E *tab = reinterpret_cast<E *>(fa.table.data());
    auto row_of = [&](uint32_t id) { return size_t(id < fa.dense_nf_end ? id : fa.dense_nf_end + (id - fa.f2)); };
    for (uint32_t i = 0; i < dense; ++i) {  // BFS order: the failure state's row is already complete
        E *row = tab + row_of(newid[i]) * nc;
        if (i != 0)
            std::memcpy(row, tab + row_of(newid[fail_b[i]]) * nc, nc * sizeof(E));
        // (root row stays all-zero: every byte without a goto edge returns to the root)
        for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; ++c) row[fa.class_of_byte[byte_b[c]]] = E(newid[c]);
    }
    E *trap_row = tab + size_t(fa.trap) * nc;
    for (size_t j = 0; j < nc; ++j) trap_row[j] = E(fa.trap);

    // LDS tile: hot rows only; a transition that leaves the tile lands on the trap row
    const size_t hot_rows = size_t(fa.trap) + 1;
    fa.lds_image.resize(hot_rows * nc * sizeof(E));
    fa.lds_image_bytes = fa.lds_image.size();
    E *img = reinterpret_cast<E *>(fa.lds_image.data());
    for (size_t i = 0; i < hot_rows * nc; ++i) img[i] = tab[i] > E(fa.trap) ? E(fa.trap) : tab[i];
}
}  // namespace

Automaton::Automaton() {
    parent_.push_back(kNone);
    in_byte_.push_back(0);
    own_.push_back(kNone);
    rehash(1024);
}

void Automaton::rehash(size_t new_cap) {
    std::vector<uint64_t> keys(new_cap, 0);
    std::vector<uint32_t> vals(new_cap, 0);
    const size_t mask = new_cap - 1;
    for (size_t i = 0; i < map_keys_.size(); ++i) {
        if (!map_keys_[i]) continue;
        size_t s = mix(map_keys_[i]) & mask;
        while (keys[s]) s = (s + 1) & mask;
        keys[s] = map_keys_[i];
        vals[s] = map_vals_[i];
    }
    map_keys_.swap(keys);
    map_vals_.swap(vals);
}

uint32_t Automaton::find_child(uint32_t node, uint8_t byte) const {
    const uint64_t key = ((uint64_t(node) << 8) | byte) + 1;
    const size_t mask = map_keys_.size() - 1;
    size_t s = mix(key) & mask;
    while (map_keys_[s]) {
        if (map_keys_[s] == key) return map_vals_[s];
        s = (s + 1) & mask;
    }
    return kNone;
}

uint32_t Automaton::add_child(uint32_t node, uint8_t byte) {
    if ((map_used_ + 1) * 2 > map_keys_.size()) rehash(map_keys_.size() * 2);
    const uint32_t child = uint32_t(parent_.size());
    parent_.push_back(node);
    in_byte_.push_back(byte);
    own_.push_back(kNone);
    const uint64_t key = ((uint64_t(node) << 8) | byte) + 1;
    const size_t mask = map_keys_.size() - 1;
    size_t s = mix(key) & mask;
    while (map_keys_[s]) s = (s + 1) & mask;
    map_keys_[s] = key;
    map_vals_[s] = child;
    ++map_used_;
    return child;
}

void Automaton::set_replacement(uint32_t index, const uint8_t *text, uint32_t length) {
    if (rep_off_.size() <= index) rep_off_.resize(size_t(index) + 1, ~0ull);
    if (!text) { rep_off_[index] = ~0ull; return; }
    const uint64_t at = rep_bytes_.size();
    uint32_t n = 0;
    while (n < length && text[n] != 0) ++n;          // strncpy: up to the first NUL ...
    rep_bytes_.insert(rep_bytes_.end(), text, text + n);
    rep_bytes_.insert(rep_bytes_.end(), length - n, uint8_t(0));   // ... zero-filled to the caller's length
    rep_off_[index] = (at << 32) | length;
}

bool Automaton::has_replacement() const {
    for (size_t i = 0; i < patterns_.size() && i < rep_off_.size(); ++i)
        if (patterns_[i].accepted && rep_off_[i] != ~0ull) return true;
    return false;
}

size_t Automaton::add_batch(const uint8_t *const *values, const uint32_t *lengths,
                            const uint32_t *id_types, const int64_t *id_numbers,
                            const uint64_t *users, size_t n) {
    const uint32_t base = uint32_t(patterns_.size());
    for (size_t k = 0; k < n; ++k) {
        PatternMeta m;
        m.byte_off = pattern_bytes_.size();
        m.length = lengths[k];
        m.id_type = id_types ? id_types[k] : 0;
        m.id_number = id_numbers ? id_numbers[k] : 0;
        m.user = users ? users[k] : 0;
        m.accepted = false;
        if (lengths[k]) pattern_bytes_.insert(pattern_bytes_.end(), values[k], values[k] + lengths[k]);
        patterns_.push_back(m);
    }
    size_t accepted = 0;
    {   // one rehash for the whole batch instead of one per doubling: every byte is at most one new node
        size_t bytes = 0;
        for (size_t k = 0; k < n; ++k)
            if (lengths[k] <= kMaxPatternLength) bytes += lengths[k];
        size_t cap = map_keys_.size();
        while ((map_used_ + bytes + 1) * 2 > cap) cap *= 2;
        if (cap != map_keys_.size()) rehash(cap);
    }
    for (size_t k = n; k-- > 0;) {  // reverse caller order (php_ahocorasick.c:410-421,457-486)
        const uint32_t len = lengths[k];
        if (len == 0 || len > kMaxPatternLength) continue;  // ahocorasick.c:101-105
        const uint8_t *p = values[k];
        uint32_t node = 0;
        uint32_t i = 0;
        for (; i < len; ++i) {
            const uint32_t nx = find_child(node, p[i]);
            if (nx == kNone) break;
            node = nx;
        }
        if (i < len) {
            // Below the first node this pattern creates there is nothing to find, and the ids of the nodes to come are
            // known (consecutive): all their map slots are requested at once -- an insert into a map of 10^7 slots is
            // a cache miss, and one after the other they were most of the time of a large add.
            const uint32_t first_new = uint32_t(parent_.size());
            const size_t mask = map_keys_.size() - 1;
            for (uint32_t j = i; j < len; ++j) {
                const uint32_t from = j == i ? node : first_new + (j - i - 1);
                const size_t s = mix(((uint64_t(from) << 8) | p[j]) + 1) & mask;
                __builtin_prefetch(&map_keys_[s], 1);
                __builtin_prefetch(&map_vals_[s], 1);
            }
            for (; i < len; ++i) node = add_child(node, p[i]);
        }
        if (own_[node] != kNone) continue;  // duplicate: first inserted wins (ahocorasick.c:123-124)
        own_[node] = base + uint32_t(k);
        patterns_[base + k].accepted = true;
        ++accepted;
    }
    accepted_ += accepted;
    return accepted;
}

// ---- chain-compressed image (automaton.hpp ChainImage) ----------------------------------------------------------------------
// All arguments by BFS position (finalize): children of i = [cfirst[i], cfirst[i+1]), in byte `byte_b`; failure position; final
// flag; records of the state's event; newid = the state id of the other tables.
static void build_chain_image(FlatAutomaton &fa, uint32_t budget, uint32_t ns, const std::vector<uint32_t> &cfirst, const std::vector<uint32_t> &parent_pos,
                              const std::vector<uint8_t> &byte_b, const std::vector<uint32_t> &fail_b, const std::vector<uint8_t> &final_b,
                              const std::vector<uint32_t> &cnt_b, const std::vector<uint32_t> &newid) {
    ChainImage &ci = fa.chain;
    ci = ChainImage{};
    if (budget == 0 || ns > 32767u || fa.n_columns > 127u || fa.out_head.empty()) return;   // (a record's key: twice the class, in a byte)
    const uint32_t row_bytes = fa.n_columns * 2u;
    if (uint64_t(ns) * 4u > budget) return;
    // D: branching states and the root; closed under "failure state of a state outside D" and "parent of a state inside D"
    // (a failure state and a parent lie before their state in BFS order: one backward pass closes both)
    std::vector<uint8_t> in_d(ns, 0);
    in_d[0] = 1;
    if (uint64_t(ns) * (row_bytes + 1u) + 4u <= budget) {
        // (every state's row fits: all of them D states -- a step is then ONE read, k_walk_chain's ALLD form)
        std::fill(in_d.begin(), in_d.end(), uint8_t(1));
    } else {
        for (uint32_t i = 0; i < ns; ++i)
            if (cfirst[i + 1] - cfirst[i] >= 2u) in_d[i] = 1;
        for (uint32_t i = ns; i-- > 1;) {
            if (in_d[i]) in_d[parent_pos[i]] = 1;
            else in_d[fail_b[i]] = 1;
        }
    }
    // (position i is only ever marked from positions behind it, all of them visited before it: its mark is final when it is read)
    uint32_t n_dense = 0;
    for (uint32_t i = 0; i < ns; ++i) n_dense += in_d[i];
    const uint64_t rows_bytes = (uint64_t(n_dense) * row_bytes + 3u) & ~uint64_t(3);
    if (rows_bytes + uint64_t(ns - n_dense) * 4u > budget) return;
    // indices: D in BFS order; then the chains (a C state whose parent is D, followed along its only children), final heads first
    std::vector<uint32_t> idx(ns, 0xFFFFFFFFu);
    uint32_t next = 0;
    for (uint32_t i = 0; i < ns; ++i)
        if (in_d[i]) idx[i] = next++;
    uint32_t cnt_n = n_dense;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t i = 1; i < ns; ++i) {
            if (in_d[i] || !in_d[parent_pos[i]] || (final_b[i] != 0) != (pass == 0)) continue;
            for (uint32_t j = i;; j = cfirst[j]) {
                idx[j] = next++;
                if (cfirst[j + 1] == cfirst[j]) break;
            }
        }
        if (pass == 0) cnt_n = next;
    }
    if (next != ns) return;   // (cannot happen: every C state hangs below a D state by a chain of C states)
    bool any_row_final = false;   // does any row lead to a final state at all?  (no: the count bytes are not needed)
    for (uint32_t i = 1; i < ns && !any_row_final; ++i)
        if (final_b[i] && in_d[parent_pos[i]]) any_row_final = true;
    if (!any_row_final) cnt_n = 0;
    if (n_dense == ns) cnt_n = ns;   // (all rows: every final state is entered through a row)
    const uint64_t rec_off = rows_bytes, cnt_off = rec_off + uint64_t(ns - n_dense) * 4u;
    const uint64_t total = (cnt_off + cnt_n + 3u) & ~uint64_t(3);
    if (total > budget) return;
    ci.n_dense = n_dense;
    ci.n_index = ns;
    ci.row_bytes = row_bytes;
    ci.rec_off = uint32_t(rec_off);
    ci.cnt_off = uint32_t(cnt_off);
    ci.cnt_n = cnt_n;
    ci.image.assign(size_t(total), 0);
    ci.head.assign(size_t(ns) * 2, 0u);
    ci.orig.assign(ns, 0u);
    // class -> the byte that has it (the "other" class: none)
    int byte_of_class[512];
    for (uint32_t c = 0; c < fa.n_columns; ++c) byte_of_class[c] = -1;
    {
        bool used[256] = {false};
        for (uint32_t v = 1; v < ns; ++v) used[byte_b[v]] = true;
        for (int b = 0; b < 256; ++b)
            if (used[b]) byte_of_class[fa.class_of_byte[b]] = b;
    }
    auto child_by = [&](uint32_t i, int b) -> uint32_t {
        for (uint32_t v = cfirst[i]; v < cfirst[i + 1]; ++v)
            if (byte_b[v] == b) return v;
        return 0xFFFFFFFFu;
    };
    uint16_t *rows = reinterpret_cast<uint16_t *>(ci.image.data());
    uint32_t *recs = reinterpret_cast<uint32_t *>(ci.image.data() + rec_off);
    uint8_t *cnt = ci.image.data() + cnt_off;
    for (uint32_t i = 0; i < ns; ++i) {
        const uint32_t k = idx[i];
        ci.orig[k] = newid[i];
        ci.head[size_t(k) * 2] = fa.out_head[size_t(newid[i]) * 2];
        ci.head[size_t(k) * 2 + 1] = fa.out_head[size_t(newid[i]) * 2 + 1];
        if (k < cnt_n) cnt[k] = uint8_t(std::min<uint32_t>(cnt_b[i], 255u));
        if (in_d[i]) {
            for (uint32_t c = 0; c < fa.n_columns; ++c) {
                uint32_t t = 0;   // the root
                const int b = byte_of_class[c];
                if (b >= 0) {
                    for (uint32_t f = i;; f = fail_b[f]) {   // ahocorasick.c:199-212, folded into the row
                        const uint32_t v = child_by(f, b);
                        if (v != 0xFFFFFFFFu) { t = v; break; }
                        if (f == 0) break;
                    }
                }
                rows[size_t(k) * fa.n_columns + c] = uint16_t(idx[t] | (final_b[t] ? 0x8000u : 0u));
            }
        } else {
            uint32_t key = 0xFFu, code = 0;
            if (cfirst[i + 1] != cfirst[i]) {
                const uint32_t v = cfirst[i];
                key = 2u * fa.class_of_byte[byte_b[v]];
                if (final_b[v]) code = 0x80u | std::min<uint32_t>(cnt_b[v], 31u);
            }
            recs[k - n_dense] = idx[fail_b[i]] | key << 16 | code << 24;
        }
    }
    // final indices by rank (every index's head is in `head`; the ranked copy is what fits an LDS next to nothing else)
    ci.frank.assign(ns, uint16_t(0xFFFF));
    ci.fhead.clear();
    for (uint32_t k = 0; k < ns; ++k) {
        if ((ci.head[size_t(k) * 2] >> 24) == 0u) continue;
        ci.frank[k] = uint16_t(ci.fhead.size() / 2);
        ci.fhead.push_back(ci.head[size_t(k) * 2]);
        ci.fhead.push_back(ci.head[size_t(k) * 2 + 1]);
    }
}

bool Automaton::finalize(const BuildOptions &opt, std::string *error, const BuildAccel *accel) {
    if (finalized_) return true;
    // ACMI_BUILD_TRACE=1: phase times on stderr
    const bool trace = std::getenv("ACMI_BUILD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[acmi build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const uint32_t ns = uint32_t(parent_.size());
    FlatAutomaton &fa = flat_;

    // ---- children in CSR form, BFS order ------------------------------------------------
    // From here on a node is named by its POSITION in BFS order (order[i] = the node at position i): the children of
    // position i are the consecutive positions [cfirst[i], cfirst[i+1]), a parent lies before its children, and so does
    // a failure target.  Every later pass over the trie then reads its arrays front to back -- indexed by node id (the
    // order the patterns were added in) each of them was a pass of cache misses over millions of nodes.
    std::vector<uint32_t> order(ns), cfirst(size_t(ns) + 1), parent_pos(ns, 0);
    std::vector<uint8_t> depth(ns, 0), byte_b(ns, 0);
    {
        std::vector<uint32_t> child_off(size_t(ns) + 1, 0), child_list(ns > 1 ? ns - 1 : 0);
        for (uint32_t v = 1; v < ns; ++v) ++child_off[parent_[v] + 1];
        for (uint32_t v = 0; v < ns; ++v) child_off[v + 1] += child_off[v];
        {
            std::vector<uint32_t> cur(child_off.begin(), child_off.end() - 1);
            for (uint32_t v = 1; v < ns; ++v) child_list[cur[parent_[v]]++] = v;
        }
        order[0] = 0;
        uint32_t tail = 1;
        for (uint32_t head = 0; head < ns; ++head) {
            const uint32_t u = order[head];
            cfirst[head] = tail;
            const uint8_t d = uint8_t(std::min<uint32_t>(depth[head] + 1u, 255u));
            for (uint32_t e = child_off[u]; e < child_off[u + 1]; ++e) {
                const uint32_t v = child_list[e];
                order[tail] = v;
                depth[tail] = d;
                parent_pos[tail] = head;
                byte_b[tail] = in_byte_[v];
                ++tail;
            }
        }
        cfirst[ns] = ns;
    }

    // ---- failure links (textbook BFS; same function as ahocorasick.c:344-368 defines) and per-state output:
    // final flag, record count, dictionary link (node.c:424-441).  Level-synchronous: on the device when the engine
    // offers it (one launch per trie depth), else here.  All by BFS position, the links as positions.
    std::vector<uint32_t> fail_b(ns, 0);
    std::vector<uint8_t> final_b(ns, 0);
    std::vector<uint32_t> cnt_b(ns, 0), dict_b(ns, kNone);
    bool done = false;
    if (accel && accel->fail_and_outputs) {
        // level d + 1 = the children of level d: it begins where level d ends and ends where the children of the first
        // position behind level d begin
        std::vector<uint32_t> level_off{0, 1};
        while (level_off.back() < ns) level_off.push_back(cfirst[level_off.back()]);
        TrieView tv{ns, parent_.data(), in_byte_.data(), own_.data(), order.data(), level_off.data(), uint32_t(level_off.size() - 1),
                    map_keys_.data(), map_vals_.data(), map_keys_.size()};
        done = accel->fail_and_outputs(accel->ctx, tv, fail_b.data(), final_b.data(), cnt_b.data(), dict_b.data());
        built_on_device_ = done;
        lap(done ? "failure links + outputs (device)" : "device build declined");
    }
    if (!done) {
        std::vector<uint32_t> pos(ns);
        for (uint32_t i = 0; i < ns; ++i) pos[order[i]] = i;
        for (uint32_t i = 1; i < ns; ++i) {
            const uint32_t p = parent_pos[i];
            if (p == 0) continue;
            const uint8_t b = byte_b[i];
            uint32_t f = fail_b[p];
            for (;;) {
                const uint32_t c = find_child(order[f], b);
                if (c != kNone) { fail_b[i] = pos[c]; break; }
                if (f == 0) break;
                f = fail_b[f];
            }
        }
        lap("children CSR + failure links");
        for (uint32_t i = 1; i < ns; ++i) {
            const uint32_t f = fail_b[i];
            const bool owns = own_[order[i]] != kNone;
            final_b[i] = owns || final_b[f];
            cnt_b[i] = (owns ? 1u : 0u) + cnt_b[f];
            dict_b[i] = own_[order[f]] != kNone ? f : dict_b[f];
        }
    }

    // ---- byte classes ----------------------------------------------------------------------
    bool used[256] = {false};
    for (uint32_t v = 1; v < ns; ++v) used[in_byte_[v]] = true;
    uint32_t nused = 0;
    for (int b = 0; b < 256; ++b) nused += used[b];
    fa.n_used_bytes = nused;
    fa.class_mode = kClassTable;
    if (nused <= 8) {
        for (uint32_t k = 0; k <= 5 && fa.class_mode != kClassSwar3; ++k) {
            bool seen[8] = {false};
            bool ok = true;
            for (int b = 0; b < 256 && ok; ++b) {
                if (!used[b]) continue;
                const uint32_t j = (uint32_t(b) >> k) & 7;
                if (seen[j]) ok = false;
                seen[j] = true;
            }
            if (!ok) continue;
            fa.class_mode = kClassSwar3;
            fa.swar_shift = k;
            uint32_t other = 8, maxb = 0;
            for (uint32_t j = 0; j < 8; ++j) {
                if (!seen[j] && other == 8) other = j;
                if (seen[j]) maxb = j;
            }
            fa.swar_other = other;
            fa.n_columns = std::max(maxb, other) + 1;
            uint8_t expect[8];
            for (uint32_t j = 0; j < 8; ++j) expect[j] = uint8_t(((j + 1) & 7) << k);  // lands in bucket j+1: never validates
            for (int b = 0; b < 256; ++b)
                if (used[b]) expect[(uint32_t(b) >> k) & 7] = uint8_t(b);
            fa.swar_expect_lo = expect[0] | (expect[1] << 8) | (expect[2] << 16) | (uint32_t(expect[3]) << 24);
            fa.swar_expect_hi = expect[4] | (expect[5] << 8) | (expect[6] << 16) | (uint32_t(expect[7]) << 24);
            for (int b = 0; b < 256; ++b)
                fa.class_of_byte[b] = uint8_t(used[b] ? ((uint32_t(b) >> k) & 7) : other);
        }
    }
    if (fa.class_mode == kClassTable) {
        if (nused == 256) {
            fa.n_columns = 256;
            for (int b = 0; b < 256; ++b) fa.class_of_byte[b] = uint8_t(b);
        } else {
            uint32_t next = 1;  // column 0 = every byte outside the alphabet
            for (int b = 0; b < 256; ++b) fa.class_of_byte[b] = uint8_t(used[b] ? next++ : 0);
            fa.n_columns = nused + 1;
        }
    }

    // ---- state numbering: [hot non-final | hot final | trap | cold non-final | cold final] ----
    fa.n_states = ns + 1;
    fa.entry_bytes = ns <= 65535 ? 2 : 4;
    const uint64_t row_bytes = uint64_t(fa.n_columns) * fa.entry_bytes;
    const uint64_t full_table_bytes = row_bytes * fa.n_states;
    const bool compact = (opt.compact > 0 || (opt.compact < 0 && full_table_bytes > (64ull << 20))) && fa.n_states < (1u << 24);
    if (!compact && full_table_bytes > opt.max_table_bytes) {
        if (error) *error = "transition table of " + std::to_string(full_table_bytes) + " bytes exceeds the limit";
        return false;
    }
    fa.compact = compact ? 1u : 0u;
    const uint64_t class_tab = fa.class_mode == kClassTable ? 512 : 0;  // uint16 column offsets
    uint64_t max_rows = opt.lds_bytes > class_tab ? (opt.lds_bytes - class_tab) / row_bytes : 0;
    if (max_rows < 2) max_rows = 2;  // root + trap always fit (row <= 1 KiB)
    const uint32_t hot = uint32_t(std::min<uint64_t>(ns, max_rows - 1));
    const uint32_t n_shallow1 = cfirst[1];  // states of trie depth <= 1: the root and its children, positions [0, cfirst[1])
    // Dense set = a prefix of the BFS order (closed under failure links, so that a row can be copied from the
    // failure state's): everything when not compact; else the LDS tile, trie depth <= 1 (the failure walk of a
    // compact state then always ends on a dense row after at most one record per level) and what the budget allows.
    uint32_t dense = ns;
    if (compact) {
        const uint64_t by_budget = opt.dense_bytes / row_bytes;
        dense = uint32_t(std::min<uint64_t>(ns, std::max<uint64_t>(std::max<uint64_t>(hot, n_shallow1), by_budget)));
    }
    // Compact: cold non-final states with exactly one trie child (the inside of a long pattern) are numbered last
    // among the cold non-final ones, chain by chain: a chain s1 -> s2 -> ... has consecutive ids and its 8-byte
    // records are read sequentially, 16 steps per 128-byte line.  Depth <= 2 keeps BFS numbering (the early stop
    // of candidate walks tests "non-final, depth <= 2" as an id range).
    // One pass counts the classes, the shallow states (non-final states of trie depth <= 2 are a prefix of their id class
    // -- ids follow BFS order within a class: [0, shallow_hot) and (trap, shallow_cold_end); the exact walk of an anchor
    // candidate stops on them, see run_epilogue: no pattern begun at the candidate's window can still be alive there) and
    // the events + tile exits per byte of uniform text over the used alphabet (see est_flag_rate).
    // Dense-row automata number the same chains the same way: the rows of one planted occurrence then share two or three
    // 128-byte lines instead of one line per byte, and walk_fast skips along them by comparing 16 bytes at a time.
    const uint32_t chain_from = compact ? dense : hot;
    std::vector<uint8_t> in_chain(ns, 0);
    uint32_t hot_nf = 0, cold_nf = 0, cold_chain = 0, dense_cold_nf = 0, dense_cold_f = 0;
    {
        uint32_t sh = 0, sc = 0;
        uint64_t d1 = 0, d3 = 0;
        std::vector<double> w(256, 0.0);
        w[0] = 1.0;
        for (int d = 1; d < 256; ++d) w[d] = w[d - 1] / double(std::max<uint32_t>(nused, 1));
        double rate = 0;
        for (uint32_t i = 0; i < ns; ++i) {
            const bool fin = final_b[i] != 0;
            const uint32_t d = depth[i];
            if (!fin) {
                (i < hot ? hot_nf : cold_nf)++;
                if (i >= chain_from && d >= 3 && cfirst[i + 1] - cfirst[i] == 1) { in_chain[i] = 1; ++cold_chain; }
                if (d <= 2) (i < hot ? sh : sc)++;
            }
            if (i >= hot && i < dense) (fin ? dense_cold_f : dense_cold_nf)++;
            d1 += d == 1;
            d3 += d == 3;
            if (i) {
                if (fin) rate += w[d];
                else if (i >= hot && parent_pos[i] < hot) rate += w[d];  // first cold state on its path
            }
        }
        fa.shallow_hot = sh;
        fa.shallow_cold_end = hot + 1 + sc;
        // When every 3-gram over the first-byte alphabet is a trie path (2048 x 16 B over 6 letters: 216 of 216)
        // a walk on in-alphabet text never falls back to depth <= 2 and the test would only cost time.
        fa.early_stop_useful = d3 < d1 * d1 * d1;
        fa.est_flag_rate = rate;
    }
    fa.f1 = hot_nf;
    fa.trap = hot;
    fa.f2 = hot + 1 + cold_nf;
    fa.dense_nf_end = compact ? hot + 1 + dense_cold_nf : fa.n_states;
    fa.dense_f_end = compact ? fa.f2 + dense_cold_f : fa.n_states;
    fa.n_dense_rows = compact ? fa.dense_nf_end + dense_cold_f : fa.n_states;
    const uint32_t chain_lo = fa.f2 - cold_chain;
    fa.chain_lo = chain_lo;
    std::vector<uint32_t> newid(ns);   // by BFS position, ascending within each class
    {
        uint32_t a = 0, b = hot_nf, c = hot + 1, d = fa.f2;
        for (uint32_t i = 0; i < ns; ++i) {
            if (i < hot) newid[i] = final_b[i] ? b++ : a++;
            else if (final_b[i]) newid[i] = d++;
            else if (!in_chain[i]) newid[i] = c++;
        }
    }
    if (cold_chain) {
        // a chain is numbered from its head (the first of its states whose parent is not of this kind) along its only
        // children; the heads in BFS order
        uint32_t c = chain_lo;
        for (uint32_t i = chain_from; i < ns; ++i) {
            if (!in_chain[i] || in_chain[parent_pos[i]]) continue;
            for (uint32_t j = i; j < ns && in_chain[j]; j = cfirst[j]) newid[j] = c++;
        }
    }

    lap("outputs, classes, numbering");
    // ---- the two-byte window table (automaton.hpp win_table) ----------------------------------------------------
    fa.win_present = false;
    fa.win_table.clear();
    fa.win_state.clear();
    fa.win_hash.clear();
    fa.win_hash_bits = 0;
    {
        uint32_t shortest = 0;
        for (const PatternMeta &m : patterns_)
            if (m.accepted && (shortest == 0 || m.length < shortest)) shortest = m.length;
        if (shortest >= 3 && opt.win_table != 0 && (compact || opt.win_table > 0)) {
            fa.win_present = true;
            fa.win_table.assign(65536, uint16_t(0));
            fa.win_state.assign(65536, newid[0]);
            // depth 1: the state after "xb" is b's node unless (x, b) is a node itself
            for (uint32_t i = 1; i < ns && depth[i] == 1; ++i)
                for (uint32_t a = 0; a < 256; ++a) fa.win_state[a << 8 | byte_b[i]] = newid[i];
            uint32_t n3 = 0;
            std::vector<uint8_t> lists;   // the child lists of the nodes with three children or more: {count, bytes ...}, each at an even offset
            for (uint32_t i = 1; i < ns && depth[i] <= 2; ++i) {
                if (depth[i] != 2) continue;
                const uint32_t cell = uint32_t(byte_b[parent_pos[i]]) << 8 | byte_b[i];
                fa.win_state[cell] = newid[i];
                const uint32_t k = cfirst[i + 1] - cfirst[i];
                n3 += k;
                uint32_t v = 0;
                if (k == 1) {
                    v = 0x8000u | byte_b[cfirst[i]];
                } else if (k == 2) {
                    v = 0x4000u | win_fp7(byte_b[cfirst[i]]) << 7 | win_fp7(byte_b[cfirst[i] + 1]);
                } else if (k >= 3) {
                    // (a list of up to 62 bytes while the 32 KiB behind the cells have room; else "always maybe")
                    if (k <= 62 && lists.size() + k + 2 <= kWinListBytes && lists.size() / 2 < 0x3FFFu) {
                        v = 0xC000u | uint32_t(lists.size() / 2);
                        lists.push_back(uint8_t(k));
                        for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; ++c) lists.push_back(byte_b[c]);
                        if (lists.size() & 1u) lists.push_back(0);
                    } else {
                        v = 0xFFFFu;
                    }
                }
                fa.win_table[cell] = uint16_t(v);
            }
            lists.resize(kWinListBytes, 0);
            for (size_t i = 0; i < lists.size(); i += 2) fa.win_table.push_back(uint16_t(lists[i] | lists[i + 1] << 8));
            // the nodes of depth 3 by their three bytes: open addressing, {key + 1, state}, at most half full
            uint32_t bits = 4;
            while ((1u << bits) < 2 * n3 + 16) ++bits;
            fa.win_hash_bits = bits;
            fa.win_hash.assign(size_t(2) << bits, 0u);
            const uint32_t mask = (1u << bits) - 1u;
            for (uint32_t i = 1; i < ns && depth[i] <= 3; ++i) {
                if (depth[i] != 3) continue;
                const uint32_t p2 = parent_pos[i], p1 = parent_pos[p2];
                const uint32_t key = uint32_t(byte_b[p1]) << 16 | uint32_t(byte_b[p2]) << 8 | byte_b[i];
                uint32_t at = win_hash_slot(key, bits);
                while (fa.win_hash[2 * at] != 0u) at = (at + 1u) & mask;
                fa.win_hash[2 * at] = key + 1u;
                // a node of one child that is not final and whose failure link is of depth <= 2 (nearly every depth-3 node of a
                // big random set) carries that child's byte: text that does not continue with it is back at depth <= 2 without
                // a step through the records (bit 31 of the state word; the byte in bits 24..30 and bit 31 of the key word)
                uint32_t st_word = newid[i];
                if (cfirst[i + 1] - cfirst[i] == 1 && !final_b[i] && depth[fail_b[i]] <= 2 && newid[i] < (1u << 24)) {
                    const uint32_t cb = byte_b[cfirst[i]];
                    st_word |= 0x80000000u | (cb & 0x7Fu) << 24;
                    fa.win_hash[2 * at] |= (cb >> 7) << 31;
                }
                fa.win_hash[2 * at + 1] = st_word;
            }
            lap("window table");
        }
    }
    // ---- tables ----------------------------------------------------------------------------------
    if (fa.entry_bytes == 2)
        build_tables<uint16_t>(fa, dense, fail_b, newid, cfirst, byte_b);
    else
        build_tables<uint32_t>(fa, dense, fail_b, newid, cfirst, byte_b);

    lap("  tables");
    // ---- compact records of the states without a dense row ------------------------------------------
    fa.cold_rec.clear();
    fa.cold_edges.clear();
    if (compact) {
        fa.cold_rec.assign(size_t(fa.n_states) * 2, 0u);
        std::vector<uint32_t> tmp;
        for (uint32_t i = dense; i < ns; ++i) {
            const uint32_t c0 = cfirst[i], nk = cfirst[i + 1] - c0;
            uint32_t x = 0, code = 0;
            if (nk == 1) {
                x = (newid[c0] << 8) | byte_b[c0];
                code = 1;
            } else if (nk >= 2) {
                tmp.clear();
                for (uint32_t v = c0; v < c0 + nk; ++v) tmp.push_back((newid[v] << 8) | byte_b[v]);
                std::sort(tmp.begin(), tmp.end(), [](uint32_t p, uint32_t q) { return (p & 0xffu) < (q & 0xffu); });
                x = uint32_t(fa.cold_edges.size());
                if (nk >= 255) { code = 255; fa.cold_edges.push_back(nk); } else code = nk;
                fa.cold_edges.insert(fa.cold_edges.end(), tmp.begin(), tmp.end());
            }
            fa.cold_rec[size_t(newid[i]) * 2] = x;
            fa.cold_rec[size_t(newid[i]) * 2 + 1] = newid[fail_b[i]] | (code << 24);
        }
    }
    lap("  cold records");
    fa.chain_byte.clear();
    fa.chain_run.clear();
    if (cold_chain) {
        fa.chain_byte.assign(size_t(cold_chain) + 16, 0);
        fa.chain_run.assign(size_t(cold_chain) + 16, 0);
        std::vector<uint32_t> child(cold_chain, 0);
        for (uint32_t i = chain_from; i < ns; ++i) {
            if (!in_chain[i]) continue;
            const uint32_t k = newid[i] - chain_lo, c0 = cfirst[i];
            fa.chain_byte[k] = byte_b[c0];
            child[k] = newid[c0];
        }
        for (uint32_t id = fa.f2; id-- > chain_lo;) {  // backwards: a state's run extends its successor's
            const uint32_t k = id - chain_lo;
            // steps that can be taken blindly: the states they LAND on must be of this kind too (non-final, deeper
            // than 2: no event, no early stop to test); a child numbered elsewhere or a final one ends the run
            if (child[k] != id + 1 || id + 1 >= fa.f2) continue;
            fa.chain_run[k] = uint8_t(std::min<uint32_t>(255u, 1u + fa.chain_run[k + 1]));
        }
    }

    lap("  chain runs");
    fa.own_pattern.assign(fa.n_states, kNone);
    fa.dict_link.assign(fa.n_states, kNone);
    fa.out_count.assign(fa.n_states, 0);
    // final state of every accepted pattern (the state whose own pattern it is)
    std::vector<uint32_t> pattern_state(patterns_.size(), kNone);
    for (uint32_t i = 0; i < ns; ++i) {
        // (a state that is not final owns no pattern, has no dictionary link and no records: the defaults -- and most states of
        // a large set are the insides of patterns, numbered chain by chain, i.e. all over the arrays in this order)
        if (!final_b[i]) continue;
        const uint32_t id = newid[i], own = own_[order[i]];
        fa.own_pattern[id] = own;
        fa.dict_link[id] = dict_b[i] == kNone ? kNone : newid[dict_b[i]];
        fa.out_count[id] = cnt_b[i];
        if (own != kNone) pattern_state[own] = id;
    }
    fa.pattern_length.resize(patterns_.size());
    fa.max_pattern_length = 0;
    for (size_t i = 0; i < patterns_.size(); ++i) {
        fa.pattern_length[i] = patterns_[i].length;
        if (patterns_[i].accepted) fa.max_pattern_length = std::max(fa.max_pattern_length, patterns_[i].length);
    }

    lap("  own/dict/count");
    // ---- flat output lists (matched[] of the reference's nodes) ------------------------------------
    fa.out_off.clear();
    fa.out_list.clear();
    fa.n_flat_outputs = 0;
    fa.max_out_count = 0;
    {
        uint64_t total = 0;
        for (uint32_t s = 0; s < fa.n_states; ++s) {
            total += fa.out_count[s];
            fa.max_out_count = std::max(fa.max_out_count, fa.out_count[s]);
        }
        if (total <= opt.max_flat_outputs) {
            fa.out_off.assign(fa.n_states, 0);
            fa.out_list.resize(size_t(total) * 2);
            fa.n_flat_outputs = total;
            uint32_t at = 0;
            for (uint32_t s = 0; s < fa.n_states; ++s) {
                if (fa.out_count[s] == 0) continue;
                fa.out_off[s] = at;
                for (uint32_t t = fa.own_pattern[s] != kNone ? s : fa.dict_link[s]; t != kNone; t = fa.dict_link[t]) {
                    const uint32_t pat = fa.own_pattern[t];
                    fa.out_list[size_t(at) * 2] = pat;
                    fa.out_list[size_t(at) * 2 + 1] = fa.pattern_length[pat];
                    ++at;
                }
            }
        }
    }

    fa.out_head.clear();
    if (patterns_.size() < (1u << 24)) {
        fa.out_head.assign(size_t(fa.n_states) * 2, 0u);
        for (uint32_t s = 0; s < fa.n_states; ++s) {
            if (fa.out_count[s] == 0) continue;
            const uint32_t t = fa.own_pattern[s] != kNone ? s : fa.dict_link[s];
            const uint32_t pat = fa.own_pattern[t];
            fa.out_head[size_t(s) * 2] = pat | (std::min<uint32_t>(fa.out_count[s], 255u) << 24);
            fa.out_head[size_t(s) * 2 + 1] = fa.pattern_length[pat];
        }
    }
    lap("rows, records, output lists");
    // the whole automaton for the LDS (k_walk_chain): chain-compressed, or -- a small one -- all of its rows
    fa.chain = ChainImage{};
    build_chain_image(fa, opt.chain_bytes, ns, cfirst, parent_pos, byte_b, fail_b, final_b, cnt_b, newid);
    lap("chain image");

    // ---- anchor filter table -------------------------------------------------------------------
    fa.min_pattern_length = 0;
    for (const PatternMeta &m : patterns_)
        if (m.accepted && (fa.min_pattern_length == 0 || m.length < fa.min_pattern_length)) fa.min_pattern_length = m.length;
    fa.anchor_q = 0;
    fa.anchor_mode = 0;
    fa.anchor_choice = 0;
    fa.anchor_offsets.clear();
    fa.anchor_table.clear();
    fa.anchor_bitmap.clear();
    fa.anchor_dense = 0;
    fa.anchor_dense_rate = 0;
    // Dense form: the shortest pattern is 4..6 B, so not every occurrence holds an aligned dword to anchor on -- every
    // pattern is anchored by its first 4 bytes instead and the kernel tests the window at every byte offset.  Worth it only
    // when few text windows are keys: the estimate draws text from the patterns' own byte frequencies (keyword lists over
    // letters: 2,048 words / 26^4 = 0.4 % of the windows; 2,048 patterns over six letters: every window -- not eligible).
    bool every_offset = false;
    if (fa.min_pattern_length >= 4 && fa.min_pattern_length < 7 && opt.anchor_dense != 0) {
        std::vector<uint32_t> prefixes;
        double freq[256] = {0};
        uint64_t total_bytes = 0;
        for (const PatternMeta &m : patterns_) {
            if (!m.accepted) continue;
            uint32_t w;
            std::memcpy(&w, pattern_bytes_.data() + m.byte_off, 4);
            prefixes.push_back(w);
            for (uint32_t i = 0; i < m.length; ++i) freq[pattern_bytes_[m.byte_off + i]] += 1.0;
            total_bytes += m.length;
        }
        std::sort(prefixes.begin(), prefixes.end());
        prefixes.erase(std::unique(prefixes.begin(), prefixes.end()), prefixes.end());
        double rate = 0;
        for (uint32_t w : prefixes)
            rate += (freq[w & 0xff] / total_bytes) * (freq[(w >> 8) & 0xff] / total_bytes) * (freq[(w >> 16) & 0xff] / total_bytes) * (freq[w >> 24] / total_bytes);
        fa.anchor_dense_rate = rate;
        // up to ~ 160 candidate windows per 8 KiB work item (its run list holds 256)
        every_offset = prefixes.size() <= kAnchorMaxKeys && (opt.anchor_dense > 0 || rate * 8192.0 <= 160.0);
    }
    if (fa.min_pattern_length >= 7 || every_offset) {
        const uint32_t q = every_offset ? 4u : fa.min_pattern_length >= 19 ? 16u : fa.min_pattern_length >= 15 ? 12u
                         : fa.min_pattern_length >= 11 ? 8u : 4u;
        // (q-gram, pattern, offset) triples, sorted by q-gram: the distinct q-grams are the filter's keys, and a key
        // that stands for exactly ONE (pattern, offset) pair lets k_walk_runs verify a candidate by comparing the
        // pattern's text instead of walking the automaton (anchor_pairs below)
        struct Triple { std::array<uint32_t, 4> key; uint32_t pattern, offset; };
        std::vector<std::array<uint32_t, 4>> keys;
        std::vector<uint32_t> key_first;                         // key k stands for key_trip[key_first[k] .. key_first[k+1])
        std::vector<std::pair<uint32_t, uint32_t>> key_trip;     // (pattern, offset)
        // anchor offset of pattern pi for alignment class a (the pattern begins a bytes before an aligned text dword
        // ... no: its first aligned dword is at offset a): a itself, unless a later aligned window was chosen (below)
        fa.anchor_offsets.assign(patterns_.size() * 4, 0);
        for (size_t pi = 0; pi < patterns_.size(); ++pi)
            for (uint32_t a = 0; a < 4; ++a) fa.anchor_offsets[pi * 4 + a] = uint8_t(a);
        auto build_keys = [&]() {
            std::vector<Triple> triples;
            for (size_t pi = 0; pi < patterns_.size(); ++pi) {
                const PatternMeta &m = patterns_[pi];
                if (!m.accepted) continue;
                for (uint32_t a = 0; a < (every_offset ? 1u : 4u); ++a) {
                    const uint32_t o = fa.anchor_offsets[pi * 4 + a];
                    Triple t{{0, 0, 0, 0}, uint32_t(pi), o};
                    std::memcpy(t.key.data(), pattern_bytes_.data() + m.byte_off + o, q);
                    triples.push_back(t);
                }
            }
            lap("    triples");
            // (by the first two words as one number, then everything: the order the whole comparison gives)
            sort_by_u64(triples, [](const Triple &t) { return uint64_t(t.key[0]) << 32 | t.key[1]; }, [](const Triple &a, const Triple &b) {
                return a.key != b.key ? a.key < b.key : (a.pattern != b.pattern ? a.pattern < b.pattern : a.offset < b.offset);
            });
            lap("    sort");
            keys.clear();
            key_first.clear();
            key_trip.clear();
            for (size_t i = 0; i < triples.size();) {
                size_t e = i + 1;
                while (e < triples.size() && triples[e].key == triples[i].key) ++e;
                keys.push_back(triples[i].key);
                key_first.push_back(uint32_t(key_trip.size()));
                for (size_t k = i; k < e; ++k) key_trip.emplace_back(triples[k].pattern, triples[k].offset);
                i = e;
            }
            key_first.push_back(uint32_t(key_trip.size()));
            fa.anchor_keys = uint32_t(keys.size());
        };
        // Once every key has its slot in the fingerprint table: the (pattern, offset) pairs a slot stands for -- those of
        // its q-gram, and of any other q-gram that shares bucket and fingerprint -- contiguous in anchor_pairs,
        // anchor_slot_pair[slot] = first | (count - 1) << 28.  More than 8 pairs: kNone (such candidates are walked).
        auto make_slot_pairs = [&](size_t n_slots, const std::vector<size_t> &slot_of_key) {
            struct Entry { size_t slot; uint32_t pattern, offset; };
            std::vector<Entry> all;
            all.reserve(key_trip.size());
            for (size_t ki = 0; ki < keys.size(); ++ki)
                for (uint32_t t = key_first[ki]; t < key_first[ki + 1]; ++t) all.push_back({slot_of_key[ki], key_trip[t].first, key_trip[t].second});
            sort_by_u64(all, [](const Entry &e) { return uint64_t(e.slot) << 32 | e.pattern; }, [](const Entry &a, const Entry &b) {
                return a.slot != b.slot ? a.slot < b.slot : (a.pattern != b.pattern ? a.pattern < b.pattern : a.offset < b.offset);
            });
            fa.anchor_slot_pair.assign(n_slots, kNone);
            fa.anchor_pairs.clear();
            for (size_t i = 0; i < all.size();) {
                size_t e = i + 1;
                while (e < all.size() && all[e].slot == all[i].slot) ++e;
                if (e - i <= 8 && fa.anchor_pairs.size() / 4 < (1u << 28)) {
                    fa.anchor_slot_pair[all[i].slot] = uint32_t(fa.anchor_pairs.size() / 4) | uint32_t(e - i - 1) << 28;
                    for (size_t k = i; k < e; ++k) {
                        const PatternMeta &m = patterns_[all[k].pattern];
                        // (the final state's record count rides along: the candidate check needs no load of its own for it)
                        fa.anchor_pairs.insert(fa.anchor_pairs.end(), {fa.out_count[pattern_state[all[k].pattern]], all[k].offset | (m.length << 8),
                                                                       uint32_t(m.byte_off), pattern_state[all[k].pattern]});
                    }
                }
                i = e;
            }
            fa.anchor_pairs_count = fa.anchor_pairs.size() / 4;
        };
        // A set that is certainly a big one (more distinct first windows than the LDS table takes) gets its keys after the
        // window choice below: listing them here as well -- a sort of every (pattern, class) triple -- only told that much.
        bool big_known = false;
        if (!every_offset && opt.anchor_choice != 0 && accepted_ * 4 > kAnchorMaxKeys) {
            // distinct 64-bit hashes of the windows in a flat open-addressing set, until there are too many for the LDS table
            // (a collision could only make the count smaller: then the keys are listed after all)
            size_t cap = 1024, distinct = 0;
            while (cap < size_t(kAnchorMaxKeys) * 4) cap *= 2;
            std::vector<uint64_t> seen(cap, 0);
            for (size_t pi = 0; pi < patterns_.size() && distinct <= kAnchorMaxKeys; ++pi) {
                const PatternMeta &m = patterns_[pi];
                if (!m.accepted) continue;
                for (uint32_t a = 0; a < 4; ++a) {
                    uint64_t w[2] = {0, 0};
                    std::memcpy(w, pattern_bytes_.data() + m.byte_off + a, q);
                    uint64_t h = mix(w[0] ^ mix(w[1] + 0x632BE59BD9B4E019ull));
                    h += h == 0;
                    size_t at = size_t(h >> 7) & (cap - 1);
                    while (seen[at] != 0 && seen[at] != h) at = (at + 1) & (cap - 1);
                    if (seen[at] == 0) { seen[at] = h; ++distinct; }
                }
            }
            big_known = distinct > kAnchorMaxKeys;
        }
        if (!big_known) build_keys();
        lap("anchor keys");
        // Big sets: WHICH aligned window of a pattern is its anchor is free -- an occurrence whose first aligned text
        // dword lies a bytes into the pattern contains the windows at a, a+4, a+8, ... -- and the LDS bitmap passes a
        // window when ANY key shares its bit, so the (pattern, class) pairs are covered with as few distinct bits as
        // possible: greedy set cover over the candidates' bits.  65,536 patterns of 8..64 B: 9.4 % of the 2^20 bits
        // instead of 22 %, i.e. 2.3 x fewer probes of the fingerprint table, which is what bounds the filter (L2 request
        // rate).  The chosen offset travels in the pair record (verification) and the run of a hit window becomes
        // [w + q - 1, w + Lmax): a pattern anchored at w ends there whatever its offset.
        // (log2_bits: 20, the bitmap the kernel has; `dry`: only count the bits a cover over a bitmap of another size would set --
        // ACMI_COVER_SWEEP with ACMI_BUILD_TRACE, the table of DESIGN 8 item 7)
        auto choose_offsets = [&](uint32_t mul_, uint32_t log2_bits = 20u, bool dry = false) {
            const uint32_t kBits = 1u << log2_bits;
            std::vector<uint32_t> pair_begin{0}, cand_bit;
            std::vector<uint8_t> cand_off;
            std::vector<uint32_t> pair_id;  // pattern * 4 + class
            cand_bit.reserve(pattern_bytes_.size());   // (a pattern has at most one candidate per byte over its four classes)
            cand_off.reserve(pattern_bytes_.size());
            pair_begin.reserve(accepted_ * 4 + 1);
            pair_id.reserve(accepted_ * 4);
            for (size_t pi = 0; pi < patterns_.size(); ++pi) {
                const PatternMeta &m = patterns_[pi];
                if (!m.accepted) continue;
                for (uint32_t a = 0; a < 4; ++a) {
                    for (uint32_t o = a; o + q <= m.length && o <= 252u; o += 4) {
                        uint32_t words[4] = {0, 0, 0, 0};
                        std::memcpy(words, pattern_bytes_.data() + m.byte_off + o, q);
                        cand_bit.push_back(anchor_hash(words, q / 4, mul_) >> (32u - log2_bits));
                        cand_off.push_back(uint8_t(o));
                    }
                    pair_begin.push_back(uint32_t(cand_bit.size()));
                    pair_id.push_back(uint32_t(pi * 4 + a));
                }
            }
            lap("    candidates");
            const uint32_t n_pairs = uint32_t(pair_id.size());
            // bit -> the pairs it is a candidate of (CSR)
            std::vector<uint32_t> bit_off(kBits + 1, 0), bit_pairs(cand_bit.size());
            for (uint32_t b : cand_bit) ++bit_off[b + 1];
            for (uint32_t b = 0; b < kBits; ++b) bit_off[b + 1] += bit_off[b];
            {
                std::vector<uint32_t> cur(bit_off.begin(), bit_off.end() - 1);
                for (uint32_t pr = 0; pr < n_pairs; ++pr)
                    for (uint32_t c = pair_begin[pr]; c < pair_begin[pr + 1]; ++c) bit_pairs[cur[cand_bit[c]]++] = pr;
            }
            lap("    bit CSR");
            // Lazy greedy: bits queued by the number of uncovered pairs they would cover when last counted, taken from the
            // largest count down; a bit whose count has dropped meanwhile moves to the queue of its fresh count (always
            // a lower one, visited later).  Counts are small (2.1 candidates per bit on average), so the queues are
            // plain vectors indexed by count: linear time, where a binary heap of 900,000 bits took 0.1 s.
            std::vector<uint8_t> covered(n_pairs, 0), bit_chosen(kBits, 0);
            uint32_t max_count = 0;
            for (uint32_t b = 0; b < kBits; ++b) max_count = std::max(max_count, bit_off[b + 1] - bit_off[b]);
            std::vector<std::vector<uint32_t>> queue(max_count + 1);
            for (uint32_t b = kBits; b-- > 0;)
                if (bit_off[b + 1] > bit_off[b]) queue[bit_off[b + 1] - bit_off[b]].push_back(b);
            for (uint32_t c = max_count; c >= 1; --c) {
                for (size_t i = 0; i < queue[c].size(); ++i) {
                    const uint32_t b = queue[c][i];
                    uint32_t now = 0;
                    for (uint32_t e = bit_off[b]; e < bit_off[b + 1]; ++e) now += !covered[bit_pairs[e]];
                    if (now == 0) continue;
                    if (now < c) { queue[now].push_back(b); continue; }  // stale count
                    bit_chosen[b] = 1;
                    for (uint32_t e = bit_off[b]; e < bit_off[b + 1]; ++e) covered[bit_pairs[e]] = 1;
                }
                std::vector<uint32_t>().swap(queue[c]);
            }
            lap("    greedy");
            if (dry) {
                uint32_t chosen = 0;
                for (uint32_t b = 0; b < kBits; ++b) chosen += bit_chosen[b];
                std::fprintf(stderr, "[acmi build] cover over 2^%u bits: %u bits set = %.2f %% of the bitmap (what passes of random windows), %u (pattern, class) pairs\n",
                             log2_bits, chosen, 100.0 * double(chosen) / double(kBits), n_pairs);
                return;
            }
            for (uint32_t pr = 0; pr < n_pairs; ++pr)
                for (uint32_t c = pair_begin[pr]; c < pair_begin[pr + 1]; ++c)
                    if (bit_chosen[cand_bit[c]]) { fa.anchor_offsets[pair_id[pr]] = cand_off[c]; break; }
        };
        fa.anchor_slot_pair.clear();
        uint32_t mul = 0x9E3779B1u;
        // (every_offset: the window is hashed raw, see anchor_hash_dense)
        auto key_hash = [&](const std::array<uint32_t, 4> &k, uint32_t mul_) { return every_offset ? anchor_hash_dense(k[0], mul_) : anchor_hash(k.data(), q / 4, mul_); };
        if (!big_known && keys.size() <= kAnchorMaxKeys) {
            // 4 slots per bucket; the dense form tries 2 first when its keys are few (a bucket of two overflows sooner:
            // 2,048 keys in 2^15 buckets find a seed in a few attempts, 4,096 never do)
            fa.anchor_slots = 4;
            for (uint32_t slots : {2u, 4u}) {
                if (fa.anchor_q) break;
                if (slots == 2 && !(every_offset && opt.anchor_slots2 != 0 && keys.size() <= 3072)) continue;
                const uint32_t bits = slots == 2 ? kAnchorBucketBits + 1 : kAnchorBucketBits, n_buckets = 1u << bits, shift = slots == 2 ? 2u : 3u;
                std::vector<uint16_t> tab(size_t(n_buckets) * slots);
                mul = 0x9E3779B1u;
                std::vector<uint8_t> used(n_buckets, 0);
                for (int attempt = 0; attempt < (slots == 2 ? 192 : 512) && fa.anchor_q == 0; ++attempt, mul = mul * 0x2545F491u + 0x6B43A9B5u) {
                    mul = (mul | 1u) & 0xFFFFFFu;
                    // A seed is first tried on the bucket COUNTS alone (a byte per bucket: 16 KiB, not the 128 KiB table): 8,192 keys
                    // in 2^14 buckets of four overflow somewhere for fifteen seeds in sixteen, and filling and clearing the table for
                    // each of them was 5.4 of the 8.7 ms a 2,048-pattern set took to finalize.  (Keys of one bucket that share a
                    // fingerprint share a slot, so a count above `slots` need not be an overflow -- such a seed is passed over, there are others.)
                    {
                        std::fill(used.begin(), used.end(), uint8_t(0));
                        bool fits = true;
                        for (size_t ki = 0; ki < keys.size() && fits; ++ki) {
                            const uint32_t h = key_hash(keys[ki], mul);
                            fits = ++used[(h >> shift) & (n_buckets - 1)] <= slots;
                        }
                        if (!fits) continue;
                    }
                    std::fill(used.begin(), used.end(), uint8_t(0));
                    for (uint32_t b = 0; b < n_buckets; ++b) std::fill_n(&tab[size_t(b) * slots], slots, anchor_empty(b, bits));
                    bool ok = true;
                    for (size_t ki = 0; ki < keys.size(); ++ki) {
                        const uint32_t h = key_hash(keys[ki], mul);
                        const uint16_t fp = uint16_t(h >> 16);                   // fingerprint: high half
                        const uint32_t bucket = (h >> shift) & (n_buckets - 1);  // bits 3..16 (2..16)
                        uint16_t *b = &tab[size_t(bucket) * slots];
                        uint32_t slot = 0;
                        while (slot < used[bucket] && b[slot] != fp) ++slot;
                        if (slot == slots) { ok = false; break; }  // bucket overflow: take another seed
                        if (slot == used[bucket]) b[used[bucket]++] = fp;
                    }
                    if (ok) {  // which slot each key went to (only for the seed that is kept)
                        std::vector<size_t> slot_of(keys.size());
                        for (size_t ki = 0; ki < keys.size(); ++ki) {
                            const uint32_t h = key_hash(keys[ki], mul);
                            const uint16_t fp = uint16_t(h >> 16);
                            const uint32_t bucket = (h >> shift) & (n_buckets - 1);
                            uint32_t slot = 0;
                            while (tab[size_t(bucket) * slots + slot] != fp) ++slot;
                            slot_of[ki] = size_t(bucket) * slots + slot;
                        }
                        make_slot_pairs(size_t(n_buckets) * slots, slot_of);
                        fa.anchor_q = q;
                        fa.anchor_mode = 1;
                        fa.anchor_mul = mul;
                        fa.anchor_table = tab;
                        fa.anchor_dense = every_offset ? 1u : 0u;
                        fa.anchor_slots = slots;
                    }
                }
            }
        } else {
            // mode 2: as many 8-slot buckets as q-grams (load 1: a bucket overflows with probability 1e-6)
            // (the table is sized by the keys of the first attempt -- listed above, or below once the windows are chosen)
            uint32_t bits = 0, nb = 0;
            {
                std::vector<uint16_t> tab;
                std::vector<uint8_t> used;
                // A bucket overflows with probability ~ 1e-6 (load 1 of 8): 0.3 per seed for 262,144 buckets.  The first
                // retries only move the bucket index to other hash bits (anchor_bucket_shift) -- the hash itself, and with
                // it the anchor-window choice and the bitmap (0.35 s for 65,536 patterns), stay; then a new multiplier.
                for (int attempt = 0; attempt < 64 && fa.anchor_q == 0; ++attempt) {
                    const uint32_t shift = uint32_t(attempt) % 7u;
                    if (shift == 0) {
                        if (attempt) mul = mul * 0x2545F491u + 0x6B43A9B5u;
                        mul = (mul | 1u) & 0xFFFFFFu;
                        if (opt.anchor_choice != 0) {
                            if (trace && std::getenv("ACMI_COVER_SWEEP") != nullptr)
                                for (uint32_t lb = 18; lb <= 22; ++lb) choose_offsets(mul, lb, true);
                            choose_offsets(mul);
                            lap("anchor window choice");
                            build_keys();
                            fa.anchor_choice = 1;
                            lap("anchor keys again");
                        }
                    }
                    if (attempt == 0) {
                        bits = 15;
                        while ((size_t(1) << bits) < keys.size() && bits < kAnchorMaxBucketBits) ++bits;
                        if ((size_t(1) << bits) < keys.size()) break;   // more q-grams than the largest table takes: no filter
                        nb = 1u << bits;
                    }
                    tab.assign(size_t(nb) * 8, uint16_t(0));
                    used.assign(nb, 0);
                    std::vector<size_t> slot_of(keys.size());
                    bool ok = true;
                    for (size_t ki = 0; ki < keys.size(); ++ki) {
                        const uint32_t h = anchor_hash(keys[ki].data(), q / 4, mul);
                        const uint16_t fp = uint16_t((h >> 16) | 1u);
                        const uint32_t bucket = (h >> shift) & (nb - 1);
                        uint16_t *b = &tab[size_t(bucket) * 8];
                        uint32_t slot = 0;
                        while (slot < used[bucket] && b[slot] != fp) ++slot;
                        if (slot == 8) { ok = false; break; }
                        if (slot == used[bucket]) b[used[bucket]++] = fp;
                        slot_of[ki] = size_t(bucket) * 8 + slot;
                    }
                    if (!ok) continue;
                    make_slot_pairs(size_t(nb) * 8, slot_of);
                    fa.anchor_q = q;
                    fa.anchor_mode = 2;
                    fa.anchor_mul = mul;
                    fa.anchor_bucket_bits = bits;
                    fa.anchor_bucket_shift = shift;
                    fa.anchor_table = tab;
                    fa.anchor_bitmap.assign(kAnchorBitmapWords, 0u);
                    for (const auto &k : keys) {
                        const uint32_t h = anchor_hash(k.data(), q / 4, mul);
                        fa.anchor_bitmap[h >> 17] |= 1u << ((h >> 12) & 31u);
                    }
                }
            }
        }
    }

    // ---- dense form: the LDS image of the verification data (automaton.hpp dense_image) ---------------------------
    fa.dense_image.clear();
    if (fa.anchor_dense && fa.anchor_q) {
        struct Rec { uint32_t window, state, length, pattern; };
        std::vector<Rec> recs;
        bool fits = fa.max_pattern_length <= 16 && fa.n_states < (1u << 24);
        for (size_t pi = 0; pi < patterns_.size() && fits; ++pi) {
            const PatternMeta &m = patterns_[pi];
            if (!m.accepted) continue;
            Rec r{0, pattern_state[pi], m.length, uint32_t(pi)};
            std::memcpy(&r.window, pattern_bytes_.data() + m.byte_off, 4);
            recs.push_back(r);
        }
        fits = fits && recs.size() <= 4096;
        if (fits) {
            std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
                return a.window != b.window ? a.window < b.window : (a.length != b.length ? a.length > b.length : a.pattern < b.pattern);
            });
            uint32_t n_keys = 0;
            for (size_t i = 0; i < recs.size(); ++i) n_keys += i == 0 || recs[i].window != recs[i - 1].window;
            uint32_t slots = 64;
            while (slots < 2 * n_keys) slots *= 2;
            std::vector<uint32_t> tails;   // words
            std::vector<uint32_t> rec_words;
            for (const Rec &r : recs) {
                const uint32_t toff = uint32_t(tails.size());
                const uint32_t tail_len = r.length - 4u;
                const uint8_t *tb = pattern_bytes_.data() + patterns_[r.pattern].byte_off + 4;
                for (uint32_t i = 0; i < tail_len; i += 4) {
                    uint32_t wv = 0;
                    std::memcpy(&wv, tb + i, std::min<uint32_t>(4u, tail_len - i));
                    tails.push_back(wv);
                }
                if (tail_len == 0) tails.push_back(0u);   // (every record has a readable tail word)
                const uint32_t oc = std::min<uint32_t>(65535u, fa.out_count[r.state]);
                rec_words.push_back(r.state | (r.length << 24));
                rec_words.push_back(toff | (oc << 16));
                if (toff >= 65536u || fa.out_count[r.state] > 65535u) fits = false;
            }
            tails.resize(tails.size() + 4, 0u);   // (three tail words are read unconditionally)
            fits = fits && n_keys <= 4096 && tails.size() * 4 <= 32768 + 16;
            if (fits) {
                const uint32_t mul = 0x9E3779B1u;
                std::vector<uint32_t> ht(size_t(slots) * 2, 0xFFFFFFFFu);
                for (size_t i = 0; i < recs.size();) {
                    size_t e = i + 1;
                    while (e < recs.size() && recs[e].window == recs[i].window) ++e;
                    uint32_t slot = ((recs[i].window * mul) >> 16) & (slots - 1);
                    while (ht[size_t(slot) * 2 + 1] != 0xFFFFFFFFu) slot = (slot + 1) & (slots - 1);
                    ht[size_t(slot) * 2] = recs[i].window;
                    ht[size_t(slot) * 2 + 1] = uint32_t(i) | (uint32_t(std::min<size_t>(e - i, 0xFFFEu)) << 16);
                    i = e;
                }
                std::vector<uint32_t> &img = fa.dense_image;
                img.assign(8, 0u);
                img[0] = slots - 1;
                img[1] = 8;
                img[2] = img[1] + uint32_t(ht.size());
                img[3] = img[2] + uint32_t(rec_words.size());
                img[5] = mul;
                img.insert(img.end(), ht.begin(), ht.end());
                img.insert(img.end(), rec_words.begin(), rec_words.end());
                img.insert(img.end(), tails.begin(), tails.end());
                while (img.size() % 4) img.push_back(0u);
                img[4] = uint32_t(img.size());
            }
        }
    }

    lap("anchor table + pairs");
    // the build-time trie is no longer needed
    std::vector<uint64_t>().swap(map_keys_);
    std::vector<uint32_t>().swap(map_vals_);
    finalized_ = true;
    return true;
}

}  // namespace acmi
