// automaton.hpp -- host-side builder: ordered pattern list -> flat device automaton.
//
// Replaces the reference's build path (ac_trie_add, ahocorasick.c:91-131;
// ac_trie_finalize, :143-155; ac_trie_set_failure, :344-368; node_collect_matches,
// node.c:424-441) with a from-scratch design whose OUTPUT is what the HIP kernels
// want: a failure-free transition table (one lookup per haystack byte), states
// renumbered so that "LDS-resident" and "final" are range tests, and per-state
// output lists kept as (own pattern, dictionary link) chains.
//
// What must be reproduced from the reference is behaviour, not code:
//   * reverse-within-call insertion, first inserted wins on duplicates
//     (php_ahocorasick.c:410-421,457-486; ahocorasick.c:123-124);
//   * empty and > 1024 B patterns are dropped (ahocorasick.c:101-105);
//   * a state reports when any pattern is a suffix of its path (`final` ORed
//     along the failure chain, node.c:436-437) and lists them longest first
//     (own pattern, then every failure ancestor's, node.c:429-435).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace acmi {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMaxPatternLength = 1024;  // actypes.h:148

struct PatternMeta {
    uint64_t byte_off;   // into Automaton::pattern_bytes
    uint32_t length;
    uint32_t id_type;
    int64_t id_number;
    uint64_t user;
    bool accepted;
};

struct BuildOptions {
    uint32_t lds_bytes = 160 * 1024 - 512;  // budget for class table + hot tile
    uint64_t max_table_bytes = 8ull << 30;  // refuse automata whose dense table would exceed this (and that cannot go compact)
    int compact = -1;                       // goto/failure records instead of dense rows for states outside the first
                                            // BFS levels: 1 = always, 0 = never, -1 = when the dense table exceeds 64 MiB
    int anchor_dense = -1;                  // pattern sets whose shortest pattern is 4..6 B: anchor filter over the 4-byte window at EVERY
                                            // byte offset, keys = the patterns' first 4 bytes (-1 = when the keys are selective enough
                                            // on text that looks like the patterns, 0 = never, 1 = whenever the shortest pattern is >= 4 B)
    int anchor_slots2 = -1;                 // dense form: two-slot buckets when the keys are few (-1 / 1 = up to 3,072 keys, 0 = never)
    int anchor_choice = -1;                 // big-set anchor filter: pick each pattern's anchor windows so that they share
                                            // bitmap bits (-1 / 1 = yes, 0 = the first aligned window as in the small-set table)
    uint64_t dense_bytes = 4ull << 20;      // compact: dense rows are kept for the LDS tile, every state of trie depth <= 1
                                            // and further BFS-first states up to this many bytes of rows
    uint64_t max_flat_outputs = 1ull << 25; // flat output lists up to this many entries (256 MiB), else dictionary links only
    // chain-compressed image of the WHOLE automaton for LDS (ChainImage): built when it fits this many bytes -- a CU's 160 KiB less
    // the byte classes (512 B) and a little slack; 0 = never
    uint32_t chain_bytes = 160 * 1024 - 512 - 256;
    // the two-byte window table (FlatAutomaton::win_table): -1 = for compact automata (sets far too big for the LDS) whose shortest
    // pattern is >= 3 B, 0 = never, 1 = whenever the shortest pattern is >= 3 B
    int win_table = -1;
};

// The whole automaton in a form that fits the LDS of one CU where dense rows for every state do not (k_chain.hip k_walk_chain):
// the goto edge + failure pointer of MultiFast's nodes (node.h:37-67) for the states that have ONE child or none -- the inside of
// the patterns, 85-90 % of a set of a few thousand -- and failure-free dense rows for the rest.
//   * D states, index [0, n_dense): every state with two or more children, the root, every failure target of a state that is not
//     one of them, and every ancestor of such a state (so that a C state's child is a C state and its failure state a D state).
//     Row = n_columns x uint16: the target's index | 0x8000 when the target is final.
//   * C states, index [n_dense, n_index): numbered chain by chain -- a C state's only child is the next index.  Record (4 B):
//     bits 0..15 index of the failure state (a D state); byte 2 twice the class of the edge byte (the column offset a step has
//     in hand; 0xFF: no child -- at most 127 classes); bits 24..28 min(records of the child's event, 31), bit 31 the child is
//     final (a row entry read as a signed halfword and a record are then "final" by the same test: negative).  delta(s, b) = b is the edge ? s + 1 : row[fail(s)][b] --
//     ahocorasick.c:199-212's hop loop, ended after one hop by construction.
//   * cnt: min(records, 255) of the states whose index is below cnt_n: the D states and the chains whose head is final -- the final
//     states a row leads to, but for the odd one inside a chain (the child of a C state on the row's failure path), whose count
//     is looked up with the event's expansion; a state entered along its chain has its count in the parent's record.
// image = rows | records | cnt (each part 4-byte aligned).  head / orig: per index, what an event's expansion needs (global memory).
struct ChainImage {
    uint32_t n_dense = 0, n_index = 0, row_bytes = 0;
    uint32_t rec_off = 0, cnt_off = 0, cnt_n = 0;
    std::vector<uint8_t> image;
    std::vector<uint32_t> head;   // 2 words per index: {first (longest) pattern | min(records, 255) << 24, its length} (out_head's form)
    std::vector<uint32_t> orig;   // per index: the state id of the other tables
    // the same heads for an LDS that has no room for 8 bytes per index (k_chain_emit): rank of every final index among the final
    // ones (0xFFFF: not final), and the heads by rank
    std::vector<uint16_t> frank;
    std::vector<uint32_t> fhead;  // 2 words per final index, in index order
    bool present() const { return !image.empty(); }
};

// The part of finalize() that a device can take over (engine.cpp / k_misc.hip k_build_level): failure links and
// per-state outputs are level-synchronous -- a node needs only shallower nodes' results -- so one kernel launch per
// trie depth computes them for all nodes of that depth at once, walking the same open-addressing child map.
struct TrieView {
    uint32_t n_nodes;
    const uint32_t *parent;        // per node
    const uint8_t *in_byte;        // per node
    const uint32_t *own;           // per node: pattern index or kNone
    const uint32_t *order;         // nodes in BFS order (order[0] = root)
    const uint32_t *level_off;     // n_levels + 1 offsets into `order`: level d = [level_off[d], level_off[d+1])
    uint32_t n_levels;
    const uint64_t *map_keys;      // (parent << 8 | byte) + 1, 0 = empty; linear probing from mix(key) & (map_size - 1)
    const uint32_t *map_vals;
    uint64_t map_size;
};
struct BuildAccel {
    void *ctx = nullptr;
    // fills fail / is_final / cnt / dict (all n_nodes entries, indexed by BFS POSITION -- entry i belongs to node order[i] --
    // and fail / dict naming their targets by position too, dict kNone when there is none); false: not done, the host
    // computes them
    bool (*fail_and_outputs)(void *ctx, const TrieView &tv, uint32_t *fail, uint8_t *is_final, uint32_t *cnt, uint32_t *dict) = nullptr;
};

enum ClassMode : uint32_t { kClassTable = 0, kClassSwar3 = 1 };

// Everything the device needs, in host memory, after finalize().
struct FlatAutomaton {
    uint32_t n_states = 0;       // real states + 1 trap row
    uint32_t n_columns = 0;      // byte classes per row
    uint32_t entry_bytes = 2;    // 2 (uint16) when n_states <= 65536, else 4
    uint32_t f1 = 0;             // [0,f1) hot non-final, [f1,trap) hot final
    uint32_t trap = 0;           // id of the trap row == number of hot states
    uint32_t f2 = 0;             // (trap,f2) cold non-final, [f2,n_states) cold final
    uint32_t shallow_hot = 0;    // [0,shallow_hot) and (trap,shallow_cold_end): non-final states of trie depth <= 2
    uint32_t shallow_cold_end = 0;
    bool early_stop_useful = false;  // some 3-gram over the first-byte alphabet is not a trie path
    // Expected flagged positions per text byte of k_scan_lds' optimistic pass on text drawn uniformly from the
    // patterns' own alphabet: arrivals in final states (events) plus transitions that leave the LDS tile, each
    // state weighed |alphabet|^-depth.  The engine starts keyword-like and dense sets on the exact whole-item
    // walk instead of finding out on their first batch.
    double est_flag_rate = 0;
    // Compact form (BuildOptions::compact).  The reference keeps 88-byte nodes with sorted edge arrays and a failure
    // pointer (node.h:37-67); a dense failure-free row per state is 1 KiB at 256 columns -- 2.3 GB for the
    // 2.27 M states of a 64K-pattern set, every deep step of a walk a far-memory access.  Compact keeps dense rows
    // for the first BFS levels only (where random text spends 99.9 % of its steps) and, for every other state, one
    // 8-byte record {edge, failure}: states with one trie child (96 % of a big random set: the inside of a
    // pattern) carry the edge in the record and are numbered along their pattern, so a walk down a pattern reads
    // 16 states per 128-byte line; states with several children point into a sorted edge array.  A miss follows
    // the failure link until it reaches a dense row -- the same transition function, computed the MultiFast way
    // where that is cheap and by one lookup where it is hot.
    uint32_t compact = 0;
    uint32_t dense_nf_end = 0;   // dense rows exist for ids [0,dense_nf_end) (hot, trap, BFS-first cold non-final) ...
    uint32_t dense_f_end = 0;    // ... and [f2,dense_f_end) (BFS-first cold final), the latter at row dense_nf_end + id - f2
    uint32_t n_dense_rows = 0;   // rows in `table` (== n_states when not compact)
    std::vector<uint32_t> cold_rec;    // compact: 2 words per state id {x, y}: y = failure id | code << 24;
                                       //   code 0: no child; 1: x = child id << 8 | byte; 2..254: that many edges at
                                       //   cold_edges[x..]; 255: cold_edges[x] = count, edges behind it
    std::vector<uint32_t> cold_edges;  // child id << 8 | byte, ascending byte
    // The single-child states that are numbered along their patterns, ids [chain_lo, f2): edge byte per state and
    // how many steps from this state on lead to the next id, which is again such a state (capped at 255).  A candidate walk that
    // reaches such a state compares up to 16 text bytes against 16 edge bytes at once instead of taking 16
    // dependent steps (kdev.hpp walk_fast).  Both arrays carry 16 bytes of padding.
    uint32_t chain_lo = 0;
    std::vector<uint8_t> chain_byte, chain_run;
    // The two-byte window table (k_win.hip k_walk_win; sets whose shortest pattern is >= 3 B: no state of trie depth <= 2 is final).
    // While the state is that shallow it is a function of the last two text bytes alone, and what the walk needs to know is
    // whether the next byte c leaves depth 2 downward from the node of (a, b): win_table[a << 8 | b], 16 bits, staged into LDS --
    //   0                          no node (a, b) or no child;
    //   0x8000 | c0                one child: exact;
    //   0x4000 | f(c1) << 7 | f(c2)   two children, 7-bit fingerprints f (win_fp7): never a false "no";
    //   0xC000 | offset            three or more: their bytes as a list {count, bytes ...} at 2-byte unit `offset` of the
    //                              kWinListBytes behind the 65,536 cells (exact); 0xFFFF: no room for the list, always "maybe".
    // win_state[a << 8 | b] = the state after the text "ab" read from the root (the node of (a, b); else b's; else the root).
    // win_hash: the nodes of trie depth 3 by their three bytes, {key + 1, state} in 2^win_hash_bits slots (open addressing from
    // win_hash_slot, linear probing; 0 = empty): a "maybe" is settled by one probe, not by a walk through records and edges.
    bool win_present = false;
    std::vector<uint16_t> win_table;   // 65,536 cells + kWinListBytes / 2 list words
    std::vector<uint32_t> win_state;
    std::vector<uint32_t> win_hash;
    uint32_t win_hash_bits = 0;
    uint32_t class_mode = kClassTable;
    uint32_t swar_shift = 0;     // class = (byte >> shift) & 7 when byte is in the alphabet
    uint32_t swar_other = 0;     // column of every byte outside the alphabet
    uint32_t swar_expect_lo = 0; // expected byte of buckets 0..3 (v_perm source)
    uint32_t swar_expect_hi = 0; // expected byte of buckets 4..7
    uint32_t max_pattern_length = 0;
    uint32_t n_used_bytes = 0;
    uint64_t table_bytes = 0;        // size of `table` (kept when the host copy is released)
    uint64_t lds_image_bytes = 0;    // size of `lds_image` (likewise)
    std::vector<uint8_t> table;      // n_dense_rows x n_columns entries, true transitions
    std::vector<uint8_t> lds_image;  // (trap+1) x n_columns entries, cold targets -> trap
    uint8_t class_of_byte[256] = {0};
    std::vector<uint32_t> own_pattern;  // per state
    std::vector<uint32_t> dict_link;    // per state: nearest failure ancestor owning a pattern
    std::vector<uint32_t> out_count;    // per state: records one event expands to
    std::vector<uint32_t> pattern_length;  // per caller-order pattern index
    // The same output lists flattened the way the reference keeps them per node (matched[], node.h:49-51,
    // filled by node_collect_matches, node.c:424-441): entries out_off[s] .. out_off[s] + out_count[s] are
    // {pattern index, pattern length} of state s's event, longest first.  The dense-match drain reads them
    // with independent loads instead of chasing dict_link.  Left empty (the drain then follows the links)
    // when the lists would exceed BuildOptions::max_flat_outputs entries -- nested pattern families can square.
    std::vector<uint32_t> out_off;      // per state (valid where out_count > 0)
    std::vector<uint32_t> out_list;     // 2 words per entry: pattern, length
    uint64_t n_flat_outputs = 0;        // entries of out_list (kept when the host copy is released)
    // The head of every final state's output: {first (longest) pattern | min(records, 255) << 24, its length} -- what an event
    // with ONE record needs (most events), in one 8-byte read instead of three dependent ones (own_pattern / dict_link,
    // pattern_length, out_count).  Empty for more than 2^24 patterns.
    std::vector<uint32_t> out_head;     // 2 words per state
    uint32_t max_out_count = 0;

    // ---- anchor filter (optional front end of the scan, see k_anchor.hip k_scan_anchor) ----
    // Every occurrence of a pattern of length >= q + 3 contains a 4-byte-aligned text dword within
    // its first 4 bytes; the q bytes from there are P[o, o+q) for o in 0..3 ("anchor q-grams").
    //   mode 1 (<= kAnchorMaxKeys q-grams): anchor_table = kAnchorBuckets buckets x 4 slots of 16-bit
    //     fingerprints, resident in LDS (empty slots: anchor_empty()).
    //   mode 2 (more): anchor_bitmap = 2^20 bits in LDS (bit = top 20 bits of the hash) in front of a
    //     fingerprint table in global memory / L2: 2^anchor_bucket_bits buckets x 8 slots, bucket = low
    //     bits of the hash (from bit anchor_bucket_shift on), fingerprint = (hash >> 16) | 1, 0 = empty slot.
    uint32_t min_pattern_length = 0;
    uint32_t anchor_q = 0;        // 0 = not eligible; else 4, 8, 12 or 16 bytes
    // "dense" form (mode 1 only, q = 4): every pattern is anchored by its FIRST four bytes and the kernel tests the window at
    // every byte offset of the text, not only the aligned ones -- for sets whose shortest pattern is 4..6 B (keyword lists),
    // which have no aligned dword inside every occurrence.  One key per pattern instead of four, hash anchor_hash_dense.
    uint32_t anchor_dense = 0;
    // slots per bucket of the LDS table (mode 1): 4 (2^14 buckets, bucket = hash bits 3..16) or -- dense form with few keys -- 2
    // (2^15 buckets, bits 2..16: one 4-byte read and one packed subtract per window instead of 8 bytes and three operations)
    uint32_t anchor_slots = 4;
    double anchor_dense_rate = 0; // expected hit windows per text byte, text drawn from the patterns' own byte frequencies
    // Dense form, small sets: everything the verification of a hit window needs, laid out to be staged into LDS by
    // k_verify_dense (a hit is decided without a single table access in global memory; the dense form has tens of hits per
    // work item).  Words: [0] hash-table mask (slots - 1), [1] word offset of the hash table, [2] of the pattern records,
    // [3] of the tail bytes, [4] total words, [5] multiplier of the slot hash, [6..7] reserved; then
    //   hash table   slots x {window (the pattern's first four bytes), first record | count << 16}   (0xFFFFFFFF in the
    //                second word: empty; open addressing, linear probing from (window * mul) >> 16 & mask);
    //   records      per pattern with that prefix {final state | length << 24, tail word offset | min(out_count, 65535) << 16};
    //   tails        the patterns' bytes from the fifth on, each padded to a whole word.
    // Empty when the set is too big for it (more than 4,096 keys or patterns, 32 KiB of tails, a state id of 2^24 or more, a
    // pattern longer than 16 B): candidates are then verified through the tables in global memory.
    std::vector<uint32_t> dense_image;
    uint32_t anchor_mode = 0;     // 0 none, 1 LDS table, 2 LDS bitmap + global table
    uint32_t anchor_mul = 0;      // odd multiplier of the window hash (seed that gave no bucket overflow)
    uint32_t anchor_keys = 0;
    uint32_t anchor_bucket_bits = 0;  // mode 2
    uint32_t anchor_bucket_shift = 0; // mode 2: bucket = (hash >> shift) & (2^bits - 1)
    std::vector<uint16_t> anchor_table;
    std::vector<uint32_t> anchor_bitmap;  // mode 2: 32768 words
    // Candidate verification by text compare (k_walk_runs).  A hit window whose q-gram belongs to exactly one
    // (pattern, offset) pair can only be the anchor of that pattern: comparing the pattern's bytes with the text
    // decides the candidate without a single automaton step.  anchor_slot_pair[bucket * slots + slot] = index of
    // the 4-word record {out_count of the final state, offset | length << 8, byte offset in pattern_bytes, final state} in anchor_pairs,
    // or kNone when the slot stands for several pairs (patterns sharing a q-gram, two q-grams sharing a fingerprint):
    // those candidates are walked.
    std::vector<uint32_t> anchor_slot_pair;
    std::vector<uint32_t> anchor_pairs;
    uint64_t anchor_pairs_count = 0;
    // anchor_offsets[pattern * 4 + a]: offset of the anchor window of an occurrence whose first aligned text dword lies
    // a bytes into the pattern (a itself; with anchor_choice a + 4j, see automaton.cpp)
    uint32_t anchor_choice = 0;
    std::vector<uint8_t> anchor_offsets;
    ChainImage chain;             // empty when the set does not fit (or fits the LDS as dense rows anyway)
};

constexpr uint32_t kAnchorBucketBits = 14;
constexpr uint32_t kAnchorBuckets = 1u << kAnchorBucketBits;  // x 4 slots x 2 B = 128 KiB of LDS
constexpr uint32_t kAnchorMaxKeys = 12288;          // mode 1
constexpr uint32_t kAnchorBitmapWords = 1u << 15;    // mode 2: 2^20 bits = 128 KiB of LDS
constexpr uint32_t kAnchorMaxBucketBits = 21;        // mode 2: up to 2 M buckets x 16 B = 32 MiB

// Window hash shared by the host builder and the device kernel: fold each little-endian dword of
// the q-gram (d ^ d >> 11, so that its top byte reaches the low 24 bits), then chain them with
// 24-bit multiply-adds, h = (h mod 2^24) * mul + fold(next): one full-rate v_mad_u32_u24 per dword on
// the device (a 32-bit v_mul_lo_u32 costs 8x as much and dominated the first version of the kernel).
// mul odd makes every step a bijection of the 24 bits it keeps.  Bucket = bits 3..16, fingerprint =
// bits 16..31: they share bit 16, so a slot value whose bit 0 differs from the bucket's top bit can
// never equal a window's fingerprint there -- that is the "empty" marker (anchor_empty).
inline uint32_t anchor_fold(uint32_t d) { return d ^ (d >> 11); }
inline uint32_t anchor_hash(const uint32_t *words, uint32_t n_words, uint32_t mul) {
    // a lone dword (4-byte windows) is hashed raw: its low three bytes times the multiplier plus the dword shifted down by 13
    // bits -- a shift and a multiply-add on the device (fold + multiply-add of the folded value took four instructions per
    // window).  The shift matters: a product spreads a byte's influence UPWARD only, and word lists are full of families that
    // share their first three letters ("conc", "cond", "conf", ...) -- with the fourth byte reaching the fingerprint bits
    // only, such a family met in one or two buckets of four slots and no seed could place it.  Shifted by 13, the fourth byte
    // lands on bits 11..18 (six bits of the bucket index), the third on bits 3..10.
    if (n_words == 1) return (words[0] & 0xFFFFFFu) * (mul & 0xFFFFFFu) + (words[0] >> 13);
    uint32_t h = anchor_fold(words[0]);
    for (uint32_t i = 1; i < n_words; ++i) h = (h & 0xFFFFFFu) * (mul & 0xFFFFFFu) + anchor_fold(words[i]);
    return h;
}
// dense form: ONE multiply-add on the raw window (v_lshrrev + v_mad_u32_u24 on the device), see anchor_hash
inline uint32_t anchor_hash_dense(uint32_t w, uint32_t mul) { return anchor_hash(&w, 1, mul); }
inline uint32_t win_fp7(uint32_t c) { return (c ^ (c >> 3) ^ (c >> 5)) & 127u; }   // (k_win.hip has the same)
constexpr uint32_t kWinListBytes = 30u * 1024u;   // child lists behind the window table's cells (LDS: 128 KiB + this)
inline uint32_t win_hash_slot(uint32_t key, uint32_t bits) { return (key * 0x9E3779B1u) >> (32u - bits); }
inline uint16_t anchor_empty(uint32_t bucket, uint32_t bucket_bits = kAnchorBucketBits) { return uint16_t(((bucket >> (bucket_bits - 1)) & 1u) ^ 1u); }

class Automaton {
public:
    Automaton();

    // One init()/add_patterns() call.  Returns number accepted.
    size_t add_batch(const uint8_t *const *values, const uint32_t *lengths,
                     const uint32_t *id_types, const int64_t *id_numbers,
                     const uint64_t *users, size_t n);

    // Returns false (and sets error) when the automaton cannot be represented.
    bool finalize(const BuildOptions &opt, std::string *error, const BuildAccel *accel = nullptr);
    bool built_on_device() const { return built_on_device_; }

    // search and replace (AC_PATTERN_t.rtext, actypes.h:88-94): replacement text of caller-order pattern `index`, kept as
    // the reference keeps it -- through strncpy (mpool_strndup, mpool.c:165-178): bytes up to the first NUL, zeros after
    // it, the length the caller's.  text == nullptr: the pattern is not to be replaced.  Before finalize.
    void set_replacement(uint32_t index, const uint8_t *text, uint32_t length);
    const std::vector<uint8_t> &replacement_bytes() const { return rep_bytes_; }
    const std::vector<uint64_t> &replacement_off() const { return rep_off_; }   // per pattern index: offset << 32 | length, ~0 = none
    bool has_replacement() const;  // some ACCEPTED pattern carries one (rd->has_replacement, replace.c:95-131)

    bool finalized() const { return finalized_; }
    size_t node_count() const { return parent_.size(); }
    size_t accepted_count() const { return accepted_; }
    const std::vector<PatternMeta> &patterns() const { return patterns_; }
    const std::vector<uint8_t> &pattern_bytes() const { return pattern_bytes_; }
    const FlatAutomaton &flat() const { return flat_; }
    // drop the host copies of the big arrays once they live on the device (acmi_debug_tables then has nothing to show)
    void release_host_tables() {
        std::vector<uint8_t>().swap(flat_.table);
        std::vector<uint8_t>().swap(flat_.lds_image);
        std::vector<uint32_t>().swap(flat_.cold_rec);
        std::vector<uint32_t>().swap(flat_.cold_edges);
        std::vector<uint8_t>().swap(flat_.chain_byte);
        std::vector<uint8_t>().swap(flat_.chain_run);
        std::vector<uint16_t>().swap(flat_.win_table);
        std::vector<uint32_t>().swap(flat_.win_state);
        std::vector<uint32_t>().swap(flat_.win_hash);
        std::vector<uint32_t>().swap(flat_.anchor_slot_pair);
        std::vector<uint32_t>().swap(flat_.anchor_pairs);
        std::vector<uint32_t>().swap(flat_.out_list);
        std::vector<uint32_t>().swap(flat_.out_head);
        std::vector<uint32_t>().swap(flat_.dense_image);
        std::vector<uint8_t>().swap(flat_.chain.image);
        std::vector<uint32_t>().swap(flat_.chain.head);
        std::vector<uint32_t>().swap(flat_.chain.orig);
        std::vector<uint16_t>().swap(flat_.chain.frank);
        std::vector<uint32_t>().swap(flat_.chain.fhead);
        host_tables_released_ = true;
    }
    bool host_tables_released() const { return host_tables_released_; }

private:
    uint32_t find_child(uint32_t node, uint8_t byte) const;
    uint32_t add_child(uint32_t node, uint8_t byte);
    void rehash(size_t new_cap);

    // trie: node 0 is the root; every other node is one edge (parent, byte) -> node
    std::vector<uint32_t> parent_;
    std::vector<uint8_t> in_byte_;
    std::vector<uint32_t> own_;  // pattern index owned by the node, or kNone

    // open-addressing map (parent << 8 | byte) -> child
    std::vector<uint64_t> map_keys_;
    std::vector<uint32_t> map_vals_;
    size_t map_used_ = 0;

    std::vector<PatternMeta> patterns_;
    std::vector<uint8_t> pattern_bytes_;
    size_t accepted_ = 0;
    std::vector<uint8_t> rep_bytes_;
    std::vector<uint64_t> rep_off_;
    bool finalized_ = false;
    bool built_on_device_ = false;
    bool host_tables_released_ = false;
    FlatAutomaton flat_;
};

}  // namespace acmi
