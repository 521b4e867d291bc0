// device_types.hpp -- plain structs shared by the host engine and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace acmi {

// Automaton as the kernels see it (all pointers are device pointers).
struct DevAutomaton {
    const void *table;           // n_states x n_columns transitions (uint16 or uint32)
    const void *lds_image;       // [class offsets (TABLE mode, 512 B)] + (trap+1) x n_columns tile
    const uint16_t *class_off;   // 256 entries: byte -> column * entry_bytes (global copy)
    const uint32_t *own_pattern; // per state
    const uint32_t *dict_link;   // per state
    const uint32_t *out_count;   // per state
    const uint32_t *pattern_length;  // per pattern
    const uint32_t *out_off;     // per state: first entry of its flat output list (null: follow dict_link)
    const uint2 *out_list;       // {pattern, length}, longest first
    const uint2 *out_head;       // per state: {first pattern | min(records, 255) << 24, its length} (null: more than 2^24 patterns)
    uint32_t max_out_count;      // longest output list
    // compact form (automaton.hpp): dense rows for ids [0,dense_nf_end) and [f2,dense_f_end) (the latter at row
    // dense_nf_end + id - f2); every other state has a record {x, failure | code << 24} and maybe sorted edges
    uint32_t compact;
    uint32_t dense_nf_end, dense_f_end;
    const uint2 *cold_rec;
    const uint32_t *cold_edges;
    // ids [chain_lo, f2): single-child states numbered along their pattern -- edge byte and length of the run of
    // "child == id + 1" states from here on (automaton.hpp); null when there are none
    uint32_t chain_lo;
    const uint8_t *chain_byte, *chain_run;
    // chain-compressed image of the whole automaton (automaton.hpp ChainImage, k_chain.hip): null when there is none
    const uint8_t *chain_img;    // rows | records | counts, staged into LDS as a whole
    uint32_t chain_img_bytes, chain_dense, chain_index, chain_rec_off, chain_cnt_off, chain_cnt_n;
    const uint2 *chain_head;     // per index: {first pattern | min(records, 255) << 24, its length}
    const uint32_t *chain_orig;  // per index: state id
    const uint16_t *chain_frank; // per index: rank among the final indices (0xFFFF: not final) ...
    const uint2 *chain_fhead;    // ... and the heads by rank (what k_chain_emit stages into LDS)
    uint32_t chain_finals;
    // the two-byte window table (automaton.hpp win_table; k_win.hip): null when the set has none
    const uint16_t *win_table;   // 65,536 cells + the child lists, staged into LDS
    const uint32_t *win_state;   // 65,536 states
    const uint2 *win_hash;       // depth-3 nodes by their three bytes: {key + 1, state}, 2^win_hash_bits slots
    uint32_t win_hash_bits;
    const uint8_t *pad16;        // 16 readable bytes: where unconditional loads of lanes with nothing to load go
    uint32_t shallow_hot, shallow_cold_end;  // non-final states of trie depth <= 2: [0,shallow_hot), (trap,shallow_cold_end)
    uint32_t early_stop;         // 1: walks of anchor candidates stop on them (run_epilogue)
    uint32_t class_identity;     // TABLE mode with 256 columns: class(byte) == byte
    uint32_t n_states;
    uint32_t n_columns;
    uint32_t row_bytes;          // n_columns * entry_bytes
    uint32_t f1, trap, f2;       // state-id ranges, see automaton.hpp
    uint32_t lds_image_bytes;    // bytes to stage into LDS (class offsets + tile)
    uint32_t tile_offset;        // byte offset of the tile inside the LDS image (0 or 512)
    uint32_t swar_shift, swar_expect_lo, swar_expect_hi, swar_other_off;  // other column * entry_bytes
    uint32_t other_byte;         // a byte value no pattern uses (its column leads every state to the root), 0x100: there is none
    // anchor filter (k_scan_anchor): 2^14 buckets x 4 x uint16 fingerprints, see automaton.hpp
    const void *anchor_table;    // mode 1: 128 KiB staged into LDS; mode 2: 2^bucket_bits x 8 x uint16, read through L2
    const uint32_t *anchor_bitmap;  // mode 2: 2^20 bits staged into LDS
    // candidate verification by text compare (automaton.hpp anchor_pairs); null: every candidate is walked
    const uint32_t *anchor_slot_pair;
    const uint4 *anchor_pairs;   // {pattern, offset | length << 8, byte offset in pattern_bytes, final state}
    const uint8_t *pattern_bytes;
    uint32_t anchor_q, anchor_mul, min_pattern_length, max_pattern_length;
    uint32_t anchor_mode, anchor_bucket_mask, anchor_bucket_shift;   // mode 2: bucket = (hash >> shift) & mask
    uint32_t anchor_choice;      // 1: a pattern's anchor window may lie deeper than its first aligned dword (offset in the pair record)
    uint32_t anchor_dense;       // 1: windows at every byte offset, keys = the patterns' first four bytes (automaton.hpp)
    uint32_t anchor_slots;       // slots per bucket of the LDS table: 4, or 2 (dense form with few keys: 2^15 buckets)
    uint32_t anchor_dense_hot;   // dense form: a dozen or more hit windows expected per 8 KiB -- the kernel variant that works out every step's hits branch-free
    const uint32_t *dense_image; // dense form, small sets: the verification data k_verify_dense stages into LDS (automaton.hpp), or null
    uint32_t dense_image_words;
};

// One batch launch.
struct BatchArgs {
    const uint8_t *blob;
    const uint64_t *offsets;     // n_haystacks + 1
    const uint64_t *item_base;   // n_haystacks + 1: first work item of each haystack
    uint32_t n_haystacks;
    uint32_t chunk_bytes;        // reporting bytes per work item
    uint32_t warm_bytes;         // max_pattern_length - 1
    uint32_t short_items;        // anchor kernel: not every task is full -- the variant that skips the rows a task lacks
    uint32_t tiny_items;         // anchor kernel: haystacks average 1 KiB or less -- the variant that packs one-row items G to a task
    uint32_t align_warm;         // k_items: 1 (anchor kernel) a chunk inside its haystack lengthens its warm-up to the 128-byte line
                                 // below, a haystack that starts unaligned is scanned from the 16-byte boundary below (ItemDesc::floor);
                                 // 2 (LDS-tile walk) both from the 128-byte line below, the bytes under `floor` masked
    uint32_t slab_events;        // K: event slots per work item
    uint32_t slab_major;         // 1: the slab is item-major (kdev.hpp slab_slot)
    uint32_t first_only;         // findAll == false
    uint32_t *item_hay;          // per item: haystack index
    uint4 *item_rec;             // per item: {scan offset lo, hi | warm << 16, scan bytes, pos0} (k_scan_anchor)
    uint32_t *ev_count;          // per item: events seen
    uint64_t *rec_count;         // per item: records those events expand to
    uint2 *slab;                 // event e of item i at slab[slab_slot(i, e)] (e * run_stride + i, or i * K + e): {pos, state}; K events per item
    uint32_t *overflow_list;     // items whose events did not fit their slab
    uint32_t *counters;          // [0] = number of overflowed items
    // main loop -> epilogue hand-over inside k_scan_lds: byte runs of an item that need the exact path
    uint4 *runs;                 // run r of item i at runs[r * run_stride + i]: {lo, hi, entry state or marker, -}, bytes
                                 // [lo, hi) in item scan coordinates.  Run-major, so that the usual single run per item is
                                 // written and read as whole lines (item-major: 16 B of every 128-byte line)
    uint64_t run_stride;         // >= number of items
    uint32_t run_cap;
    uint32_t *run_count;         // per item: runs appended by k_scan_anchor, walked by k_walk_runs (> run_cap: whole item)
    uint32_t run_dense;          // 1 (dense anchor form), 2 (the aligned forms): `runs` holds 4-byte window-list entries (kdev.hpp) --
                                 // the hit windows' start offsets; a run's bounds follow from its start
    uint32_t run_major;          // window lists: 1 item-major (entry r of item i at ((uint32_t *)runs)[i * run_cap + r]: the dense form,
                                 // and long lists), 0 entry-major ([r * run_stride + i])
    uint32_t *busy_list;         // ... their indices (k_walk_runs appends, counters[20] counts)
    uint32_t wave_runs;          // 1: items with more than wave_min windows are left to k_runs_wave (a wave per item, a lane per window)
    uint32_t *dense_list;        // null, or: k_expand leaves items of more than wave_min events to k_expand_dense, their indices here (counters[23] counts)
    uint32_t relist;             // 1: an item whose window list overflowed is left undone -- the host runs the batch again with longer lists
    uint32_t wave_min;           // >= kWaveRunsMin: a wave per item only pays where the item's one lane would take longer than the waves' turns
    uint32_t poison_quads;       // ceil(max_pattern_length / 16): quads flagged after a lost state
    uint32_t persistent;         // 1: one workgroup per CU, waves pull item batches from counters[1]
    uint32_t wave_items;         // k_scan_anchor: items a wave takes per batch (1..64)
    uint32_t wave_grab;          // persistent mode: batches a wave takes from counters[1] per atomic (>= 1; see engine.cpp)
    uint32_t reserve_cus;        // persistent scan kernels launch this many workgroups fewer than the chip has CUs ("reserve_cus")
    uint32_t zero_counters;      // 1: the item table is the slot's own from the batch before (same uniform shape): nobody zeroed the
                                 // per-item counters -- k_walk_runs writes every item's
};

// one trie depth of the automaton build (k_build_level); all pointers are device pointers
struct BuildArgs {
    const uint32_t *parent;
    const uint8_t *in_byte;
    const uint32_t *own;
    const uint32_t *order;
    const uint64_t *map_keys;
    const uint32_t *map_vals;
    uint64_t map_size;
    uint32_t *fail, *cnt, *dict;
    uint8_t *is_final;
};

// search and replace over a matched batch (k_replace_plan / k_replace_emit)
struct ReplaceArgs {
    const uint8_t *blob;         // haystacks
    const uint64_t *offsets;     // n_haystacks + 1
    uint32_t n_haystacks;
    uint32_t mode;               // 1 normal, 2 lazy
    const uint64_t *prefix;      // per-haystack prefix of the record array
    const uint4 *records;        // {hay, pos, start, pattern}
    const uint32_t *pattern_length;
    const uint64_t *rep_off;     // per pattern: offset << 32 | length of its replacement text, ~0: none
    const uint8_t *rep_bytes;
    uint2 *plan;                 // nominees {end, pattern}, haystack h's from plan[prefix[h]]
    uint64_t *plan_out;          // per kept nominee: where its piece (the text before it + its replacement) begins in the haystack's output
    uint64_t *tail_out;          // per haystack: where the text behind its last nominee begins in its output
    uint32_t *nom_count;         // per haystack
    uint64_t *out_len;           // per haystack
    const uint64_t *out_off;     // per haystack (k_replace_emit)
    uint8_t *out_blob;
};

// k_win.hip: bytes of child lists behind the window table's 65,536 cells (automaton.hpp kWinListBytes: engine.cpp checks they agree)
constexpr uint32_t kWinListBytesDev = 30u * 1024u;

// multi-GPU exchange (engine.cpp acmi_comm_*): at most kMaxRanks ranks
constexpr uint32_t kMaxRanks = 64;

}  // namespace acmi
