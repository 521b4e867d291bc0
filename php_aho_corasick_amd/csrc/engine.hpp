// engine.hpp -- what the engine's translation units share: the handle, device buffers, and the helpers behind the C ABI
// (engine.cpp: lifecycle, device upload, options and statistics; engine_pipeline.cpp: the batch pipeline and the match entry
// points; engine_choice.cpp: the measured kernel choice; engine_comm.cpp: the multi-GPU exchange; engine_replace.cpp: search
// and replace).  Internal: nothing here is part of include/acmi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/acmi.h"
#include "automaton.hpp"
#include "device_types.hpp"
#include "kernels.hpp"

using namespace acmi;

namespace acmi_engine {

extern uint64_t g_dev_allocs;   // device allocations made so far (a batch during whose enqueue one happened is not timed: hipMalloc stalls the host between its launches)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        ++g_dev_allocs;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;  // grow-only workspace with a little slack
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    // Growth while kernels that use the old allocation may still be queued (a second batch begun while the first is in
    // flight): the old block is parked in `retired` and freed when nothing is in flight -- hipFree synchronises the whole
    // device, which would serialise the two batches, and the older batch's arguments keep pointing at live memory.
    hipError_t ensure_keep(size_t bytes, std::vector<void *> &retired) {
        if (bytes <= cap) return hipSuccess;
        ++g_dev_allocs;
        void *fresh = nullptr;
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&fresh, want);
        if (e != hipSuccess) return e;
        if (p) retired.push_back(p);
        p = fresh;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

extern std::mutex g_live_mutex;
extern std::unordered_set<const acmi_handle *> g_live;

}  // namespace acmi_engine

using namespace acmi_engine;

struct acmi_stream {
    std::vector<uint8_t> carry;     // last max_pattern_length-1 bytes fed
    std::vector<uint8_t> buf;       // carry + the chunk being scanned
    std::vector<acmi_record> out;   // records of the last feed, stream offsets
    uint64_t base = 0;              // bytes fed so far
};

struct acmi_handle {
    int init_ok = 0;
    int device_request = -1;
    Automaton aut;
    BuildOptions bopt;

    // Split pattern set (ensure_finalized): a set that is not eligible for the anchor filter as a whole because of a FEW short
    // patterns -- the reference takes any length from 1 to 1,024 B in one trie (ahocorasick.c:101-105) -- is matched as two
    // automata over the same batch: part[0] the accepted patterns of >= 7 B (the filter), part[1] the rest (the table walk),
    // both with the caller-order pattern indices of this handle; their match arrays are merged per haystack (k_merge_parts).
    acmi_handle *part[2] = {nullptr, nullptr};
    bool is_part = false;
    int split_opt = -1;   // -1 automatic, 0 never, 1 whenever there are patterns on both sides of 7 B
    std::vector<std::pair<std::string, int64_t>> option_log;   // every option set on this handle: replayed on the parts
    DevBuf d_merge_counts, d_merge_sync;
    // (part 1 of a split set) the batch being begun may merge part 0's match arrays into the parent's result buffer as it writes
    // its own records (k_walk_tiles, lane-list variant): set by the parent around the part's begin
    struct MergeInto { const uint64_t *prefix = nullptr; const void *records = nullptr; void *out = nullptr; uint64_t cap = 0; } merge_into;

    // options
    bool dense_output = false;   // the last batch produced a record per two bytes of text or more (which drain: refill_mode)
    uint32_t run_auto = 0;       // run-list entries per item the data has asked for so far
    uint32_t sparse_slab_batches = 0;   // ... likewise for the event slabs
    uint32_t sparse_batches = 0; // filter batches in a row whose longest window list fitted the short form (the long lists are given up after eight)
    bool expand_armed = false;   // the last batch averaged four records a work item or more: k_expand_dense is enqueued with the next
    uint32_t anchor_chunk_shift = 0;   // the anchor filter's work items are 8 KiB >> this (learned from the windows per item; "anchor_chunk_shift")
    bool wave_armed = false;     // the last filter batch had items with more than a few candidate windows: k_runs_wave is enqueued with the next
    int win_walk = 1;            // "win_walk": kernel 4 over the two-byte window table where the set has one (k_walk_win), 0 = k_walk_items
    int relist_opt = 1;          // "relist": 0 = an item whose window list overflows is walked whole at once (no second run of the batch)
    uint32_t relist_reruns = 0;  // batches run again for longer window lists ("relist_reruns")
    int wave_runs_opt = 1;       // "wave_runs": 0 never launches k_runs_wave (every item is k_walk_runs')
    uint32_t busy_items = 0, max_windows = 0, serial_items = 0;   // of the last filter batch ("busy_items", "max_windows", "serial_items")
    bool wave_used = false;      // ... which ran k_runs_wave ("wave_runs_used")
    uint32_t slab_auto = 0;      // slots per item the data has asked for so far (automatic slab size, enqueue_batch)
    uint32_t slab_events = 16;   // (8 until round 3: an item with a ninth match went to the drain, one with a ninth candidate was walked whole)
    uint32_t run_cap = 16;
    bool slab_set = false;     // "slab_events" set by the caller; else learned from the data (slab_auto: 16 / 32 .. 256 slots)
    bool run_cap_set = false;  // by the caller; else learned (run_auto: 16 .. 64 entries); the dense anchor form takes 256 window starts
    uint64_t chunk_opt = 0;
    int kernel = 3;  // 0 plain global table, 1 LDS-tile walk, 2 anchor filter (if eligible), 3 automatic, 4 whole-item walk, 6 tile walk (single pass, ordered output)
    int tile_w = 0, tile_p = 0;   // k_walk_tiles: bytes per lane / lanes per item (0: tiles_geometry chooses)
    int reserve_cus = 0;          // compute units the persistent scan kernels leave free (a collective's kernels need some while a scan runs)
    int chain_walk = -1;          // kernel 6 over the chain-compressed LDS image where the set has one: -1 / 1 yes, 0 never (k_walk_tiles)
    uint32_t chain_finals = 0;    // final indices of the chain image (k_chain_emit's LDS table)
    bool chain_used = false;      // ... the last batch did ("chain_walk_used")
    int rerun_kernel = -1;        // >= 0 while a batch is enqueued AGAIN: the kernel of its first run (enqueue_batch takes it, no probe, nothing learned)
    int part_kernel = -1;         // split sets: the kernel forced on the SHORT part (-1: its own automatic choice)
    uint64_t max_result_bytes = 0;   // 0: unbounded; else a batch hands out at most this much (ACMI_E_PARTIAL + "resume_haystack")
    uint64_t resume_haystack = 0;    // after ACMI_E_PARTIAL: the first haystack that is NOT in the result
    // automatic mode: a batch on which the fast kernel lost (most bytes had to be walked exactly anyway) moves the
    // handle to the slower, safer kernel -- for `*_left` batches, then the fast one is probed again; every further
    // loss doubles the span (16 .. 1024 batches), a win resets it.  Batches under kAutoMinBytes decide nothing.
    // The drain of items whose events overflowed their slab (k_refill_wave) is a launch of its own, ~ 8 us with its kernel
    // boundary even when there is nothing to drain: it is enqueued with the pipeline only while batches need it (the batch
    // before did, or the pattern set is a dense one); a batch that needs it unexpectedly gets it at its end.
    bool refill_armed = false;
    // Automatic mode (kernel = 3) is a MEASURED choice (choose_strategy / learn_strategy below): per class of batch (findAll x
    // size) the handle keeps what a byte cost on every scan strategy it has run -- from the two events every batch records
    // around its pipeline -- runs the cheapest, and probes an alternative when the current one costs more than 1.5 x what that
    // alternative is known (or, untried, expected) to cost, each alternative on its own back-off (16, 32 .. 1,024 batches after a
    // lost probe).  Round 3 had thresholds here (bytes walked, matches per byte, candidates per byte), each found as a 5-10 x cliff.
    enum Strategy { kStratAnchor = 0, kStratLds = 1, kStratTiles = 2, kStratItems = 3, kStratCount = 4 };
    struct ChoiceClass {
        float cost[kStratCount] = {0, 0, 0, 0};      // ms per GiB of haystack bytes: the lower of its last two batches (0: never run)
        float last[kStratCount] = {0, 0, 0, 0};      // ... what its last batch measured
        float density = 0;                           // records per byte of the last batch measured
        uint32_t seen = 0;                           // batches in a row the strategy in use has been measured on (probes of others wait for two)
        uint32_t wait[kStratCount] = {0, 0, 0, 0};   // batches until this alternative may be probed again
        uint32_t span[kStratCount] = {16, 16, 16, 16};   // its back-off after the next lost probe
        int cur = -1;                                // strategy in use (-1: not seeded yet)
        uint32_t pending = 0;                        // bit s: probe strategy s with one of the next batches
    };
    ChoiceClass choice[2][8];   // [findAll][size class | short haystacks << 2]
    uint32_t probes = 0, switches = 0;   // statistics ("choice_probes", "choice_switches")
    uint32_t merged_batches = 0;         // split sets: batches whose records the short part's walk wrote merged ("merged_batches")
    int ilp = 1;
    int group_quads = 8;
    int small_path = 1;   // 1: host batches of <= 64 KiB / <= 1024 haystacks take the one-launch kernel
    uint8_t *pin_small = nullptr;  // pinned host memory of the small path: [summary | prefix | records | offsets | bytes]
    uint8_t *pin_small_dev = nullptr;   // its device address
    uint64_t small_seq = 0;        // calls of the small path so far: the kernel stores it into the summary when it is done
    int wave_items = 0;   // 0 automatic, else 1..64 items per wave batch of the anchor kernel
    int persistent = -1;  // -1 automatic, 0 one workgroup per 1024 x ILP items, 1 persistent workgroups
    int tiny_rows = 1;    // anchor kernel: 1 (default) haystacks of <= 1 KiB on average take the variant that packs one-row items
    int short_rows = -1;  // anchor kernel variant that skips the rows a task lacks: -1 automatic, 0 never, 1 always
    int walk_group = 4;   // k_walk_items: 16-byte haystack blocks a lane requests back to back (1, 4 or 8 = a whole line)
    int early_stop = 1;   // walks of anchor candidates end once no pattern begun at the hit window can be alive: 0 off,
                          // 1 when the automaton can fall back to shallow states at all (default), 2 always
    int refill = 0;       // 0 automatic, 1 one thread per overflowed item (k_refill), 2 one wave (k_refill_wave)
    int profile = 0;
    int verify = 1;            // 1: anchor candidates whose q-gram names one pattern are decided by comparing its text
    int dense_hot = -1;        // dense anchor form: the scan variant that works out every step's hits branch-free: -1 by the expected hit rate, 0 never, 1 always
    int device_build = -1;     // failure links + outputs of the build on the device: -1 from 200,000 trie nodes, 0 never, 1 always
    int keep_host_tables = 0;  // 1: the host copies of table / tile / records stay after the upload (acmi_debug_tables)

    // device state
    bool dev_ready = false;
    bool dev_touched = false;  // some device resource may exist (ensure_device got as far as picking the device)
    int dev_id = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    DevAutomaton dA{};
    DevBuf d_table, d_lds, d_class, d_own, d_dict, d_cnt, d_plen, d_anchor, d_bitmap, d_out_off, d_out_list, d_out_head, d_cold_rec, d_cold_edges, d_chain_byte, d_chain_run, d_win_table, d_win_state, d_win_hash, d_chain_img, d_chain_head, d_chain_orig, d_chain_frank, d_chain_fhead, d_slot_pair, d_pairs, d_pat_bytes, d_dense_img;

    // workspace (grow-only)
    DevBuf d_blob, d_off, d_rec_base, d_slab, d_ovf, d_counters, d_runs, d_run_count, d_first, d_busy, d_dense;
    DevBuf d_tile_queue, d_tile_sync;   // k_walk_tiles: the waves' event queues; {ticket, look-back words}
    DevBuf d_chain_log, d_chain_tiles;  // k_walk_chain: the waves' event logs; per tile {log offset | events per lane}
    uint32_t chain_log_div = 32;        // ... the logs hold an event per this many bytes of text (learned: 32 -> 4 -> 1)
    uint32_t chain_log_reruns = 0;      // batches run again because they overflowed the logs ("chain_log_reruns")
    // uniform batches (acmi_match_batch_device_begin_uniform): the offsets array the library makes itself, one per slot, and the
    // geometry it holds; uniform_len != 0 while such a batch is being enqueued
    DevBuf d_uni_off[2];
    uint64_t uni_nh[2] = {0, 0}, uni_len[2] = {0, 0};
    // the item table a slot holds (uniform batches through the filter: written once, not per batch): what it was made from
    // (the kept table stands only while k_walk_runs -- which zeroes every item's counters itself -- is the one kernel that touches them
    // before the expansion: the kernel is part of the key, tests/test_gpu_variants.py compares kept with fresh tables)
    struct ItemKey { uint64_t nh = 0, len = 0, max_items = 0; uint32_t chunk = 0, warm = 0, align = 0, blob_mis = 0; int kernel = 0; const void *rec = nullptr, *hay = nullptr, *base = nullptr, *zero = nullptr; bool valid = false;
        bool operator==(const ItemKey &o) const { return kernel == o.kernel && nh == o.nh && len == o.len && max_items == o.max_items && chunk == o.chunk && warm == o.warm && align == o.align && blob_mis == o.blob_mis && rec == o.rec && hay == o.hay && base == o.base && zero == o.zero; } };
    ItemKey item_key[2];
    uint64_t item_tables_kept = 0;   // batches that reused their slot's item table (acmi_get_stat "item_tables_kept")
    uint64_t uniform_len = 0;
    // The item table of a batch (item_base, item -> haystack, packed item records, zeroed per-item counters, k_items' own sync
    // words) exists once per slot: k_items needs nothing from the batch before, so with "items_ahead" it runs on a side stream
    // while that batch's scan is still on the chip, and the pipeline proper waits for its event.
    DevBuf d_item_base[2], d_item_hay[2], d_item_rec[2], d_zero[2], d_isync[2];
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_items[2] = {nullptr, nullptr};
    hipEvent_t ev_scanned[2] = {nullptr, nullptr};   // behind a batch's scan kernel: the NEXT batch's k_items waits for it (not for the batch)
    int items_ahead = 0;
    bool isync_clean[2] = {false, false};   // slot's k_items sync words are zero (the batch before zeroed them in its k_expand)
    uint64_t *pinned = nullptr;  // host copy of the device counters blocks (2 x 128 B, one per batch in flight), written by k_expand through pinned_dev
    uint32_t *pinned_dev = nullptr;

    // last result
    uint64_t last_total = 0;
    uint32_t last_nh = 0;
    bool have_result = false;
    std::vector<acmi_record> h_records;
    std::vector<uint64_t> h_prefix;
    // search and replace
    DevBuf d_rep_off, d_rep_bytes, d_plan, d_plan_out, d_tail_out, d_nom_count, d_out_len, d_rep_out_off, d_rep_out, d_rep_sync;
    uint64_t rep_cap = 0;  // bytes of replace output the largest batch so far produced: the optimistic capacity
    bool rep_uploaded = false;
    std::vector<uint8_t> h_rep_blob;
    std::vector<uint64_t> h_rep_off;

    // Batches whose pipeline is enqueued but not yet completed (acmi_match_batch_device_begin/_end): up to TWO, ended in
    // the order they were begun.  The second one's kernels are already queued on the stream when the first one's last
    // kernel retires, so the device does not idle through the host's turn-around (stream wake-up, the caller, the next
    // begin's launches: ~35 us of a 0.8 ms step).  Each slot owns what the host reads or hands out after the end --
    // result buffers, the pinned summary, the events -- while the intermediate buffers are shared: the pipelines run
    // one after the other on the stream.  (Only the rare re-expansion of a batch whose records outgrew the optimistic
    // capacity needs intermediates that a later begin has overwritten: that batch is then run again as a whole.)
    struct FlySlot {
        uint64_t seq = 0;              // 1-based begin number
        BatchArgs args{};
        uint64_t max_items = 0, blob_bytes = 0;
        const uint8_t *d_blob = nullptr;
        const uint64_t *d_off = nullptr;
        size_t nh = 0;
        int find_all = 1;
        bool trivial = false;
        uint32_t *sync_next = nullptr;
        uint32_t sync_words = 0;
        int kernel_used = 0;
        uint64_t chunk_used = 0;
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipStream_t stream = nullptr;  // the stream this batch was enqueued on (acmi_set_stream is refused while batches are in flight)
        uint32_t anchor_variant = 0;   // template of k_scan_anchor that ran (acmi_get_stat "anchor_variant")
        uint32_t slab_used = 0;        // event slots per item of this batch
        uint64_t est_items = 0;
        uint32_t run_used = 0, run_max = 0;   // run-list entries per item of this batch; what it could have had
        uint32_t tile_w = 0, tile_p = 0;      // k_walk_tiles geometry of this batch
        int strategy = -1, choice_class = 0;  // automatic mode: the strategy this batch ran on (-1: the caller fixed the kernel)
        bool probe = false;                   // ... as a probe of an alternative
        bool timed = false;                   // nothing was allocated while it was enqueued: its events say what it cost
        uint64_t out_cap = 0;                 // records the batch's kernels were allowed to write
        bool chain = false;                   // kernel 6 ran as k_walk_chain (the whole automaton in LDS, chain-compressed)
        ChainBuffers chain_buf{};             // ... with these buffers (k_chain_emit again into a grown record buffer)
        bool merged = false;                  // (part 1 of a split set) this batch wrote the MERGED records into the parent's buffer
        bool reran = false;                   // the end had to complete or repeat the pipeline: what ran behind it on the stream saw unfinished results
        bool relist = false;                  // items whose window list overflows are left undone and the batch is run again with longer lists
        bool win_lists = false, wave_runs = false;   // the filter handed over window lists; k_runs_wave was enqueued behind k_walk_runs
        bool refill_launched = false;  // the drain of overflowed items was enqueued with the pipeline (else: only if the batch turns out to need it)
        DevBuf d_prefix, d_out;        // valid from this batch's end until the second begin after its own
    };
    FlySlot fly[2];
    uint64_t begin_seq = 0, end_seq = 0;  // batches begun / ended; begin_seq - end_seq (0..2) are in flight
    uint64_t enqueue_seq = 0;             // seq of the batch whose pipeline was enqueued last (it owns the intermediates)
    size_t out_want = 0;                  // bytes of records the largest batch so far produced: both slots' optimistic capacity
    int cur = 0;                          // slot of the last completed result (have_result)
    DevBuf &d_prefix() { return fly[cur].d_prefix; }
    DevBuf &d_out() { return fly[cur].d_out; }
    DevBuf d_sync[2];          // counters, scan tickets and status words: two copies that take turns (run_batch_begin)
    uint32_t sync_tiles = 0, sync_turn = 0;
    std::vector<void *> retired;   // device blocks outgrown while a batch was in flight: freed when none is (DevBuf::ensure_keep)

    // statistics of the last call
    double scan_ms = 0, walk_ms = 0, total_ms = 0;
    int kernel_used = 0;
    uint64_t items = 0, chunk_used = 0;
    uint32_t overflow_items = 0;
    uint32_t anchor_variant = 0;  // k_scan_anchor template of the last batch: QW (bits 0-2) | BIG 8 | SHORT 16 | TINY 32 | DENSE 64; 0: another kernel
    uint32_t tiny_waves = 0;      // wave batches of the last call that took the TINY path (items of one row, eight to a step)
    uint64_t walked_bytes = 0;
    uint64_t fallback_bytes = 0;  // anchor path: bytes of candidate clusters that were walked because verification could not decide

    std::string last_error;
};

namespace acmi_engine {

bool live(const acmi_handle *h);
int fail_hip(acmi_handle *h, hipError_t e, const char *what);

#define HIPCHK(h, call)                                   \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return fail_hip(h, _e, #call); \
    } while (0)

constexpr uint64_t kAutoMinBytes = 1ull << 20;
constexpr uint32_t kPinWords = 32;  // 32-bit words of a batch's sync block that k_expand mirrors into pinned host memory
constexpr uint64_t kSmallItemBytes = 192;  // automatic mode: average haystack size under which the anchor kernel is not used

hipError_t copy_h2d(void *dst, const void *src, size_t bytes);
int ensure_finalized(acmi_handle *h);
int ensure_device(acmi_handle *h);
int check_ready(acmi_handle *h);

// engine_choice.cpp
int choice_class(uint64_t blob_bytes, size_t nh);
int choose_strategy(acmi_handle *h, acmi_handle::FlySlot &F, uint64_t blob_bytes, size_t nh, int find_all);
void learn_strategy(acmi_handle *h, const acmi_handle::FlySlot &F, double ms_per_gib, double density);

// engine_pipeline.cpp
int run_batch_begin(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh, int find_all);
int run_batch_end(acmi_handle *h);
int run_batch(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh, int find_all);

// No C++ exception may cross the C ABI (the callers are ctypes, cgo-style FFIs and the Zend engine):
// std::bad_alloc from the builder's vectors or the result buffers becomes ACMI_E_NOMEM, anything else
// ACMI_E_DEVICE, with the text kept for acmi_last_error().
template <typename F>
int guarded(const acmi_handle *hc, F &&f) noexcept {
    acmi_handle *h = const_cast<acmi_handle *>(hc);
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { if (live(h)) h->last_error = "out of host memory"; } catch (...) {}
        return ACMI_E_NOMEM;
    } catch (const std::exception &e) {
        try { if (live(h)) h->last_error = std::string("internal error: ") + e.what(); } catch (...) {}
        return ACMI_E_DEVICE;
    } catch (...) {
        return ACMI_E_DEVICE;
    }
}


}  // namespace acmi_engine
