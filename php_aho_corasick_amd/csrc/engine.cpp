// engine.cpp -- handle lifecycle, device memory, batch pipeline and the C ABI (include/acmi.h).
//
// Host-side counterpart of the reference's Zend glue around MultiFast
// (src/php_ahocorasick.c): what PHP_FUNCTION(ahocorasick_match) + the match callback do per
// haystack (:664-746, :542-589) is done here per BATCH: stage haystacks, run the HIP scan,
// drain the device match buffer into {hay,pos,start,pattern} records.  There is no CPU
// matching path in this library: without a usable gfx950 device every match call fails with
// ACMI_E_NO_DEVICE / ACMI_E_DEVICE.
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

namespace {

uint64_t g_dev_allocs = 0;   // device allocations made so far (a batch during whose enqueue one happened is not timed: hipMalloc stalls the host between its launches)

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

std::mutex g_live_mutex;
std::unordered_set<const acmi_handle *> g_live;

}  // namespace

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
    uint32_t slab_auto = 0;      // slots per item the data has asked for so far (automatic slab size, enqueue_batch)
    uint32_t slab_events = 16;   // (8 until round 3: an item with a ninth match went to the drain, one with a ninth candidate was walked whole)
    uint32_t run_cap = 16;
    bool slab_set = false;     // "slab_events" set by the caller; else learned from the data (slab_auto: 16 / 32 .. 256 slots)
    bool run_cap_set = false;  // by the caller; else learned (run_auto: 16 .. 64 entries); the dense anchor form takes 256 window starts
    uint64_t chunk_opt = 0;
    int kernel = 3;  // 0 plain global table, 1 LDS-tile walk, 2 anchor filter (if eligible), 3 automatic, 4 whole-item walk, 6 tile walk (single pass, ordered output)
    int tile_w = 0, tile_p = 0;   // k_walk_tiles: bytes per lane / lanes per item (0: tiles_geometry chooses)
    int reserve_cus = 0;          // compute units the persistent scan kernels leave free (a collective's kernels need some while a scan runs)
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
        float cost[kStratCount] = {0, 0, 0, 0};      // ms per GiB of haystack bytes, moving average of the batches run on it (0: never run)
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
    DevBuf d_table, d_lds, d_class, d_own, d_dict, d_cnt, d_plen, d_anchor, d_bitmap, d_out_off, d_out_list, d_cold_rec, d_cold_edges, d_chain_byte, d_chain_run, d_slot_pair, d_pairs, d_pat_bytes, d_dense_img;

    // workspace (grow-only)
    DevBuf d_blob, d_off, d_rec_base, d_slab, d_ovf, d_counters, d_runs, d_run_count, d_first;
    DevBuf d_tile_queue, d_tile_sync;   // k_walk_tiles: the waves' event queues; {ticket, look-back words}
    // uniform batches (acmi_match_batch_device_begin_uniform): the offsets array the library makes itself, one per slot, and the
    // geometry it holds; uniform_len != 0 while such a batch is being enqueued
    DevBuf d_uni_off[2];
    uint64_t uni_nh[2] = {0, 0}, uni_len[2] = {0, 0};
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
        bool merged = false;                  // (part 1 of a split set) this batch wrote the MERGED records into the parent's buffer
        bool reran = false;                   // the end had to complete or repeat the pipeline: what ran behind it on the stream saw unfinished results
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

namespace {

bool live(const acmi_handle *h) {
    if (!h) return false;
    std::lock_guard<std::mutex> g(g_live_mutex);
    return g_live.count(h) != 0;
}

int fail_hip(acmi_handle *h, hipError_t e, const char *what) {
    h->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? ACMI_E_NOMEM : ACMI_E_DEVICE;
}

#define HIPCHK(h, call)                                   \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return fail_hip(h, _e, #call); \
    } while (0)

constexpr uint64_t kAutoMinBytes = 1ull << 20;
constexpr uint32_t kPinWords = 32;  // 32-bit words of a batch's sync block that k_expand mirrors into pinned host memory
constexpr uint64_t kSmallItemBytes = 192;  // automatic mode: average haystack size under which the anchor kernel is not used

// Synchronous upload of a host array: a plain pageable hipMemcpy (20-55 GB/s on the MI355X box, tools/micro/upload_cost.hip).
// The FIRST copy of a process takes ~ 30 ms more, and ~ 140 ms in a process that has libacmi.so's code object to load:
// the runtime's own start, paid once, by whichever call touches the device first (the device-side build, or the first
// match).  (Registering the source for the duration of the copy was tried: no faster once the runtime is up, and long test
// runs then aborted inside later pageable copies of other libraries from address ranges that had once been registered.)
hipError_t copy_h2d(void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

// Failure links and per-state outputs on the device, one launch per trie depth (k_misc.hip k_build_level).  Used for
// big tries when a gfx950 device is there ("device_build": -1 automatic from 200,000 nodes, 0 never, 1 always);
// anything that goes wrong hands the work back to the host loop -- the results are the same words either way.
bool device_fail_and_outputs(void *ctx, const TrieView &tv, uint32_t *fail, uint8_t *is_final, uint32_t *cnt, uint32_t *dict) {
    acmi_handle *h = static_cast<acmi_handle *>(ctx);
    const bool trace = std::getenv("ACMI_BUILD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[acmi build]   device: %-20s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return false; }
    // The caller's current device is put back on every way out, and the work runs on a private non-blocking stream:
    // a host such as torch is neither serialised against its blocking streams (the NULL stream would be) nor left on
    // another device.
    int caller_dev = -1;
    if (hipGetDevice(&caller_dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    int dev = h->device_request;
    if (dev < 0) dev = caller_dev;
    struct Restore {
        int dev;
        hipStream_t st = nullptr;
        ~Restore() {
            if (st) (void)hipStreamDestroy(st);
            (void)hipSetDevice(dev);
        }
    } restore{caller_dev};
    if (dev >= count || hipSetDevice(dev) != hipSuccess) return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return false;
    if (hipStreamCreateWithFlags(&restore.st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); restore.st = nullptr; return false; }
    hipStream_t st = restore.st;
    lap("runtime + device");
    const size_t n = tv.n_nodes;
    DevBuf d_parent, d_byte, d_own, d_order, d_keys, d_vals, d_fail, d_cnt, d_dict, d_final, d_pos, d_fail_b, d_cnt_b, d_dict_b, d_final_b;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) {
        return b.ensure(bytes) == hipSuccess && copy_h2d(b.p, src, bytes) == hipSuccess;
    };
    bool ok = up(d_parent, tv.parent, n * 4) && up(d_byte, tv.in_byte, n) && up(d_own, tv.own, n * 4) && up(d_order, tv.order, n * 4) &&
              up(d_keys, tv.map_keys, tv.map_size * 8) && up(d_vals, tv.map_vals, tv.map_size * 4) &&
              d_fail.ensure(n * 4) == hipSuccess && d_cnt.ensure(n * 4) == hipSuccess && d_dict.ensure(n * 4) == hipSuccess &&
              d_final.ensure(n) == hipSuccess && d_pos.ensure(n * 4) == hipSuccess && d_fail_b.ensure(n * 4) == hipSuccess &&
              d_cnt_b.ensure(n * 4) == hipSuccess && d_dict_b.ensure(n * 4) == hipSuccess && d_final_b.ensure(n) == hipSuccess;
    if (ok) {
        // the root: no failure link, no output (the level loop starts at depth 1 and reads these)
        ok = hipMemsetAsync(d_fail.p, 0, 4, st) == hipSuccess && hipMemsetAsync(d_cnt.p, 0, 4, st) == hipSuccess &&
             hipMemsetAsync(d_dict.p, 0xFF, 4, st) == hipSuccess && hipMemsetAsync(d_final.p, 0, 1, st) == hipSuccess;
    }
    lap("trie upload (+ first use of the device)");
    if (ok) {
        BuildArgs T{d_parent.as<uint32_t>(), d_byte.as<uint8_t>(), d_own.as<uint32_t>(), d_order.as<uint32_t>(), d_keys.as<uint64_t>(),
                    d_vals.as<uint32_t>(), tv.map_size, d_fail.as<uint32_t>(), d_cnt.as<uint32_t>(), d_dict.as<uint32_t>(), d_final.as<uint8_t>()};
        for (uint32_t d = 1; d < tv.n_levels; ++d) launch_build_level(T, tv.level_off[d], tv.level_off[d + 1], st);
        // results by BFS position, links as positions (what the host goes on with)
        launch_build_by_position(T, uint32_t(n), d_pos.as<uint32_t>(), d_fail_b.as<uint32_t>(), d_cnt_b.as<uint32_t>(), d_dict_b.as<uint32_t>(),
                                 d_final_b.as<uint8_t>(), st);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        lap("level kernels");
        ok = ok && hipMemcpy(fail, d_fail_b.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(cnt, d_cnt_b.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(dict, d_dict_b.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(is_final, d_final_b.p, n, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    lap("download");
    for (DevBuf *b : {&d_parent, &d_byte, &d_own, &d_order, &d_keys, &d_vals, &d_fail, &d_cnt, &d_dict, &d_final, &d_pos, &d_fail_b, &d_cnt_b, &d_dict_b, &d_final_b}) b->release();
    lap("release");
    return ok;
}

int ensure_finalized(acmi_handle *h);

// The two parts of a split set: handles of their own (automaton, device tables, pipeline buffers), built from this handle's
// ACCEPTED patterns -- the others, and those of the other part, are passed with length 0, which add_batch drops while keeping
// the caller-order index (duplicates were settled when this handle took the patterns: first inserted wins across calls).
int build_parts(acmi_handle *h) {
    const std::vector<PatternMeta> &pats = h->aut.patterns();
    const uint8_t *bytes = h->aut.pattern_bytes().data();
    for (int side = 0; side < 2; ++side) {
        acmi_handle *c = new (std::nothrow) acmi_handle();
        if (!c) return ACMI_E_NOMEM;
        c->device_request = h->device_request;
        c->init_ok = 1;
        c->is_part = true;
        {
            std::lock_guard<std::mutex> g(g_live_mutex);
            g_live.insert(c);
        }
        h->part[side] = c;
        for (const auto &kv : h->option_log) (void)acmi_set_option(c, kv.first.c_str(), kv.second);
        std::vector<const uint8_t *> vals(pats.size());
        std::vector<uint32_t> lens(pats.size()), types(pats.size());
        std::vector<int64_t> ids(pats.size());
        std::vector<uint64_t> users(pats.size());
        for (size_t i = 0; i < pats.size(); ++i) {
            const bool mine = pats[i].accepted && ((pats[i].length >= 7) == (side == 0));
            vals[i] = bytes + pats[i].byte_off;
            lens[i] = mine ? pats[i].length : 0u;
            types[i] = pats[i].id_type;
            ids[i] = pats[i].id_number;
            users[i] = pats[i].user;
        }
        c->aut.add_batch(vals.data(), lens.data(), types.data(), ids.data(), users.data(), pats.size());
        const int rc = ensure_finalized(c);
        if (rc < 0) { h->last_error = c->last_error; return rc; }
        if (side == 1 && h->part_kernel >= 0) (void)acmi_set_option(c, "kernel", h->part_kernel);
    }
    return ACMI_OK;
}

// A split that could not be built leaves no half of itself behind: the handle goes on as the single automaton it is.
void drop_parts(acmi_handle *h) {
    for (acmi_handle *&c : h->part)
        if (c) { (void)acmi_destroy(c); c = nullptr; }
}

int ensure_finalized(acmi_handle *h) {
    if (h->aut.finalized()) return 0;
    std::string err;
    BuildAccel accel;
    accel.ctx = h;
    accel.fail_and_outputs = device_fail_and_outputs;
    const bool use_device = h->device_build > 0 || (h->device_build < 0 && h->aut.node_count() >= 200000);
    if (!h->aut.finalize(h->bopt, &err, use_device ? &accel : nullptr)) {
        h->last_error = err;
        return ACMI_E_TOO_LARGE;
    }
    // Automatic kernel choice, seeded from the automaton instead of from a first slow batch: when uniform text over
    // the patterns' alphabet would flag more positions of an 8 KiB work item than its run list holds, the optimistic
    // LDS-tile pass ends up walking whole items anyway (keyword lists, dense sets) -- start on the exact walk.  The
    // usual probing applies from there (the fast kernel gets its chance after `exact_left` batches).
    const FlatAutomaton &fa = h->aut.flat();
    // (not for the short part of a split set: its few patterns were split off because they are unlike the rest -- the text is
    // expected to look like the LONG patterns, and the estimate "text over the short patterns' own alphabet" says nothing; the
    // first batch's statistics decide)
    if (fa.est_flag_rate * 8192.0 > double(h->run_cap) && !h->is_part) {
        h->refill_armed = true;   // (dense sets: the first batch will want its overflowed items drained)
        h->dense_output = fa.est_flag_rate >= 0.5;   // (an event every other byte of the patterns' own alphabet: the wave drain from the first batch on)
    }
    // Split?  The set as a whole cannot use the anchor filter, but without its few short patterns it could.
    if (!h->is_part && h->split_opt != 0 && fa.anchor_q == 0 && h->kernel == 3) {
        size_t n_long = 0, n_short = 0;
        for (const PatternMeta &m : h->aut.patterns())
            if (m.accepted) (m.length >= 7 ? n_long : n_short)++;
        const bool want = n_long > 0 && n_short > 0 && (h->split_opt > 0 || (n_short <= 256 && n_long >= 16 * n_short));
        if (want) {
            int rc;
            try { rc = build_parts(h); } catch (...) { drop_parts(h); throw; }
            // (out of memory, a part that cannot be finalized: matched unsplit, through the table walk -- slower, same records)
            if (rc < 0) drop_parts(h);
        }
    }
    return 1;
}

int ensure_device(acmi_handle *h) {
    if (h->dev_ready) {
        HIPCHK(h, hipSetDevice(h->dev_id));
        return ACMI_OK;
    }
    if (h->part[0]) {   // a split set: the parts hold the device tables; this handle the stream, the merged results
        for (acmi_handle *c : h->part) {
            const int rc = ensure_device(c);
            if (rc < 0) { h->last_error = c->last_error; return rc; }
        }
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        h->last_error = "no HIP device visible (this library has no CPU matching path)";
        return ACMI_E_NO_DEVICE;
    }
    int dev = h->device_request;
    if (dev < 0) HIPCHK(h, hipGetDevice(&dev));
    if (dev >= count) {
        h->last_error = "device ordinal out of range";
        return ACMI_E_NO_DEVICE;
    }
    HIPCHK(h, hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        h->last_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return ACMI_E_NO_DEVICE;
    }
    h->dev_id = dev;
    h->dev_touched = true;
    // a call that fails further down (e.g. out of device memory for a large table) leaves dev_ready false and is
    // retried by the next match call: create every resource once only
    if (!h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    if (!h->side_stream) {
        // lowest priority: its workgroups (k_items of the NEXT batch) take the compute units the current batch's scan leaves
        // free at its tail, not the ones it is about to use -- at equal priority the scan got 2.5 % slower and nothing was won
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIPCHK(h, hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, least));
    }
    for (auto &evt : h->ev_items)
        if (!evt) HIPCHK(h, hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    for (auto &evt : h->ev_scanned)
        if (!evt) HIPCHK(h, hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    if (!h->stream) h->stream = h->own_stream;
    for (auto &evt : h->ev)
        if (!evt) HIPCHK(h, hipEventCreate(&evt));
    for (auto &slot : h->fly)
        for (auto &evt : slot.ev)
            if (!evt) HIPCHK(h, hipEventCreate(&evt));
    if (!h->pinned) {
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->pinned), 2 * kPinWords * 4, hipHostMallocMapped));
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->pinned, 0));
        h->pinned_dev = static_cast<uint32_t *>(dp);
    }

    if (h->part[0]) {
        // (of its own automaton the handle needs the pattern lengths only: search and replace)
        const FlatAutomaton &fp = h->aut.flat();
        const size_t np4 = std::max<size_t>(4, fp.pattern_length.size() * 4);
        HIPCHK(h, h->d_plen.ensure(np4));
        if (!fp.pattern_length.empty()) HIPCHK(h, copy_h2d(h->d_plen.p, fp.pattern_length.data(), fp.pattern_length.size() * 4));
        h->dA = DevAutomaton{};
        h->dA.pattern_length = h->d_plen.as<uint32_t>();
        h->dA.max_pattern_length = fp.max_pattern_length;
        for (acmi_handle *c : h->part) c->stream = h->stream;
        h->dev_ready = true;
        if (!h->keep_host_tables) h->aut.release_host_tables();
        return ACMI_OK;
    }
    // upload the flat automaton
    const FlatAutomaton &fa = h->aut.flat();
    const bool table_mode = fa.class_mode == kClassTable;
    uint16_t class_off[256];
    for (int b = 0; b < 256; ++b) class_off[b] = uint16_t(fa.class_of_byte[b] * fa.entry_bytes);
    std::vector<uint8_t> image;
    if (table_mode) image.insert(image.end(), reinterpret_cast<uint8_t *>(class_off), reinterpret_cast<uint8_t *>(class_off) + 512);
    image.insert(image.end(), fa.lds_image.begin(), fa.lds_image.end());
    while (image.size() % 16) image.push_back(0);

    HIPCHK(h, h->d_table.ensure(fa.table.size()));
    HIPCHK(h, copy_h2d(h->d_table.p, fa.table.data(), fa.table.size()));
    HIPCHK(h, h->d_lds.ensure(image.size()));
    HIPCHK(h, copy_h2d(h->d_lds.p, image.data(), image.size()));
    HIPCHK(h, h->d_class.ensure(512));
    HIPCHK(h, copy_h2d(h->d_class.p, class_off, 512));
    const size_t ns4 = size_t(fa.n_states) * 4;
    HIPCHK(h, h->d_own.ensure(ns4));
    HIPCHK(h, copy_h2d(h->d_own.p, fa.own_pattern.data(), ns4));
    HIPCHK(h, h->d_dict.ensure(ns4));
    HIPCHK(h, copy_h2d(h->d_dict.p, fa.dict_link.data(), ns4));
    HIPCHK(h, h->d_cnt.ensure(ns4));
    HIPCHK(h, copy_h2d(h->d_cnt.p, fa.out_count.data(), ns4));
    const size_t np4 = std::max<size_t>(4, fa.pattern_length.size() * 4);
    HIPCHK(h, h->d_plen.ensure(np4));
    if (!fa.pattern_length.empty())
        HIPCHK(h, copy_h2d(h->d_plen.p, fa.pattern_length.data(), fa.pattern_length.size() * 4));

    if (!fa.out_off.empty()) {
        HIPCHK(h, h->d_out_off.ensure(ns4));
        HIPCHK(h, copy_h2d(h->d_out_off.p, fa.out_off.data(), ns4));
        HIPCHK(h, h->d_out_list.ensure(std::max<size_t>(8, fa.out_list.size() * 4)));
        if (!fa.out_list.empty())
            HIPCHK(h, copy_h2d(h->d_out_list.p, fa.out_list.data(), fa.out_list.size() * 4));
    }

    if (fa.compact) {
        HIPCHK(h, h->d_cold_rec.ensure(std::max<size_t>(8, fa.cold_rec.size() * 4)));
        HIPCHK(h, copy_h2d(h->d_cold_rec.p, fa.cold_rec.data(), fa.cold_rec.size() * 4));
        HIPCHK(h, h->d_cold_edges.ensure(std::max<size_t>(8, fa.cold_edges.size() * 4)));
        if (!fa.cold_edges.empty())
            HIPCHK(h, copy_h2d(h->d_cold_edges.p, fa.cold_edges.data(), fa.cold_edges.size() * 4));
        if (!fa.chain_byte.empty()) {
            HIPCHK(h, h->d_chain_byte.ensure(fa.chain_byte.size()));
            HIPCHK(h, copy_h2d(h->d_chain_byte.p, fa.chain_byte.data(), fa.chain_byte.size()));
            HIPCHK(h, h->d_chain_run.ensure(fa.chain_run.size()));
            HIPCHK(h, copy_h2d(h->d_chain_run.p, fa.chain_run.data(), fa.chain_run.size()));
        }
    }

    if (fa.anchor_q) {
        HIPCHK(h, h->d_anchor.ensure(fa.anchor_table.size() * 2));
        HIPCHK(h, copy_h2d(h->d_anchor.p, fa.anchor_table.data(), fa.anchor_table.size() * 2));
        if (fa.anchor_mode == 2) {
            HIPCHK(h, h->d_bitmap.ensure(fa.anchor_bitmap.size() * 4));
            HIPCHK(h, copy_h2d(h->d_bitmap.p, fa.anchor_bitmap.data(), fa.anchor_bitmap.size() * 4));
        }
    }

    if (fa.anchor_dense && !fa.dense_image.empty() && h->verify) {
        HIPCHK(h, h->d_dense_img.ensure(fa.dense_image.size() * 4));
        HIPCHK(h, copy_h2d(h->d_dense_img.p, fa.dense_image.data(), fa.dense_image.size() * 4));
        h->dA.dense_image = h->d_dense_img.as<uint32_t>();
        h->dA.dense_image_words = uint32_t(fa.dense_image.size());
    } else {
        h->dA.dense_image = nullptr;
        h->dA.dense_image_words = 0;
    }
    const bool verify = fa.anchor_q && h->verify && !fa.anchor_pairs.empty() && !fa.anchor_slot_pair.empty();
    if (verify) {
        const std::vector<uint8_t> &pb = h->aut.pattern_bytes();
        HIPCHK(h, h->d_pat_bytes.ensure(pb.size() + 32));
        HIPCHK(h, hipMemset(h->d_pat_bytes.p, 0, pb.size() + 32));
        HIPCHK(h, copy_h2d(h->d_pat_bytes.p, pb.data(), pb.size()));
        HIPCHK(h, h->d_slot_pair.ensure(fa.anchor_slot_pair.size() * 4));
        HIPCHK(h, copy_h2d(h->d_slot_pair.p, fa.anchor_slot_pair.data(), fa.anchor_slot_pair.size() * 4));
        HIPCHK(h, h->d_pairs.ensure(fa.anchor_pairs.size() * 4));
        HIPCHK(h, copy_h2d(h->d_pairs.p, fa.anchor_pairs.data(), fa.anchor_pairs.size() * 4));
    }

    DevAutomaton &A = h->dA;
    A.anchor_slot_pair = verify ? h->d_slot_pair.as<uint32_t>() : nullptr;
    A.anchor_pairs = verify ? h->d_pairs.as<uint4>() : nullptr;
    A.pattern_bytes = verify ? h->d_pat_bytes.as<uint8_t>() : nullptr;
    A.anchor_table = fa.anchor_q ? h->d_anchor.p : nullptr;
    A.anchor_q = fa.anchor_q;
    A.anchor_mode = fa.anchor_mode;
    A.anchor_bitmap = fa.anchor_mode == 2 ? h->d_bitmap.as<uint32_t>() : nullptr;
    A.anchor_bucket_mask = fa.anchor_mode == 2 ? (1u << fa.anchor_bucket_bits) - 1u : 0u;
    A.anchor_choice = fa.anchor_choice;
    A.anchor_dense = fa.anchor_dense;
    A.anchor_slots = fa.anchor_slots;
    A.anchor_dense_hot = (h->dense_hot > 0 || (h->dense_hot < 0 && fa.anchor_dense_rate * 8192.0 > 12.0)) ? 1u : 0u;
    A.anchor_bucket_shift = fa.anchor_bucket_shift;
    A.anchor_mul = fa.anchor_mul;
    A.min_pattern_length = fa.min_pattern_length;
    A.max_pattern_length = fa.max_pattern_length;
    A.table = h->d_table.p;
    A.lds_image = h->d_lds.p;
    A.class_off = h->d_class.as<uint16_t>();
    A.own_pattern = h->d_own.as<uint32_t>();
    A.dict_link = h->d_dict.as<uint32_t>();
    A.out_count = h->d_cnt.as<uint32_t>();
    A.pattern_length = h->d_plen.as<uint32_t>();
    A.out_off = fa.out_off.empty() ? nullptr : h->d_out_off.as<uint32_t>();
    A.out_list = fa.out_off.empty() ? nullptr : h->d_out_list.as<uint2>();
    A.max_out_count = fa.max_out_count;
    A.compact = fa.compact;
    A.dense_nf_end = fa.dense_nf_end;
    A.dense_f_end = fa.dense_f_end;
    A.cold_rec = fa.compact ? h->d_cold_rec.as<uint2>() : nullptr;
    A.cold_edges = fa.compact ? h->d_cold_edges.as<uint32_t>() : nullptr;
    A.chain_lo = fa.chain_lo;
    A.chain_byte = fa.chain_byte.empty() ? nullptr : h->d_chain_byte.as<uint8_t>();
    A.chain_run = fa.chain_byte.empty() ? nullptr : h->d_chain_run.as<uint8_t>();
    A.pad16 = h->d_class.as<uint8_t>();  // (512 bytes, always present)
    A.shallow_hot = fa.shallow_hot;
    A.shallow_cold_end = fa.shallow_cold_end;
    A.early_stop = (h->early_stop == 2 || (h->early_stop == 1 && fa.early_stop_useful)) ? 1u : 0u;
    // identity byte classes (all 256 byte values occur in patterns); 255 used values also give 256 columns, but
    // with column 0 = "other" -- so test the map itself
    A.class_identity = (fa.class_mode == kClassTable && fa.n_columns == 256) ? 1u : 0u;
    for (int b = 0; b < 256 && A.class_identity; ++b)
        if (fa.class_of_byte[b] != b) A.class_identity = 0u;
    A.n_states = fa.n_states;
    A.n_columns = fa.n_columns;
    A.row_bytes = fa.n_columns * fa.entry_bytes;
    A.f1 = fa.f1;
    A.trap = fa.trap;
    A.f2 = fa.f2;
    A.lds_image_bytes = uint32_t(image.size());
    A.tile_offset = table_mode ? 512u : 0u;
    A.swar_shift = fa.swar_shift;
    A.swar_expect_lo = fa.swar_expect_lo;
    A.swar_expect_hi = fa.swar_expect_hi;
    A.swar_other_off = fa.swar_other * fa.entry_bytes;
    A.other_byte = 0x100u;
    if (fa.n_used_bytes < 256) {
        const uint32_t other_class = fa.class_mode == kClassSwar3 ? fa.swar_other : 0u;
        for (int b = 255; b >= 0; --b)
            if (fa.class_of_byte[b] == other_class) A.other_byte = uint32_t(b);
    }
    h->dev_ready = true;
    // the device has everything now; the host copies (up to GiBs for a dense table) are only of interest to
    // acmi_debug_tables
    if (!h->keep_host_tables) h->aut.release_host_tables();
    return ACMI_OK;
}

// Which drain takes the items whose events overflowed their slabs: the caller's choice ("refill" 1 / 2), else the whole-item walk
// again in its record-writing form for the batches that kernel scanned (mode 3), else launch_refill's own choice.
// (not when the text answers with a record per two bytes or more -- a^1..a^16, every string over {a,b}: thousands of records
// per item, which only the wave drain writes as whole rows; `dense_output` is what the batch before produced)
int refill_mode(const acmi_handle *h, int kernel) { return h->refill != 0 ? h->refill : (kernel == 4 && !h->dense_output ? 3 : 0); }

uint32_t choose_chunk(const acmi_handle *h, uint64_t blob_bytes, size_t nh, int kernel) {
    const uint32_t warm = h->aut.flat().max_pattern_length ? h->aut.flat().max_pattern_length - 1 : 0;
    if (h->chunk_opt) {
        uint64_t c = (h->chunk_opt + 15) & ~uint64_t(15);
        return uint32_t(std::min<uint64_t>(std::max<uint64_t>(c, 16), 1u << 30));
    }
    // enough work items to fill 256 CUs x 1024 lanes x 2 streams, but never so small that the
    // (Lmax-1)-byte warm-up of a chunk costs more than 1/8 of its work
    // (the exact whole-item walk, kernel 4, has every lane slot of the chip take ONE item and no persistent mode: four rounds of
    // smaller items (not under 1 KiB), so that an item count just above a multiple of the slots costs a quarter of the
    // kernel, not all of it again)
    const uint64_t target_items = 256ull * 1024 * 2 * (kernel == 4 ? 4 : 1);
    uint64_t c0 = (blob_bytes + target_items - 1) / target_items;
    uint64_t cmin = std::max<uint64_t>(kernel == 4 ? 1024 : 256, (uint64_t(warm) * 8 + 127) & ~uint64_t(127));
    // ... and a batch with fewer 1-KiB items than the chip has lanes (under 256 MiB) takes them smaller still: the walk is a
    // chain of dependent LDS reads per lane and only more lanes hide it (32 MiB of a^1..a^16: 32 K items on 256 K lanes)
    if (kernel == 4 && blob_bytes < 1024ull * 256 * 1024) {
        const uint64_t per_lane = (blob_bytes / (256ull * 1024) + 127) & ~uint64_t(127);
        cmin = std::max<uint64_t>(std::max<uint64_t>(256, per_lane), (uint64_t(warm) * 8 + 127) & ~uint64_t(127));
    }
    // the anchor kernel streams an item 8 KiB (one wave step) at a time: smaller items idle its lanes
    if (kernel == 2) {
        cmin = std::max<uint64_t>(cmin, 8192);
        // ... and where haystacks are longer than that (they get chunked), a chunk inside its haystack must fit one step
        // WITH its warm-up, which k_items rounds up to a 128-byte boundary (align_warm): 8 KiB chunks were 513-514 quads,
        // i.e. a second, almost empty step per item (24-KiB haystacks: scan kernel 0.83 instead of 0.67 ms per 4 GiB)
        if (nh && blob_bytes / nh > 8192 && uint64_t(warm) + 32 < 4096) {
            const uint64_t steps = std::max<uint64_t>(1, (c0 + 8191) / 8192);  // whole steps per chunk, less the warm-up's room
            const uint64_t c = steps * 8192 - ((uint64_t(warm) + 128 + 127) & ~uint64_t(127));  // (warm-up rounded to a whole line)
            return uint32_t(std::min<uint64_t>(c, 1u << 30));
        }
    }
    uint64_t c = std::max<uint64_t>(cmin, (c0 + 127) & ~uint64_t(127));
    return uint32_t(std::min<uint64_t>(c, 1u << 30));
}

// ---- automatic mode: the measured choice of the scan strategy --------------------------------------------------------------
// Batches are told apart by size (small ones are launch-bound whatever scans them) and by whether their haystacks are short
// (under 512 B on average: the anchor kernel spends a wave step per work item, the walks a lane): what one kind of batch
// taught the handle says little about the other.
int choice_class(uint64_t blob_bytes, size_t nh) {
    const int size = blob_bytes < (16ull << 20) ? 0 : blob_bytes < (256ull << 20) ? 1 : blob_bytes < (4ull << 30) ? 2 : 3;
    return size | ((nh && blob_bytes / nh < 512) ? 4 : 0);
}

// What a GiB is EXPECTED to cost (ms) on a strategy this handle has not run yet: the rate the kernel reaches on text it
// suits.  Only decides whether an untried alternative is worth a probe (one whose best case is no better than what the current
// strategy measures is never tried); every decision between strategies that have run is taken on measured times.
float nominal_cost(const acmi_handle *h, int strat, uint64_t blob_bytes, size_t nh, double density) {
    const FlatAutomaton &fa = h->aut.flat();
    const double avg = nh ? double(blob_bytes) / double(nh) : 0.0;
    // `density` = records per byte of the batch just ended: what the text is like is known even for a strategy that never ran on
    // it -- the filter pays for every candidate (a match is one), the optimistic pass walks again wherever there is an event,
    // every record is 16 bytes written (17 GB per GiB of text at a record per byte).  Without these terms a^1..a^16 kept
    // probing kernels that had no chance (each probe a batch at a third of the rate).
    switch (strat) {
    case acmi_handle::kStratAnchor: {
        double c = fa.anchor_dense ? 0.45 : fa.anchor_mode == 2 ? 0.30 : 0.19;
        if (avg < 2048.0) c *= 2048.0 / std::max(avg, 128.0) * 0.5 + 0.5;          // (a wave step per work item: short haystacks idle its lanes)
        return float(c + density * 250.0);
    }
    case acmi_handle::kStratLds: return float((fa.compact ? 2.0 : 0.33) + density * 300.0);
    case acmi_handle::kStratTiles: {
        const bool allhot = uint64_t(fa.trap) * (fa.n_columns * fa.entry_bytes + 4u) <= 60u * 1024u && fa.trap + 1u == fa.n_states;
        return float((fa.compact ? 6.0 : allhot ? 1.0 : 1.7) + density * (density > 0.25 ? 20.0 : 4.5));   // (past an event per 4 bytes: the second walk)
    }
    default: return float((fa.compact ? 8.0 : 1.3) + density * 3.5);
    }
}

bool strategy_eligible(const acmi_handle *h, int strat, uint64_t blob_bytes, size_t nh, int find_all) {
    const FlatAutomaton &fa = h->aut.flat();
    uint32_t w = 0, p = 0;
    switch (strat) {
    case acmi_handle::kStratAnchor: return fa.anchor_q != 0;
    case acmi_handle::kStratTiles: return find_all && tiles_geometry(fa.max_pattern_length, fa.n_states, blob_bytes, nh, &w, &p);
    default: return true;
    }
}

// Where a handle starts, before anything is measured: what the pattern set suggests.
int seed_strategy(const acmi_handle *h, uint64_t blob_bytes, size_t nh, int find_all) {
    const FlatAutomaton &fa = h->aut.flat();
    // text over the patterns' own alphabet would flag more of an 8 KiB item than its run list holds (keyword lists, dense sets):
    // the optimistic LDS-tile pass ends up walking whole items anyway
    const bool flagged = fa.est_flag_rate * 8192.0 > double(h->run_cap) && !h->is_part;
    if (fa.anchor_q && !(blob_bytes >= kAutoMinBytes && nh && blob_bytes / nh < kSmallItemBytes && !flagged)) return acmi_handle::kStratAnchor;
    if (!flagged) return acmi_handle::kStratLds;
    // the wave drain writes output denser than a record per two bytes as whole rows; everything else in one pass
    if (!h->dense_output && strategy_eligible(h, acmi_handle::kStratTiles, blob_bytes, nh, find_all)) return acmi_handle::kStratTiles;
    return acmi_handle::kStratItems;
}

int choose_strategy(acmi_handle *h, acmi_handle::FlySlot &F, uint64_t blob_bytes, size_t nh, int find_all) {
    acmi_handle::ChoiceClass &C = h->choice[find_all ? 1 : 0][F.choice_class];
    F.probe = false;
    if (C.cur < 0 || !strategy_eligible(h, C.cur, blob_bytes, nh, find_all)) C.cur = seed_strategy(h, blob_bytes, nh, find_all);
    if (blob_bytes < kAutoMinBytes) return C.cur;   // (launch-bound either way: decides nothing, probes nothing)
    for (int s = 0; s < acmi_handle::kStratCount; ++s)
        if (C.wait[s]) --C.wait[s];
    for (int s = 0; s < acmi_handle::kStratCount; ++s) {
        if (!(C.pending >> s & 1u)) continue;
        C.pending &= ~(1u << s);
        if (s == C.cur || !strategy_eligible(h, s, blob_bytes, nh, find_all)) continue;
        F.probe = true;
        ++h->probes;
        return s;
    }
    return C.cur;
}

void learn_strategy(acmi_handle *h, const acmi_handle::FlySlot &F, double ms_per_gib, double density) {
    acmi_handle::ChoiceClass &C = h->choice[F.find_all ? 1 : 0][F.choice_class];
    const int s = F.strategy;
    C.cost[s] = C.cost[s] > 0.f ? 0.5f * C.cost[s] + 0.5f * float(ms_per_gib) : float(ms_per_gib);
    if (F.probe && s != C.cur) {
        // a probe wins when it is at least 1.25 x cheaper than what the strategy in use costs now; a lost one backs off
        if (C.cost[C.cur] > 0.f && C.cost[s] * 1.25f < C.cost[C.cur]) {
            C.cur = s;
            C.span[s] = 16;
            ++h->switches;
        } else {
            C.wait[s] = C.span[s];
            C.span[s] = std::min<uint32_t>(C.span[s] * 2, 1024);
        }
        return;
    }
    if (s != C.cur) return;
    // The strategy in use: every alternative whose kernel, on text that suits it, would be clearly cheaper than this is probed --
    // once its back-off allows: what it cost on EARLIER text (a filter that lost on dense matches) does not keep it out for good,
    // the text may have changed.  (Expected cost: the kernel's rate plus what any pipeline costs whatever it scans, ~ 60 us --
    // small batches are launch-bound on every strategy and probe nothing.)
    for (int a = 0; a < acmi_handle::kStratCount; ++a) {
        if (a == s || C.wait[a] || !strategy_eligible(h, a, F.blob_bytes, F.nh, F.find_all)) continue;
        const float ref = nominal_cost(h, a, F.blob_bytes, F.nh, density) + 0.06f * float(1ull << 30) / float(F.blob_bytes);
        if (C.cost[s] > 1.5f * ref) C.pending |= 1u << a;
    }
}

// The batch pipeline on device-resident input, first half: enqueue everything, no host wait.
int enqueue_batch(acmi_handle *h, acmi_handle::FlySlot &F, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off,
                  size_t nh, int find_all);

int run_batch_end(acmi_handle *h);

int run_batch_begin(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh,
                    int find_all) {
    if (h->begin_seq - h->end_seq >= 2) {
        h->last_error = "two batches are in flight already: end the older one first";
        return ACMI_E_ARG;
    }
    if (nh > 0xFFFFFFF0ull) { h->last_error = "too many haystacks"; return ACMI_E_TOO_LARGE; }
    if (h->part[0]) {   // split set: both parts scan the batch, one after the other on the stream; merged at the end
        acmi_handle::FlySlot &F = h->fly[(h->begin_seq + 1) & 1];
        if (int(&F - h->fly) == h->cur) h->have_result = false;
        for (acmi_handle *c : h->part) {
            c->stream = h->stream;
            c->profile = h->profile;
            c->uniform_len = h->uniform_len;
            c->merge_into = acmi_handle::MergeInto{};
            if (c == h->part[1] && find_all && nh && tiles_can_merge(c->dA, int(c->aut.flat().entry_bytes))) {
                // the short part's tile walk can place the long part's records among its own as it writes them: no merge pass
                // over the whole result (the short part's 70 M records per GiB were copied once more, 0.5 ms)
                acmi_handle *a = h->part[0];
                acmi_handle::FlySlot &FA = a->fly[a->begin_seq & 1];   // the batch part 0 has just begun
                if (!FA.trivial && FA.d_out.p) {
                    HIPCHK(h, F.d_out.ensure(std::max<size_t>(16, h->out_want)));
                    c->merge_into.prefix = FA.d_prefix.as<uint64_t>();
                    c->merge_into.records = FA.d_out.p;
                    c->merge_into.out = F.d_out.p;
                    c->merge_into.cap = F.d_out.cap / 16;
                }
            }
            const int rc = run_batch_begin(c, d_blob, blob_bytes, d_off, nh, find_all);
            c->merge_into = acmi_handle::MergeInto{};
            if (rc < 0) {
                h->last_error = c->last_error;
                if (c == h->part[1]) (void)run_batch_end(h->part[0]);   // (the first part's batch is not left in flight)
                return rc;
            }
        }
        F.nh = nh;
        F.find_all = find_all;
        F.blob_bytes = blob_bytes;
        F.stream = h->stream;
        F.trivial = false;
        F.seq = ++h->begin_seq;
        h->enqueue_seq = F.seq;
        return ACMI_OK;
    }
    // This batch takes the slot of the batch before the previous one, whose prefix array and records end here: a
    // result is valid from its batch's end until the SECOND begin after its own.
    acmi_handle::FlySlot &F = h->fly[(h->begin_seq + 1) & 1];
    // (the last completed result lives in slot `cur`: it stays readable through ONE further begin, as acmi.h says)
    if (int(&F - h->fly) == h->cur) h->have_result = false;
    const int rc = enqueue_batch(h, F, d_blob, blob_bytes, d_off, nh, find_all);
    // a pipeline that was not enqueued to its end has not zeroed the next batch's sync block: have both zeroed again
    if (rc < 0) { h->sync_tiles = 0; h->isync_clean[0] = h->isync_clean[1] = false; return rc; }
    F.seq = ++h->begin_seq;
    h->enqueue_seq = F.seq;
    return rc;
}

int enqueue_batch(acmi_handle *h, acmi_handle::FlySlot &F, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off,
                  size_t nh, int find_all) {
    hipStream_t st = h->stream;
    F.stream = st;
    F.anchor_variant = 0;
    HIPCHK(h, F.d_prefix.ensure((nh + 1) * 8));
    F.d_blob = d_blob;
    F.d_off = d_off;
    F.nh = nh;
    F.blob_bytes = blob_bytes;
    F.find_all = find_all;
    F.kernel_used = 0;
    F.chunk_used = 0;
    F.max_items = 0;
    const uint64_t allocs_before = g_dev_allocs;
    F.timed = false;
    HIPCHK(h, hipEventRecord(F.ev[0], st));   // (with ev[4]: what this batch's pipeline took -- the measured kernel choice, "total_ms")
    const int slot_index = int(&F - h->fly);
    uint32_t *const pinned_dev = h->pinned_dev + kPinWords * slot_index;

    if (nh == 0 || h->aut.accepted_count() == 0) {  // empty batch / empty automaton -> no records
        HIPCHK(h, hipMemsetAsync(F.d_prefix.p, 0, (nh + 1) * 8, st));
        HIPCHK(h, hipEventRecord(F.ev[4], st));
        F.trivial = true;
        return ACMI_OK;
    }
    F.trivial = false;

    const FlatAutomaton &fa = h->aut.flat();
    int kernel = (h->kernel == 2 && !fa.anchor_q) ? 1 : h->kernel;
    F.strategy = -1;
    F.choice_class = choice_class(blob_bytes, nh);
    if (h->kernel == 3) {
        F.strategy = choose_strategy(h, F, blob_bytes, nh, find_all);
        kernel = F.strategy == acmi_handle::kStratAnchor ? 2 : F.strategy == acmi_handle::kStratLds ? 1 : F.strategy == acmi_handle::kStratTiles ? 6 : 4;
    }
    // k_walk_tiles: one pass, records placed by a running sum instead of slab -> overflow -> second walk (findAll = true, patterns
    // of <= 129 B; else the whole-item walk)
    uint32_t tile_w = 0, tile_p = 0;
    if (kernel == 6 && find_all && tiles_geometry(fa.max_pattern_length, fa.n_states, blob_bytes, nh, &tile_w, &tile_p)) {
        if (h->tile_w) tile_w = uint32_t(h->tile_w);
        if (h->tile_p) tile_p = uint32_t(h->tile_p);
    } else if (kernel == 6) {
        kernel = 4;
    }
    F.kernel_used = kernel;
    F.tile_w = tile_w;
    F.tile_p = tile_p;
    const uint32_t chunk = kernel == 6 ? tile_w * tile_p : choose_chunk(h, blob_bytes, nh, kernel);
    const uint64_t max_items = blob_bytes / chunk + nh;
    // Event slots per item.  An item with more events than its slab holds is walked AGAIN by the drain, a wave per item, and
    // for text with tens to hundreds of matches per item -- a word list over ordinary prose -- that costs ten times what a
    // bigger slab does (tools/exp_kw_hits.py: 53 records per 8 KiB haystack 8.8 -> 1.1 ms per GiB, 200: 10.7 -> 3.1).  Unless the
    // caller fixes the size, a handle starts small (16 slots, 32 with the dense anchor form) and LEARNS: a batch that sent more
    // than 1 item in 64 to the drain quadruples the slabs of the batches after it (slab_auto, set at the batch's end) -- up to 256
    // slots per item, half the batch's bytes (8 bytes a slot, the expected item count) and 2 GiB in all.
    uint32_t K = std::max<uint32_t>(1, h->slab_events);
    const uint64_t est_items = std::max<uint64_t>(1, std::max<uint64_t>(blob_bytes / chunk, nh));
    if (!h->slab_set) {
        const uint32_t base = (kernel == 2 && fa.anchor_dense) ? 32u : 16u;
        uint64_t k = blob_bytes / 2 / (est_items * 8);
        k = std::min<uint64_t>(k, (2ull << 30) / ((max_items + 328) * 8));
        uint32_t kmax = base;
        while (kmax * 2 <= k && kmax < 256) kmax *= 2;
        K = std::min(std::max(base, h->slab_auto), kmax);
    }
    F.slab_used = K;
    F.est_items = est_items;
    F.chunk_used = chunk;

    DevBuf &d_item_base = h->d_item_base[slot_index], &d_item_hay = h->d_item_hay[slot_index], &d_item_rec = h->d_item_rec[slot_index],
           &d_zero = h->d_zero[slot_index], &d_isync = h->d_isync[slot_index];
    HIPCHK(h, d_item_base.ensure_keep((nh + 1) * 8, h->retired));
    HIPCHK(h, d_item_hay.ensure_keep(max_items * 4, h->retired));
    HIPCHK(h, d_item_rec.ensure_keep(max_items * 16, h->retired));
    // per-item counters [rec_count (items+1) x 8 | ev_count items x 4]: zeroed per item by k_items
    const size_t rec_cnt_off = 0, ev_off = rec_cnt_off + (max_items + 1) * 8;
    HIPCHK(h, d_zero.ensure_keep(ev_off + max_items * 4, h->retired));
    const size_t isync_bytes = (8 + size_t(scan_tiles(nh + 1) + 1) * 8 + 15) & ~size_t(15);
    if (isync_bytes > d_isync.cap) {
        HIPCHK(h, d_isync.ensure_keep(isync_bytes, h->retired));
        h->isync_clean[slot_index] = false;
    }
    // sync block of this batch: [counters 16 x u32 | tickets 2 x u32 | pad | status words of the two scans], one of
    // two copies that take turns; this batch's k_expand zeroes the other one.  (Re)allocation zeroes both.
    const uint32_t tiles = std::max(scan_tiles(nh + 1), scan_tiles(max_items + 1)) + 1u;
    if (tiles > h->sync_tiles || !h->d_sync[0].p) {
        const uint32_t want = tiles + tiles / 4 + 16;
        for (DevBuf &b : h->d_sync) {
            HIPCHK(h, b.ensure_keep(128 + size_t(want) * 16, h->retired));
            HIPCHK(h, hipMemsetAsync(b.p, 0, b.cap, st));
        }
        h->sync_tiles = want;
    }
    const uint32_t sync_words = uint32_t(32 + size_t(h->sync_tiles) * 4);
    uint32_t *sync_cur = h->d_sync[h->sync_turn & 1].as<uint32_t>();
    uint32_t *sync_next = h->d_sync[(h->sync_turn + 1) & 1].as<uint32_t>();
    ++h->sync_turn;
    // (status words of k_expand's scan: tiles of 1,024 items -- the block holds 2 x sync_tiles words, sized for tiles of 2,048;
    // k_items has sync words of its own, d_isync)
    uint64_t *status1 = reinterpret_cast<uint64_t *>(sync_cur + 32);
    HIPCHK(h, h->d_rec_base.ensure_keep((max_items + 2) * 8, h->retired));
    // Run lists and event slabs are index-major (entry r of every item, then entry r + 1 ...): the distance between an item's
    // consecutive entries must not be a power of two.  2^17 haystacks of 8 KiB made it 4 MiB, an item's six candidate windows
    // fell into one cache set of one channel, and k_walk_runs took 2.4 ms instead of 0.075 (tools/exp_hits.py: five plants per
    // haystack fine, six 35 x slower; any run_cap but 8 fine too, which moved the buffers).  Hence 328 entries of padding.
    const uint64_t run_stride = max_items + 328;
    HIPCHK(h, h->d_slab.ensure_keep(run_stride * K * 8, h->retired));
    HIPCHK(h, h->d_ovf.ensure_keep(max_items * 4, h->retired));
    // dense anchor form: hit windows per item by the dozen -- 4-byte entries, 256 of them unless the caller says otherwise
    const bool dense = kernel == 2 && fa.anchor_dense != 0;
    // (run lists learn like the slabs: 16 entries, quadrupled -- up to 64 and an eighth of the batch's bytes -- after a batch
    // whose filter lost because items overflowed them, before the automatic choice gives the filter up: run_batch_end)
    uint32_t run_max = 16;
    while (run_max < 64u && uint64_t(run_max) * 2 * 16 * est_items * 8 <= blob_bytes) run_max *= 2;
    // (dense form: an item of L bytes has at most L hit windows, and 256 entries for every 256-byte haystack would be 8 x the batch:
    // what an average item can hold, at least 16, and never more than the batch's own bytes in all)
    uint32_t dense_q = 256u;
    if (dense) {
        const uint64_t avg = std::max<uint64_t>(16, blob_bytes / est_items);
        while (dense_q > 16u && (dense_q > avg || run_stride * dense_q * 8 > std::max<uint64_t>(blob_bytes, 64ull << 20))) dense_q /= 2;
    }
    const uint32_t Q = h->run_cap_set ? std::max<uint32_t>(1, h->run_cap) : dense ? dense_q : std::min(std::max(16u, h->run_auto), run_max);
    F.run_used = Q;
    F.run_max = (h->run_cap_set || dense) ? Q : run_max;
    HIPCHK(h, h->d_runs.ensure_keep(run_stride * Q * (dense ? 8 : 16), h->retired));   // (dense: window starts, then k_dense_decide's results)
    HIPCHK(h, h->d_run_count.ensure_keep(max_items * 4, h->retired));
    uint8_t *zero = d_zero.as<uint8_t>();
    uint32_t *d_counters = sync_cur;
    uint64_t *d_rec_cnt = reinterpret_cast<uint64_t *>(zero + rec_cnt_off);
    uint32_t *d_ev = reinterpret_cast<uint32_t *>(zero + ev_off);

    BatchArgs B{};
    B.blob = d_blob;
    B.offsets = d_off;
    B.item_base = d_item_base.as<uint64_t>();
    B.n_haystacks = uint32_t(nh);
    B.chunk_bytes = chunk;
    B.warm_bytes = fa.max_pattern_length ? fa.max_pattern_length - 1 : 0;
    B.align_warm = kernel == 2 ? 1u : (kernel == 1 && h->dA.other_byte < 0x100u ? 2u : 0u);
    // anchor kernel: the variant that skips the 1-KiB rows a task lacks costs the perfectly regular batch -- haystacks of
    // exactly k x 8 KiB on an aligned blob, every task full: the bench's shape -- 4 % (0.704 vs 0.677 ms) and gains up to 2 x
    // on everything else (short items, last chunks, the 513th quad of an 8-KiB haystack that starts unaligned)
    const bool regular = blob_bytes % nh == 0 && (blob_bytes / nh) % 8192 == 0 && (reinterpret_cast<uintptr_t>(d_blob) & 15u) == 0;
    B.short_items = kernel != 2 ? 0u : h->short_rows >= 0 ? uint32_t(h->short_rows) : (regular ? 0u : 1u);
    B.tiny_items = (kernel == 2 && h->tiny_rows != 0 && blob_bytes / nh <= 2048) ? 1u : 0u;   // (waves whose items do not all fit a row take the usual path: the average may be skewed by a few long haystacks)
    B.slab_events = K;
    B.first_only = find_all ? 0u : 1u;
    B.item_hay = d_item_hay.as<uint32_t>();
    B.item_rec = d_item_rec.as<uint4>();
    B.ev_count = d_ev;
    B.rec_count = d_rec_cnt;
    B.slab = h->d_slab.as<uint2>();
    B.overflow_list = h->d_ovf.as<uint32_t>();
    B.counters = d_counters;
    B.runs = h->d_runs.as<uint4>();
    B.run_count = h->d_run_count.as<uint32_t>();
    B.run_cap = Q;
    B.run_dense = dense ? 1u : 0u;
    B.run_stride = run_stride;
    B.poison_quads = std::max<uint32_t>(1, (fa.max_pattern_length + 15) / 16);
    // persistent workgroups pay off once a CU would otherwise run several workgroups back to back
    // (measured: 1 KiB items -18 %, 2 x 8 KiB per lane +2.5 %)
    {   // anchor kernel: items per wave batch -- 64 when the batch can give every wave that many
        // (at least one item per haystack: a GiB of 256-byte haystacks is 4 M items, not the 131,072 its bytes suggest -- with
        // 32 items per wave batch the batch's fixed costs were most of the tiny-item kernel's time)
        const uint64_t est = std::max<uint64_t>(1, std::max<uint64_t>(blob_bytes / chunk, nh));
        const uint64_t waves = 256ull * 16;
        B.wave_items = uint32_t(std::min<uint64_t>(64, std::max<uint64_t>(1, est / waves)));
        if (h->wave_items > 0) B.wave_items = uint32_t(h->wave_items);
    }
    B.persistent = h->persistent >= 0 ? uint32_t(h->persistent != 0)
                                      : uint32_t(blob_bytes / chunk > 3ull * 256 * 1024 * uint64_t(h->ilp));
    B.wave_grab = 1u;
    if (h->persistent < 0 && (kernel == 1 || kernel == 2)) {
        // A static split gives a workgroup 16 x wave_items items (anchor kernel) or 1,024 x ILP (tile walk), and a CU runs
        // one workgroup at a time: fine when the workgroups are a multiple of the 256 CUs (the headline batch: 512, and
        // 2.5 % faster than pulling batches from a counter), but 528 of them -- 1 MiB haystacks in 7,936-byte chunks -- are
        // three rounds of which the last is almost empty (0.87 instead of 0.67 ms per 4 GiB), and 513 -- 524,352
        // haystacks through the tile walk -- end with ONE workgroup that lasts as long as the others (1.50 vs 1.25 ms).
        // The item count is only known here when all haystacks have one length: any other batch, and any uneven count,
        // gets persistent workgroups.
        const uint64_t per_group = kernel == 2 ? 16ull * B.wave_items : 1024ull * uint64_t(std::max(1, h->ilp));
        const uint64_t hay_len = blob_bytes / nh;
        const uint64_t items = nh * ((hay_len + chunk - 1) / chunk);
        const uint64_t groups = (items + per_group - 1) / per_group;
        if (blob_bytes % nh != 0 || (groups > 256 && groups % 256 != 0)) B.persistent = 1u;
    }
    // "reserve_cus": a scan that holds every CU's whole LDS for most of a step leaves a collective's kernels (RCCL's need LDS and
    // CUs of their own) only the gaps between scans: N CUs are kept free of scan workgroups -- which takes persistent workgroups
    B.reserve_cus = uint32_t(h->reserve_cus);
    if (h->reserve_cus > 0 && (kernel == 1 || kernel == 2)) B.persistent = 1u;
    if (B.persistent) {
        // Every batch a wave pulls is one device atomic on ONE word, and the chip retires those at ~ 12 ns each whoever issues
        // them: 4 GiB of 256-byte haystacks are 262,144 wave batches = 3.1 ms of atomics under a kernel that needs 0.5 (the same
        // went for the statistic every tiny batch used to add to: both found as "a wave batch costs 50 us whatever it holds").
        // A wave therefore takes up to 16 batches per atomic, so that a launch has ~ 16,384 of them at most.
        const uint64_t est = std::max<uint64_t>(1, std::max<uint64_t>(blob_bytes / chunk, nh));
        const uint64_t per_batch = kernel == 2 ? B.wave_items : 64ull * uint64_t(std::max(1, h->ilp));
        const uint64_t batches = (est + per_batch - 1) / per_batch;
        B.wave_grab = uint32_t(std::min<uint64_t>(16, std::max<uint64_t>(1, (batches + 16383) / 16384)));
        if (max_items > (1ull << 31)) B.wave_grab = 1u;   // (the 32-bit item counter overshoots by what the waves take past the end)
    }

    // item counts, their scan, the item table, zeroed per-item counters: one launch
    {
        // "items_ahead": on the side stream -- it starts at once, next to the batch before (whose slot this is not), and the
        // pipeline waits for it.  The caller vouches that offsets and blob are complete when the batch is begun.
        hipStream_t ist = (h->items_ahead && h->side_stream) ? h->side_stream : st;
        // ... but not next to that batch's SCAN, which wants every compute unit to itself (k_items' workgroups, placed first,
        // kept the scan's off their units: scan + 24 us for 14 us saved): behind it, next to its candidate walk and expansion
        if (ist != st && h->begin_seq != h->end_seq) HIPCHK(h, hipStreamWaitEvent(ist, h->ev_scanned[slot_index ^ 1], 0));
        // (the sync words of k_items are zeroed by the k_expand of the batch before -- no fill launch per batch -- unless that
        // batch had no pipeline, failed, or this k_items runs ahead of it)
        if (!h->isync_clean[slot_index] || ist != st) HIPCHK(h, hipMemsetAsync(d_isync.p, 0, d_isync.cap & ~size_t(15), ist));
        h->isync_clean[slot_index] = false;
        if (h->uniform_len && blob_bytes == h->uniform_len * nh) launch_items_uniform(B, h->uniform_len, d_item_base.as<uint64_t>(), ist);
        else launch_items(B, blob_bytes, max_items, d_item_base.as<uint64_t>(), d_isync.as<uint32_t>(), ist);
        if (ist != st) {
            HIPCHK(h, hipEventRecord(h->ev_items[slot_index], ist));
            HIPCHK(h, hipStreamWaitEvent(st, h->ev_items[slot_index], 0));
        }
    }

    // Records are written optimistically into a buffer of the capacity the batches before needed (at least): ONE host
    // synchronisation per batch; a batch that produced more is run again into a grown buffer (run_batch_end).
    HIPCHK(h, F.d_out.ensure(std::max<size_t>(16, h->max_result_bytes ? std::min<size_t>(h->out_want, h->max_result_bytes) : h->out_want)));
    // ("max_result_bytes": no record past the cap is written even where the allocation has slack -- what the host hands out is
    // whole haystacks below it)
    const uint64_t cap = h->max_result_bytes ? std::min<uint64_t>(F.d_out.cap / 16, h->max_result_bytes / 16) : F.d_out.cap / 16;
    F.out_cap = cap;
    if (kernel == 6) {
        HIPCHK(h, h->d_tile_queue.ensure_keep(tiles_queue_entries(tile_w) * 8, h->retired));
        HIPCHK(h, h->d_tile_sync.ensure_keep(8 + tiles_status_words(max_items, tile_p) * 8, h->retired));
        HIPCHK(h, hipMemsetAsync(h->d_tile_sync.p, 0, 8 + tiles_status_words(max_items, tile_p) * 8, st));
    }
    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[1], st));
    F.merged = false;
    F.reran = false;
    if (kernel == 6) {
        F.merged = h->merge_into.out != nullptr;
        HIPCHK(h, launch_walk_tiles(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), tile_w, tile_p, h->d_tile_queue.p,
                                    h->d_tile_sync.as<uint32_t>(), h->d_tile_sync.as<uint64_t>() + 1, F.merged ? h->merge_into.out : F.d_out.p,
                                    F.merged ? h->merge_into.cap : cap, st, F.merged ? h->merge_into.prefix : nullptr, F.merged ? h->merge_into.records : nullptr));
    } else if (kernel == 4) {
        // exact walk only (k_walk_items): for text on which k_scan_lds' optimistic
        // pass flags everything anyway (events in every wave-quad)
        HIPCHK(h, launch_walk_items(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), h->walk_group, st));
    } else {
        HIPCHK(h, launch_scan(h->dA, B, max_items, int(fa.entry_bytes), kernel, int(fa.class_mode), h->ilp, h->group_quads, st, &F.anchor_variant));
    }
    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[2], st));
    if (h->items_ahead) HIPCHK(h, hipEventRecord(h->ev_scanned[slot_index], st));
    // the anchor kernel only flags byte runs; their exact walk is a kernel of its own (one thread per item)
    if (kernel == 2) HIPCHK(h, launch_walk_runs(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), st));
    if (h->profile > 1) HIPCHK(h, hipEventRecord(F.ev[3], st));

    if (!find_all) {
        HIPCHK(h, h->d_first.ensure_keep(nh * 4, h->retired));
        HIPCHK(h, hipMemsetAsync(h->d_first.p, 0xFF, nh * 4, st));
        launch_first_only(B, h->d_first.as<uint32_t>(), max_items, st);
    }

    // Expansion is enqueued optimistically with the output capacity of the previous call, so the
    // whole pipeline needs ONE host synchronisation (the read-back of the summary at the end).  Only
    // when the batch produced more records than fit is the buffer grown and the expansion re-run.
    // (also: per-haystack prefix, the summary for the host next to the overflow counter -- counters[2..3] = total
    // records, [4..5] = items -- and the zeroing of the next batch's sync block)
    launch_expand(h->dA, B, max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, F.d_prefix.as<uint64_t>(),
                  pinned_dev, sync_next, sync_words, sync_cur + 17, status1, d_isync.as<uint32_t>() + 1, true, st,
                  h->items_ahead ? nullptr : h->d_isync[slot_index ^ 1].as<uint32_t>(),
                  h->items_ahead ? 0u : uint32_t((h->d_isync[slot_index ^ 1].cap & ~size_t(15)) / 4));
    // ("items_ahead": the next batch's k_items may already be running on the side stream, behind this batch's SCAN only -- its
    // sync words are zeroed by that stream's own memset, never from here)
    h->isync_clean[slot_index ^ 1] = !h->items_ahead && h->d_isync[slot_index ^ 1].p != nullptr;
    F.refill_launched = find_all && h->refill_armed && kernel != 6;   // (k_walk_tiles leaves no item to drain)
    if (F.refill_launched) launch_refill(h->dA, B, max_items, int(fa.entry_bytes), h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, refill_mode(h, kernel), st, int(fa.class_mode));
    F.sync_next = sync_next;
    F.sync_words = sync_words;
    HIPCHK(h, hipEventRecord(F.ev[4], st));  // (always: the end waits for THIS batch, not for the one queued behind it)
    HIPCHK(h, hipGetLastError());
    F.timed = g_dev_allocs == allocs_before;
    F.args = B;
    F.max_items = max_items;
    return ACMI_OK;
}

// Second half, for the OLDEST batch in flight: the one host synchronisation, the capacity check and (rarely) the re-run.
// The end of a split set's batch: both parts' ends, then the merge of their match arrays into this handle's result buffers.
int run_split_end(acmi_handle *h) {
    acmi_handle::FlySlot &F = h->fly[(h->end_seq + 1) & 1];
    ++h->end_seq;
    acmi_handle *a = h->part[0], *b = h->part[1];
    int rc = run_batch_end(a);
    const int rc2 = run_batch_end(b);
    if (rc < 0 || rc2 < 0) { h->last_error = rc < 0 ? a->last_error : b->last_error; return rc < 0 ? rc : rc2; }
    hipStream_t st = F.stream;
    const uint32_t nh = uint32_t(F.nh), first_only = F.find_all ? 0u : 1u;
    const uint64_t ta = a->last_total, tb = b->last_total;
    HIPCHK(h, F.d_prefix.ensure((size_t(nh) + 1) * 8));
    acmi_handle::FlySlot &FA = a->fly[a->cur], &FB = b->fly[b->cur];
    bool merged = FB.merged;
    h->out_want = std::max<size_t>(h->out_want, size_t(ta + tb) * 16);
    if (merged && (FA.reran || (ta + tb) * 16 > F.d_out.cap)) {
        // Part 1 merged part 0's records while writing its own -- but part 0's were not final when it ran (their buffer had to
        // grow, items had to be drained at the end), or the merged array outgrew this handle's buffer: part 1's batch once more,
        // now that everything is known.  (A later begin may have taken part 1's intermediates: the whole pipeline, as
        // run_batch_end does for a batch whose records outgrew its buffer.)
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, F.d_out.ensure((ta + tb) * 16));
        b->merge_into.prefix = FA.d_prefix.as<uint64_t>();
        b->merge_into.records = FA.d_out.p;
        b->merge_into.out = F.d_out.p;
        b->merge_into.cap = F.d_out.cap / 16;
        const int rr = enqueue_batch(b, FB, FB.d_blob, FB.blob_bytes, FB.d_off, FB.nh, FB.find_all);
        b->merge_into = acmi_handle::MergeInto{};
        if (rr < 0) { b->sync_tiles = 0; h->last_error = b->last_error; return rr; }
        b->enqueue_seq = FB.seq;
        HIPCHK(h, hipStreamSynchronize(st));
        merged = FB.merged;
    }
    HIPCHK(h, F.d_out.ensure(std::max<uint64_t>(16, (ta + tb) * 16)));
    const uint64_t *pa = a->d_prefix().as<uint64_t>(), *pb = b->d_prefix().as<uint64_t>();
    uint64_t total = ta + tb;
    if (merged) {
        ++h->merged_batches;
        launch_merge_prefix_sum(pa, pb, nh, F.d_prefix.as<uint64_t>(), st);   // (the records are in place already)
    } else if (!first_only) {
        launch_merge_prefix_sum(pa, pb, nh, F.d_prefix.as<uint64_t>(), st);
    } else {
        // findAll = false: a haystack keeps the earlier of the parts' first events (both at equal positions): counts, then their scan
        HIPCHK(h, h->d_merge_counts.ensure(std::max<size_t>(8, size_t(nh) * 8)));
        const size_t sync_bytes = 8 + size_t(scan_tiles(uint64_t(nh) + 1) + 1) * 8;
        HIPCHK(h, h->d_merge_sync.ensure(sync_bytes));
        HIPCHK(h, hipMemsetAsync(h->d_merge_sync.p, 0, sync_bytes, st));
        launch_merge_counts(pa, a->d_out().p, pb, b->d_out().p, nh, 1u, h->d_merge_counts.as<uint64_t>(), st);
        const int slot_index = int(&F - h->fly);
        launch_scan_u64(h->d_merge_counts.as<uint64_t>(), nh, F.d_prefix.as<uint64_t>(), h->d_merge_sync.as<uint32_t>(),
                        reinterpret_cast<uint64_t *>(h->pinned_dev + kPinWords * slot_index), st);
    }
    if (!merged) launch_merge_parts(pa, a->d_out().p, ta, pb, b->d_out().p, tb, nh, first_only, F.d_prefix.as<uint64_t>(), F.d_out.p, st);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(st));
    if (first_only) total = h->pinned[(kPinWords / 2) * int(&F - h->fly)];
    h->cur = int(&F - h->fly);
    h->last_nh = nh;
    h->last_total = total;
    h->have_result = true;
    h->kernel_used = a->kernel_used;
    h->anchor_variant = a->anchor_variant;
    h->tiny_waves = a->tiny_waves;
    h->chunk_used = a->chunk_used;
    h->items = a->items + b->items;
    h->overflow_items = a->overflow_items + b->overflow_items;
    h->walked_bytes = a->walked_bytes + b->walked_bytes;
    h->fallback_bytes = a->fallback_bytes + b->fallback_bytes;
    h->scan_ms = a->scan_ms + b->scan_ms;
    h->walk_ms = a->walk_ms + b->walk_ms;
    h->total_ms = a->total_ms + b->total_ms;
    return ACMI_OK;
}

int run_batch_end(acmi_handle *h) {
    if (h->begin_seq == h->end_seq) { h->last_error = "no batch in flight"; return ACMI_E_ARG; }
    if (h->part[0]) return run_split_end(h);
    acmi_handle::FlySlot &F = h->fly[(h->end_seq + 1) & 1];
    ++h->end_seq;
    const int slot_index = int(&F - h->fly);
    const uint64_t *const pinned = h->pinned + (kPinWords / 2) * slot_index;
    const uint32_t *const pinned32 = reinterpret_cast<const uint32_t *>(pinned);
    hipStream_t st = F.stream;
    {
        // the wait is for THIS batch's last kernel (its own event, on the stream it was enqueued on), not for whatever
        // else the stream holds -- a second batch queued behind it, the caller's own work
        const hipError_t e = hipEventSynchronize(F.ev[4]);
        if (e != hipSuccess) { h->sync_tiles = 0; h->end_seq = h->begin_seq; return fail_hip(h, e, "hipEventSynchronize"); }
    }
    if (h->begin_seq == h->end_seq && !h->retired.empty()) {  // nothing in flight: blocks outgrown meanwhile can go
        for (void *q : h->retired) (void)hipFree(q);
        h->retired.clear();
    }
    h->cur = slot_index;
    h->last_nh = uint32_t(F.nh);
    h->last_total = 0;
    h->scan_ms = h->walk_ms = h->total_ms = 0;
    h->items = 0;
    h->overflow_items = 0;
    h->kernel_used = F.kernel_used;
    h->chunk_used = F.chunk_used;
    h->anchor_variant = F.anchor_variant;
    h->tiny_waves = 0;
    if (F.trivial) {
        h->have_result = true;
        return ACMI_OK;
    }
    const FlatAutomaton &fa = h->aut.flat();
    if (pinned32[6] != 0u) {
        h->last_error = "offsets must be non-decreasing, inside the blob, and every haystack shorter than 4 GiB";
        return ACMI_E_ARG;
    }
    const uint32_t n_ovf = F.find_all ? pinned32[0] : 0u;
    uint64_t total = pinned[1];
    h->items = pinned[2];
    h->overflow_items = n_ovf;
    h->dense_output = total * 2 >= F.blob_bytes && F.blob_bytes != 0;
    if (!h->slab_set && uint64_t(n_ovf) * 64 >= F.est_items && F.slab_used < 256u) h->slab_auto = std::max(h->slab_auto, std::min(256u, F.slab_used * 4u));
    h->walked_bytes = 0;  // counters[7..15]: k_scan_lds adds to [7], k_walk_runs spreads its workgroups over [8..15]
    for (int i = 7; i < 16; ++i) h->walked_bytes += uint64_t(pinned32[i]) * 16;
    h->fallback_bytes = F.kernel_used == 2 ? uint64_t(pinned32[7]) * 16 : 0;
    if (F.kernel_used == 2) h->walked_bytes -= h->fallback_bytes;  // (counters[7] is k_scan_lds' own share otherwise)
    h->tiny_waves = pinned32[18];
    // run lists learn like the slabs: longer ones for the batches after one whose filter walked more than a 64th of the bytes
    // whole (items whose candidate windows overflowed the list)
    if (h->kernel == 3 && F.blob_bytes >= kAutoMinBytes && F.kernel_used == 2 && h->fallback_bytes * 64 > F.blob_bytes && F.run_used < F.run_max)
        h->run_auto = F.run_used * 4;
    float pipeline_ms = 0;
    HIPCHK(h, hipEventElapsedTime(&pipeline_ms, F.ev[0], F.ev[4]));
    h->total_ms = pipeline_ms;
    if (h->profile) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, F.ev[1], F.ev[2]));
        h->scan_ms = ms;
        if (h->profile > 1) {
            HIPCHK(h, hipEventElapsedTime(&ms, F.ev[2], F.ev[3]));
            h->walk_ms = F.kernel_used == 2 ? ms : 0.0;
        }
    }
    // "max_result_bytes": a batch whose records exceed the cap hands out the haystacks that fit below it
    const uint64_t bound = (h->max_result_bytes && !h->is_part) ? h->max_result_bytes / 16 : ~0ull;
    const bool partial = total > bound;
    const uint64_t need = partial ? bound : total;   // records the buffer must hold
    if (!F.merged) h->out_want = std::max<size_t>(h->out_want, size_t(need) * 16);
    // (a merged batch wrote into the parent's buffer: whether THAT was big enough is the parent's business, run_split_end)
    const bool too_small = !F.merged && need > F.out_cap;   // did not fit the optimistic capacity: grow, expand again
    const bool undrained = n_ovf > 0 && !F.refill_launched;      // items overflowed their slabs and the drain was not enqueued
    h->refill_armed = n_ovf > 0;
    // (a pipeline that has to be completed or run again below says nothing about what the strategy costs)
    if (F.strategy >= 0 && F.blob_bytes >= kAutoMinBytes) {
        if (F.timed && !too_small && !undrained) learn_strategy(h, F, double(pipeline_ms) * double(1ull << 30) / double(F.blob_bytes), double(total) / double(F.blob_bytes));
        else if (F.probe) h->choice[F.find_all ? 1 : 0][F.choice_class].pending |= 1u << F.strategy;   // (the probe is taken again)
    }
    F.reran = too_small || undrained;
    if (too_small || undrained) {
        if (h->enqueue_seq == F.seq) {
            if (too_small) HIPCHK(h, F.d_out.ensure(need * 16));
            const uint64_t cap = partial ? bound : F.d_out.cap / 16;
            F.out_cap = cap;
            if (F.kernel_used == 6) {   // the walk again, into the grown buffer (prefix array and summary stand)
                const size_t sync_bytes = 8 + tiles_status_words(F.max_items, F.tile_p) * 8;
                HIPCHK(h, hipMemsetAsync(h->d_tile_sync.p, 0, sync_bytes, st));
                HIPCHK(h, launch_walk_tiles(h->dA, F.args, F.max_items, int(fa.entry_bytes), int(fa.class_mode), F.tile_w, F.tile_p, h->d_tile_queue.p,
                                            h->d_tile_sync.as<uint32_t>(), h->d_tile_sync.as<uint64_t>() + 1, F.d_out.p, cap, st));
            } else if (too_small)
                launch_expand(h->dA, F.args, F.max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, F.d_prefix.as<uint64_t>(),
                              h->pinned_dev + kPinWords * slot_index, F.sync_next, F.sync_words, nullptr, nullptr,
                              h->d_isync[slot_index].as<uint32_t>() + 1, false, st);
            if (F.find_all && F.kernel_used != 6)
                launch_refill(h->dA, F.args, F.max_items, int(fa.entry_bytes), h->d_rec_base.as<uint64_t>(),
                              F.d_out.p, cap, refill_mode(h, F.kernel_used), st, int(fa.class_mode));
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipStreamSynchronize(st));
        } else {
            // a later begin has reused the item table, slabs and record bases: this batch runs again as a whole, into a
            // buffer of the size it is now known to need (the later batch's own result and summary are its slot's)
            {
                const hipError_t e = hipStreamSynchronize(st);
                if (e != hipSuccess) { h->sync_tiles = 0; return fail_hip(h, e, "hipStreamSynchronize"); }
            }
            if (too_small) { h->out_want = std::max<size_t>(h->out_want, size_t(need) * 16); HIPCHK(h, F.d_out.ensure(need * 16)); }
            h->refill_armed = true;
            const int rc = enqueue_batch(h, F, F.d_blob, F.blob_bytes, F.d_off, F.nh, F.find_all);
            if (rc < 0) { h->sync_tiles = 0; return rc; }
            h->enqueue_seq = F.seq;
            HIPCHK(h, hipStreamSynchronize(st));
            if (pinned[1] != total) { h->last_error = "record count changed between two runs of one batch"; return ACMI_E_DEVICE; }
        }
    }
    h->last_total = total;
    h->have_result = true;
    h->resume_haystack = F.nh;
    if (partial) {
        // the haystacks whose records lie wholly below the cap: the largest k with prefix[k] <= bound (the prefix array is on the
        // device: a binary search by single-word copies -- this path is an exception, not a rate)
        const uint64_t *d_prefix = F.d_prefix.as<uint64_t>();
        uint64_t lo = 0, hi = F.nh;   // prefix[lo] <= bound < prefix[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            uint64_t v = 0;
            HIPCHK(h, hipMemcpy(&v, d_prefix + mid, 8, hipMemcpyDeviceToHost));
            if (v <= bound) lo = mid; else hi = mid;
        }
        if (lo == 0) {
            h->have_result = false;
            h->last_error = "the records of one haystack alone exceed max_result_bytes";
            return ACMI_E_TOO_LARGE;
        }
        uint64_t v = 0;
        HIPCHK(h, hipMemcpy(&v, d_prefix + lo, 8, hipMemcpyDeviceToHost));
        h->last_nh = uint32_t(lo);
        h->last_total = v;
        h->resume_haystack = lo;
        h->last_error = "the batch's records exceed max_result_bytes: the result holds the first haystacks only (resume_haystack)";
        return ACMI_E_PARTIAL;
    }
    return ACMI_OK;
}

int run_batch(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh,
              int find_all) {
    if (h->begin_seq != h->end_seq) {  // a synchronous call drops batches that were begun and never ended
        for (auto &slot : h->fly)
            if (slot.stream) (void)hipStreamSynchronize(slot.stream);
        h->end_seq = h->begin_seq;
        h->have_result = false;
        for (acmi_handle *c : h->part)
            if (c) { c->end_seq = c->begin_seq; c->have_result = false; }
    }
    int rc = run_batch_begin(h, d_blob, blob_bytes, d_off, nh, find_all);
    if (rc < 0) return rc;
    return run_batch_end(h);
}

// Small host batches: one launch that reads and writes pinned host memory (k_match_small).  Returns 1 when
// the batch does not qualify or its records do not fit, and the caller takes the general path.
constexpr uint32_t kSmallCap = 8192;  // records
constexpr size_t kSmallSummaryOff = 0, kSmallPrefixOff = 128, kSmallRecordsOff = kSmallPrefixOff + (kSmallPathMaxHay + 8) * 8,
                 kSmallOffsetsOff = kSmallRecordsOff + size_t(kSmallCap) * 16, kSmallBytesOff = kSmallOffsetsOff + (kSmallPathMaxHay + 8) * 8,
                 kSmallPinBytes = kSmallBytesOff + kSmallPathMaxBytes + 64;

int run_small(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t nh, int find_all) {
    if (!h->small_path || nh == 0 || nh > kSmallPathMaxHay || h->part[0]) return 1;   // (a split set has two automata: the pipeline)
    const uint64_t span = offsets[nh] - offsets[0];
    if (span > kSmallPathMaxBytes) return 1;
    if (!h->pin_small) {
        if (hipHostMalloc(reinterpret_cast<void **>(&h->pin_small), kSmallPinBytes, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            h->pin_small = nullptr;
            return 1;
        }
    }
    uint8_t *pin = h->pin_small;
    std::memcpy(pin + kSmallOffsetsOff, offsets, (nh + 1) * 8);
    if (span) std::memcpy(pin + kSmallBytesOff, blob + offsets[0], span);
    void *dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer(&dev, pin, 0));
    uint8_t *d = static_cast<uint8_t *>(dev);
    const FlatAutomaton &fa = h->aut.flat();
    hipStream_t st = h->stream;
    if (h->profile) HIPCHK(h, hipEventRecord(h->ev[0], st));
    HIPCHK(h, launch_match_small(h->dA, reinterpret_cast<const uint64_t *>(d + kSmallOffsetsOff), d + kSmallBytesOff, uint32_t(nh),
                                 find_all, fa.max_pattern_length, int(fa.entry_bytes), int(fa.class_mode),
                                 reinterpret_cast<uint64_t *>(d + kSmallSummaryOff), reinterpret_cast<uint64_t *>(d + kSmallPrefixOff),
                                 d + kSmallRecordsOff, kSmallCap, uint32_t(span), ++h->small_seq, st));
    if (h->profile) HIPCHK(h, hipEventRecord(h->ev[4], st));
    const uint64_t *summary = reinterpret_cast<const uint64_t *>(pin + kSmallSummaryOff);
    {
        // The kernel's last store is the call's sequence number into the pinned summary: poll it (the whole call is ~ 15 us of
        // device time; hipStreamSynchronize sleeps and is woken by an interrupt, ~ 8 us more) -- for a few hundred microseconds,
        // then wait the ordinary way.  (With events on the stream -- "profile" -- the stream is synchronised anyway.)
        const volatile uint64_t *flag = reinterpret_cast<const volatile uint64_t *>(summary + 3);
        const auto t_poll = std::chrono::steady_clock::now();
        bool done = false;
        while (!h->profile) {
            if (*flag == h->small_seq) { done = true; break; }
            if (std::chrono::steady_clock::now() - t_poll > std::chrono::microseconds(300)) break;
        }
        if (done) std::atomic_thread_fence(std::memory_order_acquire);
        else HIPCHK(h, hipStreamSynchronize(st));
    }
    if (std::getenv("ACMI_SMALL_TRACE"))
        std::fprintf(stderr, "[acmi small] staged %.1f  items %.1f  walked %.1f  scanned %.1f  written %.1f us\n", summary[4] / 100.0, summary[5] / 100.0,
                     summary[6] / 100.0, summary[7] / 100.0, summary[8] / 100.0);
    if (summary[2] != 0) return 1;  // more records than the pinned buffer holds: general path
    const uint64_t total = summary[0];
    h->items = summary[1];
    h->h_prefix.assign(reinterpret_cast<const uint64_t *>(pin + kSmallPrefixOff), reinterpret_cast<const uint64_t *>(pin + kSmallPrefixOff) + nh + 1);
    h->h_records.assign(reinterpret_cast<const acmi_record *>(pin + kSmallRecordsOff), reinterpret_cast<const acmi_record *>(pin + kSmallRecordsOff) + total);
    h->last_total = total;
    h->last_nh = uint32_t(nh);
    h->have_result = false;  // the result lives in host memory only
    h->kernel_used = 5;  // k_match_small
    h->overflow_items = 0;
    h->walked_bytes = 0;
    if (h->profile) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[4]));
        h->scan_ms = h->total_ms = ms;
    }
    return ACMI_OK;
}

int check_ready(acmi_handle *h) {
    if (!live(h)) return ACMI_E_INVALID;
    if (h->init_ok != 1) return ACMI_E_NOT_INIT;
    int f = ensure_finalized(h);  // lazy finalize (php_ahocorasick.c:707)
    if (f < 0) return f;
    return ensure_device(h);
}

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

}  // namespace

extern "C" {

int acmi_create(int device, acmi_handle **out) {
    return guarded(nullptr, [&]() -> int {
        if (!out) return ACMI_E_ARG;
        acmi_handle *h = new (std::nothrow) acmi_handle();
        if (!h) return ACMI_E_NOMEM;
        h->device_request = device;
        h->init_ok = 1;
        {
            std::lock_guard<std::mutex> g(g_live_mutex);
            g_live.insert(h);
        }
        *out = h;
        return ACMI_OK;
    });
}

int acmi_add_patterns(acmi_handle *h, const acmi_pattern *patterns, size_t n, size_t *accepted) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (h->init_ok != 1) return ACMI_E_NOT_INIT;
        if (h->aut.finalized()) return ACMI_E_FINALIZED;
        if (n && !patterns) return ACMI_E_ARG;
        std::vector<const uint8_t *> vals(n);
        std::vector<uint32_t> lens(n), types(n);
        std::vector<int64_t> ids(n);
        std::vector<uint64_t> users(n);
        for (size_t k = 0; k < n; ++k) {
            if (patterns[k].length && !patterns[k].value) return ACMI_E_ARG;
            vals[k] = patterns[k].value;
            lens[k] = patterns[k].length;
            types[k] = patterns[k].id_type;
            ids[k] = patterns[k].id_number;
            users[k] = patterns[k].user;
        }
        size_t acc = h->aut.add_batch(vals.data(), lens.data(), types.data(), ids.data(), users.data(), n);
        if (accepted) *accepted = acc;
        return ACMI_OK;
    });
}

int acmi_finalize(acmi_handle *h) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (h->init_ok != 1) return ACMI_E_NOT_INIT;
        return ensure_finalized(h);
    });
}

int acmi_is_valid(const acmi_handle *h) { return live(h) && h->init_ok == 1 ? 1 : 0; }

int acmi_destroy(acmi_handle *h) {
    return guarded(h, [&]() -> int {
        {
            std::lock_guard<std::mutex> g(g_live_mutex);
            if (!h || !g_live.erase(h)) return ACMI_E_INVALID;
        }
        h->init_ok = 0;
        for (acmi_handle *&c : h->part)
            if (c) { (void)acmi_destroy(c); c = nullptr; }
        if (h->dev_touched) {  // also after a failed upload: release whatever was created
            (void)hipSetDevice(h->dev_id);
            if (h->stream) (void)hipStreamSynchronize(h->stream);
            for (DevBuf *b : {&h->d_table, &h->d_lds, &h->d_class, &h->d_own, &h->d_dict, &h->d_cnt, &h->d_plen, &h->d_anchor, &h->d_bitmap, &h->d_out_off, &h->d_out_list, &h->d_cold_rec, &h->d_cold_edges, &h->d_chain_byte, &h->d_chain_run, &h->d_slot_pair, &h->d_pairs, &h->d_pat_bytes, &h->d_dense_img, &h->d_merge_counts, &h->d_merge_sync, &h->d_rep_off, &h->d_rep_bytes, &h->d_plan, &h->d_plan_out, &h->d_tail_out, &h->d_rep_sync, &h->d_nom_count, &h->d_out_len, &h->d_rep_out_off, &h->d_rep_out, &h->d_blob,
                              &h->d_off, &h->d_sync[0], &h->d_sync[1], &h->d_item_base[0], &h->d_item_hay[0], &h->d_item_rec[0], &h->d_zero[0], &h->d_isync[0],
                              &h->d_item_base[1], &h->d_item_hay[1], &h->d_item_rec[1], &h->d_zero[1], &h->d_isync[1],
                              &h->d_rec_base, &h->d_slab, &h->d_ovf, &h->d_counters, &h->d_first, &h->d_tile_queue, &h->d_tile_sync, &h->d_uni_off[0], &h->d_uni_off[1], &h->fly[0].d_prefix, &h->fly[0].d_out, &h->fly[1].d_prefix, &h->fly[1].d_out, &h->d_runs, &h->d_run_count})
                b->release();
            for (void *q : h->retired) (void)hipFree(q);
            for (auto &evt : h->ev)
                if (evt) (void)hipEventDestroy(evt);
            for (auto &slot : h->fly)
                for (auto &evt : slot.ev)
                    if (evt) (void)hipEventDestroy(evt);
            if (h->pinned) (void)hipHostFree(h->pinned);
            if (h->pin_small) (void)hipHostFree(h->pin_small);
            if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
            if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
            for (auto &evt : h->ev_items)
                if (evt) (void)hipEventDestroy(evt);
            for (auto &evt : h->ev_scanned)
                if (evt) (void)hipEventDestroy(evt);
        }
        delete h;
        return ACMI_OK;
    });
}

int acmi_match_batch_device(acmi_handle *h, const void *d_blob, uint64_t blob_bytes, const void *d_offsets,
                            size_t n_haystacks, int find_all, uint64_t *n_records) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !d_offsets) return ACMI_E_ARG;
        rc = run_batch(h, static_cast<const uint8_t *>(d_blob), blob_bytes, static_cast<const uint64_t *>(d_offsets),
                       n_haystacks, find_all);
        if (rc < 0 && rc != ACMI_E_PARTIAL) return rc;
        if (n_records) *n_records = h->last_total;
        return rc;
    });
}

int acmi_match_batch_device_begin(acmi_handle *h, const void *d_blob, uint64_t blob_bytes, const void *d_offsets,
                                  size_t n_haystacks, int find_all) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !d_offsets) return ACMI_E_ARG;
        return run_batch_begin(h, static_cast<const uint8_t *>(d_blob), blob_bytes, static_cast<const uint64_t *>(d_offsets),
                               n_haystacks, find_all);
    });
}

int acmi_match_batch_device_begin_uniform(acmi_handle *h, const void *d_blob, uint64_t haystack_bytes, size_t n_haystacks, int find_all) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (haystack_bytes >= 0xFFFFFFFFull) { h->last_error = "haystack of 4 GiB or more"; return ACMI_E_ARG; }
        if (h->begin_seq - h->end_seq >= 2) { h->last_error = "two batches are in flight already: end the older one first"; return ACMI_E_ARG; }
        // the offsets array of this geometry: the slot's own copy, written again only when the geometry changes
        const int slot = int((h->begin_seq + 1) & 1);
        if (n_haystacks && (h->uni_nh[slot] != n_haystacks || h->uni_len[slot] != haystack_bytes || !h->d_uni_off[slot].p)) {
            HIPCHK(h, h->d_uni_off[slot].ensure_keep((n_haystacks + 1) * 8, h->retired));
            launch_uniform_offsets(h->d_uni_off[slot].as<uint64_t>(), n_haystacks, haystack_bytes, h->stream);
            HIPCHK(h, hipGetLastError());
            h->uni_nh[slot] = n_haystacks;
            h->uni_len[slot] = haystack_bytes;
        }
        h->uniform_len = haystack_bytes;
        rc = run_batch_begin(h, static_cast<const uint8_t *>(d_blob), haystack_bytes * n_haystacks, h->d_uni_off[slot].as<uint64_t>(), n_haystacks, find_all);
        h->uniform_len = 0;
        return rc;
    });
}

int acmi_match_batch_device_end(acmi_handle *h, uint64_t *n_records) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        HIPCHK(h, hipSetDevice(h->dev_id));
        int rc = run_batch_end(h);
        if (rc < 0 && rc != ACMI_E_PARTIAL) return rc;
        if (n_records) *n_records = h->last_total;
        return rc;
    });
}

int acmi_result_device(acmi_handle *h, const void **d_prefix, const void **d_records, uint64_t *n_records) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!h->have_result) return ACMI_E_ARG;
        if (d_prefix) *d_prefix = h->d_prefix().p;
        if (d_records) *d_records = h->d_out().p;
        if (n_records) *n_records = h->last_total;
        return ACMI_OK;
    });
}

int acmi_result_to_host(acmi_handle *h, uint64_t *prefix, acmi_record *records) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!h->have_result) return ACMI_E_ARG;
        HIPCHK(h, hipSetDevice(h->dev_id));
        hipStream_t st = h->stream;
        if (prefix) HIPCHK(h, hipMemcpyAsync(prefix, h->d_prefix().p, (size_t(h->last_nh) + 1) * 8, hipMemcpyDeviceToHost, st));
        if (records && h->last_total)
            HIPCHK(h, hipMemcpyAsync(records, h->d_out().p, h->last_total * sizeof(acmi_record), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        return ACMI_OK;
    });
}

int acmi_result_to_device(acmi_handle *h, void *d_prefix_dst, void *d_records_dst) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!h->have_result) return ACMI_E_ARG;
        HIPCHK(h, hipSetDevice(h->dev_id));
        hipStream_t st = h->stream;
        if (d_prefix_dst)
            HIPCHK(h, hipMemcpyAsync(d_prefix_dst, h->d_prefix().p, (size_t(h->last_nh) + 1) * 8, hipMemcpyDeviceToDevice, st));
        if (d_records_dst && h->last_total)
            HIPCHK(h, hipMemcpyAsync(d_records_dst, h->d_out().p, h->last_total * sizeof(acmi_record), hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipStreamSynchronize(st));
        return ACMI_OK;
    });
}

int acmi_match_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t n_haystacks, int find_all,
                     const uint64_t **prefix, const acmi_record **records, size_t *n_records) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !offsets) return ACMI_E_ARG;
        uint64_t blob_bytes = 0;
        if (n_haystacks) {
            for (size_t k = 0; k < n_haystacks; ++k) {
                if (offsets[k + 1] < offsets[k]) { h->last_error = "offsets must be non-decreasing"; return ACMI_E_ARG; }
                if (offsets[k + 1] - offsets[k] >= 0xFFFFFFFFull) { h->last_error = "haystack of 4 GiB or more"; return ACMI_E_ARG; }
            }
            blob_bytes = offsets[n_haystacks];
            if (blob_bytes && !blob) return ACMI_E_ARG;
        }
        if (n_haystacks) {
            rc = run_small(h, blob, offsets, n_haystacks, find_all);
            if (rc < 0) return rc;
            if (rc == ACMI_OK) {
                if (prefix) *prefix = h->h_prefix.data();
                if (records) *records = h->h_records.data();
                if (n_records) *n_records = h->last_total;
                return ACMI_OK;
            }
        }
        hipStream_t st = h->stream;
        // stage the borrowed host buffers (php_ahocorasick.c:725-727 borrows the zend_string)
        HIPCHK(h, h->d_blob.ensure(blob_bytes + 64));
        HIPCHK(h, h->d_off.ensure((n_haystacks + 1) * 8));
        if (blob_bytes) HIPCHK(h, hipMemcpyAsync(h->d_blob.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
        if (n_haystacks) HIPCHK(h, hipMemcpyAsync(h->d_off.p, offsets, (n_haystacks + 1) * 8, hipMemcpyHostToDevice, st));
        rc = run_batch(h, h->d_blob.as<uint8_t>(), blob_bytes, h->d_off.as<uint64_t>(), n_haystacks, find_all);
        if (rc < 0 && rc != ACMI_E_PARTIAL) return rc;
        // drain: device match buffer -> host records
        h->h_prefix.assign(n_haystacks + 1, 0);
        if (rc == ACMI_OK) {
            h->h_records.resize(h->last_total);
            rc = acmi_result_to_host(h, h->h_prefix.data(), h->h_records.data());
            if (rc < 0) return rc;
        } else {
            // "max_result_bytes": the device hands out the haystacks whose records fit its result buffer; the rest of the batch is
            // matched again from the first haystack that did not, until all of it is drained (the overflow -> drain -> resume of
            // BASELINE configs[4]: the host's arrays are the only ones that hold the whole result)
            h->h_records.clear();
            size_t first = 0;
            uint64_t before = 0;
            std::vector<uint64_t> seg;
            for (;;) {
                const size_t k = rc == ACMI_E_PARTIAL ? size_t(h->resume_haystack) : n_haystacks - first;
                const uint64_t n = h->last_total;
                seg.assign(k + 1, 0);
                const size_t at = h->h_records.size();
                h->h_records.resize(at + n);
                const int rr = acmi_result_to_host(h, seg.data(), n ? h->h_records.data() + at : nullptr);
                if (rr < 0) return rr;
                for (size_t i = 0; i <= k; ++i) h->h_prefix[first + i] = before + seg[i];
                if (first)
                    for (size_t i = at; i < at + n; ++i) h->h_records[i].hay += uint32_t(first);
                before += n;
                first += k;
                if (rc != ACMI_E_PARTIAL) break;
                rc = run_batch(h, h->d_blob.as<uint8_t>(), blob_bytes, h->d_off.as<uint64_t>() + first, n_haystacks - first, find_all);
                if (rc < 0 && rc != ACMI_E_PARTIAL) return rc;
            }
            h->last_total = before;
            h->last_nh = uint32_t(n_haystacks);
            h->have_result = false;   // (the whole result is in host memory only)
        }
        if (prefix) *prefix = h->h_prefix.data();
        if (records) *records = h->h_records.data();
        if (n_records) *n_records = h->last_total;
        return ACMI_OK;
    });
}

int acmi_match(acmi_handle *h, const uint8_t *text, size_t length, int find_all, const acmi_record **records,
               size_t *n_records) {
    return guarded(h, [&]() -> int {
        const uint64_t off[2] = {0, length};
        if (length >= 0xFFFFFFFFull) return live(h) ? ACMI_E_ARG : ACMI_E_INVALID;
        return acmi_match_batch(h, text, off, 1, find_all, nullptr, records, n_records);
    });
}

int acmi_stream_open(acmi_handle *h, acmi_stream **out) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!out) return ACMI_E_ARG;
        acmi_stream *s = new (std::nothrow) acmi_stream();
        if (!s) return ACMI_E_NOMEM;
        *out = s;
        return ACMI_OK;
    });
}

int acmi_stream_feed(acmi_handle *h, acmi_stream *s, const uint8_t *data, size_t length, const acmi_record **records,
                     size_t *n_records) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (!s || (length && !data)) return ACMI_E_ARG;
        if (s->base + length >= 0xFFFFFFFFull) { h->last_error = "stream of 4 GiB or more"; return ACMI_E_TOO_LARGE; }
        const size_t carry = s->carry.size();
        s->buf.resize(carry + length);
        if (carry) std::memcpy(s->buf.data(), s->carry.data(), carry);
        if (length) std::memcpy(s->buf.data() + carry, data, length);
        const acmi_record *recs = nullptr;
        size_t n = 0;
        rc = acmi_match(h, s->buf.data(), s->buf.size(), 1, &recs, &n);
        if (rc < 0) return rc;
        // events that end inside the carry were reported by the feed that delivered those bytes
        s->out.clear();
        const uint32_t shift = uint32_t(s->base - carry);  // buffer offset -> stream offset
        for (size_t i = 0; i < n; ++i) {
            if (recs[i].pos <= carry) continue;
            acmi_record r = recs[i];
            r.pos += shift;
            r.start += shift;
            s->out.push_back(r);
        }
        s->base += length;
        const uint32_t lmax = h->aut.flat().max_pattern_length;
        const size_t keep = std::min<size_t>(lmax ? lmax - 1 : 0, s->buf.size());
        s->carry.assign(s->buf.end() - keep, s->buf.end());
        if (records) *records = s->out.data();
        if (n_records) *n_records = s->out.size();
        return ACMI_OK;
    });
}

uint64_t acmi_stream_position(const acmi_stream *s) { return s ? s->base : 0; }

void acmi_stream_close(acmi_stream *s) { delete s; }

size_t acmi_pattern_count(const acmi_handle *h) { return live(h) ? h->aut.patterns().size() : 0; }

int acmi_pattern_info(const acmi_handle *h, uint32_t index, acmi_pattern *out) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!out || index >= h->aut.patterns().size()) return ACMI_E_ARG;
        const PatternMeta &m = h->aut.patterns()[index];
        out->value = h->aut.pattern_bytes().data() + m.byte_off;
        out->length = m.length;
        out->id_type = m.id_type;
        out->id_number = m.id_number;
        out->user = m.user;
        return m.accepted ? 1 : 0;
    });
}

// One option validated and applied to ONE handle (acmi_set_option logs and forwards it afterwards).
static int apply_option(acmi_handle *h, const std::string &k, int64_t value) {
        if (k == "lds_bytes") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 2048 || value > 160 * 1024 - 64) return ACMI_E_ARG;
            h->bopt.lds_bytes = uint32_t(value);
        } else if (k == "slab_events") {
            if (value < 1 || value > 4096) return ACMI_E_ARG;
            h->slab_events = uint32_t(value);
            h->slab_set = true;
        } else if (k == "run_cap") {
            if (value < 1 || value > 1024) return ACMI_E_ARG;
            h->run_cap = uint32_t(value);
            h->run_cap_set = true;
        } else if (k == "chunk_bytes") {
            if (value < 0) return ACMI_E_ARG;
            h->chunk_opt = uint64_t(value);
        } else if (k == "kernel") {
            if (value < 0 || value > 6 || value == 5) return ACMI_E_ARG;
            h->kernel = int(value);
            for (auto &row : h->choice)
                for (auto &c : row) c = acmi_handle::ChoiceClass{};
        } else if (k == "reserve_cus") {
            if (value < 0 || value > 128) return ACMI_E_ARG;
            h->reserve_cus = int(value);
        } else if (k == "max_result_bytes") {
            if (value != 0 && value < 4096) return ACMI_E_ARG;
            h->max_result_bytes = uint64_t(value);
        } else if (k == "part_kernel") {   // split sets: "kernel" of the short part alone (the long part keeps the filter); -1 automatic
            if (value < -1 || value > 6 || value == 5 || value == 2) return ACMI_E_ARG;
            h->part_kernel = int(value);
            if (h->is_part && h->aut.accepted_count() && h->aut.flat().max_pattern_length < 7 && h->aut.finalized()) return apply_option(h, "kernel", value < 0 ? 3 : value);
        } else if (k == "tile_w") {   // k_walk_tiles: bytes per lane (0: automatic)
            if (value != 0 && value != 16 && value != 32 && value != 64) return ACMI_E_ARG;
            h->tile_w = int(value);
        } else if (k == "tile_p") {   // k_walk_tiles: lanes per work item (0: automatic)
            if (value < 0 || value > 64) return ACMI_E_ARG;
            h->tile_p = int(value);
        } else if (k == "ilp") {
            if (value != 1 && value != 2) return ACMI_E_ARG;
            h->ilp = int(value);
        } else if (k == "group_quads") {
            if (value != 4 && value != 8) return ACMI_E_ARG;
            h->group_quads = int(value);
        } else if (k == "small_path") {
            h->small_path = value ? 1 : 0;
        } else if (k == "wave_items") {
            if (value < 0 || value > 64) return ACMI_E_ARG;
            h->wave_items = int(value);
        } else if (k == "persistent") {
            h->persistent = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "tiny_rows") {
            h->tiny_rows = value ? 1 : 0;
        } else if (k == "short_rows") {
            h->short_rows = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "walk_group") {
            if (value != 1 && value != 4 && value != 8) return ACMI_E_ARG;
            h->walk_group = int(value);
        } else if (k == "early_stop") {
            if (value < 0 || value > 2) return ACMI_E_ARG;
            h->early_stop = int(value);
            if (h->aut.finalized())
                h->dA.early_stop = (value == 2 || (value == 1 && h->aut.flat().early_stop_useful)) ? 1u : 0u;
        } else if (k == "refill") {
            if (value < 0 || value > 2) return ACMI_E_ARG;
            h->refill = int(value);
        } else if (k == "profile") {
            if (value < 0 || value > 2) return ACMI_E_ARG;
            h->profile = int(value);
        } else if (k == "compact") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->bopt.compact = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "anchor_slots2") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->bopt.anchor_slots2 = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "anchor_dense") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->bopt.anchor_dense = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "anchor_choice") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->bopt.anchor_choice = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "dense_bytes") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 0) return ACMI_E_ARG;
            h->bopt.dense_bytes = uint64_t(value);
        } else if (k == "verify") {
            if (h->dev_ready) return ACMI_E_FINALIZED;  // the tables are uploaded (or not) by the first match call
            h->verify = value ? 1 : 0;
        } else if (k == "items_ahead") {
            h->items_ahead = value ? 1 : 0;
        } else if (k == "dense_hot") {
            if (h->dev_ready) return ACMI_E_FINALIZED;
            h->dense_hot = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "device_build") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->device_build = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "keep_host_tables") {
            h->keep_host_tables = value ? 1 : 0;
        } else if (k == "max_flat_outputs") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 0) return ACMI_E_ARG;
            h->bopt.max_flat_outputs = uint64_t(value);
        } else if (k == "max_table_bytes") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 1024) return ACMI_E_ARG;
            h->bopt.max_table_bytes = uint64_t(value);
        } else {
            return ACMI_E_ARG;
        }
        return ACMI_OK;
}

int acmi_set_option(acmi_handle *h, const char *key, int64_t value) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!key) return ACMI_E_ARG;
        const std::string k(key);
        if (k == "split") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->split_opt = value < 0 ? -1 : (value ? 1 : 0);
            return ACMI_OK;
        }
        // this handle first: a value it rejects is neither logged (the log is replayed on the parts of a split set) nor
        // forwarded, so that the handle and its parts cannot come to differ
        int rc = apply_option(h, k, value);
        if (rc != ACMI_OK) return rc;
        if (!h->is_part) {
            h->option_log.emplace_back(k, value);
            for (acmi_handle *c : h->part) {   // ("any time" options reach the parts of a split set too)
                if (!c) continue;
                const int rc2 = acmi_set_option(c, key, value);
                if (rc2 < 0) { h->last_error = "a part of the split set refused the option"; rc = rc2; }
            }
        }
        return rc;
    });
}

int acmi_get_stat(const acmi_handle *h, const char *key, double *value) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!key || !value) return ACMI_E_ARG;
        const std::string k(key);
        if (k == "split") { *value = h->part[0] ? 1 : 0; return ACMI_OK; }
        if (k == "merged_batches") { *value = h->merged_batches; return ACMI_OK; }
        if (k == "resume_haystack") { *value = double(h->resume_haystack); return ACMI_OK; }
        if (k == "split_long_patterns" || k == "split_short_patterns") {
            *value = h->part[0] ? double(h->part[k == "split_long_patterns" ? 0 : 1]->aut.accepted_count()) : 0.0;
            return ACMI_OK;
        }
        if (h->part[0] && (k.rfind("anchor", 0) == 0 || k == "compact" || k == "built_on_device"))
            return acmi_get_stat(h->part[0], key, value);   // (the filter is the long part's)
        const FlatAutomaton &fa = h->aut.flat();
        if (k == "nodes") *value = double(h->aut.node_count());
        else if (k == "accepted") *value = double(h->aut.accepted_count());
        else if (k == "patterns") *value = double(h->aut.patterns().size());
        else if (k == "finalized") *value = h->aut.finalized() ? 1 : 0;
        else if (k == "est_flag_rate") *value = fa.est_flag_rate;
        else if (k == "states") *value = fa.n_states;
        else if (k == "hot_states") *value = fa.trap;
        else if (k == "columns") *value = fa.n_columns;
        else if (k == "entry_bytes") *value = fa.entry_bytes;
        else if (k == "class_mode") *value = fa.class_mode;
        else if (k == "lds_image_bytes") *value = double(fa.lds_image_bytes + (fa.class_mode == kClassTable ? 512 : 0));
        else if (k == "table_bytes") *value = double(fa.table_bytes);
        else if (k == "automaton_bytes") *value = double(fa.table_bytes + (fa.compact ? uint64_t(fa.n_states) * 8 : 0));
        else if (k == "max_pattern_length") *value = fa.max_pattern_length;
        else if (k == "used_bytes") *value = fa.n_used_bytes;
        else if (k == "items") *value = double(h->items);
        else if (k == "chunk_bytes") *value = double(h->chunk_used);
        else if (k == "overflow_items") *value = h->overflow_items;
        else if (k == "walked_bytes") *value = double(h->walked_bytes);
        else if (k == "fallback_bytes") *value = double(h->fallback_bytes);
        else if (k == "choice_probes") *value = h->probes;
        else if (k == "choice_switches") *value = h->switches;
        else if (k == "scan_ms") *value = h->scan_ms;
        else if (k == "walk_ms") *value = h->walk_ms;
        else if (k == "kernel_used") *value = h->kernel_used;
        else if (k == "anchor_variant") *value = h->anchor_variant;
        else if (k == "tiny_waves") *value = h->tiny_waves;
        else if (k == "anchor_q") *value = fa.anchor_q;
        else if (k == "anchor_keys") *value = fa.anchor_keys;
        else if (k == "anchor_mode") *value = fa.anchor_mode;
        else if (k == "min_pattern_length") *value = fa.min_pattern_length;
        else if (k == "total_ms") *value = h->total_ms;
        else if (k == "records") *value = double(h->last_total);
        else if (k == "flat_outputs") *value = double(fa.n_flat_outputs);
        else if (k == "max_out_count") *value = fa.max_out_count;
        else if (k == "anchor_pairs") *value = double(fa.anchor_pairs_count);
        else if (k == "anchor_choice") *value = fa.anchor_choice;
        else if (k == "anchor_dense") *value = fa.anchor_dense;
        else if (k == "anchor_slots") *value = fa.anchor_slots;
        else if (k == "anchor_dense_rate") *value = fa.anchor_dense_rate;
        else if (k == "dense_image_bytes") *value = double(h->dA.dense_image_words) * 4;
        else if (k == "built_on_device") *value = h->aut.built_on_device() ? 1 : 0;
        else if (k == "compact") *value = fa.compact;
        else if (k == "dense_rows") *value = fa.n_dense_rows;
        else if (k == "cold_edges") *value = double(fa.cold_edges.size());
        else return ACMI_E_ARG;
        return ACMI_OK;
    });
}

int acmi_set_stream(acmi_handle *h, void *hip_stream) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (h->begin_seq != h->end_seq) {  // a begun batch is ended on the stream it was enqueued on
            h->last_error = "acmi_set_stream while a batch is in flight";
            return ACMI_E_ARG;
        }
        h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
        for (acmi_handle *c : h->part)
            if (c) c->stream = h->stream;
        return ACMI_OK;
    });
}

int acmi_debug_tables(const acmi_handle *h, acmi_tables *out) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!out || !h->aut.finalized()) return ACMI_E_ARG;
        if (h->aut.host_tables_released()) return ACMI_E_ARG;  // uploaded and dropped: set "keep_host_tables" before the first match
        const FlatAutomaton &fa = h->aut.flat();
        out->n_states = fa.n_states;
        out->n_columns = fa.n_columns;
        out->entry_bytes = fa.entry_bytes;
        out->first_hot_final = fa.f1;
        out->trap = fa.trap;
        out->first_cold_final = fa.f2;
        out->class_mode = fa.class_mode;
        out->swar_shift = fa.swar_shift;
        out->swar_other = fa.swar_other;
        out->max_pattern_length = fa.max_pattern_length;
        out->table = fa.table.data();
        out->lds_image = fa.lds_image.data();
        out->class_of_byte = fa.class_of_byte;
        out->own_pattern = fa.own_pattern.data();
        out->dict_link = fa.dict_link.data();
        out->out_count = fa.out_count.data();
        out->anchor_q = fa.anchor_q;
        out->anchor_mul = fa.anchor_mul;
        out->min_pattern_length = fa.min_pattern_length;
        out->anchor_keys = fa.anchor_keys;
        out->anchor_mode = fa.anchor_mode;
        out->anchor_bucket_bits = fa.anchor_bucket_bits;
        out->anchor_bucket_shift = fa.anchor_bucket_shift;
        out->anchor_bitmap = fa.anchor_mode == 2 ? fa.anchor_bitmap.data() : nullptr;
        out->anchor_table = fa.anchor_q ? fa.anchor_table.data() : nullptr;
        out->anchor_choice = fa.anchor_choice;
        out->anchor_offsets = fa.anchor_offsets.empty() ? nullptr : fa.anchor_offsets.data();
        out->compact = fa.compact;
        out->dense_nf_end = fa.dense_nf_end;
        out->dense_f_end = fa.dense_f_end;
        out->n_dense_rows = fa.n_dense_rows;
        out->n_cold_edges = uint32_t(fa.cold_edges.size());
        out->cold_rec = fa.cold_rec.empty() ? nullptr : fa.cold_rec.data();
        out->cold_edges = fa.cold_edges.empty() ? nullptr : fa.cold_edges.data();
        out->chain_lo = fa.chain_lo;
        out->chain_byte = fa.chain_byte.empty() ? nullptr : fa.chain_byte.data();
        out->chain_run = fa.chain_run.empty() ? nullptr : fa.chain_run.data();
        out->n_flat_outputs = uint32_t(fa.out_list.size() / 2);
        out->out_off = fa.out_off.empty() ? nullptr : fa.out_off.data();
        out->out_list = fa.out_list.empty() ? nullptr : fa.out_list.data();
        out->anchor_dense = fa.anchor_dense;
        out->anchor_slots = fa.anchor_slots;
        return ACMI_OK;
    });
}

int acmi_debug_stream_read(acmi_handle *h, const void *d_src, uint64_t bytes, int repeats, double *ms_per_pass) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (!d_src || bytes < 16 || repeats < 1 || !ms_per_pass) return ACMI_E_ARG;
        HIPCHK(h, h->d_counters.ensure(64));
        hipStream_t st = h->stream;
        launch_stream_read(d_src, bytes, h->d_counters.as<uint32_t>(), st);  // warm-up
        HIPCHK(h, hipEventRecord(h->ev[0], st));
        for (int i = 0; i < repeats; ++i) launch_stream_read(d_src, bytes, h->d_counters.as<uint32_t>(), st);
        HIPCHK(h, hipEventRecord(h->ev[1], st));
        HIPCHK(h, hipEventSynchronize(h->ev[1]));
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
        *ms_per_pass = double(ms) / repeats;
        return ACMI_OK;
    });
}

const char *acmi_strerror(int status) {
    switch (status) {
    case ACMI_OK: return "ok";
    case ACMI_E_INVALID: return "Invalid resource.";
    case ACMI_E_NOT_INIT: return "Not initialized.";
    case ACMI_E_FINALIZED: return "automaton already finalized";
    case ACMI_E_ARG: return "bad argument";
    case ACMI_E_NOMEM: return "out of memory";
    case ACMI_E_DEVICE: return "HIP error";
    case ACMI_E_NO_DEVICE: return "no usable gfx950 device (no CPU fallback)";
    case ACMI_E_TOO_LARGE: return "automaton or result too large";
    case ACMI_E_NO_REPLACEMENT: return "no pattern has a replacement text";
    case ACMI_E_PARTIAL: return "result holds the first haystacks only (max_result_bytes)";
    default: return status > 0 ? "ok" : "unknown error";
    }
}

const char *acmi_last_error(const acmi_handle *h) { return live(h) ? h->last_error.c_str() : "Invalid resource."; }

const char *acmi_version(void) { return ACMI_VERSION; }

}  // extern "C"

// ---- multi-GPU: contiguous haystack shards, one exchange step ------------------------------------------
// RCCL is bound at run time (dlopen): libacmi.so has no link-time dependency on it, a host without RCCL simply gets
// ACMI_E_NO_DEVICE from acmi_comm_*.  torch.distributed's "nccl" backend is the same library: a process that has
// torch loaded reuses its copy.
//
// The exchange is SIZE-PROPORTIONAL: after a 32-byte header per rank (records, haystacks, status) every rank knows every
// shard's size, writes its own shard -- haystack ids and prefix words already global -- into ITS slice of the global
// arrays, and one group of per-rank ncclBroadcast calls (root r, exactly r's bytes, in place for the root) fills in the
// other slices: no padding to the largest shard, no staging copy, no unpacking kernel.  A rank that cannot take part (no
// result, wrong haystack count, out of memory) says so IN THE HEADER and still joins the header round, so that every rank
// returns an error together instead of leaving its peers inside a collective.
//
// The collectives sit behind a small transport interface with two implementations: RCCL (one process or thread per GPU),
// and a process-local fabric (acmi_comm_create_local: W communicators of one process, each used by its own thread, copies
// instead of collectives) -- the same packing / offset logic runs on a box with ONE GPU, which is how the W > 1 layout is
// tested where RCCL refuses several ranks per device.
namespace {

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, acmi_comm_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = std::getenv("ACMI_RCCL_LIBRARY");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {  // a copy that is already in the process first (torch's)
            if (n && (r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        }
        for (const char *n : names) {
            if (r.lib) break;
            if (n) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.error = "librccl.so not found (set ACMI_RCCL_LIBRARY)"; return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.Broadcast || !r.GroupStart || !r.GroupEnd || !r.CommDestroy)
            r.error = "librccl.so lacks a needed symbol";
    });
    return r;
}

constexpr int kNcclUint8 = 1;  // ncclDataType_t (rccl.h:460)

// One broadcast of a group: `bytes` from rank `root`'s `send` into every rank's `recv` (the root passes send == recv).
struct BcastOp {
    const void *send;
    void *recv;
    size_t bytes;
    int root;
};

// The process-local fabric: W communicators of one process meet here.  Every collective is a rendezvous (all ranks
// post their buffers, wait for each other, copy what they receive on their own stream, synchronise it, meet again):
// as blocking as can be, which is all a test transport and a single-process multi-GPU host need.
struct LocalFabric {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, refs = 0;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;                 // all-gather: per rank
    std::vector<std::vector<BcastOp>> ops;          // broadcast group: per rank
    std::vector<hipEvent_t> ready;                  // per rank: its send data is complete on its stream
    void barrier() {
        std::unique_lock<std::mutex> g(m);
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(g, [&] { return generation != gen; });
    }
};

}  // namespace

struct acmi_comm {
    void *comm = nullptr;           // ncclComm_t (RCCL transport)
    LocalFabric *fabric = nullptr;  // process-local transport
    int rank = 0, world = 1, dev = 0;
    int backend_ranks = 1;          // ranks the backend itself reports (ncclCommCount)
    DevBuf d_hdr, d_prefix, d_records, d_wire;   // (d_wire: every rank's records as they travel, 8 bytes each {pos, pattern})
    uint64_t *h_hdr = nullptr;      // pinned: [mine 4 | all world x 4] words
    // capacities every rank holds, kept identical on all ranks (they grow by the same rule from the same headers)
    uint64_t cap_rec = 0, cap_hay = 0;
    bool cap_ok = true;             // this rank's buffers really have that capacity
    uint64_t total_rec = 0, total_hay = 0;
    uint64_t bytes_sent = 0, bytes_received = 0, rounds = 0;  // of the last exchange (statistics)
    std::string last_error;
};

namespace {

constexpr uint64_t kHdrWords = 4;  // {records, haystacks, status, reserved}
enum : uint64_t { kCommOk = 0, kCommNoResult = 1, kCommNoMemory = 2, kCommDevice = 3 };

int comm_fail(acmi_handle *h, acmi_comm *c, int code, const std::string &text) {
    c->last_error = text;
    if (h && live(h)) h->last_error = text;
    return code;
}

// all-gather of `bytes` per rank (device buffers), enqueued on `st`
int comm_allgather(acmi_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st) {
    if (c->world == 1) {
        if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ACMI_E_DEVICE;
        return ACMI_OK;
    }
    if (c->fabric) {
        LocalFabric &F = *c->fabric;
        // (an error on this rank travels THROUGH both barriers: a rank that left early would leave its peers waiting for ever)
        int rc = hipEventRecord(F.ready[size_t(c->rank)], st) == hipSuccess ? ACMI_OK : ACMI_E_DEVICE;
        F.send[size_t(c->rank)] = send;
        F.barrier();
        for (int r = 0; r < c->world && rc == ACMI_OK; ++r) {
            if (hipStreamWaitEvent(st, F.ready[size_t(r)], 0) != hipSuccess ||
                hipMemcpyAsync(static_cast<uint8_t *>(recv) + size_t(r) * bytes, F.send[size_t(r)], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                rc = ACMI_E_DEVICE;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = ACMI_E_DEVICE;
        F.barrier();  // nobody reuses its send buffer before everybody has copied it
        return rc;
    }
    Rccl &R = rccl();
    const int e = R.AllGather(send, recv, bytes, kNcclUint8, c->comm, st);
    if (e != 0) { c->last_error = std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
    return ACMI_OK;
}

// one group of broadcasts; every rank passes the same list (same order, sizes and roots).  dest >= 0: only rank `dest` receives
// (a gather: a send from every root to dest, ncclSend / ncclRecv in one group)
int comm_bcast_group(acmi_comm *c, const std::vector<BcastOp> &ops, hipStream_t st, int dest = -1) {
    if (c->world == 1 || ops.empty()) return ACMI_OK;
    if (c->fabric) {
        LocalFabric &F = *c->fabric;
        int rc = hipEventRecord(F.ready[size_t(c->rank)], st) == hipSuccess ? ACMI_OK : ACMI_E_DEVICE;
        F.ops[size_t(c->rank)] = ops;
        F.barrier();
        for (size_t i = 0; i < ops.size() && rc == ACMI_OK; ++i) {
            const BcastOp &mine = ops[i];
            if (mine.root == c->rank || mine.bytes == 0 || (dest >= 0 && dest != c->rank)) continue;
            const BcastOp &theirs = F.ops[size_t(mine.root)][i];
            if (hipStreamWaitEvent(st, F.ready[size_t(mine.root)], 0) != hipSuccess ||
                hipMemcpyAsync(mine.recv, theirs.send, mine.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                rc = ACMI_E_DEVICE;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = ACMI_E_DEVICE;
        F.barrier();
        return rc;
    }
    Rccl &R = rccl();
    if (dest >= 0 && (!R.Send || !R.Recv)) { c->last_error = "librccl.so lacks ncclSend / ncclRecv"; return ACMI_E_NO_DEVICE; }
    int e = R.GroupStart();
    for (size_t i = 0; i < ops.size() && e == 0; ++i) {
        if (!ops[i].bytes) continue;
        if (dest < 0) e = R.Broadcast(ops[i].send, ops[i].recv, ops[i].bytes, kNcclUint8, ops[i].root, c->comm, st);
        else if (ops[i].root == dest) continue;
        else if (c->rank == ops[i].root) e = R.Send(ops[i].send, ops[i].bytes, kNcclUint8, dest, c->comm, st);
        else if (c->rank == dest) e = R.Recv(ops[i].recv, ops[i].bytes, kNcclUint8, ops[i].root, c->comm, st);
    }
    const int e2 = R.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) { c->last_error = std::string("ncclBroadcast: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
    return ACMI_OK;
}

// header round: every rank's {records, haystacks, status} -> c->h_hdr[4 ..]; the exchange's one host wait
int comm_headers(acmi_comm *c, uint64_t n_rec, uint64_t n_hay, uint64_t status, hipStream_t st) {
    const int W = c->world;
    uint64_t *mine = c->h_hdr, *all = c->h_hdr + kHdrWords;
    mine[0] = n_rec; mine[1] = n_hay; mine[2] = status; mine[3] = 0;
    uint64_t *d_mine = c->d_hdr.as<uint64_t>(), *d_all = d_mine + kHdrWords;
    if (hipMemcpyAsync(d_mine, mine, kHdrWords * 8, hipMemcpyHostToDevice, st) != hipSuccess) return ACMI_E_DEVICE;
    int rc = comm_allgather(c, d_mine, d_all, kHdrWords * 8, st);
    if (rc < 0) return rc;
    if (hipMemcpyAsync(all, d_all, size_t(W) * kHdrWords * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return ACMI_E_DEVICE;
    if (hipStreamSynchronize(st) != hipSuccess) return ACMI_E_DEVICE;
    ++c->rounds;
    return ACMI_OK;
}

acmi_comm *comm_new(acmi_handle *h, int rank, int world) {
    acmi_comm *c = new acmi_comm();
    c->rank = rank;
    c->world = world;
    c->dev = h->dev_id;
    c->backend_ranks = world;
    // the header buffers exist from the start: the header round itself must not be able to fail on an allocation
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_hdr), size_t(world + 1) * kHdrWords * 8, hipHostMallocDefault) != hipSuccess ||
        c->d_hdr.ensure(size_t(world + 1) * kHdrWords * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        c->d_hdr.release();
        delete c;
        return nullptr;
    }
    return c;
}

}  // namespace

extern "C" {

int acmi_comm_get_id(acmi_comm_id *id) {
    return guarded(nullptr, [&]() -> int {
        if (!id) return ACMI_E_ARG;
        Rccl &R = rccl();
        if (!R.error.empty()) return ACMI_E_NO_DEVICE;
        return R.GetUniqueId(id) == 0 ? ACMI_OK : ACMI_E_DEVICE;
    });
}

int acmi_comm_create(acmi_handle *h, int rank, int world, const acmi_comm_id *id, acmi_comm **out) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);  // the communicator lives on the handle's device
        if (rc < 0) return rc;
        if (!out || world < 1 || world > int(kMaxRanks) || rank < 0 || rank >= world || (world > 1 && !id)) return ACMI_E_ARG;
        acmi_comm *c = comm_new(h, rank, world);
        if (!c) return ACMI_E_NOMEM;
        if (world > 1) {
            Rccl &R = rccl();
            if (!R.error.empty()) { h->last_error = R.error; acmi_comm_destroy(c); return ACMI_E_NO_DEVICE; }
            const int e = R.CommInitRank(&c->comm, world, *id, rank);
            if (e != 0) {
                h->last_error = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(e) : "error");
                c->comm = nullptr;
                acmi_comm_destroy(c);
                return ACMI_E_DEVICE;
            }
            int n = 0;
            if (R.CommCount && R.CommCount(c->comm, &n) == 0) c->backend_ranks = n;
        }
        *out = c;
        return ACMI_OK;
    });
}

int acmi_comm_create_local(acmi_handle *const *handles, int world, acmi_comm **out) {
    return guarded(nullptr, [&]() -> int {
        if (!handles || !out || world < 1 || world > int(kMaxRanks)) return ACMI_E_ARG;
        for (int r = 0; r < world; ++r) {
            const int rc = check_ready(handles[r]);
            if (rc < 0) return rc;
        }
        LocalFabric *F = new LocalFabric();
        F->world = world;
        F->send.assign(size_t(world), nullptr);
        F->ops.resize(size_t(world));
        F->ready.assign(size_t(world), nullptr);
        int made = 0;
        for (int r = 0; r < world; ++r) {
            acmi_comm *c = nullptr;
            if (hipSetDevice(handles[r]->dev_id) == hipSuccess && hipEventCreateWithFlags(&F->ready[size_t(r)], hipEventDisableTiming) == hipSuccess)
                c = comm_new(handles[r], r, world);
            if (!c) {
                for (int k = 0; k < made; ++k) { out[k]->fabric = nullptr; acmi_comm_destroy(out[k]); out[k] = nullptr; }
                for (hipEvent_t e : F->ready) if (e) (void)hipEventDestroy(e);
                delete F;
                return ACMI_E_NOMEM;
            }
            c->fabric = F;
            ++F->refs;
            out[r] = c;
            ++made;
        }
        return ACMI_OK;
    });
}

int acmi_comm_destroy(acmi_comm *c) {
    return guarded(nullptr, [&]() -> int {
        if (!c) return ACMI_E_INVALID;
        (void)hipSetDevice(c->dev);
        if (c->comm) rccl().CommDestroy(c->comm);
        if (c->fabric) {
            bool last;
            {
                std::lock_guard<std::mutex> g(c->fabric->m);
                last = --c->fabric->refs == 0;
            }
            if (last) {
                for (hipEvent_t e : c->fabric->ready) if (e) (void)hipEventDestroy(e);
                delete c->fabric;
            }
        }
        for (DevBuf *b : {&c->d_hdr, &c->d_prefix, &c->d_records, &c->d_wire}) b->release();
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        delete c;
        return ACMI_OK;
    });
}

// root < 0: all-gatherv (every rank ends with the global arrays); else a gather to `root`
static int comm_exchange(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, int root, const void **d_prefix,
                         const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    return guarded(h, [&]() -> int {
        if (!c) return ACMI_E_ARG;
        if (!live(h)) return ACMI_E_INVALID;   // (no stream to join a collective on: the caller broke the contract)
        if (root >= c->world) return ACMI_E_ARG;
        HIPCHK(h, hipSetDevice(h->dev_id));
        hipStream_t st = h->stream;
        const int W = c->world;
        const bool receiver = root < 0 || root == c->rank;
        c->rounds = c->bytes_sent = c->bytes_received = 0;
        // what is wrong on THIS rank travels in the header; the rank joins the header round all the same
        uint64_t status = kCommOk;
        if (!h->have_result || n_haystacks_local != h->last_nh) status = kCommNoResult;
        else if (!c->cap_ok) status = kCommNoMemory;
        const uint64_t my_rec = status == kCommOk ? h->last_total : 0, my_hay = status == kCommOk ? h->last_nh : 0;
        int rc = comm_headers(c, my_rec, my_hay, status, st);
        if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "header exchange failed" : c->last_error);
        const uint64_t *all = c->h_hdr + kHdrWords;
        uint64_t rec_base[kMaxRanks + 1] = {0}, hay_base[kMaxRanks + 1] = {0};
        int bad = -1;
        for (int r = 0; r < W; ++r) {
            rec_base[r + 1] = rec_base[r] + all[size_t(r) * kHdrWords];
            hay_base[r + 1] = hay_base[r] + all[size_t(r) * kHdrWords + 1];
            if (all[size_t(r) * kHdrWords + 2] != kCommOk && bad < 0) bad = r;
        }
        if (bad >= 0) {  // every rank sees the same headers and leaves here together
            const uint64_t why = all[size_t(bad) * kHdrWords + 2];
            const std::string text = "rank " + std::to_string(bad) + (why == kCommNoResult ? " has no result for the haystack count it gave"
                                                                      : why == kCommNoMemory ? " is out of device memory" : " reported a device error");
            return comm_fail(h, c, bad == c->rank ? (why == kCommNoMemory ? ACMI_E_NOMEM : ACMI_E_ARG) : ACMI_E_DEVICE, text);
        }
        const uint64_t total_rec = rec_base[W], total_hay = hay_base[W];
        // Global arrays: grown by the same rule on every rank (the totals are the same everywhere), and a rank whose
        // allocation fails tells the others in a second, tiny round -- taken by all ranks or by none.  (A rank that only
        // sends -- a gather's non-root -- holds its own slice of the prefix words and of the wire, no more.)
        if (total_rec > c->cap_rec || total_hay > c->cap_hay || !c->d_prefix.p) {
            c->cap_rec = std::max(c->cap_rec, total_rec + total_rec / 4);
            c->cap_hay = std::max(c->cap_hay, total_hay + total_hay / 4);
            c->cap_ok = c->d_prefix.ensure((c->cap_hay + 1) * 8) == hipSuccess && c->d_wire.ensure(std::max<uint64_t>(16, c->cap_rec * 8)) == hipSuccess &&
                        (!receiver || c->d_records.ensure(std::max<uint64_t>(16, c->cap_rec * 16)) == hipSuccess);
            if (!c->cap_ok) (void)hipGetLastError();
            rc = comm_headers(c, 0, 0, c->cap_ok ? kCommOk : kCommNoMemory, st);
            if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "status exchange failed" : c->last_error);
            for (int r = 0; r < W; ++r)
                if (all[size_t(r) * kHdrWords + 2] != kCommOk) {
                    // (every rank: the next call tries again from scratch -- also the rank whose allocation failed, which
                    // would otherwise report "out of memory" in every later header round without ever trying again)
                    c->cap_rec = c->cap_hay = 0;
                    c->cap_ok = true;
                    c->d_prefix.release();
                    c->d_records.release();
                    c->d_wire.release();
                    return comm_fail(h, c, r == c->rank ? ACMI_E_NOMEM : ACMI_E_DEVICE, "rank " + std::to_string(r) + " is out of device memory");
                }
        }
        if (receiver && c->d_records.cap < std::max<uint64_t>(16, total_rec * 16)) {   // (a rank that was a gather's sender so far)
            if (c->d_records.ensure(std::max<uint64_t>(16, c->cap_rec * 16)) != hipSuccess) { (void)hipGetLastError(); return comm_fail(h, c, ACMI_E_NOMEM, "out of device memory for the gathered records"); }
        }
        // this rank's slice, already global: haystack ids + hay_base, prefix words + rec_base (and the closing word); its
        // records a second time in the 8-byte form that travels
        uint64_t *g_prefix = c->d_prefix.as<uint64_t>();
        uint4 *g_records = receiver ? c->d_records.as<uint4>() : nullptr;
        uint2 *g_wire = c->d_wire.as<uint2>();
        launch_gather_pack(h->d_prefix().as<uint64_t>(), h->d_out().as<uint4>(), my_hay, my_rec, hay_base[c->rank], rec_base[c->rank],
                           g_prefix + hay_base[c->rank], g_records ? g_records + rec_base[c->rank] : nullptr, g_wire + rec_base[c->rank],
                           g_prefix + total_hay, total_rec, st);
        HIPCHK(h, hipGetLastError());
        // one group: rank r's slices from rank r, exactly their size
        std::vector<BcastOp> ops;
        for (int r = 0; r < W; ++r) {
            const uint64_t nr = rec_base[r + 1] - rec_base[r], nh = hay_base[r + 1] - hay_base[r];
            ops.push_back({g_wire + rec_base[r], g_wire + rec_base[r], size_t(nr) * 8, r});
            ops.push_back({g_prefix + hay_base[r], g_prefix + hay_base[r], size_t(nh) * 8, r});
            if (r == c->rank) c->bytes_sent = (W > 1 && root != c->rank ? nr * 8 + nh * 8 : 0);
            else if (receiver) c->bytes_received += nr * 8 + nh * 8;
        }
        rc = comm_bcast_group(c, ops, st, root);
        if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "payload exchange failed" : c->last_error);
        ++c->rounds;
        // the other ranks' records, expanded where they belong: haystack from the prefix array, start = pos - pattern length
        if (receiver && W > 1)
            launch_gather_unpack(g_prefix, total_hay, g_wire, total_rec, rec_base[c->rank], rec_base[c->rank + 1], h->dA.pattern_length, g_records, st);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
        c->total_rec = receiver ? total_rec : 0;
        c->total_hay = receiver ? total_hay : 0;
        if (d_prefix) *d_prefix = receiver ? c->d_prefix.p : nullptr;
        if (d_records) *d_records = receiver ? c->d_records.p : nullptr;
        if (n_records) *n_records = total_rec;
        if (n_haystacks) *n_haystacks = total_hay;
        return ACMI_OK;
    });
}

int acmi_comm_allgather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, const void **d_prefix,
                               const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    return comm_exchange(c, h, n_haystacks_local, -1, d_prefix, d_records, n_records, n_haystacks);
}

int acmi_comm_gather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, int root, const void **d_prefix,
                            const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    if (root < 0) return ACMI_E_ARG;
    return comm_exchange(c, h, n_haystacks_local, root, d_prefix, d_records, n_records, n_haystacks);
}

int acmi_comm_result_to_host(acmi_comm *c, uint64_t *prefix, acmi_record *records) {
    return guarded(nullptr, [&]() -> int {
        if (!c) return ACMI_E_ARG;
        if (!c->d_prefix.p || (c->total_rec && !c->d_records.p)) return ACMI_E_ARG;   // (no gathered result on this rank)
        if (hipSetDevice(c->dev) != hipSuccess) return ACMI_E_DEVICE;
        if (prefix && hipMemcpy(prefix, c->d_prefix.p, (c->total_hay + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return ACMI_E_DEVICE;
        if (records && c->total_rec && hipMemcpy(records, c->d_records.p, c->total_rec * 16, hipMemcpyDeviceToHost) != hipSuccess) return ACMI_E_DEVICE;
        return ACMI_OK;
    });
}

int acmi_comm_get_stat(const acmi_comm *c, const char *key, double *value) {
    return guarded(nullptr, [&]() -> int {
        if (!c || !key || !value) return ACMI_E_ARG;
        const std::string k(key);
        if (k == "world") *value = c->world;
        else if (k == "rank") *value = c->rank;
        else if (k == "backend_ranks") *value = c->backend_ranks;          // ncclCommCount (RCCL transport)
        else if (k == "transport") *value = c->world == 1 ? 0 : (c->fabric ? 2 : 1);  // 0 none, 1 RCCL, 2 process-local
        else if (k == "rounds") *value = double(c->rounds);                // collectives of the last exchange
        else if (k == "bytes_sent") *value = double(c->bytes_sent);
        else if (k == "bytes_received") *value = double(c->bytes_received);
        else if (k == "records") *value = double(c->total_rec);
        else if (k == "haystacks") *value = double(c->total_hay);
        else return ACMI_E_ARG;
        return ACMI_OK;
    });
}

const char *acmi_comm_last_error(const acmi_comm *c) { return c ? c->last_error.c_str() : "no communicator"; }

}  // extern "C"

// ---- search and replace (multifast_replace, reference src/multifast/replace.c) ------------------------------------
extern "C" {

int acmi_set_replacement(acmi_handle *h, uint32_t pattern_index, const uint8_t *text, uint32_t length) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (h->init_ok != 1) return ACMI_E_NOT_INIT;
        if (h->aut.finalized()) return ACMI_E_FINALIZED;
        if (pattern_index >= h->aut.patterns().size() || (length && !text)) return ACMI_E_ARG;
        h->aut.set_replacement(pattern_index, text, length);
        return ACMI_OK;
    });
}

namespace {

// multifast_replace over a batch that is resident in device memory; the output stays there (h->d_rep_out / d_rep_out_off).
// One host wait: the match pipeline's own, then plan -> scan -> emit are enqueued behind it with an OPTIMISTIC output
// capacity (the largest output so far, at least the input's size plus a quarter) and the total read back at the end; only
// an output that outgrew the buffer costs a second emit.
int replace_on_device(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh, int mode,
                      uint64_t *out_bytes) {
    hipStream_t st = h->stream;
    if (!h->rep_uploaded) {
        std::vector<uint64_t> off = h->aut.replacement_off();
        off.resize(std::max<size_t>(1, h->aut.patterns().size()), ~0ull);
        for (size_t i = 0; i < h->aut.patterns().size(); ++i)
            if (!h->aut.patterns()[i].accepted) off[i] = ~0ull;
        HIPCHK(h, h->d_rep_off.ensure(off.size() * 8));
        HIPCHK(h, copy_h2d(h->d_rep_off.p, off.data(), off.size() * 8));
        const std::vector<uint8_t> &rb = h->aut.replacement_bytes();
        HIPCHK(h, h->d_rep_bytes.ensure(rb.size() + 16));
        if (!rb.empty()) HIPCHK(h, copy_h2d(h->d_rep_bytes.p, rb.data(), rb.size()));
        h->rep_uploaded = true;
    }
    *out_bytes = 0;
    if (nh == 0) {
        HIPCHK(h, h->d_rep_out_off.ensure(8));
        HIPCHK(h, hipMemsetAsync(h->d_rep_out_off.p, 0, 8, st));
        HIPCHK(h, h->d_rep_out.ensure(16));
        HIPCHK(h, hipStreamSynchronize(st));
        return ACMI_OK;
    }
    // 1. every match of the batch (results left on the device)
    int rc = run_batch(h, d_blob, blob_bytes, d_off, nh, 1);
    if (rc < 0) return rc;
    // 2. nominees and output lengths per haystack, 3. output offsets, 4. the output texts
    HIPCHK(h, h->d_plan.ensure(std::max<uint64_t>(8, h->last_total * 8)));
    HIPCHK(h, h->d_plan_out.ensure(std::max<uint64_t>(8, h->last_total * 8)));
    HIPCHK(h, h->d_tail_out.ensure(nh * 8));
    HIPCHK(h, h->d_nom_count.ensure(nh * 4));
    HIPCHK(h, h->d_out_len.ensure(nh * 8));
    HIPCHK(h, h->d_rep_out_off.ensure((nh + 1) * 8));
    const size_t sync_bytes = 8 + size_t(scan_tiles(nh + 1) + 1) * 8;
    HIPCHK(h, h->d_rep_sync.ensure(sync_bytes));
    HIPCHK(h, hipMemsetAsync(h->d_rep_sync.p, 0, sync_bytes, st));
    h->rep_cap = std::max<uint64_t>(h->rep_cap, blob_bytes + blob_bytes / 4 + 4096);
    HIPCHK(h, h->d_rep_out.ensure(h->rep_cap + 16));
    ReplaceArgs R{};
    R.blob = d_blob;
    R.offsets = d_off;
    R.n_haystacks = uint32_t(nh);
    R.mode = mode == 2 ? 2u : 1u;   // MF_REPLACE_MODE_DEFAULT behaves as NORMAL (replace.c:252-254)
    R.prefix = h->d_prefix().as<uint64_t>();
    R.records = h->d_out().as<uint4>();
    R.pattern_length = h->dA.pattern_length;
    R.rep_off = h->d_rep_off.as<uint64_t>();
    R.rep_bytes = h->d_rep_bytes.as<uint8_t>();
    R.plan = h->d_plan.as<uint2>();
    R.plan_out = h->d_plan_out.as<uint64_t>();
    R.tail_out = h->d_tail_out.as<uint64_t>();
    R.nom_count = h->d_nom_count.as<uint32_t>();
    R.out_len = h->d_out_len.as<uint64_t>();
    R.out_off = h->d_rep_out_off.as<uint64_t>();
    R.out_blob = h->d_rep_out.as<uint8_t>();
    // the total travels through the pinned block of the slot the match result does not use
    uint64_t *total_host = h->pinned + (kPinWords / 2) * (h->cur ^ 1);
    uint64_t *total_dev = reinterpret_cast<uint64_t *>(h->pinned_dev + kPinWords * (h->cur ^ 1));
    launch_replace_plan(R, st);
    launch_scan_u64(R.out_len, nh, h->d_rep_out_off.as<uint64_t>(), h->d_rep_sync.as<uint32_t>(), total_dev, st);
    launch_replace_emit(R, h->rep_cap, st);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(st));
    const uint64_t total = *total_host;
    if (total > h->rep_cap) {
        h->rep_cap = total + total / 8;
        HIPCHK(h, h->d_rep_out.ensure(h->rep_cap + 16));
        R.out_blob = h->d_rep_out.as<uint8_t>();
        launch_replace_emit(R, h->rep_cap, st);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
    }
    *out_bytes = total;
    return ACMI_OK;
}

int replace_args_ok(acmi_handle *h, int mode) {
    if (mode < 0 || mode > 2) return ACMI_E_ARG;
    if (!h->aut.has_replacement()) {  // multifast_replace returns -2 and emits nothing (replace.c:482-483)
        h->last_error = "no pattern has a replacement text";
        return ACMI_E_NO_REPLACEMENT;
    }
    return ACMI_OK;
}

}  // namespace

int acmi_replace_batch_device(acmi_handle *h, const void *d_blob, uint64_t blob_bytes, const void *d_offsets, size_t n_haystacks,
                              int mode, const void **d_out_blob, const void **d_out_offsets, uint64_t *out_bytes) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !d_offsets) return ACMI_E_ARG;
        rc = replace_args_ok(h, mode);
        if (rc < 0) return rc;
        uint64_t total = 0;
        rc = replace_on_device(h, static_cast<const uint8_t *>(d_blob), blob_bytes, static_cast<const uint64_t *>(d_offsets), n_haystacks, mode, &total);
        if (rc < 0) return rc;
        if (d_out_blob) *d_out_blob = h->d_rep_out.p;
        if (d_out_offsets) *d_out_offsets = h->d_rep_out_off.p;
        if (out_bytes) *out_bytes = total;
        return ACMI_OK;
    });
}

int acmi_replace_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t n_haystacks, int mode,
                       const uint8_t **out_blob, const uint64_t **out_offsets) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !offsets) return ACMI_E_ARG;
        rc = replace_args_ok(h, mode);
        if (rc < 0) return rc;
        uint64_t blob_bytes = 0;
        for (size_t k = 0; k < n_haystacks; ++k) {
            if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] >= 0xFFFFFFFFull) return ACMI_E_ARG;
        }
        if (n_haystacks) blob_bytes = offsets[n_haystacks];
        if (blob_bytes && !blob) return ACMI_E_ARG;
        hipStream_t st = h->stream;
        HIPCHK(h, h->d_blob.ensure(blob_bytes + 64));
        HIPCHK(h, h->d_off.ensure((n_haystacks + 1) * 8));
        if (blob_bytes) HIPCHK(h, hipMemcpyAsync(h->d_blob.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
        if (n_haystacks) HIPCHK(h, hipMemcpyAsync(h->d_off.p, offsets, (n_haystacks + 1) * 8, hipMemcpyHostToDevice, st));
        uint64_t total = 0;
        rc = replace_on_device(h, h->d_blob.as<uint8_t>(), blob_bytes, h->d_off.as<uint64_t>(), n_haystacks, mode, &total);
        if (rc < 0) return rc;
        h->h_rep_off.assign(n_haystacks + 1, 0);
        h->h_rep_blob.resize(total);
        HIPCHK(h, hipMemcpyAsync(h->h_rep_off.data(), h->d_rep_out_off.p, (n_haystacks + 1) * 8, hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(h, hipMemcpyAsync(h->h_rep_blob.data(), h->d_rep_out.p, total, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (out_blob) *out_blob = h->h_rep_blob.data();
        if (out_offsets) *out_offsets = h->h_rep_off.data();
        return ACMI_OK;
    });
}

}  // extern "C"
