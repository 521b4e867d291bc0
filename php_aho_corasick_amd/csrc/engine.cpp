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
#include <cstdio>
#include <cstdlib>
#include <chrono>
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

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;  // grow-only workspace with a little slack
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
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

    // options
    uint32_t slab_events = 8;
    uint32_t run_cap = 8;
    uint64_t chunk_opt = 0;
    int kernel = 3;  // 0 plain global table, 1 LDS-tile walk, 2 anchor filter (if eligible), 3 automatic
    // automatic mode: a batch on which the fast kernel lost (most bytes had to be walked exactly anyway) moves the
    // handle to the slower, safer kernel -- for `*_left` batches, then the fast one is probed again; every further
    // loss doubles the span (16 .. 1024 batches), a win resets it.  Batches under kAutoMinBytes decide nothing.
    bool anchor_off = false;  // the anchor filter lost on this handle's data: table walk meanwhile
    bool exact_only = false;  // k_scan_lds' optimistic pass flagged most of a batch: exact whole-item walk meanwhile
    uint32_t anchor_left = 0, anchor_span = 16, exact_left = 0, exact_span = 16;
    int ilp = 1;
    int group_quads = 8;
    int small_path = 1;   // 1: host batches of <= 64 KiB / <= 1024 haystacks take the one-launch kernel
    uint8_t *pin_small = nullptr;  // pinned host memory of the small path: [summary | prefix | records | offsets | bytes]
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
    DevBuf d_table, d_lds, d_class, d_own, d_dict, d_cnt, d_plen, d_anchor, d_bitmap, d_out_off, d_out_list, d_cold_rec, d_cold_edges, d_chain_byte, d_chain_run, d_slot_pair, d_pairs, d_pat_bytes;

    // workspace (grow-only)
    DevBuf d_blob, d_off, d_item_base, d_item_hay, d_item_rec, d_zero, d_rec_base, d_slab, d_ovf, d_counters,
        d_runs, d_run_count;
    uint64_t *pinned = nullptr;  // host copy of the device counters blocks (2 x 64 B), written by k_expand through pinned_dev
    uint32_t *pinned_dev = nullptr;

    // last result
    uint64_t last_total = 0;
    uint32_t last_nh = 0;
    bool have_result = false;
    std::vector<acmi_record> h_records;
    std::vector<uint64_t> h_prefix;
    // search and replace
    DevBuf d_rep_off, d_rep_bytes, d_plan, d_nom_count, d_out_len, d_rep_out_off, d_rep_out;
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

    // statistics of the last call
    double scan_ms = 0, walk_ms = 0, total_ms = 0;
    int kernel_used = 0;
    uint64_t items = 0, chunk_used = 0;
    uint32_t overflow_items = 0;
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

// Failure links and per-state outputs on the device, one launch per trie depth (kernels.hip k_build_level).  Used for
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
    int dev = h->device_request;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= count || hipSetDevice(dev) != hipSuccess) return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return false;
    lap("runtime + device");
    const size_t n = tv.n_nodes;
    DevBuf d_parent, d_byte, d_own, d_order, d_keys, d_vals, d_fail, d_cnt, d_dict, d_final;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) {
        return b.ensure(bytes) == hipSuccess && copy_h2d(b.p, src, bytes) == hipSuccess;
    };
    bool ok = up(d_parent, tv.parent, n * 4) && up(d_byte, tv.in_byte, n) && up(d_own, tv.own, n * 4) && up(d_order, tv.order, n * 4) &&
              up(d_keys, tv.map_keys, tv.map_size * 8) && up(d_vals, tv.map_vals, tv.map_size * 4) &&
              d_fail.ensure(n * 4) == hipSuccess && d_cnt.ensure(n * 4) == hipSuccess && d_dict.ensure(n * 4) == hipSuccess &&
              d_final.ensure(n) == hipSuccess;
    if (ok) {
        // the root: no failure link, no output (the level loop starts at depth 1 and reads these)
        ok = hipMemset(d_fail.p, 0, 4) == hipSuccess && hipMemset(d_cnt.p, 0, 4) == hipSuccess &&
             hipMemset(d_dict.p, 0xFF, 4) == hipSuccess && hipMemset(d_final.p, 0, 1) == hipSuccess;
    }
    lap("trie upload (+ first use of the device)");
    if (ok) {
        BuildArgs T{d_parent.as<uint32_t>(), d_byte.as<uint8_t>(), d_own.as<uint32_t>(), d_order.as<uint32_t>(), d_keys.as<uint64_t>(),
                    d_vals.as<uint32_t>(), tv.map_size, d_fail.as<uint32_t>(), d_cnt.as<uint32_t>(), d_dict.as<uint32_t>(), d_final.as<uint8_t>()};
        for (uint32_t d = 1; d < tv.n_levels; ++d) launch_build_level(T, tv.level_off[d], tv.level_off[d + 1], nullptr);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        lap("level kernels");
        ok = ok && hipMemcpy(fail, d_fail.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(cnt, d_cnt.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(dict, d_dict.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(is_final, d_final.p, n, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    lap("download");
    for (DevBuf *b : {&d_parent, &d_byte, &d_own, &d_order, &d_keys, &d_vals, &d_fail, &d_cnt, &d_dict, &d_final}) b->release();
    lap("release");
    return ok;
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
    if (fa.est_flag_rate * 8192.0 > double(h->run_cap)) {
        h->exact_only = true;
        h->exact_left = h->exact_span;
    }
    return 1;
}

int ensure_device(acmi_handle *h) {
    if (h->dev_ready) {
        HIPCHK(h, hipSetDevice(h->dev_id));
        return ACMI_OK;
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
    if (!h->stream) h->stream = h->own_stream;
    for (auto &evt : h->ev)
        if (!evt) HIPCHK(h, hipEventCreate(&evt));
    for (auto &slot : h->fly)
        for (auto &evt : slot.ev)
            if (!evt) HIPCHK(h, hipEventCreate(&evt));
    if (!h->pinned) {
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&h->pinned), 128, hipHostMallocMapped));
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->pinned, 0));
        h->pinned_dev = static_cast<uint32_t *>(dp);
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

// The batch pipeline on device-resident input, first half: enqueue everything, no host wait.
int enqueue_batch(acmi_handle *h, acmi_handle::FlySlot &F, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off,
                  size_t nh, int find_all);

int run_batch_begin(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh,
                    int find_all) {
    if (h->begin_seq - h->end_seq >= 2) {
        h->last_error = "two batches are in flight already: end the older one first";
        return ACMI_E_ARG;
    }
    if (nh > 0xFFFFFFF0ull) { h->last_error = "too many haystacks"; return ACMI_E_TOO_LARGE; }
    // This batch takes the slot of the batch before the previous one, whose prefix array and records end here: a
    // result is valid from its batch's end until the SECOND begin after its own.
    acmi_handle::FlySlot &F = h->fly[(h->begin_seq + 1) & 1];
    h->have_result = false;
    const int rc = enqueue_batch(h, F, d_blob, blob_bytes, d_off, nh, find_all);
    // a pipeline that was not enqueued to its end has not zeroed the next batch's sync block: have both zeroed again
    if (rc < 0) { h->sync_tiles = 0; return rc; }
    F.seq = ++h->begin_seq;
    h->enqueue_seq = F.seq;
    return rc;
}

int enqueue_batch(acmi_handle *h, acmi_handle::FlySlot &F, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off,
                  size_t nh, int find_all) {
    hipStream_t st = h->stream;
    HIPCHK(h, F.d_prefix.ensure((nh + 1) * 8));
    F.d_blob = d_blob;
    F.d_off = d_off;
    F.nh = nh;
    F.blob_bytes = blob_bytes;
    F.find_all = find_all;
    F.kernel_used = 0;
    F.chunk_used = 0;
    F.max_items = 0;
    if (h->profile > 1) HIPCHK(h, hipEventRecord(F.ev[0], st));
    const int slot_index = int(&F - h->fly);
    uint32_t *const pinned_dev = h->pinned_dev + 16 * slot_index;

    if (nh == 0 || h->aut.accepted_count() == 0) {  // empty batch / empty automaton -> no records
        HIPCHK(h, hipMemsetAsync(F.d_prefix.p, 0, (nh + 1) * 8, st));
        HIPCHK(h, hipEventRecord(F.ev[4], st));
        F.trivial = true;
        return ACMI_OK;
    }
    F.trivial = false;

    const FlatAutomaton &fa = h->aut.flat();
    // the anchor filter needs a pattern set that allows it; automatic mode also leaves it once a batch
    // showed that this handle's text defeats it (see run_batch_end)
    if (h->kernel == 3 && blob_bytes >= kAutoMinBytes) {  // time to probe the fast kernel again?
        if (h->anchor_off && h->anchor_left-- == 0) h->anchor_off = false;
        else if (h->exact_only && !(fa.anchor_q && !h->anchor_off) && h->exact_left-- == 0) h->exact_only = false;
    }
    int kernel = h->kernel == 3 ? ((fa.anchor_q && !h->anchor_off) ? 2 : 1)
                                : ((h->kernel == 2 && !fa.anchor_q) ? 1 : h->kernel);
    // Small haystacks: the anchor kernel gives a wave one work item per step (its short-item variant skips the 1-KiB rows
    // an item lacks; its tiny-item variant, from 1 KiB per haystack down, packs one-row items eight to a step), so its
    // rate falls with the item size -- headline patterns over 4 GiB: 5.4 TB/s at 8 KiB, 4.8 at 4 KiB, 3.5 at 2 KiB, 3.95 at
    // 1 KiB, 2.6 at 512 B, 1.44 at 256 B, 0.75 at 128 B -- while the LDS-tile walk gives every LANE its own items (2.3 /
    // 1.75 / 1.27 / 0.84 TB/s at 1 KiB / 512 / 256 / 128 B).  Under 192 B per haystack on average automatic mode takes
    // the tile walk -- unless the automaton's estimate or an earlier batch says its optimistic pass does not work on this
    // handle's data (exact_only).  (Batches under 1 MiB are launch-bound either way and keep the kernel the pattern set
    // asks for.)
    if (h->kernel == 3 && kernel == 2 && !h->exact_only && blob_bytes >= kAutoMinBytes && blob_bytes / nh < kSmallItemBytes) kernel = 1;
    if (h->kernel == 3 && kernel == 1 && h->exact_only) kernel = 4;
    F.kernel_used = kernel;
    const uint32_t chunk = choose_chunk(h, blob_bytes, nh, kernel);
    const uint64_t max_items = blob_bytes / chunk + nh;
    const uint32_t K = std::max<uint32_t>(1, h->slab_events);
    F.chunk_used = chunk;

    HIPCHK(h, h->d_item_base.ensure((nh + 1) * 8));
    HIPCHK(h, h->d_item_hay.ensure(max_items * 4));
    HIPCHK(h, h->d_item_rec.ensure(max_items * 16));
    // per-item counters [rec_count (items+1) x 8 | ev_count items x 4]: zeroed per item by k_items
    const size_t rec_cnt_off = 0, ev_off = rec_cnt_off + (max_items + 1) * 8;
    HIPCHK(h, h->d_zero.ensure(ev_off + max_items * 4));
    // sync block of this batch: [counters 16 x u32 | tickets 2 x u32 | pad | status words of the two scans], one of
    // two copies that take turns; this batch's k_expand zeroes the other one.  (Re)allocation zeroes both.
    const uint32_t tiles = std::max(scan_tiles(nh + 1), scan_tiles(max_items + 1)) + 1u;
    if (tiles > h->sync_tiles || !h->d_sync[0].p) {
        const uint32_t want = tiles + tiles / 4 + 16;
        for (DevBuf &b : h->d_sync) {
            HIPCHK(h, b.ensure(128 + size_t(want) * 16));
            HIPCHK(h, hipMemsetAsync(b.p, 0, b.cap, st));
        }
        h->sync_tiles = want;
    }
    const uint32_t sync_words = uint32_t(32 + size_t(h->sync_tiles) * 4);
    uint32_t *sync_cur = h->d_sync[h->sync_turn & 1].as<uint32_t>();
    uint32_t *sync_next = h->d_sync[(h->sync_turn + 1) & 1].as<uint32_t>();
    ++h->sync_turn;
    uint64_t *status0 = reinterpret_cast<uint64_t *>(sync_cur + 32), *status1 = status0 + h->sync_tiles;
    HIPCHK(h, h->d_rec_base.ensure((max_items + 2) * 8));
    HIPCHK(h, h->d_slab.ensure(max_items * K * 8));
    HIPCHK(h, h->d_ovf.ensure(max_items * 4));
    const uint32_t Q = std::max<uint32_t>(1, h->run_cap);
    HIPCHK(h, h->d_runs.ensure(max_items * Q * 16));
    HIPCHK(h, h->d_run_count.ensure(max_items * 4));
    uint8_t *zero = h->d_zero.as<uint8_t>();
    uint32_t *d_counters = sync_cur;
    uint64_t *d_rec_cnt = reinterpret_cast<uint64_t *>(zero + rec_cnt_off);
    uint32_t *d_ev = reinterpret_cast<uint32_t *>(zero + ev_off);

    BatchArgs B{};
    B.blob = d_blob;
    B.offsets = d_off;
    B.item_base = h->d_item_base.as<uint64_t>();
    B.n_haystacks = uint32_t(nh);
    B.chunk_bytes = chunk;
    B.warm_bytes = fa.max_pattern_length ? fa.max_pattern_length - 1 : 0;
    B.align_warm = kernel == 2 ? 1u : (kernel == 1 && h->dA.other_byte < 0x100u ? 2u : 0u);
    // anchor kernel: the variant that skips the 1-KiB rows a task lacks costs the perfectly regular batch -- haystacks of
    // exactly k x 8 KiB on an aligned blob, every task full: the bench's shape -- 4 % (0.704 vs 0.677 ms) and gains up to 2 x
    // on everything else (short items, last chunks, the 513th quad of an 8-KiB haystack that starts unaligned)
    const bool regular = blob_bytes % nh == 0 && (blob_bytes / nh) % 8192 == 0 && (reinterpret_cast<uintptr_t>(d_blob) & 15u) == 0;
    B.short_items = kernel != 2 ? 0u : h->short_rows >= 0 ? uint32_t(h->short_rows) : (regular ? 0u : 1u);
    B.tiny_items = (kernel == 2 && h->tiny_rows != 0 && blob_bytes / nh <= 1024) ? 1u : 0u;
    B.slab_events = K;
    B.first_only = find_all ? 0u : 1u;
    B.item_hay = h->d_item_hay.as<uint32_t>();
    B.item_rec = h->d_item_rec.as<uint4>();
    B.ev_count = d_ev;
    B.rec_count = d_rec_cnt;
    B.slab = h->d_slab.as<uint2>();
    B.overflow_list = h->d_ovf.as<uint32_t>();
    B.counters = d_counters;
    B.runs = h->d_runs.as<uint4>();
    B.run_count = h->d_run_count.as<uint32_t>();
    B.run_cap = Q;
    B.run_stride = max_items;
    B.poison_quads = std::max<uint32_t>(1, (fa.max_pattern_length + 15) / 16);
    // persistent workgroups pay off once a CU would otherwise run several workgroups back to back
    // (measured: 1 KiB items -18 %, 2 x 8 KiB per lane +2.5 %)
    {   // anchor kernel: items per wave batch -- 64 when the batch can give every wave that many
        const uint64_t est = std::max<uint64_t>(1, blob_bytes / chunk);
        const uint64_t waves = 256ull * 16;
        B.wave_items = uint32_t(std::min<uint64_t>(64, std::max<uint64_t>(1, est / waves)));
        if (h->wave_items > 0) B.wave_items = uint32_t(h->wave_items);
    }
    B.persistent = h->persistent >= 0 ? uint32_t(h->persistent != 0)
                                      : uint32_t(blob_bytes / chunk > 3ull * 256 * 1024 * uint64_t(h->ilp));
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

    // item counts, their scan, the item table, zeroed per-item counters: one launch
    launch_items(B, blob_bytes, max_items, h->d_item_base.as<uint64_t>(), sync_cur + 16, status0, st);

    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[1], st));
    if (kernel == 4) {
        // exact walk only (k_walk_items): for text on which k_scan_lds' optimistic
        // pass flags everything anyway (events in every wave-quad)
        HIPCHK(h, launch_walk_items(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), h->walk_group, st));
    } else {
        HIPCHK(h, launch_scan(h->dA, B, max_items, int(fa.entry_bytes), kernel, int(fa.class_mode), h->ilp, h->group_quads, st));
    }
    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[2], st));
    // the anchor kernel only flags byte runs; their exact walk is a kernel of its own (one thread per item)
    if (kernel == 2) HIPCHK(h, launch_walk_runs(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), st));
    if (h->profile > 1) HIPCHK(h, hipEventRecord(F.ev[3], st));

    if (!find_all) launch_first_only(B, st);
    launch_rec_scan(B, max_items, h->d_rec_base.as<uint64_t>(), sync_cur + 17, status1, st);

    // Expansion is enqueued optimistically with the output capacity of the previous call, so the
    // whole pipeline needs ONE host synchronisation (the read-back of the summary at the end).  Only
    // when the batch produced more records than fit is the buffer grown and the expansion re-run.
    HIPCHK(h, F.d_out.ensure(std::max<size_t>(16, h->out_want)));  // (at least what any earlier batch of this handle needed)
    const uint64_t cap = F.d_out.cap / 16;
    // (also: per-haystack prefix, the summary for the host next to the overflow counter -- counters[2..3] = total
    // records, [4..5] = items -- and the zeroing of the next batch's sync block)
    launch_expand(h->dA, B, max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, F.d_prefix.as<uint64_t>(),
                  pinned_dev, sync_next, sync_words, st);
    if (find_all) launch_refill(h->dA, B, max_items, int(fa.entry_bytes), h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, h->refill, st);
    F.sync_next = sync_next;
    F.sync_words = sync_words;
    HIPCHK(h, hipEventRecord(F.ev[4], st));  // (always: the end waits for THIS batch, not for the one queued behind it)
    HIPCHK(h, hipGetLastError());
    F.args = B;
    F.max_items = max_items;
    return ACMI_OK;
}

// Second half, for the OLDEST batch in flight: the one host synchronisation, the capacity check and (rarely) the re-run.
int run_batch_end(acmi_handle *h) {
    if (h->begin_seq == h->end_seq) { h->last_error = "no batch in flight"; return ACMI_E_ARG; }
    acmi_handle::FlySlot &F = h->fly[(h->end_seq + 1) & 1];
    ++h->end_seq;
    const int slot_index = int(&F - h->fly);
    const uint64_t *const pinned = h->pinned + 8 * slot_index;
    const uint32_t *const pinned32 = reinterpret_cast<const uint32_t *>(pinned);
    hipStream_t st = h->stream;
    {
        // (with a second batch queued behind this one the wait is for this batch's last kernel only)
        const hipError_t e = h->begin_seq != h->end_seq ? hipEventSynchronize(F.ev[4]) : hipStreamSynchronize(st);
        if (e != hipSuccess) { h->sync_tiles = 0; h->end_seq = h->begin_seq; return fail_hip(h, e, "hipStreamSynchronize"); }
    }
    h->cur = slot_index;
    h->last_nh = uint32_t(F.nh);
    h->last_total = 0;
    h->scan_ms = h->walk_ms = h->total_ms = 0;
    h->items = 0;
    h->overflow_items = 0;
    h->kernel_used = F.kernel_used;
    h->chunk_used = F.chunk_used;
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
    h->walked_bytes = 0;  // counters[7..15]: k_scan_lds adds to [7], k_walk_runs spreads its workgroups over [8..15]
    for (int i = 7; i < 16; ++i) h->walked_bytes += uint64_t(pinned32[i]) * 16;
    h->fallback_bytes = F.kernel_used == 2 ? uint64_t(pinned32[7]) * 16 : 0;
    if (F.kernel_used == 2) h->walked_bytes -= h->fallback_bytes;  // (counters[7] is k_scan_lds' own share otherwise)
    // automatic mode: a batch of which more than a quarter had to be walked exactly defeats the filter
    const bool decisive = h->kernel == 3 && F.blob_bytes >= kAutoMinBytes;
    if (decisive && F.kernel_used == 2) {
        if (h->walked_bytes * 4 > F.blob_bytes) {
            h->anchor_off = true;
            h->anchor_left = h->anchor_span;
            h->anchor_span = std::min<uint32_t>(h->anchor_span * 2, 1024);
        } else {
            h->anchor_span = 16;
        }
    }
    // ... and one of which more than half was flagged by the LDS-tile kernel's optimistic pass (events or tile
    // exits in most wave-quads: keyword-like and dense-match text) is walked exactly right away
    if (decisive && F.kernel_used == 1) {
        if (h->walked_bytes * 2 > F.blob_bytes) {
            h->exact_only = true;
            h->exact_left = h->exact_span;
            h->exact_span = std::min<uint32_t>(h->exact_span * 2, 1024);
        } else {
            h->exact_span = 16;
        }
    }
    if (h->profile) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, F.ev[1], F.ev[2]));
        h->scan_ms = ms;
        if (h->profile > 1) {
            HIPCHK(h, hipEventElapsedTime(&ms, F.ev[2], F.ev[3]));
            h->walk_ms = F.kernel_used == 2 ? ms : 0.0;
            HIPCHK(h, hipEventElapsedTime(&ms, F.ev[0], F.ev[4]));
            h->total_ms = ms;
        }
    }
    h->out_want = std::max<size_t>(h->out_want, size_t(total) * 16);
    if (total > F.d_out.cap / 16) {  // did not fit the optimistic capacity: grow, expand again
        if (h->enqueue_seq == F.seq) {
            HIPCHK(h, F.d_out.ensure(total * 16));
            const uint64_t cap = F.d_out.cap / 16;
            launch_expand(h->dA, F.args, F.max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, F.d_prefix.as<uint64_t>(),
                          h->pinned_dev + 16 * slot_index, F.sync_next, F.sync_words, st);
            if (F.find_all)
                launch_refill(h->dA, F.args, F.max_items, int(fa.entry_bytes), h->d_rec_base.as<uint64_t>(),
                              F.d_out.p, cap, h->refill, st);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipStreamSynchronize(st));
        } else {
            // a later begin has reused the item table, slabs and record bases: this batch runs again as a whole, into a
            // buffer of the size it is now known to need (the later batch's own result and summary are its slot's)
            {
                const hipError_t e = hipStreamSynchronize(st);
                if (e != hipSuccess) { h->sync_tiles = 0; return fail_hip(h, e, "hipStreamSynchronize"); }
            }
            HIPCHK(h, F.d_out.ensure(total * 16));
            const int rc = enqueue_batch(h, F, F.d_blob, F.blob_bytes, F.d_off, F.nh, F.find_all);
            if (rc < 0) { h->sync_tiles = 0; return rc; }
            h->enqueue_seq = F.seq;
            HIPCHK(h, hipStreamSynchronize(st));
            total = pinned[1];
            if (total > F.d_out.cap / 16) { h->last_error = "record count changed between two runs of one batch"; return ACMI_E_DEVICE; }
        }
    }
    h->last_total = total;
    h->have_result = true;
    return ACMI_OK;
}

int run_batch(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh,
              int find_all) {
    if (h->begin_seq != h->end_seq) {  // a synchronous call drops batches that were begun and never ended
        (void)hipStreamSynchronize(h->stream);
        h->end_seq = h->begin_seq;
    }
    int rc = run_batch_begin(h, d_blob, blob_bytes, d_off, nh, find_all);
    if (rc < 0) return rc;
    return run_batch_end(h);
}

// Small host batches: one launch that reads and writes pinned host memory (k_match_small).  Returns 1 when
// the batch does not qualify or its records do not fit, and the caller takes the general path.
constexpr uint32_t kSmallCap = 8192;  // records
constexpr size_t kSmallSummaryOff = 0, kSmallPrefixOff = 64, kSmallRecordsOff = kSmallPrefixOff + (kSmallPathMaxHay + 8) * 8,
                 kSmallOffsetsOff = kSmallRecordsOff + size_t(kSmallCap) * 16, kSmallBytesOff = kSmallOffsetsOff + (kSmallPathMaxHay + 8) * 8,
                 kSmallPinBytes = kSmallBytesOff + kSmallPathMaxBytes + 64;

int run_small(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t nh, int find_all) {
    if (!h->small_path || nh == 0 || nh > kSmallPathMaxHay) return 1;
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
                                 d + kSmallRecordsOff, kSmallCap, st));
    if (h->profile) HIPCHK(h, hipEventRecord(h->ev[4], st));
    HIPCHK(h, hipStreamSynchronize(st));
    const uint64_t *summary = reinterpret_cast<const uint64_t *>(pin + kSmallSummaryOff);
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
        if (h->dev_touched) {  // also after a failed upload: release whatever was created
            (void)hipSetDevice(h->dev_id);
            if (h->stream) (void)hipStreamSynchronize(h->stream);
            for (DevBuf *b : {&h->d_table, &h->d_lds, &h->d_class, &h->d_own, &h->d_dict, &h->d_cnt, &h->d_plen, &h->d_anchor, &h->d_bitmap, &h->d_out_off, &h->d_out_list, &h->d_cold_rec, &h->d_cold_edges, &h->d_chain_byte, &h->d_chain_run, &h->d_slot_pair, &h->d_pairs, &h->d_pat_bytes, &h->d_rep_off, &h->d_rep_bytes, &h->d_plan, &h->d_nom_count, &h->d_out_len, &h->d_rep_out_off, &h->d_rep_out, &h->d_blob,
                              &h->d_off, &h->d_sync[0], &h->d_sync[1], &h->d_item_base, &h->d_item_hay, &h->d_item_rec, &h->d_zero,
                              &h->d_rec_base, &h->d_slab, &h->d_ovf, &h->d_counters, &h->fly[0].d_prefix, &h->fly[0].d_out, &h->fly[1].d_prefix, &h->fly[1].d_out, &h->d_runs, &h->d_run_count})
                b->release();
            for (auto &evt : h->ev)
                if (evt) (void)hipEventDestroy(evt);
            for (auto &slot : h->fly)
                for (auto &evt : slot.ev)
                    if (evt) (void)hipEventDestroy(evt);
            if (h->pinned) (void)hipHostFree(h->pinned);
            if (h->pin_small) (void)hipHostFree(h->pin_small);
            if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
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
        if (rc < 0) return rc;
        if (n_records) *n_records = h->last_total;
        return ACMI_OK;
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

int acmi_match_batch_device_end(acmi_handle *h, uint64_t *n_records) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        HIPCHK(h, hipSetDevice(h->dev_id));
        int rc = run_batch_end(h);
        if (rc < 0) return rc;
        if (n_records) *n_records = h->last_total;
        return ACMI_OK;
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
        if (rc < 0) return rc;
        // drain: device match buffer -> host records
        h->h_prefix.assign(n_haystacks + 1, 0);
        h->h_records.resize(h->last_total);
        rc = acmi_result_to_host(h, h->h_prefix.data(), h->h_records.data());
        if (rc < 0) return rc;
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

int acmi_set_option(acmi_handle *h, const char *key, int64_t value) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!key) return ACMI_E_ARG;
        const std::string k(key);
        if (k == "lds_bytes") {
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 2048 || value > 160 * 1024 - 64) return ACMI_E_ARG;
            h->bopt.lds_bytes = uint32_t(value);
        } else if (k == "slab_events") {
            if (value < 1 || value > 4096) return ACMI_E_ARG;
            h->slab_events = uint32_t(value);
        } else if (k == "run_cap") {
            if (value < 1 || value > 1024) return ACMI_E_ARG;
            h->run_cap = uint32_t(value);
        } else if (k == "chunk_bytes") {
            if (value < 0) return ACMI_E_ARG;
            h->chunk_opt = uint64_t(value);
        } else if (k == "kernel") {
            if (value < 0 || value > 4) return ACMI_E_ARG;
            h->kernel = int(value);
            h->anchor_off = false;
            h->exact_only = false;
            h->anchor_span = h->exact_span = 16;
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
    });
}

int acmi_get_stat(const acmi_handle *h, const char *key, double *value) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (!key || !value) return ACMI_E_ARG;
        const std::string k(key);
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
        else if (k == "scan_ms") *value = h->scan_ms;
        else if (k == "walk_ms") *value = h->walk_ms;
        else if (k == "kernel_used") *value = h->kernel_used;
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
        h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
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
namespace {

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, acmi_comm_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
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
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.error = "librccl.so lacks a needed symbol";
    });
    return r;
}

constexpr int kNcclUint8 = 1;  // ncclDataType_t (rccl.h:460)

}  // namespace

struct acmi_comm {
    void *comm = nullptr;  // ncclComm_t; null for a world of one
    int rank = 0, world = 1, dev = 0;
    DevBuf d_hdr, d_send, d_recv, d_prefix, d_records;
    uint64_t *h_hdr = nullptr;  // pinned: world x {records, haystacks}
    uint64_t total_rec = 0, total_hay = 0;
    std::string last_error;
};

namespace {

// Unpack `world` payloads (see GatherArgs) into the comm's global prefix / record arrays.
int comm_compact(acmi_comm *c, int world, const uint8_t *d_recv, uint64_t stride, uint64_t nh_cap, const uint64_t *hdr, hipStream_t st) {
    GatherArgs G{};
    G.recv = d_recv;
    G.stride = stride;
    G.nh_cap = nh_cap;
    G.world = uint32_t(world);
    for (int r = 0; r < world; ++r) {
        G.rec_base[r + 1] = G.rec_base[r] + hdr[2 * r];
        G.hay_base[r + 1] = G.hay_base[r] + hdr[2 * r + 1];
    }
    G.total_rec = G.rec_base[world];
    G.total_hay = G.hay_base[world];
    if (c->d_prefix.ensure((G.total_hay + 1) * 8) != hipSuccess || c->d_records.ensure(std::max<uint64_t>(16, G.total_rec * 16)) != hipSuccess)
        return ACMI_E_NOMEM;
    G.out_prefix = c->d_prefix.as<uint64_t>();
    G.out_records = c->d_records.as<uint4>();
    launch_gather_compact(G, st);
    c->total_rec = G.total_rec;
    c->total_hay = G.total_hay;
    return hipGetLastError() == hipSuccess ? ACMI_OK : ACMI_E_DEVICE;
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
        acmi_comm *c = new acmi_comm();
        c->rank = rank;
        c->world = world;
        c->dev = h->dev_id;
        if (world > 1) {
            Rccl &R = rccl();
            if (!R.error.empty()) { h->last_error = R.error; delete c; return ACMI_E_NO_DEVICE; }
            const int e = R.CommInitRank(&c->comm, world, *id, rank);
            if (e != 0) {
                h->last_error = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(e) : "error");
                delete c;
                return ACMI_E_DEVICE;
            }
        }
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_hdr), size_t(world) * 16, hipHostMallocDefault) != hipSuccess) {
            if (c->comm) rccl().CommDestroy(c->comm);
            delete c;
            return ACMI_E_NOMEM;
        }
        *out = c;
        return ACMI_OK;
    });
}

int acmi_comm_destroy(acmi_comm *c) {
    return guarded(nullptr, [&]() -> int {
        if (!c) return ACMI_E_INVALID;
        (void)hipSetDevice(c->dev);
        if (c->comm) rccl().CommDestroy(c->comm);
        for (DevBuf *b : {&c->d_hdr, &c->d_send, &c->d_recv, &c->d_prefix, &c->d_records}) b->release();
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        delete c;
        return ACMI_OK;
    });
}

int acmi_comm_allgather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, const void **d_prefix,
                               const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    return guarded(h, [&]() -> int {
        if (!c) return ACMI_E_ARG;
        if (!live(h)) return ACMI_E_INVALID;
        if (!h->have_result || n_haystacks_local != h->last_nh) return ACMI_E_ARG;
        HIPCHK(h, hipSetDevice(h->dev_id));
        hipStream_t st = h->stream;
        Rccl &R = rccl();
        const int W = c->world;
        // round 1: sizes (16 bytes per rank) -- RCCL has no all-gatherv, the payload round is padded to the largest shard
        HIPCHK(h, c->d_hdr.ensure(size_t(W + 1) * 16));
        uint64_t *d_mine = c->d_hdr.as<uint64_t>(), *d_all = d_mine + 2;
        const uint64_t mine[2] = {h->last_total, h->last_nh};
        HIPCHK(h, hipMemcpyAsync(d_mine, mine, 16, hipMemcpyHostToDevice, st));
        if (W > 1) {
            const int e = R.AllGather(d_mine, d_all, 16, kNcclUint8, c->comm, st);
            if (e != 0) { h->last_error = std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
        } else {
            HIPCHK(h, hipMemcpyAsync(d_all, d_mine, 16, hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(h, hipMemcpyAsync(c->h_hdr, d_all, size_t(W) * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        uint64_t rec_cap = 0, nh_cap = 0;
        for (int r = 0; r < W; ++r) { rec_cap = std::max(rec_cap, c->h_hdr[2 * r]); nh_cap = std::max(nh_cap, c->h_hdr[2 * r + 1]); }
        // round 2: [prefix (nh_cap+1) | records rec_cap], every rank the same size
        const uint64_t stride = (nh_cap + 1) * 8 + rec_cap * 16;
        HIPCHK(h, c->d_send.ensure(stride));
        HIPCHK(h, c->d_recv.ensure(stride * uint64_t(W)));
        uint8_t *send = c->d_send.as<uint8_t>();
        HIPCHK(h, hipMemcpyAsync(send, h->d_prefix().p, (size_t(h->last_nh) + 1) * 8, hipMemcpyDeviceToDevice, st));
        if (h->last_total)
            HIPCHK(h, hipMemcpyAsync(send + (nh_cap + 1) * 8, h->d_out().p, h->last_total * 16, hipMemcpyDeviceToDevice, st));
        const uint8_t *recv = send;
        if (W > 1) {
            const int e = R.AllGather(send, c->d_recv.p, stride, kNcclUint8, c->comm, st);
            if (e != 0) { h->last_error = std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
            recv = c->d_recv.as<uint8_t>();
        }
        int rc = comm_compact(c, W, recv, stride, nh_cap, c->h_hdr, st);
        if (rc < 0) return rc;
        HIPCHK(h, hipStreamSynchronize(st));
        if (d_prefix) *d_prefix = c->d_prefix.p;
        if (d_records) *d_records = c->d_records.p;
        if (n_records) *n_records = c->total_rec;
        if (n_haystacks) *n_haystacks = c->total_hay;
        return ACMI_OK;
    });
}

int acmi_comm_debug_compact(acmi_comm *c, acmi_handle *h, int n_payloads, const void *d_payloads, uint64_t stride, uint64_t nh_cap,
                            const uint64_t *sizes, const void **d_prefix, const void **d_records, uint64_t *n_records,
                            uint64_t *n_haystacks) {
    return guarded(h, [&]() -> int {
        if (!c || !sizes || !d_payloads || n_payloads < 1 || n_payloads > int(kMaxRanks)) return ACMI_E_ARG;
        int rc = check_ready(h);
        if (rc < 0) return rc;
        rc = comm_compact(c, n_payloads, static_cast<const uint8_t *>(d_payloads), stride, nh_cap, sizes, h->stream);
        if (rc < 0) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (d_prefix) *d_prefix = c->d_prefix.p;
        if (d_records) *d_records = c->d_records.p;
        if (n_records) *n_records = c->total_rec;
        if (n_haystacks) *n_haystacks = c->total_hay;
        return ACMI_OK;
    });
}

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

int acmi_replace_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t n_haystacks, int mode,
                       const uint8_t **out_blob, const uint64_t **out_offsets) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (mode < 0 || mode > 2 || (n_haystacks && !offsets)) return ACMI_E_ARG;
        if (!h->aut.has_replacement()) {  // multifast_replace returns -2 and emits nothing (replace.c:482-483)
            h->last_error = "no pattern has a replacement text";
            return ACMI_E_NO_REPLACEMENT;
        }
        uint64_t blob_bytes = 0;
        for (size_t k = 0; k < n_haystacks; ++k) {
            if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] >= 0xFFFFFFFFull) return ACMI_E_ARG;
        }
        if (n_haystacks) blob_bytes = offsets[n_haystacks];
        if (blob_bytes && !blob) return ACMI_E_ARG;
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
        // 1. every match of the batch (the pipeline of acmi_match_batch, results left on the device)
        HIPCHK(h, h->d_blob.ensure(blob_bytes + 64));
        HIPCHK(h, h->d_off.ensure((n_haystacks + 1) * 8));
        if (blob_bytes) HIPCHK(h, hipMemcpyAsync(h->d_blob.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
        h->h_rep_off.assign(n_haystacks + 1, 0);
        if (n_haystacks == 0) {
            h->h_rep_blob.clear();
            if (out_blob) *out_blob = h->h_rep_blob.data();
            if (out_offsets) *out_offsets = h->h_rep_off.data();
            return ACMI_OK;
        }
        HIPCHK(h, hipMemcpyAsync(h->d_off.p, offsets, (n_haystacks + 1) * 8, hipMemcpyHostToDevice, st));
        rc = run_batch(h, h->d_blob.as<uint8_t>(), blob_bytes, h->d_off.as<uint64_t>(), n_haystacks, 1);
        if (rc < 0) return rc;
        // 2. nominees and output lengths per haystack
        HIPCHK(h, h->d_plan.ensure(std::max<uint64_t>(8, h->last_total * 8)));
        HIPCHK(h, h->d_nom_count.ensure(n_haystacks * 4));
        HIPCHK(h, h->d_out_len.ensure(n_haystacks * 8));
        HIPCHK(h, h->d_rep_out_off.ensure((n_haystacks + 1) * 8));
        ReplaceArgs R{};
        R.blob = h->d_blob.as<uint8_t>();
        R.offsets = h->d_off.as<uint64_t>();
        R.n_haystacks = uint32_t(n_haystacks);
        R.mode = mode == 2 ? 2u : 1u;   // MF_REPLACE_MODE_DEFAULT behaves as NORMAL (replace.c:252-254)
        R.prefix = h->d_prefix().as<uint64_t>();
        R.records = h->d_out().as<uint4>();
        R.pattern_length = h->dA.pattern_length;
        R.rep_off = h->d_rep_off.as<uint64_t>();
        R.rep_bytes = h->d_rep_bytes.as<uint8_t>();
        R.plan = h->d_plan.as<uint2>();
        R.nom_count = h->d_nom_count.as<uint32_t>();
        R.out_len = h->d_out_len.as<uint64_t>();
        launch_replace_plan(R, st);
        // 3. output offsets: exclusive scan of the lengths (nh words: on the host)
        std::vector<uint64_t> lens(n_haystacks);
        HIPCHK(h, hipMemcpyAsync(lens.data(), h->d_out_len.p, n_haystacks * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        for (size_t k = 0; k < n_haystacks; ++k) h->h_rep_off[k + 1] = h->h_rep_off[k] + lens[k];
        const uint64_t total = h->h_rep_off[n_haystacks];
        HIPCHK(h, hipMemcpyAsync(h->d_rep_out_off.p, h->h_rep_off.data(), (n_haystacks + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(h, h->d_rep_out.ensure(total + 16));
        // 4. the output texts
        R.out_off = h->d_rep_out_off.as<uint64_t>();
        R.out_blob = h->d_rep_out.as<uint8_t>();
        launch_replace_emit(R, st);
        h->h_rep_blob.resize(total);
        if (total) HIPCHK(h, hipMemcpyAsync(h->h_rep_blob.data(), h->d_rep_out.p, total, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, hipGetLastError());
        if (out_blob) *out_blob = h->h_rep_blob.data();
        if (out_offsets) *out_offsets = h->h_rep_off.data();
        return ACMI_OK;
    });
}

}  // extern "C"
