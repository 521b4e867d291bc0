// engine.cpp -- handle lifecycle, device memory, batch pipeline and the C ABI (include/acmi.h).
//
// Host-side counterpart of the reference's Zend glue around MultiFast
// (src/php_ahocorasick.c): what PHP_FUNCTION(ahocorasick_match) + the match callback do per
// haystack (:664-746, :542-589) is done here per BATCH: stage haystacks, run the HIP scan,
// drain the device match buffer into {hay,pos,start,pattern} records.  There is no CPU
// matching path in this library: without a usable gfx950 device every match call fails with
// ACMI_E_NO_DEVICE / ACMI_E_DEVICE.
#include "engine.hpp"

namespace acmi_engine {

uint64_t g_dev_allocs = 0;
std::mutex g_live_mutex;
std::unordered_set<const acmi_handle *> g_live;

bool live(const acmi_handle *h) {
    if (!h) return false;
    std::lock_guard<std::mutex> g(g_live_mutex);
    return g_live.count(h) != 0;
}

int fail_hip(acmi_handle *h, hipError_t e, const char *what) {
    h->last_error = std::string(what) + ": " + hipGetErrorString(e);
    h->item_key[0].valid = h->item_key[1].valid = false;   // (whatever was enqueued may not have run: no table is taken on trust)
    return (e == hipErrorOutOfMemory) ? ACMI_E_NOMEM : ACMI_E_DEVICE;
}

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

    if (!fa.out_head.empty()) {
        HIPCHK(h, h->d_out_head.ensure(fa.out_head.size() * 4));
        HIPCHK(h, copy_h2d(h->d_out_head.p, fa.out_head.data(), fa.out_head.size() * 4));
    }
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
        static_assert(kWinListBytes == kWinListBytesDev, "the window table's list area");
        if (fa.win_present) {
            HIPCHK(h, h->d_win_table.ensure(fa.win_table.size() * 2));
            HIPCHK(h, copy_h2d(h->d_win_table.p, fa.win_table.data(), fa.win_table.size() * 2));
            HIPCHK(h, h->d_win_state.ensure(fa.win_state.size() * 4));
            HIPCHK(h, copy_h2d(h->d_win_state.p, fa.win_state.data(), fa.win_state.size() * 4));
            HIPCHK(h, h->d_win_hash.ensure(fa.win_hash.size() * 4));
            HIPCHK(h, copy_h2d(h->d_win_hash.p, fa.win_hash.data(), fa.win_hash.size() * 4));
        }
        if (!fa.chain_byte.empty()) {
            HIPCHK(h, h->d_chain_byte.ensure(fa.chain_byte.size()));
            HIPCHK(h, copy_h2d(h->d_chain_byte.p, fa.chain_byte.data(), fa.chain_byte.size()));
            HIPCHK(h, h->d_chain_run.ensure(fa.chain_run.size()));
            HIPCHK(h, copy_h2d(h->d_chain_run.p, fa.chain_run.data(), fa.chain_run.size()));
        }
    }

    if (fa.chain.present()) {
        const ChainImage &ci = fa.chain;
        HIPCHK(h, h->d_chain_img.ensure(ci.image.size() + 16));
        HIPCHK(h, copy_h2d(h->d_chain_img.p, ci.image.data(), ci.image.size()));
        HIPCHK(h, h->d_chain_head.ensure(ci.head.size() * 4));
        HIPCHK(h, copy_h2d(h->d_chain_head.p, ci.head.data(), ci.head.size() * 4));
        HIPCHK(h, h->d_chain_orig.ensure(ci.orig.size() * 4));
        HIPCHK(h, copy_h2d(h->d_chain_orig.p, ci.orig.data(), ci.orig.size() * 4));
        HIPCHK(h, h->d_chain_frank.ensure(ci.frank.size() * 2 + 16));
        HIPCHK(h, copy_h2d(h->d_chain_frank.p, ci.frank.data(), ci.frank.size() * 2));
        HIPCHK(h, h->d_chain_fhead.ensure(ci.fhead.size() * 4 + 16));
        if (!ci.fhead.empty()) HIPCHK(h, copy_h2d(h->d_chain_fhead.p, ci.fhead.data(), ci.fhead.size() * 4));
        h->chain_finals = uint32_t(ci.fhead.size() / 2);
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
    A.out_head = fa.out_head.empty() ? nullptr : h->d_out_head.as<uint2>();
    A.max_out_count = fa.max_out_count;
    A.compact = fa.compact;
    A.dense_nf_end = fa.dense_nf_end;
    A.dense_f_end = fa.dense_f_end;
    A.cold_rec = fa.compact ? h->d_cold_rec.as<uint2>() : nullptr;
    A.cold_edges = fa.compact ? h->d_cold_edges.as<uint32_t>() : nullptr;
    A.chain_lo = fa.chain_lo;
    A.chain_byte = fa.chain_byte.empty() ? nullptr : h->d_chain_byte.as<uint8_t>();
    A.chain_run = fa.chain_byte.empty() ? nullptr : h->d_chain_run.as<uint8_t>();
    A.win_table = fa.win_present ? h->d_win_table.as<uint16_t>() : nullptr;
    A.win_state = fa.win_present ? h->d_win_state.as<uint32_t>() : nullptr;
    A.win_hash = fa.win_present ? h->d_win_hash.as<uint2>() : nullptr;
    A.win_hash_bits = fa.win_hash_bits;
    A.chain_img = fa.chain.present() ? h->d_chain_img.as<uint8_t>() : nullptr;
    A.chain_img_bytes = uint32_t(fa.chain.image.size());
    A.chain_dense = fa.chain.n_dense;
    A.chain_index = fa.chain.n_index;
    A.chain_rec_off = fa.chain.rec_off;
    A.chain_cnt_off = fa.chain.cnt_off;
    A.chain_cnt_n = fa.chain.cnt_n;
    A.chain_head = fa.chain.present() ? h->d_chain_head.as<uint2>() : nullptr;
    A.chain_orig = fa.chain.present() ? h->d_chain_orig.as<uint32_t>() : nullptr;
    A.chain_frank = fa.chain.present() ? h->d_chain_frank.as<uint16_t>() : nullptr;
    A.chain_fhead = fa.chain.present() ? h->d_chain_fhead.as<uint2>() : nullptr;
    A.chain_finals = fa.chain.present() ? h->chain_finals : 0u;
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

int check_ready(acmi_handle *h) {
    if (!live(h)) return ACMI_E_INVALID;
    if (h->init_ok != 1) return ACMI_E_NOT_INIT;
    int f = ensure_finalized(h);  // lazy finalize (php_ahocorasick.c:707)
    if (f < 0) return f;
    return ensure_device(h);
}

}  // namespace acmi_engine

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
            for (DevBuf *b : {&h->d_table, &h->d_lds, &h->d_class, &h->d_own, &h->d_dict, &h->d_cnt, &h->d_plen, &h->d_anchor, &h->d_bitmap, &h->d_out_off, &h->d_out_list, &h->d_out_head, &h->d_cold_rec, &h->d_cold_edges, &h->d_chain_byte, &h->d_chain_run, &h->d_win_table, &h->d_win_state, &h->d_win_hash, &h->d_chain_img, &h->d_chain_head, &h->d_chain_orig, &h->d_chain_frank, &h->d_chain_fhead, &h->d_slot_pair, &h->d_pairs, &h->d_pat_bytes, &h->d_dense_img, &h->d_merge_counts, &h->d_merge_sync, &h->d_rep_off, &h->d_rep_bytes, &h->d_plan, &h->d_plan_out, &h->d_tail_out, &h->d_rep_sync, &h->d_nom_count, &h->d_out_len, &h->d_rep_out_off, &h->d_rep_out, &h->d_blob,
                              &h->d_off, &h->d_sync[0], &h->d_sync[1], &h->d_item_base[0], &h->d_item_hay[0], &h->d_item_rec[0], &h->d_zero[0], &h->d_isync[0],
                              &h->d_item_base[1], &h->d_item_hay[1], &h->d_item_rec[1], &h->d_zero[1], &h->d_isync[1],
                              &h->d_rec_base, &h->d_slab, &h->d_ovf, &h->d_counters, &h->d_first, &h->d_tile_queue, &h->d_tile_sync, &h->d_chain_log, &h->d_chain_tiles, &h->d_uni_off[0], &h->d_uni_off[1], &h->fly[0].d_prefix, &h->fly[0].d_out, &h->fly[1].d_prefix, &h->fly[1].d_out, &h->d_runs, &h->d_run_count, &h->d_busy, &h->d_dense})
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
            if (value < 0 || value > 1024) return ACMI_E_ARG;
            h->run_cap = value ? uint32_t(value) : 16u;
            h->run_cap_set = value != 0;   // (0: learned from the data again)
        } else if (k == "chunk_bytes") {
            if (value < 0) return ACMI_E_ARG;
            h->chunk_opt = uint64_t(value);
        } else if (k == "kernel") {
            if (value < 0 || value > 6 || value == 5) return ACMI_E_ARG;
            h->kernel = int(value);
            for (auto &row : h->choice)
                for (auto &c : row) c = acmi_handle::ChoiceClass{};
            h->anchor_chunk_shift = 0;   // (what the batches before taught the filter's hand-over goes with the measured choice)
            h->wave_armed = h->expand_armed = false;
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
            if (value != 0 && value != 16 && value != 32 && value != 64 && value != 128) return ACMI_E_ARG;
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
        } else if (k == "relist") {
            h->relist_opt = value ? 1 : 0;
        } else if (k == "wave_runs") {
            h->wave_runs_opt = value ? 1 : 0;
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
        } else if (k == "chain_walk") {   // kernel 6: -1 / 1 k_walk_chain where the set carries the chain-compressed image, 0 always k_walk_tiles
            if (value < -1 || value > 1) return ACMI_E_ARG;
            h->chain_walk = int(value);
        } else if (k == "chain_bytes") {   // budget of the chain-compressed LDS image of the whole automaton (0: none is built)
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            if (value < 0 || value > 160 * 1024 - 512 - 256) return ACMI_E_ARG;
            h->bopt.chain_bytes = uint32_t(value);
        } else if (k == "win_table") {   // the two-byte window table of the whole-item walk (kernel 4): -1 compact automata, 0 never, 1 whenever the shortest pattern is >= 3 B
            if (h->aut.finalized()) return ACMI_E_FINALIZED;
            h->bopt.win_table = value < 0 ? -1 : (value ? 1 : 0);
        } else if (k == "win_walk") {    // 0: kernel 4 walks the tables (k_walk_items) although the set has the window table
            h->win_walk = value ? 1 : 0;
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
        if (k == "item_tables_kept") { *value = double(h->item_tables_kept); return ACMI_OK; }
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
        else if (k == "anchor_chunk_shift") *value = h->anchor_chunk_shift;
        else if (k == "win_table") *value = fa.win_present ? 1 : 0;
        else if (k == "relist_reruns") *value = h->relist_reruns;
        else if (k == "busy_items") *value = h->busy_items;
        else if (k == "max_windows") *value = h->max_windows;
        else if (k == "serial_items") *value = h->serial_items;
        else if (k == "wave_runs_used") *value = h->wave_used ? 1 : 0;
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
        else if (k == "chain_walk_used") *value = h->chain_used ? 1 : 0;
        else if (k == "chain_log_reruns") *value = h->chain_log_reruns;
        else if (k == "chain_image_bytes") *value = double(h->aut.flat().chain.image.size() ? h->aut.flat().chain.image.size() : h->dA.chain_img_bytes);
        else if (k == "chain_dense_states") *value = h->aut.flat().chain.n_dense;
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
        const ChainImage &ci = fa.chain;
        out->chain_image = ci.present() ? ci.image.data() : nullptr;
        out->chain_bytes = uint32_t(ci.image.size());
        out->chain_dense = ci.n_dense;
        out->chain_index = ci.n_index;
        out->chain_rec_off = ci.rec_off;
        out->chain_cnt_off = ci.cnt_off;
        out->chain_cnt_n = ci.cnt_n;
        out->chain_orig = ci.present() ? ci.orig.data() : nullptr;
        out->chain_head = ci.present() ? ci.head.data() : nullptr;
        out->win_table = fa.win_present && !fa.win_table.empty() ? fa.win_table.data() : nullptr;
        out->win_state = fa.win_present && !fa.win_state.empty() ? fa.win_state.data() : nullptr;
        out->win_hash = fa.win_present && !fa.win_hash.empty() ? fa.win_hash.data() : nullptr;
        out->win_hash_bits = fa.win_hash_bits;
        out->win_list_bytes = kWinListBytes;
        out->shallow_hot = fa.shallow_hot;
        out->shallow_cold_end = fa.shallow_cold_end;
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
