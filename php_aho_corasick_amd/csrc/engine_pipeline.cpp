// engine_pipeline.cpp -- the batch pipeline (enqueue, end, the split set's two parts, the one-launch small path) and the match
// entry points of the C ABI (acmi_match*, acmi_result_*, acmi_stream_*).
#include "engine.hpp"

namespace acmi_engine {

// Which drain takes the items whose events overflowed their slabs: the caller's choice ("refill" 1 / 2), else the whole-item walk
// again in its record-writing form for the batches that kernel scanned (mode 3), else launch_refill's own choice.
// (not when the text answers with a record per two bytes or more -- a^1..a^16, every string over {a,b}: thousands of records
// per item, which only the wave drain writes as whole rows; `dense_output` is what the batch before produced)
int refill_mode(const acmi_handle *h, int kernel) { return h->refill != 0 ? h->refill : (kernel == 4 && !h->dense_output ? 3 : 0); }

constexpr uint32_t kWaveRunsMinHost = 3;   // (kdev.hpp kWaveRunsMin)

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
        // (text with hundreds of candidate windows per 8 KiB: smaller items -- a window list, an occurrence list in LDS and an event
        // slab of an item hold a few hundred entries -- "anchor_chunk_shift", learned: run_batch_end)
        const uint64_t step = 8192u >> std::min<uint32_t>(h->anchor_chunk_shift, 3u);
        cmin = std::max<uint64_t>(cmin, step);
        // ... and where haystacks are longer than that (they get chunked), a chunk inside its haystack must fit one step
        // WITH its warm-up, which k_items rounds up to a 128-byte boundary (align_warm): 8 KiB chunks were 513-514 quads,
        // i.e. a second, almost empty step per item (24-KiB haystacks: scan kernel 0.83 instead of 0.67 ms per 4 GiB)
        if (nh && blob_bytes / nh > step && (uint64_t(warm) + 32) * 2 < step) {
            const uint64_t steps = h->anchor_chunk_shift ? 1 : std::max<uint64_t>(1, (c0 + step - 1) / step);  // whole steps per chunk, less the warm-up's room
            const uint64_t c = steps * step - ((uint64_t(warm) + 128 + 127) & ~uint64_t(127));  // (warm-up rounded to a whole line)
            return uint32_t(std::min<uint64_t>(c, 1u << 30));
        }
        if (h->anchor_chunk_shift) return uint32_t(cmin);
    }
    uint64_t c = std::max<uint64_t>(cmin, (c0 + 127) & ~uint64_t(127));
    return uint32_t(std::min<uint64_t>(c, 1u << 30));
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
            c->max_result_bytes = h->max_result_bytes;   // (each part alone is bounded like the whole: run_split_end)
            if (c == h->part[1] && find_all && nh && !h->max_result_bytes && tiles_can_merge(c->dA, int(c->aut.flat().entry_bytes))) {
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
    if (h->rerun_kernel >= 0) {
        // a batch that is run again (its records outgrew the buffer; a split set's short part after the long part's records
        // changed) takes the kernel of its first run: the automatic choice may hold a pending probe of another one, and what the
        // caller of the re-run set up -- the merged output of k_walk_tiles -- belongs to that kernel
        kernel = h->rerun_kernel;
    } else if (h->kernel == 3) {
        F.strategy = choose_strategy(h, F, blob_bytes, nh, find_all);
        kernel = F.strategy == acmi_handle::kStratAnchor ? 2 : F.strategy == acmi_handle::kStratLds ? 1 : F.strategy == acmi_handle::kStratTiles ? 6 : 4;
    }
    // k_walk_tiles: one pass, records placed by a running sum instead of slab -> overflow -> second walk (findAll = true, patterns
    // of <= 129 B; else the whole-item walk)
    uint32_t tile_w = 0, tile_p = 0;
    if (kernel == 6 && find_all && tiles_geometry(fa.max_pattern_length, fa.n_states, blob_bytes, nh, &tile_w, &tile_p)) {
        // ("tile_w" 128: k_walk_chain only -- k_walk_tiles' queues are laid out for pieces of at most 64 bytes)
        const bool will_chain = h->chain_walk != 0 && chain_walk_ok(h->dA) && h->merge_into.out == nullptr && (h->chain_walk > 0 || h->dA.chain_dense < h->dA.chain_index);
        if (h->tile_w) tile_w = h->tile_w == 128 && !will_chain ? 64u : uint32_t(h->tile_w);
        else if (will_chain && tile_w == 64u && blob_bytes >= (64ull << 20) && blob_bytes / nh >= 256u && h->chain_log_div >= 32u) {
            // k_walk_chain over a big batch (a tile of 16 KiB for every wave of the chip: 64 MiB): pieces of 128 bytes -- the Lmax - 1
            // bytes of warm-up are a ninth of a piece's steps instead of a fifth, and half as many items and tiles go through k_items, k_expand and k_chain_emit (1 GiB of the
            // headline set at 64 occurrences per 8 KiB: the walk 0.738 -> 0.734 ms, the call 0.880 -> 0.824).  Not for text whose
            // events have overflowed the logs' first size (more than one per 26 bytes): a piece's event list is written and read
            // back through L2, and at an event per 9 bytes the lists of 128-byte pieces no longer fit it (the word list over
            // nothing but listed words: walk 1.19 ms with pieces of 64 bytes, 1.82 with 128)
            tile_w = 128u;
            tile_p = uint32_t(std::min<uint64_t>(64u, (blob_bytes / nh + 127u) / 128u));
        }
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
        uint64_t k = blob_bytes / (est_items * 8);   // (at most the batch's own bytes; round 5: half of them -- 2 KiB items of dense text need 256 slots)
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
    // the aligned forms of the anchor filter hand over window lists (4-byte entries, kdev.hpp): up to 1,024 windows an item -- what
    // k_runs_wave orders -- and half the batch's bytes (one window per eight bytes of an average item)
    const bool win = kernel == 2 && !dense;
    if (win) {
        run_max = 16;
        while (run_max < 1024u && uint64_t(run_max) * 2 * 8 * est_items <= blob_bytes) run_max *= 2;
    }
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
    HIPCHK(h, h->d_runs.ensure_keep(run_stride * Q * (dense ? 8 : win ? 4 : 16), h->retired));   // (dense: window starts, then k_dense_decide's results)
    HIPCHK(h, h->d_run_count.ensure_keep(max_items * 4, h->retired));
    // k_runs_wave (the items with many windows, a wave each) is a launch of its own: enqueued while batches report such items
    const bool wave_runs = win && find_all && h->wave_armed && h->dA.anchor_pairs != nullptr && h->wave_runs_opt != 0;
    if (wave_runs) HIPCHK(h, h->d_busy.ensure_keep(max_items * 4, h->retired));
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
    const bool regular = blob_bytes % nh == 0 && (blob_bytes / nh) % 8192 == 0 && (reinterpret_cast<uintptr_t>(d_blob) & 15u) == 0 && chunk >= 8192u;
    B.short_items = kernel != 2 ? 0u : h->short_rows >= 0 ? uint32_t(h->short_rows) : (regular ? 0u : 1u);
    B.tiny_items = (kernel == 2 && h->tiny_rows != 0 && blob_bytes / nh <= 2048) ? 1u : 0u;   // (waves whose items do not all fit a row take the usual path: the average may be skewed by a few long haystacks)
    B.slab_events = K;
    B.slab_major = K > 16u ? 1u : 0u;   // (big slabs -- text full of matches -- are item-major: kdev.hpp slab_slot)
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
    B.run_dense = dense ? 1u : win ? 2u : 0u;
    B.run_major = (dense || (win && Q > 16u)) ? 1u : 0u;   // (short lists entry-major: k_walk_runs' thread per item reads whole lines)
    B.wave_runs = wave_runs ? 1u : 0u;
    B.relist = (win && Q < F.run_max && !h->run_cap_set && h->relist_opt != 0) ? 1u : 0u;   // (an item whose list overflows is left undone: run_batch_end runs the batch again with longer lists)
    // a wave per item against a lane per item: the chip holds ~ 8,192 of k_runs_wave's waves at a time and 262,144 of k_walk_runs' lanes,
    // a window costs either the same chain of round trips -- so an item is the wave kernel's when it has more windows than the
    // batch's items take turns of waves (64 MiB of 8 KiB items: from four windows on; of 256-byte haystacks: from nine)
    // (... up to eight: beyond that the one lane's way is not a window's chain eight times over -- clusters of more than six
    // occurrences are WALKED, through tables in global memory for a big set: 1 KiB items with 64 windows each took 5.6 ms per GiB)
    B.wave_min = uint32_t(std::min<uint64_t>(8, std::max<uint64_t>(kWaveRunsMinHost, est_items / 8192)));
    // ... and the same for the expansion: k_expand_dense (a wave per item, a lane per event) is enqueued behind k_expand while the
    // batches before averaged four records an item or more (kernel 6 writes its records itself)
    const bool expand_dense = find_all && kernel != 6 && h->expand_armed && h->wave_runs_opt != 0;
    if (expand_dense) HIPCHK(h, h->d_dense.ensure_keep(max_items * 4, h->retired));
    B.dense_list = expand_dense ? h->d_dense.as<uint32_t>() : nullptr;
    B.busy_list = wave_runs ? h->d_busy.as<uint32_t>() : nullptr;
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
        // A uniform batch's item table is a function of its shape alone: a slot that holds the table of the same shape keeps it
        // (through the filter + k_walk_runs, which then writes every item's counters itself: a launch and its boundary less per
        // batch, ~ 10 us of the headline's step)
        const bool uniform = h->uniform_len && blob_bytes == h->uniform_len * nh;
        acmi_handle::ItemKey key;
        key.kernel = kernel;   // (findAll is not part of it: the table is the same, the counters are k_walk_runs' either way)
        key.nh = nh; key.len = h->uniform_len; key.max_items = max_items; key.chunk = B.chunk_bytes; key.warm = B.warm_bytes; key.align = B.align_warm;
        key.blob_mis = uint32_t(reinterpret_cast<uintptr_t>(B.blob) & 127u);
        key.rec = d_item_rec.p; key.hay = d_item_hay.p; key.base = d_item_base.p; key.zero = d_zero.p;
        acmi_handle::ItemKey &held = h->item_key[slot_index];
        const bool reuse = uniform && held.valid && held == key && kernel == 2 && !fa.anchor_dense && ist == st;
        B.zero_counters = reuse ? 1u : 0u;
        if (reuse) {
            ++h->item_tables_kept;
        } else {
            held.valid = false;
            if (uniform) launch_items_uniform(B, h->uniform_len, d_item_base.as<uint64_t>(), ist);
            else launch_items(B, blob_bytes, max_items, d_item_base.as<uint64_t>(), d_isync.as<uint32_t>(), ist);
            if (uniform) { held = key; held.valid = true; }
        }
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
    // kernel 6 over the whole automaton in LDS (k_walk_chain) when the set carries the chain-compressed image -- not for the short part
    // of a split set that merges the long part's records as it writes (k_walk_tiles' lane-list variant does that)
    // (a small automaton, every row of which fits the LDS, keeps k_walk_tiles' lane-list form unless "chain_walk" 1 asks: 8 patterns of
    // 2-6 B over their own alphabet, 540 records per 8 KiB, 1.32 ms per GiB there against 1.55 here)
    F.chain = kernel == 6 && h->chain_walk != 0 && chain_walk_ok(h->dA) && h->merge_into.out == nullptr &&
              (h->chain_walk > 0 || h->dA.chain_dense < h->dA.chain_index);
    if (kernel == 6 && F.chain) {
        // k_walk_chain's buffers: the pieces' overflow lists (per wave, reused tile after tile), the waves' event logs and the tiles'
        // {log offset, events per lane}.  The logs are sized optimistically -- an event per chain_log_div bytes of text, learned like
        // the record buffer: a batch that overflows them is run again (run_batch_end) -- and never beyond an event per byte.
        const uint64_t tiles = chain_tiles(max_items, tile_p), waves = chain_waves(B.reserve_cus);
        const uint64_t active = std::min(tiles, waves);   // (a small batch has fewer tiles than the grid has waves: the others touch nothing)
        const uint64_t worst = ((tiles + waves - 1) / waves) * 64ull * chain_np() * (tile_w + 16ull);
        const uint64_t want = (blob_bytes / std::max<uint32_t>(1, h->chain_log_div)) / waves * 5 / 4 + 64ull * chain_np() * (tile_w + 16ull);
        // (at an event per byte the AVERAGE wave's share is no bound -- tiles are dealt t % waves, and haystacks whose density recurs
        // with the grid's period give one wave more than the mean: the last size is the true per-wave bound, `worst`)
        const uint32_t log_cap = uint32_t(std::min<uint64_t>(h->chain_log_div <= 1u ? worst : std::min(worst, want), 0xFFFFFF00ull));
        HIPCHK(h, h->d_tile_queue.ensure_keep(chain_spill_words(tile_w, B.reserve_cus) / waves * active * 4, h->retired));
        HIPCHK(h, h->d_chain_log.ensure_keep(active * log_cap * 4, h->retired));
        HIPCHK(h, h->d_chain_tiles.ensure_keep(tiles * 8, h->retired));
        F.chain_buf = ChainBuffers{h->d_tile_queue.as<uint32_t>(), h->d_chain_log.as<uint32_t>(), log_cap, h->d_chain_tiles.as<uint32_t>()};
    } else if (kernel == 6) {
        HIPCHK(h, h->d_tile_queue.ensure_keep(tiles_queue_entries(tile_w) * 8, h->retired));
        HIPCHK(h, h->d_tile_sync.ensure_keep(8 + tiles_status_words(max_items, tile_p) * 8, h->retired));
        HIPCHK(h, hipMemsetAsync(h->d_tile_sync.p, 0, 8 + tiles_status_words(max_items, tile_p) * 8, st));
    }
    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[1], st));
    F.merged = false;
    F.reran = false;
    if (kernel == 6 && F.chain) {
        HIPCHK(h, launch_walk_chain(h->dA, B, max_items, int(fa.class_mode), tile_w, tile_p, F.chain_buf, st));
    } else if (kernel == 6) {
        F.merged = h->merge_into.out != nullptr;
        HIPCHK(h, launch_walk_tiles(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), tile_w, tile_p, h->d_tile_queue.p,
                                    h->d_tile_sync.as<uint32_t>(), h->d_tile_sync.as<uint64_t>() + 1, F.merged ? h->merge_into.out : F.d_out.p,
                                    F.merged ? h->merge_into.cap : cap, st, F.merged ? h->merge_into.prefix : nullptr, F.merged ? h->merge_into.records : nullptr));
    } else if (kernel == 4) {
        // exact walk only (k_walk_items): for text on which k_scan_lds' optimistic
        // pass flags everything anyway (events in every wave-quad)
        if (h->dA.win_table != nullptr && h->win_walk) HIPCHK(h, launch_walk_win(h->dA, B, max_items, int(fa.entry_bytes), st));
        else HIPCHK(h, launch_walk_items(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), h->walk_group, st));
    } else {
        HIPCHK(h, launch_scan(h->dA, B, max_items, int(fa.entry_bytes), kernel, int(fa.class_mode), h->ilp, h->group_quads, st, &F.anchor_variant));
    }
    if (h->profile) HIPCHK(h, hipEventRecord(F.ev[2], st));
    if (h->items_ahead) HIPCHK(h, hipEventRecord(h->ev_scanned[slot_index], st));
    // the anchor kernel only flags byte runs; their exact walk is a kernel of its own (one thread per item)
    // (its LDS tile only where a walk may be needed: no verification by compare, or the batch before walked a cluster)
    if (kernel == 2) HIPCHK(h, launch_walk_runs(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), st,
                                                h->dA.anchor_slot_pair == nullptr || h->fallback_bytes != 0 || !find_all));
    if (B.wave_runs) HIPCHK(h, launch_runs_wave(h->dA, B, max_items, int(fa.entry_bytes), int(fa.class_mode), st));
    F.wave_runs = B.wave_runs != 0u;
    F.win_lists = win;
    F.relist = B.relist != 0u;
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
    if (B.dense_list != nullptr) launch_expand_dense(h->dA, B, max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, st);
    // (k_walk_chain left events in the waves' logs: records at the bases k_expand's scan has just written)
    if (kernel == 6 && F.chain) HIPCHK(h, launch_chain_emit(h->dA, B, max_items, tile_w, tile_p, F.chain_buf, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, st));
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
    if ((rc < 0 && rc != ACMI_E_PARTIAL) || (rc2 < 0 && rc2 != ACMI_E_PARTIAL)) {
        const bool first = rc < 0 && rc != ACMI_E_PARTIAL;
        h->last_error = first ? a->last_error : b->last_error;
        return first ? rc : rc2;
    }
    hipStream_t st = F.stream;
    uint32_t nh = uint32_t(F.nh);
    const uint32_t first_only = F.find_all ? 0u : 1u;
    uint64_t ta = a->last_total, tb = b->last_total;
    bool partial = false;
    if (h->max_result_bytes) {
        // "max_result_bytes" on a split set: each part was bounded on its own (ACMI_E_PARTIAL: it holds the records of its first
        // resume_haystack haystacks); the merged result takes the haystacks BOTH hold whose records together fit the bound --
        // the largest k with pa[k] + pb[k] <= bound, by single-word copies (an exception, not a rate)
        const uint64_t bound = h->max_result_bytes / 16;
        uint64_t k = std::min<uint64_t>(rc == ACMI_E_PARTIAL ? a->resume_haystack : nh, rc2 == ACMI_E_PARTIAL ? b->resume_haystack : nh);
        const uint64_t *dpa = a->d_prefix().as<uint64_t>(), *dpb = b->d_prefix().as<uint64_t>();
        auto both = [&](uint64_t i, uint64_t *va, uint64_t *vb) -> hipError_t {
            hipError_t e = hipMemcpy(va, dpa + i, 8, hipMemcpyDeviceToHost);
            return e != hipSuccess ? e : hipMemcpy(vb, dpb + i, 8, hipMemcpyDeviceToHost);
        };
        uint64_t va = 0, vb = 0;
        HIPCHK(h, both(k, &va, &vb));
        if (va + vb > bound) {
            uint64_t lo = 0, hi = k;   // sum[lo] <= bound < sum[hi]
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                HIPCHK(h, both(mid, &va, &vb));
                if (va + vb <= bound) lo = mid; else hi = mid;
            }
            k = lo;
            HIPCHK(h, both(k, &va, &vb));
        }
        if (k < nh) {
            if (k == 0) {
                h->have_result = false;
                h->last_error = "the records of one haystack alone exceed max_result_bytes";
                return ACMI_E_TOO_LARGE;
            }
            partial = true;
            nh = uint32_t(k);
            ta = va;
            tb = vb;
        }
    }
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
        b->rerun_kernel = FB.kernel_used;
        int rr = enqueue_batch(b, FB, FB.d_blob, FB.blob_bytes, FB.d_off, FB.nh, FB.find_all);
        b->merge_into = acmi_handle::MergeInto{};
        if (rr >= 0 && !FB.merged) {
            // (cannot happen while the kernel is pinned and the geometry is a function of the batch: if the re-run did not merge
            // after all, the part's OWN buffer must hold its records before k_merge_parts reads them)
            b->out_want = std::max<size_t>(b->out_want, size_t(tb) * 16);
            rr = enqueue_batch(b, FB, FB.d_blob, FB.blob_bytes, FB.d_off, FB.nh, FB.find_all);
        }
        b->rerun_kernel = -1;
        if (rr < 0) { b->sync_tiles = 0; h->last_error = b->last_error; return rr; }
        b->enqueue_seq = FB.seq;
        HIPCHK(h, hipStreamSynchronize(st));
        merged = FB.merged;
        if (!merged && FB.out_cap < tb) { h->last_error = "split set: the short part's records did not fit its buffer on the second run"; return ACMI_E_DEVICE; }
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
    h->resume_haystack = nh;
    if (partial) {
        h->last_error = "the batch's records exceed max_result_bytes: the result holds the first haystacks only (resume_haystack)";
        return ACMI_E_PARTIAL;
    }
    return ACMI_OK;
}

// What the hand-over behind the filter learns from a batch (window lists, round 6): the scan reports the longest list the batch
// asked for (word 21), k_walk_runs the items with more than a few windows (word 20) -- the lists of the batches behind are sized
// in one step, and k_runs_wave is enqueued while there are such items.
static void learn_handover(acmi_handle *h, const acmi_handle::FlySlot &F, const uint32_t *pinned32) {
    if (F.win_lists) {
        h->busy_items = pinned32[20];
        h->max_windows = pinned32[21];
        h->serial_items = pinned32[22];
        h->wave_used = F.wave_runs;
        // (armed by the longest list as well: on a batch whose lists all overflowed k_walk_runs saw no item it could have listed)
        h->wave_armed = pinned32[20] != 0u || pinned32[21] > F.args.wave_min;
        if (pinned32[21] > F.run_used && F.run_used < F.run_max) {
            uint32_t want = 16;
            while (want < pinned32[21] + pinned32[21] / 4u && want < F.run_max) want *= 2;
            h->run_auto = std::max(h->run_auto, want);
        }
        // ... and back: eight batches in a row whose longest list had room in the short form (16 entries, entry-major: what the thread
        // per item of a sparse batch reads as whole lines) give the long, item-major lists up again
        if (!h->run_cap_set && F.run_used > 16u && pinned32[21] <= 12u) {
            if (++h->sparse_batches >= 8u) { h->run_auto = 16; h->sparse_batches = 0; }
        } else {
            h->sparse_batches = 0;
        }
        // (a window is an event at most -- but for patterns that share an anchor: the slabs follow the lists in one step as well)
        if (!h->slab_set && pinned32[21] > F.slab_used && h->wave_armed) {
            uint32_t want = 16;
            while (want < pinned32[21] && want < 256u) want *= 2;
            h->slab_auto = std::max(h->slab_auto, want);
        }
        // more than ~ 200 windows an item: smaller items (8 KiB -> 4, 2, 1 KiB), so that an item has ~ 128 at most
        if (!h->chunk_opt && pinned32[21] > 192u && F.chunk_used >= 2048u && h->anchor_chunk_shift < 3u) {
            uint32_t per = pinned32[21], shift = h->anchor_chunk_shift;
            uint64_t c = F.chunk_used;
            while (per > 160u && c >= 2048u && shift < 3u) { per = (per + 1u) / 2u; c /= 2; ++shift; }
            h->anchor_chunk_shift = shift;
        } else if (!h->chunk_opt && h->anchor_chunk_shift && pinned32[21] <= 60u) {
            --h->anchor_chunk_shift;   // (sparser text again: items twice the size would still have ~ 120 windows at most)
        }
    }
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
    h->chain_used = F.chain && F.kernel_used == 6;
    if (F.trivial) {
        h->have_result = true;
        return ACMI_OK;
    }
    const FlatAutomaton &fa = h->aut.flat();
    // k_walk_chain's event logs are sized for the batches before: one that overflowed them (word 19) is run again with logs for eight
    // times the events -- from an event per 32 bytes of text to, at most, one per byte
    for (int tries = 0; F.kernel_used == 6 && F.chain && pinned32[19] != 0u; ++tries) {
        if (tries == 3 || h->chain_log_div == 1u) { h->last_error = "k_walk_chain: event log overflow at its largest size"; return ACMI_E_DEVICE; }
        {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { h->sync_tiles = 0; return fail_hip(h, e, "hipStreamSynchronize"); }
        }
        h->chain_log_div = std::max<uint32_t>(1u, h->chain_log_div / 8u);
        ++h->chain_log_reruns;
        h->rerun_kernel = 6;
        const int rc = enqueue_batch(h, F, F.d_blob, F.blob_bytes, F.d_off, F.nh, F.find_all);
        h->rerun_kernel = -1;
        if (rc < 0) { h->sync_tiles = 0; return rc; }
        h->enqueue_seq = F.seq;
        F.timed = false;
        HIPCHK(h, hipStreamSynchronize(st));
    }
    // Window lists that overflowed while they could still grow: k_walk_runs left those items undone (BatchArgs::relist) instead of
    // walking them whole -- 8 KiB through tables in global memory per item, then the drain's second walk: 34 ms for the first 64 MiB of
    // dense text a handle of the 65,536-pattern set met -- and the batch runs again with the lists, slabs and items the scan's own
    // count asks for (0.4 ms).
    for (int tries = 0; F.relist && pinned32[21] > F.run_used && pinned32[6] == 0u && tries < 4; ++tries) {
        {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { h->sync_tiles = 0; return fail_hip(h, e, "hipStreamSynchronize"); }
        }
        learn_handover(h, F, pinned32);
        ++h->relist_reruns;
        h->rerun_kernel = F.kernel_used;
        const int rc = enqueue_batch(h, F, F.d_blob, F.blob_bytes, F.d_off, F.nh, F.find_all);
        h->rerun_kernel = -1;
        if (rc < 0) { h->sync_tiles = 0; return rc; }
        h->enqueue_seq = F.seq;
        F.timed = false;
        HIPCHK(h, hipStreamSynchronize(st));
    }
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
    learn_handover(h, F, pinned32);
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
    const uint64_t bound = h->max_result_bytes ? h->max_result_bytes / 16 : ~0ull;   // (a split set's parts too: run_split_end takes what both hold)
    const bool partial = total > bound;
    const uint64_t need = partial ? bound : total;   // records the buffer must hold
    if (!F.merged) h->out_want = std::max<size_t>(h->out_want, size_t(need) * 16);
    // (a merged batch wrote into the parent's buffer: whether THAT was big enough is the parent's business, run_split_end)
    const bool too_small = !F.merged && need > F.out_cap;   // did not fit the optimistic capacity: grow, expand again
    const bool undrained = n_ovf > 0 && !F.refill_launched;      // items overflowed their slabs and the drain was not enqueued
    h->refill_armed = n_ovf > 0;
    // (a pipeline that has to be completed or run again below says nothing about what the strategy costs -- nor does a batch whose
    // hand-over behind the filter was still sized for other text: its lists overflowed, its items are about to shrink, or the
    // kernels that take items with many windows / events a wave each were not enqueued yet.  Such a batch cost the filter 25 ms
    // per 64 MiB of the 65,536-pattern set at 128 occurrences per 8 KiB, the next one 0.19 -- and the measured choice had left by then)
    bool handover_learning = false;
    if (F.win_lists) {
        const uint32_t busy = pinned32[20], maxw = pinned32[21];
        handover_learning = (maxw > F.run_used && F.run_used < F.run_max) ||
                            ((busy != 0 || maxw > F.args.wave_min) && !F.wave_runs && F.find_all && h->wave_runs_opt != 0 && h->dA.anchor_pairs != nullptr) ||
                            (!h->chunk_opt && maxw > 192u && F.chunk_used >= 2048u && h->anchor_chunk_shift < 3u);
    }
    if (F.find_all && F.kernel_used != 6 && F.args.dense_list == nullptr && h->wave_runs_opt != 0 && total >= 4 * std::max<uint64_t>(1, h->items))
        handover_learning = true;   // (k_expand_dense joins from the next batch on)
    if (F.strategy >= 0 && F.blob_bytes >= kAutoMinBytes) {
        if (F.timed && !too_small && !undrained && handover_learning) {
            if (F.probe) h->choice[F.find_all ? 1 : 0][F.choice_class].pending |= 1u << F.strategy;   // (the probe is taken again)
        } else if (F.timed && !too_small && !undrained) learn_strategy(h, F, double(pipeline_ms) * double(1ull << 30) / double(F.blob_bytes), double(total) / double(F.blob_bytes));
        else if (F.probe) h->choice[F.find_all ? 1 : 0][F.choice_class].pending |= 1u << F.strategy;   // (the probe is taken again)
    }
    F.reran = too_small || undrained;
    if (too_small || undrained) {
        if (h->enqueue_seq == F.seq) {
            if (too_small) HIPCHK(h, F.d_out.ensure(need * 16));
            const uint64_t cap = partial ? bound : F.d_out.cap / 16;
            F.out_cap = cap;
            if (F.kernel_used == 6 && F.chain) {   // the logs once more, into the grown buffer (bases, prefix array and summary stand)
                HIPCHK(h, launch_chain_emit(h->dA, F.args, F.max_items, F.tile_w, F.tile_p, F.chain_buf, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, st));
            } else if (F.kernel_used == 6) {   // the walk again, into the grown buffer (prefix array and summary stand)
                const size_t sync_bytes = 8 + tiles_status_words(F.max_items, F.tile_p) * 8;
                HIPCHK(h, hipMemsetAsync(h->d_tile_sync.p, 0, sync_bytes, st));
                HIPCHK(h, launch_walk_tiles(h->dA, F.args, F.max_items, int(fa.entry_bytes), int(fa.class_mode), F.tile_w, F.tile_p, h->d_tile_queue.p,
                                            h->d_tile_sync.as<uint32_t>(), h->d_tile_sync.as<uint64_t>() + 1, F.d_out.p, cap, st));
            } else if (too_small) {
                // (every item by k_expand itself this time: the dense items' list was filled by the first expansion)
                BatchArgs again = F.args;
                again.dense_list = nullptr;
                launch_expand(h->dA, again, F.max_items, h->d_rec_base.as<uint64_t>(), F.d_out.p, cap, F.d_prefix.as<uint64_t>(),
                              h->pinned_dev + kPinWords * slot_index, F.sync_next, F.sync_words, nullptr, nullptr,
                              h->d_isync[slot_index].as<uint32_t>() + 1, false, st);
            }
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
            h->rerun_kernel = F.kernel_used;
            const int rc = enqueue_batch(h, F, F.d_blob, F.blob_bytes, F.d_off, F.nh, F.find_all);
            h->rerun_kernel = -1;
            if (rc < 0) { h->sync_tiles = 0; return rc; }
            h->enqueue_seq = F.seq;
            HIPCHK(h, hipStreamSynchronize(st));
            if (pinned[1] != total) { h->last_error = "record count changed between two runs of one batch"; return ACMI_E_DEVICE; }
        }
    }
    h->last_total = total;
    h->expand_armed = total >= 4 * std::max<uint64_t>(1, h->items);
    // (big slabs -- item-major, learned from text full of matches -- are given up after eight batches in a row with fewer than two
    // records an item and nothing drained: the sparse batch's thread per item reads the small, event-major slab as whole lines)
    if (!h->slab_set && F.slab_used > 32u && total < 2 * std::max<uint64_t>(1, h->items) && n_ovf == 0) {
        if (++h->sparse_slab_batches >= 8u) { h->slab_auto = 0; h->sparse_slab_batches = 0; }
    } else {
        h->sparse_slab_batches = 0;
    }
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
    if (!h->pin_small_dev) {   // (asked once: the mapping does not move)
        void *dev = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dev, pin, 0));
        h->pin_small_dev = static_cast<uint8_t *>(dev);
    }
    uint8_t *d = h->pin_small_dev;
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
    if (small_trace())
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

}  // namespace acmi_engine

extern "C" {

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

}  // extern "C"
