// k_small.hip -- k_match_small: a whole small host batch in one launch
#include "kdev.hpp"

namespace acmi {

// ------------------------------------------------------------------------------------------
// k_match_small: a whole small batch (<= 64 KiB of haystacks, <= 1024 of them) in ONE launch.
//
// The batch pipeline above is a dozen dependent launches, ~110 us end to end whatever the size --
// which is what one ahocorasick_match() of a few KiB costs, and PHP code calls it in a loop
// (examples/benchmark.php:65-67).  Here one 1024-thread workgroup reads offsets and bytes straight from
// pinned host memory into LDS (coalesced 16-byte reads over PCIe), walks 64-byte work items with the
// usual Lmax-1 warm-up through an LDS tile of the transition table (cold rows from the global table),
// prefix-sums the record counts in LDS and writes prefix array and records back to pinned host memory:
// one launch, one synchronisation, no copies.  Same exact walk, same order, same findAll=false rule.
// A batch whose records do not fit `cap` reports so and the host takes the general path.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kSmallMaxHay = 1024;
constexpr uint32_t kSmallMaxItems = 2048;       // small_scan handles 2 x 1024 values
constexpr uint32_t kSmallEvents = 2;            // events kept per item; more -> the item is walked again
constexpr uint32_t kSmallThreads = 1024;
constexpr uint32_t kSmallLds = 160u * 1024u - 256u;

struct SmallLayout {
    uint32_t chunk;       // bytes per work item: 16, 32 or 64 -- the smallest that leaves the batch <= 2,048 items
    uint32_t max_items;   // item slots (a multiple of 2, <= kSmallMaxItems)
    uint32_t in_bytes;    // LDS for the haystack bytes (their span rounded up to 16)
    uint32_t tile_bytes;  // LDS for the transition tile: what is left
};

static SmallLayout small_layout(uint32_t span, uint32_t nh, uint32_t hot_bytes, bool table_mode) {
    SmallLayout Y{};
    // (the smallest items that still fit the item arrays: more lanes walk, each fewer steps -- the warm-up of Lmax - 1 bytes per
    // item is cheap next to a chip that would otherwise run two waves)
    // (8 bytes when that gives every thread ONE item: one 8 KiB haystack = 1,024 items of 8 + 15 steps instead of 512 of 16 + 15)
    Y.chunk = (span + 7u) / 8u + nh <= kSmallThreads + 1u ? 8u : (span + 15u) / 16u + nh <= kSmallMaxItems ? 16u : (span + 31u) / 32u + nh <= kSmallMaxItems ? 32u : 64u;
    Y.max_items = std::min<uint32_t>(kSmallMaxItems, ((span + Y.chunk - 1u) / Y.chunk + nh + 1u) & ~1u);
    if (Y.max_items < 2u) Y.max_items = 2u;
    Y.in_bytes = (span + 15u) & ~15u;
    (void)table_mode;
    const uint32_t fixed = Y.max_items * (kSmallEvents * 8u + 8u) + (kSmallMaxHay + 8u) * 8u + 64u + 512u;
    const uint32_t room = kSmallLds > Y.in_bytes + fixed ? kSmallLds - Y.in_bytes - fixed : 0u;
    Y.tile_bytes = std::min<uint32_t>(room, (hot_bytes + 15u) & ~15u) & ~15u;
    return Y;
}

// exclusive scan of v[0..n) (n <= 2 * kSmallThreads) in LDS, in place; returns the total.  All threads call.
__device__ __forceinline__ uint32_t small_scan(uint32_t *v, uint32_t n, uint32_t *wave_tmp) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t a = 2 * t < n ? v[2 * t] : 0u, b = 2 * t + 1 < n ? v[2 * t + 1] : 0u;
    const uint32_t sum = a + b;
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= uint32_t(d)) inc += o;
    }
    if (lane == 63) wave_tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < kSmallThreads / 64u; ++w) {
        const uint32_t x = wave_tmp[w];
        if (w < wave) base += x;
        total += x;
    }
    const uint32_t ex = base + inc - sum;
    if (2 * t < n) v[2 * t] = ex;
    if (2 * t + 1 < n) v[2 * t + 1] = ex + a;
    __syncthreads();
    return total;
}

template <typename E, int MODE>
__global__ __launch_bounds__(kSmallThreads) void k_match_small(DevAutomaton A, const uint64_t *offsets, const uint8_t *blob,
                                                               uint32_t nh, uint32_t find_all, uint32_t max_pattern_length,
                                                               uint64_t *out_summary, uint64_t *out_prefix, uint4 *out_records,
                                                               uint32_t cap, SmallLayout Y, uint64_t seq, uint32_t n_bytes) {
    // LDS, laid out by the host for THIS batch (small_layout): the bytes and per-item arrays take what the batch needs, the tile
    // of the transition table everything else -- one 8 KiB haystack leaves ~ 140 KiB to the tile (every state of the first nine
    // trie levels of the headline set) where a fixed 64 KiB for the bytes left 36: with two waves on the chip nothing hides a row
    // that comes from L2, and 94 % of the lockstep steps had one (24 of the kernel's 31 us).
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t kChunk = Y.chunk, kItems = Y.max_items;
    uint8_t *in_bytes = smem;
    const uint8_t *tile = smem + Y.in_bytes;
    uint2 *ev = reinterpret_cast<uint2 *>(smem + Y.in_bytes + Y.tile_bytes);
    uint32_t *cnt_ev = reinterpret_cast<uint32_t *>(ev + kItems * kSmallEvents);
    uint32_t *cnt_rec = cnt_ev + kItems;
    uint32_t *item_base = cnt_rec + kItems;                                          // 1025 x 4 (+ pad) = 4.5 KiB
    uint32_t *hay_off = item_base + kSmallMaxHay + 8;                                // 1025 x 4
    uint32_t *wave_tmp = hay_off + kSmallMaxHay + 8;                                 // 16 x 4
    uint16_t *cls = reinterpret_cast<uint16_t *>(wave_tmp + 16);                     // 512 B (TABLE mode)

    const uint32_t t = threadIdx.x;
    // (phase stamps of thread 0, 100 MHz: summary[4..10] -- ACMI_SMALL_TRACE prints them; seven stores to pinned memory)
    const uint64_t t_start = wall_clock64();
    // (only when asked for -- find_all bit 1: seven stores into pinned host memory that the system-scope fence at the end waits for)
    const bool stamps = (find_all & 2u) != 0u;
    find_all &= 1u;
    auto stamp = [&](int i) { if (stamps && t == 0) out_summary[4 + i] = wall_clock64() - t_start; };
    // offsets and haystack bytes (16 per lane, coalesced, 16-byte aligned by the host) from pinned host memory, requested together
    // -- the bytes' length comes with the launch: one trip over PCIe, not two (k_match_small_chain below has the story)
    const uint32_t n16 = (n_bytes + 15u) / 16u;
    const uint4 *src = reinterpret_cast<const uint4 *>(blob);
    const uint64_t off0 = offsets[0];
    const uint64_t off_t = t <= nh ? offsets[t] : 0ull;
    const uint4 text_t = t < n16 ? src[t] : make_uint4(0u, 0u, 0u, 0u);
    {
        uint4 *dst = reinterpret_cast<uint4 *>(in_bytes);
        const uint32_t tile_rows_b = A.row_bytes ? (Y.tile_bytes / A.row_bytes < A.trap ? Y.tile_bytes / A.row_bytes : A.trap) * A.row_bytes : 0u;
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + Y.in_bytes);
        for (uint32_t i = t; i < (tile_rows_b + 15u) / 16u; i += kSmallThreads) tdst[i] = tsrc[i];
        if (MODE == 0 && t < 256) cls[t] = A.class_off[t];
        if (t < n16) dst[t] = text_t;
        for (uint32_t i = t + kSmallThreads; i < n16; i += kSmallThreads) dst[i] = src[i];
        if (t <= nh) hay_off[t] = uint32_t(off_t - off0);   // (relative to the first one)
        for (uint32_t h = t + kSmallThreads; h <= nh; h += kSmallThreads) hay_off[h] = uint32_t(offsets[h] - off0);
    }
    __syncthreads();
    for (uint32_t h = t; h < kSmallMaxHay + 1; h += kSmallThreads)   // items per haystack
        item_base[h] = h < nh ? (hay_off[h + 1] - hay_off[h] + kChunk - 1) / kChunk : 0u;
    __syncthreads();
    stamp(0);   // offsets, bytes and tile are in LDS
    const uint32_t tile_rows = A.row_bytes ? (Y.tile_bytes / A.row_bytes < A.trap ? Y.tile_bytes / A.row_bytes : A.trap) : 0u;
    const uint32_t n_items = small_scan(item_base, nh + 1, wave_tmp);
    stamp(1);  // item_base[h] = first item of haystack h; [nh] = total

    auto step = [&](uint32_t s, uint32_t byte) -> uint32_t {
        const uint32_t coff = class_offset<E, MODE>(A, cls, byte);
        uint32_t w = A.trap;
        if (s < tile_rows) w = lds_read<E>(tile, s * A.row_bytes + coff);
        if (w >= A.trap) w = cold_step<E>(A, s, byte, coff);
        return w;
    };
    // item i -> (haystack, first byte, bytes, warm-up bytes, position of its first byte)
    auto decode = [&](uint32_t i, uint32_t &h, uint32_t &begin, uint32_t &len, uint32_t &warm, uint32_t &pos0) {
        uint32_t lo = 0, hi = nh;  // item_base[lo] <= i < item_base[hi]
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (item_base[mid] <= i) lo = mid; else hi = mid;
        }
        h = lo;
        pos0 = (i - item_base[h]) * kChunk;
        begin = hay_off[h] + pos0;
        const uint32_t rest = hay_off[h + 1] - begin;
        len = rest < kChunk ? rest : kChunk;
        const uint32_t wmax = max_pattern_length ? max_pattern_length - 1u : 0u;
        warm = pos0 < wmax ? pos0 : wmax;
    };

    // ---- walk: count events and records per item, keep the first kSmallEvents events ----
    for (uint32_t i = t; i < kItems; i += kSmallThreads) {
        uint32_t n_ev = 0, n_rec = 0;
        if (i < n_items) {
            uint32_t h, begin, len, warm, pos0;
            decode(i, h, begin, len, warm, pos0);
            uint32_t s = 0;
            const uint8_t *p = in_bytes + begin - warm;
            const uint32_t n = warm + len;
            auto one = [&](uint32_t byte, uint32_t j) {
                s = step(s, byte);
                if (j >= warm && is_final(A, s)) {
                    n_rec += A.out_count[s];
                    if (n_ev < kSmallEvents) ev[i * kSmallEvents + n_ev] = make_uint2(pos0 + (j - warm) + 1u, s);
                    ++n_ev;
                }
            };
            // (the bytes of four steps are read together, ahead of the steps: the text does not depend on the state, and with a
            // handful of waves on the chip a dependent LDS read per byte in front of the table's was a third of a step)
            uint32_t j = 0;
            for (; j + 4u <= n; j += 4u) {
                const uint32_t b0 = p[j], b1 = p[j + 1u], b2 = p[j + 2u], b3 = p[j + 3u];
                one(b0, j); one(b1, j + 1u); one(b2, j + 2u); one(b3, j + 3u);
            }
            for (; j < n; ++j) one(p[j], j);
        }
        cnt_ev[i] = n_ev;
        cnt_rec[i] = n_rec;
    }
    __syncthreads();
    stamp(2);   // walked
    // ---- findAll == false: only the first event of each haystack (php_ahocorasick.c:588) ----
    if (!find_all) {
        for (uint32_t h = t; h < nh; h += kSmallThreads) {
            bool found = false;
            for (uint32_t i = item_base[h]; i < item_base[h + 1]; ++i) {
                if (found) { cnt_ev[i] = 0; cnt_rec[i] = 0; }
                else if (cnt_ev[i]) {
                    found = true;
                    cnt_ev[i] = 1;
                    cnt_rec[i] = A.out_count[ev[i * kSmallEvents].y];
                }
            }
        }
        __syncthreads();
    }
    // ---- record offsets ----
    const uint32_t total = small_scan(cnt_rec, kItems, wave_tmp);  // cnt_rec[i] = first record of item i
    for (uint32_t h = t; h <= nh; h += kSmallThreads)
        out_prefix[h] = (h < nh && item_base[h] < kItems) ? cnt_rec[item_base[h]] : total;
    if (t == 0) { out_summary[0] = total; out_summary[1] = n_items; out_summary[2] = total > cap ? 1u : 0u; }
    stamp(3);   // scanned, prefix written
    // summary[3] = the call's sequence number, stored LAST behind a system-scope fence: the host polls it in its pinned memory
    // instead of sleeping on the stream (an interrupt and a thread wake-up, ~ 8 us of a 30 us call)
    if (total > cap) {
        __threadfence_system();
        __syncthreads();
        if (t == 0) __hip_atomic_store(out_summary + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // ---- records ----
    for (uint32_t i = t; i < n_items; i += kSmallThreads) {
        const uint32_t n_ev = cnt_ev[i];
        if (n_ev == 0) continue;
        uint32_t h, begin, len, warm, pos0;
        decode(i, h, begin, len, warm, pos0);
        uint64_t w = cnt_rec[i];
        if (n_ev <= kSmallEvents) {
            for (uint32_t e = 0; e < n_ev; ++e) w = emit_event(A, out_records, w, h, ev[i * kSmallEvents + e].x, ev[i * kSmallEvents + e].y);
        } else {  // dense item: walk it again, writing records straight to their slots
            uint32_t s = 0;
            const uint8_t *p = in_bytes + begin - warm;
            for (uint32_t j = 0; j < warm + len; ++j) {
                s = step(s, p[j]);
                if (j >= warm && is_final(A, s)) w = emit_event(A, out_records, w, h, pos0 + (j - warm) + 1u, s);
            }
        }
    }
    stamp(4);   // records written
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(out_summary + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// k_match_small_chain: the same call for a pattern set that carries the chain-compressed image (automaton.hpp ChainImage, k_chain.hip):
// the WHOLE automaton in LDS.  One 8 KiB haystack is 23 dependent steps of 1,024 lanes (8-byte items + Lmax - 1 bytes of warm-up);
// with a tile of dense rows the lanes that meet the haystack's occurrence leave the tile and every one of their steps is an L2
// round trip the whole workgroup waits for (round 4: 19.4 us in the kernel) -- here every step is two LDS reads.  The image takes
// 130-142 KiB for a few thousand patterns: the per-item arrays are cut to the bone (ONE event kept per item, 12 B per item) and
// the call falls back to k_match_small when text + items do not fit beside the image.
// ------------------------------------------------------------------------------------------
struct SmallChainLayout {
    uint32_t chunk, max_items, in_off, items_off, hay_off, lds;
};

static bool small_chain_layout(const DevAutomaton &A, uint32_t span, uint32_t nh, bool table_mode, SmallChainLayout &Y) {
    if (A.chain_img == nullptr || A.chain_head == nullptr || A.chain_orig == nullptr) return false;
    const uint32_t img = (A.chain_img_bytes + 15u) & ~15u;
    const uint32_t in_bytes = (span + 15u) & ~15u;
    Y.in_off = img + (table_mode ? 512u : 0u);
    Y.items_off = Y.in_off + in_bytes;
    for (uint32_t chunk : {8u, 16u, 32u, 64u}) {
        const uint32_t items = ((span + chunk - 1u) / chunk + nh + 1u) & ~1u;
        if (items > kSmallMaxItems || (chunk == 8u && items > kSmallThreads + 2u)) continue;
        const uint32_t hay = Y.items_off + std::max(items, 2u) * 12u;
        const uint32_t lds = hay + (nh + 8u) * 8u + 64u;
        if (lds > kSmallLds) continue;
        Y.chunk = chunk;
        Y.max_items = std::max(items, 2u);
        Y.hay_off = hay;
        Y.lds = lds;
        return true;
    }
    return false;
}

template <int MODE>
__global__ __launch_bounds__(kSmallThreads) void k_match_small_chain(DevAutomaton A, const uint64_t *offsets, const uint8_t *blob, uint32_t nh, uint32_t find_all,
                                                                     uint32_t max_pattern_length, uint64_t *out_summary, uint64_t *out_prefix, uint4 *out_records,
                                                                     uint32_t cap, SmallChainLayout Y, uint64_t seq, uint32_t n_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t kChunk = Y.chunk, kItems = Y.max_items;
    const uint32_t img_bytes = (A.chain_img_bytes + 15u) & ~15u;
    uint16_t *cls = reinterpret_cast<uint16_t *>(smem + img_bytes);   // 512 B (TABLE mode)
    uint8_t *in_bytes = smem + Y.in_off;
    uint32_t *ev0 = reinterpret_cast<uint32_t *>(smem + Y.items_off);   // per item: its first event {index | byte of the item << 15}
    uint32_t *cnt_ev = ev0 + kItems;
    uint32_t *cnt_rec = cnt_ev + kItems;
    uint32_t *item_base = reinterpret_cast<uint32_t *>(smem + Y.hay_off);
    uint32_t *hay_off = item_base + nh + 8;
    uint32_t *wave_tmp = hay_off + nh + 8;
    const uint32_t t = threadIdx.x;
    const uint64_t t_start = wall_clock64();
    const bool stamps = (find_all & 2u) != 0u;
    find_all &= 1u;
    auto stamp = [&](int i) { if (stamps && t == 0) out_summary[4 + i] = wall_clock64() - t_start; };
    // Offsets and text lie in the host's pinned memory: a read is a trip over PCIe, ~ 2 us.  The text's length comes with the launch
    // (n_bytes), so that its first 16 KiB, every thread's offset and the first offset are REQUESTED TOGETHER, the image's loads
    // (device memory) behind them: one trip where the text used to wait for the offsets' (staging 4.4 us of a 15 us kernel).
    const uint32_t n16 = (n_bytes + 15u) / 16u;
    const uint4 *src = reinterpret_cast<const uint4 *>(blob);
    const uint64_t off0 = offsets[0];
    const uint64_t off_t = t <= nh ? offsets[t] : 0ull;
    const uint4 text_t = t < n16 ? src[t] : make_uint4(0u, 0u, 0u, 0u);
    {
        const uint4 *isrc = reinterpret_cast<const uint4 *>(A.chain_img);   // (the allocation has 16 bytes of slack)
        uint4 *idst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = t; i < (img_bytes >> 4); i += kSmallThreads) idst[i] = isrc[i];
        if (MODE == 0 && t < 256) cls[t] = A.class_off[t];
        uint4 *dst = reinterpret_cast<uint4 *>(in_bytes);
        if (t < n16) dst[t] = text_t;
        for (uint32_t i = t + kSmallThreads; i < n16; i += kSmallThreads) dst[i] = src[i];
        if (t <= nh) hay_off[t] = uint32_t(off_t - off0);
        for (uint32_t h = t + kSmallThreads; h <= nh; h += kSmallThreads) hay_off[h] = uint32_t(offsets[h] - off0);
    }
    __syncthreads();
    for (uint32_t h = t; h < nh + 1u; h += kSmallThreads)
        item_base[h] = h < nh ? (hay_off[h + 1] - hay_off[h] + kChunk - 1) / kChunk : 0u;
    __syncthreads();
    stamp(0);
    const uint32_t n_items = small_scan(item_base, nh + 1, wave_tmp);
    stamp(1);
    const uint32_t n_dense = A.chain_dense, rec_base = A.chain_rec_off - 4u * A.chain_dense, row_bytes = A.n_columns * 2u, cnt_n = A.chain_cnt_n;
    const uint8_t *cnt = smem + A.chain_cnt_off;
    // one transition (k_chain.hip chain_step2, one walk): the state's record, then the row entry of the state itself (a D state) or of
    // its failure state; `u` negative: the new state is final
    auto step = [&](uint32_t &e, uint32_t byte, uint32_t &u, bool &hit) {
        const uint32_t coff = class_offset<uint16_t, MODE>(A, cls, byte);
        const uint32_t r = *reinterpret_cast<const uint32_t *>(smem + ((e << 2) + rec_base));
        const bool is_d = e < n_dense;
        const uint32_t f = is_d ? e : (r & 0xFFFFu);
        const uint32_t tt = uint32_t(int32_t(*reinterpret_cast<const int16_t *>(smem + (f * row_bytes + coff))));
        hit = !is_d && ((r >> 16) & 0xFFu) == coff;
        u = hit ? r : tt;
        e = hit ? e + 1u : (tt & 0x7FFFu);
    };
    auto records_of = [&](uint32_t idx, uint32_t u, bool hit) -> uint32_t {
        const uint32_t code = hit ? (u >> 24) & 31u : 31u;
        if (code != 31u) return code;
        if (idx < cnt_n) {
            const uint32_t c = cnt[idx];
            if (c != 255u) return c;
        }
        const uint32_t c = A.chain_head[idx].x >> 24;
        return c < 255u ? c : A.out_count[A.chain_orig[idx]];
    };
    auto decode = [&](uint32_t i, uint32_t &h, uint32_t &begin, uint32_t &len, uint32_t &warm, uint32_t &pos0) {
        uint32_t lo = 0, hi = nh;  // item_base[lo] <= i < item_base[hi]
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (item_base[mid] <= i) lo = mid; else hi = mid;
        }
        h = lo;
        pos0 = (i - item_base[h]) * kChunk;
        begin = hay_off[h] + pos0;
        const uint32_t rest = hay_off[h + 1] - begin;
        len = rest < kChunk ? rest : kChunk;
        const uint32_t wmax = max_pattern_length ? max_pattern_length - 1u : 0u;
        warm = pos0 < wmax ? pos0 : wmax;
    };
    // ---- walk: count events and records per item, keep the first event ----
    for (uint32_t i = t; i < kItems; i += kSmallThreads) {
        uint32_t n_ev = 0, n_rec = 0;
        if (i < n_items) {
            uint32_t h, begin, len, warm, pos0;
            decode(i, h, begin, len, warm, pos0);
            uint32_t e = 0;
            const uint8_t *p = in_bytes + begin - warm;
            const uint32_t n = warm + len;
            auto one = [&](uint32_t byte, uint32_t j) {
                uint32_t u;
                bool hit;
                step(e, byte, u, hit);
                if (j >= warm && int32_t(u) < 0) {
                    n_rec += records_of(e, u, hit);
                    if (n_ev == 0u) ev0[i] = e | (j - warm) << 15;
                    ++n_ev;
                }
            };
            uint32_t j = 0;
            for (; j + 4u <= n; j += 4u) {
                const uint32_t b0 = p[j], b1 = p[j + 1u], b2 = p[j + 2u], b3 = p[j + 3u];
                one(b0, j); one(b1, j + 1u); one(b2, j + 2u); one(b3, j + 3u);
            }
            for (; j < n; ++j) one(p[j], j);
        }
        cnt_ev[i] = n_ev;
        cnt_rec[i] = n_rec;
    }
    __syncthreads();
    stamp(2);
    // ---- findAll == false: only the first event of each haystack (php_ahocorasick.c:588) ----
    if (!find_all) {
        for (uint32_t h = t; h < nh; h += kSmallThreads) {
            bool found = false;
            for (uint32_t i = item_base[h]; i < item_base[h + 1]; ++i) {
                if (found) { cnt_ev[i] = 0; cnt_rec[i] = 0; }
                else if (cnt_ev[i]) {
                    found = true;
                    cnt_ev[i] = 1;
                    const uint32_t idx = ev0[i] & 0x7FFFu;
                    const uint32_t c = A.chain_head[idx].x >> 24;
                    cnt_rec[i] = c < 255u ? c : A.out_count[A.chain_orig[idx]];
                }
            }
        }
        __syncthreads();
    }
    const uint32_t total = small_scan(cnt_rec, kItems, wave_tmp);  // cnt_rec[i] = first record of item i
    for (uint32_t h = t; h <= nh; h += kSmallThreads)
        out_prefix[h] = (h < nh && item_base[h] < kItems) ? cnt_rec[item_base[h]] : total;
    if (t == 0) { out_summary[0] = total; out_summary[1] = n_items; out_summary[2] = total > cap ? 1u : 0u; }
    stamp(3);
    if (total > cap) {
        __threadfence_system();
        __syncthreads();
        if (t == 0) __hip_atomic_store(out_summary + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // ---- records ----
    for (uint32_t i = t; i < n_items; i += kSmallThreads) {
        const uint32_t n_ev = cnt_ev[i];
        if (n_ev == 0) continue;
        uint32_t h, begin, len, warm, pos0;
        decode(i, h, begin, len, warm, pos0);
        uint64_t w = cnt_rec[i];
        if (n_ev == 1u) {
            // (an event of one record -- most -- from the index's own output head: one load, not state id -> head)
            const uint32_t ent = ev0[i], idx = ent & 0x7FFFu, pos = pos0 + (ent >> 15) + 1u;
            const uint2 hd = A.chain_head[idx];
            if ((hd.x >> 24) == 1u) out_records[w] = make_uint4(h, pos, pos - hd.y, hd.x & 0xFFFFFFu);
            else (void)emit_event(A, out_records, w, h, pos, A.chain_orig[idx]);
        } else {  // more than one event: the item is walked again, writing records straight to their slots
            uint32_t e = 0;
            const uint8_t *p = in_bytes + begin - warm;
            for (uint32_t j = 0; j < warm + len; ++j) {
                uint32_t u;
                bool hit;
                step(e, p[j], u, hit);
                if (j >= warm && int32_t(u) < 0) w = emit_event(A, out_records, w, h, pos0 + (j - warm) + 1u, A.chain_orig[e]);
            }
        }
    }
    stamp(4);
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(out_summary + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool small_trace() {
    static const bool on = std::getenv("ACMI_SMALL_TRACE") != nullptr;
    return on;
}

hipError_t launch_match_small(const DevAutomaton &A, const uint64_t *offsets, const uint8_t *blob, uint32_t nh, int find_all,
                              uint32_t max_pattern_length, int entry_bytes, int class_mode, uint64_t *out_summary,
                              uint64_t *out_prefix, void *out_records, uint32_t cap, uint32_t span_bytes, uint64_t seq, hipStream_t st) {
    static const bool no_chain = [] { const char *e = std::getenv("ACMI_SMALL_CHAIN"); return e && e[0] == '0'; }();
    SmallChainLayout YC{};
    // the whole automaton in LDS when the set carries the chain-compressed image and the batch fits beside it (one 8 KiB haystack does)
    if (!no_chain && A.trap + 1u < A.n_states && small_chain_layout(A, span_bytes, nh, class_mode == 0, YC)) {
        auto goc = [&](auto fn) {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), kSmallLds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(fn, dim3(1), dim3(kSmallThreads), YC.lds, st, A, offsets, blob, nh, uint32_t(find_all ? 1 : 0) | (small_trace() ? 2u : 0u),
                               max_pattern_length, out_summary, out_prefix, static_cast<uint4 *>(out_records), cap, YC, seq, span_bytes);
            return hipGetLastError();
        };
        return class_mode == 1 ? goc(k_match_small_chain<1>) : goc(k_match_small_chain<0>);
    }
    const SmallLayout Y = small_layout(span_bytes, nh, A.trap * A.row_bytes, class_mode == 0);
    const uint32_t lds = Y.in_bytes + Y.tile_bytes + Y.max_items * (kSmallEvents * 8u + 8u) + (kSmallMaxHay + 8u) * 8u + 64u + 512u;
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), kSmallLds);
        if (e != hipSuccess) return e;
        const bool trace = small_trace();
        hipLaunchKernelGGL(fn, dim3(1), dim3(kSmallThreads), lds, st, A, offsets, blob, nh, uint32_t(find_all ? 1 : 0) | (trace ? 2u : 0u),
                           max_pattern_length, out_summary, out_prefix, static_cast<uint4 *>(out_records), cap, Y, seq, span_bytes);
        return hipGetLastError();
    };
    if (entry_bytes == 2) return class_mode == 1 ? go(k_match_small<uint16_t, 1>) : go(k_match_small<uint16_t, 0>);
    return class_mode == 1 ? go(k_match_small<uint32_t, 1>) : go(k_match_small<uint32_t, 0>);
}

}  // namespace acmi
