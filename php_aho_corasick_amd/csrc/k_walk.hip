// k_walk.hip -- the exact walks behind the filter: k_walk_runs and the dense form's decision kernels (k_walk_items: k_walk_items.hip)
#include <cstdlib>

#include "kdev.hpp"

namespace acmi {

// ------------------------------------------------------------------------------------------
// k_walk_runs: exact walk of the byte runs k_scan_anchor flagged, one thread per work item.
//
// The anchor kernel used to do this itself, lane k walking item k after the wave had streamed its 64
// items: a chain of ~26 dependent memory round trips (run list, haystack dwords, 11 cold table rows)
// during which the wave streams nothing -- 0.15 ms of a 0.97 ms kernel for one hit per 8 KiB.  As a
// kernel of its own the same walks run at full occupancy, all 2^19 of them at once, and the latency
// chain is paid once instead of once per wave batch.  The first rows of the transition tile are
// staged in LDS (same image as k_scan_lds uses) so that only deep states go to L2.  Items without runs
// (run_count 0) cost one coalesced 4-byte load.
// ------------------------------------------------------------------------------------------
// (k_walk_runs keeps 79 registers, so a CU holds ONE of its 16-wave workgroups.  Measured alternatives on config 3 /
// the headline batch: 64 registers forced (24 spilled) 0.183 / 0.077 ms, three 8-wave workgroups with 48 KiB tiles
// 0.160 / 0.084 ms, against 0.163 / 0.077 ms as it is: occupancy is not what bounds the kernel.)

// Items whose run list overflowed are walked by the WAVE, a 64th of the item per lane (entered from the root Lmax-1 bytes
// early, like every chunk), and only COUNTED: events and records go to the item's counters, the item to the overflow list,
// and the drain (k_refill_wave) writes its records as it does for any item with more events than its slab holds.  One
// lane walking its 8 KiB alone took 2.4 ms -- the whole kernel, for ONE such item in a batch of any size (tools/exp_hits.py:
// six matches per haystack and one accidental candidate more somewhere: 0.075 -> 2.4 ms).  `whole`: this lane's item is
// one of them (`rec` its packed record, `item` its index); must be reached by every lane of the wave.
template <typename E, int MODE>
__device__ __forceinline__ void wave_walk_whole(const DevAutomaton &A, const BatchArgs &B, const uint8_t *tile, uint32_t tile_rows,
                                                bool whole, const uint4 &rec, uint64_t item, uint32_t &walked, uint32_t &fallback16) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t todo = __ballot(whole); todo != 0; todo &= todo - 1) {
        const uint32_t l = uint32_t(__builtin_ctzll(todo));
        const uint64_t item_l = item - lane + l;
        ItemDesc it;
        unpack_item(uint32_t(__builtin_amdgcn_readlane(int(rec.x), int(l))), uint32_t(__builtin_amdgcn_readlane(int(rec.y), int(l))),
                    uint32_t(__builtin_amdgcn_readlane(int(rec.z), int(l))), uint32_t(__builtin_amdgcn_readlane(int(rec.w), int(l))), it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        const uint32_t seg = (it.len + 63u) >> 6;
        const uint32_t s0 = it.warm + lane * seg, s1 = s0 + seg < n ? s0 + seg : n;
        EventRegs ev{0, 0};
        if (s0 < s1) {
            uint32_t from = s0 > B.warm_bytes ? s0 - B.warm_bytes : 0u;
            if (from < it.floor) from = it.floor;
            exact_range<E, MODE, false>(A, B, nullptr, tile, tile_rows, item_l, p, from, s1, s0, it.pos0 - it.warm + 1u, 0u, ev, 0xFFFFFFFFu, false);
        }
        uint32_t n_ev = ev.n_ev, rec_lo = uint32_t(ev.n_rec), rec_hi = uint32_t(ev.n_rec >> 32);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            n_ev += __shfl_xor(n_ev, d, 64);
            const uint32_t lo2 = __shfl_xor(rec_lo, d, 64), hi2 = __shfl_xor(rec_hi, d, 64);
            const uint64_t sum = (uint64_t(rec_hi) << 32 | rec_lo) + (uint64_t(hi2) << 32 | lo2);
            rec_lo = uint32_t(sum);
            rec_hi = uint32_t(sum >> 32);
        }
        if (lane == 0) {
            if (n_ev) {
                B.ev_count[item_l] = n_ev > B.slab_events ? n_ev : B.slab_events + 1u;   // (its events are not in the slab: the drain's)
                B.rec_count[item_l] = uint64_t(rec_hi) << 32 | rec_lo;
                B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item_l);
            }
            walked += n;
            fallback16 += (n + 15u) >> 4;
        }
    }
}

template <typename E, int MODE, bool EXTRA>
__global__ __launch_bounds__(kWalkThreads) void k_walk_runs(DevAutomaton A, BatchArgs B, uint32_t tile_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t total_items = B.item_base[B.n_haystacks];
    if (uint64_t(blockIdx.x) * blockDim.x >= total_items) return;
    __shared__ uint32_t s_walked, s_done, s_fallback, s_busy;
    if (threadIdx.x == 0) { s_walked = 0; s_done = 0; s_fallback = 0; s_busy = 0; }
    const uint64_t item = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    // issue the item's own loads first: they are in flight while the tile is staged
    uint32_t nr = 0, entry0 = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (item < total_items) {
        nr = B.run_count[item];
        // (batches of tiny haystacks -- 4 M items a GiB, one in thirty with a candidate -- ask for the 16-byte record and the list's first
        // entry only where there is a list: 84 of the kernel's 130 MB)
        if (!B.tiny_items || nr != 0u) {
            rec = B.item_rec[item];
            entry0 = reinterpret_cast<const uint32_t *>(B.runs)[win_slot(B, item, 0u)];   // (valid memory whatever the list holds; used when nr != 0)
        }
        if (B.zero_counters) {   // (the table is the batch before's: so are these, until an item with events overwrites them below)
            B.ev_count[item] = 0u;
            B.rec_count[item] = 0ull;
        }
    }
    const uint32_t tile_rows = A.row_bytes ? (tile_bytes / A.row_bytes < A.trap ? tile_bytes / A.row_bytes : A.trap) : 0u;
    if (tile_bytes != 0u) {   // (0: the handle's candidates are decided by text compares -- a walk, should one be needed, reads its rows through the L2)
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (tile_rows * A.row_bytes + 15u) >> 4;  // <= 4096: at most 4 per thread, loaded back to back
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = threadIdx.x + uint32_t(u) * kWalkThreads;
            v[u] = tsrc[t < n16 ? t : 0u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = threadIdx.x + uint32_t(u) * kWalkThreads;
            if (t < n16) tdst[t] = v[u];
        }
    }
    __syncthreads();
    uint32_t walked = 0, fallback16 = 0;
    bool whole = false, busy = false;
    if (nr != 0) {
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        // (a run list that overflowed -- more candidate windows than it holds -- means the whole item: not by this lane alone,
        // see below; findAll = false keeps the old way, the first event needs the order)
        // (... and an item with more than a few windows is a latency chain per window for its one lane -- eight windows per 8 KiB item,
        // and a batch of 8,192 items took 0.21 ms instead of 0.08: such items are k_runs_wave's, a wave per item and a lane per
        // window, launched behind this kernel once a batch has reported some: counters[20])
        if (nr > B.run_cap && B.relist) { /* left undone: the batch runs again with longer lists (run_batch_end) */ }
        else if (nr > B.run_cap && !B.first_only) whole = true;
        else if (nr > B.wave_min && B.wave_runs && !B.first_only) busy = true;
        else {
            busy = nr > B.wave_min;
            walked = run_epilogue_fast<E, MODE, EXTRA>(A, B, smem, tile_rows, item, it, nr, fallback16, entry0, true);
        }
    }
    wave_walk_whole<E, MODE>(A, B, smem, tile_rows, whole, rec, item, walked, fallback16);
    // (armed: the busy items' indices into k_runs_wave's list, one atomic per wave that has any; else they are only counted, per workgroup)
    uint32_t n_busy = 0;
    {
        const uint64_t bm = __ballot(busy);
        if (bm != 0) {
            n_busy = uint32_t(__builtin_popcountll(bm));
            if (B.wave_runs) {
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0) base = atomicAdd(&B.counters[20], n_busy);
                base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
                if (busy) B.busy_list[base + __builtin_amdgcn_mbcnt_hi(uint32_t(bm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bm), 0u))] = uint32_t(item);
                n_busy = 0;
            }
        }
    }
    // statistic: bytes walked, one device atomic per WORKGROUP (8,192 same-address device atomics took 0.1 ms on their own,
    // and per-wave atomics spread over eight words of one line doubled this kernel's time), handed in by the workgroup's
    // LAST wave to finish -- no barrier: a barrier made every wave of a workgroup wait for its slowest lane (a cluster that
    // has to be WALKED takes ~ 100 us) and kept the CU from starting the next workgroup; with one such item in a thousand
    // almost every workgroup had one (0.66 ms instead of 0.16 on such a batch; 0.39 without the barrier).
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        walked += __shfl_xor(walked, d, 64);
        fallback16 += __shfl_xor(fallback16, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (walked) atomicAdd(&s_walked, (walked + 15u) >> 4);
        if (fallback16) atomicAdd(&s_fallback, fallback16);
        if (n_busy) atomicAdd(&s_busy, n_busy);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (atomicAdd(&s_done, 1u) == blockDim.x / 64u - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t total = atomicAdd(&s_walked, 0u), fb = atomicAdd(&s_fallback, 0u), nb = atomicAdd(&s_busy, 0u);
            if (total) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], total);
            if (fb) atomicAdd(&B.counters[7], fb);
            if (nb) atomicAdd(&B.counters[20], nb);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_runs_wave: the items of a batch that have MANY candidate windows, a wave per item and a lane per window.
//
// k_walk_runs gives an item one lane, and a window costs that lane a chain of four or five dependent memory round trips
// (window text, fingerprint bucket, slot, pair record, the two texts): fine for the one or two windows of a sparse batch's item,
// a millisecond for an item of the 65,536-pattern set with 64 planted occurrences, and -- once a list overflowed -- the whole
// item walked through tables in global memory (round 5: 31-44 ms per 64 MiB forced through the filter at 32 occurrences per
// 8 KiB).  Here every window of an item is decided at once: lane r verifies window r (verify_candidate: the text compared
// with the patterns its fingerprint slot names), the occurrences found are compacted into the wave's list in LDS, and the
// wave ORDERS them -- an occurrence's rank among the distinct ends is its slot in the item's event slab; of several that end
// at the same byte the one with the most records stays (the longest pattern: its output list carries the others,
// src/multifast/node.c:424-441).  No cluster logic is needed: with every window of the item decided, the events are the
// occurrences' ends in ascending order (src/multifast/ahocorasick.c:214-231 reports in text order).  An item that holds
// something this cannot decide -- an unaligned head or tail entry, a fingerprint slot with too many pairs, more than
// kWaveOccMax occurrences -- goes through run_epilogue_fast on the wave's first lane, as k_walk_runs would have done it.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kWaveThreads = 256;   // (four waves: a workgroup lives as long as its slowest wave)

// occ_max: entries of a wave's occurrence list (LDS: {window, end, state, records} x occ_max per wave), a power of two >= the longest
// window list of the batch, 256 .. kWaveOccMax
template <typename E, int MODE, bool EXTRA>
__global__ __launch_bounds__(kWaveThreads) void k_runs_wave(DevAutomaton A, BatchArgs B, uint32_t occ_max) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *const ow = reinterpret_cast<uint32_t *>(smem) + wave * (4u * occ_max), *const oj = ow + occ_max, *const os = oj + occ_max, *const oc = os + occ_max;
    const uint32_t n_list = B.counters[20];   // (k_walk_runs' list of the items it left to this kernel)
    const uint32_t *starts = reinterpret_cast<const uint32_t *>(B.runs);
    const uint32_t l_max = A.max_pattern_length;
    uint32_t walked = 0, fallback16 = 0, serial = 0;
#pragma unroll 1
    for (uint32_t w = blockIdx.x * (kWaveThreads / 64u) + wave; w < n_list; w += gridDim.x * (kWaveThreads / 64u)) {
        const uint64_t item = uint32_t(__builtin_amdgcn_readfirstlane(int(B.busy_list[w])));
        const uint32_t nr = uint32_t(__builtin_amdgcn_readfirstlane(int(B.run_count[item])));
        const uint4 rec = B.item_rec[item];
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        const uint32_t pos_bias = it.pos0 - it.warm + 1u;
        // ---- every window decided, a lane each; the occurrences into the wave's list, in window order ----
        uint32_t M = 0;
        bool ok = A.anchor_pairs != nullptr;
#pragma unroll 1
        for (uint32_t c = 0; c < nr && ok; c += 64u) {
            const uint32_t r = c + lane;
            const bool active = r < nr;
            const uint32_t e = active ? starts[win_slot(B, item, r)] : 0u;
            Verified v[kClusterMax];
#pragma unroll
            for (uint32_t i = 0; i < kClusterMax; ++i) v[i] = Verified{0xFFFFFFFFu, 0u, 0u};
            uint32_t nv = 0;
            bool decided = (e & kWinSpecial) == 0u;
            if (active && decided) decided = verify_candidate(A, p, n, it.floor, e, v, nv);
            if (__any(!decided)) { ok = false; break; }
            const uint32_t incl = wave_inclusive_sum(nv);
            const uint32_t tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (M + tot > occ_max) { ok = false; break; }
            const uint32_t at = M + incl - nv;
#pragma unroll
            for (uint32_t i = 0; i < kClusterMax; ++i) {
                if (i < nv) {
                    ow[at + i] = e;
                    // (ends in the warm-up are the item before's: they take no part in the order)
                    oj[at + i] = v[i].j >= it.warm ? v[i].j : 0xFFFFFFFFu;
                    os[at + i] = v[i].state;
                    oc[at + i] = v[i].cnt;
                }
            }
            M += tot;
        }
        if (!ok) {
            if (lane == 0) {
                walked += run_epilogue_fast<E, MODE, EXTRA>(A, B, smem, 0u, item, it, nr, fallback16);
                ++serial;
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (M == 0u) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- the order.  The list is in window order, and an occurrence anchored at window w ends in [w + 3, w + Lmax): only
        // entries whose windows lie less than Lmax bytes apart can end in the other order or at the same byte.  So an entry looks
        // at its NEIGHBOURHOOD only (a few entries each way, whatever the list's length): first the duplicates of an end go -- all
        // but the one with the most records, the earlier of equals --, then an entry's rank is its index among the kept entries,
        // less the kept neighbours before it that end later, plus those behind it that end earlier.
        const uint32_t rounds = (M + 63u) >> 6;
        uint32_t keep_bits = 0;   // bit t: entry lane + 64 t is an event
#pragma unroll 1
        for (uint32_t t = 0; t < rounds; ++t) {
            const uint32_t idx = lane + 64u * t;
            if (idx < M) {
                const uint32_t wj = ow[idx], j = oj[idx], cnt = oc[idx];
                bool keep = j != 0xFFFFFFFFu;
                for (uint32_t i = idx; keep && i > 0u && ow[i - 1u] + l_max > wj; --i)
                    if (oj[i - 1u] == j && oc[i - 1u] >= cnt) keep = false;
                for (uint32_t i = idx + 1u; keep && i < M && ow[i] < wj + l_max; ++i)
                    if (oj[i] == j && oc[i] > cnt) keep = false;
                keep_bits |= keep ? (1u << t) : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (uint32_t t = 0; t < rounds; ++t) {
            const uint32_t idx = lane + 64u * t;
            if (idx < M && ((keep_bits >> t) & 1u) == 0u) oj[idx] = 0xFFFFFFFFu;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t before = 0;      // kept entries of the rounds before (wave-uniform)
        uint32_t rec_lo = 0, rec_hi = 0;
#pragma unroll 1
        for (uint32_t t = 0; t < rounds; ++t) {
            const uint32_t idx = lane + 64u * t;
            const bool keep = ((keep_bits >> t) & 1u) != 0u;
            const uint64_t km = __ballot(keep);
            if (keep) {
                const uint32_t wj = ow[idx], j = oj[idx];
                uint32_t rank = before + __builtin_amdgcn_mbcnt_hi(uint32_t(km >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(km), 0u));
                for (uint32_t i = idx; i > 0u && ow[i - 1u] + l_max > wj; --i) {
                    const uint32_t ji = oj[i - 1u];
                    rank -= (ji != 0xFFFFFFFFu && ji > j) ? 1u : 0u;
                }
                for (uint32_t i = idx + 1u; i < M && ow[i] < wj + l_max; ++i) rank += oj[i] < j ? 1u : 0u;
                if (rank < B.slab_events) B.slab[slab_slot(B, item, rank)] = make_uint2(pos_bias + j, os[idx]);
                const uint32_t c = oc[idx];
                rec_hi += rec_lo + c < rec_lo ? 1u : 0u;
                rec_lo += c;
            }
            before += uint32_t(__builtin_popcountll(km));
        }
        const uint32_t n_ev = before;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo2 = __shfl_xor(rec_lo, d, 64), hi2 = __shfl_xor(rec_hi, d, 64);
            const uint64_t sum = (uint64_t(rec_hi) << 32 | rec_lo) + (uint64_t(hi2) << 32 | lo2);
            rec_lo = uint32_t(sum);
            rec_hi = uint32_t(sum >> 32);
        }
        if (lane == 0 && n_ev) {
            B.ev_count[item] = n_ev;
            B.rec_count[item] = uint64_t(rec_hi) << 32 | rec_lo;
            if (n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
        }
        __builtin_amdgcn_wave_barrier();   // (the list is the next item's from here on)
    }
    // statistics, once per wave (a wave that took the serial way for nothing adds nothing)
    if (lane == 0) {
        if (walked) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], (walked + 15u) >> 4);
        if (fallback16) atomicAdd(&B.counters[7], fallback16);
        if (serial) atomicAdd(&B.counters[22], serial);
    }
}

hipError_t launch_runs_wave(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode, hipStream_t st) {
    if (max_items == 0 || B.first_only) return hipSuccess;
    uint32_t occ_max = 256;
    while (occ_max < B.run_cap && occ_max < kWaveOccMax) occ_max *= 2;
    const uint32_t lds = (kWaveThreads / 64u) * 4u * occ_max * 4u;   // 16 KiB .. 64 KiB
    // (the list's length is on the device: a wave per item the batch could have, at most eight workgroups a CU)
    const uint32_t blocks = uint32_t(std::min<uint64_t>((max_items + kWaveThreads / 64u - 1u) / (kWaveThreads / 64u), uint64_t(num_cus()) * 8u));
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWaveThreads), lds, st, A, B, occ_max);
        return hipGetLastError();
    };
    const bool extra = A.early_stop != 0u;
#define ACMI_RUNS_WAVE(E, M) (extra ? go(k_runs_wave<E, M, true>) : go(k_runs_wave<E, M, false>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_RUNS_WAVE(uint16_t, 1) : ACMI_RUNS_WAVE(uint16_t, 0);
    return class_mode == 1 ? ACMI_RUNS_WAVE(uint32_t, 1) : ACMI_RUNS_WAVE(uint32_t, 0);
#undef ACMI_RUNS_WAVE
}

static hipError_t launch_verify_dense(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                                      hipStream_t st);

hipError_t launch_walk_runs(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                            hipStream_t st, bool stage_tile) {
    if (max_items == 0) return hipSuccess;
    if (B.run_dense == 1u) return launch_verify_dense(A, B, max_items, entry_bytes, class_mode, st);   // (2: window lists of the aligned forms, this kernel's)
    // (512- and 256-thread workgroups where no tile is staged -- three / six workgroups a CU instead of one of sixteen waves -- were
    // measured on the headline batch, round 6: 49.8 -> 51.5 / 54.0 us)
    const uint32_t threads = kWalkThreads;
    const uint32_t blocks = uint32_t((max_items + threads - 1) / threads);
    // No tile while the handle's candidates are all decided by text compares (the engine asks for it again as soon as a batch
    // walked a cluster: every fallback cluster through the L2 is not worth it): 64 KiB staged per 1,024 items were 52 -> 49.5 us on
    // the headline batch, 0.148 -> 0.141 ms on config 3 -- and most of the kernel for haystacks of 256 B, an item each.
    const uint32_t tile_bytes = stage_tile ? kWalkTileBytes : 0u;
    // (without a tile the workgroup still owns one row's worth of LDS: walk_fast reads row 0's column unconditionally and discards
    // the value -- the read must lie inside the allocation)
    const uint32_t lds = tile_bytes ? tile_bytes : std::max<uint32_t>(A.row_bytes, 16u);
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, st, A, B, tile_bytes);
        return hipGetLastError();
    };
    const bool extra = A.early_stop != 0u;
#define ACMI_WALK_RUNS(E, M) (extra ? go(k_walk_runs<E, M, true>) : go(k_walk_runs<E, M, false>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_WALK_RUNS(uint16_t, 1) : ACMI_WALK_RUNS(uint16_t, 0);
    return class_mode == 1 ? ACMI_WALK_RUNS(uint32_t, 1) : ACMI_WALK_RUNS(uint32_t, 0);
#undef ACMI_WALK_RUNS
}

// ------------------------------------------------------------------------------------------
// k_verify_dense: k_walk_runs for the dense anchor form (windows at every byte offset, shortest pattern 4..6 B).
//
// A work item has tens of hit windows there (a keyword list: ~ 100 per 8 KiB), and deciding each through the tables in
// global memory -- fingerprint bucket, slot, pair record, pattern text: five dependent round trips, ~ 15 divergent
// requests -- made this step three times the filter's own time.  For small sets (<= 4,096 patterns of <= 16 B) everything
// the decision needs is staged into LDS instead (automaton.hpp dense_image: a hash table keyed by the window itself, a
// record per pattern, the patterns' bytes from the fifth on), and a hit costs ONE global access, the 16 text bytes at the
// window, requested a hit ahead.  The run list is the 4-byte window starts of the item, contiguous (item-major); a run's
// bounds follow from its start: patterns BEGIN at their window, so everything anchored at `start` ends in
// [start + m_min - 1, start + Lmax).  Runs that touch or overlap form a cluster, decided as in run_epilogue_fast: its
// occurrences in position order, or -- more than six of them, a text load that would cross the item's end -- one exact
// walk from the cluster's first window.  Bigger sets verify through the global tables (VLDS = false).
// ------------------------------------------------------------------------------------------
template <bool VLDS>
__device__ __forceinline__ bool verify_dense(const DevAutomaton &A, const uint32_t *img, const uint8_t *p, uint32_t n, uint32_t floor,
                                             uint32_t start, const uint4 &t, Verified (&v)[kClusterMax], uint32_t &nv) {
    if (start + 4u > n) return true;   // no pattern fits
    if constexpr (!VLDS) {
        if (A.anchor_pairs == nullptr) return false;   // "verify" 0: every candidate is walked
        return verify_candidate(A, p, n, floor, start, v, nv);
    } else {
        if (start + 16u > n) return false;   // (the 16 text bytes were not loaded: the cluster is walked)
        const uint32_t mask = img[0];
        const uint32_t *ht = img + img[1], *recs = img + img[2], *tails = img + img[3];
        uint32_t slot = ((t.x * img[5]) >> 16) & mask;
        uint32_t e1;
        for (;;) {
            e1 = ht[2u * slot + 1u];
            if (e1 == 0xFFFFFFFFu) return true;   // not a key at all: a fingerprint twin
            if (ht[2u * slot] == t.x) break;
            slot = (slot + 1u) & mask;
        }
        const uint32_t first = e1 & 0xFFFFu, count = e1 >> 16;
#pragma unroll 1
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t r0 = recs[2u * (first + k)], r1 = recs[2u * (first + k) + 1u];
            const uint32_t len = r0 >> 24, state = r0 & 0xFFFFFFu, toff = r1 & 0xFFFFu;
            if (start < floor || start + len > n) continue;   // would begin before / end after this item's bytes
            const uint32_t tl = len - 4u;                      // 0..12 bytes behind the window
            const uint32_t m0 = tl >= 4u ? 0xFFFFFFFFu : (1u << (8u * tl)) - 1u;
            const uint32_t m1 = tl >= 8u ? 0xFFFFFFFFu : tl > 4u ? (1u << (8u * (tl - 4u))) - 1u : 0u;
            const uint32_t m2 = tl >= 12u ? 0xFFFFFFFFu : tl > 8u ? (1u << (8u * (tl - 8u))) - 1u : 0u;
            const uint32_t diff = ((t.y ^ tails[toff]) & m0) | ((t.z ^ tails[toff + 1u]) & m1) | ((t.w ^ tails[toff + 2u]) & m2);
            if (diff) continue;
            if (nv >= kClusterMax) return false;
            uint32_t oc = r1 >> 16;
            if (oc == 65535u) oc = A.out_count[state];
            const Verified vv{start + len - 1u, state, oc};
#pragma unroll
            for (uint32_t i = 0; i < kClusterMax; ++i) if (i == nv) v[i] = vv;
            ++nv;
        }
        return true;
    }
}

template <typename E, int MODE, bool VLDS>
__global__ __launch_bounds__(kWalkThreads) void k_verify_dense(DevAutomaton A, BatchArgs B) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t total_items = B.item_base[B.n_haystacks];
    if (uint64_t(blockIdx.x) * kWalkThreads >= total_items) return;
    __shared__ uint32_t s_walked, s_done, s_fallback;
    if (threadIdx.x == 0) { s_walked = 0; s_done = 0; s_fallback = 0; }
    const uint64_t item = uint64_t(blockIdx.x) * kWalkThreads + threadIdx.x;
    uint32_t nr = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (item < total_items) {
        nr = B.run_count[item];
        rec = B.item_rec[item];
    }
    if constexpr (VLDS) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.dense_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = A.dense_image_words >> 2;
        for (uint32_t t = threadIdx.x; t < n16; t += kWalkThreads) dst[t] = src[t];
    }
    __syncthreads();
    const uint32_t *img = reinterpret_cast<const uint32_t *>(smem);
    uint32_t walked = 0, fallback16 = 0;
    bool whole = false;
    if (nr != 0) {
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        const uint32_t pos_bias = it.pos0 - it.warm + 1u;
        const uint32_t m_min = A.min_pattern_length, l_max = A.max_pattern_length;
        EventRegs ev{0, 0};
        if (nr > B.run_cap && !B.first_only) {   // more hit windows than the list holds (or an unaligned head): the whole item,
            whole = true;                         // exactly -- by the wave, below (wave_walk_whole)
        } else if (nr > B.run_cap) {
            walked = n;
            exact_range<E, MODE, false>(A, B, nullptr, nullptr, 0u, item, p, it.floor, n, it.warm, pos_bias, 0u, ev);
            fallback16 += (n + 15u) >> 4;
        } else {
            const uint32_t *starts = reinterpret_cast<const uint32_t *>(B.runs) + item * uint64_t(B.run_cap);
            uint32_t r = 0;
            uint32_t cur = starts[0];
            // the 16 text bytes at a window, requested one window ahead (where they lie inside the item)
            auto text_at = [&](uint32_t st) -> uint4 { return (VLDS && st + 16u <= n) ? load16u(p + st) : make_uint4(0u, 0u, 0u, 0u); };
            uint4 t_cur = text_at(cur);
#pragma unroll 1
            while (r < nr) {
                const uint32_t first = cur;
                uint32_t hi = first + l_max < n ? first + l_max : n, nv = 0;
                Verified v[kClusterMax];
#pragma unroll
                for (uint32_t i = 0; i < kClusterMax; ++i) v[i] = Verified{0xFFFFFFFFu, 0u, 0u};
                bool windows = true;
#pragma unroll 1
                for (;;) {
                    const uint32_t st = cur;
                    const uint4 t = t_cur;
                    ++r;
                    if (r < nr) {   // (the next window's text is in flight while this one is decided)
                        cur = starts[r];
                        t_cur = text_at(cur);
                    }
                    if (windows) windows = verify_dense<VLDS>(A, img, p, n, it.floor, st, t, v, nv);
                    if (r >= nr || cur + m_min - 1u > hi) break;   // neither adjacent nor overlapping: the next cluster
                    const uint32_t h2 = cur + l_max < n ? cur + l_max : n;
                    hi = h2 > hi ? h2 : hi;
                }
                const uint32_t lo = first + m_min - 1u;
                if (lo >= n) continue;
                const uint32_t rep_from = lo > it.warm ? lo : it.warm;
                walked += hi - lo;
                if (windows) {
                    emit_verified(B, item, v, rep_from, hi, pos_bias, ev);
                    continue;
                }
                // every pattern that ends in the cluster begins at one of its windows, the first being the leftmost
                const uint32_t from = first > it.floor ? first : it.floor;
                exact_range<E, MODE, false>(A, B, nullptr, nullptr, 0u, item, p, from, hi, rep_from, pos_bias, 0u, ev);
                fallback16 += (hi - from + 15u) >> 4;
            }
        }
        if (ev.n_ev) {
            B.ev_count[item] = ev.n_ev;
            B.rec_count[item] = ev.n_rec;
            if (ev.n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
        }
    }
    wave_walk_whole<E, MODE>(A, B, nullptr, 0u, whole, rec, item, walked, fallback16);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        walked += __shfl_xor(walked, d, 64);
        fallback16 += __shfl_xor(fallback16, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (walked) atomicAdd(&s_walked, (walked + 15u) >> 4);
        if (fallback16) atomicAdd(&s_fallback, fallback16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (atomicAdd(&s_done, 1u) == kWalkThreads / 64u - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t total = atomicAdd(&s_walked, 0u), fb = atomicAdd(&s_fallback, 0u);
            if (total) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], total);
            if (fb) atomicAdd(&B.counters[7], fb);
        }
    }
}

// With the LDS image the decision is split in two kernels, because what bounds it is memory LATENCY, not work: one thread per
// item deciding its ~ 100 windows one after the other waits for a window-list load and a text load per window, a few
// microseconds each (0.89 ms per GiB of keyword text, more than the filter), and a per-item pass over ALL windows reads its
// 4-byte entries a 128-byte line at a time (4 GB of L2 misses per GiB of text).
//   k_dense_decide   one WAVE per item, 64 hit windows per step: the window list in one coalesced read, the text loads (they
//                    fall into the item's own few lines), the LDS lookup, and the windows that hold an occurrence compacted
//                    in place -- with the record index of the occurrence, or kDenseMany (several patterns with that prefix
//                    are there, or the text load would cross the item's end);
//   k_dense_emit     one thread per item over the FEW entries left (nine hit windows in ten hold nothing): clusters those
//                    whose ranges touch, emits the events in position order; only kDenseMany windows are looked at again.
// What is left of the cost is the text itself: the decide kernel reads every line of it again (TCC_MISS x 128 B = the
// batch), 0.29 ms per GiB -- the price of deciding in a kernel of its own instead of inside the scan.
constexpr uint32_t kDenseNone = 0xFFFFFFFFu, kDenseMany = 0xFFFFFFFEu;

// One WAVE per item, 64 hit windows per step: the window starts in one coalesced read, the text loads (they fall into the
// item's own few lines), the LDS lookup, and the windows that hold an occurrence COMPACTED (ballot + mbcnt, running base):
// found[item * Q + i] = start of the i-th such window (written over the window list: i never exceeds the index read) and
// its record index -- or kDenseMany -- behind the lists; run_count[item] becomes the number found.  Most hit windows hold
// nothing (a keyword list: nine in ten are a word's first four letters and no more), so k_dense_emit is left with a
// handful of entries per item.
// IMG = false (sets the LDS image does not hold: more than 4,096 patterns, patterns over 16 B): the same pass as a FILTER --
// a window is kept when the text compare through the global tables (verify_candidate) finds an occurrence or cannot decide,
// marked kDenseMany, and k_dense_emit decides the few that are left again.  One thread per item deciding its hundreds of
// windows one after the other (k_verify_dense) took 1.1 ms per GiB for 8,192 words.
template <bool IMG>
__global__ __launch_bounds__(1024) void k_dense_decide(DevAutomaton A, BatchArgs B, uint32_t *found_idx) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t total_items = B.item_base[B.n_haystacks];
    const uint32_t Q = B.run_cap;
    if (uint64_t(blockIdx.x) * 16u >= total_items) return;
    if constexpr (IMG) {   // the image is staged once per workgroup; the workgroups are persistent (a few per CU) and stride over the items
        const uint4 *src = reinterpret_cast<const uint4 *>(A.dense_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = A.dense_image_words >> 2;
        for (uint32_t t = threadIdx.x; t < n16; t += 1024u) dst[t] = src[t];
    }
    __syncthreads();
    const uint32_t *img = reinterpret_cast<const uint32_t *>(smem);
    const uint32_t mask = IMG ? img[0] : 0u;
    const uint32_t *ht = IMG ? img + img[1] : nullptr, *recs = IMG ? img + img[2] : nullptr, *tails = IMG ? img + img[3] : nullptr;
    const uint2 *recs2 = reinterpret_cast<const uint2 *>(recs);
    const uint32_t hmul = IMG ? img[5] : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blob_end = B.offsets[B.n_haystacks];
    uint32_t *starts_all = reinterpret_cast<uint32_t *>(B.runs);
    // IMG: the wave's own 64 {window start, record index} behind the image -- an item whose windows are all decided here and hold at
    // most 64 occurrences is finished by this wave (below), k_dense_emit finds nothing to do for it
    uint2 *kept = reinterpret_cast<uint2 *>(smem + ((IMG ? A.dense_image_words * 4u : 0u) + 15u & ~15u)) + (threadIdx.x >> 6) * 64u;
    for (uint64_t item = uint64_t(blockIdx.x) * 16u + (threadIdx.x >> 6); item < total_items; item += uint64_t(gridDim.x) * 16u) {
        const uint32_t nr = __builtin_amdgcn_readfirstlane(B.run_count[item]);
        if (nr == 0 || nr > Q) continue;   // (more windows than the list holds: k_dense_emit walks the whole item)
        const uint4 rec = B.item_rec[item];
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        const uint64_t room64 = blob_end - (it.begin - it.warm);   // bytes that can be read from p on
        const uint32_t room = room64 < 0xFFFFFFFFull ? uint32_t(room64) : 0xFFFFFFFFu;
        uint32_t *starts = starts_all + item * uint64_t(Q), *fidx = found_idx + item * uint64_t(Q);
        uint32_t base = 0;
        bool many = false;   // (wave-uniform) a window is left to k_dense_emit
#pragma unroll 1
        for (uint32_t c = 0; c < nr; c += 64u) {
            const uint32_t r = c + lane;
            const bool active = r < nr;
            const uint32_t start = active ? starts[r] : 0u;
            uint32_t out = kDenseNone;
            if constexpr (!IMG) {
                if (active) {
                    Verified v[kClusterMax];
                    uint32_t nv = 0;
                    const bool decided = verify_candidate(A, p, n, it.floor, start, v, nv);
                    if (!decided || nv != 0) out = kDenseMany;
                }
            } else {
                // (the 16 bytes at every active window at once, then the LDS look-ups: one pass without a branch for the usual window --
                // key found at its first slot or not there, one pattern with that prefix)
                const bool readable = active && start + 16u <= room;   // (not: the last bytes of the BLOB -- an item's own end is no bound, what lies behind it is masked)
                if (active && !readable) out = start + 4u <= n ? kDenseMany : kDenseNone;   // (decided by k_dense_emit, which reads single bytes)
                const uint4 tx = load16u(readable ? p + start : A.pad16);
                uint32_t slot = ((tx.x * hmul) >> 16) & mask;
                const uint2 *ht2 = reinterpret_cast<const uint2 *>(ht);
                uint2 e = ht2[slot];
                while (readable && e.y != 0xFFFFFFFFu && e.x != tx.x) {
                    slot = (slot + 1u) & mask;
                    e = ht2[slot];
                }
                if (readable && e.y != 0xFFFFFFFFu) {
                    const uint32_t first = e.y & 0xFFFFu, count = e.y >> 16;
#pragma unroll 1
                    for (uint32_t k = 0; k < count; ++k) {
                        const uint2 rr = recs2[first + k];   // (the records begin at an even word of the image: automaton.cpp)
                        const uint32_t r0 = rr.x, r1 = rr.y;
                        const uint32_t len = r0 >> 24, toff = r1 & 0xFFFFu;
                        const uint32_t tl = len - 4u;
                        // mask of the tail's bytes: tl of the 12 behind the window's four
                        const uint64_t lo64 = tl >= 8u ? ~0ull : (1ull << (8u * tl)) - 1ull;
                        const uint32_t m0 = uint32_t(lo64), m1 = uint32_t(lo64 >> 32);
                        const uint32_t m2 = tl > 8u ? 0xFFFFFFFFu >> (8u * (12u - tl)) : 0u;
                        const uint32_t diff = ((tx.y ^ tails[toff]) & m0) | ((tx.z ^ tails[toff + 1u]) & m1) | ((tx.w ^ tails[toff + 2u]) & m2);
                        const bool hit = diff == 0u && start >= it.floor && start + len <= n;
                        if (hit) out = out == kDenseNone ? first + k : kDenseMany;
                    }
                }
            }
            const bool keep = out != kDenseNone;
            const uint64_t km = __ballot(keep);
            if (km != 0) {
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi(uint32_t(km >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(km), 0u));
                many = many || __ballot(out == kDenseMany) != 0;
                // (an item of one step that this wave finishes itself, below, leaves nothing in the lists)
                const bool mine = IMG && nr <= 64u && !many && !B.first_only;
                if (keep) {
                    if (!mine) {
                        starts[at] = start;   // at <= r: every lane of this step has read its start, earlier steps' are consumed
                        fidx[at] = out;
                    }
                    if (IMG && at < 64u) kept[at] = make_uint2(start, out);
                }
                base += uint32_t(__builtin_popcountll(km));
            }
        }
        if constexpr (IMG) {
            // The item's events, by the wave: every occurrence is known (start, pattern record), none is left to be decided, so
            // there is nothing to cluster -- an occurrence that ends in the item's own range is an event, of several that end at
            // the same byte the one that starts first (the longest pattern: its output list carries the others), in order of
            // their ends (src/multifast/ahocorasick.c:214-231 reports in text order).  k_dense_emit did this a thread per item,
            // entry by entry: 77 us per GiB of the keyword set for what this wave, waiting for its text lines anyway, does in passing.
            if (base != 0u && base <= 64u && !many && !B.first_only) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool have = lane < base;
                const uint2 k = have ? kept[lane] : make_uint2(0u, 0u);
                const uint2 rr = have ? recs2[k.y] : make_uint2(0u, 0u);
                const uint32_t r0 = rr.x, r1 = rr.y;
                uint32_t oc = r1 >> 16;
                if (have && oc == 65535u) oc = A.out_count[r0 & 0xFFFFFFu];
                const uint32_t j = k.x + (r0 >> 24) - 1u;          // last byte of the occurrence
                const bool valid = have && j >= it.warm;          // (ends in the warm-up are the item before's)
                const uint64_t vm = __ballot(valid);
                // (the usual item: ends ascend with the starts -- every entry is an event of its own, in list order)
                const uint32_t j_left = uint32_t(__builtin_amdgcn_update_dpp(0, int(j), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                const bool ascending = __all(!have || lane == 0u || j > j_left) && __all(oc < (1u << 24));
                bool ev_mine = valid;
                uint64_t em = vm;
                uint32_t before, n_rec_lo, n_rec_hi = 0;
                if (ascending) {
                    before = __builtin_amdgcn_mbcnt_hi(uint32_t(vm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(vm), 0u));
                    if (__ballot(valid && oc != 1u) == 0) n_rec_lo = uint32_t(__builtin_popcountll(vm));   // (a record per event: the usual list)
                    else n_rec_lo = uint32_t(__builtin_amdgcn_readlane(int(wave_inclusive_sum(valid ? oc : 0u)), 63));
                } else {
                    bool dup = false;
                    for (uint64_t m = vm; m != 0; m &= m - 1) {        // (entries ascend by start: of equal ends the first is the longest)
                        const uint32_t l = uint32_t(__builtin_ctzll(m));
                        const uint32_t jl = uint32_t(__builtin_amdgcn_readlane(int(j), int(l)));
                        dup = dup || (l < lane && jl == j);
                    }
                    ev_mine = valid && !dup;
                    em = __ballot(ev_mine);
                    before = 0;
                    n_rec_lo = 0;
                    for (uint64_t m = em; m != 0; m &= m - 1) {
                        const uint32_t l = uint32_t(__builtin_ctzll(m));
                        const uint32_t jl = uint32_t(__builtin_amdgcn_readlane(int(j), int(l)));
                        before += jl < j ? 1u : 0u;
                        const uint32_t c = uint32_t(__builtin_amdgcn_readlane(int(oc), int(l)));
                        n_rec_hi += n_rec_lo + c < n_rec_lo ? 1u : 0u;
                        n_rec_lo += c;
                    }
                }
                const uint32_t n_ev = uint32_t(__builtin_popcountll(em));
                if (ev_mine && before < B.slab_events)
                    B.slab[slab_slot(B, item, before)] = make_uint2(it.pos0 - it.warm + 1u + j, r0 & 0xFFFFFFu);
                if (lane == 0) {
                    if (n_ev) {
                        B.ev_count[item] = n_ev;
                        B.rec_count[item] = uint64_t(n_rec_hi) << 32 | n_rec_lo;
                        if (n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
                    }
                    B.run_count[item] = 0u;
                }
                continue;
            }
        }
        if (lane == 0) B.run_count[item] = base;
    }
}

template <typename E, int MODE>
__global__ __launch_bounds__(256) void k_dense_emit(DevAutomaton A, BatchArgs B, const uint32_t *found_idx) {
    const uint64_t total_items = B.item_base[B.n_haystacks];
    const uint64_t item = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    __shared__ uint32_t s_walked, s_fallback;
    if (threadIdx.x == 0) { s_walked = 0; s_fallback = 0; }
    __syncthreads();
    uint32_t walked = 0, fallback16 = 0;
    bool whole = false;
    const uint32_t nf = item < total_items ? B.run_count[item] : 0u;   // occurrences found (> run_cap: the window list overflowed)
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (nf != 0) {
        rec = B.item_rec[item];
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        const uint32_t pos_bias = it.pos0 - it.warm + 1u;
        const uint32_t m_min = A.min_pattern_length, l_max = A.max_pattern_length;
        const uint32_t *img = A.dense_image;              // (records read through the cache: a few KiB, shared by everyone; none: every
        const uint32_t *recs = img ? img + img[2] : nullptr;   //  kept window is marked kDenseMany)
        EventRegs ev{0, 0};
        if (nf > B.run_cap && !B.first_only) {
            whole = true;   // (the wave walks it together, below: wave_walk_whole)
        } else if (nf > B.run_cap) {
            walked = n;
            exact_range<E, MODE, false>(A, B, nullptr, nullptr, 0u, item, p, it.floor, n, it.warm, pos_bias, 0u, ev);
            fallback16 += (n + 15u) >> 4;
        } else {
            const uint32_t *starts = reinterpret_cast<const uint32_t *>(B.runs) + item * uint64_t(B.run_cap);
            const uint32_t *fidx = found_idx + item * uint64_t(B.run_cap);
            // (entry r + 1 -- window start and verdict -- is requested before entry r is worked on: 89 -> 77 us per GiB of the keyword
            // set; the occurrence record one entry ahead as well measured the same)
            uint32_t r = 0;
            uint32_t st_n = starts[0], got_n = fidx[0];
#pragma unroll 1
            while (r < nf) {
                // a cluster: occurrences whose possible ranges [start + m_min - 1, start + Lmax) touch or overlap
                const uint32_t first = st_n;
                uint32_t hi = first + l_max < n ? first + l_max : n, nv = 0;
                Verified v[kClusterMax];
#pragma unroll
                for (uint32_t i = 0; i < kClusterMax; ++i) v[i] = Verified{0xFFFFFFFFu, 0u, 0u};
                bool windows = true;
                uint32_t st = first;
#pragma unroll 1
                for (;;) {
                    const uint32_t got = got_n;
                    if (r + 1u < nf) {
                        st_n = starts[r + 1u];
                        got_n = fidx[r + 1u];
                    }
                    if (got == kDenseMany) {
                        // several patterns with this prefix, or a window in the item's last bytes: decided here
                        if (windows) windows = A.anchor_pairs != nullptr && verify_candidate(A, p, n, it.floor, st, v, nv);
                    } else if (windows) {
                        if (nv >= kClusterMax) windows = false;
                        else {
                            const uint32_t r0 = recs[2u * got], r1 = recs[2u * got + 1u];
                            uint32_t oc = r1 >> 16;
                            if (oc == 65535u) oc = A.out_count[r0 & 0xFFFFFFu];
                            const Verified vv{st + (r0 >> 24) - 1u, r0 & 0xFFFFFFu, oc};
#pragma unroll
                            for (uint32_t i = 0; i < kClusterMax; ++i) if (i == nv) v[i] = vv;
                            ++nv;
                        }
                    }
                    ++r;
                    if (r >= nf) break;
                    st = st_n;
                    if (st + m_min - 1u > hi) break;   // neither adjacent nor overlapping: the next cluster
                    const uint32_t h2 = st + l_max < n ? st + l_max : n;
                    hi = h2 > hi ? h2 : hi;
                }
                const uint32_t lo = first + m_min - 1u;
                if (lo >= n) continue;
                const uint32_t rep_from = lo > it.warm ? lo : it.warm;
                walked += hi - lo;
                if (windows) {
                    if (nv) emit_verified(B, item, v, rep_from, hi, pos_bias, ev);
                    continue;
                }
                // every occurrence that ends in the cluster begins at one of its windows, the first being the leftmost
                const uint32_t from = first > it.floor ? first : it.floor;
                exact_range<E, MODE, false>(A, B, nullptr, nullptr, 0u, item, p, from, hi, rep_from, pos_bias, 0u, ev);
                fallback16 += (hi - from + 15u) >> 4;
            }
        }
        if (ev.n_ev) {
            B.ev_count[item] = ev.n_ev;
            B.rec_count[item] = ev.n_rec;
            if (ev.n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
        }
    }
    wave_walk_whole<E, MODE>(A, B, nullptr, 0u, whole, rec, item, walked, fallback16);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        walked += __shfl_xor(walked, d, 64);
        fallback16 += __shfl_xor(fallback16, d, 64);
    }
    if ((threadIdx.x & 63u) == 0 && walked) atomicAdd(&s_walked, (walked + 15u) >> 4);
    if ((threadIdx.x & 63u) == 0 && fallback16) atomicAdd(&s_fallback, fallback16);
    __syncthreads();
    if (threadIdx.x == 0 && s_walked) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], s_walked);
    if (threadIdx.x == 0 && s_fallback) atomicAdd(&B.counters[7], s_fallback);
}

static hipError_t launch_verify_dense(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                                      hipStream_t st) {
    const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);
    if (A.dense_image != nullptr || A.anchor_pairs != nullptr) {
        // results live behind the window lists in the same buffer (engine.cpp sizes it: 8 bytes per window slot)
        uint32_t *results = reinterpret_cast<uint32_t *>(B.runs) + B.run_stride * uint64_t(B.run_cap);
        const bool image = A.dense_image != nullptr;
        const uint32_t lds1 = image ? ((A.dense_image_words * 4u + 15u) & ~15u) + 16u * 64u * 8u : 16u;   // (+ the waves' kept entries)
        hipError_t e = image ? ensure_dynamic_lds(reinterpret_cast<const void *>(k_dense_decide<true>), uint32_t(lds1))
                             : hipSuccess;
        if (e != hipSuccess) return e;
        const uint32_t per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(2u, (160u * 1024u) / std::max<uint32_t>(lds1, 1u)));
        const uint64_t blocks1 = std::min<uint64_t>((max_items + 15) / 16, uint64_t(num_cus()) * per_cu);
        if (image) hipLaunchKernelGGL(k_dense_decide<true>, dim3(uint32_t(blocks1)), dim3(1024), lds1, st, A, B, results);
        else hipLaunchKernelGGL(k_dense_decide<false>, dim3(uint32_t(blocks1)), dim3(1024), lds1, st, A, B, results);
        const uint32_t blocks2 = uint32_t((max_items + 255) / 256);
        if (entry_bytes == 2) {
            if (class_mode == 1) hipLaunchKernelGGL((k_dense_emit<uint16_t, 1>), dim3(blocks2), dim3(256), 0, st, A, B, results);
            else hipLaunchKernelGGL((k_dense_emit<uint16_t, 0>), dim3(blocks2), dim3(256), 0, st, A, B, results);
        } else {
            if (class_mode == 1) hipLaunchKernelGGL((k_dense_emit<uint32_t, 1>), dim3(blocks2), dim3(256), 0, st, A, B, results);
            else hipLaunchKernelGGL((k_dense_emit<uint32_t, 0>), dim3(blocks2), dim3(256), 0, st, A, B, results);
        }
        return hipGetLastError();
    }
    const uint32_t lds = 16u;   // (no image: candidates are verified through the tables in global memory)
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B);
        return hipGetLastError();
    };
#define ACMI_VERIFY_DENSE(E, M) go(k_verify_dense<E, M, false>)
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_VERIFY_DENSE(uint16_t, 1) : ACMI_VERIFY_DENSE(uint16_t, 0);
    return class_mode == 1 ? ACMI_VERIFY_DENSE(uint32_t, 1) : ACMI_VERIFY_DENSE(uint32_t, 0);
#undef ACMI_VERIFY_DENSE
}

}  // namespace acmi
