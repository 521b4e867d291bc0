// kdev.hpp -- device-side helpers shared by the kernel families (k_*.hip): work-item records, the exact table
// walk (ac_trie_search, reference src/multifast/ahocorasick.c:175-241, as one lookup per byte), the event sink, the
// single-pass prefix sums, run lists, candidate verification.  Everything here is __device__ inline or a constant.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.hpp"
#include "kernels.hpp"

namespace acmi {

int num_cus();

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

struct ItemDesc {
    uint64_t begin;   // blob offset of the first byte this item REPORTS on
    uint32_t len;     // bytes it reports on
    uint32_t warm;    // bytes before `begin` consumed silently from the root: the same haystack's, except the first ...
    uint32_t floor;   // ... `floor` of them (anchor path, haystacks that start unaligned: the scan begins on the 16-byte
                      // boundary below the haystack, in ANOTHER haystack's bytes -- hashed, never part of an occurrence)
    uint32_t pos0;    // begin - haystack begin
    uint32_t hay;
};

// Work item i -> (haystack, chunk) given its haystack index.
__device__ __forceinline__ bool decode_item(const BatchArgs &B, uint64_t i, ItemDesc &it) {
    if (i >= B.item_base[B.n_haystacks]) return false;
    const uint32_t h = B.item_hay[i];
    const uint64_t hb = B.offsets[h], he = B.offsets[h + 1];
    const uint64_t k = i - B.item_base[h];
    const uint64_t p0 = k * B.chunk_bytes;
    it.hay = h;
    it.pos0 = uint32_t(p0);
    it.begin = hb + p0;
    const uint64_t rest = he - it.begin;
    it.len = rest < B.chunk_bytes ? uint32_t(rest) : B.chunk_bytes;
    it.warm = p0 < B.warm_bytes ? uint32_t(p0) : B.warm_bytes;
    it.floor = 0;
    return true;
}

// Load through the constant address space: with a wave-uniform address hipcc then emits a scalar
// s_load (counted by lgkmcnt) instead of a vector load whose wait would drain the haystack
// prefetches queued on vmcnt.  Only for arrays this kernel never writes.
template <typename T>
__device__ __forceinline__ T const_load(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

// Haystack bytes are read exactly once: a non-temporal load (global_load ... nt) keeps them from
// displacing the transition table in L2 and, measured, streams 11-13 % faster than a plain load
// (k_scan_anchor 0.79 -> 0.69 ms per 4 GiB, k_stream_read 6.35 -> 7.07 TB/s).  Only for fully
// coalesced accesses: a line that is fetched in pieces by several instructions is not retained.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream_load(const uint4 *p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// decode_item for a wave-uniform item index, all loads scalar
__device__ __forceinline__ bool decode_item_uniform(const BatchArgs &B, uint64_t i, ItemDesc &it) {
    if (i >= const_load(B.item_base + B.n_haystacks)) return false;
    const uint32_t h = const_load(B.item_hay + i);
    const uint64_t hb = const_load(B.offsets + h), he = const_load(B.offsets + h + 1);
    const uint64_t k = i - const_load(B.item_base + h);
    const uint64_t p0 = k * B.chunk_bytes;
    it.hay = h;
    it.pos0 = uint32_t(p0);
    it.begin = hb + p0;
    const uint64_t rest = he - it.begin;
    it.len = rest < B.chunk_bytes ? uint32_t(rest) : B.chunk_bytes;
    it.warm = p0 < B.warm_bytes ? uint32_t(p0) : B.warm_bytes;
    it.floor = 0;
    return true;
}

// Packed work-item record for k_scan_anchor and k_walk_runs: everything they need about an
// item in one 16-byte load, so that opening an item costs no chain of dependent loads
// (item_hay -> offsets, item_base).  x, y[15:0]: blob offset of the first scanned byte (begin - warm);
// y[31:16]: warm (<= 1023); z: scanned bytes (warm + len, >= 1); w: pos0.
__device__ __forceinline__ uint4 pack_item(const ItemDesc &it) {
    const uint64_t off = it.begin - it.warm;
    // y = offset bits 32..45 | warm (< 2^11) << 14 | floor (< 2^7) << 25
    return make_uint4(uint32_t(off), (uint32_t(off >> 32) & 0x3FFFu) | it.warm << 14 | it.floor << 25, it.warm + it.len, it.pos0);
}
__device__ __forceinline__ void unpack_item(uint32_t x, uint32_t y, uint32_t z, uint32_t w, ItemDesc &it) {
    it.warm = (y >> 14) & 0x7FFu;
    it.floor = y >> 25;
    it.begin = (uint64_t(y & 0x3FFFu) << 32 | x) + it.warm;
    it.len = z - it.warm;
    it.pos0 = w;
    it.hay = 0;  // not carried: the scan kernels do not need it
}

// window hash (automaton.hpp anchor_hash): fold each dword, then a multiply-add chain in 24-bit
// arithmetic -- one full-rate v_mad_u32_u24 per dword after the first
__device__ __forceinline__ uint32_t anchor_fold_dev(uint32_t d) { return d ^ (d >> 11); }
__device__ __forceinline__ uint32_t anchor_mad(uint32_t h, uint32_t mul, uint32_t f) {
    uint32_t r;  // spelled out: from `__umul24(h, mul) + f` hipcc makes a quarter-rate v_mad_u64_u32 half of the time
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(h), "s"(mul), "v"(f));
    return r;
}

// A read from LDS through a pointer the compiler only knows as generic (a function parameter, a pointer that may be null): as a
// plain dereference hipcc emits flat_load, which counts on vmcnt AND lgkmcnt -- the read then waits for every global load and
// store in flight.  The low 32 bits of a generic LDS address are its LDS offset.
template <typename T>
__device__ __forceinline__ T lds_read(const void *generic_lds, uint32_t byte_offset) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>(uint32_t(reinterpret_cast<uintptr_t>(generic_lds)) + byte_offset);
}

__device__ __forceinline__ bool is_final(const DevAutomaton &A, uint32_t s) {
    return (s >= A.f1 && s < A.trap) || s >= A.f2;
}

// Event e of item i in the slab: event-major for small slabs (a thread per item reads and writes whole lines across the items), item-major
// for big ones (BatchArgs::slab_major: an item's events are one contiguous piece -- what a wave per item writes and reads; event-major,
// the 67 M events of a GiB of dense text were 67 M partial lines spread over a 1.2 GB slab)
__device__ __forceinline__ uint64_t slab_slot(const BatchArgs &B, uint64_t item, uint32_t e) {
    return B.slab_major ? item * uint64_t(B.slab_events) + e : uint64_t(e) * B.run_stride + item;
}

// Per-item event sink.  Events, not records, are stored: one {pos, state} per arrival in a
// final state; the per-state output list is expanded later (k_expand / k_refill).  The
// counters live in global memory (zeroed before the launch) rather than in registers: events
// are rare on the hot path and the scan kernel needs its VGPRs for haystack lines.
__device__ __forceinline__ void on_event(const DevAutomaton &A, const BatchArgs &B, uint64_t item, uint32_t pos,
                                         uint32_t s) {
    const uint32_t n = B.ev_count[item];
    if (B.first_only && n) return;  // findAll=false: the first event only (php_ahocorasick.c:588)
    B.rec_count[item] += A.out_count[s];
    if (n < B.slab_events) B.slab[slab_slot(B, item, n)] = make_uint2(pos, s);
    B.ev_count[item] = n + 1;
}

__device__ __forceinline__ void finish_item(const BatchArgs &B, uint64_t item) {
    if (B.ev_count[item] > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
}

// One transition from a state whose row is not in LDS: the dense global table, or -- compact automata -- the
// goto / failure records of the states that have no dense row (ahocorasick.c:199-211 does the same walk per byte:
// edge search, else failure link; here it only happens below the first trie levels, where text rarely is, and it
// ends on the first dense row, whose entry already has the rest of the failure chain folded in).
template <typename E>
__device__ __forceinline__ uint32_t cold_step(const DevAutomaton &A, uint32_t s, uint32_t byte, uint32_t coff) {
    const uint8_t *table = static_cast<const uint8_t *>(A.table);
    if (!A.compact) return *reinterpret_cast<const E *>(table + uint64_t(s) * A.row_bytes + coff);
    for (;;) {
        if (s < A.dense_nf_end) return *reinterpret_cast<const E *>(table + uint64_t(s) * A.row_bytes + coff);
        if (s >= A.f2 && s < A.dense_f_end)
            return *reinterpret_cast<const E *>(table + uint64_t(A.dense_nf_end + (s - A.f2)) * A.row_bytes + coff);
        const uint2 r = A.cold_rec[s];
        const uint32_t code = r.y >> 24;
        if (code == 1u) {  // one trie child: the inside of a pattern
            if ((r.x & 0xffu) == byte) return r.x >> 8;
        } else if (code >= 2u) {  // several: sorted by byte
            const uint32_t *e = A.cold_edges + r.x;
            uint32_t n = code;
            if (code == 255u) n = *e++;
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((e[mid] & 0xffu) < byte) lo = mid + 1; else hi = mid;
            }
            if (lo < n) {
                const uint32_t w = e[lo];
                if ((w & 0xffu) == byte) return w >> 8;
            }
        }
        s = r.y & 0xFFFFFFu;  // failure link: strictly shallower, the root's row is dense
    }
}

// One exact transition.  `tile` (LDS) may be null; hot rows are read from it unless the
// stored target is the trap row, which stands for "the real target is cold".
template <typename E>
__device__ __forceinline__ uint32_t exact_step(const DevAutomaton &A, const uint8_t *tile, uint32_t s,
                                               uint32_t byte) {
    const uint32_t coff = A.class_off[byte];
    if (tile != nullptr && s < A.trap) {
        const uint32_t w = lds_read<E>(tile, s * A.row_bytes + coff);
        if (w < A.trap) return w;
    }
    return cold_step<E>(A, s, byte, coff);
}

// Walk bytes [j0, j1) of a stream one by one with full semantics (events, cold states).
template <typename E>
__device__ __forceinline__ uint32_t exact_bytes(const DevAutomaton &A, const BatchArgs &B, const uint8_t *tile,
                                                uint64_t item, const uint8_t *p, uint32_t j0, uint32_t j1,
                                                uint32_t warm, uint32_t pos0, uint32_t s) {
#pragma unroll 1
    for (uint32_t j = j0; j < j1; ++j) {
        s = exact_step<E>(A, tile, s, p[j]);
        if (j >= warm && is_final(A, s)) on_event(A, B, item, pos0 + (j - warm) + 1, s);
    }
    return s;
}

// ------------------------------------------------------------------------------------------
// Single-pass prefix sums (decoupled look-back) for the two scans of the pipeline.
//
// The batch pipeline used to open with five launches (zero block, tile sums, tile scan, item table) and close
// with four more (tile sums, tile scan, haystack prefix, expansion): 5-9 us each, dependent, host-paced -- 10 %
// of a 0.86 ms step and all of a small batch.  Now
//   k_items      item counts per haystack from the offsets (validated), their exclusive scan, the item table
//                (item -> haystack, packed item records) and the zeroed per-item counters: one launch;
//   k_rec_scan   exclusive scan of the per-item record counts: one launch;
//   k_expand     also writes the per-haystack prefix array, the batch summary, and zeroes the NEXT batch's sync block.
// A tile (2048 elements, one workgroup) takes a ticket, sums its elements, publishes {aggregate}, adds up the
// tiles before it by looking back until it meets a published {inclusive prefix}, publishes its own, and writes
// its results.  Status words are 64-bit {flag:2 | value:62}, written and read with relaxed agent-scope atomics
// (8-byte atomics on both sides need no further ordering; tickets make the look-back deadlock-free whatever the
// dispatch order).  The sync block -- counters, tickets, status words -- alternates between two copies; the
// last pipeline kernel of a batch zeroes the other copy, so no memset is launched.
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanThreads * kScanPerThread;
constexpr uint64_t kChainAgg = 1ull << 62, kChainInc = 2ull << 62, kChainVal = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t chain_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The first WAVE of a workgroup (all 64 lanes call): publish this tile's sum, return the sum of all tiles before it.
// Look-back 64 tiles at a time: the lanes read the status words of the 64 tiles before the window's end, the
// nearest published inclusive prefix closes the sum (everything nearer must at least have its aggregate out, else
// the window is read again).  A single thread walking back one word per L2 round trip made the 257 tiles of a
// 2^19-item batch a 17 us serial chain.
__device__ __forceinline__ uint64_t chain_exclusive(uint64_t *status, uint32_t tile, uint64_t tile_sum) {
    const uint32_t lane = threadIdx.x & 63u;
    if (tile == 0) {
        if (lane == 0) chain_store(status, kChainInc | tile_sum);
        return 0;
    }
    if (lane == 0) chain_store(status + tile, kChainAgg | tile_sum);
    uint64_t before = 0;
    uint32_t end = tile;  // tiles [.., end) are still to be added
    for (;;) {
        const bool mine = lane < end;
        uint64_t v = 0;
        if (mine) v = chain_load(status + (end - 1u - lane));  // lane 0 = the nearest tile
        const uint64_t inc = __ballot(mine && (v >> 62) == 2u);
        const uint64_t missing = __ballot(mine && (v >> 62) == 0u);
        const uint32_t stop = inc ? uint32_t(__builtin_ctzll(inc)) : 64u;             // first lane holding an inclusive prefix
        const uint32_t gap = missing ? uint32_t(__builtin_ctzll(missing)) : 64u;      // first lane with nothing published
        if (gap < stop && gap < 64u) {  // a nearer tile has not published yet: look again
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        uint64_t part = (mine && lane <= stop) ? (v & kChainVal) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        before += part;
        if (stop < 64u) break;
        if (end <= 64u) break;  // (cannot happen: tile 0 publishes an inclusive prefix)
        end -= 64u;
    }
    if (lane == 0) chain_store(status + tile, kChainInc | (before + tile_sum));
    return before;
}

// Block-wide exclusive scan of kScanPerThread values per thread; returns this thread's exclusive base inside the
// tile and, through tile_sum (valid in every thread), the tile total.
__device__ __forceinline__ uint64_t tile_exclusive(const uint64_t (&v)[kScanPerThread], uint64_t *wave_sums, uint64_t &tile_sum) {
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) sum += v[i];
    uint64_t inc = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint64_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const uint64_t x = wave_sums[w];
        if (w < wave) wave_base += x;
        total += x;
    }
    tile_sum = total;
    return wave_base + inc - sum;
}

// Work item `g` = chunk `c` of haystack `h`: its table entries and zeroed counters.
__device__ __forceinline__ void fill_item(const BatchArgs &B, uint64_t h, uint64_t c, uint64_t g) {
    const uint64_t hb = B.offsets[h], he = B.offsets[h + 1];
    const uint64_t p0 = c * B.chunk_bytes;
    ItemDesc it;
    it.hay = uint32_t(h);
    it.pos0 = uint32_t(p0);
    it.begin = hb + p0;
    const uint64_t rest = he - it.begin;
    it.len = rest < B.chunk_bytes ? uint32_t(rest) : B.chunk_bytes;
    it.warm = p0 < B.warm_bytes ? uint32_t(p0) : B.warm_bytes;
    it.floor = 0;
    if (B.align_warm) {
        // Anchor kernel: the scan begins on an aligned address, so that there is no unaligned head to flag wholesale and
        // walk from the root (0.3-0.4 ms per 4 GiB of unaligned items).  A chunk inside its haystack has the haystack's own
        // bytes in front of its warm-up: it takes them up to the 128-byte line below (whole lines per wave-load).  Where the
        // haystack itself begins unaligned the bytes below are ANOTHER haystack's (or none of the blob's: the aligned
        // 16-byte block of a valid byte is in that byte's page): they are hashed, and `floor` keeps them out of matches.
        uint64_t avail = p0 - it.warm;
        const uintptr_t base = reinterpret_cast<uintptr_t>(B.blob);
        const uint64_t extra = (base + it.begin - it.warm) & 127u;
        const uint64_t e = extra < avail ? extra : avail;
        it.warm += uint32_t(e);
        // (non-zero only when the haystack's start was reached; the tile walk, mode 2, wants whole lines here too)
        const uint32_t mis = uint32_t((base + it.begin - it.warm) & (B.align_warm == 2u ? 127u : 15u));
        if (it.begin >= uint64_t(it.warm) + mis) {  // (not below the blob's own first byte: that item keeps an unaligned head)
            it.floor = mis;
            it.warm += mis;
        }
    }
    B.item_hay[g] = uint32_t(h);
    B.item_rec[g] = pack_item(it);
    B.ev_count[g] = 0u;
    B.rec_count[g] = 0ull;
}

constexpr int kLdsThreads = 1024;

// Append byte run [lo, hi) of `item` (scan coordinates) to its fix-up list; adjacent runs merge.
// `entry` is the exact automaton state at `lo` when the stream knew it (else ~0: the epilogue
// re-derives it by walking the Lmax-1 bytes before `lo` from the root).  The list length and the
// end of the last run live in registers (`n`, `last_hi`): an append is a fire-and-forget store,
// never a load, so it does not stall the wave in the main loop.  n > run_cap == "whole item".
__device__ __forceinline__ void append_run(const BatchArgs &B, uint64_t item, uint32_t lo, uint32_t hi,
                                           uint32_t entry, uint32_t &n, uint32_t &last_hi) {
    if (n > B.run_cap) return;
    uint4 *runs = B.runs + item;
    if (n > 0 && last_hi == lo) {
        runs[uint64_t(n - 1) * B.run_stride].y = hi;
    } else {
        if (n < B.run_cap) runs[uint64_t(n) * B.run_stride] = make_uint4(lo, hi, entry, 0u);
        ++n;
    }
    last_hi = hi;
}

// Run hand-over of k_scan_anchor, where a wave handles ONE item together.  Runs are STAGED in LDS
// (2 KiB per wave: 128 x {item k | index << 8, lo, hi, entry}) and written to the items' run lists
// after the streaming loop.  A global store inside that loop cost 0.09 ms per 4 GiB with one hit per
// 8 KiB: the store shares the in-order vmcnt queue with the haystack prefetches, and every wait behind it
// sat out the store's round trip.
struct RunStage {
    uint4 *slots;        // this wave's staging slots (LDS)
    uint32_t count;      // staged entries (wave-uniform)
    uint32_t cap;        // 8-byte entries the stage holds: 256 (2 KiB per wave), or 96 in the big-set kernel, whose probe queue shares the 2 KiB
    bool flushed;        // entries of this wave batch have left for the lists already (the stage does not hold all of them)
};
constexpr uint32_t kStageSlots = 128;
constexpr uint32_t kStageSlotsBig = 48;   // 768 B behind 1 KiB of probe queue and 256 B of hit masks
constexpr uint32_t kProbeQueue = 128;     // entries {hash, window | lane << 5} per wave

// `entry` of a run: the exact automaton state at `lo` (< 2^24), or
constexpr uint32_t kEntryUnknown = 0xFFFFFFFFu;    // walk from the root Lmax-1 bytes before lo
// ... or, derived from a window-list entry of k_scan_anchor (load_run), kEntryWindow | start of the hit window the run
// belongs to (item scan coordinates): every pattern anchored there ends in [start + m_min - 4, start + Lmax).  One run per
// hit window, never merged or clamped: k_walk_runs sees which runs touch and what every run was flagged for.
constexpr uint32_t kEntryWindow = 0x80000000u;
__device__ __forceinline__ bool entry_is_window(uint32_t e) { return (e & kEntryWindow) != 0u && e != kEntryUnknown; }

// ------------------------------------------------------------------------------------------
// Window lists: what k_scan_anchor hands to the kernels behind it (every form of the filter since round 6; the dense form
// always did).  An entry is 4 bytes: the start of a hit window in item scan coordinates -- a run's bounds follow from it
// ([start + lo_bias, min(start + Lmax, n)): load_run below) -- or, bit 31 set, one of the two byte ranges that name no window:
//   kWinSpecial | hi             the unaligned head of an item: bytes [0, hi), walked from the root at byte 0;
//   kWinSpecial | kWinTail | lo  the tail whose windows the streaming loop did not test: bytes [lo, n), entered Lmax-1 bytes early.
// Entry r of item i lives at win_slot(): entry-major (r * run_stride + i: a thread per item reads whole lines -- the sparse
// batches' layout) or item-major (i * run_cap + r: a wave per item does -- the layout of long lists and of the dense form).
// (Until round 5 the aligned forms wrote 16-byte runs {lo, hi, entry, -}, clamped against each other one hit at a time through
// the scalar unit; 4-byte entries have no order dependence, so a task's hits are placed by vector code, and a dense batch's
// lists are a quarter of the size.)
// ------------------------------------------------------------------------------------------
constexpr uint32_t kWinSpecial = 0x80000000u, kWinTail = 0x40000000u;
constexpr uint32_t kWaveRunsMin = 3;     // items with more windows than this are k_runs_wave's where the batch has that kernel
constexpr uint32_t kWaveOccMax = 1024;   // occurrences of one item k_runs_wave orders at most (its LDS list per wave); also the longest window list
__device__ __forceinline__ uint64_t win_slot(const BatchArgs &B, uint64_t item, uint32_t r) {
    return B.run_major ? item * uint64_t(B.run_cap) + r : uint64_t(r) * B.run_stride + item;
}

constexpr uint32_t kDenseStage = 256;   // 8-byte entries in the wave's 2 KiB
constexpr uint32_t kDenseStageBig = 96; // ... in the 768 B the big-set kernel's probe queue leaves
__device__ __forceinline__ void dense_flush(const BatchArgs &B, uint64_t wave_first, RunStage &st, uint32_t lane) {
    const uint2 *slots2 = reinterpret_cast<const uint2 *>(st.slots);
    uint32_t *runs32 = reinterpret_cast<uint32_t *>(B.runs);
    for (uint32_t i = lane; i < st.count; i += 64u) {
        const uint2 e = slots2[i];
        runs32[win_slot(B, wave_first + (e.x & 0xffu), e.x >> 8)] = e.y;
    }
    st.flushed = st.flushed || st.count != 0u;
    st.count = 0;
}

// One entry of item k's window list, from the scalar path of k_scan_anchor (wave-uniform arguments).  The count keeps running
// past the list's capacity (the host learns the size a batch asked for from it); entries beyond it are not stored, and
// n > run_cap means "the whole item" to the kernels behind.
__device__ __forceinline__ void append_win_wave(const BatchArgs &B, uint64_t wave_first, RunStage &st, uint32_t k, uint32_t value,
                                                uint32_t &n, uint32_t lane) {
    if (n < B.run_cap) {
        if (st.count == st.cap) dense_flush(B, wave_first, st, lane);
        if (lane == 0) reinterpret_cast<uint2 *>(st.slots)[st.count] = make_uint2(k | (n << 8), value);
        ++st.count;
    }
    ++n;
}

// Run r of an item as run_epilogue_fast wants it, {lo, hi, entry, -}, from its window list.
__device__ __forceinline__ uint4 run_of_entry(const DevAutomaton &A, uint32_t e, uint32_t n);
__device__ __forceinline__ uint4 load_run(const DevAutomaton &A, const BatchArgs &B, uint64_t item, uint32_t r, uint32_t n) {
    return run_of_entry(A, reinterpret_cast<const uint32_t *>(B.runs)[win_slot(B, item, r)], n);
}
__device__ __forceinline__ uint4 run_of_entry(const DevAutomaton &A, uint32_t e, uint32_t n) {
    if (e & kWinSpecial) {
        const uint32_t v = e & 0x3FFFFFFFu;
        return (e & kWinTail) ? make_uint4(v, n, kEntryUnknown, 0u) : make_uint4(0u, v, 0u, 0u);
    }
    // the end of a pattern anchored at window e: at least m_min - 4 past the window start when the window is the pattern's FIRST
    // aligned dword, at least q - 1 when the builder chose a deeper window (anchor_choice); the scan stores no window whose
    // lower bound lies behind the item's end
    const uint32_t lo_bias = A.anchor_choice ? A.anchor_q - 1u : A.min_pattern_length - 4u;
    const uint32_t hi = e + A.max_pattern_length < n ? e + A.max_pattern_length : n;
    return make_uint4(e + lo_bias, hi, kEntryWindow | e, 0u);
}

// 4 haystack bytes -> 4 column offsets (one per byte lane of the result), SWAR3 mode.
template <typename E>
__device__ __forceinline__ uint32_t swar_columns(const DevAutomaton &A, uint32_t d) {
    uint32_t c4 = (d >> A.swar_shift) & 0x07070707u;
    const uint32_t e4 = __builtin_amdgcn_perm(A.swar_expect_hi, A.swar_expect_lo, c4);
    uint32_t x = e4 ^ d;
    if (__builtin_expect(x != 0, 0)) {  // some byte is outside the alphabet -> "other" column
        x |= x >> 4;
        x |= x >> 2;
        x |= x >> 1;
        const uint32_t mask = (x & 0x01010101u) * 0xffu;
        c4 = (c4 & ~mask) | (mask & (A.swar_other_off / uint32_t(sizeof(E)) * 0x01010101u));
    }
    return c4 * uint32_t(sizeof(E));
}

// byte -> column offset inside a row.  SWAR3: pure arithmetic.  TABLE: the 256-entry table, from
// LDS when the caller staged it there (`cls_lds`), else from global memory.
template <typename E, int MODE>
__device__ __forceinline__ uint32_t class_offset(const DevAutomaton &A, const uint16_t *cls_lds, uint32_t byte) {
    if (MODE == 1) {
        const uint32_t c = (byte >> A.swar_shift) & 7u;
        // selector bytes 1..3 = 0x0c pick the constant 0x00
        const uint32_t e = __builtin_amdgcn_perm(A.swar_expect_hi, A.swar_expect_lo, c | 0x0c0c0c00u);
        return e == byte ? c * uint32_t(sizeof(E)) : A.swar_other_off;
    }
    if (A.class_identity) return byte * uint32_t(sizeof(E));  // all 256 byte values in use: no lookup (one dependent load less per step)
    // (explicitly an LDS read: through the generic pointer hipcc emitted flat_load_ushort -- which counts on vmcnt as well, so a
    // byte's class waited for every global load and store in flight; the low 32 bits of a generic LDS address are its LDS offset)
    if (cls_lds != nullptr)
        return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(uint32_t(reinterpret_cast<uintptr_t>(cls_lds)) + byte * 2u);
    return A.class_off[byte];
}

// Event counters of the item being walked by the epilogue, kept in registers and stored once.
struct EventRegs {
    uint64_t n_rec;
    uint32_t n_ev;
};

// EXTRA compiles in what only some automata use -- the early stop of single-window anchor runs: without it the
// headline's k_walk_runs takes 65 us, with the test in the loop 72.
// Exact walk of bytes [j0, j1) of one item: hot rows from the LDS tile when one is resident
// (`tile`), everything else from the global table; haystack bytes by aligned dword loads.
// Reports (records an event) for final states at byte index j >= rep_from, at position
// pos_bias + j.  Returns the state.
template <typename E, int MODE, bool EXTRA = false>
__device__ __forceinline__ uint32_t exact_range(const DevAutomaton &A, const BatchArgs &B, const uint16_t *cls_lds,
                                                const uint8_t *tile, uint32_t tile_rows, uint64_t item,
                                                const uint8_t *p, uint32_t j0, uint32_t j1, uint32_t rep_from,
                                                uint32_t pos_bias, uint32_t s, EventRegs &ev,
                                                uint32_t stop_from = 0xFFFFFFFFu, bool store = true) {
    // Haystack bytes come in aligned 16-byte blocks, the next block requested one block ahead: one
    // exposed memory round trip per call instead of one per dword (the walks are latency chains).
    // A block is only touched when it holds a needed byte, so it lies inside a mapped page.
    const uintptr_t base = reinterpret_cast<uintptr_t>(p);
    if (j0 >= j1) return s;
    uintptr_t blk = (base + j0) & ~uintptr_t(15);
    const uintptr_t end = base + j1;
    // (explicitly global: through an address computed as an integer hipcc emits flat_load, whose wait also drains the LDS queue)
    auto text16 = [](uintptr_t a) __attribute__((always_inline)) -> uint4 {
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(a);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    uint4 cur = text16(blk), nxt = cur;
    if (blk + 16 < end) nxt = text16(blk + 16);
#pragma unroll 1
    for (uint32_t j = j0; j < j1; ++j) {
        const uintptr_t a = base + j;
        if (j != j0 && (a & 15u) == 0) {
            cur = nxt;
            blk += 16;
            if (blk + 16 < end) nxt = text16(blk + 16);
        }
        const uint32_t sel = uint32_t(a >> 2) & 3u;
        const uint32_t word = sel == 0 ? cur.x : sel == 1 ? cur.y : sel == 2 ? cur.z : cur.w;
        const uint32_t byte = (word >> (8u * uint32_t(a & 3u))) & 0xffu;
        const uint32_t coff = class_offset<E, MODE>(A, cls_lds, byte);
        uint32_t w = A.trap;
        if (s < tile_rows) w = lds_read<E>(tile, s * A.row_bytes + coff);  // LDS-resident row
        if (w >= A.trap) w = cold_step<E>(A, s, byte, coff);  // row not in LDS
        s = w;
        if (j >= rep_from && is_final(A, s) && !(B.first_only && ev.n_ev)) {
            ev.n_rec += A.out_count[s];
            if (store && ev.n_ev < B.slab_events) B.slab[slab_slot(B, item, ev.n_ev)] = make_uint2(pos_bias + j, s);
            ++ev.n_ev;
        }
        // an anchor candidate's walk: from byte stop_from on, a non-final state of trie depth <= 2 means that
        // no pattern begun at or before the hit window is still alive -- nothing more can end in this run
        if (EXTRA && j >= stop_from && (s < A.shallow_hot || (s > A.trap && s < A.shallow_cold_end))) break;
    }
    return s;
}

// Exact walk of one item's flagged byte runs (k_scan_lds' in-kernel epilogue, k_walk_runs).  Events only
// arise here, in ascending position order (runs are disjoint and ascending).  A run [lo, hi) whose
// entry state is unknown is entered from the root Lmax-1 bytes earlier -- exact for every pattern
// ending inside the run -- and only events ending inside it (and inside the item's reporting range)
// are recorded.  nr > run_cap means the run list overflowed: the whole item is walked.
// Returns the number of bytes it walked exactly (statistic for the kernel choice).
template <typename E, int MODE, bool EXTRA = false>
__device__ __forceinline__ uint32_t run_epilogue(const DevAutomaton &A, const BatchArgs &B, const uint16_t *cls_lds,
                                                 const uint8_t *tile, uint32_t tile_rows, uint64_t item,
                                                 const ItemDesc &it, uint32_t nr) {
    const uint8_t *p = B.blob + (it.begin - it.warm);
    const uint32_t n = it.warm + it.len;
    const uint32_t pos_bias = it.pos0 - it.warm + 1u;
    EventRegs ev{0, 0};
    uint32_t walked = 0;
    if (nr > B.run_cap) {
        walked = n;
        exact_range<E, MODE, EXTRA>(A, B, cls_lds, tile, tile_rows, item, p, it.floor, n, it.warm, pos_bias, 0u, ev);
    } else {
        const uint4 *runs = B.runs + item;
#pragma unroll 1
        for (uint32_t r = 0; r < nr; ++r) {
            const uint4 run = runs[uint64_t(r) * B.run_stride];
            const uint32_t lo = run.x, hi = run.y;
            uint32_t st = run.z, from = lo;
            walked += hi - lo;
            if (st == kEntryUnknown) {  // entry state unknown: silent warm-up from the root, Lmax-1 bytes
                from = lo > B.warm_bytes ? lo - B.warm_bytes : 0u;
                if (from < it.floor) from = it.floor;  // (the bytes below are another haystack's)
                st = 0;
            }
            // warm-up and run in one loop: nothing before `lo` is reported
            exact_range<E, MODE, EXTRA>(A, B, cls_lds, tile, tile_rows, item, p, from, hi, lo > it.warm ? lo : it.warm, pos_bias, st, ev);
        }
    }
    if (ev.n_ev) {
        B.ev_count[item] = ev.n_ev;
        B.rec_count[item] = ev.n_rec;
        // events only arise here, so the overflow test needs no reload of ev_count (cf. finish_item)
        if (ev.n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
    }
    return walked;
}

// ------------------------------------------------------------------------------------------
// walk_fast: the exact walk of ONE candidate run, built for k_walk_runs.
//
// exact_range above costs ~200 wave-instructions per byte-step once 64 lanes diverge over its branches (block
// reloads, per-byte dword selects, cold-row loop), and a wave takes as many steps as its longest walk has bytes:
// k_walk_runs was bound by instruction issue, not by memory (config 3: 6,200 VALU + 7,000 SALU per wave for ~67
// lockstep steps).  Here every lane has its own cursor and every loop iteration is useful work for every lane
// that is still walking:
//   * a lane holds 16 text bytes in registers, read by ONE unaligned 16-byte load at its cursor (byte k of the
//     window = a dword select and a shift, no memory access);
//   * a state inside a pattern (compact automata: ids [chain_lo, f2) are numbered along the patterns) is not
//     stepped through byte by byte with dependent loads: its next 16 edge bytes are compared with the 16 text
//     bytes at the cursor in one go and the cursor jumps over what matched -- no event test on the way (such
//     states are non-final and deeper than 2).  A 36-byte pattern is ~10 iterations instead of ~40;
//   * events, the early stop and the cold step keep the semantics of exact_range, which also finishes the last
//     bytes of an item (where a 16-byte load would cross its end).
// ------------------------------------------------------------------------------------------
typedef u32x4 u32x4_u __attribute__((aligned(1)));
// (explicitly global: through a generic pointer hipcc emits flat_load, whose wait also drains the LDS queue)
__device__ __forceinline__ uint4 load16u(const uint8_t *a) {
    const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_u *>(reinterpret_cast<uintptr_t>(a));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t load8g(const uint8_t *a) {
    return *reinterpret_cast<const __attribute__((address_space(1))) uint8_t *>(reinterpret_cast<uintptr_t>(a));
}

// The loop body is written as straight-line predicated code (three loads issued together, one wait, selects):
// as nested divergent branches the same logic compiled to ~120 scalar branch / exec-mask sequences and a
// vmcnt(0) behind every load, and the kernel was bound by instruction issue again.
template <typename E, int MODE, bool EXTRA>
__device__ __forceinline__ uint32_t walk_fast(const DevAutomaton &A, const BatchArgs &B, const uint8_t *tile,
                                              uint32_t tile_rows, uint64_t item, const uint8_t *p, uint32_t from,
                                              uint32_t hi, uint32_t safe_end, uint32_t rep_from, uint32_t pos_bias,
                                              uint32_t s, EventRegs &ev, uint32_t stop_from) {
    uint32_t j = from;
    uint32_t jbase = from;  // text[jbase .. jbase+16) is what the window load fetches
    const bool chains = A.chain_byte != nullptr;
    const uint32_t chain_lo = A.chain_lo, f2 = A.f2, trap = A.trap, row_bytes = A.row_bytes;
    // 16-byte loads must stay inside the item: the last bytes are walked with single loads
    const uint32_t fast_hi = safe_end >= 16u ? (hi < safe_end - 15u ? hi : safe_end - 15u) : 0u;
#pragma unroll 1
    while (j < fast_hi) {
        const bool in_chain = chains && s >= chain_lo && s < f2;
        uint32_t k = j - jbase;
        if (k >= 16u || (in_chain && k != 0u)) { jbase = j; k = 0; }
        const uint32_t idx = in_chain ? s - chain_lo : 0u;
        const uint4 t = load16u(p + jbase);           // (a line this lane has touched before, unless the window moved)
        const uint4 cb = load16u((chains ? A.chain_byte : A.pad16) + idx);
        const uint32_t run = load8g((chains ? A.chain_run : A.pad16) + idx);
        // the 16 text bytes at the cursor against the 16 edge bytes from this state on
        uint32_t m = 16u;
        uint32_t x = t.w ^ cb.w;
        m = x ? 12u + (uint32_t(__builtin_ctz(x)) >> 3) : m;
        x = t.z ^ cb.z;
        m = x ? 8u + (uint32_t(__builtin_ctz(x)) >> 3) : m;
        x = t.y ^ cb.y;
        m = x ? 4u + (uint32_t(__builtin_ctz(x)) >> 3) : m;
        x = t.x ^ cb.x;
        m = x ? (uint32_t(__builtin_ctz(x)) >> 3) : m;
        m = m < run ? m : run;
        m = m < hi - j ? m : hi - j;
        m = in_chain ? m : 0u;                        // m steps along the pattern: states s+1 .. s+m are of this kind too
        // one ordinary step (taken when m == 0)
        const uint32_t word = k < 8u ? (k < 4u ? t.x : t.y) : (k < 12u ? t.z : t.w);
        const uint32_t byte = (word >> (8u * (k & 3u))) & 0xffu;
        const uint32_t coff = class_offset<E, MODE>(A, nullptr, byte);
        const uint32_t row = s < tile_rows ? s : 0u;
        uint32_t w = lds_read<E>(tile, row * row_bytes + coff);
        w = s < tile_rows ? w : trap;
        if (m == 0u && w >= trap) w = cold_step<E>(A, s, byte, coff);
        const bool stepped = m == 0u;
        s = stepped ? w : s + m;
        if (stepped && j >= rep_from && is_final(A, s) && !(B.first_only && ev.n_ev)) {
            ev.n_rec += A.out_count[s];
            if (ev.n_ev < B.slab_events) B.slab[slab_slot(B, item, ev.n_ev)] = make_uint2(pos_bias + j, s);
            ++ev.n_ev;
        }
        if (EXTRA && stepped && j >= stop_from && (s < A.shallow_hot || (s > trap && s < A.shallow_cold_end))) return s;
        j += stepped ? 1u : m;
    }
    if (j < hi) s = exact_range<E, MODE, EXTRA>(A, B, nullptr, tile, tile_rows, item, p, j, hi, rep_from, pos_bias, s, ev, stop_from);
    return s;
}

// ------------------------------------------------------------------------------------------
// verify_candidate: decide a single-window candidate of the anchor filter WITHOUT walking.
//
// A run that serves one hit window (the case the early stop is also built on: nothing anchored at another window
// can end inside it) can only contain occurrences of patterns whose anchor q-gram is the window's.  When that
// q-gram belongs to exactly one (pattern, offset) pair -- every q-gram of a random signature set, most of any
// real one -- the run holds one event or none: the pattern's text either is there or is not.  So: hash the
// window again, find its slot in the filter's own fingerprint table, fetch the pair, compare the pattern's bytes
// with the text in 16-byte pieces, and on equality report the occurrence with the pattern's final state (its output
// list carries the shorter patterns that end there too, exactly as after a walk).  Four dependent memory round
// trips and ~100 instructions, the same for every lane of a wave, against one round trip and ~200 lockstep
// instructions per BYTE of the pattern for the walk.  Returns false when the slot stands for several pairs
// (patterns sharing a q-gram, fingerprint twins): the caller walks.
// ------------------------------------------------------------------------------------------
// Result of verify_candidate for one window
struct Verified {
    uint32_t j;      // index of the occurrence's last byte, 0xFFFFFFFF: the pattern is not there
    uint32_t state;  // its final state
    uint32_t cnt;    // out_count[state], requested together with the texts that are compared (one round trip less)
};

typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
typedef u32x2v u32x2_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
__device__ __forceinline__ uint2 load8u(const uint8_t *a) {
    const u32x2v v = *reinterpret_cast<const __attribute__((address_space(1))) u32x2_u *>(reinterpret_cast<uintptr_t>(a));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t load4u(const uint8_t *a) {
    return *reinterpret_cast<const __attribute__((address_space(1))) u32_u *>(reinterpret_cast<uintptr_t>(a));
}

constexpr uint32_t kClusterMax = 6;   // occurrences a cluster of candidate windows may hold (more: the cluster is walked)

// Appends the occurrences anchored at window w to v[0 .. nv) (kept in registers: the index is lane-divergent, hence the
// select chain).  False when the window cannot be decided here.
__device__ __forceinline__ bool verify_candidate(const DevAutomaton &A, const uint8_t *p, uint32_t n, uint32_t floor, uint32_t w,
                                                 Verified (&v)[kClusterMax], uint32_t &nv) {
    if (w + A.anchor_q > n) return true;   // the window reaches past the item's end: no pattern anchored there fits
    // the window's hash (automaton.hpp anchor_hash; the window lies inside the item: the filter tested it)
    const uint32_t nw = A.anchor_q >> 2;
    uint32_t h;
    if (nw == 1u) {  // the raw window, one multiply-add (automaton.hpp anchor_hash)
        const uint32_t t4 = load4u(p + w);
        h = anchor_mad(t4, A.anchor_mul, t4 >> 13);
    } else if (w + 16u <= n) {  // one request for the whole window (what bounds this kernel is its count of divergent requests)
        const uint4 t = load16u(p + w);
        h = anchor_mad(anchor_fold_dev(t.x), A.anchor_mul, anchor_fold_dev(t.y));
        if (nw >= 3u) h = anchor_mad(h, A.anchor_mul, anchor_fold_dev(t.z));
        if (nw >= 4u) h = anchor_mad(h, A.anchor_mul, anchor_fold_dev(t.w));
    } else {
        h = anchor_fold_dev(load4u(p + w));
#pragma unroll 1
        for (uint32_t i = 1; i < nw; ++i) h = anchor_mad(h, A.anchor_mul, anchor_fold_dev(load4u(p + w + 4u * i)));
    }
    // its slot in the fingerprint table
    uint32_t slot = 0xFFFFFFFFu;
    if (A.anchor_mode == 2u) {
        const uint32_t bucket = (h >> A.anchor_bucket_shift) & A.anchor_bucket_mask;
        const uint4 b = static_cast<const uint4 *>(A.anchor_table)[bucket];
        const uint32_t fp = (h >> 16) | 1u;
        const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if ((bv[i] >> 16) == fp) slot = bucket * 8u + uint32_t(i) * 2u + 1u;
            if ((bv[i] & 0xffffu) == fp) slot = bucket * 8u + uint32_t(i) * 2u;
        }
    } else if (A.anchor_slots == 2u) {
        const uint32_t bucket = (h >> 2) & 0x7FFFu;
        const uint32_t b = static_cast<const uint32_t *>(A.anchor_table)[bucket];
        const uint32_t fp = h >> 16;
        if ((b >> 16) == fp) slot = bucket * 2u + 1u;
        if ((b & 0xffffu) == fp) slot = bucket * 2u;
    } else {
        const uint32_t bucket = (h >> 3) & 0x3FFFu;
        const uint2 b = static_cast<const uint2 *>(A.anchor_table)[bucket];
        const uint32_t fp = h >> 16;
        const uint32_t bv[2] = {b.x, b.y};
#pragma unroll
        for (int i = 1; i >= 0; --i) {
            if ((bv[i] >> 16) == fp) slot = bucket * 4u + uint32_t(i) * 2u + 1u;
            if ((bv[i] & 0xffffu) == fp) slot = bucket * 4u + uint32_t(i) * 2u;
        }
    }
    if (slot == 0xFFFFFFFFu) return false;  // (cannot happen for a window the filter flagged; walk to be safe)
    const uint32_t sp = A.anchor_slot_pair[slot];
    if (sp == 0xFFFFFFFFu) return false;    // too many pairs behind this slot
    const uint32_t first = sp & 0x0FFFFFFFu, count = (sp >> 28) + 1u;
    // every (pattern, offset) pair the slot stands for: the q-gram's own, those of q-grams that share its fingerprint
#pragma unroll 1
    for (uint32_t k = 0; k < count; ++k) {
        const uint4 rec = A.anchor_pairs[first + k];
        const uint32_t o = rec.y & 0xffu, len = rec.y >> 8;
        if (w < o + floor || w - o + len > n) continue;  // would begin before / end after this item's bytes: no occurrence here
        const uint8_t *tp = p + (w - o), *pp = A.pattern_bytes + rec.z;
        const uint32_t oc = rec.x;   // out_count of the pattern's final state
        uint32_t diff = 0;
        if (len >= 16u) {
            // (requesting four clamped 16-byte pieces of both texts up front -- one round trip whatever the length -- was
            // measured and is slower, 0.166 against 0.162 ms on config 3: the kernel is bound by the number of divergent
            // requests a CU's vector memory path takes per clock, ~ 15 per candidate, not by the length of the chain)
#pragma unroll 1
            for (uint32_t c = 0; c + 16u <= len; c += 16u) {
                const uint4 a = load16u(tp + c), b = load16u(pp + c);
                diff |= (a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w);
            }
            if (len & 15u) {
                const uint4 a = load16u(tp + len - 16u), b = load16u(pp + len - 16u);  // the last 16 bytes (may overlap)
                diff |= (a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w);
            }
        } else if (len >= 8u) {
            const uint2 a0 = load8u(tp), b0 = load8u(pp), a1 = load8u(tp + len - 8u), b1 = load8u(pp + len - 8u);
            diff = (a0.x ^ b0.x) | (a0.y ^ b0.y) | (a1.x ^ b1.x) | (a1.y ^ b1.y);
        } else {  // 4..7 bytes (the dense form anchors patterns from 4 B on)
            diff = (load4u(tp) ^ load4u(pp)) | (load4u(tp + len - 4u) ^ load4u(pp + len - 4u));
        }
        if (diff) continue;
        if (nv >= kClusterMax) return false;
        const Verified vv{w - o + len - 1u, rec.w, oc};  // index of the occurrence's last byte, the pattern's final state
#pragma unroll
        for (uint32_t i = 0; i < kClusterMax; ++i) if (i == nv) v[i] = vv;
        ++nv;
    }
    return true;
}

// The events of a cluster whose windows were all decided by verification: the occurrences in position order (windows
// ascend, so do pattern starts per window, ends need not); of several that end at the same byte one event, the deepest
// final state among them (the longest pattern: its output list carries the others).
__device__ __forceinline__ void emit_verified(const BatchArgs &B, uint64_t item, Verified (&v)[kClusterMax], uint32_t rep_from,
                                              uint32_t hi, uint32_t pos_bias, EventRegs &ev) {
#pragma unroll
    for (uint32_t a = 0; a + 1 < kClusterMax; ++a)
#pragma unroll
        for (uint32_t b = 0; b + 1 < kClusterMax - a; ++b)
            if (v[b + 1].j < v[b].j) { const Verified t = v[b]; v[b] = v[b + 1]; v[b + 1] = t; }
#pragma unroll
    for (uint32_t i = 0; i < kClusterMax; ++i) {
        const uint32_t j = v[i].j;
        if (j == 0xFFFFFFFFu) continue;
        // same end as the next one: after the sort equal ends are neighbours
        if (i + 1 < kClusterMax && v[i + 1].j == j) {
            if (v[i + 1].cnt < v[i].cnt) { v[i + 1].state = v[i].state; v[i + 1].cnt = v[i].cnt; }
            continue;
        }
        if (j >= rep_from && j < hi && !(B.first_only && ev.n_ev)) {
            const uint32_t s = v[i].state;
            ev.n_rec += v[i].cnt;
            if (ev.n_ev < B.slab_events) B.slab[slab_slot(B, item, ev.n_ev)] = make_uint2(pos_bias + j, s);
            ++ev.n_ev;
        }
    }
}

// run_epilogue for k_walk_runs.  k_scan_anchor leaves one window-list entry per hit window, ascending; the run of a window is
// the byte range in which a pattern anchored there can end (load_run).  Runs that touch or overlap form a CLUSTER: everything
// that ends inside a cluster is an occurrence of a pattern anchored at one of the cluster's windows (an occurrence anchored
// elsewhere would have flagged a run of its own, overlapping this range, hence in the cluster).  So a cluster of
// window runs is decided by verifying its windows (verify_candidate) and emitting the occurrences found in
// position order -- of two that end at the same byte only the longer is an event, the shorter is in its output
// list.  Clusters with a run that names no window (unaligned head, untested tail), with a fingerprint slot
// that stands for more than 8 pairs, or with more than kClusterMax occurrences are walked as one range.
template <typename E, int MODE, bool EXTRA>
__device__ __forceinline__ uint32_t run_epilogue_fast(const DevAutomaton &A, const BatchArgs &B, const uint8_t *tile,
                                                      uint32_t tile_rows, uint64_t item, const ItemDesc &it, uint32_t nr,
                                                      uint32_t &fallback16, uint32_t entry0 = 0u, bool have_entry0 = false) {
    const uint8_t *p = B.blob + (it.begin - it.warm);
    const uint32_t n = it.warm + it.len;
    const uint32_t pos_bias = it.pos0 - it.warm + 1u;
    EventRegs ev{0, 0};
    uint32_t walked = 0;
    if (nr > B.run_cap) {
        walked = n;
        exact_range<E, MODE, EXTRA>(A, B, nullptr, tile, tile_rows, item, p, it.floor, n, it.warm, pos_bias, 0u, ev);
        fallback16 += (n + 15u) >> 4;  // statistic: walked through the automaton, not decided by a text compare (summed per workgroup:
                                       // an atomic per cluster on one word serialises the launch at ~ 12 ns each)
    } else {
        uint32_t r = 0;
        // (entry0: the list's first entry, requested by the caller together with the list's length -- a round trip less per item)
        uint4 run = nr ? (have_entry0 ? run_of_entry(A, entry0, n) : load_run(A, B, item, 0u, n)) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
        while (r < nr) {
            // the cluster that starts with run r
            const uint4 first = run;
            uint32_t hi = first.y, count = 1, nv = 0;
            bool windows = entry_is_window(first.z) && A.anchor_pairs != nullptr;
            Verified v[kClusterMax];
#pragma unroll
            for (uint32_t i = 0; i < kClusterMax; ++i) v[i] = Verified{0xFFFFFFFFu, 0u, 0u};
            if (windows) windows = verify_candidate(A, p, n, it.floor, first.z & ~kEntryWindow, v, nv);
#pragma unroll 1
            for (;;) {
                ++r;
                if (r >= nr) break;
                run = load_run(A, B, item, r, n);
                if (run.x > hi) break;  // neither adjacent nor overlapping (runs are not clamped against each other): the next cluster
                hi = run.y > hi ? run.y : hi;
                windows = windows && entry_is_window(run.z);
                if (windows) windows = verify_candidate(A, p, n, it.floor, run.z & ~kEntryWindow, v, nv);
                ++count;
            }
            const uint32_t lo = first.x;
            const uint32_t rep_from = lo > it.warm ? lo : it.warm;
            walked += hi - lo;
            if (windows) {
                emit_verified(B, item, v, rep_from, hi, pos_bias, ev);
                continue;
            }
            // walk the cluster as one range: from the root, m_min-1 bytes before a window's own lower bound (every
            // pattern that ends in the cluster begins at or after the first window - 3), Lmax-1 bytes otherwise
            uint32_t st = 0, from = lo, stop_from = 0xFFFFFFFFu;
            if (A.anchor_dense && entry_is_window(first.z)) {
                // dense form: every pattern BEGINS at its window, and the cluster's first window is its leftmost
                from = first.z & ~kEntryWindow;
            } else if (!A.anchor_choice && entry_is_window(first.z) && lo + 4u == (first.z & ~kEntryWindow) + A.min_pattern_length) {
                const uint32_t warm = A.min_pattern_length - 1u;
                from = lo > warm ? lo - warm : 0u;
                // a lone window: nothing is alive once the state is shallow two bytes past the window start
                if (EXTRA && A.early_stop && count == 1u && lo + 6u >= A.min_pattern_length) stop_from = lo + 6u - A.min_pattern_length;
            } else if (first.z == 0u && lo == 0u) {
                from = 0;  // the item's first bytes: the scan begins in the root
            } else {
                from = lo > B.warm_bytes ? lo - B.warm_bytes : 0u;
            }
            if (from < it.floor) from = it.floor;  // (the bytes below are another haystack's)
            walk_fast<E, MODE, EXTRA>(A, B, tile, tile_rows, item, p, from, hi, n, rep_from, pos_bias, st, ev, stop_from);
            fallback16 += (hi - from + 15u) >> 4;  // statistic: bytes walked because verification could not decide
        }
    }
    if (ev.n_ev) {
        B.ev_count[item] = ev.n_ev;
        B.rec_count[item] = ev.n_rec;
        if (ev.n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
    }
    return walked;
}

// Inclusive prefix sum over the 64 lanes of a wave in registers (DPP): within the rows of 16 lanes by shifts of 1, 2, 4, 8,
// then row 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the first half's total into the second
// (row_bcast:31).  All lanes must be active.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112 /* row_shr:2 */, 0xf, 0xf, true));
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114 /* row_shr:4 */, 0xf, 0xf, true));
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118 /* row_shr:8 */, 0xf, 0xf, true));
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return x;
}

// Sum `v` over the wave and add it (in units of 16 bytes) to counters[7]: how much of the batch the
// exact path had to walk.  Must be reached by all lanes of the wave.
__device__ __forceinline__ void account_walked(const BatchArgs &B, uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&B.counters[7], (v + 15u) >> 4);
}

// Work item of k_scan_lds: from the item table when k_items aligned the items (the record carries the lengthened
// warm-up and the floor), else recomputed from the offsets.
__device__ __forceinline__ bool lds_item(const BatchArgs &B, uint64_t i, ItemDesc &it) {
    if (B.align_warm != 2u) return decode_item(B, i, it);
    if (i >= B.item_base[B.n_haystacks]) return false;
    const uint4 rec = B.item_rec[i];
    unpack_item(rec.x, rec.y, rec.z, rec.w, it);
    it.hay = B.item_hay[i];
    return true;
}

// The quad a stream feeds the automaton: bytes below the item's floor (quads of an aligned item that lie in front of its
// haystack) and behind its end (last quad) replaced by A.other_byte.  hf = head | floor << 8 | valid bytes of the last
// quad << 16; the test is two compares for every quad, the masking runs for the first and the last quads of an item only.
__device__ __forceinline__ uint4 masked_quad(const DevAutomaton &A, const uint4 &v, uint32_t q, uint32_t hf, uint32_t last) {
    const uint32_t floor = (hf >> 8) & 0xffu, tv = hf >> 16;
    if (__builtin_expect((q << 4) >= floor && !(q == last && tv < 16u), 1)) return v;
    const uint32_t other4 = A.other_byte * 0x01010101u;
    const int lead = int(floor) - int(q << 4);                       // bytes of this quad below the floor
    const int keep = q == last ? int(tv) : 16;                        // bytes of this quad inside the item
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int l = lead - 4 * w, t = 4 * w + 4 - keep;             // leading / trailing bytes of dword w to replace
        uint32_t m = 0;
        if (l > 0) m |= l >= 4 ? 0xFFFFFFFFu : (1u << (8 * l)) - 1u;
        if (t > 0) m |= t >= 4 ? 0xFFFFFFFFu : ~((1u << (8 * (4 - t))) - 1u);
        d[w] = (d[w] & ~m) | (other4 & m);
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
}

constexpr uint32_t kWalkThreads = 1024;
constexpr uint32_t kWalkTileBytes = 64u * 1024u;
constexpr uint32_t kItemsTileBytes = 78u * 1024u;

// Expand one event into records, longest pattern first (node.c:424-441 order).
__device__ __forceinline__ uint64_t emit_event(const DevAutomaton &A, uint4 *out, uint64_t w, uint32_t hay,
                                               uint32_t pos, uint32_t state) {
    if (A.out_head != nullptr) {   // most events stand for one record: everything it needs in one 8-byte read
        const uint2 h = A.out_head[state];
        if ((h.x >> 24) == 1u) {
            out[w] = make_uint4(hay, pos, pos - h.y, h.x & 0xFFFFFFu);
            return w + 1;
        }
    }
    uint32_t t = A.own_pattern[state] != 0xFFFFFFFFu ? state : A.dict_link[state];
    while (t != 0xFFFFFFFFu) {
        const uint32_t pat = A.own_pattern[t];
        out[w++] = make_uint4(hay, pos, pos - A.pattern_length[pat], pat);
        t = A.dict_link[t];
    }
    return w;
}

}  // namespace acmi
