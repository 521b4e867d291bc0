// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// The hot path replaces ac_trie_search() (reference src/multifast/ahocorasick.c:175-241)
// and node_find_next_bs() (src/multifast/node.c:119-140): instead of a goto edge binary
// search plus failure hops (~1.9 pointer-chased node visits per byte), every haystack byte
// costs ONE lookup in a failure-free transition table whose hot rows live in LDS.
//
// Pipeline for one batch (all on one stream, one host synchronisation at the very end):
//   k_tile_sums<COUNT> + k_tile_scan<COUNT>            work items per haystack from the offsets, scanned
//   k_build_items                                      item -> haystack, packed item records
//   the scan kernel, by pattern set:
//     k_scan_anchor + k_walk_runs                      every pattern >= 11 B: q-gram anchor filter at HBM
//                                                      speed; the byte runs it flags are walked exactly
//                                                      by a kernel of their own
//     k_scan_lds                                       otherwise: optimistic LDS-tile table walk, flagged
//                                                      byte runs walked exactly in its dense epilogue
//                                                      events -> per-item slabs
//   k_first_only_* (findAll == false)                  keep only the first event per haystack
//   k_tile_sums + k_tile_scan of per-item record counts -> record base of every item
//   k_hay_prefix                                       -> per-haystack prefix array + batch summary
//   k_expand                                           slab events -> records (callback order)
//   k_refill                                           items whose events overflowed the slab re-scanned
//                                                      straight into their final slots (dense inputs)
// k_scan_simple (plain global-table walk) is the cross-check.  Every record comes out of the same exact
// table walk (exact_range); the scan kernels only differ in how they find the bytes worth walking.
//
// No MFMA anywhere: this is byte/integer indexing, bounded by HBM reads of the haystack (anchor
// path) or by the LDS's random-lookup rate (table walk).  Wave64, 1024-thread workgroups.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.hpp"
#include "kernels.hpp"
#include "scan_asm.inc"

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

__device__ __forceinline__ bool is_final(const DevAutomaton &A, uint32_t s) {
    return (s >= A.f1 && s < A.trap) || s >= A.f2;
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
    if (n < B.slab_events) B.slab[uint64_t(n) * B.run_stride + item] = make_uint2(pos, s);
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
        const uint32_t w = *reinterpret_cast<const E *>(tile + s * A.row_bytes + coff);
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

// item_base[h] = first work item of haystack h (h = 0..nh; item_base[nh] = number of items), the item table and
// the zeroed per-item counters.  Validates what the host cannot see for device-resident offsets: decreasing
// offsets, a haystack of 4 GiB or more (positions are 32-bit) or one that ends past the blob (every per-item
// workspace is sized from blob_bytes) raise *bad_offsets (k_expand hands it to the host as counters[6]), and that haystack gets no
// item.  The kernel touches nothing of the batch's sync block: it may run on another stream, ahead of the batch before.
__global__ __launch_bounds__(kScanThreads) void k_items(BatchArgs B, uint64_t blob_bytes, uint64_t *item_base_out, uint32_t *ticket,
                                                        uint32_t *bad_offsets, uint64_t *status, bool fill_later) {
    __shared__ uint64_t wave_sums[kScanThreads / 64];
    __shared__ uint64_t s_before;
    __shared__ uint32_t s_tile;
    __shared__ uint64_t local[kScanTile + 1];  // exclusive item prefix inside the tile
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t n = uint64_t(B.n_haystacks) + 1;  // elements: one per haystack + the closing one
    const uint64_t base = uint64_t(tile) * kScanTile + uint64_t(threadIdx.x) * kScanPerThread;
    uint64_t v[kScanPerThread];
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        const uint64_t h = base + i;
        v[i] = 0;
        if (h + 1 < n) {
            const uint64_t lo = B.offsets[h], hi = B.offsets[h + 1];
            if (hi < lo || hi - lo >= 0xFFFFFFFFull || hi > blob_bytes) *bad_offsets = 1u;
            else v[i] = (hi - lo + B.chunk_bytes - 1) / B.chunk_bytes;
        }
    }
    uint64_t tile_sum;
    uint64_t run = tile_exclusive(v, wave_sums, tile_sum);
    if (threadIdx.x < 64) {
        const uint64_t bf = chain_exclusive(status, tile, tile_sum);
        if (threadIdx.x == 0) s_before = bf;
    }
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        local[threadIdx.x * kScanPerThread + i] = run;
        run += v[i];
    }
    if (threadIdx.x == kScanThreads - 1) local[kScanTile] = run;
    __syncthreads();
    const uint64_t before = s_before;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i)
        if (base + i < n) item_base_out[base + i] = before + local[threadIdx.x * kScanPerThread + i];
    // the tile's work items, shared out over the workgroup whatever their distribution over its haystacks (when haystacks
    // are long -- many items each -- a tile of 2,048 of them is too much for one workgroup: k_item_fill does it instead)
    if (fill_later) return;
    for (uint64_t k = threadIdx.x; k < tile_sum; k += kScanThreads) {
        uint32_t lo = 0, hi = kScanTile;  // local[lo] <= k < local[hi]
        // one item per haystack so far (the usual batch): haystack k, two LDS reads instead of eleven
        if (k < kScanTile && local[k] == k && local[k + 1] == k + 1) { lo = uint32_t(k); hi = lo + 1; }
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (local[mid] <= k) lo = mid; else hi = mid;
        }
        fill_item(B, uint64_t(tile) * kScanTile + lo, k - local[lo], before + k);
    }
}

// The item table when haystacks are long (4,096 haystacks of 1 MiB: 129 items each -- two workgroups of k_items took
// 1 ms to write 528,384 records): one thread per item, its haystack found by binary search in item_base.
__global__ __launch_bounds__(256) void k_item_fill(BatchArgs B) {
    const uint64_t items = B.item_base[B.n_haystacks];
    for (uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; g < items; g += uint64_t(gridDim.x) * blockDim.x) {
        uint64_t lo = 0, hi = B.n_haystacks;  // item_base[lo] <= g < item_base[hi]; the LAST such lo (empty haystacks share a base)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (B.item_base[mid] <= g) lo = mid; else hi = mid;
        }
        fill_item(B, lo, g - B.item_base[lo], g);
    }
}

uint32_t scan_tiles(uint64_t n) { return uint32_t((n + kScanTile - 1) / kScanTile); }

// ------------------------------------------------------------------------------------------
// k_scan_simple: one thread per work item, every lookup from the global table.  The plain,
// obviously-correct form; also what the LDS kernel degenerates to for states outside the tile.
// ------------------------------------------------------------------------------------------
template <typename E>
__global__ void k_scan_simple(DevAutomaton A, BatchArgs B) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    ItemDesc it;
    if (!decode_item(B, i, it)) return;
    const uint8_t *p = B.blob + (it.begin - it.warm);
    exact_bytes<E>(A, B, nullptr, i, p, 0, it.warm + it.len, it.warm, it.pos0, 0u);
    finish_item(B, i);
}

// ------------------------------------------------------------------------------------------
// k_scan_lds: the hot kernel.
//
// * Workgroup = 1024 threads (16 waves, 4 per SIMD), one workgroup per CU; it stages the hot
//   tile of the transition table (up to ~160 KiB: every state within the first few trie
//   levels) into LDS once and then streams its work items.
// * Each lane owns ILP independent work items (streams): the per-byte step is a dependent
//   LDS lookup (~100+ cycles issue-to-use under load), so independent chains per lane are what
//   fills the SIMD besides the 4 waves it can hold at this LDS footprint.
// * A lane reads its haystack 64 bytes (4 x 16-byte loads, issued back to back) at a time into
//   one of two register buffers; the group after the current one is in flight while the
//   current one is scanned, so HBM latency hides behind 64 byte-steps.  Loads are
//   unconditional (index clamped to the stream's last quad) so the compiler can count them
//   with s_waitcnt vmcnt(N) instead of draining the queue.
// * The kernel is PURELY optimistic and records no events itself.  16 bytes (a quad) are pushed
//   through the LDS tile with no branch at all (per byte: extract the column, one multiply-add
//   for the address, one ds_read, one max).  Transitions that would leave the tile land on an
//   absorbing TRAP row and final states are numbered >= f1, so "max state seen >= f1" after the
//   quad says it holds an event or lost the true state.  Such a quad is appended to the item's
//   run list for the dense epilogue, which re-walks just those bytes exactly (with the (Lmax-1)-byte
//   warm-up that makes a mid-text restart exact).  A stream that hit the trap restarts from the
//   root and keeps flagging the next ceil(Lmax/16) quads: after Lmax bytes a walk from the root
//   is in the true state again, and until then the epilogue owns the events.  SIMT cost of a rare
//   event is therefore one short divergent append, not a 16-step replay by the whole wave.
// * SWAR3 mode maps 4 haystack bytes to 4 table columns with a few VALU ops when the pattern
//   alphabet has <= 8 symbols distinguishable by a 3-bit window (abcdef, ACGT, ...):
//   column = (byte >> shift) & 7, validated against the expected byte of that bucket with one
//   v_perm_b32; bytes outside the alphabet are redirected to the "other" column.  TABLE mode
//   reads a 256-entry column-offset table from LDS instead.
// ------------------------------------------------------------------------------------------
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
    uint32_t cap;        // slots: 128 (2 KiB per wave), or 48 in the big-set kernel, whose probe queue shares the 2 KiB
};
constexpr uint32_t kStageSlots = 128;
constexpr uint32_t kStageSlotsBig = 48;   // 768 B behind 1 KiB of probe queue and 256 B of hit masks
constexpr uint32_t kProbeQueue = 128;     // entries {hash, window | lane << 5} per wave

// `entry` of a run: the exact automaton state at `lo` (< 2^24), or
constexpr uint32_t kEntryUnknown = 0xFFFFFFFFu;    // walk from the root Lmax-1 bytes before lo
// ... or, from k_scan_anchor, kEntryWindow | start of the hit window the run belongs to (item scan coordinates):
// every pattern anchored there ends in [start + m_min - 4, start + Lmax); the stored `lo` is that lower bound unless
// an earlier window's run reached further (then it was clamped to that run's end: the two are adjacent).  Runs
// are never merged, one per hit window: k_walk_runs sees which runs touch and what every run was flagged for.
constexpr uint32_t kEntryWindow = 0x80000000u;
__device__ __forceinline__ bool entry_is_window(uint32_t e) { return (e & kEntryWindow) != 0u && e != kEntryUnknown; }

__device__ __forceinline__ void stage_flush(const BatchArgs &B, uint64_t wave_first, RunStage &st, uint32_t lane) {
    for (uint32_t i = lane; i < st.count; i += 64u) {
        const uint4 e = st.slots[i];
        B.runs[uint64_t(e.x >> 8) * B.run_stride + wave_first + (e.x & 0xffu)] = make_uint4(e.y, e.z, e.w, 0u);
    }
    st.count = 0;
}

constexpr uint32_t kDenseStage = 256;   // 8-byte entries in the wave's 2 KiB
__device__ __forceinline__ void dense_flush(const BatchArgs &B, uint64_t wave_first, RunStage &st, uint32_t lane) {
    const uint2 *slots2 = reinterpret_cast<const uint2 *>(st.slots);
    uint32_t *runs32 = reinterpret_cast<uint32_t *>(B.runs);
    for (uint32_t i = lane; i < st.count; i += 64u) {
        const uint2 e = slots2[i];
        runs32[(wave_first + (e.x & 0xffu)) * uint64_t(B.run_cap) + (e.x >> 8)] = e.y;
    }
    st.count = 0;
}

// Bookkeeping as in append_run (n, last_hi stay in scalar registers), but runs are never merged: adjacent runs
// stay separate entries (k_walk_runs joins them); k = item index inside the wave batch
__device__ __forceinline__ void append_run_wave(const BatchArgs &B, uint64_t wave_first, RunStage &st, uint32_t k,
                                                uint32_t lo, uint32_t hi, uint32_t entry, uint32_t &n,
                                                uint32_t &last_hi, uint32_t lane) {
    if (n > B.run_cap) return;
    if (n < B.run_cap) {
        if (st.count == st.cap) stage_flush(B, wave_first, st, lane);
        if (lane == 0) st.slots[st.count] = make_uint4(k | (n << 8), lo, hi, entry);
        ++st.count;
    }
    ++n;
    last_hi = hi;
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
    return cls_lds ? cls_lds[byte] : A.class_off[byte];
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
    uint4 cur = *reinterpret_cast<const uint4 *>(blk), nxt = cur;
    if (blk + 16 < end) nxt = *reinterpret_cast<const uint4 *>(blk + 16);
#pragma unroll 1
    for (uint32_t j = j0; j < j1; ++j) {
        const uintptr_t a = base + j;
        if (j != j0 && (a & 15u) == 0) {
            cur = nxt;
            blk += 16;
            if (blk + 16 < end) nxt = *reinterpret_cast<const uint4 *>(blk + 16);
        }
        const uint32_t sel = uint32_t(a >> 2) & 3u;
        const uint32_t word = sel == 0 ? cur.x : sel == 1 ? cur.y : sel == 2 ? cur.z : cur.w;
        const uint32_t byte = (word >> (8u * uint32_t(a & 3u))) & 0xffu;
        const uint32_t coff = class_offset<E, MODE>(A, cls_lds, byte);
        uint32_t w = A.trap;
        if (s < tile_rows) w = *reinterpret_cast<const E *>(tile + s * A.row_bytes + coff);  // LDS-resident row
        if (w >= A.trap) w = cold_step<E>(A, s, byte, coff);  // row not in LDS
        s = w;
        if (j >= rep_from && is_final(A, s) && !(B.first_only && ev.n_ev)) {
            ev.n_rec += A.out_count[s];
            if (store && ev.n_ev < B.slab_events) B.slab[uint64_t(ev.n_ev) * B.run_stride + item] = make_uint2(pos_bias + j, s);
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
        uint32_t w = *reinterpret_cast<const E *>(tile + row * row_bytes + coff);
        w = s < tile_rows ? w : trap;
        if (m == 0u && w >= trap) w = cold_step<E>(A, s, byte, coff);
        const bool stepped = m == 0u;
        s = stepped ? w : s + m;
        if (stepped && j >= rep_from && is_final(A, s) && !(B.first_only && ev.n_ev)) {
            ev.n_rec += A.out_count[s];
            if (ev.n_ev < B.slab_events) B.slab[uint64_t(ev.n_ev) * B.run_stride + item] = make_uint2(pos_bias + j, s);
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
            if (ev.n_ev < B.slab_events) B.slab[uint64_t(ev.n_ev) * B.run_stride + item] = make_uint2(pos_bias + j, s);
            ++ev.n_ev;
        }
    }
}

// run_epilogue for k_walk_runs.  k_scan_anchor leaves one run per hit window, ascending and disjoint, a run that
// begins where its predecessor ends being adjacent to it.  Runs that touch form a CLUSTER: everything that ends
// inside a cluster is an occurrence of a pattern anchored at one of the cluster's windows (an occurrence anchored
// elsewhere would have flagged a run of its own, overlapping this range, hence in the cluster).  So a cluster of
// window runs is decided by verifying its windows (verify_candidate) and emitting the occurrences found in
// position order -- of two that end at the same byte only the longer is an event, the shorter is in its output
// list.  Clusters with a run that names no window (unaligned head, tail, a swallowed window), with a fingerprint slot
// that stands for more than 8 pairs, or with more than kClusterMax occurrences are walked as one range.
template <typename E, int MODE, bool EXTRA>
__device__ __forceinline__ uint32_t run_epilogue_fast(const DevAutomaton &A, const BatchArgs &B, const uint8_t *tile,
                                                      uint32_t tile_rows, uint64_t item, const ItemDesc &it, uint32_t nr,
                                                      uint32_t &fallback16) {
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
        const uint4 *runs = B.runs + item;
        uint32_t r = 0;
        uint4 run = nr ? runs[0] : make_uint4(0u, 0u, 0u, 0u);
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
                run = runs[uint64_t(r) * B.run_stride];
                if (run.x > hi) break;  // neither adjacent nor overlapping (the dense form does not clamp its runs): the next cluster
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

template <typename E, int MODE, int ILP, int G>
__global__ __launch_bounds__(kLdsThreads) void k_scan_lds(DevAutomaton A, BatchArgs B) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kTileOff = MODE == 1 ? 0u : 512u;  // TABLE mode: 256 x uint16 column offsets first
    const uint64_t total_items = B.item_base[B.n_haystacks];
    if (uint64_t(blockIdx.x) * (B.persistent ? 64u * ILP : kLdsThreads * ILP) >= total_items) return;  // no work

    {   // stage [column offsets +] hot tile into LDS
        const uint4 *src = static_cast<const uint4 *>(A.lds_image);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (A.lds_image_bytes + 15u) >> 4;
        for (uint32_t t = threadIdx.x; t < n16; t += kLdsThreads) dst[t] = src[t];
    }
    __syncthreads();
    // the hand-scheduled path addresses the tile with raw LDS offsets: the dynamic LDS block must
    // start at LDS address 0 (true while this kernel declares no static __shared__ object)
    if (uint32_t(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) uint8_t *)smem)) != 0u) __builtin_trap();
    const uint8_t *tile = smem + kTileOff;
    const uint32_t row_bytes = A.row_bytes;
    const uint32_t f1 = A.f1, trap = A.trap;

    // Persistent workgroups (one per CU, the tile is staged once): each WAVE pulls batches of
    // 64 x ILP consecutive work items from a global counter until none are left.  Waves drift out
    // of phase, so while one wave sits in its latency-bound exact epilogue the other 15 keep the
    // LDS and HBM busy, and the tail of the launch is one wave-batch, not one workgroup.
    uint32_t grab_next = 0, grab_left = 0;   // persistent mode: the batches taken with one atomic (BatchArgs::wave_grab)
    uint32_t walked_all = 0;                 // this lane's exactly-walked bytes over all its batches
    for (;;) {
    uint32_t batch_first = (blockIdx.x * (kLdsThreads / 64u) + (threadIdx.x >> 6)) * (64u * ILP);  // static split
    if (B.persistent) {
        if (grab_left == 0) {
            if ((threadIdx.x & 63u) == 0) grab_next = atomicAdd(&B.counters[1], 64u * ILP * (B.wave_grab ? B.wave_grab : 1u));
            grab_next = __builtin_amdgcn_readfirstlane(grab_next);
            grab_left = B.wave_grab ? B.wave_grab : 1u;
        }
        batch_first = grab_next;
        grab_next += 64u * ILP;
        --grab_left;
    }
    if (batch_first >= total_items) break;
    const uint64_t item0 = uint64_t(batch_first) + uint64_t(threadIdx.x & 63u) * ILP;

    // per-stream state kept across the main loop: 7 VGPRs + the line buffers
    uint32_t s[ILP], nq[ILP], last[ILP], head[ILP], poison[ILP], nrun[ILP], run_hi[ILP];
    const uint4 *qp[ILP];
    uint32_t nq_max = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        ItemDesc it;
        s[k] = 0; nq[k] = 0; last[k] = 0; head[k] = 0; poison[k] = 0; nrun[k] = 0; run_hi[k] = 0;
        qp[k] = static_cast<const uint4 *>(A.lds_image);  // any readable 16-byte aligned address
        if (lds_item(B, item0 + k, it)) {
            const uint8_t *p = B.blob + (it.begin - it.warm);
            const uint32_t n = it.warm + it.len;
            const uint32_t mis = uint32_t(-reinterpret_cast<uintptr_t>(p)) & 15u;
            head[k] = mis < n ? mis : n;
            if (head[k]) {  // unaligned head: exact state walk here, events left to the epilogue
#pragma unroll 1
                for (uint32_t j = 0; j < head[k]; ++j) s[k] = exact_step<E>(A, tile, s[k], p[j]);
                append_run(B, item0 + k, 0, head[k], 0u, nrun[k], run_hi[k]);  // the scan range starts at the root
                if (s[k] >= trap) { s[k] = 0; poison[k] = B.poison_quads; }
                nq[k] = (n - head[k]) >> 4;
                head[k] |= 16u << 16;  // (every quad is whole; the tail is a run of its own)
            } else if (B.align_warm == 2u) {
                // Aligned item (k_items, align_warm 2: the scan begins on a 128-byte line, so that a lane's 8-quad groups
                // are whole lines -- unaligned, every line was fetched by two groups and ragged batches ran at half the
                // rate).  The `floor` bytes in front of a haystack that starts unaligned are another haystack's and the
                // bytes behind the item's end likewise: both are replaced by a byte no pattern uses before they reach
                // the automaton (every state goes to the root on it), so the item needs neither a head walked exactly
                // nor a tail flagged.  head[k] = 0 | floor << 8 | valid bytes of the last quad << 16.
                nq[k] = (n + 15u) >> 4;
                head[k] = it.floor << 8 | (((n - 1u) & 15u) + 1u) << 16;
            } else {
                nq[k] = n >> 4;
                head[k] = 16u << 16;
            }
            if (nq[k]) { qp[k] = reinterpret_cast<const uint4 *>(p + (head[k] & 0xffu)); last[k] = nq[k] - 1; }
            nq_max = nq[k] > nq_max ? nq[k] : nq_max;
        }
    }

    uint4 buf0[ILP][G], buf1[ILP][G];
    auto load_group = [&](uint4 (&buf)[ILP][G], uint32_t g) {
#pragma unroll
        for (int k = 0; k < ILP; ++k)
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const uint32_t q = g * G + r;
                buf[k][r] = qp[k][q < last[k] ? q : last[k]];  // plain load: a lane takes its 128-byte line in 8 pieces, and
                                                               // non-temporal lines are not kept for the later ones (2.3x slower)
            }
    };
    auto scan_group = [&](const uint4 (&buf)[ILP][G], uint32_t g) {
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const uint32_t q = g * G + r;
            if (q < nq_max) {
                uint32_t s_in[ILP], m[ILP];
#pragma unroll
                for (int k = 0; k < ILP; ++k) { s_in[k] = s[k]; m[k] = 0; }
                // ---- optimistic 16 byte-steps per stream, branch-free ----
                if constexpr (MODE == 1 && sizeof(E) == 2 && (ILP == 1 || ILP == 2)) {
                    // hand-scheduled gfx950 sequence (scan_asm.inc): streams interleaved per step with
                    // counted lgkmcnt waits, address = one v_mad_u32_u24; tile starts at LDS address 0
                    uint32_t w[ILP][4], t0[ILP], t1[ILP];
#pragma unroll
                    for (int k = 0; k < ILP; ++k) {
                        const uint4 cur = masked_quad(A, buf[k][r], q, head[k], last[k]);
                        w[k][0] = swar_columns<E>(A, cur.x);
                        w[k][1] = swar_columns<E>(A, cur.y);
                        w[k][2] = swar_columns<E>(A, cur.z);
                        w[k][3] = swar_columns<E>(A, cur.w);
                    }
                    if constexpr (ILP == 1) ACMI_SCAN_QUAD_ASM_1(t0, t1, m, s, w, row_bytes);
                    else ACMI_SCAN_QUAD_ASM_2(t0, t1, m, s, w, row_bytes);
#pragma unroll
                    for (int k = 0; k < ILP; ++k) s[k] = t1[k];
                } else {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t o[ILP][4];
#pragma unroll
                        for (int k = 0; k < ILP; ++k) {
                            const uint4 cur = masked_quad(A, buf[k][r], q, head[k], last[k]);
                            const uint32_t d = w == 0 ? cur.x : w == 1 ? cur.y : w == 2 ? cur.z : cur.w;
                            if (MODE == 1) {
                                const uint32_t o4 = swar_columns<E>(A, d);
#pragma unroll
                                for (int b = 0; b < 4; ++b) o[k][b] = __builtin_amdgcn_ubfe(o4, 8 * b, 8);
                            } else {
                                const uint16_t *coff = reinterpret_cast<const uint16_t *>(smem);
#pragma unroll
                                for (int b = 0; b < 4; ++b) o[k][b] = coff[__builtin_amdgcn_ubfe(d, 8 * b, 8)];
                            }
                        }
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
#pragma unroll
                            for (int k = 0; k < ILP; ++k) {
                                const uint32_t addr = __umul24(s[k], row_bytes) + o[k][b];
                                s[k] = *reinterpret_cast<const E *>(tile + addr);
                                m[k] = s[k] > m[k] ? s[k] : m[k];
                            }
                        }
                    }
                }
                // ---- rare: hand the quad to the epilogue ----
#pragma unroll
                for (int k = 0; k < ILP; ++k) {
                    if (q < nq[k]) {
                        if (m[k] >= f1 || poison[k] != 0) {
                            // (clipped to the item's own bytes: below `floor` the automaton saw masked bytes and is in the root)
                            const uint32_t fl = (head[k] >> 8) & 0xffu;
                            uint32_t lo = (head[k] & 0xffu) + (q << 4), hi = lo + (q == last[k] ? head[k] >> 16 : 16u);
                            uint32_t entry = poison[k] ? 0xFFFFFFFFu : s_in[k];
                            if (lo < fl) { lo = fl; entry = poison[k] ? 0xFFFFFFFFu : 0u; }
                            if (lo < hi) append_run(B, item0 + k, lo, hi, entry, nrun[k], run_hi[k]);
                            if (s[k] >= trap) { s[k] = 0; poison[k] = B.poison_quads; }  // lost the state: restart
                            else if (poison[k]) --poison[k];
                        }
                    } else {
                        s[k] = s_in[k];
                    }
                }
            }
        }
    };

    load_group(buf0, 0);
    for (uint32_t g = 0; g * G < nq_max; g += 2) {
        load_group(buf1, g + 1);
        scan_group(buf0, g);
        load_group(buf0, g + 2);
        scan_group(buf1, g + 1);
        // dense-match or mostly-cold inputs: once every stream of the wave has overflowed its run list
        // (its whole item will be walked exactly anyway) the optimistic pass is pure waste -- leave it
        bool alive = false;
#pragma unroll
        for (int k = 0; k < ILP; ++k) alive |= nrun[k] <= B.run_cap && (g + 2) * G < nq[k];
        if (!__any(alive)) break;
    }

    // ---- dense epilogue: every lane now walks the flagged runs of its own items exactly ----
    uint32_t walked = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        ItemDesc it;
        if (lds_item(B, item0 + k, it)) {
            const uint32_t n = it.warm + it.len;
            const uint32_t tail_lo = (head[k] & 0xffu) + (nq[k] << 4);
            if (tail_lo < n) append_run(B, item0 + k, tail_lo, n, poison[k] ? 0xFFFFFFFFu : s[k], nrun[k], run_hi[k]);
            walked += run_epilogue<E, MODE>(A, B, MODE == 1 ? nullptr : reinterpret_cast<const uint16_t *>(smem), tile, trap,
                                            item0 + k, it, nrun[k]);
        }
    }
    walked_all += walked;
    if (!B.persistent) break;
    }  // next batch of this wave
    account_walked(B, walked_all);   // (once per wave, not per batch: same-address device atomics retire at ~ 12 ns each)
}

// ------------------------------------------------------------------------------------------
// k_scan_anchor: sampled q-gram front end for pattern sets whose SHORTEST pattern is long.
//
// The per-byte table walk of k_scan_lds is bound by the LDS's random-lookup rate (one dependent
// ds_read per byte).  When every pattern is at least q + 3 bytes long (q = 8, 12 or 16) the walk
// can be skipped almost everywhere: an occurrence starting at text address x contains the
// 4-byte-aligned dword at ceil4(x), i.e. within its first 4 bytes, and the q bytes from there are
// one of <= 4 q-grams of that pattern (offsets 0..3).  So it suffices to test, for every ALIGNED
// text dword, whether the q bytes starting there are such an "anchor" q-gram: one hash + one
// ds_read_b64 (a 4-slot bucket of 16-bit fingerprints, 128 KiB table in LDS) per 4 bytes instead
// of four dependent ds_read_u16.  No false negatives by construction; a hit (true anchor or
// fingerprint collision) only flags the byte run in which a match could END, and k_walk_runs walks
// it through the Aho-Corasick table with the same exact walk k_scan_lds uses -- results are
// bit-identical.  Windows are independent (no state), so a wave can stream one item with fully
// coalesced non-temporal loads and their LDS reads pipeline freely; the kernel is bound by HBM.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kAnchorLdsBytes = 16384u * 8u;


// Probe one window: bucket = bits 3..16 of the hash (so the LDS byte address is one v_and),
// fingerprint = its high half.  The bucket's four 16-bit slots minus the fingerprint (packed
// subtract, the fingerprint broadcast by op_sel) are folded into a running packed minimum: some
// halfword of `z` is zero afterwards iff a slot equalled the fingerprint.
// The table sits at LDS address 0 (the kernel has no other LDS object and checks it), so the masked
// hash IS the ds_read address; going through the generic `smem` pointer costs a v_add per window.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2 *LdsU2Ptr;
typedef const __attribute__((address_space(3))) uint32_t *LdsU1Ptr;
__device__ __forceinline__ u32x2 anchor_bucket(uint32_t h) { return *reinterpret_cast<LdsU2Ptr>(h & 0x1FFF8u); }
__device__ __forceinline__ uint32_t anchor_compare(u32x2 b, uint32_t h, uint32_t z) {
    uint32_t x, y;
    asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x) : "v"(b.x), "v"(h));
    asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(y) : "v"(b.y), "v"(h));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(z) : "v"(z), "v"(x));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(z) : "v"(z), "v"(y));
    return z;
}
__device__ __forceinline__ bool has_zero_half(uint32_t z) { return ((z - 0x00010001u) & ~z & 0x80008000u) != 0; }

// One wave streams one work item at a time, 8 KiB per step, with fully coalesced loads (lane l holds
// quads r*64 + l).  Windows that begin in the previous quad get its last hashed dwords from the
// neighbouring lane through a DPP wave shift.  A wave owns 64 consecutive items; lane k remembers
// item k's run count, handed to k_walk_runs when the batch is done.
struct AnchorCursor {      // wave-uniform position in the wave's (item, chunk) task list
    uint32_t k;            // item index inside the wave, 0..63
    uint32_t c, chunks;    // current 8-KiB chunk of the item, number of chunks (>= 1)
    uint32_t n, head, nq;  // scan bytes, unaligned head bytes, whole 16-byte quads after the head
    const uint4 *qp;       // first quad
    bool valid;
};

// `rec` = lane k's packed record of item wave_first + k (z == 0: no such item)
template <uint32_t TQ>  // quads per task (a wave step): 512 = 8 KiB, or 256 for the BIG filter
__device__ __forceinline__ void anchor_open(const DevAutomaton &A, const BatchArgs &B, const uint4 &rec, uint32_t k,
                                            AnchorCursor &cur) {
    cur.k = k;
    cur.c = 0;
    cur.chunks = 1;
    cur.n = cur.head = cur.nq = 0;
    cur.qp = static_cast<const uint4 *>(A.anchor_table);
    cur.valid = false;
    if (k >= B.wave_items) return;
    ItemDesc it;
    const uint32_t z = __builtin_amdgcn_readlane(rec.z, k);
    if (z == 0) return;
    unpack_item(__builtin_amdgcn_readlane(rec.x, k), __builtin_amdgcn_readlane(rec.y, k), z, 0u, it);
    cur.valid = true;
    const uint8_t *p = B.blob + (it.begin - it.warm);
    cur.n = z;
    const uint32_t mis = uint32_t(-reinterpret_cast<uintptr_t>(p)) & 15u;
    cur.head = mis < cur.n ? mis : cur.n;
    // whole quads after the head; with aligned items (align_warm) also the last, partial one: its windows are tested like
    // any other (the bytes behind the item's end are another haystack's, or the rest of the blob's last aligned block),
    // what they yield past the end is clipped or rejected, and there is no tail left to flag wholesale
    cur.nq = (cur.n - cur.head + (B.align_warm && cur.head == 0u ? 15u : 0u)) >> 4;
    if (cur.nq) cur.qp = reinterpret_cast<const uint4 *>(p + cur.head);
    cur.chunks = cur.nq ? (cur.nq + TQ - 1u) / TQ : 1u;
}

// advance to the next (item, chunk) task of the wave; false when the wave's items are done
template <uint32_t TQ>
__device__ __forceinline__ bool anchor_next(const DevAutomaton &A, const BatchArgs &B, const uint4 &rec, AnchorCursor &cur) {
    if (!cur.valid) return false;
    if (++cur.c < cur.chunks) return true;
    anchor_open<TQ>(A, B, rec, cur.k + 1u, cur);
    return cur.valid;
}

// BIG (anchor mode 2, more q-grams than the LDS table holds): LDS carries a 2^20-bit map of the hashes'
// top 20 bits; a window whose bit is set (a fifth of them for 262,144 q-grams) is looked up in an
// 8-slot fingerprint table in global memory, which L2 serves.  QW = 1 (4-byte windows, shortest pattern
// 7..10 B) needs no hand-over between quads.
// SHORT (small-set kernel only): items may be shorter than a task -- the rows a task does not have are not hashed.
// A variant of its own because the test, one scalar branch per row, costs the 8-KiB-item kernel its schedule.
// TINY: a wave whose items are all at most one row (1 KiB, aligned: haystacks of a few hundred bytes) takes them G to a task,
// ROW r of the task = item t*G + r, instead of spending a whole step on each (512-byte haystacks: 8 x fewer steps).
// DENSE (QW == 1, small sets): pattern sets whose shortest pattern is 4..6 B (keyword lists) have no aligned dword inside
// every occurrence, so every pattern is anchored by its FIRST four bytes and the 4-byte window at EVERY byte offset is
// tested -- 16 windows per quad instead of 4, hashed raw (anchor_hash_dense: a shift and a multiply-add).  Hit windows are
// an order of magnitude more frequent than with aligned anchors (a keyword list: ~ 20 per 8 KiB of text), so they are
// turned into runs by VECTOR code -- a wave prefix sum over the lanes' hit counts gives every hit its slot in the item's
// run list -- not one by one through the scalar unit; runs keep their natural bounds (they may overlap: k_walk_runs joins
// them into clusters).
// HOT (dense form): many hit windows expected (a dozen or more per 8 KiB) -- which of a step's four windows hit is worked out
// for every step, branch-free (31 VALU), instead of a cheap any-hit test (21) with the exact one behind a branch (+ 35 when
// taken: with three hits per step that was most of the kernel's instructions).
// SL2 (dense form, few keys): the LDS table has two slots per bucket, 2^15 buckets -- one ds_read_b32 and one packed subtract
// per window.
template <int QW, bool BIG, bool SHORT, bool TINY = false, bool DENSE = false, bool HOT = false, bool SL2 = false>
__global__ __launch_bounds__(kLdsThreads) void k_scan_anchor(DevAutomaton A, BatchArgs B) {
    static_assert(!SL2 || DENSE, "two-slot buckets exist for the dense form only");
    static_assert(!DENSE || (QW == 1 && !BIG), "the dense form exists for 4-byte windows and the LDS table only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = BIG ? 4 : 8;  // 16-byte quads per lane and task
    constexpr uint32_t TQ = G * 64u;  // quads per task: 8 KiB, BIG 4 KiB (two tasks in flight in the registers of one)
    constexpr int NC = QW > 1 ? QW - 1 : 1;  // carries between quads (none for QW == 1; arrays keep one element)
    if (uint64_t(blockIdx.x) * (B.persistent ? 1u : kLdsThreads / 64u) * B.wave_items >= B.item_base[B.n_haystacks]) return;
    {   // LDS: the anchor table (or bitmap), 128 KiB; behind it 2 KiB of run staging per wave
        const uint4 *src = BIG ? reinterpret_cast<const uint4 *>(A.anchor_bitmap) : static_cast<const uint4 *>(A.anchor_table);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t t = threadIdx.x; t < kAnchorLdsBytes / 16u; t += kLdsThreads) dst[t] = src[t];
    }
    __syncthreads();
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)smem)) != 0u) __builtin_trap();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mul = A.anchor_mul;
    const uint32_t q = QW * 4u, m_min = A.min_pattern_length, l_max = A.max_pattern_length;
    const uint32_t lo_bias = DENSE ? m_min - 1u : (A.anchor_choice ? q - 1u : m_min - 4u);
    const uint64_t total_items = B.item_base[B.n_haystacks];

    // Persistent mode: waves pull batches of items from counters[1] until none are left, wave_grab batches per atomic.
    uint32_t grab_next = 0, grab_left = 0, tiny_batches = 0;
    for (;;) {
    // a wave takes `wave_items` (1..64) consecutive items per batch: 64 for big batches, fewer when
    // there are not enough items to give every wave of the chip 64 of them
    uint32_t batch_first = (blockIdx.x * (kLdsThreads / 64u) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * B.wave_items;
    if (B.persistent) {
        if (grab_left == 0) {
            if (lane == 0) grab_next = atomicAdd(&B.counters[1], B.wave_items * (B.wave_grab ? B.wave_grab : 1u));
            grab_next = __builtin_amdgcn_readfirstlane(grab_next);
            grab_left = B.wave_grab ? B.wave_grab : 1u;
        }
        batch_first = grab_next;
        grab_next += B.wave_items;
        --grab_left;
    }
    if (batch_first >= total_items) break;
    const uint64_t wave_first = batch_first;

    // lane k holds the packed record of item wave_first + k for the whole batch: opening an item is
    // three v_readlane and no memory access
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (lane < B.wave_items && wave_first + lane < total_items) rec = B.item_rec[wave_first + lane];
    // TINY: every item of this batch is one aligned row at most?  (wave-uniform; else the wave takes the usual path)
    bool tiny_on = false;
    uint32_t tcnt = 0, tscan = 0, tload = 0;  // items of the batch; task being scanned / being loaded (G x tpack items each)
    uint32_t tpack = 1, tshift = 6;           // items per row of a task, log2 of the lanes per item
    if constexpr (TINY) {
        const bool fits = rec.z == 0u || (((reinterpret_cast<uintptr_t>(B.blob) + rec.x) & 15u) == 0u && rec.z <= 1024u);
        tiny_on = B.align_warm == 1u && __all(fits);
        tcnt = uint32_t(__builtin_popcountll(__ballot(rec.z != 0u)));
        tiny_batches += tiny_on ? 1u : 0u;  // statistic ("tiny_waves"): which path the batch's waves took -- summed up at the kernel's end
        // items of a quarter / half row at most (haystacks of <= 256 / 512 B) are packed FOUR / TWO to a row: lane l of row r holds
        // quad l mod 16 (32) of item (8 t + r) * 4 (2) + l / 16 (32) -- else three lanes in four of a 256-byte item's row idle
        if (!BIG && !DENSE && tiny_on) {
            if (__all(rec.z <= 256u)) { tpack = 4u; tshift = 4u; }
            else if (__all(rec.z <= 512u)) { tpack = 2u; tshift = 5u; }
        }
    }
    AnchorCursor ld, sc;  // load cursor runs one task ahead of the scan cursor
    anchor_open<TQ>(A, B, rec, 0, ld);
    if (!ld.valid) break;
    sc = ld;
    // per wave, behind the table: 2 KiB of run staging -- BIG: [1 KiB probe queue | 256 B hit masks | 768 B run staging]
    uint8_t *wave_lds = smem + kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u;
    RunStage stage{reinterpret_cast<uint4 *>(wave_lds + (BIG ? 1280u : 0u)), 0u, BIG ? kStageSlotsBig : kStageSlots};
    uint2 *const queue = reinterpret_cast<uint2 *>(wave_lds);           // BIG only
    uint32_t *const hitmask = reinterpret_cast<uint32_t *>(wave_lds + 1024u);
    if (BIG) hitmask[lane] = 0u;
    uint32_t nrun = 0, run_hi = 0;  // run list of the item being scanned (wave-uniform)
    uint32_t nrun_vec = 0;          // lane k: run count of item wave_first + k
    // carry[n-1] = hash prefix over the LAST n dwords of a quad (n = 1..QW-1): what a window that
    // begins there and ends in the next quad continues from.  `tail` = the carries of the previous
    // wave-load (its lane 63 is the left neighbour of lane 0).
    uint32_t tail[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) tail[i] = 0;
    if (sc.head) {
        // (dense form: its run lists hold window starts only -- an item with an unaligned head, the first haystack of an unaligned
        // blob, is walked whole)
        if constexpr (DENSE) nrun = B.run_cap + 1u;
        else append_run_wave(B, wave_first, stage, sc.k, 0, (sc.head + l_max) < sc.n ? sc.head + l_max : sc.n, 0u, nrun, run_hi, lane);
    }

    uint4 buf0[G], buf1[G];
    // Lane l holds quads r*64 + l (r = 0..7) of the wave's current 8 KiB: every wave-load covers 1 KiB
    // of contiguous memory (8 whole 128-byte lines).  A micro-benchmark of the access patterns alone
    // (tools/micro/stream_pattern.hip) gives 4.6 TB/s when each lane reads its own 128-byte line --
    // the line is split over 8 wave-instructions of 64 different lines each -- against 6.4 TB/s coalesced.
    auto load_quad = [&](const AnchorCursor &cur, int r) -> uint4 {
        const uint4 *base = cur.qp;
        uint32_t last = cur.nq ? cur.nq - 1u : 0u;
        uint32_t qi = cur.c * TQ + uint32_t(r) * 64u + lane;
        if constexpr (TINY) {
            if (tiny_on) {  // row r of task `tload` = item tload * G + r (a row that is not there repeats the last item)
                if (tpack == 1u) {
                    uint32_t k = tload * uint32_t(G) + uint32_t(r);
                    k = k < tcnt ? k : tcnt - 1u;
                    const uint32_t x = __builtin_amdgcn_readlane(rec.x, k), y = __builtin_amdgcn_readlane(rec.y, k);
                    base = reinterpret_cast<const uint4 *>(B.blob + (uint64_t(y & 0x3FFFu) << 32 | x));
                    last = ((__builtin_amdgcn_readlane(rec.z, k) + 15u) >> 4) - 1u;
                    qi = lane;
                } else {  // several items per row: every lane fetches its own item's record from the lane that holds it
                    uint32_t k = (tload * uint32_t(G) + uint32_t(r)) * tpack + (lane >> tshift);
                    k = k < tcnt ? k : tcnt - 1u;
                    const uint32_t x = uint32_t(__shfl(int(rec.x), int(k), 64)), y = uint32_t(__shfl(int(rec.y), int(k), 64));
                    base = reinterpret_cast<const uint4 *>(B.blob + (uint64_t(y & 0x3FFFu) << 32 | x));
                    last = ((uint32_t(__shfl(int(rec.z), int(k), 64)) + 15u) >> 4) - 1u;
                    qi = lane & ((1u << tshift) - 1u);
                }
            }
        }
        return stream_load(base + (qi < last ? qi : last));  // (one load instruction either way: only the address differs)
    };
    auto load_task = [&](uint4 (&buf)[G], const AnchorCursor &cur) {
#pragma unroll
        for (int r = 0; r < G; ++r) buf[r] = load_quad(cur, r);
    };
    auto scan_task = [&](uint4 (&buf)[G]) {
        uint32_t z[G];  // per quad: bit w set = the window ENDING at dword w of the quad hit (written on hits only)
#pragma unroll
        for (int r = 0; r < G; ++r) z[r] = 0;
        uint32_t zany = 0;
        if (sc.c == 0) {  // first chunk of an item: nothing to the left of quad 0 (wave-uniform)
#pragma unroll
            for (int i = 0; i < NC; ++i) tail[i] = 0;
        }
        // hashes of the 4 windows ending in quad r
        auto hash_quad = [&](int r, uint32_t (&h4)[4]) {
            if constexpr (TINY && QW > 1) {
                if (tiny_on) {  // every row begins an item: nothing to the left of its quad 0
#pragma unroll
                    for (int i = 0; i < NC; ++i) tail[i] = 0;
                }
            }
            if constexpr (QW == 1) {  // lone dwords are hashed raw (automaton.hpp anchor_hash): a shift and a multiply-add
                h4[0] = anchor_mad(buf[r].x, mul, buf[r].x >> 13);
                h4[1] = anchor_mad(buf[r].y, mul, buf[r].y >> 13);
                h4[2] = anchor_mad(buf[r].z, mul, buf[r].z >> 13);
                h4[3] = anchor_mad(buf[r].w, mul, buf[r].w >> 13);
                return;
            }
            uint32_t f[4];
            f[0] = anchor_fold_dev(buf[r].x);
            f[1] = anchor_fold_dev(buf[r].y);
            f[2] = anchor_fold_dev(buf[r].z);
            f[3] = anchor_fold_dev(buf[r].w);
            uint32_t mine[NC], left[NC];
#pragma unroll
            for (int n = 1; n < QW; ++n) {  // prefix over this quad's last n dwords
                uint32_t h = f[4 - n];
#pragma unroll
                for (int i = 4 - n + 1; i < 4; ++i) h = anchor_mad(h, mul, f[i]);
                mine[n - 1] = h;
                // lane 0 <- lane 63 of the previous wave-load (rotate), lanes 1..63 <- lane l-1 of this one
                const uint32_t wrap = __builtin_amdgcn_mov_dpp(tail[n - 1], 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
                left[n - 1] = __builtin_amdgcn_update_dpp(wrap, h, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if constexpr (TINY) {  // packed rows: the lane that holds an item's first quad has nothing to its left
                    if (tpack > 1u && (lane & ((1u << tshift) - 1u)) == 0u) left[n - 1] = 0u;
                }
            }
#pragma unroll
            for (int n = 1; n < QW; ++n) tail[n - 1] = mine[n - 1];
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // window ending at dword w: QW-1-w dwords of the left quad, then f[..w]
                const int need = QW - 1 - w;
                uint32_t h;
                int i0;
                if (need > 0) { h = left[need - 1]; i0 = 0; } else { h = f[w - (QW - 1)]; i0 = w - (QW - 1) + 1; }
#pragma unroll
                for (int i = i0; i <= w; ++i) h = anchor_mad(h, mul, f[i]);
                h4[w] = h;
            }
        };
        const uint32_t bucket_mask = A.anchor_bucket_mask;
        const uint4 *big_table = static_cast<const uint4 *>(A.anchor_table);
        if constexpr (BIG) {
            // Big sets.  The LDS bitmap passes a fifth of all windows (262,144 q-grams in 2^20 bits); each of those
            // needs its 16-byte fingerprint bucket from the table behind L2.  Probing where the window is found --
            // four predicated loads and compares per quad, 14 of 64 lanes active in each, the wave waiting for every
            // quad's round trip -- made the kernel both VALU-bound (52 wave-instructions per window) and
            // latency-bound (2.1 ms per 4 GiB).  Instead the "maybes" of a wave are COMPACTED: {hash, position} goes
            // into a 128-entry LDS ring (ballot + mbcnt give the slot), and whenever 64 are queued the wave probes
            // them as one dense round: one 16-byte load and one compare per LANE, the load consumed one round later.
            // A fingerprint match sets the window's bit in the owner lane's hit mask (ds_or); the masks are read
            // back when the task is done and flagged in text order exactly as in the small-set kernel.
            uint32_t qhead = 0, qcount = 0;   // ring state (wave-uniform)
            uint32_t ph = 0, ppos = 0xFFFFFFFFu;  // the probe round in flight: hash, position (0xFFFFFFFF: idle lane)
            uint4 ptb = make_uint4(0u, 0u, 0u, 0u);
            bool pending = false;             // wave-uniform
            // (two rounds in flight were tried: no faster -- 1.71 against 1.75 ms per 4 GiB -- and the 12 registers
            // they hold spilled inside the loop: the kernel is bound by the L2's request rate, 70 M probes per GiB)
            auto consume = [&]() {
                if (!pending) return;
                uint32_t zz = 0xFFFFFFFFu;
                const uint32_t fp = ph | 0x10000u;  // fingerprint = (h >> 16) | 1: never the empty 0
                zz = anchor_compare(u32x2{ptb.x, ptb.y}, fp, zz);
                zz = anchor_compare(u32x2{ptb.z, ptb.w}, fp, zz);
                const bool hit = ppos != 0xFFFFFFFFu && has_zero_half(zz);
                if (__builtin_expect(__any(hit), 0)) {
                    if (hit) atomicOr(&hitmask[ppos >> 5], 1u << (ppos & 31u));
                }
                pending = false;
            };
            auto probe_round = [&]() {  // the oldest (up to) 64 queued windows
                consume();
                const uint32_t nq_ = qcount < 64u ? qcount : 64u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                ppos = 0xFFFFFFFFu;
                ph = 0;
                if (lane < nq_) {
                    const uint2 e = queue[(qhead + lane) & (kProbeQueue - 1u)];
                    ph = e.x;
                    ppos = e.y;
                }
                ptb = big_table[(ph >> A.anchor_bucket_shift) & bucket_mask];  // (idle lanes read bucket 0: no branch around the load)
                qhead += nq_;
                qcount -= nq_;
                pending = true;
            };
            // quads of this task that exist (an item of 4 KiB fills half a task: the rest is skipped, wave-uniform)
            uint32_t task_quads = sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ;
            if constexpr (TINY) {
                if (tiny_on) task_quads = (tcnt - tscan * uint32_t(G) < uint32_t(G) ? tcnt - tscan * uint32_t(G) : uint32_t(G)) * 64u;
            }
            // The bitmap words of quad r+1 are requested before quad r's are tested (sched_barrier keeps hipcc from
            // sinking the reads to their use): a wave never sits on the LDS round trip of its own four reads.
            typedef __attribute__((address_space(3))) u32x2 *RingPtr;
            const uint32_t ring_base = kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u;  // 1-KiB aligned: slot address = base | offset
            uint32_t hh[2][4], ww[2][4];
            auto fetch_quad = [&](int r) {
                hash_quad(r, hh[r & 1]);
#pragma unroll
                for (int w = 0; w < 4; ++w) ww[r & 1][w] = *reinterpret_cast<LdsU1Ptr>((hh[r & 1][w] >> 15) & 0x1FFFCu);
            };
            fetch_quad(0);
#pragma unroll
            for (int r = 0; r < G; ++r) {
                if (uint32_t(r) * 64u >= task_quads) break;
                if (r + 1 < G) fetch_quad(r + 1);  // (of a quad past the item's end: hashed, never tested)
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t *h4 = hh[r & 1], *wd = ww[r & 1];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // (one bit-field extract feeds both the ballot and the predicated append: written as a shift and a
                    // mask hipcc computed the bit twice, 7 VALU instead of 3)
                    const bool maybe = __builtin_amdgcn_ubfe(wd[w], __builtin_amdgcn_ubfe(h4[w], 12u, 5u), 1u) != 0u;
                    const uint64_t mm = __ballot(maybe);
                    if (mm != 0) {
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(mm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mm), 0u));
                        if (maybe) {
                            const uint32_t slot = ((((qhead + qcount) << 3) + (at << 3)) & (kProbeQueue * 8u - 8u)) | ring_base;
                            *reinterpret_cast<RingPtr>(slot) = u32x2{h4[w], uint32_t(r * 4 + w) | (lane << 5)};
                        }
                        qcount += uint32_t(__builtin_popcountll(mm));
                        if (qcount >= 64u) probe_round();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            while (qcount != 0) probe_round();
            load_task(buf, ld);  // the next task (issued behind the last probe round, which is consumed without waiting for it)
            consume();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t m = hitmask[lane];
            if (__builtin_expect(__any(m != 0), 0)) {
                hitmask[lane] = 0u;
#pragma unroll
                for (int r = 0; r < G; ++r) z[r] = (m >> (4 * r)) & 15u;
                zany = m;
            }
        } else if constexpr (DENSE) {
        // 16 windows per quad, four at a time: step t = 4 r + j takes dword j of row r -- the three windows that begin in
        // dword j-1 and end in dword j, and dword j itself -- i.e. the windows ENDING at bytes 4j .. 4j+3 of the quad (bit e
        // of z[r] = the window ending at byte e).  Same software pipeline as below: the four bucket reads of step t are in
        // flight while step t-1's are compared.  The dword to the left of a quad's first is the neighbouring lane's last
        // (DPP), lane 0's the last dword of lane 63 in the previous row (tail[0]: raw text here, not a hash prefix).
        uint32_t hh[2][4];
        u32x2 bb[2][4];   // (SL2: .x holds the bucket's two slots, .y a copy -- the compares below then see every slot twice)
        uint32_t rows = SHORT ? ((sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ) + 63u) >> 6 : uint32_t(G);
        if constexpr (TINY) {
            if (tiny_on) rows = tcnt - tscan * uint32_t(G) < uint32_t(G) ? tcnt - tscan * uint32_t(G) : uint32_t(G);
        }
        uint32_t left = 0;  // the dword in front of the current row's quad
#pragma unroll
        for (int t = 0; t <= 4 * G; ++t) {
            const int r = t >> 2, j = t & 3;
            if (t < 4 * G) {
                const bool live = !SHORT || uint32_t(r) < rows;
                if (j == 0 && live) {
                    if constexpr (TINY) {
                        if (tiny_on) tail[0] = 0;  // every row begins an item: nothing to the left of its quad 0
                    }
                    const uint32_t wrap = __builtin_amdgcn_mov_dpp(tail[0], 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
                    left = __builtin_amdgcn_update_dpp(wrap, buf[r].w, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    tail[0] = buf[r].w;
                }
                if (live) {
                    const uint32_t dj = j == 0 ? buf[r].x : j == 1 ? buf[r].y : j == 2 ? buf[r].z : buf[r].w;
                    const uint32_t dm = j == 0 ? left : j == 1 ? buf[r].x : j == 2 ? buf[r].y : buf[r].z;
                    uint32_t *h4 = hh[t & 1];
                    const uint32_t w1 = __builtin_amdgcn_alignbyte(dj, dm, 1), w2 = __builtin_amdgcn_alignbyte(dj, dm, 2), w3 = __builtin_amdgcn_alignbyte(dj, dm, 3);
                    h4[0] = anchor_mad(w1, mul, w1 >> 13);
                    h4[1] = anchor_mad(w2, mul, w2 >> 13);
                    h4[2] = anchor_mad(w3, mul, w3 >> 13);
                    h4[3] = anchor_mad(dj, mul, dj >> 13);
                }
                // the row is consumed: reload its registers with the same row of the task two steps ahead
                if (j == 3) buf[r] = load_quad(ld, r);
                if (live) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if constexpr (SL2) bb[t & 1][w].x = *reinterpret_cast<LdsU1Ptr>(hh[t & 1][w] & 0x1FFFCu);   // bucket = hash bits 2..16
                        else bb[t & 1][w] = anchor_bucket(hh[t & 1][w]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0 && (!SHORT || uint32_t((t - 1) >> 2) < rows)) {
                const u32x2 *b4 = bb[(t - 1) & 1];
                const uint32_t *h4 = hh[(t - 1) & 1];
                if constexpr (SL2) {
                    // two slots: the subtract's result is the whole test.  HOT: exact per window, every step; else a running
                    // minimum with the exact test behind a branch
                    if constexpr (HOT) {
                        // (written as plain compares hipcc emits two v_cmp_eq_u32_sdwa per window -- halfword operands, the verdict in a
                        // scalar register pair -- an s_or and a v_cndmask: 14 VALU + 4 SALU per step against 19 VALU for the packed
                        // subtract and zero-halfword test; filter 0.442 -> 0.431 ms per GiB of the keyword set)
                        uint32_t m = 0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const uint32_t fp = h4[w] >> 16;
                            const bool hit = ((b4[w].x & 0xffffu) == fp) | ((b4[w].x >> 16) == fp);
                            m |= hit ? (1u << w) : 0u;
                        }
                        z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                        zany |= m;
                    } else {
                        uint32_t x4[4], zz = 0xFFFFFFFFu;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x4[w]) : "v"(b4[w].x), "v"(h4[w]));
                            asm("v_pk_min_u16 %0, %1, %2" : "=v"(zz) : "v"(zz), "v"(x4[w]));
                        }
                        if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                            uint32_t m = 0;
#pragma unroll
                            for (int w = 0; w < 4; ++w) m |= has_zero_half(x4[w]) ? (1u << w) : 0u;
                            z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                            zany |= m;
                        }
                    }
                } else if constexpr (HOT) {
                    // per window: the bucket's four slots minus the fingerprint, their minimum per halfword pair, the
                    // zero-halfword test (bits 15 / 31), shifted to the window's own bit positions and merged
                    uint32_t dd = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t x, y, tm;
                        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x) : "v"(b4[w].x), "v"(h4[w]));
                        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(y) : "v"(b4[w].y), "v"(h4[w]));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(tm) : "v"(x), "v"(y));
                        const uint32_t d = (tm - 0x00010001u) & ~tm;
                        dd |= (d >> (3 - w)) & (0x80008000u >> (3 - w));   // window w at bits 12 + w and 28 + w
                    }
                    const uint32_t m = __builtin_amdgcn_ubfe(dd | (dd >> 16), 12u, 4u);   // bit w: window w hit
                    z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                    zany |= m;
                } else {
                uint32_t zz = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 4; ++w) zz = anchor_compare(b4[w], h4[w], zz);
                if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                    uint32_t m = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) m |= has_zero_half(anchor_compare(b4[w], h4[w], 0xFFFFFFFFu)) ? (1u << w) : 0u;
                    z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                    zany |= m;
                }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
        // Software pipeline over the quads, pinned with sched_barriers: the four bucket reads of quad r
        // are in flight while quad r-1's buckets are compared, so a wave never sits on LDS latency
        // (hipcc on its own put an lgkmcnt(0) right behind every ds_read).
        uint32_t hh[2][4];
        u32x2 bb[2][4];   // the four fingerprint buckets of a quad
        // SHORT: rows of 64 quads this task has (an item of 4 KiB fills half a task); the others are only reloaded
        uint32_t rows = SHORT ? ((sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ) + 63u) >> 6 : uint32_t(G);
        if constexpr (TINY) {
            if (tiny_on) {
                const uint32_t left_items = tcnt - tscan * uint32_t(G) * tpack;
                rows = (left_items + tpack - 1u) / tpack < uint32_t(G) ? (left_items + tpack - 1u) / tpack : uint32_t(G);
            }
        }
#pragma unroll
        for (int r = 0; r <= G; ++r) {
            if (r < G) {
                if (!SHORT || uint32_t(r) < rows) hash_quad(r, hh[r & 1]);
                // the quad is consumed: reload its registers with the same quad of the task two steps
                // ahead at once, so that the wave keeps close to 16 KiB of loads in flight
                buf[r] = load_quad(ld, r);
                if (!SHORT || uint32_t(r) < rows) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) bb[r & 1][w] = anchor_bucket(hh[r & 1][w]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r > 0 && (!SHORT || uint32_t(r - 1) < rows)) {
                const u32x2 *b4 = bb[(r - 1) & 1];
                const uint32_t *h4 = hh[(r - 1) & 1];
                uint32_t zz = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 4; ++w) zz = anchor_compare(b4[w], h4[w], zz);
                // Rare, wave-uniform: some lane's quad hit.  Which of its four windows is worked out here,
                // while the buckets are still in registers: k_walk_runs then knows the window of every run.
                if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                    uint32_t m = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) m |= has_zero_half(anchor_compare(b4[w], h4[w], 0xFFFFFFFFu)) ? (1u << w) : 0u;
                    z[r - 1] = m;
                    zany |= m;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // Rare, wave-uniform: flag where a match anchored at a hit window can end.  A pattern whose first
        // aligned dword is the window starting at byte `start` begins in [start-3, start] and ends at byte
        // index [start + m_min - 4, start + Lmax).
        auto flag = [&](uint32_t r, uint32_t l, uint32_t wmask) {
            const uint32_t qi = sc.c * TQ + r * 64u + l;
            if (qi >= sc.nq) return;
            for (; wmask != 0; wmask &= wmask - 1) {
                const uint32_t w = uint32_t(__builtin_ctz(wmask));
                // the window ending at dword w of quad qi starts QW-1 dwords earlier
                const int32_t start = int32_t(sc.head + (qi << 4) + 4u * w) - int32_t(q - 4u);
                // the end of a pattern anchored here: at least m_min - 4 past the window start when the window is the
                // pattern's FIRST aligned dword, at least q - 1 when the builder chose a deeper window (anchor_choice)
                int32_t lo = start + int32_t(lo_bias);
                uint32_t hi = uint32_t(start + int32_t(l_max));
                if (lo < 0) lo = 0;
                if (hi > sc.n) hi = sc.n;
                // a window at the item's very end or in the padding of its last 16-byte block: whatever is anchored there
                // ends behind the item.  (Not the swallowed-window case below: marking the previous run for a walk on
                // every such accidental hit put one slow lane into almost every workgroup of k_walk_runs -- 0.65 instead
                // of 0.16 ms on 4,001-byte haystacks.)
                if (uint32_t(lo) >= sc.n) continue;
                // one run per hit window, tagged with the window's start; where an earlier window's run reaches
                // into this one's range the run starts at that run's end (disjoint, ascending, adjacent)
                if (nrun > 0 && uint32_t(lo) < run_hi) lo = int32_t(run_hi);
                // (a window that begins before the item's first quad was hashed over zeros: not a real window)
                const uint32_t entry = start >= 0 ? (kEntryWindow | uint32_t(start)) : kEntryUnknown;
                if (uint32_t(lo) < hi) {
                    append_run_wave(B, wave_first, stage, sc.k, uint32_t(lo), hi, entry, nrun, run_hi, lane);
                } else if (nrun > 0 && nrun <= B.run_cap) {
                    // its whole range lies inside the previous run (both end with the item): that run now stands for
                    // a window it does not name and must be walked.  The previous run is the last staged entry:
                    // the stage is only flushed right before an append.
                    if (lane == 0) stage.slots[stage.count - 1].w = kEntryUnknown;
                }
            }
        };
        if constexpr (DENSE) {
            // Hits -> runs by vector code.  Per row (its quads ascend with the lane, so (row, lane, byte) order is text
            // order): a lane keeps the hit windows that can hold a pattern at all -- they begin inside the item and a pattern
            // of the shortest length would end inside it -- counts them, a wave prefix sum gives each its index in the
            // item's run list, and the lanes write their own entries into the LDS stage (flushed to the run lists in bulk).
            // Runs are not clamped against each other: [start + m_min - 1, start + Lmax) each, possibly overlapping.
            bool by_row = __any(zany != 0);
            bool tiny_rows_now = false;
            if constexpr (TINY) tiny_rows_now = tiny_on;
            if (by_row && !tiny_rows_now) {
                // One item per task (the usual case): the hits of all eight rows are placed with THREE wave prefix sums -- the
                // lanes' hit counts of the task, then the per-row counts packed four to a word (a task of at most 255 hits
                // cannot carry between the bytes) -- instead of one per row, and the masks for windows that stick out of the
                // item run for its first row and its last two quads only: per row this part cost as much as two of its four
                // probe steps.
                if (nrun > B.run_cap) return;   // the item is walked whole anyway
                const uint32_t qi0 = sc.c * TQ + lane, pos00 = sc.head + (qi0 << 4);   // row r: + 64 r quads, + 1024 r bytes
                uint32_t pk_lo = 0, pk_hi = 0;
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    uint32_t zr = qi0 + 64u * uint32_t(r) < sc.nq ? z[r] : 0u;
                    const uint32_t row_first = sc.c * TQ + 64u * uint32_t(r);   // (wave-uniform)
                    if (row_first == 0u || row_first + 65u >= sc.nq) {
                        const uint32_t pos0 = pos00 + 1024u * uint32_t(r);
                        if (pos0 < 3u) zr &= ~((1u << (3u - pos0)) - 1u);   // start = pos0 + e - 3 >= 0
                        const int32_t emax = int32_t(sc.n + 3u) - int32_t(m_min) - int32_t(pos0);   // start + m_min <= n
                        if (emax < 15) zr = emax < 0 ? 0u : (zr & ((2u << emax) - 1u));
                    }
                    z[r] = zr;
                    const uint32_t c = uint32_t(__builtin_popcount(zr));
                    if (r < 4) pk_lo |= c << (8 * r); else pk_hi |= c << (8 * (r - 4));
                }
                const uint32_t mine = ((pk_lo * 0x01010101u) >> 24) + ((pk_hi * 0x01010101u) >> 24);
                const uint32_t T = __builtin_amdgcn_readlane(wave_inclusive_sum(mine), 63);
                if (T == 0u) return;
                if (T <= 255u) {
                    if (nrun + T > B.run_cap) { nrun = B.run_cap + 1u; return; }   // more windows than the run list holds: the item is walked
                    if (stage.count + T > kDenseStage) dense_flush(B, wave_first, stage, lane);
                    const uint32_t in_lo = wave_inclusive_sum(pk_lo), in_hi = wave_inclusive_sum(pk_hi);
                    const uint32_t tot_lo = __builtin_amdgcn_readlane(in_lo, 63), tot_hi = __builtin_amdgcn_readlane(in_hi, 63);
                    uint2 *slots2 = reinterpret_cast<uint2 *>(stage.slots);
                    uint32_t rowbase = 0;   // hits of the rows before this one (wave-uniform)
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        const uint32_t sh = 8u * uint32_t(r & 3);
                        const uint32_t cnt = ((r < 4 ? pk_lo : pk_hi) >> sh) & 0xffu, incl = ((r < 4 ? in_lo : in_hi) >> sh) & 0xffu;
                        uint32_t at = stage.count + rowbase + incl - cnt, idx = nrun + rowbase + incl - cnt;
                        const uint32_t pos0 = pos00 + 1024u * uint32_t(r);
                        for (uint32_t m = z[r]; m != 0; m &= m - 1, ++at, ++idx)
                            slots2[at] = make_uint2(sc.k | (idx << 8), pos0 + uint32_t(__builtin_ctz(m)) - 3u);
                        rowbase += ((r < 4 ? tot_lo : tot_hi) >> sh) & 0xffu;
                    }
                    stage.count += T;
                    nrun += T;
                    return;
                }
                // (more than 255 hits in one task: row by row below, with the masks applied already)
            }
            if (by_row) {
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    if (__ballot(z[r] != 0) == 0) continue;
                    uint32_t k_item = sc.k, n_item = sc.n, head = sc.head, qi = sc.c * TQ + uint32_t(r) * 64u + lane, nq_item = sc.nq;
                    if constexpr (TINY) {
                        if (tiny_on) {  // row r is item tscan * G + r
                            k_item = tscan * uint32_t(G) + uint32_t(r);
                            if (k_item >= tcnt) continue;
                            n_item = __builtin_amdgcn_readlane(rec.z, k_item);
                            head = 0;
                            qi = lane;
                            nq_item = (n_item + 15u) >> 4;
                            nrun = 0;
                        }
                    }
                    // windows that cannot anchor anything: begin before the item's first byte, or too near its end
                    uint32_t zr = qi < nq_item ? z[r] : 0u;
                    const uint32_t pos0 = head + (qi << 4);            // scan coordinate of the quad's first byte
                    if (pos0 < 3u) zr &= ~((1u << (3u - pos0)) - 1u);   // start = pos0 + e - 3 >= 0
                    {
                        // start + m_min <= n_item  <=>  e <= n_item + 3 - m_min - pos0
                        const int32_t emax = int32_t(n_item + 3u) - int32_t(m_min) - int32_t(pos0);
                        if (emax < 15) zr = emax < 0 ? 0u : (zr & ((2u << emax) - 1u));
                    }
                    const uint32_t cnt = uint32_t(__builtin_popcount(zr));
                    const uint32_t incl = wave_inclusive_sum(cnt);
                    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                    if (total != 0) {
                        if (nrun <= B.run_cap && nrun + total <= B.run_cap && total <= kDenseStage) {
                            // (dense stage: 256 entries of 8 bytes {item in the wave | run index << 8, window start}, flushed to
                            // the items' run lists -- 4 bytes per hit, item-major: an item's hits are one contiguous piece)
                            uint2 *slots2 = reinterpret_cast<uint2 *>(stage.slots);
                            if (stage.count + total > kDenseStage) dense_flush(B, wave_first, stage, lane);
                            uint32_t at = stage.count + incl - cnt, idx = nrun + incl - cnt;
                            for (uint32_t m = zr; m != 0; m &= m - 1, ++at, ++idx)
                                slots2[at] = make_uint2(k_item | (idx << 8), pos0 + uint32_t(__builtin_ctz(m)) - 3u);
                            stage.count += total;
                            nrun += total;
                        } else {
                            nrun = B.run_cap + 1u;  // more windows than the run list holds: the whole item is walked
                        }
                    }
                    if constexpr (TINY) {
                        if (tiny_on) nrun_vec = lane == k_item ? nrun : nrun_vec;
                    }
                }
            }
            if constexpr (TINY) {
                if (tiny_on) return;
            }
            return;
        }
        if constexpr (TINY) {
            if (tiny_on) {  // row r is item tscan * G + r: its hits in lane order, its run count handed to its lane
                if (__any(zany != 0)) {
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        uint64_t who = __ballot(z[r] != 0);
                        if (who == 0) continue;
                        // the row's items in lane order (one, or tpack of them); an item's run count goes to its lane when its
                        // last hit has been flagged
                        uint32_t cur_k = 0xFFFFFFFFu;
                        while (who != 0) {
                            const uint32_t l = uint32_t(__builtin_ctzll(who));
                            who &= who - 1;
                            const uint32_t k = (tscan * uint32_t(G) + uint32_t(r)) * tpack + (l >> tshift);
                            if (k >= tcnt) continue;   // (a slot behind the batch's last item repeats that item: not its hits again)
                            if (k != cur_k) {
                                if (cur_k != 0xFFFFFFFFu) nrun_vec = lane == cur_k ? nrun : nrun_vec;
                                anchor_open<TQ>(A, B, rec, k, sc);
                                nrun = 0;
                                run_hi = 0;
                                cur_k = k;
                            }
                            flag(0u, l & ((1u << tshift) - 1u), __builtin_amdgcn_readlane(z[r], l));
                        }
                        if (cur_k != 0xFFFFFFFFu) nrun_vec = lane == cur_k ? nrun : nrun_vec;
                    }
                }
                return;
            }
        }
        const uint64_t cand = __ballot(zany != 0);
        if (__builtin_expect(cand != 0, 0)) {
            if ((cand & (cand - 1)) == 0) {
                // one lane hit (the usual case): its eight masks go through the scalar unit
                const uint32_t l = uint32_t(__builtin_ctzll(cand));
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    const uint32_t m = __builtin_amdgcn_readlane(z[r], l);
                    if (m) flag(uint32_t(r), l, m);
                }
            } else {
                uint32_t hits = 0;  // bit r: some window ENDING in this lane's quad r hit
#pragma unroll
                for (int r = 0; r < G; ++r) hits |= z[r] ? (1u << r) : 0u;
                uint32_t rmask = 0;
                for (uint64_t w = cand; w != 0; w &= w - 1) rmask |= __builtin_amdgcn_readlane(hits, uint32_t(__builtin_ctzll(w)));
                while (rmask != 0) {  // ascending quad order (r outer, lane inner) keeps the run list sorted
                    const uint32_t r = uint32_t(__builtin_ctz(rmask));
                    rmask &= rmask - 1;
                    uint32_t zr = 0;
#pragma unroll
                    for (int i = 0; i < G; ++i) zr = uint32_t(i) == r ? z[i] : zr;
                    uint64_t who = __ballot(zr != 0);
                    while (who != 0) {
                        const uint32_t l = uint32_t(__builtin_ctzll(who));
                        flag(r, l, __builtin_amdgcn_readlane(zr, l));
                        who &= who - 1;
                    }
                }
            }
        }
    };
    auto close_or_advance = [&]() -> bool {  // scan cursor: next chunk, or finish the item and open the next
        if (sc.c + 1u == sc.chunks) {
            // anchors the main loop did not test start after head + 16*nq - q: what they could match
            // ends at or after head + 16*nq - q + m_min
            uint32_t lo = sc.head + (sc.nq << 4) + (A.anchor_choice ? q : m_min);
            lo = lo > q ? lo - q : 0u;
            if (nrun > 0 && lo < run_hi) lo = run_hi;
            if (!DENSE && sc.head + (sc.nq << 4) < sc.n) {
                if (lo < sc.n) {
                    append_run_wave(B, wave_first, stage, sc.k, lo, sc.n, kEntryUnknown, nrun, run_hi, lane);
                } else if (nrun > 0 && nrun <= B.run_cap) {
                    // the previous run already reaches the item's end: it now also stands for the untested windows
                    // of the tail and must be walked (it is the last staged entry, see flag())
                    if (lane == 0) stage.slots[stage.count - 1].w = kEntryUnknown;
                }
            }
            nrun_vec = lane == sc.k ? nrun : nrun_vec;
        }
        const uint32_t k_before = sc.k;
        const bool more = anchor_next<TQ>(A, B, rec, sc);
        if (more && sc.k != k_before) {
            nrun = 0;
            run_hi = 0;
            if (sc.head) {
        // (dense form: its run lists hold window starts only -- an item with an unaligned head, the first haystack of an unaligned
        // blob, is walked whole)
        if constexpr (DENSE) nrun = B.run_cap + 1u;
        else append_run_wave(B, wave_first, stage, sc.k, 0, (sc.head + l_max) < sc.n ? sc.head + l_max : sc.n, 0u, nrun, run_hi, lane);
    }
        }
        return more;
    };

    // Two tasks are in flight: buf0 / buf1 hold tasks t and t+1, and each quad's registers are reloaded
    // with task t+2 as soon as the quad has been hashed (scan_task).  Loads are unconditional -- past
    // the last task the cursor is an invalid item whose loads fall on the anchor table -- because a load
    // under a branch makes hipcc drain vmcnt(0) instead of counting.
    // (BIG: tasks are 4 KiB, so the two buffers take the 32 registers one 8-KiB task would.  The version with a
    // single buffer loaded at the end of each scan left every task's HBM latency exposed at its first quad; now the
    // loads of task t+1 are a whole scan old when it begins, and the probe loads -- whose in-order vmcnt wait also
    // drains whatever was issued before them -- come two thirds into a task at a 10 % pass rate.)
    bool done_tiny = false;
    if constexpr (TINY) {
        if (tiny_on) {
            tload = 0;
            load_task(buf0, ld);
            tload = 1;
            load_task(buf1, ld);
            for (;;) {
                tload = tscan + 2u;
                scan_task(buf0);
                if (++tscan * uint32_t(G) * tpack >= tcnt) break;
                tload = tscan + 2u;
                scan_task(buf1);
                if (++tscan * uint32_t(G) * tpack >= tcnt) break;
            }
            done_tiny = true;
        }
    }
    if (!done_tiny) {
    load_task(buf0, ld);
    anchor_next<TQ>(A, B, rec, ld);
    load_task(buf1, ld);
    for (;;) {
        anchor_next<TQ>(A, B, rec, ld);
        scan_task(buf0);
        if (!close_or_advance()) break;
        anchor_next<TQ>(A, B, rec, ld);
        scan_task(buf1);
        if (!close_or_advance()) break;
    }
    }

    // hand the run lists over to k_walk_runs (lane k: item wave_first + k)
    if constexpr (DENSE) dense_flush(B, wave_first, stage, lane);
    else stage_flush(B, wave_first, stage, lane);
    if (rec.z != 0) B.run_count[wave_first + lane] = nrun_vec;
    if (!B.persistent) break;
    }  // next batch of this wave
    if constexpr (TINY) {
        // one device atomic per WORKGROUP (through the waves' own LDS regions, free by now): one per batch on a single word
        // was the longest thing in the launch (see wave_grab)
        uint32_t *mine = reinterpret_cast<uint32_t *>(smem + kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u);
        if (lane == 0) *mine = tiny_batches;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (uint32_t w = 0; w < kLdsThreads / 64u; ++w) sum += *reinterpret_cast<const uint32_t *>(smem + kAnchorLdsBytes + w * 2048u);
            if (sum) atomicAdd(&B.counters[18], sum);
        }
    }
}

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
constexpr uint32_t kWalkThreads = 1024;
constexpr uint32_t kWalkTileBytes = 64u * 1024u;
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
    if (uint64_t(blockIdx.x) * kWalkThreads >= total_items) return;
    __shared__ uint32_t s_walked, s_done, s_fallback;
    if (threadIdx.x == 0) { s_walked = 0; s_done = 0; s_fallback = 0; }
    const uint64_t item = uint64_t(blockIdx.x) * kWalkThreads + threadIdx.x;
    // issue the item's own loads first: they are in flight while the tile is staged
    uint32_t nr = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (item < total_items) {
        nr = B.run_count[item];
        rec = B.item_rec[item];
    }
    const uint32_t tile_rows = A.row_bytes ? (tile_bytes / A.row_bytes < A.trap ? tile_bytes / A.row_bytes : A.trap) : 0u;
    {
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
    bool whole = false;
    if (nr != 0) {
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        // (a run list that overflowed -- more candidate windows than it holds -- means the whole item: not by this lane alone,
        // see below; findAll = false keeps the old way, the first event needs the order)
        if (nr > B.run_cap && !B.first_only) whole = true;
        else walked = run_epilogue_fast<E, MODE, EXTRA>(A, B, smem, tile_rows, item, it, nr, fallback16);
    }
    wave_walk_whole<E, MODE>(A, B, smem, tile_rows, whole, rec, item, walked, fallback16);
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (atomicAdd(&s_done, 1u) == kWalkThreads / 64u - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t total = atomicAdd(&s_walked, 0u), fb = atomicAdd(&s_fallback, 0u);
            if (total) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], total);
            if (fb) atomicAdd(&B.counters[7], fb);
        }
    }
}

static hipError_t launch_verify_dense(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                                      hipStream_t st);

hipError_t launch_walk_runs(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                            hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    if (B.run_dense) return launch_verify_dense(A, B, max_items, entry_bytes, class_mode, st);
    const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);
    // (no tile at all when candidates are decided by text compares: 52 -> 49.5 us on the headline batch, 0.148 -> 0.141 ms on
    // config 3 -- not worth walking every fallback cluster through L2)
    const uint32_t tile_bytes = kWalkTileBytes;
    const uint32_t lds = tile_bytes;
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B, tile_bytes);
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
    const uint32_t hmul = IMG ? img[5] : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *starts_all = reinterpret_cast<uint32_t *>(B.runs);
    for (uint64_t item = uint64_t(blockIdx.x) * 16u + (threadIdx.x >> 6); item < total_items; item += uint64_t(gridDim.x) * 16u) {
        const uint32_t nr = __builtin_amdgcn_readfirstlane(B.run_count[item]);
        if (nr == 0 || nr > Q) continue;   // (more windows than the list holds: k_dense_emit walks the whole item)
        const uint4 rec = B.item_rec[item];
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        uint32_t *starts = starts_all + item * uint64_t(Q), *fidx = found_idx + item * uint64_t(Q);
        uint32_t base = 0;
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
            } else
            if (active) {
                if (start + 16u > n) {
                    out = start + 4u <= n ? kDenseMany : kDenseNone;   // (decided by k_dense_emit, which reads single bytes)
                } else {
                    const uint4 tx = load16u(p + start);
                    uint32_t slot = ((tx.x * hmul) >> 16) & mask, e1;
                    for (;;) {
                        e1 = ht[2u * slot + 1u];
                        if (e1 == 0xFFFFFFFFu || ht[2u * slot] == tx.x) break;
                        slot = (slot + 1u) & mask;
                    }
                    if (e1 != 0xFFFFFFFFu) {
                        const uint32_t first = e1 & 0xFFFFu, count = e1 >> 16;
#pragma unroll 1
                        for (uint32_t k = 0; k < count; ++k) {
                            const uint32_t r0 = recs[2u * (first + k)], r1 = recs[2u * (first + k) + 1u];
                            const uint32_t len = r0 >> 24, toff = r1 & 0xFFFFu;
                            if (start < it.floor || start + len > n) continue;
                            const uint32_t tl = len - 4u;
                            const uint32_t m0 = tl >= 4u ? 0xFFFFFFFFu : (1u << (8u * tl)) - 1u;
                            const uint32_t m1 = tl >= 8u ? 0xFFFFFFFFu : tl > 4u ? (1u << (8u * (tl - 4u))) - 1u : 0u;
                            const uint32_t m2 = tl >= 12u ? 0xFFFFFFFFu : tl > 8u ? (1u << (8u * (tl - 8u))) - 1u : 0u;
                            const uint32_t diff = ((tx.y ^ tails[toff]) & m0) | ((tx.z ^ tails[toff + 1u]) & m1) | ((tx.w ^ tails[toff + 2u]) & m2);
                            if (diff) continue;
                            out = out == kDenseNone ? first + k : kDenseMany;
                        }
                    }
                }
            }
            const bool keep = out != kDenseNone;
            const uint64_t km = __ballot(keep);
            if (km != 0) {
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi(uint32_t(km >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(km), 0u));
                if (keep) {
                    starts[at] = start;   // at <= r: every lane of this step has read its start, earlier steps' are consumed
                    fidx[at] = out;
                }
                base += uint32_t(__builtin_popcountll(km));
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
            uint32_t r = 0;
#pragma unroll 1
            while (r < nf) {
                // a cluster: occurrences whose possible ranges [start + m_min - 1, start + Lmax) touch or overlap
                const uint32_t first = starts[r];
                uint32_t hi = first + l_max < n ? first + l_max : n, nv = 0;
                Verified v[kClusterMax];
#pragma unroll
                for (uint32_t i = 0; i < kClusterMax; ++i) v[i] = Verified{0xFFFFFFFFu, 0u, 0u};
                bool windows = true;
                uint32_t st = first;
#pragma unroll 1
                for (;;) {
                    const uint32_t got = fidx[r];
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
                    st = starts[r];
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
        const uint32_t lds1 = image ? A.dense_image_words * 4u : 16u;
        hipError_t e = image ? hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_decide<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds1))
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B);
        return hipGetLastError();
    };
#define ACMI_VERIFY_DENSE(E, M) go(k_verify_dense<E, M, false>)
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_VERIFY_DENSE(uint16_t, 1) : ACMI_VERIFY_DENSE(uint16_t, 0);
    return class_mode == 1 ? ACMI_VERIFY_DENSE(uint32_t, 1) : ACMI_VERIFY_DENSE(uint32_t, 0);
#undef ACMI_VERIFY_DENSE
}

// ------------------------------------------------------------------------------------------
// k_walk_items: exact walk of WHOLE work items, one per thread (kernel 4).
//
// For text on which k_scan_lds' optimistic pass flags everything anyway -- keyword-like sets with events
// in every wave-quad, the dense sets of BASELINE configs[4] -- the table walk is all there is.  One item per
// thread, 2,048 walks per CU.  What bounded the first version was HBM, not the walk: a lane's bytes share a
// 128-byte line with no other lane's, 2^19 lanes hold 64 MiB of such lines open at once -- twice the L2 --
// and with one 16-byte block requested at a time every line was fetched from HBM again for each of its
// blocks (TCC_MISS x 128 B = 8.5 x the haystack bytes, 6.3 TB/s of HBM traffic for 0.74 TB/s of text).  So a
// lane requests LB blocks of one line back to back, one group ahead, and walks the group it holds: LB x 16
// unrolled steps, no per-byte select.  LB = 4 (64 bytes) is the most that fits the 64 VGPRs of 8 waves per
// SIMD and is the default: 1.16 ms per GiB of the keyword set of tools/bench_words.py against 1.42 ms with
// single blocks; LB = 8 (whole lines, 96 VGPRs, 4 waves per SIMD) 1.59 ms.  Byte classes and the first 64 KiB of
// the tile are in LDS; the cold-row load and the event block are the only branches.  Two items per thread
// (their blocks side by side, conditional or unconditional cold loads) measured equal or slower and was dropped.
// Same event rule and slab layout as run_epilogue.
// ------------------------------------------------------------------------------------------
// two workgroups per CU share the 160 KiB: 79 KiB each, 512 B of it the byte classes
constexpr uint32_t kItemsTileBytes = 78u * 1024u;

__device__ __forceinline__ uint64_t emit_event(const DevAutomaton &A, uint4 *out, uint64_t w, uint32_t hay, uint32_t pos, uint32_t state);

// EMIT: the same walk as the DRAIN of a batch this kernel scanned -- a thread per OVERFLOWED item (its events did not fit
// the slab), records written straight into their final slots.  Items of <= 1 KiB are what this kernel is given, and at tens of
// events per item a wave per item (k_refill_wave: a round of dependent loads per 256 positions) was five times slower than
// walking them again one per thread (a word list over text made of its words: 14 -> 3.6 ms per GiB).
template <typename E, int MODE, int LB, bool EMIT = false>
__global__ __launch_bounds__(kWalkThreads, (LB <= 4 ? 8 : 4)) void k_walk_items(DevAutomaton A, BatchArgs B, const uint64_t *rec_base,
                                                                                uint4 *out, uint64_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t total_items = EMIT ? uint64_t(B.counters[0]) : B.item_base[B.n_haystacks];   // (EMIT: entries of the overflow list)
    if (uint64_t(blockIdx.x) * kWalkThreads >= total_items) return;
    const uint64_t slot = uint64_t(blockIdx.x) * kWalkThreads + threadIdx.x;
    const uint64_t item = EMIT ? (slot < total_items ? uint64_t(B.overflow_list[slot]) : 0ull) : slot;
    __shared__ uint32_t s_walked;
    if (threadIdx.x == 0) s_walked = 0;
    uint4 rec = slot < total_items ? B.item_rec[item] : make_uint4(0u, 0u, 0u, 0u);
    uint64_t wr = 0;   // EMIT: where the item's next record goes
    uint32_t hay = 0;
    if constexpr (EMIT) {
        if (slot < total_items && rec_base[item + 1] <= cap) {
            wr = rec_base[item];
            hay = B.item_hay[item];
        } else {
            rec.z = 0u;   // (nothing to do: past the list, or the output buffer is too small -- the host grows it and launches again)
        }
    }
    // LDS: [512 B byte classes (TABLE mode)] [tile rows]
    const uint32_t cls_bytes = (MODE == 0) ? 512u : 0u;
    const uint32_t tile_rows = A.row_bytes ? (kItemsTileBytes / A.row_bytes < A.trap ? kItemsTileBytes / A.row_bytes : A.trap) : 0u;
    const uint16_t *cls_lds = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem) : nullptr;
    const uint8_t *tile = smem + cls_bytes;
    // a tile that leaves room takes its states' record counts along: where nearly every byte ends a pattern (a^1..a^16)
    // the count was a dependent global load per byte, and that load was most of the kernel (0.33 ms per 32 MiB)
    const uint32_t tile_bytes = (tile_rows * A.row_bytes + 15u) & ~15u;
    const uint32_t cnt_rows = tile_bytes + tile_rows * 4u <= kItemsTileBytes ? tile_rows : 0u;
    const uint32_t *cnt_lds = reinterpret_cast<const uint32_t *>(tile + tile_bytes);
    {
        if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem)[threadIdx.x] = A.class_off[threadIdx.x];
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + cls_bytes);
        const uint32_t n16 = tile_bytes >> 4;
        for (uint32_t t = threadIdx.x; t < n16; t += kWalkThreads) tdst[t] = tsrc[t];
        uint32_t *cdst = reinterpret_cast<uint32_t *>(smem + cls_bytes + tile_bytes);
        for (uint32_t t = threadIdx.x; t < cnt_rows; t += kWalkThreads) cdst[t] = A.out_count[t];
    }
    __syncthreads();

    ItemDesc it;
    unpack_item(rec.x, rec.y, rec.z, rec.w, it);
    const uint32_t n = rec.z;  // 0 for threads past the last item
    const uint8_t *p = B.blob + (n ? it.begin - it.warm : 0);
    const uint32_t warm = it.warm, pos_bias = it.pos0 - it.warm + 1u;
    uint32_t st = 0;
    EventRegs ev{0, 0};
    auto step = [&](uint32_t byte, uint32_t jj) __attribute__((always_inline)) {
        const uint32_t coff = class_offset<E, MODE>(A, cls_lds, byte);
        uint32_t w = A.trap;
        if (st < tile_rows) w = *reinterpret_cast<const E *>(tile + __umul24(st, A.row_bytes) + coff);
        if (w >= A.trap) w = cold_step<E>(A, st, byte, coff);
        st = w;
        if (jj >= warm && is_final(A, st) && !(B.first_only && ev.n_ev)) {
            if constexpr (EMIT) {
                wr = emit_event(A, out, wr, hay, pos_bias + jj, st);
            } else {
                ev.n_rec += st < cnt_rows ? cnt_lds[st] : A.out_count[st];
                if (ev.n_ev < B.slab_events) B.slab[uint64_t(ev.n_ev) * B.run_stride + item] = make_uint2(pos_bias + jj, st);
            }
            ++ev.n_ev;
        }
    };
    // (written out: as a loop over an array of the four words hipcc kept the array in scratch memory)
#define ACMI_STEP4(wd, jj) \
    step((wd) & 0xffu, (jj)); step(((wd) >> 8) & 0xffu, (jj) + 1u); step(((wd) >> 16) & 0xffu, (jj) + 2u); step((wd) >> 24, (jj) + 3u);
    auto step_block = [&](const uint4 &blk, uint32_t jj) __attribute__((always_inline)) {
        ACMI_STEP4(blk.x, jj) ACMI_STEP4(blk.y, jj + 4u) ACMI_STEP4(blk.z, jj + 8u) ACMI_STEP4(blk.w, jj + 12u)
    };
#undef ACMI_STEP4
    uint32_t j = 0;
    // head: single bytes up to a 16-byte boundary, single blocks up to a group boundary
    for (; j < n && (reinterpret_cast<uintptr_t>(p + j) & 15u) != 0; ++j) step(p[j], j);
    for (; j + 16u <= n && (reinterpret_cast<uintptr_t>(p + j) & (16u * LB - 1u)) != 0; j += 16u)
        step_block(*reinterpret_cast<const uint4 *>(p + j), j);
    // body: groups of LB blocks of one line, the next group requested (back to back, clamped index) while this
    // one is walked
    const uint32_t ng = (n - j) / (16u * LB);
    if (ng) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p + j);
        uint4 cur[LB], nxt[LB];
#pragma unroll
        for (int i = 0; i < LB; ++i) cur[i] = q[i];
#pragma unroll 1
        for (uint32_t g = 0; g < ng; ++g) {
            const uint4 *qn = q + (g + 1u < ng ? g + 1u : g) * LB;
#pragma unroll
            for (int i = 0; i < LB; ++i) nxt[i] = qn[i];
#pragma unroll
            for (int i = 0; i < LB; ++i) step_block(cur[i], j + 16u * uint32_t(i));
#pragma unroll
            for (int i = 0; i < LB; ++i) cur[i] = nxt[i];
            j += 16u * LB;
        }
    }
    // what is left: whole blocks, then bytes
    for (; j + 16u <= n; j += 16u) step_block(*reinterpret_cast<const uint4 *>(p + j), j);
    for (; j < n; ++j) step(p[j], j);

    if constexpr (EMIT) return;
    uint32_t walked = n;
    if (ev.n_ev) {
        B.ev_count[item] = ev.n_ev;
        B.rec_count[item] = ev.n_rec;
        if (ev.n_ev > B.slab_events) B.overflow_list[atomicAdd(&B.counters[0], 1u)] = uint32_t(item);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) walked += __shfl_xor(walked, d, 64);
    if ((threadIdx.x & 63u) == 0 && walked) atomicAdd(&s_walked, (walked + 15u) >> 4);
    __syncthreads();
    if (threadIdx.x == 0 && s_walked) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], s_walked);
}

hipError_t launch_walk_items(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                             int group, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    const uint32_t lds = kItemsTileBytes + (class_mode == 0 ? 512u : 0u);
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B, static_cast<const uint64_t *>(nullptr), static_cast<uint4 *>(nullptr),
                           uint64_t(0));
        return hipGetLastError();
    };
    // group = 16-byte blocks a lane requests back to back: 1 (the first version), 4 or 8 (a whole line)
#define ACMI_WALK_ITEMS(E, M) (group >= 8 ? go(k_walk_items<E, M, 8>) : group >= 4 ? go(k_walk_items<E, M, 4>) : go(k_walk_items<E, M, 1>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_WALK_ITEMS(uint16_t, 1) : ACMI_WALK_ITEMS(uint16_t, 0);
    return class_mode == 1 ? ACMI_WALK_ITEMS(uint32_t, 1) : ACMI_WALK_ITEMS(uint32_t, 0);
#undef ACMI_WALK_ITEMS
}

// ------------------------------------------------------------------------------------------
// findAll == false: keep the first event of the first item (per haystack) that has one
// ------------------------------------------------------------------------------------------
// Item-parallel, two launches: the first item with an event per haystack by atomicMin, then everything behind it dropped.  (A
// thread per HAYSTACK looping over its items took 5.8 ms on a batch with one 256-MiB haystack: 32,768 items behind one thread.)
__global__ void k_first_only_mark(BatchArgs B, uint32_t *first_item) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B.item_base[B.n_haystacks]) return;
    // (a long haystack's items all name the same word: an item behind what is recorded already says nothing, and a wave whose
    // lanes share the haystack sends its smallest candidate only -- 33,000 same-address atomics were 0.4 ms a haystack)
    const uint32_t hay = B.item_hay[i];
    const bool cand = B.ev_count[i] != 0 && first_item[hay] > uint32_t(i);
    const uint64_t who = __ballot(cand);
    if (who == 0) return;
    const uint32_t lead = uint32_t(__builtin_ctzll(who));   // the wave's first candidate: items ascend with the lane
    const uint32_t hay0 = uint32_t(__builtin_amdgcn_readlane(int(hay), int(lead)));
    if (cand && (hay != hay0 || (threadIdx.x & 63u) == lead)) atomicMin(&first_item[hay], uint32_t(i));
}
__global__ void k_first_only_drop(BatchArgs B, const uint32_t *first_item) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B.item_base[B.n_haystacks]) return;
    if (B.ev_count[i] && first_item[B.item_hay[i]] != uint32_t(i)) { B.ev_count[i] = 0; B.rec_count[i] = 0; }
}

// Expand one event into records, longest pattern first (node.c:424-441 order).
__device__ __forceinline__ uint64_t emit_event(const DevAutomaton &A, uint4 *out, uint64_t w, uint32_t hay,
                                               uint32_t pos, uint32_t state) {
    uint32_t t = A.own_pattern[state] != 0xFFFFFFFFu ? state : A.dict_link[state];
    while (t != 0xFFFFFFFFu) {
        const uint32_t pat = A.own_pattern[t];
        out[w++] = make_uint4(hay, pos, pos - A.pattern_length[pat], pat);
        t = A.dict_link[t];
    }
    return w;
}

// `cap` = records the output buffer can hold: the host launches this optimistically with the
// capacity of the previous call and re-runs it after growing the buffer if the batch did not fit.
//
// SCAN = true (the pipeline's launch): the exclusive scan of the per-item record counts is done HERE, single pass with
// look-back like k_items -- a tile of 1,024 items per workgroup, one per thread -- so that the record base of an item is known
// in the kernel that expands it (k_rec_scan used to be a launch of its own: 14 us + a kernel boundary per batch).  rec_base[]
// is still written (the drain of overflowed items and a re-expansion read it).  The per-haystack prefix of the record array
// comes from the items too: the thread of a haystack's FIRST item writes that haystack's word -- and those of the empty
// haystacks right in front of it --, the workgroup that holds the closing element the words behind the last item, the batch
// summary {total records, work items} for the host's single read-back next to the overflow counter; and the first workgroup
// zeroes the sync block of the NEXT batch (counters, tickets, scan status words).
// SCAN = false (re-expansion into a grown buffer): rec_base[] is read.
// (one item per thread, tiles of 1,024: with two items per thread and 2,048-item tiles the expansion of a batch with a dozen
// events per item -- chains of dependent loads per event -- took 74 us on half the compute units)
constexpr int kExpandThreads = 1024;
constexpr int kExpandPerThread = 1;
constexpr int kExpandTile = kExpandThreads * kExpandPerThread;

template <bool SCAN>
__global__ __launch_bounds__(kExpandThreads) void k_expand(DevAutomaton A, BatchArgs B, uint64_t *rec_base, uint4 *out, uint64_t cap,
                                                            uint64_t *prefix, uint32_t *host_counters, uint32_t *next_sync,
                                                            uint32_t next_sync_words, uint32_t *ticket, uint64_t *status,
                                                            const uint32_t *bad_offsets, uint32_t *next_isync, uint32_t next_isync_words) {
    __shared__ uint64_t wave_sums[kExpandThreads / 64];
    __shared__ uint64_t s_before, s_total;
    __shared__ uint32_t s_tile, s_closing;
    if (blockIdx.x == 0) {
        for (uint32_t k = threadIdx.x; k < next_sync_words; k += blockDim.x) next_sync[k] = 0u;
        for (uint32_t k = threadIdx.x; k < next_isync_words; k += blockDim.x) next_isync[k] = 0u;   // (the next batch's k_items: ticket, flag, status)
        // the batch's counters straight into the host's pinned block (words 2..5, the summary, come from the closing element
        // below): no copy command behind the pipeline
        // (words 16..31: the scans' tickets and, word 18, the wave batches k_scan_anchor's TINY path took)
        if (threadIdx.x < 32u && (threadIdx.x < 2u || threadIdx.x > 5u))
            host_counters[threadIdx.x] = threadIdx.x == 6u ? *bad_offsets : B.counters[threadIdx.x];   // ([6]: k_items' verdict on the offsets)
    }
    if (threadIdx.x == 0) { s_tile = SCAN ? atomicAdd(ticket, 1u) : blockIdx.x; s_closing = 0u; }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t items = B.item_base[B.n_haystacks];
    if (uint64_t(tile) * kExpandTile > items) return;  // (so does every later tile: nobody looks back at this one)
    const uint64_t base = uint64_t(tile) * kExpandTile + uint64_t(threadIdx.x) * kExpandPerThread;
    uint64_t v[kExpandPerThread], rb[kExpandPerThread + 1];
#pragma unroll
    for (int i = 0; i < kExpandPerThread; ++i) v[i] = base + i < items ? B.rec_count[base + i] : 0ull;
    if constexpr (SCAN) {
        uint64_t sum = 0;
#pragma unroll
        for (int i = 0; i < kExpandPerThread; ++i) sum += v[i];
        uint64_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint64_t wave_base = 0, tile_sum = 0;
#pragma unroll
        for (int w = 0; w < kExpandThreads / 64; ++w) {
            const uint64_t x = wave_sums[w];
            if (w < wave) wave_base += x;
            tile_sum += x;
        }
        if (threadIdx.x < 64) {
            const uint64_t bf = chain_exclusive(status, tile, tile_sum);
            if (threadIdx.x == 0) s_before = bf;
        }
        __syncthreads();
        uint64_t run = s_before + wave_base + inc - sum;
#pragma unroll
        for (int i = 0; i <= kExpandPerThread; ++i) {
            rb[i] = run;
            if (i < kExpandPerThread) {
                if (base + i <= items) rec_base[base + i] = run;
                run += v[i];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i <= kExpandPerThread; ++i) rb[i] = base + i <= items ? rec_base[base + i] : 0ull;
    }
    // prefix words and the events of this thread's items
#pragma unroll
    for (int i = 0; i < kExpandPerThread; ++i) {
        const uint64_t g = base + i;
        if (g > items) break;
        if (g == items) {  // the closing element: the total; the haystacks behind the last item are this workgroup's (below)
            uint64_t *summary = reinterpret_cast<uint64_t *>(host_counters + 2);
            summary[0] = rb[i];
            summary[1] = items;
            s_total = rb[i];
            s_closing = 1u;
            break;
        }
        const uint32_t hay = B.item_hay[g];
        const uint32_t prev = g ? B.item_hay[g - 1] : 0xFFFFFFFFu;
        if (prev != hay) {  // the haystack's first item: its prefix word, and those of the empty haystacks right before it
            for (uint32_t hh = prev + 1u; hh != hay + 1u; ++hh) prefix[hh] = rb[i];
        }
        const uint32_t n = B.ev_count[g];
        if (n == 0 || n > B.slab_events) continue;  // overflowed items are k_refill's
        if (rb[i + 1] > cap) continue;
        uint64_t w = rb[i];
        for (uint32_t e = 0; e < n; ++e) {
            const uint2 ev = B.slab[uint64_t(e) * B.run_stride + g];  // event-major, like the run lists
            w = emit_event(A, out, w, hay, ev.x, ev.y);
        }
    }
    __syncthreads();
    if (s_closing) {  // prefix words of the haystacks behind the last item (all of them when the batch has no item), the closing word
        const uint64_t total = s_total;
        const uint32_t from = items ? B.item_hay[items - 1] + 1u : 0u;
        for (uint64_t hh = uint64_t(from) + threadIdx.x; hh <= B.n_haystacks; hh += blockDim.x) prefix[hh] = total;
    }
}

// Overflowed items: scan again, writing records straight into their final slots.  Grid-stride over
// the overflow list, whose length the host does not need to know at launch time.  One THREAD per item:
// kept for automata whose longest pattern makes the per-lane warm-up of k_refill_wave too expensive.
template <typename E>
__global__ void k_refill(DevAutomaton A, BatchArgs B, const uint64_t *rec_base, uint4 *out, uint64_t cap) {
    const uint32_t n_overflow = B.counters[0];
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_overflow; t += gridDim.x * blockDim.x) {
        const uint64_t i = B.overflow_list[t];
        ItemDesc it;
        if (!decode_item(B, i, it) || rec_base[i + 1] > cap) continue;
        uint64_t w = rec_base[i];
        const uint8_t *p = B.blob + (it.begin - it.warm);
        const uint32_t n = it.warm + it.len;
        uint32_t s = 0;
#pragma unroll 1
        for (uint32_t j = 0; j < n; ++j) {
            s = exact_step<E>(A, nullptr, s, p[j]);
            if (j >= it.warm && is_final(A, s)) w = emit_event(A, out, w, it.hay, it.pos0 + (j - it.warm) + 1, s);
        }
    }
}

// The dense-match drain (BASELINE configs[4]: >= 1 record per byte, 256 B written per byte read for
// a^1..a^16): one WAVE per overflowed item, and the wave writes the item's records as whole 1 KiB rows.
// An item is taken in tiles of 64 x S positions.  Per tile
//   1. lane l walks positions [l*S, l*S+S) exactly (entered from the root Lmax-1 bytes early, like every
//      chunk in this file) and keeps the state and the record count (out_count) of each;
//   2. a wave prefix sum over the counts gives every position its slot in the tile's output
//      (`cum`, in LDS next to the states);
//   3. 64 positions at a time, lane l expands position 64 g + l into an LDS staging window at its slot
//      (the state's flat output list, longest pattern first: independent 8-byte loads; the dictionary
//      links when the automaton has no flat lists), and the wave copies the window to the output with
//      64 consecutive 16-byte records per store instruction.
//      A group whose records exceed the window (more than 16 per position on average) takes several
//      windows; its lanes walk their chains again for each.
// One thread per item (k_refill) wrote the same bytes as 64 separate 16-byte streams per wave
// instruction: 1.2-1.5 TB/s of records.
constexpr uint32_t kRefillThreads = 256;
constexpr uint32_t kRefillS = 4;
constexpr uint32_t kRefillTile = 64u * kRefillS;
// Staged records per wave: NB slots, padded by one slot per 16 (slot i lives at i + i / 16).  Lanes write
// at a stride of their record counts -- 16 records = 256 B = the whole bank row for a^1..a^16 -- and
// without the padding all 64 lanes of a ds_write_b128 met in the same four banks.
__host__ __device__ constexpr uint32_t refill_wave_lds(uint32_t nb) {
    return (nb + nb / 16u) * 16u + ((kRefillTile + 1u + kRefillTile) * 4u + 15u) / 16u * 16u;
}

template <typename E, uint32_t NB>
__global__ __launch_bounds__(kRefillThreads) void k_refill_wave(DevAutomaton A, BatchArgs B, const uint64_t *rec_base,
                                                                uint4 *out, uint64_t cap, uint32_t tile_budget) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = reinterpret_cast<uint4 *>(smem + wave * refill_wave_lds(NB));
    uint32_t *cum = reinterpret_cast<uint32_t *>(stage + (NB + NB / 16u));  // kRefillTile + 1 slots
    uint32_t *stt = cum + kRefillTile + 1;
    // Behind the waves' regions: the byte classes and as many of the first (shallowest) table rows as the launch left room
    // for.  A lane walks Lmax-1 + 4 bytes per tile of 256 positions, every step a dependent lookup: from the global table
    // that was ~ 1 us a step, and an item with tens -- not thousands -- of records per tile spent its time there (short
    // patterns over their own alphabet: 70 M records per GiB at 0.19 TB/s of records).
    const uint16_t *cls = reinterpret_cast<const uint16_t *>(smem + (kRefillThreads / 64u) * refill_wave_lds(NB));
    const uint8_t *tile = reinterpret_cast<const uint8_t *>(cls + 256);
    const uint32_t tile_rows = A.row_bytes ? (tile_budget / A.row_bytes < A.trap ? tile_budget / A.row_bytes : A.trap) : 0u;
    {
        uint16_t *cdst = const_cast<uint16_t *>(cls);
        for (uint32_t t = threadIdx.x; t < 256u; t += kRefillThreads) cdst[t] = A.class_off[t];
        const uint32_t *tsrc = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint32_t *tdst = reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(tile));
        const uint32_t n4 = (tile_rows * A.row_bytes + 3u) >> 2;
        for (uint32_t t = threadIdx.x; t < n4; t += kRefillThreads) tdst[t] = tsrc[t];
    }
    __syncthreads();
    auto step = [&](uint32_t s, uint32_t byte) __attribute__((always_inline)) -> uint32_t {
        const uint32_t coff = cls[byte];
        if (s < tile_rows) {
            const uint32_t w = *reinterpret_cast<const E *>(tile + s * A.row_bytes + coff);
            if (w < A.trap) return w;   // (the image's rows name a cold target by the trap row)
        }
        return cold_step<E>(A, s, byte, coff);
    };
    const uint32_t n_overflow = B.counters[0];
    const uint32_t waves_total = gridDim.x * (kRefillThreads / 64);
    for (uint32_t t = blockIdx.x * (kRefillThreads / 64) + wave; t < n_overflow; t += waves_total) {
        const uint64_t i = __builtin_amdgcn_readfirstlane(B.overflow_list[t]);
        ItemDesc it;
        if (!decode_item_uniform(B, i, it)) continue;
        if (rec_base[i + 1] > cap) continue;
        uint64_t w = rec_base[i];
        const uint8_t *p = B.blob + (it.begin - it.warm);
#pragma unroll 1
        for (uint32_t q_tile = 0; q_tile < it.len; q_tile += kRefillTile) {
            // 1. exact walk of this lane's S positions.  The tile's text -- its 256 positions and the warm-up in front of them,
            // at most 319 bytes -- comes into LDS with one round of coalesced byte loads (the staging window is idle until step
            // 3): read from memory by every lane for itself it was a round trip per step.
            const uint32_t q0 = q_tile + lane * kRefillS;
            uint8_t *txt = reinterpret_cast<uint8_t *>(stage);
            const uint32_t r_end = it.warm + (q_tile + kRefillTile < it.len ? q_tile + kRefillTile : it.len);
            const uint32_t r0 = it.warm + q_tile > B.warm_bytes ? it.warm + q_tile - B.warm_bytes : 0u;   // (wave-uniform)
            const bool staged = B.warm_bytes + kRefillTile <= 320u;   // (longer warm-ups -- the kernel forced on long patterns -- read memory)
            if (staged) {
                uint32_t tb[5];
#pragma unroll
                for (uint32_t u = 0; u < 5u; ++u) tb[u] = r0 + lane + 64u * u < r_end ? load8g(p + r0 + lane + 64u * u) : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 5u; ++u) txt[lane + 64u * u] = uint8_t(tb[u]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            uint32_t st[kRefillS], cnt[kRefillS], sum = 0;
#pragma unroll
            for (uint32_t k = 0; k < kRefillS; ++k) { st[k] = 0; cnt[k] = 0; }
            if (q0 < it.len) {
                const uint32_t j_rep = it.warm + q0;  // scan coordinate of the first reported byte
                uint32_t j = j_rep > B.warm_bytes ? j_rep - B.warm_bytes : 0u;
                uint32_t s = 0;
#pragma unroll 1
                for (; j < j_rep; ++j) s = step(s, staged ? uint32_t(txt[j - r0]) : uint32_t(p[j]));
#pragma unroll
                for (uint32_t k = 0; k < kRefillS; ++k) {
                    if (q0 + k < it.len) {
                        s = step(s, staged ? uint32_t(txt[j_rep + k - r0]) : uint32_t(p[j_rep + k]));
                        st[k] = s;
                        cnt[k] = is_final(A, s) ? A.out_count[s] : 0u;
                        sum += cnt[k];
                    }
                }
            }
            // 2. slots: exclusive prefix over the wave
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if (lane >= uint32_t(d)) incl += v;
            }
            const uint32_t T = __shfl(incl, 63, 64);
            uint32_t run = incl - sum;
#pragma unroll
            for (uint32_t k = 0; k < kRefillS; ++k) {
                cum[lane * kRefillS + k] = run;
                stt[lane * kRefillS + k] = st[k];
                run += cnt[k];
            }
            if (lane == 63u) cum[kRefillTile] = T;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 3. expand.  A tile whose records fit ONE window (tens of records, not thousands: short patterns over their own
            // alphabet) is expanded by the lanes that walked it, each its own kRefillS positions from the states it still holds:
            // the loads of all four positions are in flight together and the tile is written with one copy -- 64 positions at
            // a time it was four rounds of two dependent round trips each, and those were the kernel (70 M records per GiB
            // at 0.19 TB/s of records).
            if (T != 0 && T <= NB && A.out_list != nullptr) {
                uint32_t off[kRefillS];
#pragma unroll
                for (uint32_t k = 0; k < kRefillS; ++k) off[k] = cnt[k] ? A.out_off[st[k]] : 0u;
                uint32_t slot = incl - sum;
#pragma unroll
                for (uint32_t k = 0; k < kRefillS; ++k) {
                    const uint32_t pos = it.pos0 + q0 + k + 1u;
                    const uint2 *ol = A.out_list + off[k];
                    for (uint32_t i = 0; i < cnt[k]; ++i, ++slot) {
                        const uint2 e = ol[i];
                        stage[slot + (slot >> 4)] = make_uint4(it.hay, pos, pos - e.y, e.x);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint4 *dst = out + w;
                for (uint32_t r = lane; r < T; r += 64u) dst[r] = stage[r + (r >> 4)];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (T != 0) {
#pragma unroll 1
                for (uint32_t g = 0; g < kRefillS; ++g) {
                    const uint32_t pp = g * 64u + lane;
                    const uint32_t start = cum[pp], c = cum[pp + 1] - start;
                    const uint32_t gbase = __builtin_amdgcn_readfirstlane(start);
                    const uint32_t gend = __builtin_amdgcn_readfirstlane(__shfl(start + c, 63, 64));
                    const uint32_t pos = it.pos0 + q_tile + pp + 1u;
                    const uint32_t s = stt[pp];
#pragma unroll 1
                    for (uint32_t wb = gbase; wb < gend; wb += NB) {
                        // records [k_lo, k_hi) of this lane's event fall into the window
                        const uint32_t k_lo = start < wb ? wb - start : 0u;
                        const uint32_t k_hi = start + c > wb + NB ? (wb + NB > start ? wb + NB - start : 0u) : c;
                        if (k_lo < k_hi) {
                            const uint32_t base = start - wb;  // slot of record 0 (mod 2^32 when k_lo > 0)
                            if (A.out_list != nullptr) {
                                const uint2 *ol = A.out_list + A.out_off[s];
#pragma unroll 4
                                for (uint32_t k = k_lo; k < k_hi; ++k) {
                                    const uint2 e = ol[k];
                                    const uint32_t slot = base + k;
                                    stage[slot + (slot >> 4)] = make_uint4(it.hay, pos, pos - e.y, e.x);
                                }
                            } else {
                                uint32_t tt = A.own_pattern[s] != 0xFFFFFFFFu ? s : A.dict_link[s];
#pragma unroll 1
                                for (uint32_t k = 0; k < k_hi; ++k) {
                                    if (k >= k_lo) {
                                        const uint32_t pat = A.own_pattern[tt];
                                        const uint32_t slot = base + k;
                                        stage[slot + (slot >> 4)] = make_uint4(it.hay, pos, pos - A.pattern_length[pat], pat);
                                    }
                                    tt = A.dict_link[tt];
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const uint32_t n = gend - wb < NB ? gend - wb : NB;
                        uint4 *dst = out + (w + wb);
                        // records are written once and not read by this pipeline again: with 16 KiB windows a
                        // non-temporal store is 9 % faster (a^1..a^16: 1.94 -> 1.77 ms per 8.6 GB), with 4 KiB
                        // windows 3 % slower
                        for (uint32_t r = lane; r < n; r += 64u) {
                            const uint4 v = stage[r + (r >> 4)];
                            if constexpr (NB >= 1024u)
                                __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(dst + r));
                            else
                                dst[r] = v;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            w += T;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

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
constexpr uint32_t kSmallMaxBytes = 64u * 1024u;
constexpr uint32_t kSmallMaxHay = 1024;
constexpr uint32_t kSmallChunk = 64;
constexpr uint32_t kSmallMaxItems = kSmallMaxBytes / kSmallChunk + kSmallMaxHay;  // 2048
constexpr uint32_t kSmallEvents = 2;            // events kept per item; more -> the item is walked again
constexpr uint32_t kSmallTileBytes = 36u * 1024u;
constexpr uint32_t kSmallThreads = 1024;

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
                                                               uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *in_bytes = smem;                                                       // 64 KiB
    const uint8_t *tile = smem + kSmallMaxBytes;                                    // 36 KiB
    uint2 *ev = reinterpret_cast<uint2 *>(smem + kSmallMaxBytes + kSmallTileBytes); // 2048 x 2 x 8 B = 32 KiB
    uint32_t *cnt_ev = reinterpret_cast<uint32_t *>(ev + kSmallMaxItems * kSmallEvents);  // 8 KiB
    uint32_t *cnt_rec = cnt_ev + kSmallMaxItems;                                     // 8 KiB
    uint32_t *item_base = cnt_rec + kSmallMaxItems;                                  // 1025 x 4 (+ pad) = 4.5 KiB
    uint32_t *hay_off = item_base + kSmallMaxHay + 8;                                // 1025 x 4
    uint32_t *wave_tmp = hay_off + kSmallMaxHay + 8;                                 // 16 x 4
    uint16_t *cls = reinterpret_cast<uint16_t *>(wave_tmp + 16);                     // 512 B (TABLE mode)

    const uint32_t t = threadIdx.x;
    const uint64_t off0 = offsets[0];
    // offsets -> LDS (relative to the first one), items per haystack
    for (uint32_t h = t; h <= nh; h += kSmallThreads) hay_off[h] = uint32_t(offsets[h] - off0);
    __syncthreads();
    const uint32_t n_bytes = hay_off[nh];
    for (uint32_t h = t; h < kSmallMaxHay + 1; h += kSmallThreads)
        item_base[h] = h < nh ? (hay_off[h + 1] - hay_off[h] + kSmallChunk - 1) / kSmallChunk : 0u;
    {   // haystack bytes: 16 bytes per lane, coalesced, from pinned host memory (16-byte aligned by the host)
        const uint4 *src = reinterpret_cast<const uint4 *>(blob);
        uint4 *dst = reinterpret_cast<uint4 *>(in_bytes);
        for (uint32_t i = t; i < (n_bytes + 15u) / 16u; i += kSmallThreads) dst[i] = src[i];
        const uint32_t tile_rows_b = A.row_bytes ? (kSmallTileBytes / A.row_bytes < A.trap ? kSmallTileBytes / A.row_bytes : A.trap) * A.row_bytes : 0u;
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + kSmallMaxBytes);
        for (uint32_t i = t; i < (tile_rows_b + 15u) / 16u; i += kSmallThreads) tdst[i] = tsrc[i];
        if (MODE == 0 && t < 256) cls[t] = A.class_off[t];
    }
    __syncthreads();
    const uint32_t tile_rows = A.row_bytes ? (kSmallTileBytes / A.row_bytes < A.trap ? kSmallTileBytes / A.row_bytes : A.trap) : 0u;
    const uint32_t n_items = small_scan(item_base, nh + 1, wave_tmp);  // item_base[h] = first item of haystack h; [nh] = total

    auto step = [&](uint32_t s, uint32_t byte) -> uint32_t {
        const uint32_t coff = class_offset<E, MODE>(A, cls, byte);
        uint32_t w = A.trap;
        if (s < tile_rows) w = *reinterpret_cast<const E *>(tile + s * A.row_bytes + coff);
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
        pos0 = (i - item_base[h]) * kSmallChunk;
        begin = hay_off[h] + pos0;
        const uint32_t rest = hay_off[h + 1] - begin;
        len = rest < kSmallChunk ? rest : kSmallChunk;
        const uint32_t wmax = max_pattern_length ? max_pattern_length - 1u : 0u;
        warm = pos0 < wmax ? pos0 : wmax;
    };

    // ---- walk: count events and records per item, keep the first kSmallEvents events ----
    for (uint32_t i = t; i < kSmallMaxItems; i += kSmallThreads) {
        uint32_t n_ev = 0, n_rec = 0;
        if (i < n_items) {
            uint32_t h, begin, len, warm, pos0;
            decode(i, h, begin, len, warm, pos0);
            uint32_t s = 0;
            const uint8_t *p = in_bytes + begin - warm;
            for (uint32_t j = 0; j < warm + len; ++j) {
                s = step(s, p[j]);
                if (j >= warm && is_final(A, s)) {
                    n_rec += A.out_count[s];
                    if (n_ev < kSmallEvents) ev[i * kSmallEvents + n_ev] = make_uint2(pos0 + (j - warm) + 1u, s);
                    ++n_ev;
                }
            }
        }
        cnt_ev[i] = n_ev;
        cnt_rec[i] = n_rec;
    }
    __syncthreads();
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
    const uint32_t total = small_scan(cnt_rec, kSmallMaxItems, wave_tmp);  // cnt_rec[i] = first record of item i
    for (uint32_t h = t; h <= nh; h += kSmallThreads)
        out_prefix[h] = (h < nh && item_base[h] < kSmallMaxItems) ? cnt_rec[item_base[h]] : total;
    if (t == 0) { out_summary[0] = total; out_summary[1] = n_items; out_summary[2] = total > cap ? 1u : 0u; }
    if (total > cap) return;
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
}

hipError_t launch_match_small(const DevAutomaton &A, const uint64_t *offsets, const uint8_t *blob, uint32_t nh, int find_all,
                              uint32_t max_pattern_length, int entry_bytes, int class_mode, uint64_t *out_summary,
                              uint64_t *out_prefix, void *out_records, uint32_t cap, hipStream_t st) {
    const uint32_t lds = kSmallMaxBytes + kSmallTileBytes + kSmallMaxItems * (kSmallEvents * 8u + 8u) + (kSmallMaxHay + 8u) * 8u + 64u + 512u;
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(1), dim3(kSmallThreads), lds, st, A, offsets, blob, nh, uint32_t(find_all ? 1 : 0),
                           max_pattern_length, out_summary, out_prefix, static_cast<uint4 *>(out_records), cap);
        return hipGetLastError();
    };
    if (entry_bytes == 2) return class_mode == 1 ? go(k_match_small<uint16_t, 1>) : go(k_match_small<uint16_t, 0>);
    return class_mode == 1 ? go(k_match_small<uint32_t, 1>) : go(k_match_small<uint32_t, 0>);
}

// ------------------------------------------------------------------------------------------
// k_stream_read: plain coalesced 16 B/lane streaming read (xor-reduced so it cannot be elided).
// Not on the product path: it measures what HBM read bandwidth a kernel can reach on this chip
// and calibrates the FETCH_SIZE counter against a known byte count (MI355X_MICROARCH.md, HBM).
// ------------------------------------------------------------------------------------------
__global__ void k_stream_read(const uint4 *src, uint64_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = stream_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // practically never: keeps the loads alive
}

void launch_stream_read(const void *src, uint64_t bytes, uint32_t *sink, hipStream_t st) {
    hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, st, static_cast<const uint4 *>(src), bytes / 16, sink);
}

// ------------------------------------------------------------------------------------------
// k_build_level: one trie depth of the automaton build on the device (automaton.hpp TrieView / BuildAccel).
//
// The reference computes a node's failure link by re-walking every suffix of its path from the root
// (ac_trie_set_failure, ahocorasick.c:344-368: 8.3 s for 65,536 patterns); the textbook recurrence -- follow the
// parent's failure chain until a node has a child on the node's byte -- needs only SHALLOWER nodes' links, so all
// nodes of one depth are independent: one thread per node, one launch per depth, the child lookups in the builder's
// own open-addressing map.  The per-state outputs (final flag, record count, dictionary link: node_collect_matches,
// node.c:424-441, folded into three words per node) depend on the node's failure target only and ride along.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t build_find_child(const BuildArgs &T, uint32_t node, uint32_t byte) {
    const uint64_t key = ((uint64_t(node) << 8) | byte) + 1;
    uint64_t k = key * 0x9E3779B97F4A7C15ull;
    k ^= k >> 29;
    uint64_t slot = k & (T.map_size - 1);
    for (;;) {
        const uint64_t have = T.map_keys[slot];
        if (have == key) return T.map_vals[slot];
        if (have == 0) return 0xFFFFFFFFu;
        slot = (slot + 1) & (T.map_size - 1);
    }
}

__global__ void k_build_level(BuildArgs T, uint32_t lo, uint32_t hi) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const uint32_t u = T.order[i];
    const uint32_t p = T.parent[u];
    uint32_t fl = 0;
    if (p != 0) {
        const uint32_t b = T.in_byte[u];
        uint32_t f = T.fail[p];
        for (;;) {
            const uint32_t c = build_find_child(T, f, b);
            if (c != 0xFFFFFFFFu) { fl = c; break; }
            if (f == 0) break;
            f = T.fail[f];
        }
    }
    T.fail[u] = fl;
    const bool owns = T.own[u] != 0xFFFFFFFFu;
    T.is_final[u] = (owns || T.is_final[fl]) ? 1 : 0;
    T.cnt[u] = (owns ? 1u : 0u) + T.cnt[fl];
    T.dict[u] = T.own[fl] != 0xFFFFFFFFu ? fl : T.dict[fl];
}

void launch_build_level(const BuildArgs &T, uint32_t lo, uint32_t hi, hipStream_t st) {
    if (hi > lo) hipLaunchKernelGGL(k_build_level, dim3((hi - lo + 255) / 256), dim3(256), 0, st, T, lo, hi);
}

// The host continues in BFS positions (automaton.cpp finalize: its passes then stream): the results are handed back indexed
// by position, links as positions.
__global__ void k_build_positions(const uint32_t *order, uint32_t n, uint32_t *pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[order[i]] = i;
}

__global__ void k_build_by_position(BuildArgs T, uint32_t n, const uint32_t *pos, uint32_t *fail_b, uint32_t *cnt_b, uint32_t *dict_b,
                                    uint8_t *final_b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = T.order[i];
    fail_b[i] = pos[T.fail[u]];
    cnt_b[i] = T.cnt[u];
    const uint32_t d = T.dict[u];
    dict_b[i] = d == 0xFFFFFFFFu ? d : pos[d];
    final_b[i] = T.is_final[u];
}

void launch_build_by_position(const BuildArgs &T, uint32_t n, uint32_t *pos, uint32_t *fail_b, uint32_t *cnt_b, uint32_t *dict_b,
                              uint8_t *final_b, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_build_positions, dim3((n + 255) / 256), dim3(256), 0, st, T.order, n, pos);
    hipLaunchKernelGGL(k_build_by_position, dim3((n + 255) / 256), dim3(256), 0, st, T, n, pos, fail_b, cnt_b, dict_b, final_b);
}

// ------------------------------------------------------------------------------------------
// Search and replace: multifast_replace + multifast_rep_flush(keep=0) (reference src/multifast/replace.c:469-568) as a
// post-pass over the match records of a batch.
//
// The reference books a NOMINEE per event -- the longest pattern of the node's list that has a replacement text
// (node.c:337-362) -- and keeps or drops nominees by mode (replace.c:222-272): NORMAL lets a nominee swallow the
// earlier ones that start at or after its own start (its factors), overlapping ones all stay; LAZY drops a nominee
// that starts before the previous one ends.  The output is then one left-to-right pass (replace.c:402-455): the text
// between the cursor and a nominee's start (nothing if the nominee starts before the cursor), its replacement, the
// cursor moved to the nominee's end; the rest of the text at the end.  The match records already hold every event
// in that order with the patterns of an event longest first, so:
//   k_replace_plan  one thread per haystack: nominees from its records, the booking rule, where in the haystack's output
//                   every kept nominee's piece (gap + replacement) begins, the output length;
//   k_scan_u64      exclusive scan of the output lengths -> output offsets, on the device (single pass, look-back);
//   k_replace_emit  the copy, cut by OUTPUT bytes, not by haystack: every wave takes one contiguous share of the output blob
//                   (a multiple of 32 KiB), finds the haystack and the piece its share begins in by binary search and copies
//                   pieces -- 16 bytes per lane, four loads in flight, stores aligned -- until the share is full.  One huge
//                   haystack (the README's "one big file") is spread over the whole chip, thousands of tiny ones do not
//                   get a wave each.
// ------------------------------------------------------------------------------------------
__global__ void k_replace_plan(ReplaceArgs R) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= R.n_haystacks) return;
    const uint64_t lo = R.prefix[h], hi = R.prefix[h + 1];
    const uint32_t len = uint32_t(R.offsets[h + 1] - R.offsets[h]);
    uint2 *plan = R.plan + lo;   // {end position, pattern}: at most one nominee per record
    uint32_t n = 0, last_pos = 0xFFFFFFFFu;
    for (uint64_t i = lo; i < hi; ++i) {
        const uint4 rec = R.records[i];  // {hay, pos, start, pattern}
        if (rec.y == last_pos) continue;              // this event has its nominee (records of one event: longest first)
        const uint64_t ro = R.rep_off[rec.w];
        if (ro == ~0ull) continue;                    // not a to-be-replaced pattern: maybe a shorter one of the event is
        last_pos = rec.y;
        const uint32_t new_start = rec.z;
        if (R.mode == 2) {                            // MF_REPLACE_MODE_LAZY (replace.c:236-250; the cursor is 0 while one text is scanned)
            if (n > 0 && new_start < plan[n - 1].x) continue;
        } else {                                      // DEFAULT / NORMAL (:252-268)
            while (n > 0) {
                const uint2 prev = plan[n - 1];
                const uint32_t prev_start = prev.x - uint32_t(R.pattern_length[prev.y]);
                if (new_start <= prev_start) --n; else break;
            }
        }
        plan[n++] = make_uint2(rec.y, rec.w);
    }
    // piece k = [text between the cursor and nominee k's start][its replacement], beginning at plan_out[k] of the output
    uint64_t out = 0;
    uint32_t cursor = 0;
    uint64_t *pout = R.plan_out + lo;
    for (uint32_t k = 0; k < n; ++k) {
        const uint2 nom = plan[k];
        const uint32_t start = nom.x - R.pattern_length[nom.y];
        pout[k] = out;
        if (start >= cursor) out += start - cursor;
        out += uint32_t(R.rep_off[nom.y]);
        cursor = nom.x;
    }
    R.tail_out[h] = out;
    if (len > cursor) out += len - cursor;
    R.nom_count[h] = n;
    R.out_len[h] = out;
}

// Exclusive scan of n 64-bit values into out[0..n] (out[n] = the total, also written to *total_host when that is not
// null: mapped host memory).  Same single-pass scheme as k_items / k_rec_scan; `sync` = {ticket, pad, status words},
// zeroed before the launch.
__global__ __launch_bounds__(kScanThreads) void k_scan_u64(const uint64_t *in, uint64_t n, uint64_t *out, uint32_t *ticket,
                                                           uint64_t *status, uint64_t *total_host) {
    __shared__ uint64_t wave_sums[kScanThreads / 64];
    __shared__ uint64_t s_before;
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t base = uint64_t(tile) * kScanTile + uint64_t(threadIdx.x) * kScanPerThread;
    if (uint64_t(tile) * kScanTile > n) return;
    uint64_t v[kScanPerThread];
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) v[i] = base + i < n ? in[base + i] : 0ull;
    uint64_t tile_sum;
    uint64_t run = tile_exclusive(v, wave_sums, tile_sum);
    if (threadIdx.x < 64) {
        const uint64_t bf = chain_exclusive(status, tile, tile_sum);
        if (threadIdx.x == 0) s_before = bf;
    }
    __syncthreads();
    run += s_before;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        if (base + i <= n) out[base + i] = run;
        if (base + i == n && total_host) *total_host = run;
        run += v[i];
    }
}

// `n` bytes from src to dst by one wave: single bytes up to dst's 16-byte boundary, then 16 bytes per lane (the load may be
// unaligned: gaps move by the difference between pattern and replacement lengths), then the last bytes.
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint64_t n, uint32_t lane) {
    const uint64_t head = (16u - uint32_t(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    if (n < 64u + head) {
        for (uint64_t i = lane; i < n; i += 64u) dst[i] = src[i];
        return;
    }
    if (lane < head) dst[lane] = src[lane];
    const uint64_t body = (n - head) >> 4;
    uint4 *d16 = reinterpret_cast<uint4 *>(dst + head);
    const uint8_t *s = src + head;
    // four loads in flight per lane (4 KiB per wave and step): one at a time, every 1 KiB step waited for its own round trip
    uint64_t i = lane;
    for (; i + 192u < body; i += 256u) {
        const uint4 v0 = load16u(s + (i << 4)), v1 = load16u(s + ((i + 64u) << 4)), v2 = load16u(s + ((i + 128u) << 4)),
                    v3 = load16u(s + ((i + 192u) << 4));
        d16[i] = v0;
        d16[i + 64u] = v1;
        d16[i + 128u] = v2;
        d16[i + 192u] = v3;
    }
    for (; i < body; i += 64u) d16[i] = load16u(s + (i << 4));
    const uint64_t done = head + (body << 4);
    if (done + lane < n) dst[done + lane] = src[done + lane];
}

constexpr uint64_t kEmitRange = 32u * 1024u;   // a wave's share of the output is a multiple of this

__global__ __launch_bounds__(256) void k_replace_emit(ReplaceArgs R, uint64_t cap) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = uint64_t(gridDim.x) * (blockDim.x / 64u);
    const uint64_t total = R.out_off[R.n_haystacks];
    if (total > cap) return;   // the optimistic output buffer is too small: the host grows it and launches again
    // a wave's ranges are consecutive -- one binary search for its first byte, the haystacks behind it in order (strided over
    // the output, every 32 KiB began with 17 dependent loads: as long as the copy itself)
    const uint64_t per_wave = ((total + kEmitRange - 1) / kEmitRange + waves - 1) / waves * kEmitRange;
    for (uint64_t g = uint64_t(blockIdx.x) * (blockDim.x / 64u) + (threadIdx.x >> 6); g < waves && g * per_wave < total; g += waves) {
        uint64_t a = g * per_wave;
        const uint64_t b = a + per_wave < total ? a + per_wave : total;
        // the haystack whose output holds byte a: the LAST h with out_off[h] <= a (haystacks with empty outputs share an offset)
        uint64_t lo = 0, hi = R.n_haystacks;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (R.out_off[mid] <= a) lo = mid; else hi = mid;
        }
        uint32_t h = uint32_t(lo);
        // the gap [at, at + n_bytes) of a haystack's output, clipped to this wave's part [la, le) of it
        auto copy_clip = [&](uint8_t *out, uint64_t at, const uint8_t *src, uint64_t n_bytes, uint64_t la, uint64_t le) {
            const uint64_t p0 = at > la ? at : la, p1 = at + n_bytes < le ? at + n_bytes : le;
            if (p0 < p1) wave_copy(out + p0, src + (p0 - at), p1 - p0, lane);
        };
        auto rl32 = [](uint32_t v, uint32_t j) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(j))); };
        auto rl64 = [&](uint64_t v, uint32_t j) { return uint64_t(rl32(uint32_t(v >> 32), j)) << 32 | rl32(uint32_t(v), j); };
        while (a < b) {
            // The bookkeeping of the NEXT 64 haystacks, a lane each, three dependent rounds for all of them -- one haystack at a
            // time the wave sat out three round trips before every 8-KiB piece, about as long as the piece's copy.  Lane j:
            // haystack h + j; with at most one nominee (the usual text) everything its pieces need is in these registers.
            const uint32_t hj = h + lane < R.n_haystacks ? h + lane : R.n_haystacks - 1u;
            const uint64_t v_ob = R.out_off[hj], v_oe = R.out_off[hj + 1], v_tb = R.offsets[hj], v_te = R.offsets[hj + 1];
            const uint64_t v_pf = R.prefix[hj], v_tail = R.tail_out[hj];
            const uint32_t v_n = R.nom_count[hj];
            uint2 v_plan = make_uint2(0u, 0u);
            uint64_t v_pout = 0, v_ro = 0;
            uint32_t v_plen = 0;
            if (v_n >= 1u) {
                v_plan = R.plan[v_pf];
                v_pout = R.plan_out[v_pf];
                v_plen = R.pattern_length[v_plan.y];
                v_ro = R.rep_off[v_plan.y];
            }
            for (uint32_t j = 0; j < 64u && a < b; ++j, ++h) {
                if (h >= R.n_haystacks) { a = b; break; }   // (cannot happen: b <= the last haystack's end)
                const uint64_t hb = rl64(v_ob, j), he = rl64(v_oe, j);
                if (he <= a) continue;
                const uint64_t seg_end = b < he ? b : he;
                const uint64_t la = a - hb, le = seg_end - hb;      // this range inside haystack h's output
                const uint64_t tb = rl64(v_tb, j);
                const uint8_t *text = R.blob + tb;
                const uint32_t len = uint32_t(rl64(v_te, j) - tb);
                const uint32_t n = rl32(v_n, j);
                const uint64_t pf = rl64(v_pf, j), tail_at = rl64(v_tail, j);
                uint8_t *out = R.out_blob + hb;
                if (n <= 1u) {
                    uint32_t cursor = 0;
                    if (n == 1u) {   // [text before the nominee][its replacement]
                        const uint32_t end = rl32(v_plan.x, j), start = end - rl32(v_plen, j);
                        const uint64_t at = rl64(v_pout, j), ro = rl64(v_ro, j);
                        copy_clip(out, at, text, start, la, le);
                        if (uint32_t(ro)) copy_clip(out, at + start, R.rep_bytes + (ro >> 32), uint32_t(ro), la, le);
                        cursor = end;
                    }
                    copy_clip(out, tail_at, text + cursor, len > cursor ? len - cursor : 0u, la, le);   // the text behind it
                } else {
                    const uint2 *plan = R.plan + pf;
                    const uint64_t *pout = R.plan_out + pf;
                    // first piece that reaches past la: the last k <= n with begin(k) <= la (begin(n) = the tail's)
                    uint32_t klo = 0, khi = n + 1u;   // begin(klo) <= la < begin(khi)
                    while (khi - klo > 1u) {
                        const uint32_t mid = (klo + khi) >> 1;
                        const uint64_t at = mid < n ? pout[mid] : tail_at;
                        if (at <= la) klo = mid; else khi = mid;
                    }
                    for (uint32_t k = klo; k <= n; ++k) {
                        const uint64_t at = k < n ? pout[k] : tail_at;
                        if (at >= le) break;
                        const uint32_t cursor = k ? plan[k - 1].x : 0u;
                        uint64_t gap_len, rep_len = 0;
                        const uint8_t *rep = nullptr;
                        if (k < n) {
                            const uint2 nom = plan[k];
                            const uint32_t start = nom.x - R.pattern_length[nom.y];
                            gap_len = start >= cursor ? start - cursor : 0u;
                            const uint64_t ro = R.rep_off[nom.y];
                            rep = R.rep_bytes + (ro >> 32);
                            rep_len = uint32_t(ro);
                        } else {
                            gap_len = len > cursor ? len - cursor : 0u;
                        }
                        copy_clip(out, at, text + cursor, gap_len, la, le);
                        if (rep_len) copy_clip(out, at + gap_len, rep, rep_len, la, le);
                    }
                }
                a = seg_end;
            }
        }
    }
}

void launch_replace_plan(const ReplaceArgs &R, hipStream_t st) {
    if (R.n_haystacks) hipLaunchKernelGGL(k_replace_plan, dim3((R.n_haystacks + 255) / 256), dim3(256), 0, st, R);
}
void launch_scan_u64(const uint64_t *in, uint64_t n, uint64_t *out, uint32_t *sync, uint64_t *total_host, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_u64, dim3(scan_tiles(n + 1)), dim3(kScanThreads), 0, st, in, n, out, sync, reinterpret_cast<uint64_t *>(sync + 2), total_host);
}
void launch_replace_emit(const ReplaceArgs &R, uint64_t cap, hipStream_t st) {
    if (!R.n_haystacks) return;
    hipLaunchKernelGGL(k_replace_emit, dim3(uint32_t(num_cus()) * 8u), dim3(256), 0, st, R, cap);
}

// ------------------------------------------------------------------------------------------
// Split pattern sets (engine.cpp): a set that as a whole is not eligible for the anchor filter -- a few patterns under 7 B
// among thousands of long ones -- is matched as TWO automata over the same batch, the long patterns through the filter, the
// rest through the table walk, and the two match arrays are merged here.  All patterns of part A are longer than all of
// part B, and the records of one position come longest pattern first (node.c:424-441): per haystack the merged order is
// position ascending, at equal position A's records before B's.  A record's final slot is its own index plus the number of
// records of the other part that go before it, found by binary search in that part's segment of the haystack: one thread per
// record, no serial merge.  findAll = false (all records of the FIRST event only): the parts each kept their own first
// event; the earlier position wins, at equal positions both stay.
// ------------------------------------------------------------------------------------------
__global__ void k_merge_counts(const uint64_t *pa, const uint4 *ra, const uint64_t *pb, const uint4 *rb, uint32_t nh, uint32_t first_only,
                               uint64_t *counts) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    const uint64_t a0 = pa[h], a1 = pa[h + 1], b0 = pb[h], b1 = pb[h + 1];
    uint64_t n = (a1 - a0) + (b1 - b0);
    if (first_only && a1 > a0 && b1 > b0) {
        const uint32_t posa = ra[a0].y, posb = rb[b0].y;
        if (posa < posb) n = a1 - a0; else if (posb < posa) n = b1 - b0;
    }
    counts[h] = n;
}

__global__ void k_merge_parts(const uint64_t *pa, const uint4 *ra, uint64_t ta, const uint64_t *pb, const uint4 *rb, uint64_t tb, uint32_t nh,
                              uint32_t first_only, const uint64_t *out_prefix, uint4 *out) {
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= ta + tb) return;
    const bool is_a = g < ta;
    const uint4 rec = is_a ? ra[g] : rb[g - ta];
    const uint32_t h = rec.x;
    const uint64_t a0 = pa[h], a1 = pa[h + 1], b0 = pb[h], b1 = pb[h + 1];
    if (first_only && a1 > a0 && b1 > b0) {   // both parts have a first event: the later one goes
        const uint32_t posa = ra[a0].y, posb = rb[b0].y;
        if ((is_a && posb < posa) || (!is_a && posa < posb)) return;
        if (posa != posb) {   // the other part's event was dropped: this part's records are the haystack's
            out[out_prefix[h] + (is_a ? g - a0 : g - ta - b0)] = rec;
            return;
        }
    }
    // records of the other part that go before this one: strictly earlier positions for A, earlier or equal for B
    const uint4 *other = is_a ? rb : ra;
    uint64_t lo = is_a ? b0 : a0, hi = is_a ? b1 : a1;
    const uint64_t seg0 = lo;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint32_t p = other[mid].y;
        if (is_a ? p < rec.y : p <= rec.y) lo = mid + 1; else hi = mid;
    }
    out[out_prefix[h] + (is_a ? g - a0 : g - ta - b0) + (lo - seg0)] = rec;
}

// find_all: the merged prefix is the sum of the parts' (no scan needed)
__global__ void k_merge_prefix_sum(const uint64_t *pa, const uint64_t *pb, uint32_t nh, uint64_t *out_prefix) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h <= nh) out_prefix[h] = pa[h] + pb[h];
}

void launch_merge_counts(const uint64_t *pa, const void *ra, const uint64_t *pb, const void *rb, uint32_t nh, uint32_t first_only, uint64_t *counts,
                         hipStream_t st) {
    if (nh) hipLaunchKernelGGL(k_merge_counts, dim3((nh + 255) / 256), dim3(256), 0, st, pa, static_cast<const uint4 *>(ra), pb,
                               static_cast<const uint4 *>(rb), nh, first_only, counts);
}
void launch_merge_prefix_sum(const uint64_t *pa, const uint64_t *pb, uint32_t nh, uint64_t *out_prefix, hipStream_t st) {
    hipLaunchKernelGGL(k_merge_prefix_sum, dim3((nh + 256) / 256), dim3(256), 0, st, pa, pb, nh, out_prefix);
}
void launch_merge_parts(const uint64_t *pa, const void *ra, uint64_t ta, const uint64_t *pb, const void *rb, uint64_t tb, uint32_t nh,
                        uint32_t first_only, const uint64_t *out_prefix, void *out, hipStream_t st) {
    if (ta + tb == 0) return;
    hipLaunchKernelGGL(k_merge_parts, dim3(uint32_t((ta + tb + 255) / 256)), dim3(256), 0, st, pa, static_cast<const uint4 *>(ra), ta, pb,
                       static_cast<const uint4 *>(rb), tb, nh, first_only, out_prefix, static_cast<uint4 *>(out));
}

// ------------------------------------------------------------------------------------------
// k_gather_pack: this rank's half of the all-gatherv of match arrays (engine.cpp acmi_comm_allgather_result).
// Ranks own contiguous haystack ranges in rank order, so the global arrays are the ranks' arrays back to back: a rank
// writes its own slice of them -- haystack ids made global, prefix words moved by the records of the ranks before it --
// and the per-rank broadcasts that follow fill in the other slices, each exactly its size.  Nothing is unpacked
// afterwards.  `closing` = the global prefix array's last word (the total), written by every rank for itself.
// ------------------------------------------------------------------------------------------
__global__ void k_gather_pack(const uint64_t *prefix, const uint4 *records, uint64_t nh, uint64_t nrec, uint64_t hay_base,
                              uint64_t rec_base, uint64_t *out_prefix, uint4 *out_records, uint64_t *closing, uint64_t total_rec) {
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g < nrec) {
        uint4 v = records[g];
        v.x += uint32_t(hay_base);
        out_records[g] = v;
    }
    if (g < nh) out_prefix[g] = prefix[g] + rec_base;
    if (g == 0) *closing = total_rec;
}

void launch_gather_pack(const uint64_t *prefix, const uint4 *records, uint64_t nh, uint64_t nrec, uint64_t hay_base, uint64_t rec_base,
                        uint64_t *out_prefix, uint4 *out_records, uint64_t *closing, uint64_t total_rec, hipStream_t st) {
    const uint64_t threads = std::max<uint64_t>(1, std::max(nrec, nh));
    hipLaunchKernelGGL(k_gather_pack, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, st, prefix, records, nh, nrec, hay_base,
                       rec_base, out_prefix, out_records, closing, total_rec);
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
void launch_items(const BatchArgs &B, uint64_t blob_bytes, uint64_t max_items, uint64_t *item_base, uint32_t *isync, hipStream_t st) {
    // isync (zeroed by the caller): {ticket, bad-offsets flag, status words of the scan}
    uint32_t *ticket = isync, *bad = isync + 1;
    uint64_t *status = reinterpret_cast<uint64_t *>(isync + 2);
    // long haystacks (more than two items each on average): the item table is written by a kernel of its own
    const bool fill_later = B.n_haystacks && blob_bytes / B.n_haystacks > 2ull * B.chunk_bytes;
    hipLaunchKernelGGL(k_items, dim3(scan_tiles(uint64_t(B.n_haystacks) + 1)), dim3(kScanThreads), 0, st, B, blob_bytes, item_base, ticket, bad, status, fill_later);
    if (fill_later) {
        const uint64_t blocks = std::min<uint64_t>((max_items + 255) / 256, 256u * 32u);
        hipLaunchKernelGGL(k_item_fill, dim3(uint32_t(blocks)), dim3(256), 0, st, B);
    }
}

int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

template <typename E, int MODE, int ILP, int G>
static hipError_t launch_lds_t(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint32_t lds_bytes,
                               hipStream_t st) {
    auto fn = k_scan_lds<E, MODE, ILP, G>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
    if (e != hipSuccess) return e;
    // persistent: at most one workgroup per CU (the tile fills the CU's LDS), fewer for small batches
    const uint64_t per_block = uint64_t(kLdsThreads) * ILP;
    const uint64_t want = (max_items + per_block - 1) / per_block;
    const uint32_t blocks = B.persistent ? uint32_t(std::min<uint64_t>(want, uint64_t(num_cus()))) : uint32_t(want);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(kLdsThreads), lds_bytes, st, A, B);
    return hipGetLastError();
}

template <typename E, int MODE>
static hipError_t launch_lds_m(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint32_t lds, int ilp,
                               int group_quads, hipStream_t st) {
    // ILP 1 reads whole 128-byte lines per lane (8 quads, double-buffered = 64 VGPRs); ILP 2 has
    // registers for 64-byte groups only
    if (ilp == 2) return launch_lds_t<E, MODE, 2, 4>(A, B, max_items, lds, st);
    if (group_quads == 4) return launch_lds_t<E, MODE, 1, 4>(A, B, max_items, lds, st);
    return launch_lds_t<E, MODE, 1, 8>(A, B, max_items, lds, st);
}

// *variant (acmi_get_stat "anchor_variant"): the template that ran -- QW | BIG << 3 | SHORT << 4 | TINY << 5
static hipError_t launch_anchor(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, hipStream_t st, uint32_t *variant) {
    if (variant) {
        const bool big = A.anchor_mode == 2, tiny = B.tiny_items != 0;
        const bool shrt = !big && (tiny || B.short_items != 0);
        *variant = (A.anchor_q / 4u) | (big ? 8u : 0u) | (shrt ? 16u : 0u) | (tiny ? 32u : 0u) | (A.anchor_dense ? 64u : 0u) | (A.anchor_dense && A.anchor_dense_hot ? 128u : 0u) | (A.anchor_slots == 2u ? 256u : 0u);
    }
    const uint64_t per_block = uint64_t(kLdsThreads / 64u) * B.wave_items;
    const uint64_t want = (max_items + per_block - 1) / per_block;
    const uint32_t blocks = B.persistent ? uint32_t(std::min<uint64_t>(want, uint64_t(num_cus()))) : uint32_t(want);
    auto go = [&](auto fn) {
        const uint32_t lds = kAnchorLdsBytes + (kLdsThreads / 64u) * kStageSlots * 16u;  // table + run staging = 160 KiB
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kLdsThreads), lds, st, A, B);
        return hipGetLastError();
    };
    if (A.anchor_dense) {  // windows at every byte offset (shortest pattern 4..6 B)
        if (A.anchor_slots == 2u) {
            if (A.anchor_dense_hot) {
                if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, true, true>);
                if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, true, true>);
                return go(k_scan_anchor<1, false, false, false, true, true, true>);
            }
            if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, false, true>);
            if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, false, true>);
            return go(k_scan_anchor<1, false, false, false, true, false, true>);
        }
        if (A.anchor_dense_hot) {
            if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, true>);
            if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, true>);
            return go(k_scan_anchor<1, false, false, false, true, true>);
        }
        if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true>);
        if (B.short_items) return go(k_scan_anchor<1, false, true, false, true>);
        return go(k_scan_anchor<1, false, false, false, true>);
    }
    if (B.tiny_items) {  // haystacks of a few hundred bytes: the variants whose waves pack one-row items G to a task
        if (A.anchor_mode == 2) {
            if (A.anchor_q == 4) return go(k_scan_anchor<1, true, false, true>);
            if (A.anchor_q == 8) return go(k_scan_anchor<2, true, false, true>);
            if (A.anchor_q == 12) return go(k_scan_anchor<3, true, false, true>);
            return go(k_scan_anchor<4, true, false, true>);
        }
        if (A.anchor_q == 4) return go(k_scan_anchor<1, false, true, true>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, false, true, true>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, false, true, true>);
        return go(k_scan_anchor<4, false, true, true>);
    }
    if (A.anchor_mode == 2) {
        if (A.anchor_q == 4) return go(k_scan_anchor<1, true, false>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, true, false>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, true, false>);
        return go(k_scan_anchor<4, true, false>);
    }
    if (B.short_items) {
        if (A.anchor_q == 4) return go(k_scan_anchor<1, false, true>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, false, true>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, false, true>);
        return go(k_scan_anchor<4, false, true>);
    }
    if (A.anchor_q == 4) return go(k_scan_anchor<1, false, false>);
    if (A.anchor_q == 8) return go(k_scan_anchor<2, false, false>);
    if (A.anchor_q == 12) return go(k_scan_anchor<3, false, false>);
    return go(k_scan_anchor<4, false, false>);
}

hipError_t launch_scan(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes,
                       int kernel, int class_mode, int ilp, int group_quads, hipStream_t st, uint32_t *anchor_variant) {
    if (anchor_variant) *anchor_variant = 0;
    if (max_items == 0) return hipSuccess;
    if (kernel == 2 && A.anchor_q != 0) return launch_anchor(A, B, max_items, st, anchor_variant);
    if (kernel == 0) {
        const uint32_t blocks = uint32_t((max_items + 255) / 256);
        if (entry_bytes == 2) hipLaunchKernelGGL(k_scan_simple<uint16_t>, dim3(blocks), dim3(256), 0, st, A, B);
        else hipLaunchKernelGGL(k_scan_simple<uint32_t>, dim3(blocks), dim3(256), 0, st, A, B);
        return hipGetLastError();
    }
    const uint32_t lds = ((A.lds_image_bytes + 15u) & ~15u) + 32u;
    if (entry_bytes == 2)
        return class_mode == 1 ? launch_lds_m<uint16_t, 1>(A, B, max_items, lds, ilp, group_quads, st)
                               : launch_lds_m<uint16_t, 0>(A, B, max_items, lds, ilp, group_quads, st);
    return class_mode == 1 ? launch_lds_m<uint32_t, 1>(A, B, max_items, lds, ilp, group_quads, st)
                           : launch_lds_m<uint32_t, 0>(A, B, max_items, lds, ilp, group_quads, st);
}

void launch_first_only(const BatchArgs &B, uint32_t *first_item, uint64_t max_items, hipStream_t st) {   // first_item: n_haystacks words of 0xFFFFFFFF
    if (B.n_haystacks == 0 || max_items == 0) return;
    const uint32_t blocks = uint32_t((max_items + 255) / 256);
    hipLaunchKernelGGL(k_first_only_mark, dim3(blocks), dim3(256), 0, st, B, first_item);
    hipLaunchKernelGGL(k_first_only_drop, dim3(blocks), dim3(256), 0, st, B, static_cast<const uint32_t *>(first_item));
}

void launch_expand(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint64_t *rec_base, void *out,
                   uint64_t cap, uint64_t *prefix, uint32_t *host_counters, uint32_t *next_sync, uint32_t next_sync_words,
                   uint32_t *ticket, uint64_t *status, const uint32_t *bad_offsets, bool scan, hipStream_t st,
                   uint32_t *next_isync, uint32_t next_isync_words) {
    const uint32_t blocks = uint32_t((max_items + 1 + kExpandTile - 1) / kExpandTile);
    if (scan)
        hipLaunchKernelGGL(k_expand<true>, dim3(blocks), dim3(kExpandThreads), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap, prefix,
                           host_counters, next_sync, next_sync_words, ticket, status, bad_offsets, next_isync, next_isync_words);
    else
        hipLaunchKernelGGL(k_expand<false>, dim3(blocks), dim3(kExpandThreads), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap, prefix,
                           host_counters, next_sync, next_sync_words, ticket, status, bad_offsets, next_isync, next_isync_words);
}

void launch_refill(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes,
                   const uint64_t *rec_base, void *out, uint64_t cap, int mode, hipStream_t st, int class_mode) {
    if (max_items == 0) return;
    if (mode == 3) {   // a batch of k_walk_items: the same walk again, a thread per overflowed item, writing records
        const uint32_t lds = kItemsTileBytes + (class_mode == 0 ? 512u : 0u);
        const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);   // (workgroups past the list's end leave at once)
        auto go = [&](auto fn) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B, rec_base, static_cast<uint4 *>(out), cap);
        };
        if (entry_bytes == 2) { if (class_mode == 1) go(k_walk_items<uint16_t, 1, 4, true>); else go(k_walk_items<uint16_t, 0, 4, true>); }
        else { if (class_mode == 1) go(k_walk_items<uint32_t, 1, 4, true>); else go(k_walk_items<uint32_t, 0, 4, true>); }
        return;
    }
    // mode 0: the wave kernel unless a lane's (Lmax-1)-byte warm-up would dwarf its kRefillS positions
    const bool wave = mode == 2 || (mode == 0 && A.max_pattern_length <= 64u);
    if (wave) {
        const uint64_t per_block = kRefillThreads / 64;
        // window = what 64 positions can produce at most (one pass per group), up to 1024 records: smaller
        // windows leave LDS for more resident waves
        const uint32_t nb = A.max_out_count <= 4u ? 256u : A.max_out_count <= 8u ? 512u : 1024u;
        const uint32_t lds_waves = refill_wave_lds(nb) * uint32_t(per_block);
        // workgroups per CU: what the staging windows allow, four at most (the rest of a workgroup's share of the LDS holds the
        // byte classes and the first rows of the table)
        const uint32_t per_cu = std::min<uint32_t>(4u, (160u * 1024u) / (lds_waves + 1024u));
        const uint32_t share = (160u * 1024u) / std::max<uint32_t>(per_cu, 1u);
        const uint32_t hot_bytes = (A.trap * A.row_bytes + 15u) & ~15u;
        const uint32_t tile_budget = std::min<uint32_t>(hot_bytes, share > lds_waves + 512u + 64u ? ((share - lds_waves - 512u - 64u) & ~15u) : 0u);
        const uint32_t lds = lds_waves + 512u + tile_budget;
        const uint32_t blocks = uint32_t(std::min<uint64_t>((max_items + per_block - 1) / per_block, uint64_t(num_cus()) * std::max<uint32_t>(per_cu, 1u)));
        auto go = [&](auto fn) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(kRefillThreads), lds, st, A, B, rec_base, static_cast<uint4 *>(out), cap, tile_budget);
        };
        if (entry_bytes == 2) {
            if (nb == 256u) go(k_refill_wave<uint16_t, 256>); else if (nb == 512u) go(k_refill_wave<uint16_t, 512>); else go(k_refill_wave<uint16_t, 1024>);
        } else {
            if (nb == 256u) go(k_refill_wave<uint32_t, 256>); else if (nb == 512u) go(k_refill_wave<uint32_t, 512>); else go(k_refill_wave<uint32_t, 1024>);
        }
        return;
    }
    const uint32_t blocks = uint32_t(std::min<uint64_t>((max_items + 63) / 64, 4096));
    if (entry_bytes == 2)
        hipLaunchKernelGGL(k_refill<uint16_t>, dim3(blocks), dim3(64), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap);
    else
        hipLaunchKernelGGL(k_refill<uint32_t>, dim3(blocks), dim3(64), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap);
}

}  // namespace acmi
