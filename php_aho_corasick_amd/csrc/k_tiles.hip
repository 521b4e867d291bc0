// k_tiles.hip -- k_walk_tiles: the exact walk with exact-sized, ordered output in ONE pass (kernel 6).
//
// What it replaces (reference: the callback of ac_trie_search hands out records one event at a time, in text order, with no
// notion of "how many" -- src/php_ahocorasick.c:542-589, src/multifast/ahocorasick.c:214-231): the other scan kernels put
// events into a fixed slab per work item, and an item with more events than its slab holds is WALKED A SECOND TIME by a
// drain.  Between "a match now and then" and "a match at every byte" -- a word list over prose, short patterns over their own
// alphabet: tens to hundreds of records per 8 KiB -- that second walk, the learned slab sizes and the per-item expansion were
// most of the call (round 3: 5 ms per GiB at 540 records per haystack, 0.05 of the roofline).
//
// Here the unit of work is small and the output is placed by a running sum instead of by a second pass:
//   * a work item (<= P x W bytes of one haystack) is walked by P lanes of a wave, W bytes each, entered from the root
//     ceil((Lmax-1)/16) blocks early (exact: the output set at a position depends on the last Lmax bytes only) or at
//     the haystack's first byte; a wave takes R = 64 / P consecutive items, a workgroup 16 R;
//   * a lane reads its own bytes (groups of 64 B requested back to back, one group ahead); neighbouring lanes hold
//     neighbouring bytes, so a wave's loads cover whole lines and every line is fetched once;
//   * an event {state, owner lane, byte index, records of this lane before it} goes to the wave's queue (global memory, L2
//     resident: written and read back by the same wave within microseconds), its slot from a ballot -- no slab, no per-item
//     capacity: a lane has at most one event per byte, the queue holds an event per 4 bytes, and a wave with more than that
//     (dense output) writes its records by walking its pieces again from the registers' point of view, the only second walk left;
//   * the records a lane's piece produces are summed over the wave, the wave totals over the workgroup, and the workgroup's
//     base in the output array comes from the workgroups before it by decoupled look-back (tickets: a tile only waits for
//     tiles that are already running) -- the same single-pass scan k_items and k_expand use, one status word per 16 R items;
//   * then the queue is expanded by ALL lanes, 64 events per round whatever their distribution over the pieces: an event's
//     records go to base + (records of the lanes before its owner) + (its own offset): the reference's order -- position
//     ascending, longest pattern first (node.c:424-441) -- without a sort and without knowing any count in advance.
// The per-item record counts are left for k_expand (per-haystack prefix, summary), which finds no event to expand.
#include <cstdlib>

#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kTilesThreads = 1024;
constexpr uint32_t kTilesWaves = kTilesThreads / 64;
constexpr uint32_t kTilesRing = 128;   // events a wave stages in LDS (flushed to its queue 64 at a time)

struct TileArgs {
    uint32_t W;           // bytes per lane piece: 16, 32 or 64
    uint32_t P;           // pieces (lanes) per item
    uint32_t R;           // items per wave = 64 / P
    uint32_t back;        // bytes a lane enters its piece early = Lmax - 1
    uint32_t qcap;        // queue entries per wave (a multiple of 64)
    uint32_t tile_bytes;  // LDS budget of the table tile (+ record counts)
    uint32_t flags;       // analysis only (ACMI_TILES_DEBUG): 1 no expansion, 2 no events
    uint2 *queue;         // grid waves x qcap entries
    uint32_t *ticket;     // zeroed
    uint64_t *status;     // zeroed, one word per workgroup tile
    uint4 *out;
    uint64_t cap;         // records `out` holds
};

__device__ __forceinline__ uint32_t tile_wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t tile_wave_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// One lane's piece: what it walks and what it reports.
struct Piece {
    const uint8_t *pb;    // 16-byte aligned address of the first block it walks
    uint32_t nblk;        // blocks it walks (0: nothing to do)
    uint32_t start_j;     // byte index (from pb) of the first byte that is walked: the state is the root before it
    uint32_t rep_lo, rep_len;   // bytes [rep_lo, rep_lo + rep_len) report
    uint32_t pos_bias;    // position of byte jj = pos_bias + jj
    uint32_t hay;
};

// The wave's event queue: a ring in LDS, flushed 64 entries at a time to the wave's region in global memory (which stays in
// L2: the wave reads it back microseconds later).  A store per event from inside the walk made every step wait for the store
// before (the stores share the counter the cold rows' loads are waited on with).
struct TileQueue {
    uint2 *ring;          // LDS, kTilesRing entries
    uint2 *spill;         // global, qcap entries
    uint32_t qcap;
    uint32_t total;       // events so far (wave-uniform)
    uint32_t flushed;     // of which in `spill` (a multiple of 64)
};

__device__ __forceinline__ void tiles_flush(TileQueue &Q, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint2 e = Q.ring[(Q.flushed + lane) & (kTilesRing - 1u)];
    if (Q.flushed + lane < Q.qcap) Q.spill[Q.flushed + lane] = e;
    Q.flushed += 64u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One exact transition of a lane: the LDS tile's row when the state has one, else the global table / the compact records.
// ALLHOT: every state has its row in the tile -- no test, no cold path compiled in.
template <typename E, bool ALLHOT>
__device__ __forceinline__ uint32_t tiles_step(const DevAutomaton &A, const uint8_t *tile, uint32_t tile_rows, uint32_t st, uint32_t byte,
                                               uint32_t coff) {
    if constexpr (ALLHOT) return *reinterpret_cast<const E *>(tile + __umul24(st, A.row_bytes) + coff);
    uint32_t w = A.trap;
    if (st < tile_rows) w = *reinterpret_cast<const E *>(tile + __umul24(st, A.row_bytes) + coff);
    if (w >= A.trap) w = cold_step<E>(A, st, byte, coff);
    return w;
}

template <bool ALLHOT>
__device__ __forceinline__ bool tiles_final(const DevAutomaton &A, uint32_t s) {
    if constexpr (ALLHOT) return s >= A.f1;   // (no cold states, and the trap row is never entered)
    return is_final(A, s);
}

// The walk of one piece, counting: events go to the wave's queue, n_rec sums the records they stand for.  All 64 lanes of
// the wave call this together and run the same number of steps (a lane past its own bytes walks on without reporting).
template <typename E, int MODE, bool ALLHOT>
__device__ __forceinline__ void tiles_walk(const DevAutomaton &A, const Piece &pc, const uint16_t *cls_lds, const uint8_t *tile,
                                           uint32_t tile_rows, const uint32_t *cnt_lds, uint32_t cnt_rows, TileQueue &Q, uint32_t lane,
                                           uint32_t &n_rec, bool events) {
    uint32_t st = 0;
    auto event = [&](uint32_t jj) __attribute__((always_inline)) {
        const bool ev = jj - pc.rep_lo < pc.rep_len && tiles_final<ALLHOT>(A, st);
        // the lanes that have an event at this step: the ballot is taken by ALL lanes, so that the queue index stays wave-uniform
        const uint64_t m = __ballot(ev);
        if (m) {
            if (ev) {
                const uint32_t c = (ALLHOT || st < cnt_rows) ? cnt_lds[st] : A.out_count[st];
                const uint32_t my = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                if (events) Q.ring[(Q.total + my) & (kTilesRing - 1u)] = make_uint2(st | lane << 24, jj | n_rec << 9);
                n_rec += c;
            }
            Q.total += uint32_t(__builtin_popcountll(m));
            if (Q.total - Q.flushed >= 64u) tiles_flush(Q, lane);
        }
    };
    // A piece is its warm-up and at most 64 bytes = one group of a line: all of it is requested at once, back to back (a line
    // fetched in pieces over the microseconds of a walk comes from HBM again for every piece, k_walk_items' lesson); what a
    // longer warm-up or an unaligned last piece adds is requested as the buffers free up.
    const uint4 *q = reinterpret_cast<const uint4 *>(pc.pb);
    const uint32_t last = pc.nblk ? pc.nblk - 1u : 0u;
    const uint32_t blocks = tile_wave_max(pc.nblk);
    if (blocks == 0u) return;
    const uint4 first = q[pc.start_j < 16u ? 0u : (last ? 1u : 0u)];   // (start_j == 16: block 0 is virtual, see the caller)
    uint4 b0 = q[1u < last ? 1u : last], b1 = q[2u < last ? 2u : last], b2 = q[3u < last ? 3u : last], b3 = q[4u < last ? 4u : last];
    // the first block: the walk begins inside it (Lmax - 1 bytes before the piece, or at the haystack's first byte) -- byte by
    // byte, the state held in the root up to the lane's own first byte; not unrolled
    const uint32_t jmin = tile_wave_min(pc.nblk ? pc.start_j : 16u);
#pragma unroll 1
    for (uint32_t jj = jmin; jj < 16u; ++jj) {
        const uint32_t wd = jj < 8u ? (jj < 4u ? first.x : first.y) : (jj < 12u ? first.z : first.w);
        const uint32_t byte = (wd >> (8u * (jj & 3u))) & 0xffu;
        const uint32_t w = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, byte, class_offset<E, MODE>(A, cls_lds, byte));
        st = jj < pc.start_j ? 0u : w;
        event(jj);
    }
    // the blocks behind it: sixteen unrolled steps per block, the buffers rotating; the column offsets of a dword's four bytes
    // are worked out together (SWAR3: a handful of operations for all four; TABLE: four independent LDS reads)
#pragma unroll 1
    for (uint32_t k = 1u; k < blocks; ++k) {
        const uint4 cur = b0;
        b0 = b1; b1 = b2; b2 = b3;
        const uint32_t kn = k + 4u;
        if (kn < blocks) b3 = q[kn < last ? kn : last];
        const uint32_t j0 = k * 16u;
        auto dword = [&](uint32_t wd, uint32_t jj) __attribute__((always_inline)) {
            uint32_t c0, c1, c2, c3;
            if constexpr (MODE == 1) {
                const uint32_t c4 = swar_columns<E>(A, wd);
                c0 = c4 & 0xffu; c1 = (c4 >> 8) & 0xffu; c2 = (c4 >> 16) & 0xffu; c3 = c4 >> 24;
            } else {
                c0 = class_offset<E, MODE>(A, cls_lds, wd & 0xffu); c1 = class_offset<E, MODE>(A, cls_lds, (wd >> 8) & 0xffu);
                c2 = class_offset<E, MODE>(A, cls_lds, (wd >> 16) & 0xffu); c3 = class_offset<E, MODE>(A, cls_lds, wd >> 24);
            }
            st = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, wd & 0xffu, c0); event(jj);
            st = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, (wd >> 8) & 0xffu, c1); event(jj + 1u);
            st = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, (wd >> 16) & 0xffu, c2); event(jj + 2u);
            st = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, wd >> 24, c3); event(jj + 3u);
        };
        dword(cur.x, j0); dword(cur.y, j0 + 4u); dword(cur.z, j0 + 8u); dword(cur.w, j0 + 12u);
    }
}

// The same walk writing records in place: for the waves whose events did not fit their queue (dense output, rare, and bound by
// its stores): every lane on its own, bytes straight from memory (the lines are in L2: the wave has just read them).
template <typename E, int MODE, bool ALLHOT>
__device__ __forceinline__ void tiles_emit(const DevAutomaton &A, const Piece &pc, const uint16_t *cls_lds, const uint8_t *tile,
                                           uint32_t tile_rows, uint4 *out, uint64_t wr) {
    uint32_t st = 0;
    const uint32_t end = pc.rep_lo + pc.rep_len;
#pragma unroll 1
    for (uint32_t jj = pc.start_j; jj < end; ++jj) {
        const uint32_t byte = load8g(pc.pb + jj);
        st = tiles_step<E, ALLHOT>(A, tile, tile_rows, st, byte, class_offset<E, MODE>(A, cls_lds, byte));
        if (jj >= pc.rep_lo && tiles_final<ALLHOT>(A, st)) wr = emit_event(A, out, wr, pc.hay, pc.pos_bias + jj, st);
    }
}

// An event's records, longest pattern first, at `dst`.
__device__ __forceinline__ void tiles_expand(const DevAutomaton &A, uint4 *dst, uint32_t hay, uint32_t pos, uint32_t s, uint32_t c) {
    if (A.out_list != nullptr) {
        const uint2 *ol = A.out_list + A.out_off[s];
        for (uint32_t k = 0; k < c; ++k) {
            const uint2 o = ol[k];
            dst[k] = make_uint4(hay, pos, pos - o.y, o.x);
        }
    } else {
        (void)emit_event(A, dst, 0, hay, pos, s);
    }
}

// Two workgroups per CU (64 registers, 8 waves per SIMD): what hides the round trips of rows that are not in LDS is the number
// of waves, not the size of the tile -- one workgroup with the whole LDS for its tile (twice the rows, 80-odd registers, no
// spills) measured 8-12 % slower on a word list and on the headline set with 64 plants per haystack.
template <typename E, int MODE, bool ALLHOT>
__global__ __launch_bounds__(kTilesThreads, 8) void k_walk_tiles(DevAutomaton A, BatchArgs B, TileArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_tile, s_walked;
    __shared__ uint64_t s_before;
    __shared__ uint64_t s_wave_tot[kTilesWaves];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    // LDS: [the waves' event rings] [512 B byte classes (TABLE mode)] [tile rows] [their record counts]
    const uint32_t ring_bytes = kTilesWaves * kTilesRing * 8u;
    const uint32_t cls_bytes = (MODE == 0) ? 512u : 0u;
    const uint32_t tile_rows = A.row_bytes ? (T.tile_bytes / (A.row_bytes + 4u) < A.trap ? T.tile_bytes / (A.row_bytes + 4u) : A.trap) : 0u;
    const uint16_t *cls_lds = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem + ring_bytes) : nullptr;
    const uint8_t *tile = smem + ring_bytes + cls_bytes;
    const uint32_t tile_bytes = (tile_rows * A.row_bytes + 15u) & ~15u;
    const uint32_t cnt_rows = tile_rows;
    const uint32_t *cnt_lds = reinterpret_cast<const uint32_t *>(tile + tile_bytes);
    {
        if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem + ring_bytes)[threadIdx.x] = A.class_off[threadIdx.x];
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + ring_bytes + cls_bytes);
        const uint32_t n16 = tile_bytes >> 4;
        for (uint32_t t = threadIdx.x; t < n16; t += kTilesThreads) tdst[t] = tsrc[t];
        uint32_t *cdst = reinterpret_cast<uint32_t *>(smem + ring_bytes + cls_bytes + tile_bytes);
        for (uint32_t t = threadIdx.x; t < cnt_rows; t += kTilesThreads) cdst[t] = A.out_count[t];
    }
    if (threadIdx.x == 0) s_walked = 0;
    TileQueue Q;
    Q.ring = reinterpret_cast<uint2 *>(smem) + wave * kTilesRing;
    Q.spill = T.queue + (uint64_t(blockIdx.x) * kTilesWaves + wave) * T.qcap;
    Q.qcap = T.qcap;
    const uint32_t r = lane / T.P, qi = lane - r * T.P;   // item of the wave, piece of the item
    const uint64_t per_group = uint64_t(kTilesWaves) * T.R;
    const uintptr_t beta = reinterpret_cast<uintptr_t>(B.blob) & 15u;
    const uint8_t *abase = B.blob - beta;   // 16-byte aligned
    uint32_t walked16 = 0;
#pragma unroll 1
    for (;;) {
        __syncthreads();   // (the tile is staged; the previous round's s_tile / s_before have been read)
        if (threadIdx.x == 0) s_tile = atomicAdd(T.ticket, 1u);
        __syncthreads();
        const uint32_t t = s_tile;
        const uint64_t first_item = uint64_t(t) * per_group;
        if (first_item >= total_items) break;
        const uint64_t item = first_item + uint64_t(wave) * T.R + r;
        const bool have = r < T.R && item < total_items;
        Piece pc;
        pc.pb = A.pad16;
        pc.nblk = 0;
        pc.start_j = pc.rep_lo = pc.rep_len = pc.pos_bias = pc.hay = 0;
        if (have) {
            const uint4 rec = B.item_rec[item];
            ItemDesc it;
            unpack_item(rec.x, rec.y, rec.z, rec.w, it);
            pc.hay = B.item_hay[item];
            // x coordinates: bytes from G, the 16-byte boundary at or below the item's first byte
            const uint64_t zb = it.begin + beta;
            const uint64_t G = zb & ~uint64_t(15);
            const uint32_t d0 = uint32_t(zb - G);
            const uint32_t a = qi * T.W;
            const uint32_t lo = a > d0 ? a : d0;
            const uint32_t end = d0 + it.len;
            const uint32_t hi = (qi + 1u == T.P) ? end : (a + T.W < end ? a + T.W : end);
            if (lo < hi) {
                const int64_t hs = int64_t(d0) - int64_t(it.pos0);   // the haystack's first byte
                int64_t xs = int64_t(lo) - int64_t(T.back);           // Lmax - 1 bytes early: exact from there on
                if (xs < hs) xs = hs;
                // (the first block is walked byte by byte from start_j on, by the whole wave as long as its earliest lane: a lane
                // whose walk begins ON a block boundary -- a haystack that starts aligned -- takes the block below as its first,
                // virtually: start_j = 16, nothing of it is loaded or walked)
                int64_t xb = xs & ~int64_t(15);
                if (xb == xs) xb -= 16;
                pc.pb = abase + G + xb;
                pc.start_j = uint32_t(xs - xb);
                pc.rep_lo = uint32_t(int64_t(lo) - xb);
                pc.rep_len = hi - lo;
                pc.nblk = (pc.rep_lo + pc.rep_len + 15u) >> 4;
                pc.pos_bias = uint32_t(xb) - d0 + it.pos0 + 1u;
                walked16 += pc.nblk;
            }
        }
        uint32_t n_rec = 0;
        Q.total = Q.flushed = 0;
        tiles_walk<E, MODE, ALLHOT>(A, pc, cls_lds, tile, tile_rows, cnt_lds, cnt_rows, Q, lane, n_rec, (T.flags & 2u) == 0u);
        // records: exclusive prefix over the wave's lanes, the items' totals, the wave's total
        const uint32_t incl = wave_inclusive_sum(n_rec);
        const uint32_t excl = incl - n_rec;
        const uint32_t wave_tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        {
            const uint32_t last_lane = r * T.P + T.P - 1u;
            const uint32_t item_tot = uint32_t(__shfl(int(incl), int(last_lane < 64u ? last_lane : 63u), 64)) - uint32_t(__shfl(int(excl), int(r * T.P < 64u ? r * T.P : 63u), 64));
            if (have && qi == 0u && item_tot) B.rec_count[item] = item_tot;
        }
        if (lane == 0) s_wave_tot[wave] = wave_tot;
        __syncthreads();
        uint64_t wave_base = 0, group_tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kTilesWaves; ++w) {
            const uint64_t x = s_wave_tot[w];
            if (w < wave) wave_base += x;
            group_tot += x;
        }
        if (wave == 0) {
            const uint64_t bf = chain_exclusive(T.status, t, group_tot);
            if (lane == 0) s_before = bf;
        }
        __syncthreads();
        const uint64_t base = s_before + wave_base;
        if (wave_tot == 0u || base + wave_tot > T.cap || (T.flags & 1u)) continue;   // (too small a buffer: the host grows it and runs the batch again)
        if (Q.total <= Q.qcap) {
            // every lane expands events, 64 per round whatever their distribution over the pieces: an event's slot = the records
            // of the lanes before its owner + its own offset.  Rounds [0, flushed) come from the wave's queue in memory (written by
            // this wave's own lanes, through the CU's write-through L1: visible to them once the stores have landed; an
            // agent-scope release here wrote the whole L2 back, 5 ms per GiB), the rest from the ring.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the flushes have landed: a wait, no cache maintenance)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 1
            for (uint32_t e0 = 0; e0 < Q.total; e0 += 64u) {
                const uint32_t e = e0 + lane;
                const bool valid = e < Q.total;
                uint2 ent = make_uint2(0u, 0u);
                if (e0 < Q.flushed) {
                    ent = Q.spill[e];
                } else if (valid) {
                    ent = Q.ring[e & (kTilesRing - 1u)];
                }
                const uint32_t owner = ent.x >> 24, s = ent.x & 0xFFFFFFu;
                const uint32_t slot = uint32_t(__shfl(int(excl), int(owner), 64)) + (ent.y >> 9);
                const uint32_t pos = uint32_t(__shfl(int(pc.pos_bias), int(owner), 64)) + (ent.y & 511u);
                const uint32_t hay = uint32_t(__shfl(int(pc.hay), int(owner), 64));
                if (valid) tiles_expand(A, T.out + base + slot, hay, pos, s, (ALLHOT || s < cnt_rows) ? cnt_lds[s] : A.out_count[s]);
            }
        } else if (pc.nblk) {
            // more events than the queue holds (dense output): every lane walks its piece again and writes its records in place
            tiles_emit<E, MODE, ALLHOT>(A, pc, cls_lds, tile, tile_rows, T.out, base + excl);
        }
    }
    // statistic: 16-byte blocks walked (what the automatic kernel choice reads), summed per workgroup
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) walked16 += __shfl_xor(walked16, d, 64);
    if (lane == 0 && walked16) atomicAdd(&s_walked, walked16);
    __syncthreads();
    if (threadIdx.x == 0 && s_walked) atomicAdd(&B.counters[8u + (blockIdx.x & 7u)], s_walked);
}

// Geometry of a batch (see the header of this file): false when the patterns are too long for per-lane warm-ups.
bool tiles_geometry(uint32_t max_pattern_length, uint32_t n_states, uint64_t blob_bytes, uint64_t n_haystacks, uint32_t *W, uint32_t *P) {
    if (max_pattern_length == 0 || n_haystacks == 0 || n_states >= (1u << 24)) return false;   // (a queue entry has 24 bits for the state)
    const uint32_t wb = (max_pattern_length - 1u + 15u) / 16u;
    if (wb > 8u) return false;   // Lmax > 129: a lane would walk two bytes of warm-up and more for every byte of text
    // a lane's piece is one 64-byte group of a line (see tiles_walk); smaller pieces only where the batch is too small to give
    // every wave of the chip a tile otherwise -- and never so small that the warm-up is most of a lane's walk
    uint32_t w = 16u;
    while (w < 64u && uint64_t(w) * 2u * 64u * 16384u <= blob_bytes) w *= 2u;
    if (wb >= 1u && w < 32u) w = 32u;
    if (wb >= 2u) w = 64u;
    const uint64_t avg = std::max<uint64_t>(1, blob_bytes / n_haystacks);
    if (avg <= 64u) w = 64u;   // (whole haystacks per lane: nothing to warm up)
    uint64_t p = (avg + w - 1u) / w;
    if (p > 64u) p = 64u;
    if (p < 1u) p = 1u;
    *W = w;
    *P = uint32_t(p);
    return true;
}

uint32_t tiles_grid() { return 2u * uint32_t(num_cus()); }
uint64_t tiles_queue_entries(uint32_t W) { return uint64_t(tiles_grid()) * kTilesWaves * (64u * W / 4u); }
uint64_t tiles_status_words(uint64_t max_items, uint32_t P) { return max_items / (uint64_t(kTilesWaves) * (64u / P)) + 2u; }

hipError_t launch_walk_tiles(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode, uint32_t W, uint32_t P,
                             void *queue, uint32_t *ticket, uint64_t *status, void *out, uint64_t cap, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    TileArgs T{};
    T.W = W;
    T.P = P;
    T.R = 64u / P;
    T.back = A.max_pattern_length ? A.max_pattern_length - 1u : 0u;
    T.qcap = 64u * W / 4u;
    const uint32_t ring_bytes = kTilesWaves * kTilesRing * 8u;
    static const uint32_t debug_flags = [] { const char *e = std::getenv("ACMI_TILES_DEBUG"); return e ? uint32_t(std::atoi(e)) : 0u; }();
    T.flags = debug_flags;
    T.queue = static_cast<uint2 *>(queue);
    T.ticket = ticket;
    T.status = status;
    T.out = static_cast<uint4 *>(out);
    T.cap = cap;
    T.tile_bytes = 79u * 1024u - ring_bytes - 768u;   // two workgroups per CU share the 160 KiB
    const uint32_t lds = ring_bytes + T.tile_bytes + 16u + (class_mode == 0 ? 512u : 0u);
    // every state's row in the tile (small automata): the variant without the cold path
    const bool allhot = A.trap + 1u == A.n_states && A.row_bytes && uint64_t(A.trap) * (A.row_bytes + 4u) <= T.tile_bytes;
    const uint64_t groups = (max_items + uint64_t(kTilesWaves) * T.R - 1u) / (uint64_t(kTilesWaves) * T.R);
    const uint32_t blocks = uint32_t(std::min<uint64_t>(groups, tiles_grid()));
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kTilesThreads), lds, st, A, B, T);
        return hipGetLastError();
    };
#define ACMI_TILES(E, M) (allhot ? go(k_walk_tiles<E, M, true>) : go(k_walk_tiles<E, M, false>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_TILES(uint16_t, 1) : ACMI_TILES(uint16_t, 0);
    return class_mode == 1 ? ACMI_TILES(uint32_t, 1) : ACMI_TILES(uint32_t, 0);
#undef ACMI_TILES
}

}  // namespace acmi
