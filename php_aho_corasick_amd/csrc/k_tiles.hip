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
#include <type_traits>

#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kTilesThreads = 1024;
constexpr uint32_t kTilesWaves = kTilesThreads / 64;
constexpr uint32_t kTilesRing = 128;   // events a wave stages in LDS (flushed to its queue 64 at a time)
constexpr uint32_t kTilesFinals = 8192;  // !ALLHOT: up to this many final states keep their record counts in LDS, a byte each
constexpr uint32_t kTilesLaneCap = 16; // LANEQ: events a LANE keeps in LDS (more: the wave walks its pieces again)

struct TileArgs {
    uint32_t W;           // bytes per lane piece: 16, 32 or 64
    uint32_t P;           // pieces (lanes) per item
    uint32_t R;           // items per wave = 64 / P
    uint32_t back;        // bytes a lane enters its piece early = Lmax - 1
    uint32_t qcap;        // queue entries per wave (a multiple of 64)
    uint32_t tile_bytes;  // LDS budget of the table tile (+ record counts)
    uint32_t flags;       // reserved (0)
    uint2 *queue;         // grid waves x qcap entries
    uint32_t *ticket;     // zeroed
    uint64_t *status;     // zeroed, one word per workgroup tile
    uint4 *out;
    uint64_t cap;         // records `out` holds
    // split pattern sets (engine.cpp): the match arrays of the OTHER part of the set over the same batch -- all its patterns longer
    // than all of this automaton's -- are merged into `out` as this kernel writes it (LANEQ variants): per haystack position
    // ascending, the other part's records first at equal positions (node.c:424-441: longest first).  Null: nothing to merge.
    const uint64_t *m_prefix;
    const uint4 *m_records;
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
    uint64_t m_first;     // merging: first record of the other part that ends inside this piece ...
    uint32_t m_count;     // ... and how many do
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
    uint32_t *lists;      // LANEQ: this wave's lane lists (LDS): entry e of lane l at lists[e * 64 + l] = row offset | byte index << 16
    uint32_t n_ev;        // LANEQ: this lane's events so far
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
// ALLHOT (every state has its row in the tile: small automata): no test, no cold path compiled in, and a state is held as the
// BYTE OFFSET of its row -- the tile's entries are stored that way (hot_stage) -- so that a step is one add and one read.
// D32 (dense rows, the whole table below 4 GiB -- everything but compact automata): the row that is not in the tile is one load
// at a 32-bit offset from the table's base (a multiply-add and the load; the general form went through a 64-bit multiply and a
// test of the automaton's form, and carried the compact form's record loop in every step of the unrolled walk).
// The tile has one row more than it holds states: row `tile_rows` = the trap's (every entry "not here"), which is where a state
// without a row looks -- a minimum instead of a test and an exec mask around the read.
template <typename E, bool ALLHOT, bool D32 = false>
__device__ __forceinline__ uint32_t tiles_step(const DevAutomaton &A, const uint8_t *tile, uint32_t tile_rows, uint32_t st, uint32_t byte,
                                               uint32_t coff) {
    if constexpr (ALLHOT) return *reinterpret_cast<const E *>(tile + st + coff);
    const uint32_t row = st < tile_rows ? st : tile_rows;
    uint32_t w = *reinterpret_cast<const E *>(tile + __umul24(row, A.row_bytes) + coff);
    if (w >= A.trap) {
        if constexpr (D32) w = *reinterpret_cast<const E *>(static_cast<const uint8_t *>(A.table) + (__umul24(st, A.row_bytes) + coff));
        else w = cold_step<E>(A, st, byte, coff);
    }
    return w;
}

// What the walk needs besides the automaton: where the tile's data lies in LDS
struct TileLds {
    const uint16_t *cls;      // byte classes (TABLE mode), else null
    const uint8_t *tile;
    uint32_t rows;            // rows of the tile
    uint32_t stride;          // ALLHOT: bytes from row to row = row_bytes + 2 x sizeof(E): {record count, state id} behind the transitions
    uint32_t final_from;      // ALLHOT: f1 x stride -- a state (row offset) is final from here on
    const uint32_t *cnt;      // !ALLHOT, many final states: record counts of the tile's rows
    const uint8_t *fcnt;      // !ALLHOT, <= kTilesFinals final states: min(records, 255) of EVERY final state, hot ones first (else null)
    uint32_t hot_finals;      // trap - f1
    const uint2 *out1;        // ALLHOT, when there is room: the first {pattern, length} of every state's output list, else null
};

template <typename E, bool ALLHOT>
__device__ __forceinline__ bool tiles_final(const DevAutomaton &A, const TileLds &L, uint32_t s) {
    if constexpr (ALLHOT) return s >= L.final_from;   // (no cold states, and the trap row is never entered)
    return is_final(A, s);
}

// {state id, record count} of a final state the walk is in
template <typename E, bool ALLHOT>
__device__ __forceinline__ uint2 tiles_meta(const DevAutomaton &A, const TileLds &L, uint32_t s) {
    if constexpr (ALLHOT) {
        const E *m = reinterpret_cast<const E *>(L.tile + s + A.row_bytes);
        return make_uint2(uint32_t(m[1]), uint32_t(m[0]));
    }
    if (L.fcnt != nullptr) {   // (what the walk asks at every event: a global read here was a second L2 round trip in the step)
        const uint32_t c = L.fcnt[s >= A.f2 ? s - A.f2 + L.hot_finals : s - A.f1];
        if (c < 255u) return make_uint2(s, c);
    } else if (s < L.rows) return make_uint2(s, L.cnt[s]);
    if (A.out_head != nullptr) {   // (the line the expansion reads again)
        const uint32_t c = A.out_head[s].x >> 24;
        if (c < 255u) return make_uint2(s, c);
    }
    return make_uint2(s, A.out_count[s]);
}

// The walk of one piece, counting: events go to the wave's queue, n_rec sums the records they stand for.  All 64 lanes of
// the wave call this together and run the same number of steps (a lane past its own bytes walks on without reporting).
// LANEQ (small automata, whose tile leaves the LDS free): events go to a list of the LANE's own -- a shift, an or, a store -- and
// are counted and expanded by the lane afterwards.  With the wave-shared queue an event costs ~ 13 operations (ballot, rank,
// record count, packing, ring address) which the whole wave issues at every step in which ANY of its lanes has one: at a match
// per 15 bytes that is 99 % of the steps, and it was more than half of the kernel's instructions.
template <typename E, int MODE, bool ALLHOT, bool LANEQ, bool D32>
__device__ __forceinline__ void tiles_walk(const DevAutomaton &A, const Piece &pc, const TileLds &L, TileQueue &Q, uint32_t lane,
                                           uint32_t &n_rec, bool events) {
    uint32_t st = 0;
    // `inside`: the byte reports (it lies in the lane's own range); the lanes that have an event at this step: the ballot is
    // taken by ALL lanes, so that the queue index stays wave-uniform
    auto event = [&](uint32_t jj, bool inside) __attribute__((always_inline)) {
        const bool ev = inside && tiles_final<E, ALLHOT>(A, L, st);
        if constexpr (LANEQ) {
            if (ev) {
                // (a list that overflows is not used -- the lane walks its piece again, tiles_emit --: its last entry may as well
                // be overwritten, and the store needs no test of its own)
                // (round 6: the list's LDS address bumped in place instead of the index -- three operations for seven, the loop 380 -> 351
                // instructions -- measured SLOWER, A/B in one run: 8 patterns of 2-6 B at 540 records per 8 KiB 1.32 -> 1.42 ms per GiB)
                const uint32_t at = Q.n_ev < kTilesLaneCap - 1u ? Q.n_ev : kTilesLaneCap - 1u;
                Q.lists[at * 64u + lane] = st + uint32_t(__builtin_amdgcn_readfirstlane(int(jj << 16)));   // (jj is wave-uniform: one scalar, one add)
                ++Q.n_ev;
            }
            return;
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(ev);
        if (m) {
            if (ev) {
                const uint2 meta = tiles_meta<E, ALLHOT>(A, L, st);
                const uint32_t my = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                if (events) Q.ring[(Q.total + my) & (kTilesRing - 1u)] = make_uint2(meta.x | lane << 24, jj | n_rec << 9);
                n_rec += meta.y;
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
        const uint32_t w = tiles_step<E, ALLHOT, D32>(A, L.tile, L.rows, st, byte, class_offset<E, MODE>(A, L.cls, byte));
        st = jj < pc.start_j ? 0u : w;
        event(jj, jj - pc.rep_lo < pc.rep_len);
    }
    // the blocks behind it: sixteen unrolled steps per block, the buffers rotating; the column offsets of a dword's four bytes
    // are worked out together (SWAR3: a handful of operations for all four; TABLE: four independent LDS reads).  A block that
    // lies inside EVERY lane's own range -- all but the first and the last of a piece -- takes the body without the range test.
#pragma unroll 1
    for (uint32_t k = 1u; k < blocks; ++k) {
        asm volatile("; acmi-hot-begin");   // (tools/check_asm.py: no scratch traffic between the markers)
        const uint4 cur = b0;
        b0 = b1; b1 = b2; b2 = b3;
        const uint32_t kn = k + 4u;
        if (kn < blocks) b3 = q[kn < last ? kn : last];
        const uint32_t j0 = k * 16u;
        auto dword = [&](uint32_t wd, uint32_t jj, auto checked) __attribute__((always_inline)) {
            uint32_t c0, c1, c2, c3;
            if constexpr (MODE == 1) {
                const uint32_t c4 = swar_columns<E>(A, wd);
                c0 = c4 & 0xffu; c1 = (c4 >> 8) & 0xffu; c2 = (c4 >> 16) & 0xffu; c3 = c4 >> 24;
            } else {
                c0 = class_offset<E, MODE>(A, L.cls, wd & 0xffu); c1 = class_offset<E, MODE>(A, L.cls, (wd >> 8) & 0xffu);
                c2 = class_offset<E, MODE>(A, L.cls, (wd >> 16) & 0xffu); c3 = class_offset<E, MODE>(A, L.cls, wd >> 24);
            }
            constexpr bool C = decltype(checked)::value;
            st = tiles_step<E, ALLHOT, D32>(A, L.tile, L.rows, st, wd & 0xffu, c0); event(jj, !C || jj - pc.rep_lo < pc.rep_len);
            st = tiles_step<E, ALLHOT, D32>(A, L.tile, L.rows, st, (wd >> 8) & 0xffu, c1); event(jj + 1u, !C || jj + 1u - pc.rep_lo < pc.rep_len);
            st = tiles_step<E, ALLHOT, D32>(A, L.tile, L.rows, st, (wd >> 16) & 0xffu, c2); event(jj + 2u, !C || jj + 2u - pc.rep_lo < pc.rep_len);
            st = tiles_step<E, ALLHOT, D32>(A, L.tile, L.rows, st, wd >> 24, c3); event(jj + 3u, !C || jj + 3u - pc.rep_lo < pc.rep_len);
        };
        const bool full = j0 >= pc.rep_lo && j0 + 16u <= pc.rep_lo + pc.rep_len;
        if (__all(full)) {
            dword(cur.x, j0, std::false_type{}); dword(cur.y, j0 + 4u, std::false_type{}); dword(cur.z, j0 + 8u, std::false_type{}); dword(cur.w, j0 + 12u, std::false_type{});
        } else {
            dword(cur.x, j0, std::true_type{}); dword(cur.y, j0 + 4u, std::true_type{}); dword(cur.z, j0 + 8u, std::true_type{}); dword(cur.w, j0 + 12u, std::true_type{});
        }
        asm volatile("; acmi-hot-end");
    }
}

// The records of a piece, counted by a plain walk (LANEQ: a lane of the wave had more events than its list holds).
template <typename E, int MODE, bool ALLHOT>
__device__ __forceinline__ uint32_t tiles_count(const DevAutomaton &A, const Piece &pc, const TileLds &L) {
    uint32_t st = 0, n = 0;
    const uint32_t end = pc.rep_lo + pc.rep_len;
#pragma unroll 1
    for (uint32_t jj = pc.start_j; jj < end; ++jj) {
        const uint32_t byte = load8g(pc.pb + jj);
        st = tiles_step<E, ALLHOT>(A, L.tile, L.rows, st, byte, class_offset<E, MODE>(A, L.cls, byte));
        if (jj >= pc.rep_lo && tiles_final<E, ALLHOT>(A, L, st)) n += tiles_meta<E, ALLHOT>(A, L, st).y;
    }
    return n;
}

// The same walk writing records in place: for the waves whose events did not fit their queue (dense output, rare, and bound by
// its stores): every lane on its own, bytes straight from memory (the lines are in L2: the wave has just read them).
template <typename E, int MODE, bool ALLHOT>
__device__ __forceinline__ void tiles_emit(const DevAutomaton &A, const Piece &pc, const TileLds &L, const uint4 *m_records, uint4 *out, uint64_t wr,
                                           uint64_t limit) {
    uint32_t st = 0, i0 = 0;
    uint32_t next0 = pc.m_count ? m_records[pc.m_first].y : 0xFFFFFFFFu;   // position of the other part's next record
    const uint32_t end = pc.rep_lo + pc.rep_len;
#pragma unroll 1
    for (uint32_t jj = pc.start_j; jj < end; ++jj) {
        const uint32_t byte = load8g(pc.pb + jj);
        st = tiles_step<E, ALLHOT>(A, L.tile, L.rows, st, byte, class_offset<E, MODE>(A, L.cls, byte));
        if (jj < pc.rep_lo) continue;
        while (next0 <= pc.pos_bias + jj) {   // (the other part's patterns are the longer ones: first at equal positions)
            if (wr < limit) out[wr] = m_records[pc.m_first + i0];
            ++wr;
            ++i0;
            next0 = i0 < pc.m_count ? m_records[pc.m_first + i0].y : 0xFFFFFFFFu;
        }
        if (tiles_final<E, ALLHOT>(A, L, st)) {
            const uint2 meta = tiles_meta<E, ALLHOT>(A, L, st);
            if (wr + meta.y <= limit) (void)emit_event(A, out, wr, pc.hay, pc.pos_bias + jj, meta.x);   // (records past the buffer's end are not written: the host grows it, or hands out what fits)
            wr += meta.y;
        }
    }
}

// An event's records, longest pattern first, at `dst`.  The first entry of a small automaton's lists is in LDS (out1): most
// events stand for one record, and the two dependent loads per event were most of the expansion.
__device__ __forceinline__ void tiles_expand(const DevAutomaton &A, const TileLds &L, uint4 *dst, uint32_t hay, uint32_t pos, uint32_t s, uint32_t c) {
    if (A.out_list != nullptr) {
        uint32_t k = 0;
        if (L.out1 != nullptr) {
            const uint2 o = L.out1[s];
            dst[0] = make_uint4(hay, pos, pos - o.y, o.x);
            if (c == 1u) return;
            k = 1;
        }
        if (k == 0 && A.out_head != nullptr) {   // (states without a row in the tile: the head of the list in one read)
            const uint2 h = A.out_head[s];
            dst[0] = make_uint4(hay, pos, pos - h.y, h.x & 0xFFFFFFu);
            if (c == 1u) return;
            k = 1;
        }
        const uint2 *ol = A.out_list + A.out_off[s];
        for (; k < c; ++k) {
            const uint2 o = ol[k];
            dst[k] = make_uint4(hay, pos, pos - o.y, o.x);
        }
    } else {
        (void)emit_event(A, dst, 0, hay, pos, s);
    }
}

// Two workgroups per CU (64 registers, 8 waves per SIMD): what hides the round trips of rows that are not in LDS -- and the
// dependent LDS reads of every step -- is the number of waves, not the size of the tile or a spill-free register file: one
// workgroup with the whole LDS for its tile (twice the rows, 80-odd registers, no spills) measured 8-12 % slower on a word list
// and on the headline set with 64 plants per haystack, three workgroups of 512 threads (6 waves per SIMD, 80 registers, no
// spills in any loop) 8-15 % slower on every workload tried.  (The dozen registers spilled at 64 are touched once per tile.)
template <typename E, int MODE, bool ALLHOT, bool LANEQ, bool D32>
__global__ __launch_bounds__(kTilesThreads, 8) void k_walk_tiles(DevAutomaton A, BatchArgs B, TileArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_tile, s_walked;
    __shared__ uint64_t s_before;
    __shared__ uint64_t s_wave_tot[kTilesWaves];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    // LDS: [the waves' event rings] [512 B byte classes (TABLE mode)] [tile rows] [record counts | first output entries]
    const uint32_t ring_bytes = LANEQ ? kTilesWaves * kTilesLaneCap * 256u : kTilesWaves * kTilesRing * 8u;
    const uint32_t cls_bytes = (MODE == 0) ? 512u : 0u;
    TileLds L;
    L.cls = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem + ring_bytes) : nullptr;
    L.tile = smem + ring_bytes + cls_bytes;
    L.out1 = nullptr;
    L.cnt = nullptr;
    L.fcnt = nullptr;
    L.hot_finals = A.trap - A.f1;
    if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem + ring_bytes)[threadIdx.x] = A.class_off[threadIdx.x];
    const uint8_t *src = static_cast<const uint8_t *>(A.lds_image) + A.tile_offset;
    if constexpr (ALLHOT) {
        // every state: its transitions as ROW OFFSETS, then {record count, state id}
        L.rows = A.trap;
        L.stride = A.row_bytes + 2u * uint32_t(sizeof(E));
        L.final_from = A.f1 * L.stride;
        const uint32_t per_row = A.n_columns + 2u;
        uint8_t *dst = smem + ring_bytes + cls_bytes;
        for (uint32_t t = threadIdx.x; t < L.rows * per_row; t += kTilesThreads) {
            const uint32_t row = t / per_row, c = t - row * per_row;
            uint32_t v;
            if (c < A.n_columns) v = uint32_t(reinterpret_cast<const E *>(src + row * A.row_bytes)[c]) * L.stride;
            else v = c == A.n_columns ? A.out_count[row] : row;
            reinterpret_cast<E *>(dst + row * L.stride)[c] = E(v);
        }
        const uint32_t used = (L.rows * L.stride + 7u) & ~7u;
        if (A.out_list != nullptr && used + L.rows * 8u <= T.tile_bytes) {
            uint2 *o1 = reinterpret_cast<uint2 *>(dst + used);
            for (uint32_t t = threadIdx.x; t < L.rows; t += kTilesThreads) o1[t] = A.out_count[t] ? A.out_list[A.out_off[t]] : make_uint2(0u, 0u);
            L.out1 = o1;
        }
    } else {
        // rows [0, L.rows) and, behind them, the row of "not here" entries (tiles_step); the image's own row `trap` is one.
        // Record counts: of every final state, a byte each, when there are few enough (a pattern set of a few thousand: all of
        // them) -- the tile's first rows are the shallow states, final states among them are rare, and the count of a final state
        // OUTSIDE the tile was a global read at every event; else of the tile's rows, a word each, as before.
        const uint32_t n_final = (A.trap - A.f1) + (A.n_states - A.f2);
        const bool bytes = n_final <= kTilesFinals;
        const uint32_t fbytes = bytes ? (n_final + 15u) & ~15u : 0u;
        const uint32_t per_row = A.row_bytes + (bytes ? 0u : 4u);
        const uint32_t fit = A.row_bytes && T.tile_bytes > A.row_bytes + 16u + fbytes ? (T.tile_bytes - A.row_bytes - 16u - fbytes) / per_row : 0u;
        L.rows = fit < A.trap ? fit : A.trap;
        L.stride = A.row_bytes;
        L.final_from = 0;
        const uint32_t tile_bytes = ((L.rows + 1u) * A.row_bytes + 15u) & ~15u;
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + ring_bytes + cls_bytes);
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(src);
        for (uint32_t t = threadIdx.x; t < (tile_bytes >> 4); t += kTilesThreads) tdst[t] = tsrc[t];
        if (L.rows < A.trap) {
            __syncthreads();   // (the copy above wrote whole 16-byte blocks across this row)
            E *trap_row = reinterpret_cast<E *>(smem + ring_bytes + cls_bytes + L.rows * A.row_bytes);
            for (uint32_t t = threadIdx.x; t < A.n_columns; t += kTilesThreads) trap_row[t] = E(A.trap);
        }
        if (bytes) {
            uint8_t *fdst = smem + ring_bytes + cls_bytes + tile_bytes;
            const uint32_t hot_finals = A.trap - A.f1;
            for (uint32_t t = threadIdx.x; t < n_final; t += kTilesThreads) {
                const uint32_t c = A.out_count[t < hot_finals ? A.f1 + t : A.f2 + (t - hot_finals)];
                fdst[t] = uint8_t(c < 255u ? c : 255u);
            }
            L.fcnt = fdst;
        } else {
            uint32_t *cdst = reinterpret_cast<uint32_t *>(smem + ring_bytes + cls_bytes + tile_bytes);
            for (uint32_t t = threadIdx.x; t < L.rows; t += kTilesThreads) cdst[t] = A.out_count[t];
            L.cnt = cdst;
        }
    }
    if (threadIdx.x == 0) s_walked = 0;
    TileQueue Q;
    Q.ring = reinterpret_cast<uint2 *>(smem) + wave * kTilesRing;
    Q.lists = reinterpret_cast<uint32_t *>(smem) + wave * kTilesLaneCap * 64u;
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
        pc.m_first = 0;
        pc.m_count = 0;
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
        Q.total = Q.flushed = Q.n_ev = 0;
        tiles_walk<E, MODE, ALLHOT, LANEQ, D32>(A, pc, L, Q, lane, n_rec, true);
        if constexpr (LANEQ) {
            if (T.m_prefix != nullptr && pc.nblk) {
                // the other part's records that END inside this piece: positions [p_lo, p_hi] of this haystack (looked up behind
                // the walk: three registers less to keep through it)
                const uint64_t s0 = T.m_prefix[pc.hay], s1 = T.m_prefix[pc.hay + 1u];
                const uint32_t p_lo = pc.pos_bias + pc.rep_lo, p_hi = p_lo + pc.rep_len - 1u;
                uint64_t a = s0, b = s1;
                while (a < b) { const uint64_t mid = (a + b) >> 1; if (T.m_records[mid].y < p_lo) a = mid + 1; else b = mid; }
                pc.m_first = a;
                b = s1;
                while (a < b) { const uint64_t mid = (a + b) >> 1; if (T.m_records[mid].y <= p_hi) a = mid + 1; else b = mid; }
                pc.m_count = uint32_t(a - pc.m_first);
            }
        }
        bool lane_over = false;   // LANEQ: some lane of the wave had more events than its list holds
        uint32_t ev_max = 0;
        if constexpr (LANEQ) {
            ev_max = tile_wave_max(Q.n_ev);
            lane_over = ev_max > kTilesLaneCap;
            if (!lane_over) {
#pragma unroll 1
                for (uint32_t e = 0; e < ev_max; ++e)
                    if (e < Q.n_ev) n_rec += tiles_meta<E, ALLHOT>(A, L, Q.lists[e * 64u + lane] & 0xffffu).y;
            } else if (pc.nblk) {
                n_rec = tiles_count<E, MODE, ALLHOT>(A, pc, L);
            }
        }
        // (merging: the running sum places the other part's records too; the item totals k_expand turns into this part's own
        // per-haystack prefix stay this part's)
        const uint32_t own_rec = n_rec;
        n_rec += pc.m_count;
        // records: exclusive prefix over the wave's lanes, the items' totals, the wave's total
        const uint32_t incl = wave_inclusive_sum(n_rec);
        const uint32_t excl = incl - n_rec;
        const uint32_t wave_tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        {
            const uint32_t last_lane = r * T.P + T.P - 1u;
            uint32_t i_incl = incl, i_excl = excl;
            if (LANEQ && T.m_prefix != nullptr) { i_incl = wave_inclusive_sum(own_rec); i_excl = i_incl - own_rec; }
            const uint32_t item_tot = uint32_t(__shfl(int(i_incl), int(last_lane < 64u ? last_lane : 63u), 64)) - uint32_t(__shfl(int(i_excl), int(r * T.P < 64u ? r * T.P : 63u), 64));
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
        // (too small a buffer: what fits is written -- whole events, in order -- and the host grows it and runs the batch again, or
        // hands out the haystacks that are complete: "max_result_bytes")
        if (wave_tot == 0u || base >= T.cap) continue;
        const uint64_t room = T.cap - base;   // records this wave may write
        if constexpr (LANEQ) {
            if (!lane_over) {
                // every lane its own events, in the order it met them: its records are one contiguous piece of the output
                uint4 *dst = T.out + base;
                uint64_t at = excl;   // this lane's next record, counted from the wave's base
                uint32_t i0 = 0;
                uint32_t next0 = pc.m_count ? T.m_records[pc.m_first].y : 0xFFFFFFFFu;   // position of the other part's next record
#pragma unroll 1
                for (uint32_t e = 0; e < ev_max; ++e) {
                    if (e >= Q.n_ev) continue;
                    const uint32_t ent = Q.lists[e * 64u + lane];
                    const uint2 meta = tiles_meta<E, ALLHOT>(A, L, ent & 0xffffu);
                    const uint32_t pos = pc.pos_bias + (ent >> 16);
                    while (next0 <= pos) {   // (rare: the other part's patterns are the long ones)
                        if (at < room) dst[at] = T.m_records[pc.m_first + i0];
                        ++at;
                        ++i0;
                        next0 = i0 < pc.m_count ? T.m_records[pc.m_first + i0].y : 0xFFFFFFFFu;
                    }
                    if (at + meta.y <= room) tiles_expand(A, L, dst + at, pc.hay, pos, meta.x, meta.y);
                    at += meta.y;
                }
                for (; i0 < pc.m_count; ++i0, ++at)
                    if (at < room) dst[at] = T.m_records[pc.m_first + i0];
            } else if (pc.nblk) {
                tiles_emit<E, MODE, ALLHOT>(A, pc, L, T.m_records, T.out, base + excl, T.cap);
            }
            continue;
        }
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
                if (valid) {
                    uint32_t c;
                    if constexpr (ALLHOT) c = uint32_t(reinterpret_cast<const E *>(L.tile + s * L.stride + A.row_bytes)[0]);
                    else c = tiles_meta<E, ALLHOT>(A, L, s).y;
                    if (uint64_t(slot) + c <= room) tiles_expand(A, L, T.out + base + slot, hay, pos, s, c);
                }
            }
        } else if (pc.nblk) {
            // more events than the queue holds (dense output): every lane walks its piece again and writes its records in place
            tiles_emit<E, MODE, ALLHOT>(A, pc, L, nullptr, T.out, base + excl, T.cap);
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

// Can this automaton's walk merge another part's match arrays as it writes (the lane-list variant)?
bool tiles_can_merge(const DevAutomaton &A, int entry_bytes) {
    const uint32_t share = 79u * 1024u - 768u, ring_l = kTilesWaves * kTilesLaneCap * 256u;
    const uint64_t hot_stride = uint64_t(A.row_bytes) + 2u * uint32_t(entry_bytes);
    return A.trap + 1u == A.n_states && A.row_bytes && A.max_out_count < 65536u && uint64_t(A.trap + 1u) * hot_stride < 65536u &&
           uint64_t(A.trap) * hot_stride + 8u <= share - ring_l;
}

hipError_t launch_walk_tiles(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode, uint32_t W, uint32_t P,
                             void *queue, uint32_t *ticket, uint64_t *status, void *out, uint64_t cap, hipStream_t st,
                             const uint64_t *merge_prefix, const void *merge_records) {
    if (max_items == 0) return hipSuccess;
    TileArgs T{};
    T.W = W;
    T.P = P;
    T.R = 64u / P;
    T.back = A.max_pattern_length ? A.max_pattern_length - 1u : 0u;
    T.qcap = 64u * W / 4u;
    // ACMI_TILES_LISTS=0 (measurements): the shared ring instead of the lanes' own lists where both are possible -- same records
    static const bool no_lists = [] { const char *e = std::getenv("ACMI_TILES_LISTS"); return e && e[0] == '0'; }();
    T.flags = 0;
    T.queue = static_cast<uint2 *>(queue);
    T.ticket = ticket;
    T.status = status;
    T.out = static_cast<uint4 *>(out);
    T.cap = cap;
    T.m_prefix = merge_prefix;
    T.m_records = static_cast<const uint4 *>(merge_records);
    // Two workgroups per CU share the 160 KiB.  Every state's row in the tile (small automata): the variant without the cold
    // path, states held as row offsets (which must fit the table's entry type); and when the tile is small enough to leave 64 KiB
    // of a workgroup's LDS free, the lanes keep their events in lists of their own there (LANEQ)
    const uint32_t share = 79u * 1024u - 768u;
    const uint64_t hot_stride = uint64_t(A.row_bytes) + 2u * uint32_t(entry_bytes);
    const uint64_t hot_bytes = uint64_t(A.trap) * hot_stride + 8u;
    const uint32_t ring_q = kTilesWaves * kTilesRing * 8u, ring_l = kTilesWaves * kTilesLaneCap * 256u;
    const bool hot_ok = A.trap + 1u == A.n_states && A.row_bytes && (entry_bytes == 4 || uint64_t(A.trap + 1u) * hot_stride < 65536u) && A.max_out_count < 65536u;
    const bool laneq = hot_ok && hot_bytes <= share - ring_l && uint64_t(A.trap + 1u) * hot_stride < 65536u && (!no_lists || merge_prefix);
    if (merge_prefix && !laneq) return hipErrorInvalidValue;   // (the caller asks tiles_can_merge first)
    const bool allhot = laneq || (hot_ok && hot_bytes <= share - ring_q);
    const uint32_t ring_bytes = laneq ? ring_l : ring_q;
    T.tile_bytes = share - ring_bytes;
    const uint32_t lds = ring_bytes + T.tile_bytes + 16u + (class_mode == 0 ? 512u : 0u);
    const uint64_t groups = (max_items + uint64_t(kTilesWaves) * T.R - 1u) / (uint64_t(kTilesWaves) * T.R);
    const uint32_t blocks = uint32_t(std::min<uint64_t>(groups, tiles_grid() - 2u * std::min<uint32_t>(B.reserve_cus, tiles_grid() / 2u - 1u)));
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kTilesThreads), lds, st, A, B, T);
        return hipGetLastError();
    };
    // (the rows outside the tile at 32-bit offsets: dense rows, the table below 4 GiB)
    const bool d32 = !A.compact && uint64_t(A.n_states) * A.row_bytes < (1ull << 32);
#define ACMI_TILES(E, M) (laneq ? go(k_walk_tiles<E, M, true, true, false>) : allhot ? go(k_walk_tiles<E, M, true, false, false>) : \
                          d32 ? go(k_walk_tiles<E, M, false, false, true>) : go(k_walk_tiles<E, M, false, false, false>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_TILES(uint16_t, 1) : ACMI_TILES(uint16_t, 0);
    return class_mode == 1 ? ACMI_TILES(uint32_t, 1) : ACMI_TILES(uint32_t, 0);
#undef ACMI_TILES
}

}  // namespace acmi
