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
#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kTilesThreads = 1024;
constexpr uint32_t kTilesWaves = kTilesThreads / 64;

struct TileArgs {
    uint32_t W;           // bytes per lane piece: 16, 32 or 64
    uint32_t P;           // pieces (lanes) per item
    uint32_t R;           // items per wave = 64 / P
    uint32_t WB;          // warm-up blocks of 16 bytes = ceil((Lmax - 1) / 16)
    uint32_t qcap;        // queue entries per wave
    uint32_t tile_bytes;  // LDS budget of the table tile (+ record counts)
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

// One lane's piece: what it walks and what it reports.
struct Piece {
    const uint8_t *pb;    // 16-byte aligned address of the first block it walks
    uint32_t nblk;        // blocks it walks (0: nothing to do)
    uint32_t start_j;     // byte index (from pb) of the first byte that is walked: the state is the root before it
    uint32_t rep_lo, rep_len;   // bytes [rep_lo, rep_lo + rep_len) report
    uint32_t pos_bias;    // position of byte jj = pos_bias + jj
    uint32_t hay;
};

// One exact transition of a lane: the LDS tile's row when the state has one, else the global table / the compact records.
template <typename E, int MODE>
__device__ __forceinline__ uint32_t tiles_step(const DevAutomaton &A, const uint16_t *cls_lds, const uint8_t *tile, uint32_t tile_rows,
                                               uint32_t st, uint32_t byte) {
    const uint32_t coff = class_offset<E, MODE>(A, cls_lds, byte);
    uint32_t w = A.trap;
    if (st < tile_rows) w = *reinterpret_cast<const E *>(tile + __umul24(st, A.row_bytes) + coff);
    if (w >= A.trap) w = cold_step<E>(A, st, byte, coff);
    return w;
}

// The walk of one piece, counting: events go to the wave's queue, n_rec sums the records they stand for.  All 64 lanes of
// the wave call this together and run the same number of steps (a lane past its own bytes walks on without reporting).
template <typename E, int MODE>
__device__ __forceinline__ void tiles_walk(const DevAutomaton &A, const Piece &pc, const uint16_t *cls_lds, const uint8_t *tile,
                                           uint32_t tile_rows, const uint32_t *cnt_lds, uint32_t cnt_rows, uint2 *queue, uint32_t qcap,
                                           uint32_t lane, uint32_t &n_rec, uint32_t &qn) {
    uint32_t st = 0;
    auto event = [&](uint32_t jj) __attribute__((always_inline)) {
        const bool ev = jj - pc.rep_lo < pc.rep_len && is_final(A, st);
        // the lanes that have an event at this step: the ballot is taken by ALL lanes, so that the queue index stays wave-uniform
        const uint64_t m = __ballot(ev);
        if (m) {
            if (ev) {
                const uint32_t c = st < cnt_rows ? cnt_lds[st] : A.out_count[st];
                const uint32_t my = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                if (qn + my < qcap) queue[qn + my] = make_uint2(st | lane << 24, jj | n_rec << 9);
                n_rec += c;
            }
            qn += uint32_t(__builtin_popcountll(m));
        }
    };
    // A piece is its warm-up block(s) and at most 64 bytes = one group of a line: all of it is requested at once, back to back
    // (a line fetched in pieces over the microseconds of a walk comes from HBM again for every piece, k_walk_items' lesson); what
    // a longer warm-up or an unaligned last piece adds is requested as the buffers free up.
    const uint4 *q = reinterpret_cast<const uint4 *>(pc.pb);
    const uint32_t last = pc.nblk ? pc.nblk - 1u : 0u;
    const uint32_t blocks = tile_wave_max(pc.nblk);
    if (blocks == 0u) return;
    const uint4 first = q[0];
    uint4 b0 = q[1u < last ? 1u : last], b1 = q[2u < last ? 2u : last], b2 = q[3u < last ? 3u : last], b3 = q[4u < last ? 4u : last];
    // the first block: the walk may begin inside it (a haystack that starts unaligned) -- byte by byte, the state held in the
    // root up to the first byte that counts; not unrolled (a sixteenth of the steps at most)
#pragma unroll 1
    for (uint32_t jj = 0; jj < 16u; ++jj) {
        const uint32_t wd = jj < 8u ? (jj < 4u ? first.x : first.y) : (jj < 12u ? first.z : first.w);
        const uint32_t byte = (wd >> (8u * (jj & 3u))) & 0xffu;
        const uint32_t w = tiles_step<E, MODE>(A, cls_lds, tile, tile_rows, st, byte);
        st = jj < pc.start_j ? 0u : w;
        event(jj);
    }
    // the blocks behind it: sixteen unrolled steps per block, the buffers rotating
#pragma unroll 1
    for (uint32_t k = 1u; k < blocks; ++k) {
        const uint4 cur = b0;
        b0 = b1; b1 = b2; b2 = b3;
        const uint32_t kn = k + 4u;
        if (kn < blocks) b3 = q[kn < last ? kn : last];
        const uint32_t j0 = k * 16u;
#define ACMI_TSTEP(wd, sh, jj) { st = tiles_step<E, MODE>(A, cls_lds, tile, tile_rows, st, ((wd) >> (sh)) & 0xffu); event(jj); }
#define ACMI_TSTEP4(wd, jj) ACMI_TSTEP(wd, 0, jj) ACMI_TSTEP(wd, 8, (jj) + 1u) ACMI_TSTEP(wd, 16, (jj) + 2u) ACMI_TSTEP(wd, 24, (jj) + 3u)
        ACMI_TSTEP4(cur.x, j0) ACMI_TSTEP4(cur.y, j0 + 4u) ACMI_TSTEP4(cur.z, j0 + 8u) ACMI_TSTEP4(cur.w, j0 + 12u)
#undef ACMI_TSTEP4
#undef ACMI_TSTEP
    }
}

// The same walk writing records in place: for the waves whose events did not fit their queue (dense output, rare, and bound by
// its stores): every lane on its own, bytes straight from memory (the lines are in L2: the wave has just read them).
template <typename E, int MODE>
__device__ __forceinline__ void tiles_emit(const DevAutomaton &A, const Piece &pc, const uint16_t *cls_lds, const uint8_t *tile,
                                           uint32_t tile_rows, uint4 *out, uint64_t wr) {
    uint32_t st = 0;
    const uint32_t end = pc.rep_lo + pc.rep_len;
#pragma unroll 1
    for (uint32_t jj = pc.start_j; jj < end; ++jj) {
        st = tiles_step<E, MODE>(A, cls_lds, tile, tile_rows, st, load8g(pc.pb + jj));
        if (jj >= pc.rep_lo && is_final(A, st)) wr = emit_event(A, out, wr, pc.hay, pc.pos_bias + jj, st);
    }
}

template <typename E, int MODE>
__global__ __launch_bounds__(kTilesThreads, 8) void k_walk_tiles(DevAutomaton A, BatchArgs B, TileArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_tile, s_walked;
    __shared__ uint64_t s_before;
    __shared__ uint64_t s_wave_tot[kTilesWaves];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    // LDS: [512 B byte classes (TABLE mode)] [tile rows] [their record counts]
    const uint32_t cls_bytes = (MODE == 0) ? 512u : 0u;
    const uint32_t tile_rows = A.row_bytes ? (T.tile_bytes / (A.row_bytes + 4u) < A.trap ? T.tile_bytes / (A.row_bytes + 4u) : A.trap) : 0u;
    const uint16_t *cls_lds = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem) : nullptr;
    const uint8_t *tile = smem + cls_bytes;
    const uint32_t tile_bytes = (tile_rows * A.row_bytes + 15u) & ~15u;
    const uint32_t cnt_rows = tile_rows;
    const uint32_t *cnt_lds = reinterpret_cast<const uint32_t *>(tile + tile_bytes);
    {
        if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem)[threadIdx.x] = A.class_off[threadIdx.x];
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(A.lds_image) + A.tile_offset);
        uint4 *tdst = reinterpret_cast<uint4 *>(smem + cls_bytes);
        const uint32_t n16 = tile_bytes >> 4;
        for (uint32_t t = threadIdx.x; t < n16; t += kTilesThreads) tdst[t] = tsrc[t];
        uint32_t *cdst = reinterpret_cast<uint32_t *>(smem + cls_bytes + tile_bytes);
        for (uint32_t t = threadIdx.x; t < cnt_rows; t += kTilesThreads) cdst[t] = A.out_count[t];
    }
    if (threadIdx.x == 0) s_walked = 0;
    uint2 *queue = T.queue + (uint64_t(blockIdx.x) * kTilesWaves + wave) * T.qcap;
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
                int64_t xs = int64_t(a) - int64_t(T.WB) * 16;
                if (xs < hs) xs = hs;
                const int64_t xb = xs & ~int64_t(15);
                pc.pb = abase + G + xb;
                pc.start_j = uint32_t(xs - xb);
                pc.rep_lo = uint32_t(int64_t(lo) - xb);
                pc.rep_len = hi - lo;
                pc.nblk = (pc.rep_lo + pc.rep_len + 15u) >> 4;
                pc.pos_bias = uint32_t(xb) - d0 + it.pos0 + 1u;
                walked16 += pc.nblk;
            }
        }
        uint32_t n_rec = 0, qn = 0;
        tiles_walk<E, MODE>(A, pc, cls_lds, tile, tile_rows, cnt_lds, cnt_rows, queue, T.qcap, lane, n_rec, qn);
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
        if (wave_tot == 0u || base + wave_tot > T.cap) continue;   // (too small a buffer: the host grows it and runs the batch again)
        if (qn <= T.qcap) {
            // every lane expands events, 64 per round: an event's slot = the records of the lanes before its owner + its own offset
#pragma unroll 1
            for (uint32_t e0 = 0; e0 < qn; e0 += 64u) {
                const uint32_t e = e0 + lane;
                const bool valid = e < qn;
                const uint2 ent = valid ? queue[e] : make_uint2(0u, 0u);
                const uint32_t owner = ent.x >> 24, s = ent.x & 0xFFFFFFu;
                const uint32_t slot = uint32_t(__shfl(int(excl), int(owner), 64)) + (ent.y >> 9);
                const uint32_t pos = uint32_t(__shfl(int(pc.pos_bias), int(owner), 64)) + (ent.y & 511u);
                const uint32_t hay = uint32_t(__shfl(int(pc.hay), int(owner), 64));
                if (!valid) continue;
                uint4 *dst = T.out + base + slot;
                if (A.out_list != nullptr) {
                    const uint32_t c = s < cnt_rows ? cnt_lds[s] : A.out_count[s];
                    const uint2 *ol = A.out_list + A.out_off[s];
                    for (uint32_t k = 0; k < c; ++k) {
                        const uint2 o = ol[k];
                        dst[k] = make_uint4(hay, pos, pos - o.y, o.x);
                    }
                } else {
                    (void)emit_event(A, dst, 0, hay, pos, s);
                }
            }
        } else {
            // more events than the queue holds (dense output): every lane walks its piece again and writes its records in place
            if (pc.nblk) tiles_emit<E, MODE>(A, pc, cls_lds, tile, tile_rows, T.out, base + excl);
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
    T.WB = A.max_pattern_length ? (A.max_pattern_length - 1u + 15u) / 16u : 0u;
    T.qcap = 64u * W / 4u;
    T.tile_bytes = 78u * 1024u;
    T.queue = static_cast<uint2 *>(queue);
    T.ticket = ticket;
    T.status = status;
    T.out = static_cast<uint4 *>(out);
    T.cap = cap;
    const uint32_t lds = T.tile_bytes + 16u + (class_mode == 0 ? 512u : 0u);
    const uint64_t groups = (max_items + uint64_t(kTilesWaves) * T.R - 1u) / (uint64_t(kTilesWaves) * T.R);
    const uint32_t blocks = uint32_t(std::min<uint64_t>(groups, tiles_grid()));
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kTilesThreads), lds, st, A, B, T);
        return hipGetLastError();
    };
    if (entry_bytes == 2) return class_mode == 1 ? go(k_walk_tiles<uint16_t, 1>) : go(k_walk_tiles<uint16_t, 0>);
    return class_mode == 1 ? go(k_walk_tiles<uint32_t, 1>) : go(k_walk_tiles<uint32_t, 0>);
}

}  // namespace acmi
