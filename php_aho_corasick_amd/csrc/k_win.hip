// k_win.hip -- k_walk_win: the exact walk of whole work items (kernel 4) for automata far too big for the LDS, over the
// two-byte WINDOW TABLE (automaton.hpp win_table): trie depth <= 2 decided in LDS from the text alone.
#include "kdev.hpp"

#include <type_traits>

namespace acmi {

// ------------------------------------------------------------------------------------------
// A row of a 256-column table is 1 KiB: the CU's LDS holds 159 of them, not even the 257 states of trie depth <= 1 of a
// big set over all byte values, and the table walk (k_walk_items, k_walk_tiles) of such a set is a request to the L2 per
// byte -- 55-90 GB/s for the 65,536 patterns of config 3 whatever the tile size (round 5's flat LDS sweep).  But where no
// pattern is shorter than three bytes no state of depth <= 2 is final, and the state after a byte -- while it is that
// shallow -- is a function of the LAST TWO BYTES alone (the longest suffix of the text that is a trie path, ahocorasick.c:199-212
// followed to its end): the walk needs no state there at all, only the answer to "does byte c leave depth 2 DOWNWARD from
// the node of (a, b)", and that is what the table's 65,536 16-bit cells say, indexed by a << 8 | b (automaton.hpp win_table):
// one child exactly, two by 7-bit fingerprints, three or more by their list behind the cells.  A lane takes sixteen bytes
// at a time: sixteen independent LDS reads and tests, no state, no dependence between consecutive bytes -- and only where a
// test says "maybe" (0.5 % of the bytes of random text over config 3's set, most of them real) anything else happens:
//   * the node of depth 3 is looked up by its three bytes in win_hash -- ONE probe in global memory;
//   * from there the walk is the usual one (goto edge or failure link of the compact records, a dense row where the state
//     has one), events and all, while the state stays deeper than 2; a failure link that leads to depth <= 2 hands the
//     byte back to the table (delta(fail(s), c) is the table's answer for the same byte).
// The lanes of a wave take these rare steps TOGETHER, each at its own next position: a loop over the sixteen positions with
// the slow step behind a branch made every block of every wave pay a chain of global round trips per position (1.0 ms per
// 64 MiB, no better than the table walk's 1.1).
// Same events, same slab, same drains as k_walk_items (whose drain form also drains this kernel's overflowed items).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kWinCellBytes = 65536u * 2u;
constexpr uint32_t kWinLdsBytes = kWinCellBytes + kWinListBytesDev;

__device__ __forceinline__ uint32_t win_fp7_dev(uint32_t c) { return (c ^ (c >> 3) ^ (c >> 5)) & 127u; }
// the cell's quick answer: exact for none / one child, fingerprints for two, "maybe" for a list
__device__ __forceinline__ bool win_maybe(uint32_t cell, uint32_t c) {
    const uint32_t kind = cell >> 14;
    const uint32_t f = win_fp7_dev(c);
    const bool one = (cell & 0xC0FFu) == (0x8000u | c);
    const bool two = kind == 1u && (f == (cell & 127u) || f == ((cell >> 7) & 127u));
    return one || two || kind == 3u;
}

template <typename E>
__global__ __launch_bounds__(kWalkThreads) void k_walk_win(DevAutomaton A, BatchArgs B, uint32_t dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t total_items = B.item_base[B.n_haystacks];
    if (uint64_t(blockIdx.x) * kWalkThreads >= total_items) return;
    const uint64_t item = uint64_t(blockIdx.x) * kWalkThreads + threadIdx.x;
    __shared__ uint32_t s_walked;
    if (threadIdx.x == 0) s_walked = 0;
    const uint4 rec = item < total_items ? B.item_rec[item] : make_uint4(0u, 0u, 0u, 0u);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.win_table);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t t = threadIdx.x; t < kWinLdsBytes / 16u; t += kWalkThreads) dst[t] = src[t];
    }
    __syncthreads();
    const uint16_t *L = reinterpret_cast<const uint16_t *>(smem);
    const uint8_t *lists = smem + kWinCellBytes;
    ItemDesc it;
    unpack_item(rec.x, rec.y, rec.z, rec.w, it);
    const uint32_t n = rec.z;  // 0 for threads past the last item
    const uint8_t *p = B.blob + (n ? it.begin - it.warm : 0);
    const uint32_t warm = it.warm, pos_bias = it.pos0 - it.warm + 1u;
    const uint32_t sh_hot = A.shallow_hot, trap = A.trap, sh_cold = A.shallow_cold_end;
    const uint32_t hbits = A.win_hash_bits, hmask = (1u << hbits) - 1u;
    EventRegs ev{0, 0};
    uint32_t st = 0;         // the state while it is deeper than 2
    bool deep = false;
    uint32_t idx = 0;        // the two bytes in front of the block, a << 8 | b
    uint32_t seen = 0;       // bytes walked so far, up to 2 (the first two bytes of a walk have no window in front of them)
    auto shallow = [&](uint32_t s) -> bool { return s < sh_hot || (s > trap && s < sh_cold); };
    auto event = [&](uint32_t jj) __attribute__((always_inline)) {
        if (jj >= warm && is_final(A, st) && !(B.first_only && ev.n_ev)) {
            ev.n_rec += A.out_count[st];
            if (ev.n_ev < B.slab_events) B.slab[slab_slot(B, item, ev.n_ev)] = make_uint2(pos_bias + jj, st);
            ++ev.n_ev;
        }
    };
    // One round: `nb` bytes (<= 16, the low bytes of blk) from item index jj on
    auto step_block = [&](const uint4 &blk, uint32_t jj, uint32_t nb) __attribute__((always_inline)) {
        const uint32_t wd[4] = {blk.x, blk.y, blk.z, blk.w};
        uint32_t mask = 0, listed = 0, ix = idx;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t c = (wd[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const uint32_t cell = L[ix];
            mask |= win_maybe(cell, c) ? (1u << k) : 0u;
            listed |= ((cell >> 14) == 3u && cell != 0xFFFFu) ? (1u << k) : 0u;
            ix = ((ix << 8) | c) & 0xFFFFu;
        }
        if (nb < 16u) mask &= (1u << nb) - 1u;
        if (seen < 2u) { mask &= ~((1u << (2u - seen)) - 1u); seen = (seen + nb) < 2u ? seen + nb : 2u; }
        listed &= mask;
        if (dbg & 2u) listed = 0u;
        if (dbg & 1u) mask = 0u;
        if (dbg & 4u) { idx = ix; return; }
        // (a 64-bit shift of the block's half: as a select of the four words by a lane-variable index hipcc keeps the block in
        // scratch memory and indexes it there)
        const uint64_t half_lo = uint64_t(blk.y) << 32 | blk.x, half_hi = uint64_t(blk.w) << 32 | blk.z;
        auto byte_at = [&](uint32_t k) -> uint32_t {
            const uint64_t m = 0ull - uint64_t((k >> 3) & 1u);   // (not a select either)
            const uint64_t h = (half_lo & ~m) | (half_hi & m);
            return uint32_t(h >> (8u * (k & 7u))) & 0xffu;
        };
        auto digram_before = [&](uint32_t k) -> uint32_t {   // the two bytes in front of byte k
            if (k >= 2u) return byte_at(k - 2u) << 8 | byte_at(k - 1u);
            return k == 1u ? ((idx << 8) | byte_at(0u)) & 0xFFFFu : idx;
        };
        // nodes of three children or more (one position in twelve on config 3's set): their lists, exactly -- LDS only, every lane at
        // its own next such position, BEFORE the loop that goes to global memory (in one loop with it nearly every turn had some lane
        // waiting for a probe: six round trips per block instead of one or two)
#pragma unroll 1
        while (__any(listed != 0u)) {
            if (listed != 0u) {
                const uint32_t k = uint32_t(__builtin_ctz(listed));
                listed &= listed - 1u;
                const uint32_t cell = L[digram_before(k)], c = byte_at(k);
                const uint8_t *lp = lists + ((cell & 0x3FFFu) << 1);
                const uint32_t cnt = lp[0];
                bool real = false;
                for (uint32_t i = 1; i <= cnt; ++i) real = real || lp[i] == c;
                if (!real) mask &= ~(1u << k);
            }
        }
        // the rare steps, every lane at its own next position
        uint32_t pos = 0;
        bool active = mask != 0u || deep;
#pragma unroll 1
        while (__any(active)) {
            if (active) {
                if (!deep) {
                    const uint32_t rem = pos < 16u ? (mask >> pos) << pos : 0u;
                    if (rem == 0u) {
                        active = false;
                    } else {
                        const uint32_t k = uint32_t(__builtin_ctz(rem));
                        const uint32_t dg = digram_before(k), c = byte_at(k);
                        {
                            const uint32_t key = (dg << 8 | c) + 1u;
                            uint32_t at = ((key - 1u) * 0x9E3779B1u) >> (32u - hbits);
                            uint2 e = A.win_hash[at];
                            while (e.x != 0u && (e.x & 0x7FFFFFFFu) != key) { at = (at + 1u) & hmask; e = A.win_hash[at]; }
                            if (e.x != 0u) {
                                // a node of one child, not final, its failure link shallow: text that does not go on with that child's
                                // byte is the table's again at the next byte -- no step through the records
                                const bool simple = (e.y >> 31) != 0u && k + 1u < nb;
                                const uint32_t cb = ((e.y >> 24) & 0x7Fu) | (e.x >> 31) << 7;
                                if (!(simple && byte_at(k + 1u) != cb)) {
                                    st = (e.y >> 31) ? (e.y & 0x00FFFFFFu) : e.y;
                                    deep = true;
                                    event(jj + k);
                                }
                            }
                        }
                        pos = k + 1u;
                    }
                } else if (pos >= nb) {
                    active = false;   // (deep into the next block)
                } else {
                    // one transition from a state deeper than 2
                    const uint32_t c = byte_at(pos);
                    const uint32_t coff = class_offset<E, 0>(A, nullptr, c);
                    const uint8_t *table = static_cast<const uint8_t *>(A.table);
                    uint32_t s = st, w = 0xFFFFFFFFu;
                    bool back = false;   // the failure links led to depth <= 2: the byte is the table's
#pragma unroll 1
                    for (;;) {
                        if (!A.compact || s < A.dense_nf_end) { w = *reinterpret_cast<const E *>(table + uint64_t(s) * A.row_bytes + coff); break; }
                        if (s >= A.f2 && s < A.dense_f_end) { w = *reinterpret_cast<const E *>(table + uint64_t(A.dense_nf_end + (s - A.f2)) * A.row_bytes + coff); break; }
                        const uint2 r = A.cold_rec[s];
                        const uint32_t code = r.y >> 24;
                        if (code == 1u) {
                            if ((r.x & 0xffu) == c) { w = r.x >> 8; break; }
                        } else if (code >= 2u) {
                            const uint32_t *e = A.cold_edges + r.x;
                            uint32_t ne = code;
                            if (code == 255u) ne = *e++;
                            uint32_t lo = 0, hi = ne;
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if ((e[mid] & 0xffu) < c) lo = mid + 1; else hi = mid;
                            }
                            if (lo < ne && (e[lo] & 0xffu) == c) { w = e[lo] >> 8; break; }
                        }
                        s = r.y & 0xFFFFFFu;
                        if (shallow(s)) { back = true; break; }
                    }
                    if (back) {
                        deep = false;   // (pos stays: the table's answer for this byte is in the mask)
                    } else {
                        st = w;
                        deep = !shallow(w);
                        if (deep) event(jj + pos);
                        ++pos;
                    }
                }
            }
        }
        idx = nb == 16u ? ix : digram_before(nb);
    };
    constexpr int LB = 4;
    uint32_t j = 0;
    // head: up to a 64-byte boundary (the bytes of one aligned 16-byte block at a time: a block is only touched where it holds a byte of the item)
    while (j < n && (reinterpret_cast<uintptr_t>(p + j) & (16u * LB - 1u)) != 0) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p + j);
        const uint32_t off = uint32_t(a & 15u), room = 16u - off, nb = room < n - j ? room : n - j;
        uint4 blk = *reinterpret_cast<const uint4 *>(a & ~uintptr_t(15));
        // shift the wanted bytes down to byte 0: whole words by selects, then the bytes (v_alignbyte)
        {
            uint32_t a0 = blk.x, a1 = blk.y, a2 = blk.z, a3 = blk.w;
            if (off & 4u) { a0 = a1; a1 = a2; a2 = a3; a3 = 0u; }
            if (off & 8u) { a0 = a2; a1 = a3; a2 = 0u; a3 = 0u; }
            const uint32_t bs = off & 3u;
            blk = make_uint4(__builtin_amdgcn_alignbyte(a1, a0, bs), __builtin_amdgcn_alignbyte(a2, a1, bs), __builtin_amdgcn_alignbyte(a3, a2, bs),
                             __builtin_amdgcn_alignbyte(0u, a3, bs));
        }
        step_block(blk, j, nb);
        j += nb;
    }
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
            for (int i = 0; i < LB; ++i) step_block(cur[i], j + 16u * uint32_t(i), 16u);
#pragma unroll
            for (int i = 0; i < LB; ++i) cur[i] = nxt[i];
            j += 16u * LB;
        }
    }
    // what is left: aligned blocks, the last one partial
    while (j < n) {
        const uint32_t nb = n - j < 16u ? n - j : 16u;
        step_block(*reinterpret_cast<const uint4 *>(p + j), j, nb);
        j += nb;
    }

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

hipError_t launch_walk_win(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), kWinLdsBytes);
        if (e != hipSuccess) return e;
        const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);
        static const uint32_t dbg = [] { const char *e = std::getenv("ACMI_WIN_DEBUG"); return e ? uint32_t(std::atoi(e)) : 0u; }();
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), kWinLdsBytes, st, A, B, dbg);
        return hipGetLastError();
    };
    return entry_bytes == 2 ? go(k_walk_win<uint16_t>) : go(k_walk_win<uint32_t>);
}

}  // namespace acmi
