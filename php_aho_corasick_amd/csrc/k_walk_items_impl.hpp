// k_walk_items_impl.hpp -- the kernel template of k_walk_items.hip (scan form) and k_walk_items_emit.hip (drain form):
// two translation units, compiled side by side
#pragma once
#include "kdev.hpp"

namespace acmi {
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

__device__ __forceinline__ uint64_t emit_event(const DevAutomaton &A, uint4 *out, uint64_t w, uint32_t hay, uint32_t pos, uint32_t state);

// EMIT: the same walk as the DRAIN of a batch this kernel scanned -- a thread per OVERFLOWED item (its events did not fit
// the slab), records written straight into their final slots.  Items of <= 1 KiB are what this kernel is given, and at tens of
// events per item a wave per item (k_refill_wave: a round of dependent loads per 256 positions) was five times slower than
// walking them again one per thread (a word list over text made of its words: 14 -> 3.6 ms per GiB).
// D32 (dense rows, the table below 4 GiB): a row outside the tile is one load at a 32-bit offset from the table's base -- no test
// of the automaton's form, no 64-bit multiply, none of the compact form's record loop in the 64 unrolled steps (k_walk_tiles'
// lesson: 17 % there).  The tile ends with a row of "not here" entries for the states it does not hold.
template <typename E, int MODE, int LB, bool EMIT = false, bool D32 = false>
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
    const uint32_t fit = A.row_bytes ? (kItemsTileBytes - 16u) / A.row_bytes - 1u : 0u;   // (one row more than states: the trap's)
    const uint32_t tile_rows = fit < A.trap ? fit : A.trap;
    const uint16_t *cls_lds = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem) : nullptr;
    const uint8_t *tile = smem + cls_bytes;
    // a tile that leaves room takes its states' record counts along: where nearly every byte ends a pattern (a^1..a^16)
    // the count was a dependent global load per byte, and that load was most of the kernel (0.33 ms per 32 MiB)
    const uint32_t tile_bytes = ((tile_rows + 1u) * A.row_bytes + 15u) & ~15u;
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
        if (tile_rows < A.trap) {   // the image's own row `trap` is such a row; a shorter tile gets one written behind its last
            __syncthreads();
            E *trap_row = reinterpret_cast<E *>(smem + cls_bytes + tile_rows * A.row_bytes);
            for (uint32_t t = threadIdx.x; t < A.n_columns; t += kWalkThreads) trap_row[t] = E(A.trap);
        }
    }
    __syncthreads();

    ItemDesc it;
    unpack_item(rec.x, rec.y, rec.z, rec.w, it);
    const uint32_t n = rec.z;  // 0 for threads past the last item
    const uint8_t *p = B.blob + (n ? it.begin - it.warm : 0);
    const uint32_t warm = it.warm, pos_bias = it.pos0 - it.warm + 1u;
    uint32_t st = 0;
    EventRegs ev{0, 0};
    auto step_c = [&](uint32_t byte, uint32_t coff, uint32_t jj) __attribute__((always_inline)) {
        const uint32_t row = st < tile_rows ? st : tile_rows;
        uint32_t w = *reinterpret_cast<const E *>(tile + __umul24(row, A.row_bytes) + coff);
        if (w >= A.trap) {
            if constexpr (D32) w = *reinterpret_cast<const E *>(static_cast<const uint8_t *>(A.table) + (__umul24(st, A.row_bytes) + coff));
            else w = cold_step<E>(A, st, byte, coff);
        }
        st = w;
        if (jj >= warm && is_final(A, st) && !(B.first_only && ev.n_ev)) {
            if constexpr (EMIT) {
                wr = emit_event(A, out, wr, hay, pos_bias + jj, st);
            } else {
                // (two loads, each in its own address space: as one select of two pointers hipcc emitted a flat load, which counts
                // on vmcnt AND lgkmcnt -- every event then waited for the text requested a group ahead)
                if (st < cnt_rows) ev.n_rec += *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(uint32_t(reinterpret_cast<uintptr_t>(cnt_lds)) + st * 4u);
                else ev.n_rec += A.out_count[st];
                if (ev.n_ev < B.slab_events) B.slab[slab_slot(B, item, ev.n_ev)] = make_uint2(pos_bias + jj, st);
            }
            ++ev.n_ev;
        }
    };
#define ACMI_STEP1(b, jj) step_c((b), class_offset<E, MODE>(A, cls_lds, (b)), (jj))
    // SWAR3: the column offsets of a dword's four bytes together (a handful of operations for all four instead of eight per byte;
    // TABLE stays byte by byte: four class reads in flight cost this kernel the registers it has; so does the drain form, which
    // the compiler then gave 37 registers and an array in scratch memory: 3.9 -> 4.7 ms on the short patterns' drain)
    auto step4 = [&](uint32_t wd, uint32_t jj) __attribute__((always_inline)) {
        const uint32_t c4 = swar_columns<E>(A, wd);
        step_c(wd & 0xffu, c4 & 0xffu, jj); step_c((wd >> 8) & 0xffu, (c4 >> 8) & 0xffu, jj + 1u);
        step_c((wd >> 16) & 0xffu, (c4 >> 16) & 0xffu, jj + 2u); step_c(wd >> 24, c4 >> 24, jj + 3u);
    };
    // (written out: as a loop over an array of the four words hipcc kept the array in scratch memory)
#define ACMI_STEP4(wd, jj) \
    if constexpr (MODE == 1 && !EMIT) { step4((wd), (jj)); } \
    else { ACMI_STEP1((wd) & 0xffu, (jj)); ACMI_STEP1(((wd) >> 8) & 0xffu, (jj) + 1u); ACMI_STEP1(((wd) >> 16) & 0xffu, (jj) + 2u); ACMI_STEP1((wd) >> 24, (jj) + 3u); }
    auto step_block = [&](const uint4 &blk, uint32_t jj) __attribute__((always_inline)) {
        ACMI_STEP4(blk.x, jj) ACMI_STEP4(blk.y, jj + 4u) ACMI_STEP4(blk.z, jj + 8u) ACMI_STEP4(blk.w, jj + 12u)
    };
#undef ACMI_STEP4
    uint32_t j = 0;
    // head: single bytes up to a 16-byte boundary, single blocks up to a group boundary
    for (; j < n && (reinterpret_cast<uintptr_t>(p + j) & 15u) != 0; ++j) ACMI_STEP1(uint32_t(p[j]), j);
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
    for (; j < n; ++j) ACMI_STEP1(uint32_t(p[j]), j);
#undef ACMI_STEP1

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

}  // namespace acmi
