// k_lds.hip -- k_scan_simple and k_scan_lds (the LDS-tile table walk), launch_scan
#include "kdev.hpp"
#include "scan_asm.inc"

namespace acmi {

hipError_t launch_anchor(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, hipStream_t st, uint32_t *variant);

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

template <typename E, int MODE, int ILP, int G>
static hipError_t launch_lds_t(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint32_t lds_bytes,
                               hipStream_t st) {
    auto fn = k_scan_lds<E, MODE, ILP, G>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds_bytes));
    if (e != hipSuccess) return e;
    // persistent: at most one workgroup per CU (the tile fills the CU's LDS), fewer for small batches
    const uint64_t per_block = uint64_t(kLdsThreads) * ILP;
    const uint64_t want = (max_items + per_block - 1) / per_block;
    const uint32_t blocks = B.persistent ? uint32_t(std::min<uint64_t>(want, uint64_t(std::max(1, num_cus() - int(B.reserve_cus))))) : uint32_t(want);
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

}  // namespace acmi
