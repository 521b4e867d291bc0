// k_drain.hip -- the drains of items whose events overflowed their slabs: k_refill, k_refill_wave
#include "kdev.hpp"

namespace acmi {

void launch_walk_items_emit(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, const uint64_t *rec_base, void *out,
                            uint64_t cap, hipStream_t st, int class_mode);

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
            const uint32_t w = lds_read<E>(tile, s * A.row_bytes + coff);
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

void launch_refill(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes,
                   const uint64_t *rec_base, void *out, uint64_t cap, int mode, hipStream_t st, int class_mode) {
    if (max_items == 0) return;
    if (mode == 3) {   // a batch of k_walk_items: the same walk again, a thread per overflowed item, writing records (k_walk.hip)
        launch_walk_items_emit(A, B, max_items, entry_bytes, rec_base, out, cap, st, class_mode);
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
            (void)ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
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
