// k_chain.hip -- k_walk_chain: the exact walk over the WHOLE automaton held in LDS in its chain-compressed form (kernel 6 for the
// pattern sets whose dense rows do not fit a CU's LDS but whose goto/failure records do: a few thousand patterns).
//
// What it replaces: ac_trie_search's per-byte loop (src/multifast/ahocorasick.c:199-212: find the edge, else follow the failure
// pointer) over nodes that hold a sorted edge array and a failure pointer (node.h:37-67).  k_walk_tiles keeps dense failure-free
// rows -- one read per byte -- but only the first few thousand states' rows fit the LDS; every step from a state outside them was a
// 128-byte line from L2 (round 4: 597 M L2 requests per GiB of prose under a 2,048-word list, 0.08-0.10 of the HBM roofline between
// "a match now and then" and "a match at every byte").  Here every state is in LDS (automaton.hpp ChainImage):
//   * D states (two or more children, and what the construction below needs): a dense failure-free row of uint16 targets;
//   * C states (one child or none: the insides of the patterns, 85-90 % of such a set): 4 bytes {class of the edge to the next
//     index, records of that child's event, failure state}.  The failure state of a C state is a D state BY CONSTRUCTION, so
//     delta(s, b) = b is the edge ? s + 1 : row[fail(s)][b] -- the reference's hop loop, ended after one hop.
// A step is the same straight-line code for every lane whatever its state: read the state's record, read the row entry (of the
// state itself when it is a D state, of its failure state otherwise), select.  Two LDS reads per byte, no branch, no L2 request.
//
// Work distribution as in k_walk_tiles (a work item <= P x W bytes of one haystack walked by P lanes, W bytes each, entered from
// the root Lmax - 1 bytes early), but a WAVE is the unit: it takes a ticket, walks its R = 64 / P items, publishes its record count,
// looks back over the waves before it (decoupled look-back, one status word per wave tile) and writes its records -- no workgroup
// barrier in the loop, so that one wave's look-back and its expansion (L2 round trips) overlap the other waves' walks (LDS and
// VALU).  Events go to a list of the LANE's own: {index, records, byte} in one word -- the first kChainLaneCap in LDS, what a lane
// has beyond that (it cannot have more than one per byte of its piece) in the wave's region of a global buffer.
#include <cstdlib>
#include <type_traits>

#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kChainThreads = 1024;
constexpr uint32_t kChainWaves = kChainThreads / 64;
constexpr uint32_t kChainLaneCap = 4;   // events per lane kept in LDS: 16 KiB for the workgroup
constexpr uint32_t kChainBatch = 16;    // wave tiles a workgroup takes from the global ticket at a time: one round of its waves -- the
                                        // first tile of a batch looks back on the LAST of the batch before, which its workgroup must
                                        // not walk rounds later (64 per ticket: every workgroup waited three rounds for its neighbour)

struct ChainArgs {
    uint32_t W, P, R, back;
    uint32_t spill_cap;   // events per lane the wave's global region holds
    uint32_t *spill;      // grid waves x spill_cap x 64 words
    uint32_t *ticket;     // zeroed
    uint64_t *status;     // zeroed, one word per wave tile
    uint4 *out;
    uint64_t cap;
};

struct ChainLds {
    const uint8_t *img;       // rows at 0
    uint32_t n_dense;
    uint32_t rec_base;        // rec_off - 4 x n_dense: record of index e at img + rec_base + 4 e
    uint32_t row_bytes;
    const uint8_t *cnt;       // min(records, 255) of the indices a row can lead to
    const uint16_t *cls;      // byte -> 2 x class (TABLE mode), else null
};

// One transition.  `coff` = 2 x class of the byte.  Returns through `e` the next index; `u` bit 15: that state is final (`hit`: it
// was entered along its chain, bits 10..14 of `u` are min(its records, 31)).
__device__ __forceinline__ void chain_step(const ChainLds &L, uint32_t &e, uint32_t coff, uint32_t &u, bool &hit) {
    const uint32_t r = *reinterpret_cast<const uint32_t *>(L.img + ((e << 2) + L.rec_base));   // (a D state reads a word of the rows: unused)
    const bool is_d = e < L.n_dense;
    const uint32_t f = is_d ? e : (r >> 16);
    uint32_t ra;   // (spelled out: hipcc makes a multiply and a three-operand add of it)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ra) : "v"(f), "s"(L.row_bytes), "v"(coff));
    const uint32_t t = *reinterpret_cast<const uint16_t *>(L.img + ra);
    hit = !is_d && (r & 0x3FFu) == coff;   // (the records' keys are doubled when the image is staged; 0x3FF -- no child -- is odd)
    u = hit ? r : t;
    e = hit ? e + 1u : (t & 0x7FFFu);
}

__device__ __forceinline__ uint32_t chain_wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t chain_wave_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// records of the event of index `e` when the short forms saturated
__device__ __forceinline__ uint32_t chain_count(const DevAutomaton &A, uint32_t e) {
    const uint32_t c = A.chain_head[e].x >> 24;
    return c < 255u ? c : A.out_count[A.chain_orig[e]];
}

template <int MODE>
__global__ __launch_bounds__(kChainThreads) void k_walk_chain(DevAutomaton A, BatchArgs B, ChainArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // Tickets.  A wave tile is 64 lane pieces (4 KiB of text): a GiB is 262,144 of them, and atomics on ONE word retire at ~ 12 ns
    // each whoever issues them -- 3.1 ms, the whole kernel's time when every wave took its own (engine_pipeline.cpp has the same
    // note on the persistent scans).  A workgroup therefore takes kChainBatch consecutive tiles per global atomic and hands them to
    // its waves, in order, through a counter in LDS: local ticket i = tile i % kChainBatch of the workgroup's batch i / kChainBatch,
    // whose global number the wave with i % kChainBatch == 0 fetches for all.  Look-back stays deadlock-free: the smallest tile that
    // has not published is either being walked, or the next one its workgroup hands out.
    __shared__ uint32_t s_cur, s_bid[4], s_base[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    if (threadIdx.x == 0) s_cur = 0u;
    if (threadIdx.x < 4) s_bid[threadIdx.x] = 0xFFFFFFFFu;
    // LDS: [image: rows | records | counts] [512 B byte classes (TABLE mode)] [the lanes' event lists]
    const uint32_t img_bytes = (A.chain_img_bytes + 15u) & ~15u;
    const uint32_t cls_bytes = (MODE == 0) ? 512u : 0u;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.chain_img);   // (the allocation has 16 bytes of slack)
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t t = threadIdx.x; t < (img_bytes >> 4); t += kChainThreads) dst[t] = src[t];
        if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem + img_bytes)[threadIdx.x] = A.class_off[threadIdx.x];
        __syncthreads();
        // the records' keys doubled: a step compares them with the byte's column OFFSET
        uint32_t *recs = reinterpret_cast<uint32_t *>(smem + A.chain_rec_off);
        const uint32_t n_rec = A.chain_index - A.chain_dense;
        for (uint32_t t = threadIdx.x; t < n_rec; t += kChainThreads) {
            const uint32_t w = recs[t], key = w & 0x3FFu;
            recs[t] = (w & ~0x3FFu) | (key == 0x3FFu ? key : key << 1);
        }
        __syncthreads();
    }
    ChainLds L;
    L.img = smem;
    L.n_dense = A.chain_dense;
    L.rec_base = A.chain_rec_off - 4u * A.chain_dense;
    L.row_bytes = A.n_columns * 2u;
    L.cnt = smem + A.chain_cnt_off;
    L.cls = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem + img_bytes) : nullptr;
    uint32_t *lists = reinterpret_cast<uint32_t *>(smem + img_bytes + cls_bytes) + threadIdx.x;   // entry k of this lane at lists[k * 1024]
    uint32_t *spill = T.spill + (uint64_t(blockIdx.x) * kChainWaves + wave) * T.spill_cap * 64u + lane;   // entry k at spill[(k - cap) * 64]
    const uint32_t r = lane / T.P, qi = lane - r * T.P;   // item of the wave, piece of the item
    const uintptr_t beta = reinterpret_cast<uintptr_t>(B.blob) & 15u;
    const uint8_t *abase = B.blob - beta;   // 16-byte aligned
    uint32_t walked16 = 0;
    const uint32_t cnt_n = A.chain_cnt_n;
#pragma unroll 1
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) {
            const uint32_t i = atomicAdd(&s_cur, 1u);
            const uint32_t b = i / kChainBatch, o = i % kChainBatch, slot = b & 3u;
            if (o == 0u) {
                const uint32_t g = atomicAdd(T.ticket, 1u);
                __hip_atomic_store(&s_base[slot], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_bid[slot], b, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (__hip_atomic_load(&s_bid[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != b) __builtin_amdgcn_s_sleep(1);
            }
            t = __hip_atomic_load(&s_base[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * kChainBatch + o;
        }
        t = uint32_t(__builtin_amdgcn_readfirstlane(int(t)));
        const uint64_t first_item = uint64_t(t) * T.R;
        if (first_item >= total_items) break;
        const uint64_t item = first_item + r;
        const bool have = r < T.R && item < total_items;
        // this lane's piece (k_walk_tiles' geometry): the first block it walks, where in it the walk begins, what reports
        const uint8_t *pb = A.pad16;
        uint32_t nblk = 0, start_j = 16u, rep_lo = 0, rep_len = 0, pos_bias = 0, hay = 0;
        if (have) {
            const uint4 rec = B.item_rec[item];
            ItemDesc it;
            unpack_item(rec.x, rec.y, rec.z, rec.w, it);
            hay = B.item_hay[item];
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
                int64_t xb = xs & ~int64_t(15);
                if (xb == xs) xb -= 16;   // (a walk that begins ON a block boundary takes the block below as its first, virtually)
                pb = abase + G + xb;
                start_j = uint32_t(xs - xb);
                rep_lo = uint32_t(int64_t(lo) - xb);
                rep_len = hi - lo;
                nblk = (rep_lo + rep_len + 15u) >> 4;
                pos_bias = uint32_t(xb) - d0 + it.pos0 + 1u;
                walked16 += nblk;
            }
        }
        // ---- the walk: all lanes in step, a lane past its own bytes walks on without reporting ----
        uint32_t e = 0, n_ev = 0, n_rec = 0;
        const uint32_t blocks = chain_wave_max(nblk);
        if (blocks) {
            // An event inside the walk is ONE store: {index, what the parent's record says about its records (entered along the
            // chain; else 0), byte} to the lane's list.  What it stands for is worked out behind the walk, where a lane runs over
            // its own events only -- in here the whole wave pays for every step in which ANY lane has one.
            auto event = [&](uint32_t jj, uint32_t u, bool hit, bool inside) __attribute__((always_inline)) {
                if ((u & 0xFFFFu) >= 0x8000u && inside) {
                    const uint32_t ent = e | ((hit ? u : 0u) & 0x7C00u) << 5 | jj << 23;
                    if (n_ev < kChainLaneCap) lists[n_ev * kChainThreads] = ent;
                    else *reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(spill + (n_ev - kChainLaneCap) * 64u)) = ent;
                    ++n_ev;
                }
            };
            // sixteen unrolled steps per block; the column offsets of a dword's four bytes are worked out together (SWAR3: a handful
            // of operations for all four; TABLE: four independent LDS reads).  CHECKED: the range test of a block that does not lie
            // inside every lane's own range; WARM: the first block, in which a lane's walk begins (the state is the root before).
            auto dword = [&](uint32_t wd, uint32_t jj, auto checked, auto warm) __attribute__((always_inline)) {
                constexpr bool C = decltype(checked)::value, W0 = decltype(warm)::value;
                uint32_t c0, c1, c2, c3;
                if constexpr (MODE == 1) {
                    const uint32_t c4 = swar_columns<uint16_t>(A, wd);
                    c0 = c4 & 0xffu; c1 = (c4 >> 8) & 0xffu; c2 = (c4 >> 16) & 0xffu; c3 = c4 >> 24;
                } else {
                    c0 = class_offset<uint16_t, MODE>(A, L.cls, wd & 0xffu); c1 = class_offset<uint16_t, MODE>(A, L.cls, (wd >> 8) & 0xffu);
                    c2 = class_offset<uint16_t, MODE>(A, L.cls, (wd >> 16) & 0xffu); c3 = class_offset<uint16_t, MODE>(A, L.cls, wd >> 24);
                }
                uint32_t u;
                bool hit;
                auto one = [&](uint32_t cx, uint32_t j) __attribute__((always_inline)) {
                    chain_step(L, e, cx, u, hit);
                    if (W0) e = j < start_j ? 0u : e;
                    event(j, u, hit, !C || j - rep_lo < rep_len);   // (rep_lo >= start_j: nothing before the walk's first byte reports)
                };
                one(c0, jj); one(c1, jj + 1u); one(c2, jj + 2u); one(c3, jj + 3u);
            };
            const uint4 *q = reinterpret_cast<const uint4 *>(pb);
            const uint32_t last = nblk ? nblk - 1u : 0u;
            const uint4 first = q[start_j < 16u ? 0u : (last ? 1u : 0u)];   // (start_j == 16: block 0 is virtual)
            uint4 b0 = q[1u < last ? 1u : last], b1 = q[2u < last ? 2u : last], b2 = q[3u < last ? 3u : last], b3 = q[4u < last ? 4u : last];
            if (chain_wave_min(nblk ? start_j : 16u) < 16u) {   // (every walk of the wave begins on a block boundary: nothing to do in block 0)
                dword(first.x, 0u, std::true_type{}, std::true_type{}); dword(first.y, 4u, std::true_type{}, std::true_type{});
                dword(first.z, 8u, std::true_type{}, std::true_type{}); dword(first.w, 12u, std::true_type{}, std::true_type{});
            }
#pragma unroll 1
            for (uint32_t k = 1u; k < blocks; ++k) {
                asm volatile("; acmi-hot-begin");   // (tools/check_asm.py: no scratch traffic between the markers)
                const uint4 cur = b0;
                b0 = b1; b1 = b2; b2 = b3;
                const uint32_t kn = k + 4u;
                if (kn < blocks) b3 = q[kn < last ? kn : last];
                const uint32_t j0 = k * 16u;
                // (a block that lies inside EVERY lane's own range -- all but the first and the last of a piece -- without the range test)
                const bool full = j0 >= rep_lo && j0 + 16u <= rep_lo + rep_len;
                if (__all(full)) {
                    dword(cur.x, j0, std::false_type{}, std::false_type{}); dword(cur.y, j0 + 4u, std::false_type{}, std::false_type{});
                    dword(cur.z, j0 + 8u, std::false_type{}, std::false_type{}); dword(cur.w, j0 + 12u, std::false_type{}, std::false_type{});
                } else {
                    dword(cur.x, j0, std::true_type{}, std::false_type{}); dword(cur.y, j0 + 4u, std::true_type{}, std::false_type{});
                    dword(cur.z, j0 + 8u, std::true_type{}, std::false_type{}); dword(cur.w, j0 + 12u, std::true_type{}, std::false_type{});
                }
                asm volatile("; acmi-hot-end");
            }
        }
        if (n_ev > kChainLaneCap) {   // (this lane's own stores: landed before they are read back)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // records of an event: five bits of the parent's record (entered along the chain), else the count byte of a state a row led
        // to -- if it has one: a row may also lead INTO a chain (the child of a C state on the row's failure path) -- else, and for
        // saturated counts, the tables in global memory
        auto records_of = [&](uint32_t ent) __attribute__((always_inline)) -> uint32_t {
            const uint32_t idx = ent & 0x7FFFu, code = (ent >> 15) & 31u;
            if (code != 0u && code != 31u) return code;
            if (code == 0u && idx < cnt_n) {
                const uint32_t c = L.cnt[idx];
                if (c != 255u) return c;
            }
            return chain_count(A, idx);
        };
#pragma unroll 1
        for (uint32_t k = 0; k < n_ev; ++k)
            n_rec += records_of(k < kChainLaneCap ? lists[k * kChainThreads] : __hip_atomic_load(spill + (k - kChainLaneCap) * 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        // ---- records: exclusive prefix over the wave's lanes, the items' totals, the wave's total; its base by look-back ----
        const uint32_t incl = wave_inclusive_sum(n_rec);
        const uint32_t excl = incl - n_rec;
        const uint32_t wave_tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        {
            const uint32_t first_lane = r * T.P, last_lane = first_lane + T.P - 1u;
            const uint32_t item_tot = uint32_t(__shfl(int(incl), int(last_lane < 64u ? last_lane : 63u), 64)) - uint32_t(__shfl(int(excl), int(first_lane < 64u ? first_lane : 63u), 64));
            if (have && qi == 0u && item_tot) B.rec_count[item] = item_tot;
        }
        const uint64_t base = chain_exclusive(T.status, t, wave_tot);
        // (too small a buffer: what fits is written -- whole events, in order -- and the host grows it and runs the batch again, or hands
        // out the haystacks that are complete: "max_result_bytes")
        if (wave_tot == 0u || base >= T.cap) continue;
        const uint64_t room = T.cap - base;
        uint4 *dst = T.out + base;
        uint64_t at = excl;
#pragma unroll 1
        for (uint32_t k = 0; k < n_ev; ++k) {
            // (its own stores, read back from L2: the CU's L1 may still hold the line as a tile before left it)
            const uint32_t ent = k < kChainLaneCap ? lists[k * kChainThreads] : __hip_atomic_load(spill + (k - kChainLaneCap) * 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t idx = ent & 0x7FFFu, pos = pos_bias + (ent >> 23);
            const uint2 h = A.chain_head[idx];
            const uint32_t c = records_of(ent);
            if (at + c <= room) {
                dst[at] = make_uint4(hay, pos, pos - h.y, h.x & 0xFFFFFFu);
                if (c > 1u) {
                    const uint32_t s = A.chain_orig[idx];
                    if (A.out_list != nullptr) {
                        const uint2 *ol = A.out_list + A.out_off[s];
                        for (uint32_t i = 1; i < c; ++i) {
                            const uint2 o = ol[i];
                            dst[at + i] = make_uint4(hay, pos, pos - o.y, o.x);
                        }
                    } else {
                        (void)emit_event(A, dst, at, hay, pos, s);
                    }
                }
            }
            at += c;
        }
    }
    // statistic: 16-byte blocks walked (what the automatic kernel choice reads)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) walked16 += __shfl_xor(walked16, d, 64);
    if (lane == 0 && walked16) atomicAdd(&B.counters[8u + ((blockIdx.x + wave) & 7u)], walked16);
}

// Does this automaton have the image, and does the kernel's LDS layout hold it?
bool chain_walk_ok(const DevAutomaton &A) {
    if (A.chain_img == nullptr || A.chain_head == nullptr) return false;
    const uint32_t img = (A.chain_img_bytes + 15u) & ~15u;
    return img + 512u + kChainWaves * kChainLaneCap * 256u <= 160u * 1024u;
}

uint32_t chain_grid() { return uint32_t(num_cus()); }
uint64_t chain_status_words(uint64_t max_items, uint32_t P) { return max_items / (64u / P) + 66u + uint64_t(chain_grid()) * kChainBatch; }
// the waves' global event regions: what a lane can have beyond its LDS list, a word each
// (an item's last piece takes the item's unaligned head as well: up to W + 15 bytes, an event each)
uint64_t chain_spill_words(uint32_t W) { return uint64_t(chain_grid()) * kChainWaves * uint64_t(W + 16u) * 64u; }

hipError_t launch_walk_chain(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int class_mode, uint32_t W, uint32_t P, void *spill,
                             uint32_t *ticket, uint64_t *status, void *out, uint64_t cap, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    if (!chain_walk_ok(A)) return hipErrorInvalidValue;
    ChainArgs T{};
    T.W = W;
    T.P = P;
    T.R = 64u / P;
    T.back = A.max_pattern_length ? A.max_pattern_length - 1u : 0u;
    T.spill_cap = W + 16u;
    T.spill = static_cast<uint32_t *>(spill);
    T.ticket = ticket;
    T.status = status;
    T.out = static_cast<uint4 *>(out);
    T.cap = cap;
    const uint32_t img = (A.chain_img_bytes + 15u) & ~15u;
    const uint32_t lds = img + (class_mode == 0 ? 512u : 0u) + kChainWaves * kChainLaneCap * 256u;
    const uint64_t tiles = (max_items + T.R - 1u) / T.R;
    const uint32_t grid = chain_grid() - std::min<uint32_t>(B.reserve_cus, chain_grid() - 1u);
    const uint32_t blocks = uint32_t(std::min<uint64_t>((tiles + kChainWaves - 1u) / kChainWaves, grid));
    auto go = [&](auto fn) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kChainThreads), lds, st, A, B, T);
        return hipGetLastError();
    };
    return class_mode == 1 ? go(k_walk_chain<1>) : go(k_walk_chain<0>);
}

}  // namespace acmi
