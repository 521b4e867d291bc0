// k_misc.hip -- k_stream_read, the device-side automaton build, search and replace, split-set merge, the exchange's packing kernel
#include <mutex>
#include <unordered_map>
#include "kdev.hpp"

namespace acmi {

hipError_t ensure_dynamic_lds(const void *kernel, uint32_t bytes) {
    static std::mutex m;
    static std::unordered_map<uint64_t, uint32_t> have;   // (device, kernel) -> what its attribute is: the attribute is the device's
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(m);
    uint32_t &cur = have[(uint64_t(uint32_t(dev)) << 56) ^ uint64_t(reinterpret_cast<uintptr_t>(kernel))];
    if (bytes <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e == hipSuccess) cur = bytes;
    return e;
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
// k_gather_pack / k_gather_unpack: the all-gatherv of match arrays (engine.cpp acmi_comm_allgather_result / _gather_result).
// Ranks own contiguous haystack ranges in rank order, so the global arrays are the ranks' arrays back to back.  A rank writes
// its own slice of them -- haystack ids made global, prefix words moved by the records of the ranks before it -- and, for the
// wire, its records as 8 bytes each {pos, pattern}: a record's haystack follows from the prefix array that travels anyway and
// its start from the pattern's length (start = pos - length, php_ahocorasick.c:578), so half of every 16-byte record need not
// cross the links.  What arrives from the other ranks is expanded in place by k_gather_unpack: the haystack of global record
// g is the h with prefix[h] <= g < prefix[h + 1].  `closing` = the global prefix array's last word (the total).
// ------------------------------------------------------------------------------------------
__global__ void k_gather_pack(const uint64_t *prefix, const uint4 *records, uint64_t nh, uint64_t nrec, uint64_t hay_base,
                              uint64_t rec_base, uint64_t *out_prefix, uint4 *out_records, uint2 *out_wire, uint64_t *closing, uint64_t total_rec) {
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g < nrec) {
        uint4 v = records[g];
        v.x += uint32_t(hay_base);
        if (out_records) out_records[g] = v;
        out_wire[g] = make_uint2(v.y, v.w);
    }
    if (g < nh) out_prefix[g] = prefix[g] + rec_base;
    if (g == 0 && closing) *closing = total_rec;
}

void launch_gather_pack(const uint64_t *prefix, const uint4 *records, uint64_t nh, uint64_t nrec, uint64_t hay_base, uint64_t rec_base,
                        uint64_t *out_prefix, uint4 *out_records, uint2 *out_wire, uint64_t *closing, uint64_t total_rec, hipStream_t st) {
    const uint64_t threads = std::max<uint64_t>(1, std::max(nrec, nh));
    hipLaunchKernelGGL(k_gather_pack, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, st, prefix, records, nh, nrec, hay_base,
                       rec_base, out_prefix, out_records, out_wire, closing, total_rec);
}

// Records [skip_lo, skip_hi) are this rank's own (written whole by k_gather_pack); every other one is expanded from the wire.
__global__ void k_gather_unpack(const uint64_t *prefix, uint64_t total_hay, const uint2 *wire, uint64_t total_rec, uint64_t skip_lo, uint64_t skip_hi,
                                const uint32_t *pattern_length, uint4 *out_records) {
    __shared__ uint64_t s_lo, s_hi;
    const uint64_t g0 = uint64_t(blockIdx.x) * blockDim.x;
    if (g0 >= total_rec) return;
    const uint64_t g1 = g0 + blockDim.x - 1u < total_rec ? g0 + blockDim.x - 1u : total_rec - 1u;
    // the haystacks this block's records lie in: the last h with prefix[h] <= g, for its first and its last record
    if (threadIdx.x < 2u) {
        const uint64_t g = threadIdx.x ? g1 : g0;
        uint64_t lo = 0, hi = total_hay;   // prefix[lo] <= g < prefix[hi] (prefix[total_hay] = total_rec > g)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= g) lo = mid; else hi = mid;
        }
        (threadIdx.x ? s_hi : s_lo) = lo;
    }
    __syncthreads();
    const uint64_t g = g0 + threadIdx.x;
    if (g >= total_rec || (g >= skip_lo && g < skip_hi)) return;
    uint64_t lo = s_lo, hi = s_hi + 1;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= g) lo = mid; else hi = mid;
    }
    const uint2 w = wire[g];
    out_records[g] = make_uint4(uint32_t(lo), w.x, w.x - pattern_length[w.y], w.y);
}

void launch_gather_unpack(const uint64_t *prefix, uint64_t total_hay, const uint2 *wire, uint64_t total_rec, uint64_t skip_lo, uint64_t skip_hi,
                          const uint32_t *pattern_length, uint4 *out_records, hipStream_t st) {
    if (total_rec == 0) return;
    hipLaunchKernelGGL(k_gather_unpack, dim3(uint32_t((total_rec + 255) / 256)), dim3(256), 0, st, prefix, total_hay, wire, total_rec, skip_lo, skip_hi,
                       pattern_length, out_records);
}

}  // namespace acmi
