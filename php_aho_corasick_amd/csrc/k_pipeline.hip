// k_pipeline.hip -- the kernels around the scan: work-item table (k_items), findAll=false, expansion of events into records (k_expand)
#include "kdev.hpp"

namespace acmi {

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

// The same tables when every haystack has the same length and they lie back to back (acmi_match_batch_device_begin_uniform):
// item g is chunk g mod ipp of haystack g / ipp, item_base[h] = h x ipp -- nothing to count, nothing to scan, no look-back
// (k_items: 19 us for 2^19 haystacks, most of it two chained scans; this: one pass of independent stores).
__global__ __launch_bounds__(256) void k_items_uniform(BatchArgs B, uint64_t items_per_hay, uint64_t *item_base_out) {
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t items = items_per_hay * B.n_haystacks;
    if (g <= B.n_haystacks) item_base_out[g] = g * items_per_hay;
    if (g >= items) return;
    const uint64_t h = g / items_per_hay;
    fill_item(B, h, g - h * items_per_hay, g);
}

// offsets[h] = h x haystack_bytes for h = 0 .. n (the library's own offsets array of a uniform batch)
__global__ void k_uniform_offsets(uint64_t *offsets, uint64_t n, uint64_t haystack_bytes) {
    const uint64_t h = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (h <= n) offsets[h] = h * haystack_bytes;
}

void launch_uniform_offsets(uint64_t *offsets, uint64_t n, uint64_t haystack_bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_uniform_offsets, dim3(uint32_t((n + 256) / 256)), dim3(256), 0, st, offsets, n, haystack_bytes);
}

void launch_items_uniform(const BatchArgs &B, uint64_t haystack_bytes, uint64_t *item_base, hipStream_t st) {
    const uint64_t ipp = (haystack_bytes + B.chunk_bytes - 1) / B.chunk_bytes;
    const uint64_t threads = std::max<uint64_t>(ipp * B.n_haystacks, uint64_t(B.n_haystacks) + 1);
    hipLaunchKernelGGL(k_items_uniform, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, st, B, ipp, item_base);
}

uint32_t scan_tiles(uint64_t n) { return uint32_t((n + kScanTile - 1) / kScanTile); }

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

// LOWREG: 64 registers (five spilled), two workgroups per CU -- for batches of millions of items (haystacks of 256 B, an item each:
// 4.3 M items 107 -> 88 us), whose tiles come in many rounds; the headline's 513 tiles are 3 us faster with 72 registers and one.
template <bool SCAN, bool LOWREG = false>
__global__ __launch_bounds__(kExpandThreads, LOWREG ? 8 : 4) void k_expand(DevAutomaton A, BatchArgs B, uint64_t *rec_base, uint4 *out, uint64_t cap,
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
    // What the item's first event needs does not depend on the scan: requested here, in flight through the scan and the
    // look-back (four dependent loads used to start behind them; an item usually has one event of one record).
    static_assert(kExpandPerThread == 1, "the prefetch below is written for one item per thread");
    uint32_t hay0 = 0, prev0 = 0xFFFFFFFFu, n0 = 0;
    uint2 ev0 = make_uint2(0u, 0u), ev1 = make_uint2(0u, 0u), head0 = make_uint2(0u, 0u);
    if (base < items) {
        hay0 = B.item_hay[base];
        prev0 = base ? B.item_hay[base - 1] : 0xFFFFFFFFu;
        n0 = B.ev_count[base];
        if (n0 != 0u && n0 <= B.slab_events) {
            ev0 = B.slab[slab_slot(B, base, 0u)];
            if (n0 > 1u) ev1 = B.slab[slab_slot(B, base, 1u)];
            if (A.out_head != nullptr) head0 = A.out_head[ev0.y];
        }
    }
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
    bool dense_item = false;   // an item of many events, left to k_expand_dense (a wave per item, a lane per event)
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
        const uint32_t hay = hay0, prev = prev0;
        if (prev != hay) {  // the haystack's first item: its prefix word, and those of the empty haystacks right before it
            for (uint32_t hh = prev + 1u; hh != hay + 1u; ++hh) prefix[hh] = rb[i];
        }
        const uint32_t n = n0;
        if (n == 0 || n > B.slab_events) continue;  // overflowed items are k_refill's
        if (rb[i + 1] > cap) continue;
        if (B.dense_list != nullptr && n > B.wave_min) { dense_item = true; continue; }
        uint64_t w = rb[i];
        // an item with many events (tens per item on text full of matches): event e + 2 and the output head of event e + 1 are
        // requested before event e is written -- one round trip per event instead of two dependent ones
        uint2 ev_a = ev0, hd_a = head0, ev_b = ev1;
        for (uint32_t e = 0; e < n; ++e) {
            uint2 ev_c = make_uint2(0u, 0u), hd_b = make_uint2(0u, 0u);
            if (e + 2u < n) ev_c = B.slab[slab_slot(B, g, e + 2u)];  // event-major, like the run lists
            if (e + 1u < n && A.out_head != nullptr) hd_b = A.out_head[ev_b.y];
            if ((hd_a.x >> 24) == 1u) out[w++] = make_uint4(hay, ev_a.x, ev_a.x - hd_a.y, hd_a.x & 0xFFFFFFu);
            else w = emit_event(A, out, w, hay, ev_a.x, ev_a.y);
            ev_a = ev_b;
            hd_a = hd_b;
            ev_b = ev_c;
        }
    }
    if (B.dense_list != nullptr) {   // (one atomic per wave that has such items; none for the usual sparse batch)
        const uint64_t dm = __ballot(dense_item);
        if (dm != 0) {
            uint32_t at = 0;
            if ((threadIdx.x & 63u) == 0) at = atomicAdd(&B.counters[23], uint32_t(__builtin_popcountll(dm)));
            at = uint32_t(__builtin_amdgcn_readfirstlane(int(at)));
            if (dense_item) B.dense_list[at + __builtin_amdgcn_mbcnt_hi(uint32_t(dm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(dm), 0u))] = uint32_t(base);
        }
    }
    __syncthreads();
    if (s_closing) {  // prefix words of the haystacks behind the last item (all of them when the batch has no item), the closing word
        const uint64_t total = s_total;
        const uint32_t from = items ? B.item_hay[items - 1] + 1u : 0u;
        for (uint64_t hh = uint64_t(from) + threadIdx.x; hh <= B.n_haystacks; hh += blockDim.x) prefix[hh] = total;
    }
}

// ------------------------------------------------------------------------------------------
// k_expand_dense: the items k_expand left in BatchArgs::dense_list -- more events than wave_min -- a wave per item, a lane per
// event.  k_expand gives an item one thread, and an event costs that thread a round trip (the event, its state's output head,
// the store): 128 events per item over 8,192 items of a 64 MiB batch were 215 us on eight workgroups, the longest kernel of
// the batch.  Here the events of an item are read 64 at a time, their record counts summed over the wave (one record per
// event as a rule: the head says so), and every lane writes its own event's records at its offset.  Enqueued behind
// k_expand while the batches before averaged several records per item.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand_dense(DevAutomaton A, BatchArgs B, const uint64_t *rec_base, uint4 *out, uint64_t cap) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_list = B.counters[23];
#pragma unroll 1
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_list; w += gridDim.x * 4u) {
        const uint64_t g = uint32_t(__builtin_amdgcn_readfirstlane(int(B.dense_list[w])));
        const uint32_t n = uint32_t(__builtin_amdgcn_readfirstlane(int(B.ev_count[g])));
        const uint32_t hay = uint32_t(__builtin_amdgcn_readfirstlane(int(B.item_hay[g])));
        const uint64_t w0 = rec_base[g];
        if (rec_base[g + 1] > cap) continue;   // (as in k_expand: the host grows the buffer and expands again)
        uint64_t run = w0;
#pragma unroll 1
        for (uint32_t c = 0; c < n; c += 64u) {
            const uint32_t e = c + lane;
            const bool active = e < n;
            const uint2 ev = active ? B.slab[slab_slot(B, g, e)] : make_uint2(0u, 0u);
            uint2 hd = make_uint2(0u, 0u);
            uint32_t cnt = 0;
            if (active) {
                if (A.out_head != nullptr) { hd = A.out_head[ev.y]; cnt = hd.x >> 24; }
                if (A.out_head == nullptr || cnt == 255u) cnt = A.out_count[ev.y];
            }
            const uint32_t incl = wave_inclusive_sum(cnt);
            const uint64_t at = run + (incl - cnt);
            if (active) {
                if ((hd.x >> 24) == 1u) out[at] = make_uint4(hay, ev.x, ev.x - hd.y, hd.x & 0xFFFFFFu);
                else emit_event(A, out, at, hay, ev.x, ev.y);
            }
            run += uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        }
    }
}

void launch_expand_dense(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, const uint64_t *rec_base, void *out, uint64_t cap, hipStream_t st) {
    if (max_items == 0 || B.dense_list == nullptr) return;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((max_items + 3u) / 4u, uint64_t(num_cus()) * 8u));
    hipLaunchKernelGGL(k_expand_dense, dim3(blocks), dim3(256), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap);
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
    if (scan && blocks >= 2048u)
        hipLaunchKernelGGL((k_expand<true, true>), dim3(blocks), dim3(kExpandThreads), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap, prefix,
                           host_counters, next_sync, next_sync_words, ticket, status, bad_offsets, next_isync, next_isync_words);
    else if (scan)
        hipLaunchKernelGGL(k_expand<true>, dim3(blocks), dim3(kExpandThreads), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap, prefix,
                           host_counters, next_sync, next_sync_words, ticket, status, bad_offsets, next_isync, next_isync_words);
    else
        hipLaunchKernelGGL(k_expand<false>, dim3(blocks), dim3(kExpandThreads), 0, st, A, B, rec_base, static_cast<uint4 *>(out), cap, prefix,
                           host_counters, next_sync, next_sync_words, ticket, status, bad_offsets, next_isync, next_isync_words);
}

}  // namespace acmi
