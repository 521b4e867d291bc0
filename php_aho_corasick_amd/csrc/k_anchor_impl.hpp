// k_anchor_impl.hpp -- k_scan_anchor, the q-gram anchor filter in all its forms (instantiated by k_anchor.hip: the aligned
// filter and its big-set form; k_anchor_dense.hip: the dense form -- two translation units that compile side by side)
#pragma once
#include "kdev.hpp"

namespace acmi {

// ------------------------------------------------------------------------------------------
// k_scan_anchor: sampled q-gram front end for pattern sets whose SHORTEST pattern is long.
//
// The per-byte table walk of k_scan_lds is bound by the LDS's random-lookup rate (one dependent
// ds_read per byte).  When every pattern is at least q + 3 bytes long (q = 8, 12 or 16) the walk
// can be skipped almost everywhere: an occurrence starting at text address x contains the
// 4-byte-aligned dword at ceil4(x), i.e. within its first 4 bytes, and the q bytes from there are
// one of <= 4 q-grams of that pattern (offsets 0..3).  So it suffices to test, for every ALIGNED
// text dword, whether the q bytes starting there are such an "anchor" q-gram: one hash + one
// ds_read_b64 (a 4-slot bucket of 16-bit fingerprints, 128 KiB table in LDS) per 4 bytes instead
// of four dependent ds_read_u16.  No false negatives by construction; a hit (true anchor or
// fingerprint collision) only flags the byte run in which a match could END, and k_walk_runs walks
// it through the Aho-Corasick table with the same exact walk k_scan_lds uses -- results are
// bit-identical.  Windows are independent (no state), so a wave can stream one item with fully
// coalesced non-temporal loads and their LDS reads pipeline freely; the kernel is bound by HBM.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kAnchorLdsBytes = 16384u * 8u;


// Probe one window: bucket = bits 3..16 of the hash (so the LDS byte address is one v_and),
// fingerprint = its high half.  The bucket's four 16-bit slots minus the fingerprint (packed
// subtract, the fingerprint broadcast by op_sel) are folded into a running packed minimum: some
// halfword of `z` is zero afterwards iff a slot equalled the fingerprint.
// The table sits at LDS address 0 (the kernel has no other LDS object and checks it), so the masked
// hash IS the ds_read address; going through the generic `smem` pointer costs a v_add per window.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2 *LdsU2Ptr;
typedef const __attribute__((address_space(3))) uint32_t *LdsU1Ptr;
__device__ __forceinline__ u32x2 anchor_bucket(uint32_t h) { return *reinterpret_cast<LdsU2Ptr>(h & 0x1FFF8u); }
__device__ __forceinline__ uint32_t anchor_compare(u32x2 b, uint32_t h, uint32_t z) {
    uint32_t x, y;
    asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x) : "v"(b.x), "v"(h));
    asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(y) : "v"(b.y), "v"(h));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(z) : "v"(z), "v"(x));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(z) : "v"(z), "v"(y));
    return z;
}
__device__ __forceinline__ bool has_zero_half(uint32_t z) { return ((z - 0x00010001u) & ~z & 0x80008000u) != 0; }

// One wave streams one work item at a time, 8 KiB per step, with fully coalesced loads (lane l holds
// quads r*64 + l).  Windows that begin in the previous quad get its last hashed dwords from the
// neighbouring lane through a DPP wave shift.  A wave owns 64 consecutive items; lane k remembers
// item k's run count, handed to k_walk_runs when the batch is done.
struct AnchorCursor {      // wave-uniform position in the wave's (item, chunk) task list
    uint32_t k;            // item index inside the wave, 0..63
    uint32_t c, chunks;    // current 8-KiB chunk of the item, number of chunks (>= 1)
    uint32_t n, head, nq;  // scan bytes, unaligned head bytes, whole 16-byte quads after the head
    const uint4 *qp;       // first quad
    bool valid;
};

// `rec` = lane k's packed record of item wave_first + k (z == 0: no such item)
template <uint32_t TQ>  // quads per task (a wave step): 512 = 8 KiB, or 256 for the BIG filter
__device__ __forceinline__ void anchor_open(const DevAutomaton &A, const BatchArgs &B, const uint4 &rec, uint32_t k,
                                            AnchorCursor &cur) {
    cur.k = k;
    cur.c = 0;
    cur.chunks = 1;
    cur.n = cur.head = cur.nq = 0;
    cur.qp = static_cast<const uint4 *>(A.anchor_table);
    cur.valid = false;
    if (k >= B.wave_items) return;
    ItemDesc it;
    const uint32_t z = __builtin_amdgcn_readlane(rec.z, k);
    if (z == 0) return;
    unpack_item(__builtin_amdgcn_readlane(rec.x, k), __builtin_amdgcn_readlane(rec.y, k), z, 0u, it);
    cur.valid = true;
    const uint8_t *p = B.blob + (it.begin - it.warm);
    cur.n = z;
    const uint32_t mis = uint32_t(-reinterpret_cast<uintptr_t>(p)) & 15u;
    cur.head = mis < cur.n ? mis : cur.n;
    // whole quads after the head; with aligned items (align_warm) also the last, partial one: its windows are tested like
    // any other (the bytes behind the item's end are another haystack's, or the rest of the blob's last aligned block),
    // what they yield past the end is clipped or rejected, and there is no tail left to flag wholesale
    cur.nq = (cur.n - cur.head + (B.align_warm && cur.head == 0u ? 15u : 0u)) >> 4;
    if (cur.nq) cur.qp = reinterpret_cast<const uint4 *>(p + cur.head);
    cur.chunks = cur.nq ? (cur.nq + TQ - 1u) / TQ : 1u;
}

// advance to the next (item, chunk) task of the wave; false when the wave's items are done
template <uint32_t TQ>
__device__ __forceinline__ bool anchor_next(const DevAutomaton &A, const BatchArgs &B, const uint4 &rec, AnchorCursor &cur) {
    if (!cur.valid) return false;
    if (++cur.c < cur.chunks) return true;
    anchor_open<TQ>(A, B, rec, cur.k + 1u, cur);
    return cur.valid;
}

// BIG (anchor mode 2, more q-grams than the LDS table holds): LDS carries a 2^20-bit map of the hashes'
// top 20 bits; a window whose bit is set (a fifth of them for 262,144 q-grams) is looked up in an
// 8-slot fingerprint table in global memory, which L2 serves.  QW = 1 (4-byte windows, shortest pattern
// 7..10 B) needs no hand-over between quads.
// SHORT (small-set kernel only): items may be shorter than a task -- the rows a task does not have are not hashed.
// A variant of its own because the test, one scalar branch per row, costs the 8-KiB-item kernel its schedule.
// TINY: a wave whose items are all at most one row (1 KiB, aligned: haystacks of a few hundred bytes) takes them G to a task,
// ROW r of the task = item t*G + r, instead of spending a whole step on each (512-byte haystacks: 8 x fewer steps).
// DENSE (QW == 1, small sets): pattern sets whose shortest pattern is 4..6 B (keyword lists) have no aligned dword inside
// every occurrence, so every pattern is anchored by its FIRST four bytes and the 4-byte window at EVERY byte offset is
// tested -- 16 windows per quad instead of 4, hashed raw (anchor_hash_dense: a shift and a multiply-add).  Hit windows are
// an order of magnitude more frequent than with aligned anchors (a keyword list: ~ 20 per 8 KiB of text), so they are
// turned into runs by VECTOR code -- a wave prefix sum over the lanes' hit counts gives every hit its slot in the item's
// run list -- not one by one through the scalar unit; runs keep their natural bounds (they may overlap: k_walk_runs joins
// them into clusters).
// HOT (dense form): many hit windows expected (a dozen or more per 8 KiB) -- which of a step's four windows hit is worked out
// for every step, branch-free (31 VALU), instead of a cheap any-hit test (21) with the exact one behind a branch (+ 35 when
// taken: with three hits per step that was most of the kernel's instructions).
// SL2 (dense form, few keys): the LDS table has two slots per bucket, 2^15 buckets -- one ds_read_b32 and one packed subtract
// per window.
template <int QW, bool BIG, bool SHORT, bool TINY = false, bool DENSE = false, bool HOT = false, bool SL2 = false>
__global__ __launch_bounds__(kLdsThreads) void k_scan_anchor(DevAutomaton A, BatchArgs B) {
    static_assert(!SL2 || DENSE, "two-slot buckets exist for the dense form only");
    static_assert(!DENSE || (QW == 1 && !BIG), "the dense form exists for 4-byte windows and the LDS table only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = (BIG || DENSE) ? 4 : 8;  // 16-byte quads per lane and task (dense form: four -- with eight its sixteen windows per quad spilled a dozen registers into the streaming loop)
    constexpr uint32_t TQ = G * 64u;  // quads per task: 8 KiB, BIG 4 KiB (two tasks in flight in the registers of one)
    constexpr int NC = QW > 1 ? QW - 1 : 1;  // carries between quads (none for QW == 1; arrays keep one element)
    if (uint64_t(blockIdx.x) * (B.persistent ? 1u : kLdsThreads / 64u) * B.wave_items >= B.item_base[B.n_haystacks]) return;
    {   // LDS: the anchor table (or bitmap), 128 KiB; behind it 2 KiB of run staging per wave
        const uint4 *src = BIG ? reinterpret_cast<const uint4 *>(A.anchor_bitmap) : static_cast<const uint4 *>(A.anchor_table);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t t = threadIdx.x; t < kAnchorLdsBytes / 16u; t += kLdsThreads) dst[t] = src[t];
    }
    __syncthreads();
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)smem)) != 0u) __builtin_trap();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mul = A.anchor_mul;
    const uint32_t q = QW * 4u, m_min = A.min_pattern_length, l_max = A.max_pattern_length;
    const uint32_t lo_bias = DENSE ? m_min - 1u : (A.anchor_choice ? q - 1u : m_min - 4u);
    const uint64_t total_items = B.item_base[B.n_haystacks];

    // Persistent mode: waves pull batches of items from counters[1] until none are left, wave_grab batches per atomic.
    uint32_t grab_next = 0, grab_left = 0, tiny_batches = 0;
    uint32_t nrun_max = 0;   // the longest window list this wave has seen (what the host sizes the next batches' lists by)
    for (;;) {
    // a wave takes `wave_items` (1..64) consecutive items per batch: 64 for big batches, fewer when
    // there are not enough items to give every wave of the chip 64 of them
    uint32_t batch_first = (blockIdx.x * (kLdsThreads / 64u) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * B.wave_items;
    if (B.persistent) {
        if (grab_left == 0) {
            if (lane == 0) grab_next = atomicAdd(&B.counters[1], B.wave_items * (B.wave_grab ? B.wave_grab : 1u));
            grab_next = __builtin_amdgcn_readfirstlane(grab_next);
            grab_left = B.wave_grab ? B.wave_grab : 1u;
        }
        batch_first = grab_next;
        grab_next += B.wave_items;
        --grab_left;
    }
    if (batch_first >= total_items) break;
    const uint64_t wave_first = batch_first;

    // lane k holds the packed record of item wave_first + k for the whole batch: opening an item is
    // three v_readlane and no memory access
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (lane < B.wave_items && wave_first + lane < total_items) rec = B.item_rec[wave_first + lane];
    // TINY: every item of this batch is one aligned row at most?  (wave-uniform; else the wave takes the usual path)
    bool tiny_on = false;
    uint32_t tcnt = 0, tscan = 0, tload = 0;  // items of the batch; task being scanned / being loaded (G x tpack items each)
    uint32_t tpack = 1, tshift = 6;           // items per row of a task, log2 of the lanes per item
    if constexpr (TINY) {
        const bool fits = rec.z == 0u || (((reinterpret_cast<uintptr_t>(B.blob) + rec.x) & 15u) == 0u && rec.z <= 1024u);
        tiny_on = B.align_warm == 1u && __all(fits);
        tcnt = uint32_t(__builtin_popcountll(__ballot(rec.z != 0u)));
        tiny_batches += tiny_on ? 1u : 0u;  // statistic ("tiny_waves"): which path the batch's waves took -- summed up at the kernel's end
        // items of a quarter / half row at most (haystacks of <= 256 / 512 B) are packed FOUR / TWO to a row: lane l of row r holds
        // quad l mod 16 (32) of item (8 t + r) * 4 (2) + l / 16 (32) -- else three lanes in four of a 256-byte item's row idle
        if (!BIG && !DENSE && tiny_on) {
            if (__all(rec.z <= 256u)) { tpack = 4u; tshift = 4u; }
            else if (__all(rec.z <= 512u)) { tpack = 2u; tshift = 5u; }
        }
    }
    AnchorCursor ld, sc;  // load cursor runs one task ahead of the scan cursor
    anchor_open<TQ>(A, B, rec, 0, ld);
    if (!ld.valid) break;
    sc = ld;
    // per wave, behind the table: 2 KiB of run staging -- BIG: [1 KiB probe queue | 256 B hit masks | 768 B run staging]
    uint8_t *wave_lds = smem + kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u;
    RunStage stage{reinterpret_cast<uint4 *>(wave_lds + (BIG ? 1280u : 0u)), 0u, BIG ? kDenseStageBig : kDenseStage, false};   // (8-byte entries: kdev.hpp, window lists)
    uint2 *const queue = reinterpret_cast<uint2 *>(wave_lds);           // BIG only
    uint32_t *const hitmask = reinterpret_cast<uint32_t *>(wave_lds + 1024u);
    if (BIG) hitmask[lane] = 0u;
    uint32_t nrun = 0;              // window-list entries of the item being scanned (wave-uniform; keeps counting past the list's capacity)
    uint32_t nrun_vec = 0;          // lane k: entry count of item wave_first + k
    // carry[n-1] = hash prefix over the LAST n dwords of a quad (n = 1..QW-1): what a window that
    // begins there and ends in the next quad continues from.  `tail` = the carries of the previous
    // wave-load (its lane 63 is the left neighbour of lane 0).
    uint32_t tail[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) tail[i] = 0;
    if (sc.head) {
        // (dense form: its run lists hold window starts only -- an item with an unaligned head, the first haystack of an unaligned
        // blob, is walked whole)
        if constexpr (DENSE) nrun = B.run_cap + 1u;
        else append_win_wave(B, wave_first, stage, sc.k, kWinSpecial | ((sc.head + l_max) < sc.n ? sc.head + l_max : sc.n), nrun, lane);
    }

    uint4 buf0[G], buf1[G];
    // Lane l holds quads r*64 + l (r = 0..7) of the wave's current 8 KiB: every wave-load covers 1 KiB
    // of contiguous memory (8 whole 128-byte lines).  A micro-benchmark of the access patterns alone
    // (tools/micro/stream_pattern.hip) gives 4.6 TB/s when each lane reads its own 128-byte line --
    // the line is split over 8 wave-instructions of 64 different lines each -- against 6.4 TB/s coalesced.
    auto load_quad = [&](const AnchorCursor &cur, int r) -> uint4 {
        const uint4 *base = cur.qp;
        uint32_t last = cur.nq ? cur.nq - 1u : 0u;
        uint32_t qi = cur.c * TQ + uint32_t(r) * 64u + lane;
        if constexpr (TINY) {
            if (tiny_on) {  // row r of task `tload` = item tload * G + r (a row that is not there repeats the last item)
                if (tpack == 1u) {
                    uint32_t k = tload * uint32_t(G) + uint32_t(r);
                    k = k < tcnt ? k : tcnt - 1u;
                    const uint32_t x = __builtin_amdgcn_readlane(rec.x, k), y = __builtin_amdgcn_readlane(rec.y, k);
                    base = reinterpret_cast<const uint4 *>(B.blob + (uint64_t(y & 0x3FFFu) << 32 | x));
                    last = ((__builtin_amdgcn_readlane(rec.z, k) + 15u) >> 4) - 1u;
                    qi = lane;
                } else {  // several items per row: every lane fetches its own item's record from the lane that holds it
                    uint32_t k = (tload * uint32_t(G) + uint32_t(r)) * tpack + (lane >> tshift);
                    k = k < tcnt ? k : tcnt - 1u;
                    const uint32_t x = uint32_t(__shfl(int(rec.x), int(k), 64)), y = uint32_t(__shfl(int(rec.y), int(k), 64));
                    base = reinterpret_cast<const uint4 *>(B.blob + (uint64_t(y & 0x3FFFu) << 32 | x));
                    last = ((uint32_t(__shfl(int(rec.z), int(k), 64)) + 15u) >> 4) - 1u;
                    qi = lane & ((1u << tshift) - 1u);
                }
            }
        }
        return stream_load(base + (qi < last ? qi : last));  // (one load instruction either way: only the address differs)
    };
    auto load_task = [&](uint4 (&buf)[G], const AnchorCursor &cur) {
#pragma unroll
        for (int r = 0; r < G; ++r) buf[r] = load_quad(cur, r);
    };
    auto scan_task = [&](uint4 (&buf)[G]) {
        uint32_t z[G];  // dense form, per quad: bit e set = the window ENDING at byte e of the quad hit (written on hits only)
#pragma unroll
        for (int r = 0; r < G; ++r) z[r] = 0;
        // aligned forms: four bits per quad -- bit w of nibble r set = the window ending at dword w of quad r hit -- in ONE register
        // (an array of G words until round 6: seven registers of the streaming loop, and eight v_readlane per hit where one does)
        uint32_t zpack = 0;
        uint32_t zany = 0;
        if (sc.c == 0) {  // first chunk of an item: nothing to the left of quad 0 (wave-uniform)
#pragma unroll
            for (int i = 0; i < NC; ++i) tail[i] = 0;
        }
        // hashes of the 4 windows ending in quad r
        auto hash_quad = [&](int r, uint32_t (&h4)[4]) {
            if constexpr (TINY && QW > 1) {
                if (tiny_on) {  // every row begins an item: nothing to the left of its quad 0
#pragma unroll
                    for (int i = 0; i < NC; ++i) tail[i] = 0;
                }
            }
            if constexpr (QW == 1) {  // lone dwords are hashed raw (automaton.hpp anchor_hash): a shift and a multiply-add
                h4[0] = anchor_mad(buf[r].x, mul, buf[r].x >> 13);
                h4[1] = anchor_mad(buf[r].y, mul, buf[r].y >> 13);
                h4[2] = anchor_mad(buf[r].z, mul, buf[r].z >> 13);
                h4[3] = anchor_mad(buf[r].w, mul, buf[r].w >> 13);
                return;
            }
            uint32_t f[4];
            f[0] = anchor_fold_dev(buf[r].x);
            f[1] = anchor_fold_dev(buf[r].y);
            f[2] = anchor_fold_dev(buf[r].z);
            f[3] = anchor_fold_dev(buf[r].w);
            uint32_t mine[NC], left[NC];
#pragma unroll
            for (int n = 1; n < QW; ++n) {  // prefix over this quad's last n dwords
                uint32_t h = f[4 - n];
#pragma unroll
                for (int i = 4 - n + 1; i < 4; ++i) h = anchor_mad(h, mul, f[i]);
                mine[n - 1] = h;
                // lane 0 <- lane 63 of the previous wave-load (rotate), lanes 1..63 <- lane l-1 of this one
                const uint32_t wrap = __builtin_amdgcn_mov_dpp(tail[n - 1], 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
                left[n - 1] = __builtin_amdgcn_update_dpp(wrap, h, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if constexpr (TINY) {  // packed rows: the lane that holds an item's first quad has nothing to its left
                    if (tpack > 1u && (lane & ((1u << tshift) - 1u)) == 0u) left[n - 1] = 0u;
                }
            }
#pragma unroll
            for (int n = 1; n < QW; ++n) tail[n - 1] = mine[n - 1];
#pragma unroll
            for (int w = 0; w < 4; ++w) {  // window ending at dword w: QW-1-w dwords of the left quad, then f[..w]
                const int need = QW - 1 - w;
                uint32_t h;
                int i0;
                if (need > 0) { h = left[need - 1]; i0 = 0; } else { h = f[w - (QW - 1)]; i0 = w - (QW - 1) + 1; }
#pragma unroll
                for (int i = i0; i <= w; ++i) h = anchor_mad(h, mul, f[i]);
                h4[w] = h;
            }
        };
        const uint32_t bucket_mask = A.anchor_bucket_mask;
        const uint4 *big_table = static_cast<const uint4 *>(A.anchor_table);
        if constexpr (BIG) {
            // Big sets.  The LDS bitmap passes a fifth of all windows (262,144 q-grams in 2^20 bits); each of those
            // needs its 16-byte fingerprint bucket from the table behind L2.  Probing where the window is found --
            // four predicated loads and compares per quad, 14 of 64 lanes active in each, the wave waiting for every
            // quad's round trip -- made the kernel both VALU-bound (52 wave-instructions per window) and
            // latency-bound (2.1 ms per 4 GiB).  Instead the "maybes" of a wave are COMPACTED: {hash, position} goes
            // into a 128-entry LDS ring (ballot + mbcnt give the slot), and whenever 64 are queued the wave probes
            // them as one dense round: one 16-byte load and one compare per LANE, the load consumed one round later.
            // A fingerprint match sets the window's bit in the owner lane's hit mask (ds_or); the masks are read
            // back when the task is done and flagged in text order exactly as in the small-set kernel.
            uint32_t qhead = 0, qcount = 0;   // ring state (wave-uniform)
            uint32_t ph = 0, ppos = 0xFFFFFFFFu;  // the probe round in flight: hash, position (0xFFFFFFFF: idle lane)
            uint4 ptb = make_uint4(0u, 0u, 0u, 0u);
            bool pending = false;             // wave-uniform
            // (two rounds in flight were tried: no faster -- 1.71 against 1.75 ms per 4 GiB -- and the 12 registers
            // they hold spilled inside the loop: the kernel is bound by the L2's request rate, 70 M probes per GiB)
            auto consume = [&]() {
                if (!pending) return;
                uint32_t zz = 0xFFFFFFFFu;
                const uint32_t fp = ph | 0x10000u;  // fingerprint = (h >> 16) | 1: never the empty 0
                zz = anchor_compare(u32x2{ptb.x, ptb.y}, fp, zz);
                zz = anchor_compare(u32x2{ptb.z, ptb.w}, fp, zz);
                const bool hit = ppos != 0xFFFFFFFFu && has_zero_half(zz);
                if (__builtin_expect(__any(hit), 0)) {
                    if (hit) atomicOr(&hitmask[ppos >> 5], 1u << (ppos & 31u));
                }
                pending = false;
            };
            auto probe_round = [&]() {  // the oldest (up to) 64 queued windows
                consume();
                const uint32_t nq_ = qcount < 64u ? qcount : 64u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                ppos = 0xFFFFFFFFu;
                ph = 0;
                if (lane < nq_) {
                    const uint2 e = queue[(qhead + lane) & (kProbeQueue - 1u)];
                    ph = e.x;
                    ppos = e.y;
                }
                ptb = big_table[(ph >> A.anchor_bucket_shift) & bucket_mask];  // (idle lanes read bucket 0: no branch around the load)
                qhead += nq_;
                qcount -= nq_;
                pending = true;
            };
            // quads of this task that exist (an item of 4 KiB fills half a task: the rest is skipped, wave-uniform)
            uint32_t task_quads = sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ;
            if constexpr (TINY) {
                if (tiny_on) task_quads = (tcnt - tscan * uint32_t(G) < uint32_t(G) ? tcnt - tscan * uint32_t(G) : uint32_t(G)) * 64u;
            }
            // The bitmap words of quad r+1 are requested before quad r's are tested (sched_barrier keeps hipcc from
            // sinking the reads to their use): a wave never sits on the LDS round trip of its own four reads.
            typedef __attribute__((address_space(3))) u32x2 *RingPtr;
            const uint32_t ring_base = kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u;  // 1-KiB aligned: slot address = base | offset
            uint32_t hh[2][4], ww[2][4];
            auto fetch_quad = [&](int r) {
                hash_quad(r, hh[r & 1]);
#pragma unroll
                for (int w = 0; w < 4; ++w) ww[r & 1][w] = *reinterpret_cast<LdsU1Ptr>((hh[r & 1][w] >> 15) & 0x1FFFCu);
            };
            fetch_quad(0);
#pragma unroll
            for (int r = 0; r < G; ++r) {
                if (uint32_t(r) * 64u >= task_quads) break;
                if (r + 1 < G) fetch_quad(r + 1);  // (of a quad past the item's end: hashed, never tested)
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t *h4 = hh[r & 1], *wd = ww[r & 1];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // (one bit-field extract feeds both the ballot and the predicated append: written as a shift and a
                    // mask hipcc computed the bit twice, 7 VALU instead of 3)
                    const bool maybe = __builtin_amdgcn_ubfe(wd[w], __builtin_amdgcn_ubfe(h4[w], 12u, 5u), 1u) != 0u;
                    const uint64_t mm = __ballot(maybe);
                    if (mm != 0) {
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(mm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mm), 0u));
                        if (maybe) {
                            const uint32_t slot = ((((qhead + qcount) << 3) + (at << 3)) & (kProbeQueue * 8u - 8u)) | ring_base;
                            *reinterpret_cast<RingPtr>(slot) = u32x2{h4[w], uint32_t(r * 4 + w) | (lane << 5)};
                        }
                        qcount += uint32_t(__builtin_popcountll(mm));
                        if (qcount >= 64u) probe_round();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            while (qcount != 0) probe_round();
            load_task(buf, ld);  // the next task (issued behind the last probe round, which is consumed without waiting for it)
            consume();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t m = hitmask[lane];
            if (__builtin_expect(__any(m != 0), 0)) {
                hitmask[lane] = 0u;
                zpack = m;
                zany = m;
            }
        } else if constexpr (DENSE) {
        // 16 windows per quad, four at a time: step t = 4 r + j takes dword j of row r -- the three windows that begin in
        // dword j-1 and end in dword j, and dword j itself -- i.e. the windows ENDING at bytes 4j .. 4j+3 of the quad (bit e
        // of z[r] = the window ending at byte e).  Same software pipeline as below: the four bucket reads of step t are in
        // flight while step t-1's are compared.  The dword to the left of a quad's first is the neighbouring lane's last
        // (DPP), lane 0's the last dword of lane 63 in the previous row (tail[0]: raw text here, not a hash prefix).
        uint32_t hh[2][4];
        u32x2 bb[2][4];   // (SL2: .x holds the bucket's two slots, .y a copy -- the compares below then see every slot twice)
        uint32_t rows = SHORT ? ((sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ) + 63u) >> 6 : uint32_t(G);
        if constexpr (TINY) {
            if (tiny_on) rows = tcnt - tscan * uint32_t(G) < uint32_t(G) ? tcnt - tscan * uint32_t(G) : uint32_t(G);
        }
        uint32_t left = 0;  // the dword in front of the current row's quad
#pragma unroll
        for (int t = 0; t <= 4 * G; ++t) {
            const int r = t >> 2, j = t & 3;
            if (t < 4 * G) {
                const bool live = !SHORT || uint32_t(r) < rows;
                if (j == 0 && live) {
                    if constexpr (TINY) {
                        if (tiny_on) tail[0] = 0;  // every row begins an item: nothing to the left of its quad 0
                    }
                    const uint32_t wrap = __builtin_amdgcn_mov_dpp(tail[0], 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
                    left = __builtin_amdgcn_update_dpp(wrap, buf[r].w, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    tail[0] = buf[r].w;
                }
                if (live) {
                    const uint32_t dj = j == 0 ? buf[r].x : j == 1 ? buf[r].y : j == 2 ? buf[r].z : buf[r].w;
                    const uint32_t dm = j == 0 ? left : j == 1 ? buf[r].x : j == 2 ? buf[r].y : buf[r].z;
                    uint32_t *h4 = hh[t & 1];
                    const uint32_t w1 = __builtin_amdgcn_alignbyte(dj, dm, 1), w2 = __builtin_amdgcn_alignbyte(dj, dm, 2), w3 = __builtin_amdgcn_alignbyte(dj, dm, 3);
                    h4[0] = anchor_mad(w1, mul, w1 >> 13);
                    h4[1] = anchor_mad(w2, mul, w2 >> 13);
                    h4[2] = anchor_mad(w3, mul, w3 >> 13);
                    h4[3] = anchor_mad(dj, mul, dj >> 13);
                }
                // the row is consumed: reload its registers with the same row of the task two steps ahead
                if (j == 3) buf[r] = load_quad(ld, r);
                if (live) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if constexpr (SL2) bb[t & 1][w].x = *reinterpret_cast<LdsU1Ptr>(hh[t & 1][w] & 0x1FFFCu);   // bucket = hash bits 2..16
                        else bb[t & 1][w] = anchor_bucket(hh[t & 1][w]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0 && (!SHORT || uint32_t((t - 1) >> 2) < rows)) {
                const u32x2 *b4 = bb[(t - 1) & 1];
                const uint32_t *h4 = hh[(t - 1) & 1];
                if constexpr (SL2) {
                    // two slots: the subtract's result is the whole test.  HOT: exact per window, every step; else a running
                    // minimum with the exact test behind a branch
                    if constexpr (HOT) {
                        // (written as plain compares hipcc emits two v_cmp_eq_u32_sdwa per window -- halfword operands, the verdict in a
                        // scalar register pair -- an s_or and a v_cndmask: 14 VALU + 4 SALU per step against 19 VALU for the packed
                        // subtract and zero-halfword test; filter 0.442 -> 0.431 ms per GiB of the keyword set)
                        uint32_t m = 0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const uint32_t fp = h4[w] >> 16;
                            const bool hit = ((b4[w].x & 0xffffu) == fp) | ((b4[w].x >> 16) == fp);
                            m |= hit ? (1u << w) : 0u;
                        }
                        z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                        zany |= m;
                    } else {
                        uint32_t x4[4], zz = 0xFFFFFFFFu;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x4[w]) : "v"(b4[w].x), "v"(h4[w]));
                            asm("v_pk_min_u16 %0, %1, %2" : "=v"(zz) : "v"(zz), "v"(x4[w]));
                        }
                        if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                            uint32_t m = 0;
#pragma unroll
                            for (int w = 0; w < 4; ++w) m |= has_zero_half(x4[w]) ? (1u << w) : 0u;
                            z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                            zany |= m;
                        }
                    }
                } else if constexpr (HOT) {
                    // per window: the bucket's four slots minus the fingerprint, their minimum per halfword pair, the
                    // zero-halfword test (bits 15 / 31), shifted to the window's own bit positions and merged
                    uint32_t dd = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t x, y, tm;
                        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(x) : "v"(b4[w].x), "v"(h4[w]));
                        asm("v_pk_sub_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(y) : "v"(b4[w].y), "v"(h4[w]));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(tm) : "v"(x), "v"(y));
                        const uint32_t d = (tm - 0x00010001u) & ~tm;
                        dd |= (d >> (3 - w)) & (0x80008000u >> (3 - w));   // window w at bits 12 + w and 28 + w
                    }
                    const uint32_t m = __builtin_amdgcn_ubfe(dd | (dd >> 16), 12u, 4u);   // bit w: window w hit
                    z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                    zany |= m;
                } else {
                uint32_t zz = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 4; ++w) zz = anchor_compare(b4[w], h4[w], zz);
                if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                    uint32_t m = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) m |= has_zero_half(anchor_compare(b4[w], h4[w], 0xFFFFFFFFu)) ? (1u << w) : 0u;
                    z[(t - 1) >> 2] |= m << (4 * ((t - 1) & 3));
                    zany |= m;
                }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
        // Software pipeline over the quads, pinned with sched_barriers: the four bucket reads of quad r
        // are in flight while quad r-1's buckets are compared, so a wave never sits on LDS latency
        // (hipcc on its own put an lgkmcnt(0) right behind every ds_read).
        uint32_t hh[2][4];
        u32x2 bb[2][4];   // the four fingerprint buckets of a quad
        // SHORT: rows of 64 quads this task has (an item of 4 KiB fills half a task); the others are only reloaded
        uint32_t rows = SHORT ? ((sc.nq - sc.c * TQ < TQ ? sc.nq - sc.c * TQ : TQ) + 63u) >> 6 : uint32_t(G);
        if constexpr (TINY) {
            if (tiny_on) {
                const uint32_t left_items = tcnt - tscan * uint32_t(G) * tpack;
                rows = (left_items + tpack - 1u) / tpack < uint32_t(G) ? (left_items + tpack - 1u) / tpack : uint32_t(G);
            }
        }
#pragma unroll
        for (int r = 0; r <= G; ++r) {
            if (r < G) {
                if (!SHORT || uint32_t(r) < rows) hash_quad(r, hh[r & 1]);
                // the quad is consumed: reload its registers with the same quad of the task two steps
                // ahead at once, so that the wave keeps close to 16 KiB of loads in flight
                buf[r] = load_quad(ld, r);
                if (!SHORT || uint32_t(r) < rows) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) bb[r & 1][w] = anchor_bucket(hh[r & 1][w]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r > 0 && (!SHORT || uint32_t(r - 1) < rows)) {
                const u32x2 *b4 = bb[(r - 1) & 1];
                const uint32_t *h4 = hh[(r - 1) & 1];
                uint32_t zz = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 4; ++w) zz = anchor_compare(b4[w], h4[w], zz);
                // Rare, wave-uniform: some lane's quad hit.  Which of its four windows is worked out here,
                // while the buckets are still in registers: k_walk_runs then knows the window of every run.
                if (__builtin_expect(__any(has_zero_half(zz)), 0)) {
                    uint32_t m = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) m |= has_zero_half(anchor_compare(b4[w], h4[w], 0xFFFFFFFFu)) ? (1u << w) : 0u;
                    zpack |= m << (4 * (r - 1));
                    zany |= m;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // Rare, wave-uniform: flag where a match anchored at a hit window can end.  A pattern whose first
        // aligned dword is the window starting at byte `start` begins in [start-3, start] and ends at byte
        // index [start + m_min - 4, start + Lmax).
        auto flag = [&](uint32_t r, uint32_t l, uint32_t wmask) {
            const uint32_t qi = sc.c * TQ + r * 64u + l;
            if (qi >= sc.nq) return;
            for (; wmask != 0; wmask &= wmask - 1) {
                const uint32_t w = uint32_t(__builtin_ctz(wmask));
                // the window ending at dword w of quad qi starts QW-1 dwords earlier
                const int32_t start = int32_t(sc.head + (qi << 4) + 4u * w) - int32_t(q - 4u);
                // a window that begins before the item's first byte was hashed over zeros, and nothing the item reports can be
                // anchored there (an occurrence that ends in the item's reporting range begins inside its warm-up)
                if (start < 0) continue;
                // a window at the item's very end or in the padding of its last 16-byte block: whatever is anchored there
                // ends behind the item
                if (uint32_t(start) + lo_bias >= sc.n) continue;
                append_win_wave(B, wave_first, stage, sc.k, uint32_t(start), nrun, lane);
            }
        };
        if constexpr (DENSE) {
            // Hits -> runs by vector code.  Per row (its quads ascend with the lane, so (row, lane, byte) order is text
            // order): a lane keeps the hit windows that can hold a pattern at all -- they begin inside the item and a pattern
            // of the shortest length would end inside it -- counts them, a wave prefix sum gives each its index in the
            // item's run list, and the lanes write their own entries into the LDS stage (flushed to the run lists in bulk).
            // Runs are not clamped against each other: [start + m_min - 1, start + Lmax) each, possibly overlapping.
            bool by_row = __any(zany != 0);
            bool tiny_rows_now = false;
            if constexpr (TINY) tiny_rows_now = tiny_on;
            if (by_row && !tiny_rows_now) {
                // One item per task (the usual case): the hits of all eight rows are placed with THREE wave prefix sums -- the
                // lanes' hit counts of the task, then the per-row counts packed four to a word (a task of at most 255 hits
                // cannot carry between the bytes) -- instead of one per row, and the masks for windows that stick out of the
                // item run for its first row and its last two quads only: per row this part cost as much as two of its four
                // probe steps.
                if (nrun > B.run_cap) return;   // the item is walked whole anyway
                const uint32_t qi0 = sc.c * TQ + lane, pos00 = sc.head + (qi0 << 4);   // row r: + 64 r quads, + 1024 r bytes
                uint32_t pk_lo = 0, pk_hi = 0;
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    uint32_t zr = qi0 + 64u * uint32_t(r) < sc.nq ? z[r] : 0u;
                    const uint32_t row_first = sc.c * TQ + 64u * uint32_t(r);   // (wave-uniform)
                    if (row_first == 0u || row_first + 65u >= sc.nq) {
                        const uint32_t pos0 = pos00 + 1024u * uint32_t(r);
                        if (pos0 < 3u) zr &= ~((1u << (3u - pos0)) - 1u);   // start = pos0 + e - 3 >= 0
                        const int32_t emax = int32_t(sc.n + 3u) - int32_t(m_min) - int32_t(pos0);   // start + m_min <= n
                        if (emax < 15) zr = emax < 0 ? 0u : (zr & ((2u << emax) - 1u));
                    }
                    z[r] = zr;
                    const uint32_t c = uint32_t(__builtin_popcount(zr));
                    if (r < 4) pk_lo |= c << (8 * r); else pk_hi |= c << (8 * (r - 4));
                }
                const uint32_t mine = ((pk_lo * 0x01010101u) >> 24) + ((pk_hi * 0x01010101u) >> 24);
                const uint32_t T = __builtin_amdgcn_readlane(wave_inclusive_sum(mine), 63);
                if (T == 0u) return;
                if (T <= 255u) {
                    if (nrun + T > B.run_cap) { nrun = B.run_cap + 1u; return; }   // more windows than the run list holds: the item is walked
                    if (stage.count + T > kDenseStage) dense_flush(B, wave_first, stage, lane);
                    const uint32_t in_lo = wave_inclusive_sum(pk_lo), in_hi = wave_inclusive_sum(pk_hi);
                    const uint32_t tot_lo = __builtin_amdgcn_readlane(in_lo, 63), tot_hi = __builtin_amdgcn_readlane(in_hi, 63);
                    uint2 *slots2 = reinterpret_cast<uint2 *>(stage.slots);
                    uint32_t rowbase = 0;   // hits of the rows before this one (wave-uniform)
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        const uint32_t sh = 8u * uint32_t(r & 3);
                        const uint32_t cnt = ((r < 4 ? pk_lo : pk_hi) >> sh) & 0xffu, incl = ((r < 4 ? in_lo : in_hi) >> sh) & 0xffu;
                        uint32_t at = stage.count + rowbase + incl - cnt, idx = nrun + rowbase + incl - cnt;
                        const uint32_t pos0 = pos00 + 1024u * uint32_t(r);
                        for (uint32_t m = z[r]; m != 0; m &= m - 1, ++at, ++idx)
                            slots2[at] = make_uint2(sc.k | (idx << 8), pos0 + uint32_t(__builtin_ctz(m)) - 3u);
                        rowbase += ((r < 4 ? tot_lo : tot_hi) >> sh) & 0xffu;
                    }
                    stage.count += T;
                    nrun += T;
                    return;
                }
                // (more than 255 hits in one task: row by row below, with the masks applied already)
            }
            if (by_row) {
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    if (__ballot(z[r] != 0) == 0) continue;
                    uint32_t k_item = sc.k, n_item = sc.n, head = sc.head, qi = sc.c * TQ + uint32_t(r) * 64u + lane, nq_item = sc.nq;
                    if constexpr (TINY) {
                        if (tiny_on) {  // row r is item tscan * G + r
                            k_item = tscan * uint32_t(G) + uint32_t(r);
                            if (k_item >= tcnt) continue;
                            n_item = __builtin_amdgcn_readlane(rec.z, k_item);
                            head = 0;
                            qi = lane;
                            nq_item = (n_item + 15u) >> 4;
                            nrun = 0;
                        }
                    }
                    // windows that cannot anchor anything: begin before the item's first byte, or too near its end
                    uint32_t zr = qi < nq_item ? z[r] : 0u;
                    const uint32_t pos0 = head + (qi << 4);            // scan coordinate of the quad's first byte
                    if (pos0 < 3u) zr &= ~((1u << (3u - pos0)) - 1u);   // start = pos0 + e - 3 >= 0
                    {
                        // start + m_min <= n_item  <=>  e <= n_item + 3 - m_min - pos0
                        const int32_t emax = int32_t(n_item + 3u) - int32_t(m_min) - int32_t(pos0);
                        if (emax < 15) zr = emax < 0 ? 0u : (zr & ((2u << emax) - 1u));
                    }
                    const uint32_t cnt = uint32_t(__builtin_popcount(zr));
                    const uint32_t incl = wave_inclusive_sum(cnt);
                    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                    if (total != 0) {
                        if (nrun <= B.run_cap && nrun + total <= B.run_cap && total <= kDenseStage) {
                            // (dense stage: 256 entries of 8 bytes {item in the wave | run index << 8, window start}, flushed to
                            // the items' run lists -- 4 bytes per hit, item-major: an item's hits are one contiguous piece)
                            uint2 *slots2 = reinterpret_cast<uint2 *>(stage.slots);
                            if (stage.count + total > kDenseStage) dense_flush(B, wave_first, stage, lane);
                            uint32_t at = stage.count + incl - cnt, idx = nrun + incl - cnt;
                            for (uint32_t m = zr; m != 0; m &= m - 1, ++at, ++idx)
                                slots2[at] = make_uint2(k_item | (idx << 8), pos0 + uint32_t(__builtin_ctz(m)) - 3u);
                            stage.count += total;
                            nrun += total;
                        } else {
                            nrun = B.run_cap + 1u;  // more windows than the run list holds: the whole item is walked
                        }
                    }
                    if constexpr (TINY) {
                        if (tiny_on) nrun_vec = lane == k_item ? nrun : nrun_vec;
                    }
                }
            }
            if constexpr (TINY) {
                if (tiny_on) return;
            }
            return;
        }
        if constexpr (TINY) {
            if (tiny_on) {  // row r is item tscan * G + r: its hits in lane order, its run count handed to its lane
                if (__any(zany != 0)) {
#pragma unroll
                    for (int r = 0; r < G; ++r) {
                        const uint32_t zrow = (zpack >> (4 * r)) & 15u;
                        uint64_t who = __ballot(zrow != 0);
                        if (who == 0) continue;
                        // the row's items in lane order (one, or tpack of them); an item's run count goes to its lane when its
                        // last hit has been flagged
                        uint32_t cur_k = 0xFFFFFFFFu;
                        while (who != 0) {
                            const uint32_t l = uint32_t(__builtin_ctzll(who));
                            who &= who - 1;
                            const uint32_t k = (tscan * uint32_t(G) + uint32_t(r)) * tpack + (l >> tshift);
                            if (k >= tcnt) continue;   // (a slot behind the batch's last item repeats that item: not its hits again)
                            if (k != cur_k) {
                                if (cur_k != 0xFFFFFFFFu) { nrun_vec = lane == cur_k ? nrun : nrun_vec; nrun_max = nrun > nrun_max ? nrun : nrun_max; }
                                anchor_open<TQ>(A, B, rec, k, sc);
                                nrun = 0;
                                cur_k = k;
                            }
                            flag(0u, l & ((1u << tshift) - 1u), __builtin_amdgcn_readlane(zrow, l));
                        }
                        if (cur_k != 0xFFFFFFFFu) { nrun_vec = lane == cur_k ? nrun : nrun_vec; nrun_max = nrun > nrun_max ? nrun : nrun_max; }
                    }
                }
                return;
            }
        }
        // Several lanes hit in one task (text with matches by the dozen per item): the task's windows are placed by VECTOR code, as
        // the dense form places its own -- every lane keeps the hit windows that can anchor something the item reports, three wave
        // prefix sums (the lanes' counts of the task, the per-row counts packed four to a word) give each its index in the item's
        // list, and the lanes write their own entries into the stage.  One hit at a time through the scalar unit -- the path below,
        // right for the usual task with one hit -- cost ~ 100 ns a hit: 0.1 ms of a 64 MiB batch with 128 occurrences per 8 KiB.
        // (the big-set kernel only.  In the small-set kernel's 128 registers the path spills: as a variant of that kernel for batches
        // known to be dense in hits it was measured on the headline set, round 6 -- 63 registers spilled, 114 scratch operations per
        // trip of the streaming loop, the scan 0.67-0.69 ms per GiB whatever the hits: better than the scalar path's 0.87 at 64
        // occurrences per 8 KiB, worse than its 0.37 at 16, and the whole-automaton walk takes such text at 0.97 ms all in.  Not kept.)
        auto flag_vector = [&]() -> bool {
            if constexpr (!BIG) return false;
            constexpr int NP = G / 2;   // two row counts to a word, sixteen bits each (a row of 64 quads has 256 windows)
            const uint32_t qi0 = sc.c * TQ + lane;
            uint32_t pk[NP], zkept = 0;   // (the windows that can anchor something the item reports, nibble r = quad r)
#pragma unroll
            for (int i = 0; i < NP; ++i) pk[i] = 0;
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const uint32_t qi = qi0 + 64u * uint32_t(r);
                uint32_t zr = qi < sc.nq ? (zpack >> (4 * r)) & 15u : 0u;
                if (zr != 0u) {
                    const int32_t s0 = int32_t(sc.head + (qi << 4)) - int32_t(q - 4u);   // the window ending at dword w starts at s0 + 4 w
                    uint32_t keep = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int32_t st = s0 + 4 * w;
                        if (((zr >> w) & 1u) != 0u && st >= 0 && uint32_t(st) + lo_bias < sc.n) keep |= 1u << w;
                    }
                    zr = keep;
                }
                zkept |= zr << (4 * r);
                pk[r >> 1] |= uint32_t(__builtin_popcount(zr)) << (16 * (r & 1));
            }
            uint32_t in[NP], tot[NP], T = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                in[i] = wave_inclusive_sum(pk[i]);
                tot[i] = __builtin_amdgcn_readlane(in[i], 63);
                T += (tot[i] & 0xffffu) + (tot[i] >> 16);
            }
            if (T == 0u) return true;
            if (nrun + T > B.run_cap) { nrun += T; return true; }   // more windows than the list holds: counted, the item is walked whole
            // few: through the stage, like the single hits; many (more than the stage has room for): straight into the item's list --
            // consecutive entries, and item-major lists (long lists are) make the lanes' stores contiguous
            const bool direct = T > stage.cap;
            stage.flushed = stage.flushed || direct;
            if (!direct && stage.count + T > stage.cap) dense_flush(B, wave_first, stage, lane);
            uint2 *slots2 = reinterpret_cast<uint2 *>(stage.slots);
            uint32_t *runs32 = reinterpret_cast<uint32_t *>(B.runs);
            uint32_t rowbase = 0;   // hits of the rows before this one (wave-uniform)
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const uint32_t sh = 16u * uint32_t(r & 1);
                const uint32_t cnt = (pk[r >> 1] >> sh) & 0xffffu, incl = (in[r >> 1] >> sh) & 0xffffu;
                uint32_t at = stage.count + rowbase + incl - cnt, idx = nrun + rowbase + incl - cnt;
                const uint32_t s0 = sc.head + ((qi0 + 64u * uint32_t(r)) << 4) - (q - 4u);
                for (uint32_t m = (zkept >> (4 * r)) & 15u; m != 0; m &= m - 1, ++at, ++idx) {
                    const uint32_t start = s0 + 4u * uint32_t(__builtin_ctz(m));
                    if (direct) runs32[win_slot(B, wave_first + sc.k, idx)] = start;
                    else slots2[at] = make_uint2(sc.k | (idx << 8), start);
                }
                rowbase += (tot[r >> 1] >> sh) & 0xffffu;
            }
            if (!direct) stage.count += T;
            nrun += T;
            return true;
        };
        const uint64_t cand = __ballot(zany != 0);
        if (__builtin_expect(cand != 0, 0)) {
            if ((cand & (cand - 1)) != 0 && flag_vector()) {
                // (placed above)
            } else if ((cand & (cand - 1)) == 0) {
                // one lane hit (the usual case): its eight masks go through the scalar unit
                const uint32_t l = uint32_t(__builtin_ctzll(cand));
                // (its windows one by one off the packed word: a turn per hit, not a test per quad)
                for (uint32_t zp = __builtin_amdgcn_readlane(zpack, l); zp != 0; zp &= zp - 1) {
                    const uint32_t b = uint32_t(__builtin_ctz(zp));
                    flag(b >> 2, l, 1u << (b & 3u));
                }
            } else {
                uint32_t hits = 0;  // bit r: some window ENDING in this lane's quad r hit
#pragma unroll
                for (int r = 0; r < G; ++r) hits |= ((zpack >> (4 * r)) & 15u) ? (1u << r) : 0u;
                uint32_t rmask = 0;
                for (uint64_t w = cand; w != 0; w &= w - 1) rmask |= __builtin_amdgcn_readlane(hits, uint32_t(__builtin_ctzll(w)));
                while (rmask != 0) {  // ascending quad order (r outer, lane inner) keeps the window list sorted
                    const uint32_t r = uint32_t(__builtin_ctz(rmask));
                    rmask &= rmask - 1;
                    const uint32_t zr = (zpack >> (4u * r)) & 15u;
                    // A row in which every hitting lane has ONE window that can anchor something (the usual row of text with matches
                    // by the dozen per item): the lanes place their entries themselves, a ballot and a bit count for the index --
                    // ~ 15 instructions a row instead of ~ 80 a hit through the scalar unit (the headline set at 16 / 64 occurrences
                    // per 8 KiB: the hits were 0.2 / 0.7 of the scan's 0.37 / 0.87 ms per GiB).
                    const uint32_t qi = sc.c * TQ + r * 64u + lane;
                    const int32_t s0 = int32_t(sc.head + (qi << 4)) - int32_t(q - 4u);
                    uint32_t keep = 0;
                    if (zr != 0u && qi < sc.nq) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const int32_t st = s0 + 4 * w;
                            if (((zr >> w) & 1u) != 0u && st >= 0 && uint32_t(st) + lo_bias < sc.n) keep |= 1u << w;
                        }
                    }
                    uint64_t who = __ballot(keep != 0u);
                    if (who == 0) continue;
                    const uint32_t cnt = uint32_t(__builtin_popcountll(who));
                    if (__ballot((keep & (keep - 1u)) != 0u) == 0 && nrun + cnt <= B.run_cap) {
                        if (stage.count + cnt > stage.cap) dense_flush(B, wave_first, stage, lane);
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(who >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(who), 0u));
                        if (keep != 0u)
                            reinterpret_cast<uint2 *>(stage.slots)[stage.count + at] = make_uint2(sc.k | ((nrun + at) << 8), uint32_t(s0) + 4u * uint32_t(__builtin_ctz(keep)));
                        stage.count += cnt;
                        nrun += cnt;
                        continue;
                    }
                    // (a lane with two windows in the row, or a list about to overflow: one by one)
                    while (who != 0) {
                        const uint32_t l = uint32_t(__builtin_ctzll(who));
                        flag(r, l, __builtin_amdgcn_readlane(keep, l));
                        who &= who - 1;
                    }
                }
            }
        }
    };
    auto close_or_advance = [&]() -> bool {  // scan cursor: next chunk, or finish the item and open the next
        if (sc.c + 1u == sc.chunks) {
            // anchors the main loop did not test start after head + 16*nq - q: what they could match
            // ends at or after head + 16*nq - q + m_min -- a tail entry, walked (only items with an unaligned head have one)
            uint32_t lo = sc.head + (sc.nq << 4) + (A.anchor_choice ? q : m_min);
            lo = lo > q ? lo - q : 0u;
            if (!DENSE && sc.head + (sc.nq << 4) < sc.n && lo < sc.n)
                append_win_wave(B, wave_first, stage, sc.k, kWinSpecial | kWinTail | lo, nrun, lane);
            nrun_max = nrun > nrun_max ? nrun : nrun_max;
            nrun_vec = lane == sc.k ? nrun : nrun_vec;
        }
        const uint32_t k_before = sc.k;
        const bool more = anchor_next<TQ>(A, B, rec, sc);
        if (more && sc.k != k_before) {
            nrun = 0;
            if (sc.head) {
        // (dense form: its run lists hold window starts only -- an item with an unaligned head, the first haystack of an unaligned
        // blob, is walked whole)
        if constexpr (DENSE) nrun = B.run_cap + 1u;
        else append_win_wave(B, wave_first, stage, sc.k, kWinSpecial | ((sc.head + l_max) < sc.n ? sc.head + l_max : sc.n), nrun, lane);
    }
        }
        return more;
    };

    // Two tasks are in flight: buf0 / buf1 hold tasks t and t+1, and each quad's registers are reloaded
    // with task t+2 as soon as the quad has been hashed (scan_task).  Loads are unconditional -- past
    // the last task the cursor is an invalid item whose loads fall on the anchor table -- because a load
    // under a branch makes hipcc drain vmcnt(0) instead of counting.
    // (BIG: tasks are 4 KiB, so the two buffers take the 32 registers one 8-KiB task would.  The version with a
    // single buffer loaded at the end of each scan left every task's HBM latency exposed at its first quad; now the
    // loads of task t+1 are a whole scan old when it begins, and the probe loads -- whose in-order vmcnt wait also
    // drains whatever was issued before them -- come two thirds into a task at a 10 % pass rate.)
    bool done_tiny = false;
    if constexpr (TINY) {
        if (tiny_on) {
            tload = 0;
            load_task(buf0, ld);
            tload = 1;
            load_task(buf1, ld);
            for (;;) {
                tload = tscan + 2u;
                scan_task(buf0);
                if (++tscan * uint32_t(G) * tpack >= tcnt) break;
                tload = tscan + 2u;
                scan_task(buf1);
                if (++tscan * uint32_t(G) * tpack >= tcnt) break;
            }
            done_tiny = true;
        }
    }
    if (!done_tiny) {
    load_task(buf0, ld);
    anchor_next<TQ>(A, B, rec, ld);
    load_task(buf1, ld);
    for (;;) {
        anchor_next<TQ>(A, B, rec, ld);
        scan_task(buf0);
        if (!close_or_advance()) break;
        anchor_next<TQ>(A, B, rec, ld);
        scan_task(buf1);
        if (!close_or_advance()) break;
    }
    }

    // (Round 6 tried deciding the items that have ONE candidate window here, a lane each behind every wave batch -- the entries are
    // still in the stage, the streaming loop's registers free -- so that k_walk_runs would find nothing to do on a sparse batch:
    // A/B in one run, profiles/r06/ab_scan_verify.txt: k_walk_runs 50.7 -> 32.8 us, this kernel 719 -> 750 us, the step 0.805 ->
    // 0.816 ms.  A wave that waits for its candidates' five round trips does not stream, and its SIMD's other three waves do not
    // make up for it.  Taken out.)
    // hand the window lists over to the kernels behind (lane k: item wave_first + k)
    dense_flush(B, wave_first, stage, lane);
    if (rec.z != 0) B.run_count[wave_first + lane] = nrun_vec;
    if (!B.persistent) break;
    }  // next batch of this wave
    if constexpr (!DENSE) {
        // the longest list a wave saw, for the host (one atomic per wave and launch, and none for the usual sparse batch)
        if (lane == 0 && nrun_max > kWaveRunsMin) atomicMax(&B.counters[21], nrun_max);
    }
    if constexpr (TINY) {
        // one device atomic per WORKGROUP (through the waves' own LDS regions, free by now): one per batch on a single word
        // was the longest thing in the launch (see wave_grab)
        uint32_t *mine = reinterpret_cast<uint32_t *>(smem + kAnchorLdsBytes + (threadIdx.x >> 6) * 2048u);
        if (lane == 0) *mine = tiny_batches;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (uint32_t w = 0; w < kLdsThreads / 64u; ++w) sum += *reinterpret_cast<const uint32_t *>(smem + kAnchorLdsBytes + w * 2048u);
            if (sum) atomicAdd(&B.counters[18], sum);
        }
    }
}

}  // namespace acmi
