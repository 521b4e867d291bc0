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
// Work distribution: a work item (<= P x W bytes of one haystack) is walked by P pieces of W bytes, each entered from the root
// Lmax - 1 bytes early (k_walk_tiles' geometry); a WAVE takes 128 consecutive pieces at a time (a tile: whole items), two per lane,
// tiles dealt round robin -- no ticket, no barrier.  Events go to a list of the piece's own {index, count code, byte}: the first two
// in LDS, what a piece has beyond that (it cannot have more than one per byte) in the wave's region of a global buffer.
//
// Records are NOT written here.  A wave that places its records needs the number of records before them, and every way of getting
// that inside the kernel made the waves wait for one another: decoupled look-back per wave tile fell behind (131 tiles finish per
// microsecond, a window of 64 status words is an L2 round trip: 0.54 of 1.33 ms per GiB was look-back), per batch of sixteen
// tiles it still cost 0.45 ms (a batch waits for the slowest wave of the batch before and for four windows of words).  Instead
// the walk leaves, per tile, its events in the wave's LOG (a region of the wave's own: no atomic) and its items' record counts;
// k_expand's single-pass scan over the item counts -- the pipeline runs it anyway, for the per-haystack prefix -- gives every item
// its base, and k_chain_emit turns the logs into records at those bases.  The walk: 0.79 ms per GiB with nothing to wait for.
#include <cstdlib>
#include <type_traits>

#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kChainThreads = 1024;
constexpr uint32_t kChainWaves = kChainThreads / 64;

struct ChainArgs {
    uint32_t W, P, R, back;
    uint32_t inv_p;       // ceil(65536 / P): (s x inv_p) >> 16 = s / P for s < 128
    uint32_t debug;       // measurements only (ACMI_CHAIN_DEBUG): 2 no walk
    uint32_t spill_cap;   // events per piece the wave's overflow region holds
    uint32_t *spill;      // grid waves x spill_cap x 128 words
    uint32_t *log;        // grid waves x log_cap words: the waves' event logs
    uint32_t log_cap;     // words per wave
    uint2 *tile_info;     // per tile: {where its events begin in its wave's log, how many}
    uint32_t n_waves;     // waves of the walk's grid: tile t belongs to wave t % n_waves
    uint4 *out;           // (k_chain_emit)
    uint64_t cap;
    const uint64_t *rec_base;   // (k_chain_emit) per item: records before it (k_expand's scan)
};

struct ChainLds {
    const uint8_t *img;       // ALLD: rows at 0
    uint32_t row_bytes;
    uint32_t rec_at;          // !ALLD: LDS offset of the records, one per index (D states too)
    const uint8_t *cnt;       // min(records, 255) of the indices a row can lead to
    const uint16_t *cls;      // ALLD: byte -> 2 x class (TABLE mode), else null
};

// The LDS image of the walk (!ALLD), made from the automaton's image (automaton.hpp ChainImage) as the workgroup stages it:
//   [byte -> class, a byte each (TABLE mode: 256 B at offset 0)] [rows of the D states] [a record per INDEX] [count bytes]
// A record's low half is not the failure state's index but where that state's row lies -- its byte offset from the rows' start
// (ROW16: the rows end below 64 KiB) or half of it -- and a D state has a record too, naming its own row, with the key no class
// has.  A step is then the same five operations whatever the state: `row = record.lo + class` (one v_add_u32_sdwa: a halfword
// and a byte selected in the instruction), `hit = record.key == class` (one v_cmp_eq_u32_sdwa), and the three selects -- round 5
// looked up "D state?" (a compare), chose between the state and its record's failure index (a select), and multiplied by the row
// length: 13.5 VALU operations per byte where this has 8, in a kernel that VALU issue bounds.  The price: 4 bytes of
// LDS per D state (the headline set: 141.6 -> 157.3 KB), and rows that end below 128 KiB.
template <int MODE>
struct ChainLayout {
    static constexpr uint32_t rows_at = MODE == 0 ? 256u : 0u;
    uint32_t rec_at, cnt_at, total;
    bool row16, ok;
};
template <int MODE>
__host__ __device__ __forceinline__ ChainLayout<MODE> chain_layout(uint32_t n_dense, uint32_t n_index, uint32_t row_bytes, uint32_t cnt_n) {
    ChainLayout<MODE> Y;
    const uint32_t rows_end = n_dense * row_bytes;
    Y.rec_at = ChainLayout<MODE>::rows_at + ((rows_end + 15u) & ~15u);
    Y.cnt_at = Y.rec_at + 4u * n_index;
    Y.total = (Y.cnt_at + cnt_n + 15u) & ~15u;
    Y.row16 = rows_end <= 65536u;
    Y.ok = rows_end <= 131072u && Y.total <= 160u * 1024u;
    return Y;
}

template <int SEL>   // byte SEL of a dword of four classes; 4: the dword is the class
__device__ __forceinline__ uint32_t chain_class(uint32_t c4) { return SEL == 4 ? c4 : (c4 >> (8 * SEL)) & 0xffu; }

// One transition of two independent walks: their reads issued together (the records', then the rows'), so that one's LDS round
// trips lie under the other's.  `ca`, `cz`: the bytes' classes (ROW16: twice the class = the column's byte offset), byte SEL of
// them.  Through `ea`, `ez` the next indices.
// ALLD: `ua`, `uz` negative: the state ENTERED is final.  Else `ua`, `uz` are the records of the states the step LEAVES, negative
// when that state is final: a record says so of its own state (set as the image is staged), the rows hold plain indices, and an
// event is found one step late, when the state's record is read anyway -- no select of "record or row entry" for the flag and no
// mask for the index: 6 VALU operations per byte and piece (lshl_add, add_sdwa, cmp_sdwa, add, cndmask; the event's compare).
template <int SEL>
__device__ __forceinline__ void chain_step2(const ChainLds &L, uint32_t &ea, uint32_t &ez, uint32_t ca, uint32_t cz, uint32_t &ua, uint32_t &uz) {
    // ALLD: every state has its row -- one read, no record, no chain
    typedef const __attribute__((address_space(3))) int16_t *RowPtr;
    const uint32_t ka = chain_class<SEL>(ca), kz = chain_class<SEL>(cz);
    uint32_t aa, az;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(aa) : "v"(ea), "s"(L.row_bytes), "v"(ka));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(az) : "v"(ez), "s"(L.row_bytes), "v"(kz));
    ua = uint32_t(int32_t(*reinterpret_cast<RowPtr>(aa)));   // (sign-extended: a final target is negative)
    uz = uint32_t(int32_t(*reinterpret_cast<RowPtr>(az)));
    ea = ua & 0x7FFFu;
    ez = uz & 0x7FFFu;
}

// The same for an image with records (!ALLD), in two halves so that the events of the states the step leaves are stored while the
// rows' reads are under way: chain_read2 -> `ua`, `uz` the records of the states, `ta`, `tz` the row entries (requested);
// chain_next2 -> the next indices.  The operand selections are spelled out: hipcc found `record.lo + class byte` and `record.key ==
// class byte` as one SDWA operation each in a small test and in half of this kernel's instantiations, and shifts, masks and
// three-input logic in the rest -- and it sank the second piece's row read into a branch on "no hit" (an exec mask, a jump, the
// read issued only when the first had come back); as the asm's operand it is read whatever the keys say.
#define ACMI_SEL(n) ((n) == 0 ? "BYTE_0" : (n) == 1 ? "BYTE_1" : (n) == 2 ? "BYTE_2" : (n) == 3 ? "BYTE_3" : "DWORD")
template <int SEL>
__device__ __forceinline__ uint32_t chain_row_of(uint32_t rec, uint32_t c) {   // record.lo + class
    uint32_t r;
    if constexpr (SEL == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "=v"(r) : "v"(rec), "v"(c));
    else if constexpr (SEL == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_1" : "=v"(r) : "v"(rec), "v"(c));
    else if constexpr (SEL == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2" : "=v"(r) : "v"(rec), "v"(c));
    else if constexpr (SEL == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3" : "=v"(r) : "v"(rec), "v"(c));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(rec), "v"(c));
    return r;
}
// e1 = record.key == class ? e1 : row entry (a halfword; two wait states between an SDWA compare's VCC and its reader)
template <int SEL>
__device__ __forceinline__ void chain_select(uint32_t &e1, uint32_t t, uint32_t rec, uint32_t c) {
#define ACMI_SELECT(S) asm volatile("v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:BYTE_2 src1_sel:" S "\n\ts_nop 1\n\t" \
                                    "v_cndmask_b32_e32 %0, %1, %0, vcc" \
                                    : "+v"(e1) : "v"(t), "v"(rec), "v"(c) : "vcc")
    if constexpr (SEL == 0) ACMI_SELECT("BYTE_0");
    else if constexpr (SEL == 1) ACMI_SELECT("BYTE_1");
    else if constexpr (SEL == 2) ACMI_SELECT("BYTE_2");
    else if constexpr (SEL == 3) ACMI_SELECT("BYTE_3");
    else ACMI_SELECT("DWORD");
#undef ACMI_SELECT
}
#undef ACMI_SEL
template <bool ROW16, uint32_t ROWS_AT, int SEL>
__device__ __forceinline__ void chain_read2(const ChainLds &L, uint32_t ea, uint32_t ez, uint32_t ca, uint32_t cz, uint32_t &ua, uint32_t &uz, uint32_t &ta, uint32_t &tz) {
    typedef const __attribute__((address_space(3))) uint32_t *RecPtr;
    typedef const __attribute__((address_space(3))) uint16_t *RowPtr;
    ua = *reinterpret_cast<RecPtr>((ea << 2) + L.rec_at);
    uz = *reinterpret_cast<RecPtr>((ez << 2) + L.rec_at);
    __builtin_amdgcn_sched_barrier(0);   // (both records requested before either is waited for: the scheduler put the first's add in between)
    uint32_t aa = chain_row_of<SEL>(ua, ca), az = chain_row_of<SEL>(uz, cz);
    if constexpr (!ROW16) { aa <<= 1; az <<= 1; }
    ta = *reinterpret_cast<RowPtr>(aa + ROWS_AT);
    tz = *reinterpret_cast<RowPtr>(az + ROWS_AT);
}
template <int SEL>
__device__ __forceinline__ void chain_next2(uint32_t &ea, uint32_t &ez, uint32_t ca, uint32_t cz, uint32_t ua, uint32_t uz, uint32_t ta, uint32_t tz) {
    ++ea;
    ++ez;
    chain_select<SEL>(ea, ta, ua, ca);   // (0xFF -- no child, or a D state -- is no class)
    chain_select<SEL>(ez, tz, uz, cz);
}

// One piece of a wave's 128 (k_walk_tiles' geometry): the first block it walks, where in it the walk begins, what reports.
struct Piece {
    const uint8_t *pb;   // 16-byte aligned address of the first block it walks
    uint32_t nblk, start_j, rep_lo, rep_len, pos_bias, hay;
    uint32_t r, qi;      // item of the wave's tile, piece of the item
    bool have;
    uint64_t item;
};
__device__ __forceinline__ Piece chain_piece(const DevAutomaton &A, const BatchArgs &B, const ChainArgs &T, uint64_t total_items, uint64_t first_item, uint32_t s) {
    const uintptr_t beta = reinterpret_cast<uintptr_t>(B.blob) & 15u;
    const uint8_t *abase = B.blob - beta;   // 16-byte aligned
    Piece pc;
    pc.pb = A.pad16;
    pc.nblk = 0; pc.start_j = 16u; pc.rep_lo = pc.rep_len = pc.pos_bias = pc.hay = 0;
    pc.r = (s * T.inv_p) >> 16;   // s / P for s < 128 (a division by a run-time value is ~ 30 instructions)
    pc.qi = s - pc.r * T.P;
    pc.item = first_item + pc.r;
    pc.have = pc.r < T.R && pc.item < total_items;
    if (pc.have) {
        const uint4 rec = B.item_rec[pc.item];
        ItemDesc it;
        unpack_item(rec.x, rec.y, rec.z, rec.w, it);
        pc.hay = B.item_hay[pc.item];
        const uint64_t zb = it.begin + beta;
        const uint64_t G = zb & ~uint64_t(15);
        const uint32_t d0 = uint32_t(zb - G);
        const uint32_t a = pc.qi * T.W;
        const uint32_t lo = a > d0 ? a : d0;
        const uint32_t end = d0 + it.len;
        const uint32_t hi = (pc.qi + 1u == T.P) ? end : (a + T.W < end ? a + T.W : end);
        if (lo < hi) {
            const int64_t hs = int64_t(d0) - int64_t(it.pos0);   // the haystack's first byte
            int64_t xs = int64_t(lo) - int64_t(T.back);           // Lmax - 1 bytes early: exact from there on
            if (xs < hs) xs = hs;
            int64_t xb = xs & ~int64_t(15);
            if (xb == xs) xb -= 16;   // (a walk that begins ON a block boundary takes the block below as its first, virtually)
            pc.pb = abase + G + xb;
            pc.start_j = uint32_t(xs - xb);
            pc.rep_lo = uint32_t(int64_t(lo) - xb);
            pc.rep_len = hi - lo;
            pc.nblk = (pc.rep_lo + pc.rep_len + 15u) >> 4;
            pc.pos_bias = uint32_t(xb) - d0 + it.pos0 + 1u;
        }
    }
    return pc;
}

// records of the event of index `e` when the short forms saturated
__device__ __forceinline__ uint32_t chain_count(const DevAutomaton &A, uint32_t e) {
    const uint32_t c = A.chain_head[e].x >> 24;
    return c < 255u ? c : A.out_count[A.chain_orig[e]];
}

template <int MODE, bool ALLD, bool ROW16>
__global__ __launch_bounds__(kChainThreads) void k_walk_chain(DevAutomaton A, BatchArgs B, ChainArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    // (the image at offset 0 -- no static shared variable in front of it: chain_step2 reads at the offsets it computes)
    if (uint32_t(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)smem)) != 0u) __builtin_trap();
    ChainLds L;
    L.img = smem;
    L.row_bytes = A.n_columns * 2u;
    L.rec_at = 0;
    L.cls = nullptr;
    uint32_t cls_at = 0;   // LDS offset of the byte classes (TABLE mode)
    constexpr uint32_t kRowsAt = ALLD ? 0u : ChainLayout<MODE>::rows_at;
    if constexpr (ALLD) {
        // LDS: [image: rows | counts] [512 B byte classes (TABLE mode)]
        const uint32_t img_bytes = (A.chain_img_bytes + 15u) & ~15u;
        const uint4 *src = reinterpret_cast<const uint4 *>(A.chain_img);   // (the allocation has 16 bytes of slack)
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t t = threadIdx.x; t < (img_bytes >> 4); t += kChainThreads) dst[t] = src[t];
        if (MODE == 0 && threadIdx.x < 256) reinterpret_cast<uint16_t *>(smem + img_bytes)[threadIdx.x] = A.class_off[threadIdx.x];
        L.cnt = smem + A.chain_cnt_off;
        L.cls = (MODE == 0) ? reinterpret_cast<const uint16_t *>(smem + img_bytes) : nullptr;
        cls_at = img_bytes;
    } else {
        // (chain_layout; the records are translated from the image's {failure index | 2 x class | ...} as they are staged)
        const ChainLayout<MODE> Y = chain_layout<MODE>(A.chain_dense, A.chain_index, L.row_bytes, A.chain_cnt_n);
        const uint32_t rows_end = A.chain_dense * L.row_bytes;
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(A.chain_img);
            uint4 *dst = reinterpret_cast<uint4 *>(smem + kRowsAt);
            // (plain indices: whether a state is final its own record says)
            for (uint32_t t = threadIdx.x; t < ((rows_end + 15u) >> 4); t += kChainThreads) {
                const uint4 v = src[t];
                dst[t] = make_uint4(v.x & 0x7FFF7FFFu, v.y & 0x7FFF7FFFu, v.z & 0x7FFF7FFFu, v.w & 0x7FFF7FFFu);
            }
        }
        {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(A.chain_img + A.chain_rec_off);
            uint32_t *dst = reinterpret_cast<uint32_t *>(smem + Y.rec_at);
            for (uint32_t e = threadIdx.x; e < A.chain_index; e += kChainThreads) {
                uint32_t f = e, hi = 0x00FF0000u;   // a D state: its own row, the key no class has
                if (e >= A.chain_dense) {
                    const uint32_t r = src[e - A.chain_dense];
                    const uint32_t key2 = (r >> 16) & 0xFFu;
                    f = r & 0xFFFFu;
                    hi = (r & 0x1F000000u) | (key2 == 0xFFu || ROW16 ? key2 : key2 >> 1) << 16;   // (bits 24..28: the child's count code)
                }
                if (A.chain_frank[e] != 0xFFFFu) hi |= 0x80000000u;   // this state is final
                dst[e] = hi | (ROW16 ? f * L.row_bytes : (f * L.row_bytes) >> 1);
            }
        }
        {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(A.chain_img + A.chain_cnt_off);   // (4-byte aligned, like the image's end)
            uint32_t *dst = reinterpret_cast<uint32_t *>(smem + Y.cnt_at);
            for (uint32_t t = threadIdx.x; t < ((A.chain_cnt_n + 3u) >> 2); t += kChainThreads) dst[t] = src[t];
        }
        if (MODE == 0 && threadIdx.x < 256) smem[threadIdx.x] = uint8_t(ROW16 ? A.class_off[threadIdx.x] : A.class_off[threadIdx.x] >> 1);
        L.rec_at = Y.rec_at;
        L.cnt = smem + Y.cnt_at;
    }
    __syncthreads();
    // A lane walks TWO pieces at once, 2 lane and 2 lane + 1 of the wave's 128: a step is two dependent LDS reads, and with the
    // whole LDS taken by the image a CU runs one workgroup -- four waves per SIMD, too few to hide them (one piece per lane: 55 % of
    // the VALU cycles used).  The two walks are independent: their reads are issued together.  A lane's records -- the first
    // piece's, then the second's -- are one contiguous stretch of the output.
    // Event lists: entry k of piece h of this lane at byte (h x spill_cap + k) x 256 + 4 lane of the wave's region of a global buffer
    // (L2: written and read back within microseconds).  Not in LDS: two entries per piece there and the rest in memory meant a
    // compare and two store paths in every step with an event -- 8 VALU and 6 SALU operations per event and piece, which is what
    // the kernel spent most of its time on once a piece had an event every fifteen bytes; one path, its offset bumped, is 3.
    uint8_t *const spill = reinterpret_cast<uint8_t *>(T.spill) + uint64_t(__builtin_amdgcn_readfirstlane(int(blockIdx.x * kChainWaves + wave))) * T.spill_cap * 512u;
    const uint32_t off_a0 = lane * 4u, off_z0 = T.spill_cap * 256u + lane * 4u;
    uint32_t walked16 = 0;
    const uint32_t cnt_n = A.chain_cnt_n;
    const uint32_t gw = blockIdx.x * kChainWaves + wave;   // this wave among the grid's
    const uint64_t tiles_all = (total_items + T.R - 1u) / T.R;
    uint32_t *const log = T.log + uint64_t(gw) * T.log_cap;
    uint32_t log_at = 0;   // words of the wave's log in use
#pragma unroll 1
    for (uint64_t t = gw; t < tiles_all; t += T.n_waves) {
        const uint64_t first_item = t * T.R;
        const Piece pa = chain_piece(A, B, T, total_items, first_item, 2u * lane), pz = chain_piece(A, B, T, total_items, first_item, 2u * lane + 1u);
        walked16 += pa.nblk + pz.nblk;
        // ---- the walk: all lanes in step, a lane past its own bytes walks on without reporting ----
        uint32_t ea = 0, ez = 0;             // states of the two pieces
        uint32_t oa = off_a0, oz = off_z0;   // where their next events go
        const uint32_t my_blocks = pa.nblk > pz.nblk ? pa.nblk : pz.nblk;
        if (__any(my_blocks != 0u) && !(T.debug & 2u)) {
            // An event inside the walk is ONE store, {index, byte}, to the piece's list, and the list's offset bumped in place: two
            // VALU operations.  What it stands for is worked out behind the walk, where a lane runs over its own events only -- in
            // here the whole wave pays for every step in which ANY lane has one.  (Round 5 carried the parent record's count code
            // along: an and, a three-way or, and the bump through a copy -- four operations at 0.4-0.9 of the steps; the code is
            // in LDS where the drain can read it, records_of.)
            auto event = [&](uint32_t e, uint32_t &off, uint32_t jj) __attribute__((always_inline)) {
                *reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(spill) + off) = e | jj << 15;
                asm volatile("v_add_u32 %0, 0x100, %0" : "+v"(off));   // (in place: hipcc bumped a copy and moved it back)
            };
            // four steps of both pieces; the column offsets of a dword's four bytes are worked out together (SWAR3: a handful of
            // operations for all four; TABLE: four independent LDS reads).  CHECKED: the range test of a block that does not lie
            // inside every piece's own range; WARM: the first block, in which a walk begins (the state is the root before).
            // the column offsets of a dword's bytes: SWAR3 a handful of operations for all four; TABLE an LDS read per byte, requested a
            // byte ahead of its step (all eight of a dword pair up front cost the two registers that spilled into this loop)
            auto cls_of = [&](uint32_t wd, uint32_t i) __attribute__((always_inline)) -> uint32_t {
                // (explicitly LDS reads at the table's offset: through class_offset's generic pointer hipcc emitted flat_load_ushort
                // behind two branches -- a flat load counts on vmcnt too: every byte's class waited for the text requested four
                // blocks ahead and for the event stores)
                if constexpr (ALLD) {
                    const uint32_t b2 = i == 0u ? (wd & 0xffu) << 1 : (wd >> (8u * i - 1u)) & 0x1feu;
                    return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(cls_at + b2);
                } else {   // (a byte per byte value at LDS offset 0: the byte is the address)
                    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((wd >> (8u * i)) & 0xffu);
                }
            };
            // QUIET: a first block in which no walk of the wave reports and every walk begins at byte 1 (or not at all: such a lane
            // is put back to the root behind the block) -- no range test, no event test, no reset; SKIP0: its first dword, less byte 0
            auto dword = [&](uint32_t wa, uint32_t wz, uint32_t jj, auto checked, auto warm, auto quiet, auto skip0, auto first) __attribute__((always_inline)) {
                constexpr bool C = decltype(checked)::value, W0 = decltype(warm)::value, Q = decltype(quiet)::value, S0 = decltype(skip0)::value, F = decltype(first)::value;
                uint32_t c4a = 0, c4z = 0, na_c = 0, nz_c = 0;
                if constexpr (MODE == 1) {
                    c4a = swar_columns<uint16_t>(A, wa);   // (bytes of 2 x class: even, so that a shift leaves the classes)
                    c4z = swar_columns<uint16_t>(A, wz);
                    if constexpr (!ALLD && !ROW16) { c4a >>= 1; c4z >>= 1; }
                } else {
                    na_c = cls_of(wa, S0 ? 1u : 0u);
                    nz_c = cls_of(wz, S0 ? 1u : 0u);
                }
                auto one = [&](auto sel) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(sel)::value;
                    if constexpr (S0 && i == 0u) return;
                    const uint32_t j = jj + i;
                    uint32_t ua, uz, ca_i = c4a, cz_i = c4z;
                    constexpr int SEL = MODE == 1 ? int(i) : 4;
                    if constexpr (MODE != 1) {
                        ca_i = na_c; cz_i = nz_c;
                        if constexpr (i < 3u) { na_c = cls_of(wa, i + 1u); nz_c = cls_of(wz, i + 1u); }
                    }
                    if constexpr (ALLD) {
                        chain_step2<SEL>(L, ea, ez, ca_i, cz_i, ua, uz);
                        if (W0) { ea = j < pa.start_j ? 0u : ea; ez = j < pz.start_j ? 0u : ez; }
                        if constexpr (Q) return;
                        // (rep_lo >= start_j: nothing before a walk's first byte reports)
                        const bool va = int32_t(ua) < 0 && (!C || j - pa.rep_lo < pa.rep_len);
                        const bool vz = int32_t(uz) < 0 && (!C || j - pz.rep_lo < pz.rep_len);
                        if (va) event(ea, oa, j);
                        if (vz) event(ez, oz, j);
                    } else {
                        uint32_t ta, tz;
                        const uint32_t ea0 = ea, ez0 = ez;
                        chain_read2<ROW16, kRowsAt, SEL>(L, ea, ez, ca_i, cz_i, ua, uz, ta, tz);
                        chain_next2<SEL>(ea, ez, ca_i, cz_i, ua, uz, ta, tz);
                        if constexpr (!Q) {
                            // (the state the step LEFT, entered with byte j - 1: a block's first step looks back into the block before,
                            // which may lie in front of the piece's own range -- that one step is tested whatever the block.  Behind
                            // the selects: between the rows' reads and the selects, the branches around the stores cut the reads
                            // off from their zero-extension -- an and per read -- and invited hipcc to sink the second read)
                            constexpr bool T = C || (F && i == 0u);
                            const bool va = int32_t(ua) < 0 && (!T || j - 1u - pa.rep_lo < pa.rep_len);
                            const bool vz = int32_t(uz) < 0 && (!T || j - 1u - pz.rep_lo < pz.rep_len);
                            if (va) event(ea0, oa, j - 1u);
                            if (vz) event(ez0, oz, j - 1u);
                        }
                        if (W0) { ea = j < pa.start_j ? 0u : ea; ez = j < pz.start_j ? 0u : ez; }
                    }
                };
                one(std::integral_constant<uint32_t, 0u>{}); one(std::integral_constant<uint32_t, 1u>{});
                one(std::integral_constant<uint32_t, 2u>{}); one(std::integral_constant<uint32_t, 3u>{});
            };
            // (explicitly global loads: through a generic pointer hipcc emits flat_load, which counts on the LDS counter too -- every
            // wait for a state's record then also waited for the text requested four blocks ahead)
            auto block = [](const uint8_t *base, uint32_t i) __attribute__((always_inline)) -> uint4 {
                const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(reinterpret_cast<uintptr_t>(base) + 16u * i);
                return make_uint4(v.x, v.y, v.z, v.w);
            };
            const uint8_t *qa = pa.pb, *qz = pz.pb;
            const uint32_t la = pa.nblk ? pa.nblk - 1u : 0u, lz = pz.nblk ? pz.nblk - 1u : 0u;
            const uint4 fa = block(qa, pa.start_j < 16u ? 0u : (la ? 1u : 0u)), fz = block(qz, pz.start_j < 16u ? 0u : (lz ? 1u : 0u));   // (start_j == 16: block 0 is virtual)
            uint4 a0 = block(qa, 1u < la ? 1u : la), a1 = block(qa, 2u < la ? 2u : la), a2 = block(qa, 3u < la ? 3u : la), a3 = block(qa, 4u < la ? 4u : la);
            uint4 z0 = block(qz, 1u < lz ? 1u : lz), z1 = block(qz, 2u < lz ? 2u : lz), z2 = block(qz, 3u < lz ? 3u : lz), z3 = block(qz, 4u < lz ? 4u : lz);
            const uint32_t sj = pa.nblk ? pa.start_j : 16u, sk = pz.nblk ? pz.start_j : 16u;
            constexpr std::true_type yes{};
            constexpr std::false_type no{};
            // (the usual tile: Lmax - 1 <= 15 bytes of warm-up in front of pieces that begin on block boundaries -- fifteen steps from
            // the root for every lane, nothing to test: 26 -> 18 VALU operations per step of the pair in a sixth of the tile's steps)
            const bool silent = (!pa.nblk || pa.rep_lo >= 16u) && (!pz.nblk || pz.rep_lo >= 16u);   // nothing reports in block 0
            const bool plain = silent && (sj == 1u || sj >= 16u) && (sk == 1u || sk >= 16u);
            if (__all(plain)) {
                if (__any(sj < 16u || sk < 16u)) {
                    dword(fa.x, fz.x, 0u, no, no, yes, yes, yes); dword(fa.y, fz.y, 4u, no, no, yes, no, no);
                    dword(fa.z, fz.z, 8u, no, no, yes, no, no); dword(fa.w, fz.w, 12u, no, no, yes, no, no);
                    ea = sj >= 16u ? 0u : ea;
                    ez = sk >= 16u ? 0u : ez;
                }
            } else if (__all(silent)) {   // (shorter patterns: the walks begin further inside the block -- the reset per step, no tests)
                dword(fa.x, fz.x, 0u, no, yes, yes, no, yes); dword(fa.y, fz.y, 4u, no, yes, yes, no, no);
                dword(fa.z, fz.z, 8u, no, yes, yes, no, no); dword(fa.w, fz.w, 12u, no, yes, yes, no, no);
            } else {   // (some walk begins inside the block, or reports in it)
                dword(fa.x, fz.x, 0u, yes, yes, no, no, yes); dword(fa.y, fz.y, 4u, yes, yes, no, no, no);
                dword(fa.z, fz.z, 8u, yes, yes, no, no, no); dword(fa.w, fz.w, 12u, yes, yes, no, no, no);
            }
            uint32_t k = 1u;
#pragma unroll 1
            for (; __any(k < my_blocks); ++k) {   // (as many blocks as the wave's longest walk has: a ballot, not a reduction)
                asm volatile("; acmi-hot-begin");   // (tools/check_asm.py: no scratch traffic between the markers)
                const uint4 ca = a0, cz = z0;
                a0 = a1; a1 = a2; a2 = a3;
                z0 = z1; z1 = z2; z2 = z3;
                const uint32_t kn = k + 4u;
                if (kn < my_blocks) { a3 = block(qa, kn < la ? kn : la); z3 = block(qz, kn < lz ? kn : lz); }
                const uint32_t j0 = k * 16u;
                // (a block that lies inside EVERY piece's own range -- all but the first and the last of a piece -- without the range test)
                const bool full = j0 >= pa.rep_lo && j0 + 16u <= pa.rep_lo + pa.rep_len && j0 >= pz.rep_lo && j0 + 16u <= pz.rep_lo + pz.rep_len;
                if (__all(full)) {
                    dword(ca.x, cz.x, j0, no, no, no, no, yes); dword(ca.y, cz.y, j0 + 4u, no, no, no, no, no);
                    dword(ca.z, cz.z, j0 + 8u, no, no, no, no, no); dword(ca.w, cz.w, j0 + 12u, no, no, no, no, no);
                } else {
                    dword(ca.x, cz.x, j0, yes, no, no, no, yes); dword(ca.y, cz.y, j0 + 4u, yes, no, no, no, no);
                    dword(ca.z, cz.z, j0 + 8u, yes, no, no, no, no); dword(ca.w, cz.w, j0 + 12u, yes, no, no, no, no);
                }
                asm volatile("; acmi-hot-end");
            }
            if constexpr (!ALLD) {   // the states the walks end in: entered with the last byte walked, 16 k - 1
                const uint32_t jl = 16u * k - 1u;
                const uint32_t ua = lds_read<uint32_t>(smem, (ea << 2) + L.rec_at), uz = lds_read<uint32_t>(smem, (ez << 2) + L.rec_at);
                if (int32_t(ua) < 0 && jl - pa.rep_lo < pa.rep_len) event(ea, oa, jl);
                if (int32_t(uz) < 0 && jl - pz.rep_lo < pz.rep_len) event(ez, oz, jl);
            }
        }
        const uint32_t na = (oa - off_a0) >> 8, nz = (oz - off_z0) >> 8;   // events of the two pieces
        if (na | nz) {   // (this lane's own stores: landed before they are read back)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // records of an event: the five bits the PARENT's record has for its child (a C state's child is the next index: the
        // insides and ends of the patterns -- where a planted occurrence ends) unless they are saturated; else the count byte of
        // a state a row can lead to -- if it has one: a row may also lead INTO a chain (the child of a C state on the row's failure
        // path) -- else, and for saturated counts, the tables in global memory
        auto records_of = [&](uint32_t ent) __attribute__((always_inline)) -> uint32_t {
            const uint32_t idx = ent & 0x7FFFu;
            if constexpr (!ALLD) {
                const uint32_t r = lds_read<uint32_t>(smem, L.rec_at + 4u * (idx - 1u));   // (the root, index 0, is no event)
                const uint32_t code = (r >> 24) & 31u;
                if (((r >> 16) & 0xFFu) != 0xFFu && code != 31u) return code;   // (a D state's record, a chain's last: no child)
            }
            if (idx < cnt_n) {
                const uint32_t c = L.cnt[idx];
                if (c != 255u) return c;
            }
            return chain_count(A, idx);
        };
        // ---- behind the walk: every event once -- its records counted, the event itself into the wave's log, piece after piece:
        // {index, byte, piece}.  (Entries beyond the LDS list are the lane's own stores, read back from L2 -- the CU's L1 may still hold
        // the line as a tile before left it -- FOUR at a time: one after the other, each a round trip of its own, they were most of
        // the kernel's time on text with an event every nine bytes.)
        const uint32_t ne = na + nz;
        const uint32_t e_incl = wave_inclusive_sum(ne);
        const uint32_t e_tot = uint32_t(__builtin_amdgcn_readlane(int(e_incl), 63));
        if (lane == 0) T.tile_info[t] = make_uint2(log_at, e_tot);
        const bool fits = log_at + e_tot <= T.log_cap;
        // (the log is sized for the batches before: the host grows it and runs the batch again)
        if (!fits && lane == 0) atomicOr(&B.counters[19], 1u);
        uint32_t *dst = log + log_at + (e_incl - ne);
        // (the lane's own stores, read back from L2 -- the CU's L1 may still hold the line as a tile before left it --, four at a time)
        auto drain = [&](uint32_t h, uint32_t n, uint32_t off0, uint32_t *to) __attribute__((always_inline)) -> uint32_t {
            uint32_t recs = 0;
            const uint32_t tag = (2u * lane + h) << 24;   // (the count code has done its work when the entry is logged: its bits name the piece)
#pragma unroll 1
            for (uint32_t k = 0; k < n; k += 4u) {
                uint32_t e[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i)
                    e[i] = k + i < n ? __hip_atomic_load(reinterpret_cast<uint32_t *>(spill + off0 + (k + i) * 256u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    if (k + i < n) {
                        recs += records_of(e[i]);
                        if (fits) to[k + i] = (e[i] & 0x00FFFFFFu) | tag;
                    }
                }
            }
            return recs;
        };
        const uint32_t ra = drain(0u, na, off_a0, dst), rz = drain(1u, nz, off_z0, dst + na);   // records of the two pieces
        log_at += e_tot;
        // ---- records: exclusive prefix over the wave's lanes, the items' totals (k_expand's scan turns them into bases) ----
        const uint32_t n_rec = ra + rz;
        const uint32_t incl = wave_inclusive_sum(n_rec);
        const uint32_t excl = incl - n_rec;
        const uint32_t wave_tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        {
            // an item's total: the running sum behind its last piece less the sum before its first (pieces 2 l, 2 l + 1 of lane l)
            auto before = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t {   // records of the pieces before piece s (s < 128)
                const uint32_t l = s >> 1;
                const uint32_t ex = uint32_t(__shfl(int(excl), int(l), 64)), mid = uint32_t(__shfl(int(excl + ra), int(l), 64));
                return (s & 1u) ? mid : ex;
            };
            auto item_total = [&](const Piece &pc) __attribute__((always_inline)) {
                const uint32_t s0 = pc.r * T.P, s1 = s0 + T.P;   // pieces [s0, s1) (s1 <= 128 for an item the wave has)
                const uint32_t lo = before(s0 < 128u ? s0 : 127u);
                const uint32_t up = before(s1 < 128u ? s1 : 127u);   // (a shuffle: by ALL lanes, whatever they then keep)
                const uint32_t hi = s1 < 128u ? up : wave_tot;
                if (pc.have && pc.qi == 0u && hi != lo) B.rec_count[pc.item] = hi - lo;
            };
            item_total(pa);
            item_total(pz);
        }
    }
    // statistic: 16-byte blocks walked (what the automatic kernel choice reads)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) walked16 += __shfl_xor(walked16, d, 64);
    if (lane == 0 && walked16) atomicAdd(&B.counters[8u + ((blockIdx.x + wave) & 7u)], walked16);
}

// k_chain_emit: the logs into records.  A wave per tile; the tile's events lie in its wave's log in text order -- piece after piece --
// and its items are consecutive, so an event's records begin at rec_base[the tile's first item] (k_expand's scan of the counts the
// walk left) + the records of the tile's events before it: a running sum over the log, 64 events per round whatever their
// distribution over the pieces -- position ascending, an event's records longest pattern first (node.c:424-441) by construction,
// and an event of one record (most) is one of 64 consecutive 16-byte stores.  What an event needs of its piece (haystack, the
// position of its first byte) comes from the lane that would have walked it.
constexpr uint32_t kEmitThreads = 1024;
__global__ __launch_bounds__(kEmitThreads) void k_chain_emit(DevAutomaton A, BatchArgs B, ChainArgs T) {
    // LDS: every final index's output head, found through its rank -- 2 bytes per index + 8 per final state: a few thousand patterns
    // take 60-90 KiB, two workgroups a CU.  (From global memory an event's head was an L2 request of its own: 119 M of them a GiB
    // of prose under a word list, 0.4 of the L2's request rate, 1.0 ms for 1.9 GB of records.)
    // (T.debug != 0 here: the table does not fit -- a set with very many final states -- and the heads come from global memory)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bool in_lds = T.debug == 0u;
    const uint32_t rank_bytes = (A.chain_index * 2u + 15u) & ~15u;
    uint16_t *const frank = reinterpret_cast<uint16_t *>(smem);
    uint2 *const fhead = reinterpret_cast<uint2 *>(smem + rank_bytes);
    if (in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.chain_frank);   // (both allocations carry 16 bytes of slack)
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < (rank_bytes >> 4); i += kEmitThreads) dst[i] = src[i];
        for (uint32_t i = threadIdx.x; i < A.chain_finals; i += kEmitThreads) fhead[i] = A.chain_fhead[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    const uint64_t tiles_all = (total_items + T.R - 1u) / T.R;
    const uint64_t n_waves = uint64_t(gridDim.x) * (kEmitThreads / 64u);
#pragma unroll 1
    for (uint64_t t = uint64_t(blockIdx.x) * (kEmitThreads / 64u) + (threadIdx.x >> 6); t < tiles_all; t += n_waves) {
        const uint2 info = T.tile_info[t];
        if (info.y == 0u) continue;
        // (a tile whose events did not fit its wave's log wrote none of them: the host has the batch run again with a bigger log,
        // counters[19] -- nothing of this tile may be read here, its entries would be another tile's or lie behind the log's end)
        if (uint64_t(info.x) + info.y > T.log_cap) continue;
        const uint32_t *ev = T.log + (t % T.n_waves) * uint64_t(T.log_cap) + info.x;
        const uint64_t first_item = t * T.R;
        const Piece pa = chain_piece(A, B, T, total_items, first_item, 2u * lane), pz = chain_piece(A, B, T, total_items, first_item, 2u * lane + 1u);
        uint64_t at = T.rec_base[first_item];
        // four rounds of 64 events at a time: a round is three dependent round trips (the event, its state's output head, the store),
        // and only the running sum ties one round to the next -- the rounds' loads are issued together, kind by kind
        auto place = [&](uint32_t ent, uint2 hd, uint32_t c) __attribute__((always_inline)) {
            const uint32_t idx = ent & 0x7FFFu, piece = (ent >> 24) & 127u, owner = piece >> 1;
            const uint32_t ba = uint32_t(__shfl(int(pa.pos_bias), int(owner), 64)), bz = uint32_t(__shfl(int(pz.pos_bias), int(owner), 64));
            const uint32_t ha = uint32_t(__shfl(int(pa.hay), int(owner), 64)), hz = uint32_t(__shfl(int(pz.hay), int(owner), 64));
            const uint32_t pos = ((piece & 1u) ? bz : ba) + ((ent >> 15) & 511u), hay = (piece & 1u) ? hz : ha;
            const uint32_t incl = wave_inclusive_sum(c);
            const uint64_t mine = at + (incl - c);
            // (too small a buffer: what fits is written -- whole events, in order -- and the host grows it and has the logs emitted
            // again, or hands out the haystacks that are complete: "max_result_bytes")
            if (c != 0u && mine + c <= T.cap) {
                T.out[mine] = make_uint4(hay, pos, pos - hd.y, hd.x & 0xFFFFFFu);
                if (c > 1u) {
                    const uint32_t s = A.chain_orig[idx];
                    if (A.out_list != nullptr) {
                        const uint2 *ol = A.out_list + A.out_off[s];
                        for (uint32_t i = 1; i < c; ++i) {
                            const uint2 o = ol[i];
                            T.out[mine + i] = make_uint4(hay, pos, pos - o.y, o.x);
                        }
                    } else {
                        (void)emit_event(A, T.out, mine, hay, pos, s);
                    }
                }
            }
            at += uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        };
        // (the next four rounds' events are requested before this round's records are placed: under 2 TB/s of record stores a read
        // takes microseconds, and one per 256 events in the open was most of a tile's time)
        auto load4 = [&](uint32_t e0, uint32_t (&ent)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) ent[i] = e0 + 64u * i + lane < info.y ? ev[e0 + 64u * i + lane] : 0xFFFFFFFFu;
        };
        uint32_t nxt[4];
        load4(0u, nxt);
#pragma unroll 1
        for (uint32_t e0 = 0; e0 < info.y; e0 += 256u) {
            uint32_t ent[4], c[4];
            uint2 hd[4];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) ent[i] = nxt[i];
            if (e0 + 256u < info.y) load4(e0 + 256u, nxt);
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                hd[i] = make_uint2(0u, 0u);
                if (ent[i] != 0xFFFFFFFFu) {
                    // (two reads in their own address spaces: as one select of two pointers hipcc emits a flat load)
                    if (in_lds) {
                        const uint32_t rk = lds_read<uint16_t>(frank, (ent[i] & 0x7FFFu) * 2u);
                        hd[i].x = lds_read<uint32_t>(fhead, rk * 8u);
                        hd[i].y = lds_read<uint32_t>(fhead, rk * 8u + 4u);
                    } else {
                        hd[i] = A.chain_head[ent[i] & 0x7FFFu];
                    }
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                c[i] = hd[i].x >> 24;   // (no event: 0 -- an entry of all ones is none: piece 127 has no byte 511)
                if (c[i] == 255u) c[i] = A.out_count[A.chain_orig[ent[i] & 0x7FFFu]];
            }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i)
                if (e0 + 64u * i < info.y) place(ent[i], hd[i], c[i]);
        }
    }
}

// Does this automaton have the image, and does the kernel's LDS layout hold it?  (TABLE mode's layout is the bigger one by 256 bytes)
bool chain_walk_ok(const DevAutomaton &A) {
    if (A.chain_img == nullptr || A.chain_head == nullptr || A.chain_frank == nullptr) return false;
    if (A.chain_dense == A.chain_index) return ((A.chain_img_bytes + 15u) & ~15u) + 512u <= 160u * 1024u;
    return chain_layout<0>(A.chain_dense, A.chain_index, A.n_columns * 2u, A.chain_cnt_n).ok;
}

// ("reserve_cus": the walk leaves that many compute units to a collective's kernels, like the persistent scan kernels -- the logs
// and spill regions are per wave of the grid that runs, so the walk, the emit and the host's sizing take the same count)
uint32_t chain_grid(uint32_t reserve_cus) { return uint32_t(std::max(1, num_cus() - int(reserve_cus))); }
uint32_t chain_waves(uint32_t reserve_cus) { return chain_grid(reserve_cus) * kChainWaves; }
uint64_t chain_tiles(uint64_t max_items, uint32_t P) { return max_items / (128u / P) + 1u; }
// the waves' overflow regions: what a piece can have beyond its LDS list, a word each
// (an item's last piece takes the item's unaligned head as well: up to W + 15 bytes, an event each)
uint64_t chain_spill_words(uint32_t W, uint32_t reserve_cus) { return uint64_t(chain_waves(reserve_cus)) * uint64_t(W + 16u) * 128u; }

static ChainArgs chain_args(const DevAutomaton &A, const BatchArgs &B, uint32_t W, uint32_t P, const ChainBuffers &C) {
    ChainArgs T{};
    T.W = W;
    T.P = P;
    T.R = 128u / P;   // (a lane takes two pieces)
    T.inv_p = (65536u + P - 1u) / P;
    T.back = A.max_pattern_length ? A.max_pattern_length - 1u : 0u;
    T.spill_cap = W + 16u;
    T.spill = C.spill;
    T.log = C.log;
    T.log_cap = C.log_cap;
    T.tile_info = reinterpret_cast<uint2 *>(C.tile_info);
    T.n_waves = chain_waves(B.reserve_cus);
    return T;
}

hipError_t launch_walk_chain(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int class_mode, uint32_t W, uint32_t P, const ChainBuffers &C,
                             hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    if (!chain_walk_ok(A)) return hipErrorInvalidValue;
    ChainArgs T = chain_args(A, B, W, P, C);
    static const uint32_t debug = [] { const char *e = std::getenv("ACMI_CHAIN_DEBUG"); return e ? uint32_t(std::atoi(e)) : 0u; }();
    T.debug = debug;
    const bool alld = A.chain_dense == A.chain_index;
    const ChainLayout<0> Y0 = chain_layout<0>(A.chain_dense, A.chain_index, A.n_columns * 2u, A.chain_cnt_n);
    const ChainLayout<1> Y1 = chain_layout<1>(A.chain_dense, A.chain_index, A.n_columns * 2u, A.chain_cnt_n);
    const uint32_t lds = alld ? ((A.chain_img_bytes + 15u) & ~15u) + (class_mode == 0 ? 512u : 0u) : class_mode == 0 ? Y0.total : Y1.total;
    // (the logs are per wave of THIS grid -- k_chain_emit finds a tile's by t % n_waves)
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(chain_grid(B.reserve_cus)), dim3(kChainThreads), lds, st, A, B, T);
        return hipGetLastError();
    };
    if (alld) return class_mode == 1 ? go(k_walk_chain<1, true, true>) : go(k_walk_chain<0, true, true>);
    // ACMI_CHAIN_ROW16=0 (measurements): the halved row offsets where the whole ones fit
    static const bool no_row16 = [] { const char *e = std::getenv("ACMI_CHAIN_ROW16"); return e && e[0] == '0'; }();
    if (Y0.row16 && !no_row16) return class_mode == 1 ? go(k_walk_chain<1, false, true>) : go(k_walk_chain<0, false, true>);
    return class_mode == 1 ? go(k_walk_chain<1, false, false>) : go(k_walk_chain<0, false, false>);
}

hipError_t launch_chain_emit(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint32_t W, uint32_t P, const ChainBuffers &C,
                             const uint64_t *rec_base, void *out, uint64_t cap, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    ChainArgs T = chain_args(A, B, W, P, C);
    T.rec_base = rec_base;
    T.out = static_cast<uint4 *>(out);
    T.cap = cap;
    const uint64_t tiles = chain_tiles(max_items, P);
    uint32_t lds = ((A.chain_index * 2u + 15u) & ~15u) + A.chain_finals * 8u;
    if (lds > 160u * 1024u) { lds = 0; T.debug = 1u; }   // (a set with very many final states: the heads from global memory)
    const uint32_t per_cu = lds <= 80u * 1024u ? 2u : 1u;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((tiles + 15u) / 16u, uint64_t(num_cus()) * per_cu));
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_chain_emit), uint32_t(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_chain_emit, dim3(blocks), dim3(kEmitThreads), lds, st, A, B, T);
    return hipGetLastError();
}

}  // namespace acmi
