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
// (Round 6: in LDS a record names its failure state's ROW, every index has a record and the record says whether its own state is
// final -- chain_layout, chain_read / chain_next below: 6 VALU operations per byte and piece where round 5 had 13.5; what bounds the
// kernel since is the LDS itself, 64 random reads per instruction.)
//
// Work distribution: a work item (<= P x W bytes of one haystack) is walked by P pieces of W bytes, each entered from the root
// Lmax - 1 bytes early (k_walk_tiles' geometry; W = 128 for batches of 64 MiB and more); a WAVE takes 128 consecutive pieces at a
// time (a tile: whole items), two per lane, tiles dealt round robin -- no ticket, no barrier.  Events go to a list of the piece's
// own {index, byte} in the wave's region of a global buffer (a piece cannot have more than one per byte).
//
// Records are NOT written here.  A wave that places its records needs the number of records before them, and every way of getting
// that inside the kernel made the waves wait for one another: decoupled look-back per wave tile fell behind (131 tiles finish per
// microsecond, a window of 64 status words is an L2 round trip: 0.54 of 1.33 ms per GiB was look-back), per batch of sixteen
// tiles it still cost 0.45 ms (a batch waits for the slowest wave of the batch before and for four windows of words).  Instead
// the walk leaves, per tile, its events in the wave's LOG (a region of the wave's own: no atomic) and its items' record counts;
// k_expand's single-pass scan over the item counts -- the pipeline runs it anyway, for the per-haystack prefix -- gives every item
// its base, and k_chain_emit turns the logs into records at those bases.  The walk: 0.79 ms per GiB with nothing to wait for
// (round 6: 0.68).
#include <cstdlib>
#include <type_traits>

#include "kdev.hpp"

namespace acmi {

constexpr uint32_t kChainThreads = 1024;
constexpr uint32_t kChainWaves = kChainThreads / 64;

struct ChainArgs {
    uint32_t W, P, R, back;
    uint32_t inv_p;       // ceil(65536 / P): (s x inv_p) >> 16 = s / P for s < 192
    uint32_t np;          // pieces a lane walks at once: a wave's tile is 64 np pieces (chain_np())
    uint32_t debug;       // measurements only (ACMI_CHAIN_DEBUG): 2 no walk
    uint32_t spill_cap;   // events per piece the wave's overflow region holds
    uint32_t *spill;      // grid waves x spill_cap x 64 np words
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
template <uint32_t NP, int SEL>
__device__ __forceinline__ void chain_step_rows(const ChainLds &L, uint32_t (&e)[NP], const uint32_t (&c)[NP], uint32_t (&u)[NP]) {
    // ALLD: every state has its row -- one read, no record, no chain
    typedef const __attribute__((address_space(3))) int16_t *RowPtr;
    uint32_t aa[NP];
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) {
        const uint32_t k = chain_class<SEL>(c[h]);
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(aa[h]) : "v"(e[h]), "s"(L.row_bytes), "v"(k));
    }
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) u[h] = uint32_t(int32_t(*reinterpret_cast<RowPtr>(aa[h])));   // (sign-extended: a final target is negative)
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) e[h] = u[h] & 0x7FFFu;
}

// The same for an image with records (!ALLD), in two halves so that the events of the states the step leaves are stored while the
// rows' reads are under way: chain_read -> `u` the records of the states, `t` the row entries (requested); chain_next -> the next
// indices.  The operand selections are spelled out: hipcc found `record.lo + class byte` and `record.key ==
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
template <uint32_t NP, bool ROW16, uint32_t ROWS_AT, int SEL>
__device__ __forceinline__ void chain_read(const ChainLds &L, const uint32_t (&e)[NP], const uint32_t (&c)[NP], uint32_t (&u)[NP], uint32_t (&t)[NP]) {
    typedef const __attribute__((address_space(3))) uint32_t *RecPtr;
    typedef const __attribute__((address_space(3))) uint16_t *RowPtr;
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) u[h] = *reinterpret_cast<RecPtr>((e[h] << 2) + L.rec_at);
    __builtin_amdgcn_sched_barrier(0);   // (every piece's record requested before any is waited for: the scheduler put the first's add in between)
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) {
        uint32_t aa = chain_row_of<SEL>(u[h], c[h]);
        if constexpr (!ROW16) aa <<= 1;
        t[h] = *reinterpret_cast<RowPtr>(aa + ROWS_AT);
    }
}
template <uint32_t NP, int SEL>
__device__ __forceinline__ void chain_next(uint32_t (&e)[NP], const uint32_t (&c)[NP], const uint32_t (&u)[NP], const uint32_t (&t)[NP]) {
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) {
        ++e[h];
        chain_select<SEL>(e[h], t[h], u[h], c[h]);   // (0xFF -- no child, or a D state -- is no class)
    }
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
    pc.r = (s * T.inv_p) >> 16;   // s / P for s < 192, P <= 64 (a division by a run-time value is ~ 30 instructions)
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

template <int MODE, bool ALLD, bool ROW16, uint32_t NP>
__global__ __launch_bounds__(kChainThreads) void k_walk_chain(DevAutomaton A, BatchArgs B, ChainArgs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t total_items = B.item_base[B.n_haystacks];
    // (the image at offset 0 -- no static shared variable in front of it: the steps read at the offsets they compute)
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
    // A lane walks NP pieces at once, NP lane .. NP lane + NP - 1 of the wave's 64 NP: a step is two dependent LDS reads, and with
    // the whole LDS taken by the image a CU runs one workgroup -- four waves per SIMD, too few to hide them (one piece per lane:
    // 55 % of the VALU cycles used).  The walks are independent: their reads are issued together.  A lane's records -- the first
    // piece's, then the next's -- are one contiguous stretch of the output.
    // Event lists: entry k of piece h of this lane at byte (h x spill_cap + k) x 256 + 4 lane of the wave's region of a global buffer
    // (L2: written and read back within microseconds).  Not in LDS: two entries per piece there and the rest in memory meant a
    // compare and two store paths in every step with an event -- 8 VALU and 6 SALU operations per event and piece, which is what
    // the kernel spent most of its time on once a piece had an event every fifteen bytes; one path, its offset bumped, is 2.
    constexpr uint32_t kPieces = 64u * NP;   // pieces of a wave's tile
    constexpr uint32_t kAhead = NP == 2 ? 4u : 3u;   // text blocks requested ahead of the one that is walked (the registers of 4 waves per SIMD: 128)
    uint8_t *const spill = reinterpret_cast<uint8_t *>(T.spill) + uint64_t(__builtin_amdgcn_readfirstlane(int(blockIdx.x * kChainWaves + wave))) * T.spill_cap * (256u * NP);
    uint32_t off0[NP];
#pragma unroll
    for (uint32_t h = 0; h < NP; ++h) off0[h] = h * T.spill_cap * 256u + lane * 4u;
    uint32_t walked16 = 0;
    const uint32_t cnt_n = A.chain_cnt_n;
    const uint32_t gw = blockIdx.x * kChainWaves + wave;   // this wave among the grid's
    const uint64_t tiles_all = (total_items + T.R - 1u) / T.R;
    uint32_t *const log = T.log + uint64_t(gw) * T.log_cap;
    uint32_t log_at = 0;   // words of the wave's log in use
#pragma unroll 1
    for (uint64_t t = gw; t < tiles_all; t += T.n_waves) {
        const uint64_t first_item = t * T.R;
        Piece pc[NP];
        uint32_t e[NP], o[NP];   // the pieces' states, where their next events go
        uint32_t my_blocks = 0;
#pragma unroll
        for (uint32_t h = 0; h < NP; ++h) {
            pc[h] = chain_piece(A, B, T, total_items, first_item, NP * lane + h);
            walked16 += pc[h].nblk;
            my_blocks = pc[h].nblk > my_blocks ? pc[h].nblk : my_blocks;
            e[h] = 0u;
            o[h] = off0[h];
        }
        // ---- the walk: all lanes in step, a lane past its own bytes walks on without reporting ----
        if (__any(my_blocks != 0u) && !(T.debug & 2u)) {
            // An event inside the walk is ONE store, {index, byte}, to the piece's list, and the list's offset bumped in place: two
            // VALU operations.  What it stands for is worked out behind the walk, where a lane runs over its own events only -- in
            // here the whole wave pays for every step in which ANY lane has one.  (Round 5 carried the parent record's count code
            // along: an and, a three-way or, and the bump through a copy -- four operations at 0.4-0.9 of the steps; the code is
            // in LDS where the drain can read it, records_of.)
            auto event = [&](uint32_t idx, uint32_t &off, uint32_t jj) __attribute__((always_inline)) {
                *reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(spill) + off) = idx | jj << 15;
                asm volatile("v_add_u32 %0, 0x100, %0" : "+v"(off));   // (in place: hipcc bumped a copy and moved it back)
            };
            // the column offsets of a dword's bytes: SWAR3 a handful of operations for all four; TABLE an LDS read per byte, requested a
            // byte ahead of its step (all of a dword's up front cost the registers that spilled into this loop)
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
            // four steps of every piece.  CHECKED: the range test of a block that does not lie inside every piece's own range; WARM:
            // the first block, in which a walk begins (the state is the root before).
            // QUIET: a first block in which no walk of the wave reports and every walk begins at byte 1 (or not at all: such a lane
            // is put back to the root behind the block) -- no range test, no event test, no reset; SKIP0: its first dword, less byte 0
            auto dword = [&](const uint32_t (&w)[NP], uint32_t jj, auto checked, auto warm, auto quiet, auto skip0, auto first) __attribute__((always_inline)) {
                constexpr bool C = decltype(checked)::value, W0 = decltype(warm)::value, Q = decltype(quiet)::value, S0 = decltype(skip0)::value, F = decltype(first)::value;
                uint32_t c4[NP], nc[NP];
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) {
                    c4[h] = nc[h] = 0u;
                    if constexpr (MODE == 1) {
                        c4[h] = swar_columns<uint16_t>(A, w[h]);   // (bytes of 2 x class: even, so that a shift leaves the classes)
                        if constexpr (!ALLD && !ROW16) c4[h] >>= 1;
                    } else {
                        nc[h] = cls_of(w[h], S0 ? 1u : 0u);
                    }
                }
                auto one = [&](auto sel) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(sel)::value;
                    if constexpr (S0 && i == 0u) return;
                    const uint32_t j = jj + i;
                    constexpr int SEL = MODE == 1 ? int(i) : 4;
                    uint32_t c[NP], u[NP];
#pragma unroll
                    for (uint32_t h = 0; h < NP; ++h) {
                        c[h] = c4[h];
                        if constexpr (MODE != 1) {
                            c[h] = nc[h];
                            if constexpr (i < 3u) nc[h] = cls_of(w[h], i + 1u);
                        }
                    }
                    if constexpr (ALLD) {
                        chain_step_rows<NP, SEL>(L, e, c, u);
#pragma unroll
                        for (uint32_t h = 0; h < NP; ++h) {
                            if (W0) e[h] = j < pc[h].start_j ? 0u : e[h];
                            if constexpr (!Q) {   // (rep_lo >= start_j: nothing before a walk's first byte reports)
                                if (int32_t(u[h]) < 0 && (!C || j - pc[h].rep_lo < pc[h].rep_len)) event(e[h], o[h], j);
                            }
                        }
                    } else {
                        uint32_t tr[NP], e0[NP];
#pragma unroll
                        for (uint32_t h = 0; h < NP; ++h) e0[h] = e[h];
                        chain_read<NP, ROW16, kRowsAt, SEL>(L, e, c, u, tr);
                        chain_next<NP, SEL>(e, c, u, tr);
#pragma unroll
                        for (uint32_t h = 0; h < NP; ++h) {
                            if constexpr (!Q) {
                                // (the state the step LEFT, entered with byte j - 1: a block's first step looks back into the block before,
                                // which may lie in front of the piece's own range -- that one step is tested whatever the block.  Behind
                                // the selects: between the rows' reads and the selects, the branches around the stores cut the reads
                                // off from their zero-extension -- an and per read -- and invited hipcc to sink a read)
                                constexpr bool TT = C || (F && i == 0u);
                                if (int32_t(u[h]) < 0 && (!TT || j - 1u - pc[h].rep_lo < pc[h].rep_len)) event(e0[h], o[h], j - 1u);
                            }
                            if (W0) e[h] = j < pc[h].start_j ? 0u : e[h];
                        }
                    }
                };
                one(std::integral_constant<uint32_t, 0u>{}); one(std::integral_constant<uint32_t, 1u>{});
                one(std::integral_constant<uint32_t, 2u>{}); one(std::integral_constant<uint32_t, 3u>{});
            };
            auto block4 = [&](const uint4 (&v)[NP], uint32_t j0, auto checked, auto warm, auto quiet, auto skip0) __attribute__((always_inline)) {
                constexpr std::true_type yes{};
                constexpr std::false_type no{};
                uint32_t w[NP];
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) w[h] = v[h].x;
                dword(w, j0, checked, warm, quiet, skip0, yes);
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) w[h] = v[h].y;
                dword(w, j0 + 4u, checked, warm, quiet, no, no);
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) w[h] = v[h].z;
                dword(w, j0 + 8u, checked, warm, quiet, no, no);
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) w[h] = v[h].w;
                dword(w, j0 + 12u, checked, warm, quiet, no, no);
            };
            // (explicitly global loads: through a generic pointer hipcc emits flat_load, which counts on the LDS counter too -- every
            // wait for a state's record then also waited for the text requested four blocks ahead)
            auto block = [](const uint8_t *base, uint32_t i) __attribute__((always_inline)) -> uint4 {
                const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(reinterpret_cast<uintptr_t>(base) + 16u * i);
                return make_uint4(v.x, v.y, v.z, v.w);
            };
            constexpr std::true_type yes{};
            constexpr std::false_type no{};
            uint32_t last[NP], sj[NP];
            uint4 fb[NP], buf[NP][kAhead];
            bool silent = true, plain = true, any_first = false;
#pragma unroll
            for (uint32_t h = 0; h < NP; ++h) {
                last[h] = pc[h].nblk ? pc[h].nblk - 1u : 0u;
                fb[h] = block(pc[h].pb, pc[h].start_j < 16u ? 0u : (last[h] ? 1u : 0u));   // (start_j == 16: block 0 is virtual)
#pragma unroll
                for (uint32_t d = 0; d < kAhead; ++d) buf[h][d] = block(pc[h].pb, d + 1u < last[h] ? d + 1u : last[h]);
                sj[h] = pc[h].nblk ? pc[h].start_j : 16u;
                silent = silent && (!pc[h].nblk || pc[h].rep_lo >= 16u);   // nothing reports in block 0
                plain = plain && (sj[h] == 1u || sj[h] >= 16u);
                any_first = any_first || sj[h] < 16u;
            }
            plain = plain && silent;
            // (the usual tile: Lmax - 1 <= 15 bytes of warm-up in front of pieces that begin on block boundaries -- fifteen steps from
            // the root for every lane, nothing to test)
            if (__all(plain)) {
                if (__any(any_first)) {
                    block4(fb, 0u, no, no, yes, yes);
#pragma unroll
                    for (uint32_t h = 0; h < NP; ++h) e[h] = sj[h] >= 16u ? 0u : e[h];
                }
            } else if (__all(silent)) {   // (shorter patterns: the walks begin further inside the block -- the reset per step, no tests)
                block4(fb, 0u, no, yes, yes, no);
            } else {   // (some walk begins inside the block, or reports in it)
                block4(fb, 0u, yes, yes, no, no);
            }
            uint32_t k = 1u;
#pragma unroll 1
            for (; __any(k < my_blocks); ++k) {   // (as many blocks as the wave's longest walk has: a ballot, not a reduction)
                asm volatile("; acmi-hot-begin");   // (tools/check_asm.py: no scratch traffic between the markers)
                uint4 cur[NP];
                const uint32_t kn = k + kAhead, j0 = k * 16u;
                bool full = true;
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) {
                    cur[h] = buf[h][0];
#pragma unroll
                    for (uint32_t d = 0; d + 1u < kAhead; ++d) buf[h][d] = buf[h][d + 1u];
                    if (kn < my_blocks) buf[h][kAhead - 1u] = block(pc[h].pb, kn < last[h] ? kn : last[h]);
                    // (a block that lies inside EVERY piece's own range -- all but the first and the last of a piece -- without the range test)
                    full = full && j0 >= pc[h].rep_lo && j0 + 16u <= pc[h].rep_lo + pc[h].rep_len;
                }
                if (__all(full)) block4(cur, j0, no, no, no, no);
                else block4(cur, j0, yes, no, no, no);
                asm volatile("; acmi-hot-end");
            }
            if constexpr (!ALLD) {   // the states the walks end in: entered with the last byte walked, 16 k - 1
                const uint32_t jl = 16u * k - 1u;
#pragma unroll
                for (uint32_t h = 0; h < NP; ++h) {
                    const uint32_t u = lds_read<uint32_t>(smem, (e[h] << 2) + L.rec_at);
                    if (int32_t(u) < 0 && jl - pc[h].rep_lo < pc[h].rep_len) event(e[h], o[h], jl);
                }
            }
        }
        uint32_t n[NP], ne = 0;   // events of the pieces
#pragma unroll
        for (uint32_t h = 0; h < NP; ++h) {
            n[h] = (o[h] - off0[h]) >> 8;
            ne += n[h];
        }
        if (ne) {   // (this lane's own stores: landed before they are read back)
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
        // {index, byte, piece}.  (The list entries are the lane's own stores, read back from L2 -- the CU's L1 may still hold the line as
        // a tile before left it -- FOUR at a time: one after the other, each a round trip of its own, they were most of the kernel's
        // time on text with an event every nine bytes.)
        const uint32_t e_incl = wave_inclusive_sum(ne);
        const uint32_t e_tot = uint32_t(__builtin_amdgcn_readlane(int(e_incl), 63));
        if (lane == 0) T.tile_info[t] = make_uint2(log_at, e_tot);
        const bool fits = log_at + e_tot <= T.log_cap;
        // (the log is sized for the batches before: the host grows it and runs the batch again)
        if (!fits && lane == 0) atomicOr(&B.counters[19], 1u);
        uint32_t *dst = log + log_at + (e_incl - ne);
        auto drain = [&](uint32_t h, uint32_t cnt, uint32_t at0, uint32_t *to) __attribute__((always_inline)) -> uint32_t {
            uint32_t recs = 0;
            const uint32_t tag = (NP * lane + h) << 24;   // (bits 24..31 name the piece)
#pragma unroll 1
            for (uint32_t q = 0; q < cnt; q += 4u) {
                uint32_t ev[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i)
                    ev[i] = q + i < cnt ? __hip_atomic_load(reinterpret_cast<uint32_t *>(spill + at0 + (q + i) * 256u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    if (q + i < cnt) {
                        recs += records_of(ev[i]);
                        if (fits) to[q + i] = (ev[i] & 0x00FFFFFFu) | tag;
                    }
                }
            }
            return recs;
        };
        uint32_t r[NP], n_rec = 0;   // records of the pieces
#pragma unroll
        for (uint32_t h = 0; h < NP; ++h) {
            r[h] = drain(h, n[h], off0[h], dst);
            dst += n[h];
            n_rec += r[h];
        }
        log_at += e_tot;
        // ---- records: exclusive prefix over the wave's lanes, the items' totals (k_expand's scan turns them into bases) ----
        const uint32_t incl = wave_inclusive_sum(n_rec);
        const uint32_t excl = incl - n_rec;
        const uint32_t wave_tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        {
            // an item's total: the running sum behind its last piece less the sum before its first (pieces NP l .. NP l + NP - 1 of lane l)
            uint32_t upto[NP];   // records before this lane's piece h
            upto[0] = excl;
#pragma unroll
            for (uint32_t h = 1; h < NP; ++h) upto[h] = upto[h - 1u] + r[h - 1u];
            auto before = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t {   // records of the pieces before piece s (s < 64 NP)
                const uint32_t l = NP == 2 ? s >> 1 : (s * 171u) >> 9;   // (s / 3 for s < 256)
                const uint32_t h = s - l * NP;
                uint32_t v = uint32_t(__shfl(int(upto[0]), int(l), 64));
#pragma unroll
                for (uint32_t g = 1; g < NP; ++g) {
                    const uint32_t x = uint32_t(__shfl(int(upto[g]), int(l), 64));
                    v = h == g ? x : v;
                }
                return v;
            };
#pragma unroll
            for (uint32_t h = 0; h < NP; ++h) {
                const uint32_t s0 = pc[h].r * T.P, s1 = s0 + T.P;   // pieces [s0, s1) (s1 <= 64 NP for an item the wave has)
                const uint32_t lo = before(s0 < kPieces ? s0 : kPieces - 1u);
                const uint32_t up = before(s1 < kPieces ? s1 : kPieces - 1u);   // (a shuffle: by ALL lanes, whatever they then keep)
                const uint32_t hi = s1 < kPieces ? up : wave_tot;
                if (pc[h].have && pc[h].qi == 0u && hi != lo) B.rec_count[pc[h].item] = hi - lo;
            }
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
    // (a tile is two dependent round trips -- where its events lie, then the events, its pieces and its base together -- and a wave
    // takes its tiles one after the other: the next tile's {log offset, count} is requested while this one's records are placed)
    uint64_t t = uint64_t(blockIdx.x) * (kEmitThreads / 64u) + (threadIdx.x >> 6);
    uint2 info_next = t < tiles_all ? T.tile_info[t] : make_uint2(0u, 0u);
#pragma unroll 1
    for (; t < tiles_all; t += n_waves) {
        const uint2 info = info_next;
        if (t + n_waves < tiles_all) info_next = T.tile_info[t + n_waves];
        if (info.y == 0u) continue;
        // (a tile whose events did not fit its wave's log wrote none of them: the host has the batch run again with a bigger log,
        // counters[19] -- nothing of this tile may be read here, its entries would be another tile's or lie behind the log's end)
        if (uint64_t(info.x) + info.y > T.log_cap) continue;
        const uint32_t *ev = T.log + (t % T.n_waves) * uint64_t(T.log_cap) + info.x;
        const uint64_t first_item = t * T.R;
        // (what an event needs of its piece: the lane that walked piece s is s / np, np = 2 or 3 pieces a lane)
        const Piece pa = chain_piece(A, B, T, total_items, first_item, T.np * lane), pz = chain_piece(A, B, T, total_items, first_item, T.np * lane + 1u);
        const Piece pm = T.np > 2u ? chain_piece(A, B, T, total_items, first_item, T.np * lane + 2u) : pz;
        uint64_t at = T.rec_base[first_item];
        // four rounds of 64 events at a time: a round is three dependent round trips (the event, its state's output head, the store),
        // and only the running sum ties one round to the next -- the rounds' loads are issued together, kind by kind
        auto place = [&](uint32_t ent, uint2 hd, uint32_t c) __attribute__((always_inline)) {
            const uint32_t idx = ent & 0x7FFFu, piece = ent >> 24, owner = T.np == 2u ? (piece >> 1) & 63u : ((piece * 171u) >> 9) & 63u, which = piece - owner * T.np;
            const uint32_t ba = uint32_t(__shfl(int(pa.pos_bias), int(owner), 64)), bz = uint32_t(__shfl(int(pz.pos_bias), int(owner), 64)), bm = uint32_t(__shfl(int(pm.pos_bias), int(owner), 64));
            const uint32_t ha = uint32_t(__shfl(int(pa.hay), int(owner), 64)), hz = uint32_t(__shfl(int(pz.hay), int(owner), 64)), hm = uint32_t(__shfl(int(pm.hay), int(owner), 64));
            const uint32_t pos = (which == 0u ? ba : which == 1u ? bz : bm) + ((ent >> 15) & 511u), hay = which == 0u ? ha : which == 1u ? hz : hm;
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
// pieces a lane walks at once.  (The kernel is written for any number; THREE were built and measured in round 6 -- text blocks three
// ahead instead of four, 128 registers with 24-41 spilled around the tile loop, none inside it -- and were slower everywhere: 1 GiB of
// the headline set at 64 / 256 occurrences per 8 KiB 0.68 -> 0.92 / 0.72 -> 1.02 ms for the walk, the word list 0.72 -> 0.99: a third
// stream of random reads queues behind the same LDS pipe, which is what bounds the kernel)
uint32_t chain_np() { return 2u; }
uint64_t chain_tiles(uint64_t max_items, uint32_t P) { return max_items / (64u * chain_np() / P) + 1u; }
// the waves' overflow regions: what a piece can have beyond its LDS list, a word each
// (an item's last piece takes the item's unaligned head as well: up to W + 15 bytes, an event each)
uint64_t chain_spill_words(uint32_t W, uint32_t reserve_cus) { return uint64_t(chain_waves(reserve_cus)) * uint64_t(W + 16u) * 64u * chain_np(); }

static ChainArgs chain_args(const DevAutomaton &A, const BatchArgs &B, uint32_t W, uint32_t P, const ChainBuffers &C) {
    ChainArgs T{};
    T.W = W;
    T.P = P;
    T.np = chain_np();
    T.R = 64u * T.np / P;   // (a lane takes np pieces)
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
    if (alld) return class_mode == 1 ? go(k_walk_chain<1, true, true, 2>) : go(k_walk_chain<0, true, true, 2>);
    // ACMI_CHAIN_ROW16=0 (measurements): the halved row offsets where the whole ones fit
    static const bool no_row16 = [] { const char *e = std::getenv("ACMI_CHAIN_ROW16"); return e && e[0] == '0'; }();
    if (Y0.row16 && !no_row16) return class_mode == 1 ? go(k_walk_chain<1, false, true, 2>) : go(k_walk_chain<0, false, true, 2>);
    return class_mode == 1 ? go(k_walk_chain<1, false, false, 2>) : go(k_walk_chain<0, false, false, 2>);
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
