// kernels.hpp -- launchers implemented in the kernel family files k_*.hip (device helpers they share: kdev.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.hpp"

namespace acmi {

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel raised to `bytes` when it is not there yet: the driver call (~ 1 us) once
// per kernel and size, not per launch (k_misc.hip)
hipError_t ensure_dynamic_lds(const void *kernel, uint32_t bytes);

bool small_trace();   // ACMI_SMALL_TRACE is set (read once: k_small.hip's launch and engine_pipeline.cpp's run_small share it)

// single-pass scans (kdev.hpp, k_pipeline.hip): tiles of 2048 elements, one status word per tile
uint32_t scan_tiles(uint64_t n);
// isync: 8 + 8 x scan_tiles(n_haystacks + 1) zeroed bytes {ticket, bad-offsets flag, status words}
void launch_items(const BatchArgs &B, uint64_t blob_bytes, uint64_t max_items, uint64_t *item_base, uint32_t *isync, hipStream_t st);
// uniform batches (every haystack `haystack_bytes` long, back to back): the item table in closed form; the offsets array itself
void launch_items_uniform(const BatchArgs &B, uint64_t haystack_bytes, uint64_t *item_base, hipStream_t st);
void launch_uniform_offsets(uint64_t *offsets, uint64_t n, uint64_t haystack_bytes, hipStream_t st);
hipError_t launch_scan(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int kernel,
                       int class_mode, int ilp, int group_quads, hipStream_t st, uint32_t *anchor_variant = nullptr);
hipError_t launch_walk_runs(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                            hipStream_t st, bool stage_tile = true);
// the items k_walk_runs left for it (BatchArgs::wave_runs): a wave per item, a lane per candidate window (k_walk.hip)
hipError_t launch_runs_wave(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode, hipStream_t st);
hipError_t launch_walk_items(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                             int ilp, hipStream_t st);
// kernel 4 over the two-byte window table (k_win.hip): trie depth <= 2 decided in LDS from the text alone
hipError_t launch_walk_win(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, hipStream_t st);
// k_walk_tiles (k_tiles.hip, kernel 6): the exact walk with exact-sized ordered output in one pass.  Geometry: W bytes per lane,
// P lanes per work item (chunk_bytes = P x W, items from k_items with align_warm = 0); false: not for this automaton.
bool tiles_geometry(uint32_t max_pattern_length, uint32_t n_states, uint64_t blob_bytes, uint64_t n_haystacks, uint32_t *W, uint32_t *P);
uint64_t tiles_queue_entries(uint32_t W);                       // 8-byte entries of the waves' event queues
uint64_t tiles_status_words(uint64_t max_items, uint32_t P);    // 8-byte look-back words (zeroed, behind an 8-byte ticket)
// merge_prefix / merge_records (tiles_can_merge only): the match arrays of the other part of a split set, merged into `out`
// k_walk_chain + k_chain_emit (k_chain.hip, kernel 6 for automata that carry the chain-compressed image): the walk over the whole
// automaton in LDS leaves its events in per-wave logs and the items' record counts; behind k_expand's scan the logs become records
struct ChainBuffers {
    uint32_t *spill;       // chain_spill_words(W) words
    uint32_t *log;         // chain_waves() x log_cap words
    uint32_t log_cap;
    uint32_t *tile_info;   // 2 x chain_tiles(max_items, P) words: per tile {where its events begin in its wave's log, how many}
};
bool chain_walk_ok(const DevAutomaton &A);
uint32_t chain_waves(uint32_t reserve_cus = 0);
uint32_t chain_np();   // pieces a lane of k_walk_chain walks at once: a wave's tile is 64 x that many pieces
uint64_t chain_tiles(uint64_t max_items, uint32_t P);
uint64_t chain_spill_words(uint32_t W, uint32_t reserve_cus = 0);
hipError_t launch_walk_chain(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int class_mode, uint32_t W, uint32_t P, const ChainBuffers &C,
                             hipStream_t st);
hipError_t launch_chain_emit(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint32_t W, uint32_t P, const ChainBuffers &C,
                             const uint64_t *rec_base, void *out, uint64_t cap, hipStream_t st);
hipError_t launch_walk_tiles(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode, uint32_t W, uint32_t P,
                             void *queue, uint32_t *ticket, uint64_t *status, void *out, uint64_t cap, hipStream_t st,
                             const uint64_t *merge_prefix = nullptr, const void *merge_records = nullptr);
bool tiles_can_merge(const DevAutomaton &A, int entry_bytes);
void launch_first_only(const BatchArgs &B, uint32_t *first_item, uint64_t max_items, hipStream_t st);
// scan = true: also the exclusive scan of the per-item record counts (ticket / status: zeroed sync words); false: rec_base is read
void launch_expand(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, uint64_t *rec_base, void *out,
                   uint64_t cap, uint64_t *prefix, uint32_t *host_counters, uint32_t *next_sync, uint32_t next_sync_words,
                   uint32_t *ticket, uint64_t *status, const uint32_t *bad_offsets, bool scan, hipStream_t st,
                   uint32_t *next_isync = nullptr, uint32_t next_isync_words = 0);
// the items k_expand left in BatchArgs::dense_list (more events than wave_min): a wave per item, a lane per event
void launch_expand_dense(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, const uint64_t *rec_base, void *out, uint64_t cap, hipStream_t st);
void launch_refill(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes,
                   const uint64_t *rec_base, void *out, uint64_t cap, int mode, hipStream_t st, int class_mode = 0);
// one-launch path for small host batches (<= 64 KiB, <= 1024 haystacks): reads / writes pinned host memory
constexpr uint32_t kSmallPathMaxBytes = 64u * 1024u, kSmallPathMaxHay = 1024u;
hipError_t launch_match_small(const DevAutomaton &A, const uint64_t *offsets, const uint8_t *blob, uint32_t nh, int find_all,
                              uint32_t max_pattern_length, int entry_bytes, int class_mode, uint64_t *out_summary,
                              uint64_t *out_prefix, void *out_records, uint32_t cap, uint32_t span_bytes, uint64_t seq, hipStream_t st);
void launch_build_level(const BuildArgs &T, uint32_t lo, uint32_t hi, hipStream_t st);
void launch_build_by_position(const BuildArgs &T, uint32_t n, uint32_t *pos, uint32_t *fail_b, uint32_t *cnt_b, uint32_t *dict_b,
                              uint8_t *final_b, hipStream_t st);
void launch_replace_plan(const ReplaceArgs &R, hipStream_t st);
void launch_replace_emit(const ReplaceArgs &R, uint64_t cap, hipStream_t st);
// exclusive scan of n 64-bit values into out[0..n]; sync = {ticket, pad, scan_tiles(n + 1) status words}, zeroed; total_host (mapped, may be null) receives out[n]
void launch_scan_u64(const uint64_t *in, uint64_t n, uint64_t *out, uint32_t *sync, uint64_t *total_host, hipStream_t st);
// all-gatherv of the match arrays (k_misc.hip): this rank's slice of the global arrays + its records in the 8-byte wire form
// {pos, pattern}; the other ranks' records expanded from the wire behind the exchange
void launch_gather_pack(const uint64_t *prefix, const uint4 *records, uint64_t nh, uint64_t nrec, uint64_t hay_base, uint64_t rec_base,
                        uint64_t *out_prefix, uint4 *out_records, uint2 *out_wire, uint64_t *closing, uint64_t total_rec, hipStream_t st);
void launch_gather_unpack(const uint64_t *prefix, uint64_t total_hay, const uint2 *wire, uint64_t total_rec, uint64_t skip_lo, uint64_t skip_hi,
                          const uint32_t *pattern_length, uint4 *out_records, hipStream_t st);
// split pattern sets: the two parts' match arrays merged per haystack (position ascending, part A before part B at equal positions)
void launch_merge_counts(const uint64_t *pa, const void *ra, const uint64_t *pb, const void *rb, uint32_t nh, uint32_t first_only, uint64_t *counts,
                         hipStream_t st);
void launch_merge_prefix_sum(const uint64_t *pa, const uint64_t *pb, uint32_t nh, uint64_t *out_prefix, hipStream_t st);
void launch_merge_parts(const uint64_t *pa, const void *ra, uint64_t ta, const uint64_t *pb, const void *rb, uint64_t tb, uint32_t nh,
                        uint32_t first_only, const uint64_t *out_prefix, void *out, hipStream_t st);
void launch_stream_read(const void *src, uint64_t bytes, uint32_t *sink, hipStream_t st);

}  // namespace acmi
