// kernels.hpp -- launchers implemented in kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.hpp"

namespace acmi {

size_t scan_temp_elems(uint64_t n);
void exclusive_scan_u64(const uint64_t *in, uint64_t *out, uint64_t n, uint64_t *temp, hipStream_t st);

void launch_build_items(const BatchArgs &B, uint64_t max_items, hipStream_t st);
void launch_item_base(const uint64_t *offsets, uint32_t nh, uint32_t chunk, uint64_t blob_bytes, uint64_t *n_items,
                      uint64_t *item_base, uint32_t *counters, uint64_t *temp, hipStream_t st);
hipError_t launch_scan(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int kernel,
                       int class_mode, int ilp, int group_quads, hipStream_t st);
hipError_t launch_walk_runs(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                            hipStream_t st);
hipError_t launch_walk_items(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                             int ilp, hipStream_t st);
void launch_first_only(const BatchArgs &B, hipStream_t st);
void launch_hay_prefix(const uint64_t *item_base, const uint64_t *rec_base, uint32_t nh, uint64_t *prefix,
                       uint64_t *summary, hipStream_t st);
void launch_expand(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, const uint64_t *rec_base, void *out,
                   uint64_t cap, hipStream_t st);
void launch_refill(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes,
                   const uint64_t *rec_base, void *out, uint64_t cap, int mode, hipStream_t st);
// one-launch path for small host batches (<= 64 KiB, <= 1024 haystacks): reads / writes pinned host memory
constexpr uint32_t kSmallPathMaxBytes = 64u * 1024u, kSmallPathMaxHay = 1024u;
hipError_t launch_match_small(const DevAutomaton &A, const uint64_t *offsets, const uint8_t *blob, uint32_t nh, int find_all,
                              uint32_t max_pattern_length, int entry_bytes, int class_mode, uint64_t *out_summary,
                              uint64_t *out_prefix, void *out_records, uint32_t cap, hipStream_t st);
void launch_stream_read(const void *src, uint64_t bytes, uint32_t *sink, hipStream_t st);

}  // namespace acmi
