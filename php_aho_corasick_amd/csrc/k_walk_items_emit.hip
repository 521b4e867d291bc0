// k_walk_items_emit.hip -- k_walk_items as the drain of a batch it scanned (records written in place)
#include "k_walk_items_impl.hpp"

namespace acmi {

// The drain of a batch this kernel scanned (launch_refill mode 3): a thread per overflowed item, records written in place.
void launch_walk_items_emit(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, const uint64_t *rec_base, void *out,
                            uint64_t cap, hipStream_t st, int class_mode) {
    const uint32_t lds = kItemsTileBytes + (class_mode == 0 ? 512u : 0u);
    const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);   // (workgroups past the list's end leave at once)
    auto go = [&](auto fn) {
        (void)ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B, rec_base, static_cast<uint4 *>(out), cap);
    };
    if (entry_bytes == 2) { if (class_mode == 1) go(k_walk_items<uint16_t, 1, 4, true>); else go(k_walk_items<uint16_t, 0, 4, true>); }
    else { if (class_mode == 1) go(k_walk_items<uint32_t, 1, 4, true>); else go(k_walk_items<uint32_t, 0, 4, true>); }
}

}  // namespace acmi
