// k_walk_items.hip -- the exact walk of whole work items (kernel 4); the kernel itself: k_walk_items_impl.hpp
#include "k_walk_items_impl.hpp"

namespace acmi {

hipError_t launch_walk_items(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, int entry_bytes, int class_mode,
                             int group, hipStream_t st) {
    if (max_items == 0) return hipSuccess;
    const uint32_t lds = kItemsTileBytes + (class_mode == 0 ? 512u : 0u);
    auto go = [&](auto fn) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        const uint32_t blocks = uint32_t((max_items + kWalkThreads - 1) / kWalkThreads);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWalkThreads), lds, st, A, B, static_cast<const uint64_t *>(nullptr), static_cast<uint4 *>(nullptr),
                           uint64_t(0));
        return hipGetLastError();
    };
    // 16-byte blocks a lane requests back to back: 4 (`group`, the "walk_group" option, is kept for its callers: single blocks --
    // the first version, 1.42 ms per GiB against 1.16 -- and whole lines of 8 are no longer compiled)
    (void)group;
    const bool d32 = !A.compact && uint64_t(A.n_states) * A.row_bytes < (1ull << 32) && A.n_states < (1u << 24);
#define ACMI_WALK_ITEMS(E, M) (d32 ? go(k_walk_items<E, M, 4, false, true>) : go(k_walk_items<E, M, 4, false, false>))
    if (entry_bytes == 2) return class_mode == 1 ? ACMI_WALK_ITEMS(uint16_t, 1) : ACMI_WALK_ITEMS(uint16_t, 0);
    return class_mode == 1 ? ACMI_WALK_ITEMS(uint32_t, 1) : ACMI_WALK_ITEMS(uint32_t, 0);
#undef ACMI_WALK_ITEMS
}

}  // namespace acmi
