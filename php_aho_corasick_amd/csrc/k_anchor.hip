// k_anchor.hip -- the aligned anchor filter (small-set LDS table, big-set bitmap + L2 table) and the launcher of all forms
#include "k_anchor_impl.hpp"

namespace acmi {

hipError_t launch_anchor_dense(const DevAutomaton &A, const BatchArgs &B, uint32_t blocks, hipStream_t st);   // k_anchor_dense.hip

// *variant (acmi_get_stat "anchor_variant"): the template that ran -- QW | BIG << 3 | SHORT << 4 | TINY << 5
hipError_t launch_anchor(const DevAutomaton &A, const BatchArgs &B, uint64_t max_items, hipStream_t st, uint32_t *variant) {
    if (variant) {
        const bool big = A.anchor_mode == 2, tiny = B.tiny_items != 0;
        const bool shrt = !big && (tiny || B.short_items != 0);
        *variant = (A.anchor_q / 4u) | (big ? 8u : 0u) | (shrt ? 16u : 0u) | (tiny ? 32u : 0u) | (A.anchor_dense ? 64u : 0u) | (A.anchor_dense && A.anchor_dense_hot ? 128u : 0u) | (A.anchor_slots == 2u ? 256u : 0u);
    }
    const uint64_t per_block = uint64_t(kLdsThreads / 64u) * B.wave_items;
    const uint64_t want = (max_items + per_block - 1) / per_block;
    const uint32_t blocks = B.persistent ? uint32_t(std::min<uint64_t>(want, uint64_t(std::max(1, num_cus() - int(B.reserve_cus))))) : uint32_t(want);
    auto go = [&](auto fn) {
        const uint32_t lds = kAnchorLdsBytes + (kLdsThreads / 64u) * kStageSlots * 16u;  // table + run staging = 160 KiB
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kLdsThreads), lds, st, A, B);
        return hipGetLastError();
    };
    if (A.anchor_dense) return launch_anchor_dense(A, B, blocks, st);   // windows at every byte offset (shortest pattern 4..6 B)
    if (B.tiny_items) {  // haystacks of a few hundred bytes: the variants whose waves pack one-row items G to a task
        if (A.anchor_mode == 2) {
            if (A.anchor_q == 4) return go(k_scan_anchor<1, true, false, true>);
            if (A.anchor_q == 8) return go(k_scan_anchor<2, true, false, true>);
            if (A.anchor_q == 12) return go(k_scan_anchor<3, true, false, true>);
            return go(k_scan_anchor<4, true, false, true>);
        }
        if (A.anchor_q == 4) return go(k_scan_anchor<1, false, true, true>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, false, true, true>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, false, true, true>);
        return go(k_scan_anchor<4, false, true, true>);
    }
    if (A.anchor_mode == 2) {
        if (A.anchor_q == 4) return go(k_scan_anchor<1, true, false>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, true, false>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, true, false>);
        return go(k_scan_anchor<4, true, false>);
    }
    if (B.short_items) {
        if (A.anchor_q == 4) return go(k_scan_anchor<1, false, true>);
        if (A.anchor_q == 8) return go(k_scan_anchor<2, false, true>);
        if (A.anchor_q == 12) return go(k_scan_anchor<3, false, true>);
        return go(k_scan_anchor<4, false, true>);
    }
    if (A.anchor_q == 4) return go(k_scan_anchor<1, false, false>);
    if (A.anchor_q == 8) return go(k_scan_anchor<2, false, false>);
    if (A.anchor_q == 12) return go(k_scan_anchor<3, false, false>);
    return go(k_scan_anchor<4, false, false>);
}

}  // namespace acmi
