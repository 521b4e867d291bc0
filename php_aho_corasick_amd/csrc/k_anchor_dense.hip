// k_anchor_dense.hip -- the dense form of the anchor filter (windows at every byte offset: shortest pattern 4..6 B), its twelve
// instantiations of k_scan_anchor
#include "k_anchor_impl.hpp"

namespace acmi {

hipError_t launch_anchor_dense(const DevAutomaton &A, const BatchArgs &B, uint32_t blocks, hipStream_t st) {
    auto go = [&](auto fn) {
        const uint32_t lds = kAnchorLdsBytes + (kLdsThreads / 64u) * kStageSlots * 16u;  // table + run staging = 160 KiB
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), uint32_t(lds));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(kLdsThreads), lds, st, A, B);
        return hipGetLastError();
    };
    {
        if (A.anchor_slots == 2u) {
            if (A.anchor_dense_hot) {
                if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, true, true>);
                if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, true, true>);
                return go(k_scan_anchor<1, false, false, false, true, true, true>);
            }
            if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, false, true>);
            if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, false, true>);
            return go(k_scan_anchor<1, false, false, false, true, false, true>);
        }
        if (A.anchor_dense_hot) {
            if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true, true>);
            if (B.short_items) return go(k_scan_anchor<1, false, true, false, true, true>);
            return go(k_scan_anchor<1, false, false, false, true, true>);
        }
        if (B.tiny_items) return go(k_scan_anchor<1, false, true, true, true>);
        if (B.short_items) return go(k_scan_anchor<1, false, true, false, true>);
        return go(k_scan_anchor<1, false, false, false, true>);
    }
}

}  // namespace acmi
