// engine_replace.cpp -- search and replace over a matched batch (acmi_set_replacement, acmi_replace_batch*)
#include "engine.hpp"

// ---- search and replace (multifast_replace, reference src/multifast/replace.c) ------------------------------------
extern "C" {

int acmi_set_replacement(acmi_handle *h, uint32_t pattern_index, const uint8_t *text, uint32_t length) {
    return guarded(h, [&]() -> int {
        if (!live(h)) return ACMI_E_INVALID;
        if (h->init_ok != 1) return ACMI_E_NOT_INIT;
        if (h->aut.finalized()) return ACMI_E_FINALIZED;
        if (pattern_index >= h->aut.patterns().size() || (length && !text)) return ACMI_E_ARG;
        h->aut.set_replacement(pattern_index, text, length);
        return ACMI_OK;
    });
}

namespace {

// multifast_replace over a batch that is resident in device memory; the output stays there (h->d_rep_out / d_rep_out_off).
// One host wait: the match pipeline's own, then plan -> scan -> emit are enqueued behind it with an OPTIMISTIC output
// capacity (the largest output so far, at least the input's size plus a quarter) and the total read back at the end; only
// an output that outgrew the buffer costs a second emit.
int replace_on_device(acmi_handle *h, const uint8_t *d_blob, uint64_t blob_bytes, const uint64_t *d_off, size_t nh, int mode,
                      uint64_t *out_bytes) {
    hipStream_t st = h->stream;
    if (!h->rep_uploaded) {
        std::vector<uint64_t> off = h->aut.replacement_off();
        off.resize(std::max<size_t>(1, h->aut.patterns().size()), ~0ull);
        for (size_t i = 0; i < h->aut.patterns().size(); ++i)
            if (!h->aut.patterns()[i].accepted) off[i] = ~0ull;
        HIPCHK(h, h->d_rep_off.ensure(off.size() * 8));
        HIPCHK(h, copy_h2d(h->d_rep_off.p, off.data(), off.size() * 8));
        const std::vector<uint8_t> &rb = h->aut.replacement_bytes();
        HIPCHK(h, h->d_rep_bytes.ensure(rb.size() + 16));
        if (!rb.empty()) HIPCHK(h, copy_h2d(h->d_rep_bytes.p, rb.data(), rb.size()));
        h->rep_uploaded = true;
    }
    *out_bytes = 0;
    if (nh == 0) {
        HIPCHK(h, h->d_rep_out_off.ensure(8));
        HIPCHK(h, hipMemsetAsync(h->d_rep_out_off.p, 0, 8, st));
        HIPCHK(h, h->d_rep_out.ensure(16));
        HIPCHK(h, hipStreamSynchronize(st));
        return ACMI_OK;
    }
    // 1. every match of the batch (results left on the device)
    int rc = run_batch(h, d_blob, blob_bytes, d_off, nh, 1);
    if (rc < 0) return rc;
    // 2. nominees and output lengths per haystack, 3. output offsets, 4. the output texts
    HIPCHK(h, h->d_plan.ensure(std::max<uint64_t>(8, h->last_total * 8)));
    HIPCHK(h, h->d_plan_out.ensure(std::max<uint64_t>(8, h->last_total * 8)));
    HIPCHK(h, h->d_tail_out.ensure(nh * 8));
    HIPCHK(h, h->d_nom_count.ensure(nh * 4));
    HIPCHK(h, h->d_out_len.ensure(nh * 8));
    HIPCHK(h, h->d_rep_out_off.ensure((nh + 1) * 8));
    const size_t sync_bytes = 8 + size_t(scan_tiles(nh + 1) + 1) * 8;
    HIPCHK(h, h->d_rep_sync.ensure(sync_bytes));
    HIPCHK(h, hipMemsetAsync(h->d_rep_sync.p, 0, sync_bytes, st));
    h->rep_cap = std::max<uint64_t>(h->rep_cap, blob_bytes + blob_bytes / 4 + 4096);
    HIPCHK(h, h->d_rep_out.ensure(h->rep_cap + 16));
    ReplaceArgs R{};
    R.blob = d_blob;
    R.offsets = d_off;
    R.n_haystacks = uint32_t(nh);
    R.mode = mode == 2 ? 2u : 1u;   // MF_REPLACE_MODE_DEFAULT behaves as NORMAL (replace.c:252-254)
    R.prefix = h->d_prefix().as<uint64_t>();
    R.records = h->d_out().as<uint4>();
    R.pattern_length = h->dA.pattern_length;
    R.rep_off = h->d_rep_off.as<uint64_t>();
    R.rep_bytes = h->d_rep_bytes.as<uint8_t>();
    R.plan = h->d_plan.as<uint2>();
    R.plan_out = h->d_plan_out.as<uint64_t>();
    R.tail_out = h->d_tail_out.as<uint64_t>();
    R.nom_count = h->d_nom_count.as<uint32_t>();
    R.out_len = h->d_out_len.as<uint64_t>();
    R.out_off = h->d_rep_out_off.as<uint64_t>();
    R.out_blob = h->d_rep_out.as<uint8_t>();
    // the total travels through the pinned block of the slot the match result does not use
    uint64_t *total_host = h->pinned + (kPinWords / 2) * (h->cur ^ 1);
    uint64_t *total_dev = reinterpret_cast<uint64_t *>(h->pinned_dev + kPinWords * (h->cur ^ 1));
    launch_replace_plan(R, st);
    launch_scan_u64(R.out_len, nh, h->d_rep_out_off.as<uint64_t>(), h->d_rep_sync.as<uint32_t>(), total_dev, st);
    launch_replace_emit(R, h->rep_cap, st);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(st));
    const uint64_t total = *total_host;
    if (total > h->rep_cap) {
        h->rep_cap = total + total / 8;
        HIPCHK(h, h->d_rep_out.ensure(h->rep_cap + 16));
        R.out_blob = h->d_rep_out.as<uint8_t>();
        launch_replace_emit(R, h->rep_cap, st);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
    }
    *out_bytes = total;
    return ACMI_OK;
}

int replace_args_ok(acmi_handle *h, int mode) {
    if (mode < 0 || mode > 2) return ACMI_E_ARG;
    if (!h->aut.has_replacement()) {  // multifast_replace returns -2 and emits nothing (replace.c:482-483)
        h->last_error = "no pattern has a replacement text";
        return ACMI_E_NO_REPLACEMENT;
    }
    return ACMI_OK;
}

}  // namespace

int acmi_replace_batch_device(acmi_handle *h, const void *d_blob, uint64_t blob_bytes, const void *d_offsets, size_t n_haystacks,
                              int mode, const void **d_out_blob, const void **d_out_offsets, uint64_t *out_bytes) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !d_offsets) return ACMI_E_ARG;
        rc = replace_args_ok(h, mode);
        if (rc < 0) return rc;
        uint64_t total = 0;
        rc = replace_on_device(h, static_cast<const uint8_t *>(d_blob), blob_bytes, static_cast<const uint64_t *>(d_offsets), n_haystacks, mode, &total);
        if (rc < 0) return rc;
        if (d_out_blob) *d_out_blob = h->d_rep_out.p;
        if (d_out_offsets) *d_out_offsets = h->d_rep_out_off.p;
        if (out_bytes) *out_bytes = total;
        return ACMI_OK;
    });
}

int acmi_replace_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t n_haystacks, int mode,
                       const uint8_t **out_blob, const uint64_t **out_offsets) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);
        if (rc < 0) return rc;
        if (n_haystacks && !offsets) return ACMI_E_ARG;
        rc = replace_args_ok(h, mode);
        if (rc < 0) return rc;
        uint64_t blob_bytes = 0;
        for (size_t k = 0; k < n_haystacks; ++k) {
            if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] >= 0xFFFFFFFFull) return ACMI_E_ARG;
        }
        if (n_haystacks) blob_bytes = offsets[n_haystacks];
        if (blob_bytes && !blob) return ACMI_E_ARG;
        hipStream_t st = h->stream;
        HIPCHK(h, h->d_blob.ensure(blob_bytes + 64));
        HIPCHK(h, h->d_off.ensure((n_haystacks + 1) * 8));
        if (blob_bytes) HIPCHK(h, hipMemcpyAsync(h->d_blob.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
        if (n_haystacks) HIPCHK(h, hipMemcpyAsync(h->d_off.p, offsets, (n_haystacks + 1) * 8, hipMemcpyHostToDevice, st));
        uint64_t total = 0;
        rc = replace_on_device(h, h->d_blob.as<uint8_t>(), blob_bytes, h->d_off.as<uint64_t>(), n_haystacks, mode, &total);
        if (rc < 0) return rc;
        h->h_rep_off.assign(n_haystacks + 1, 0);
        h->h_rep_blob.resize(total);
        HIPCHK(h, hipMemcpyAsync(h->h_rep_off.data(), h->d_rep_out_off.p, (n_haystacks + 1) * 8, hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(h, hipMemcpyAsync(h->h_rep_blob.data(), h->d_rep_out.p, total, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (out_blob) *out_blob = h->h_rep_blob.data();
        if (out_offsets) *out_offsets = h->h_rep_off.data();
        return ACMI_OK;
    });
}

}  // extern "C"

