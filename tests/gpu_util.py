"""Helpers for the -m gpu tests: build a resource from a synth.Dataset, call through the C ABI."""
from __future__ import annotations

import numpy as np

from php_aho_corasick_amd import api


BUILD_OPTIONS = ("lds_bytes", "max_table_bytes", "max_flat_outputs", "compact", "dense_bytes", "keep_host_tables", "verify", "anchor_choice", "device_build", "anchor_dense", "dense_hot", "split", "anchor_slots2", "chain_bytes", "win_table")


def resource_for(ds, calls: int = 1, **options):
    """ahocorasick_init() over the dataset's patterns (id => caller index), optionally split into
    `calls` add_patterns() calls; options are acmi_set_option() knobs applied before finalize."""
    pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
    step = (len(pats) + calls - 1) // calls if pats else 1
    res = api.ahocorasick_init(pats[:step] if calls > 1 else pats)
    assert res is not False
    for k, v in options.items():
        if k in BUILD_OPTIONS:
            res.set_option(k, v)
    for lo in range(step, len(pats), step):
        if calls > 1:
            assert api.ahocorasick_add_patterns(res, pats[lo:lo + step])
    for k, v in options.items():
        if k not in BUILD_OPTIONS:
            res.set_option(k, v)
    return res


def host_match(res, ds, find_all=True):
    return api.match_arrays(res, ds.hay_blob, ds.hay_off, find_all)


def device_match(res, ds, find_all=True):
    import torch
    from php_aho_corasick_amd import batch
    d_blob = torch.from_numpy(ds.hay_blob if ds.hay_blob.size else np.zeros(16, np.uint8)).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    total = batch.match_device(res, d_blob, d_off, find_all)
    torch.cuda.synchronize()
    return batch.result_numpy(res, ds.n_haystacks, total)
