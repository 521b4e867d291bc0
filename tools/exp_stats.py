"""Print the engine's batch statistics for the headline workload (analysis only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
for plant in (True, False):
    ds = synth.bench_batch(8192, planted=plant)
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    res.set_option("profile", 2)
    nh = 1 << 19
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
    batch.bind_current_stream(res)
    for _ in range(3):
        batch.match_device(res, d_blob, d_off)
    print("plant", plant, {k: res.stat(k) for k in ("records", "items", "walked_bytes", "overflow_items", "kernel_used", "scan_ms", "total_ms", "anchor_keys", "anchor_q")}, flush=True)
