"""Experiment: does the scan-kernel time drift within one process (DVFS) or is it fixed per process?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
ds = synth.bench_batch(8192)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
res.set_option("profile", 2)
nh = 1 << 19
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
batch.bind_current_stream(res)
t0 = time.time()
out = []
for i in range(1500):
    batch.match_device(res, d_blob, d_off)
    out.append(res.stat("scan_ms"))
    if i % 100 == 99:
        print(f"t={time.time()-t0:6.2f}s steps {i-99:4d}-{i:4d}: kernel mean {np.mean(out[-100:]):.4f} min {np.min(out[-100:]):.4f} max {np.max(out[-100:]):.4f}", flush=True)
    if i == 700:
        time.sleep(3.0)   # idle gap: does the clock state change?
