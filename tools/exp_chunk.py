"""Experiment: scan-kernel time vs work-item size inside ONE process (A/B without the per-process
clock / placement bimodality).  python tools/exp_chunk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
ds = synth.bench_batch(8192)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
res.set_option("profile", 2)
if len(sys.argv) > 1: res.set_option("kernel", int(sys.argv[1]))
nh = 1 << 19
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
batch.bind_current_stream(res)
for rep in range(3):
  for pers in (0, 1):
    res.set_option("persistent", pers)
    for chunk in (0, 1024):
        res.set_option("chunk_bytes", chunk)
        ms = []
        for _ in range(6):
            batch.match_device(res, d_blob, d_off)
            ms.append(res.stat("scan_ms"))
        print(f"rep {rep} persistent {pers} chunk {chunk:5d}: kernel {np.mean(ms[2:]):.4f} ms  items {int(res.stat('items'))}", flush=True)
