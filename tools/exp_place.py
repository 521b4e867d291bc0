"""Experiment: does the scan kernel's two-level run-to-run time depend on WHERE the buffers are?
Several blobs / handles inside one process, same content."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
ds = synth.bench_batch(8192, planted=True)
pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
nh = 1 << 19
tile = torch.from_numpy(ds.hay_blob).cuda()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
blobs = []
pads = []
for i in range(3):
    pads.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda"))  # shift the next allocation
    blobs.append(tile.repeat(nh // 8192).contiguous())
handles = [api.ahocorasick_init(pats) for _ in range(2)]
for h in handles:
    h.set_option("profile", 2)
    batch.bind_current_stream(h)
for rep in range(2):
    for hi, h in enumerate(handles):
        for bi, b in enumerate(blobs):
            ms = []
            for _ in range(8):
                batch.match_device(h, b, d_off)
                ms.append(h.stat("scan_ms"))
            print(f"rep {rep} handle {hi} blob {bi} @0x{b.data_ptr():x}: kernel {np.mean(ms[2:]):.4f} ms (min {min(ms[2:]):.4f} max {max(ms[2:]):.4f})", flush=True)
