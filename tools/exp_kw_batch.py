"""The keyword set (bench.py keyword_set: 2,048 patterns of 4..12 letters and spaces over such text) through the dense filter form at
batch sizes from 32 MiB to 1 GiB: does the decision kernel's second read of the text come cheaper while the batch still fits the
memory-side cache?  Kernel times per GiB of text (rocprofv3: tools/prof_cmd.sh kw_batch python .../exp_kw_batch.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
alpha = bytes(range(ord("a"), ord("z") + 1)) + b"     "
ds = synth.generate("kw", 33, 2048, 4, 12, 4096, 8192, alpha)
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(ds.patterns())])
api.ahocorasick_finalize(res)
res.set_option("profile", 2)
batch.bind_current_stream(res)
tile = torch.from_numpy(ds.hay_blob).cuda()
for mib in (32, 64, 128, 256, 1024):
    reps = (mib << 20) // tile.numel()
    d_blob = tile.repeat(reps).contiguous()
    nh = reps * ds.n_haystacks
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
    sc, wk, tt = [], [], []
    for i in range(8):
        total = batch.match_device(res, d_blob, d_off, True)
        if i >= 3:
            sc.append(res.stat("scan_ms")); wk.append(res.stat("walk_ms")); tt.append(res.stat("total_ms"))
    g = 1024.0 / mib
    print(f"{mib:5d} MiB: scan {np.mean(sc)*g:.3f} decide+emit {np.mean(wk)*g:.3f} whole pipeline {np.mean(tt)*g:.3f} ms per GiB, records {total}", flush=True)
    del d_blob, d_off
