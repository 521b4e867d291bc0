"""Experiment: ILP 1 vs 2 (and group size) inside one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
plant = "--no-plant" not in sys.argv
ds = synth.bench_batch(8192, planted=plant)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
res.set_option("profile", 2)
nh = 1 << 19
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
batch.bind_current_stream(res)
for rep in range(2):
    for ilp, g in ((1, 8), (1, 4), (2, 4)):
        res.set_option("ilp", ilp); res.set_option("group_quads", g)
        ms = []
        for _ in range(8):
            batch.match_device(res, d_blob, d_off)
            ms.append(res.stat("scan_ms"))
        print(f"rep {rep} plant {plant} ilp {ilp} G {g}: kernel {np.mean(ms[2:]):.4f} ms", flush=True)
