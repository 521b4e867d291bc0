"""k_walk_chain's kernel time on the headline set (64 plants per 8 KiB, 1 GiB), with parts switched off (ACMI_CHAIN_DEBUG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
kind, per = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("headline", 64)
base = synth.density_set(kind)
ds = synth.plant_density(base, per, 8192)
nh = 1 << 17
reps = nh // ds.n_haystacks
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(base.patterns())])
res.set_option("kernel", 6); res.set_option("profile", 1)
batch.bind_current_stream(res)
ms = []
for i in range(8):
    total = batch.match_device(res, d_blob, d_off, True)
    ms.append(res.stat("scan_ms"))
print(kind, per, "debug", os.environ.get("ACMI_CHAIN_DEBUG", "0"), "chain", res.stat("chain_walk_used"), "records", total, "kernel ms", round(float(np.median(ms[3:])), 4))
