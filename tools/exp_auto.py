"""The automatic mode on one density cell (bench.py's density leg): per call ms, kernel, chain walk, log reruns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
kind, per = sys.argv[1], float(sys.argv[2])
base = synth.density_set(kind)
ds = synth.plant_density(base, int(per), 8192) if kind == "headline" else synth.prose(base, per)
nh = 1 << 17
reps = nh // ds.n_haystacks
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(base.patterns())])
if len(sys.argv) > 3: res.set_option("kernel", int(sys.argv[3]))
batch.bind_current_stream(res)
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    total = batch.match_device(res, d_blob, d_off, True)
    ms = (time.perf_counter() - t0) * 1e3
    print(i, round(ms, 3), "kernel", int(res.stat("kernel_used")), "chain", int(res.stat("chain_walk_used")), "log reruns", int(res.stat("chain_log_reruns")), "probes", int(res.stat("choice_probes")), "records", total)
