"""Experiment: two batches in flight (two handles on two streams) vs one, same workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
ds = synth.bench_batch(8192, planted="--no-plant" not in sys.argv)
pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
nh = 1 << 19
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
handles = []
for s in streams:
    h = api.ahocorasick_init(pats)
    h.set_option("profile", 2)
    with torch.cuda.stream(s):
        batch.bind_current_stream(h)
    handles.append(h)
torch.cuda.synchronize()
def run(depth, steps=40):
    for h in handles[:depth]:
        batch.match_device(h, d_blob, d_off)
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    inflight = []
    for i in range(steps):
        h = handles[i % depth]
        if len(inflight) == depth:
            hh = inflight.pop(0)
            batch.match_device_end(hh); ms.append(hh.stat("scan_ms"))
        batch.match_device_begin(h, d_blob, d_off, True)
        inflight.append(h)
    for hh in inflight:
        batch.match_device_end(hh); ms.append(hh.stat("scan_ms"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"depth {depth}: {dt*1e3:.4f} ms per step = {nh*8192/dt/1e9:.0f} GB/s, scan kernel mean {np.mean(ms):.4f} ms", flush=True)
for rep in range(2):
    run(1); run(2)
