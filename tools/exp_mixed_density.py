"""A handle that sees dense text and then sparse text again: the hand-over behind the filter (lists, slabs, item size, the wave
kernels) follows the text both ways.  cfg3's set, 64 MiB batches: 12 sparse, 6 dense (128 per 8 KiB), 14 sparse; ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
base = synth.density_set(kind)
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(base.patterns())])
res.set_option("kernel", 2)
batch.bind_current_stream(res)
bufs = {}
for per in (1, 128):
    ds = synth.plant_density(base, per, 8192)
    reps = (64 << 20) // len(ds.hay_blob)
    bufs[per] = (torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous(), torch.arange(reps * ds.n_haystacks + 1, dtype=torch.int64, device="cuda") * 8192)
for phase, per, n in (("sparse", 1, 12), ("dense", 128, 6), ("sparse", 1, 14)):
    out = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); total = batch.match_device(res, *bufs[per], True); ms = (time.perf_counter() - t0) * 1e3
        out.append(f"{ms:.3f}")
    print(phase, per, "per 8 KiB:", " ".join(out), "| chunk", int(res.stat("chunk_bytes")), "max_windows", int(res.stat("max_windows")), "wave", int(res.stat("wave_runs_used")), flush=True)
