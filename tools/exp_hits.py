"""How the whole call degrades with the number of matches per haystack: the headline patterns over 1 GiB of 8-KiB haystacks with
P planted occurrences each (P = 1 is the bench's shape; more than slab_events = 8 events per work item overflow its slab and
take the drain).  python tools/exp_hits.py [P ...] [key=value ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

plants = [int(a) for a in sys.argv[1:] if "=" not in a] or [1, 4, 8, 9, 16, 64]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
HL, TILE, NH = 8192, 4096, int(os.environ.get("EXP_NH", 1 << 17))
BIG = os.environ.get("EXP_SET") == "cfg3"   # the 65,536-pattern set of BASELINE configs[2] (random bytes) instead of the headline's
for P in plants:
    ds = synth.g6(n_hay=TILE, hay_len=HL) if BIG else synth.generate("L", 1, 2048, 16, 16, TILE, HL, b"abcdef")
    r = np.random.default_rng(7)
    for h in range(TILE):
        for k in range(P):
            pat = np.frombuffer(ds.pattern(int(r.integers(ds.n_patterns))), dtype=np.uint8)
            at = int(ds.hay_off[h]) + (HL // P) * k + int(r.integers(0, max(1, HL // P - len(pat))))
            ds.hay_blob[at:at + len(pat)] = pat
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    for k, v in opts.items():
        res.set_option(k, int(v))
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(NH // TILE).contiguous()
    d_off = torch.arange(NH + 1, dtype=torch.int64, device="cuda") * HL
    batch.bind_current_stream(res)
    for _ in range(3):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res.set_option("profile", 2)
    batch.match_device(res, d_blob, d_off)
    extra = f", scan {res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} ms, walked {int(res.stat('walked_bytes'))>>20} MiB, fallback {int(res.stat('fallback_bytes'))>>20} MiB"
    print(f"{P:3d} plants per 8 KiB: {dt*1e3:.3f} ms per GiB = {d_blob.numel()/dt/1e9:7.1f} GB/s, records {total} ({total/NH:.1f} per haystack), "
          f"overflow items {int(res.stat('overflow_items'))}, kernel {int(res.stat('kernel_used'))}" + extra, flush=True)
    api.ahocorasick_deinit(res)
    del d_blob, d_off
    torch.cuda.empty_cache()
