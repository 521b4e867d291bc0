"""A batch of very unequal haystacks: one of 256 MiB in front of 786,432 of 1 KiB (1 GiB in all), and the same bytes as equal
haystacks of 8 KiB -- the headline patterns, one planted occurrence per 8 KiB.  python tools/exp_mixed.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

ds = synth.bench_batch(8192, 8192, seed=1, planted=True)
pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(16).contiguous()   # 1 GiB
total_bytes = d_blob.numel()
shapes = {
    "131072 x 8 KiB": torch.arange(131072 + 1, dtype=torch.int64) * 8192,
    "1 x 256 MiB + 786432 x 1 KiB": torch.cat([torch.tensor([0], dtype=torch.int64), (256 << 20) + torch.arange(786432 + 1, dtype=torch.int64) * 1024]),
    "786432 x 1 KiB + 1 x 256 MiB": torch.cat([torch.arange(786432 + 1, dtype=torch.int64) * 1024, torch.tensor([total_bytes], dtype=torch.int64)]),
    "4 x 256 MiB": torch.arange(5, dtype=torch.int64) * (256 << 20),
}
FIND_ALL = os.environ.get("EXP_FIND_ALL", "1") != "0"
for name, off in shapes.items():
    res = api.ahocorasick_init(pats)
    d_off = off.cuda()
    batch.bind_current_stream(res)
    for _ in range(3):
        total = batch.match_device(res, d_blob, d_off, FIND_ALL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off, FIND_ALL)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res.set_option("profile", 2)
    batch.match_device(res, d_blob, d_off)
    print(f"{name:32s}: {dt*1e3:7.3f} ms per GiB = {total_bytes/dt/1e9:7.1f} GB/s, records {total}, kernel {int(res.stat('kernel_used'))}, scan "
          f"{res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} ms, item {int(res.stat('chunk_bytes'))} B", flush=True)
    api.ahocorasick_deinit(res)
