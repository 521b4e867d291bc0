"""A big dictionary of short words: N words of 4..12 letters over 1 GiB of letters and spaces (8-KiB haystacks).
python tools/exp_dict.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

letters = bytes(range(ord("a"), ord("z") + 1))
HL, TILE, NH = 8192, 2048, 1 << 17
for N in [int(a) for a in sys.argv[1:]] or [2048, 8192, 16384, 65536]:
    text = synth.generate("words", 33, 1, 4, 12, TILE, HL, letters + b"     ")
    pats = synth.generate("words", 34, N, 4, 12, 1, 16, letters)
    ds = synth.Dataset("words", pats.pat_blob, pats.pat_off, text.hay_blob, text.hay_off)
    t0 = time.perf_counter()
    res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(ds.patterns())])
    if os.environ.get('KERNEL'):
        res.set_option('kernel', int(os.environ['KERNEL']))
    api.ahocorasick_finalize(res)
    build = time.perf_counter() - t0
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(NH // TILE).contiguous()
    d_off = torch.arange(NH + 1, dtype=torch.int64, device="cuda") * HL
    batch.bind_current_stream(res)
    for _ in range(12):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{N:6d} words: {dt*1e3:8.3f} ms per GiB = {d_blob.numel()/dt/1e9:7.1f} GB/s, {total/NH:7.1f} records per haystack, kernel {int(res.stat('kernel_used'))}, "
          f"anchor variant {int(res.stat('anchor_variant'))}, build {build*1e3:.0f} ms, table {int(res.stat('automaton_bytes'))>>20} MiB, overflow items {int(res.stat('overflow_items'))}", flush=True)
    api.ahocorasick_deinit(res)
    del d_blob, d_off
    torch.cuda.empty_cache()
