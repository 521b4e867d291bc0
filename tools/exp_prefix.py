"""Patterns that share a long prefix (a URL block list: "http://www." + 12 letters + ".com") over text in which that prefix is
frequent (a line of 128 bytes every ...): every occurrence of the prefix is a candidate of the anchor filter, and a q-gram that
stands for hundreds of patterns cannot be verified by a text compare.  python tools/exp_prefix.py [every ...] [key=value ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch  # noqa: E402

every = [int(a) for a in sys.argv[1:] if "=" not in a] or [8192, 1024, 256, 128]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
HL, TILE, NH = 8192, 2048, 1 << 17
r = np.random.default_rng(5)
letters = np.frombuffer(bytes(range(ord("a"), ord("z") + 1)), dtype=np.uint8)
def word(n):
    return bytes(r.choice(letters, size=n).tolist())
TAIL = int(os.environ.get("EXP_TAIL", 12))
pats = [b"http://www." + word(TAIL) + (b".com" if TAIL >= 8 else b"") for _ in range(2048)]
for ev in every:
    text = r.choice(np.frombuffer(bytes(range(ord("a"), ord("z") + 1)) + b" /.:", dtype=np.uint8), size=TILE * HL).astype(np.uint8)
    n_urls = 0
    for at in range(0, TILE * HL - 64, ev):
        u = (pats[int(r.integers(len(pats)))] if r.random() < 0.02 else b"http://www." + word(TAIL) + b".org")
        text[at:at + len(u)] = np.frombuffer(u, dtype=np.uint8)
        n_urls += 1
    res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(pats)])
    for k, v in opts.items():
        res.set_option(k, int(v))
    d_blob = torch.from_numpy(text).cuda().repeat(NH // TILE).contiguous()
    d_off = torch.arange(NH + 1, dtype=torch.int64, device="cuda") * HL
    batch.bind_current_stream(res)
    for _ in range(3):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res.set_option("profile", 2)
    batch.match_device(res, d_blob, d_off)
    print(f"a URL every {ev:5d} B: {dt*1e3:7.3f} ms per GiB = {d_blob.numel()/dt/1e9:7.1f} GB/s, {total/NH:6.1f} records per haystack, kernel "
          f"{int(res.stat('kernel_used'))}, scan {res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} ms, fallback {int(res.stat('fallback_bytes'))>>20} MiB, "
          f"overflow items {int(res.stat('overflow_items'))}", flush=True)
    api.ahocorasick_deinit(res)
    del d_blob, d_off
    torch.cuda.empty_cache()
