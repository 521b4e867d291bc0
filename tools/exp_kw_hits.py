"""How the keyword path degrades with the density of matches: 2,048 words of 4..12 letters over 1 GiB of 8-KiB haystacks of
letters and spaces into which a share of the text is REPLACED by words of the list (share = 0: the keyword_words leg without
its plants; 1: nothing but listed words).  python tools/exp_kw_hits.py [share ...] [key=value ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

shares = [float(a) for a in sys.argv[1:] if "=" not in a] or [0.0, 0.01, 0.05, 0.2, 0.5, 1.0]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
HL, TILE, NH = 8192, 2048, 1 << 17
letters = bytes(range(ord("a"), ord("z") + 1))
for share in shares:
    ds = synth.generate("words", 33, 2048, 4, 12, TILE, HL, letters + b"     ")
    pats = synth.generate("words", 34, 2048, 4, 12, 1, 16, letters)
    ds = synth.Dataset("words", pats.pat_blob, pats.pat_off, ds.hay_blob, ds.hay_off)
    words = ds.patterns()
    r = np.random.default_rng(11)
    for h in range(TILE):
        at, end = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        while at < end - 16:
            if r.random() < share:
                w = np.frombuffer(words[int(r.integers(len(words)))], dtype=np.uint8)
                ds.hay_blob[at:at + len(w)] = w
                ds.hay_blob[at + len(w)] = 32
                at += len(w) + 1
            else:
                at += 8
    res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(words)])
    for k, v in opts.items():
        res.set_option(k, int(v))
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(NH // TILE).contiguous()
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
    print(f"share {share:4.2f}: {dt*1e3:7.3f} ms per GiB = {d_blob.numel()/dt/1e9:7.1f} GB/s, {total/NH:7.1f} records per haystack, overflow items "
          f"{int(res.stat('overflow_items'))}, kernel {int(res.stat('kernel_used'))}, scan {res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} ms, "
          f"fallback {int(res.stat('fallback_bytes'))>>20} MiB", flush=True)
    api.ahocorasick_deinit(res)
    del d_blob, d_off
    torch.cuda.empty_cache()
