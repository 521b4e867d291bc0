"""The host-buffer entry (what the Zend shim calls) on batches of small haystacks: per-call time and rate.
python tools/exp_host_small.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import api as _api, synth  # noqa: E402

ds0 = synth.generate("L", 1, 2048, 16, 16, 1, 16, b"abcdef")
res = api.ahocorasick_init([{"id": k, "value": ds0.pattern(k)} for k in range(ds0.n_patterns)])
for nh, hl in ((1, 8192), (256, 8192), (16384, 64), (1 << 20, 64), (1 << 16, 1024), (1 << 14, 8192), (1 << 10, 1 << 20)):
    ds = synth.generate("H", 3, 1, 16, 16, nh, hl, b"abcdef")
    for h in range(0, nh, max(1, nh // 64)):
        at = int(ds.hay_off[h]) + (hl - 16) // 2
        ds.hay_blob[at:at + 16] = np.frombuffer(ds0.pattern(h % 2048), dtype=np.uint8)
    for _ in range(2):
        prefix, rec = _api.match_arrays(res, ds.hay_blob, ds.hay_off, True)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        prefix, rec = _api.match_arrays(res, ds.hay_blob, ds.hay_off, True)
    dt = (time.perf_counter() - t0) / n
    print(f"{nh:8d} haystacks x {hl:8d} B ({nh*hl/2**20:8.1f} MiB): {dt*1e3:9.3f} ms per call = {nh*hl/dt/1e9:6.2f} GB/s, {dt/nh*1e9:9.1f} ns per haystack, "
          f"records {len(rec)}, kernel {int(res.stat('kernel_used'))}", flush=True)
api.ahocorasick_deinit(res)
