"""Long patterns: 2,048 patterns of L bytes (abcdef) over 1 GiB of 8-KiB / 64-KiB haystacks with one planted occurrence each.
python tools/exp_long.py [L ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

lens = [int(a) for a in sys.argv[1:]] or [16, 64, 256, 1000]
for L in lens:
    for HL in (8192, 65536):
        TILE = (32 << 20) // HL
        NH = (1 << 30) // HL
        ds = synth.generate("L", 1, 2048, L, L, TILE, HL, b"abcdef")
        for h in range(TILE):
            p = np.frombuffer(ds.pattern((7 * h + 3) % 2048), dtype=np.uint8)
            at = int(ds.hay_off[h]) + (31 * h + 5) % (HL - L)
            ds.hay_blob[at:at + L] = p
        t0 = time.perf_counter()
        res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
        api.ahocorasick_finalize(res)
        build = time.perf_counter() - t0
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
        print(f"patterns of {L:4d} B, haystacks of {HL:6d} B: {dt*1e3:7.3f} ms per GiB = {d_blob.numel()/dt/1e9:7.1f} GB/s, records {total} (expected {NH}), "
              f"kernel {int(res.stat('kernel_used'))}, scan {res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} ms, build {build*1e3:.0f} ms, "
              f"walked {int(res.stat('walked_bytes'))>>20} MiB fallback {int(res.stat('fallback_bytes'))>>20} MiB, table {int(res.stat('automaton_bytes'))>>20} MiB, item {int(res.stat('chunk_bytes'))} B", flush=True)
        api.ahocorasick_deinit(res)
        del d_blob, d_off
        torch.cuda.empty_cache()
