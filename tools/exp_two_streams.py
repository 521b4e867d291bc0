"""Experiment: does the chip overlap one batch's small kernels (k_walk_runs, k_expand, k_items) with the NEXT batch's scan when
the two batches are enqueued on different streams?  Two handles with the same pattern set, each bound to a stream of its
own, take turns (begin A, begin B, end A, begin A, end B, ...) on the headline batch; against one handle with two batches in
flight on one stream (what bench.py times).  Prints ms per batch for both."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import php_aho_corasick_amd as api  # noqa: E402
from php_aho_corasick_amd import batch, synth  # noqa: E402

NH = int(os.environ.get("EXP_NH", 1 << 19))
HL = 8192
TILE = 8192


def main():
    dev = torch.device("cuda:0")
    ds = synth.bench_batch(TILE, HL, seed=1, planted=True)
    pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(NH // TILE).contiguous()
    d_off = torch.arange(NH + 1, dtype=torch.int64, device=dev) * HL
    torch.cuda.synchronize()

    def one_handle(n=12):
        res = api.ahocorasick_init(pats)
        batch.bind_current_stream(res)
        for _ in range(3):
            batch.match_device(res, d_blob, d_off, True)
        batch.match_device_begin(res, d_blob, d_off, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            batch.match_device_begin(res, d_blob, d_off, True)
            batch.match_device_end(res)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        batch.match_device_end(res)
        api.ahocorasick_deinit(res)
        return t * 1e3

    def two_handles(n=12):
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        hs = []
        for s in streams:
            with torch.cuda.stream(s):
                res = api.ahocorasick_init(pats)
                batch.bind_current_stream(res)
                for _ in range(3):
                    batch.match_device(res, d_blob, d_off, True)
                hs.append(res)
        torch.cuda.synchronize()
        totals = []
        batch.match_device_begin(hs[0], d_blob, d_off, True)
        t0 = time.perf_counter()
        for i in range(n):
            a, b = hs[i & 1], hs[(i + 1) & 1]
            batch.match_device_begin(b, d_blob, d_off, True)
            totals.append(batch.match_device_end(a))
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        batch.match_device_end(hs[n & 1])
        for res in hs:
            api.ahocorasick_deinit(res)
        return t * 1e3, totals[-1]

    a = one_handle()
    b, total = two_handles()
    gb = NH * HL / 1e9
    print(f"one handle, two in flight on one stream : {a:.4f} ms per batch = {gb / a * 1e3:.0f} GB/s")
    print(f"two handles on two streams, alternating : {b:.4f} ms per batch = {gb / b * 1e3:.0f} GB/s (records {total})")


if __name__ == "__main__":
    main()
