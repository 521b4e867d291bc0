"""The headline pattern set (2048 x 16 B, abcdef) over 4 GiB of haystacks of other lengths: how the scan rate depends
on the haystack size (work items shorter than the anchor kernel's 8 KiB wave step).  python tools/bench_haylen.py [len ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth

lens = [int(a) for a in sys.argv[1:] if "=" not in a] or [8192, 4096, 2048, 1024, 512, 24576]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for hl in lens:
    n_tile = max(1, (64 << 20) // hl)
    ds = synth.generate("L", 1, 2048, 16, 16, n_tile, hl, b"abcdef")
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    for k, v in opts.items():
        res.set_option(k, int(v))
    res.set_option("profile", 2)
    reps = (4 << 30) // (n_tile * hl)
    nh = reps * n_tile
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hl
    batch.bind_current_stream(res)
    for _ in range(3):
        total = batch.match_device(res, d_blob, d_off)
    scan, walk = res.stat("scan_ms"), res.stat("walk_ms")
    res.set_option("profile", 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"haystack {hl:6d} B x {nh:8d}: whole call {dt*1e3:.3f} ms = {d_blob.numel()/dt/1e9:7.1f} GB/s, scan kernel {scan:.3f} ms, walk {walk:.3f} ms, "
          f"records {total}, kernel {int(res.stat('kernel_used'))}, item {int(res.stat('chunk_bytes'))} B", flush=True)
    api.ahocorasick_deinit(res)
    del d_blob, d_off
    torch.cuda.empty_cache()
