"""Kernel 6 (k_walk_tiles) against kernel 4 (k_walk_items + drains) and the automatic choice, whole calls on resident batches.
usage: python tools/bench_tiles.py [case ...]   cases: short8 words headline64 headline256 split"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from php_aho_corasick_amd import api, batch, synth

dev = torch.device("cuda", 0)
HL = 8192


def timed(res, d_blob, d_off, n=5):
    for _ in range(10):
        total = batch.match_device(res, d_blob, d_off, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        total = batch.match_device(res, d_blob, d_off, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, total


def run(name, pats, hay_blob, nh, opts_list):
    reps = nh // (len(hay_blob) // HL)
    d_blob = torch.from_numpy(hay_blob).to(dev).repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HL
    for opts in opts_list:
        res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(pats)])
        for k, v in opts.items():
            res.set_option(k, v)
        batch.bind_current_stream(res)
        ms, total = timed(res, d_blob, d_off)
        algo = d_blob.numel() + 16 * total + 8 * nh
        print(f"{name:12s} {str(opts):40s} {ms:8.3f} ms  kernel {int(res.stat('kernel_used'))}  records/hay {total / nh:7.1f}  frac {algo / ms / 1e6 / 8000:.3f}", flush=True)
        api.ahocorasick_deinit(res)


cases = sys.argv[1:] or ["short8", "words", "headline64", "headline256", "split"]
nh = 1 << 17
base = synth.bench_batch(8192, HL, seed=1, planted=True)
for c in cases:
    if c == "short8":
        sp = synth.generate("short", 41, 8, 2, 6, 1, 16, b"abcdef")
        run(c, sp.patterns(), base.hay_blob, nh, [dict(kernel=6)] if os.environ.get("ONLY6") else [dict(kernel=4), dict(kernel=6), dict(kernel=6, chain_walk=1), dict(kernel=6, tile_w=32), dict(kernel=3)])
    elif c == "split":
        sp = synth.generate("short", 41, 8, 2, 6, 1, 16, b"abcdef")
        run(c, base.patterns() + sp.patterns(), base.hay_blob, nh, [dict(kernel=3), dict(kernel=6, split=0), dict(kernel=4, split=0)])
    elif c == "words":
        alpha = bytes(range(ord("a"), ord("z") + 1)) + b"     "
        ds = synth.generate("words", 33, 2048, 4, 12, 8192, HL, alpha)
        run(c, ds.patterns(), ds.hay_blob, nh, [dict(kernel=4), dict(kernel=6), dict(kernel=3)])
    elif c.startswith("prose"):
        w = synth.density_set("words")
        ds = synth.prose(w, int(c[5:]) / 100.0)
        run(c, w.patterns(), ds.hay_blob, nh, [dict(kernel=int(os.environ['K']))] if os.environ.get('K') else [dict(kernel=3), dict(kernel=2), dict(kernel=4), dict(kernel=6), dict(kernel=1)])
    elif c.startswith("headline"):
        per = int(c[8:])
        ds = synth.bench_batch(2048, HL, seed=1, planted=False)
        r = np.random.default_rng(5)
        blob = ds.hay_blob.copy()
        for h in range(2048):
            for k in range(per):
                p = np.frombuffer(ds.pattern(int(r.integers(2048))), dtype=np.uint8)
                at = h * HL + (HL // per) * k + int(r.integers(0, max(1, HL // per - 16)))
                blob[at:at + 16] = p
        run(c, ds.patterns(), blob, nh, [dict(kernel=2, chunk_bytes=1024), dict(kernel=2, chunk_bytes=512), dict(kernel=2, chunk_bytes=2048), dict(kernel=2, run_cap=64), dict(kernel=2, run_cap=256, slab_events=256)][(-1 if os.environ.get('ANCHOR') == '2' else 0):] if os.environ.get('ANCHOR') else [dict(kernel=int(os.environ['K']))] if os.environ.get('K') else [dict(kernel=3), dict(kernel=2), dict(kernel=1), dict(kernel=4), dict(kernel=6)])
