"""One cell of the density matrix (tests/test_gpu_cliffs.py) on its own: set x planted occurrences per 8 KiB x haystack size, a 64 MiB
batch, N calls -- per call the host time, the kernel times and what the filter's hand-over reported.  Analysis tool.
    python tools/bench_density.py cfg3 128 8192 [calls=8] [kernel=2] [option=value ...]      (under tools/prof_cmd.sh for the kernel table)
    python tools/bench_density.py words prose:1.0 8192 ...      the word list over prose of which that share is listed words (bench.py's cells)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth

kind, per, hay_len = sys.argv[1], sys.argv[2], int(sys.argv[3])
opts = dict(a.split("=") for a in sys.argv[4:])
calls = int(opts.pop("calls", 8))
mib = int(opts.pop("mib", 64))
base = synth.density_set(kind)
ds = synth.prose(base, float(per[6:])) if per.startswith("prose:") else synth.plant_density(base, int(per), hay_len)
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(base.patterns())])
for k, v in opts.items():
    res.set_option(k, int(v))
api.ahocorasick_finalize(res)
reps = (mib << 20) // len(ds.hay_blob)
nh = reps * ds.n_haystacks
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hay_len
batch.bind_current_stream(res)
res.set_option("profile", 2)
for i in range(calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = batch.match_device(res, d_blob, d_off, True)
    ms = (time.perf_counter() - t0) * 1e3
    print(f"call {i}: {ms:.3f} ms  kernel {int(res.stat('kernel_used'))} scan {res.stat('scan_ms'):.3f} walk {res.stat('walk_ms'):.3f} total {res.stat('total_ms'):.3f}  records {total} "
          f"busy {int(res.stat('busy_items'))} max_windows {int(res.stat('max_windows'))} serial {int(res.stat('serial_items'))} wave {int(res.stat('wave_runs_used'))} "
          f"overflow {int(res.stat('overflow_items'))} fallback {int(res.stat('fallback_bytes'))} chunk {int(res.stat('chunk_bytes'))}", flush=True)
res.set_option("profile", 0)
best = 1e9
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = batch.match_device(res, d_blob, d_off, True)
    best = min(best, (time.perf_counter() - t0) * 1e3)
algo = d_blob.numel() + 16 * total + 8 * nh
print(f"best of 5 without events: {best:.4f} ms, frac {algo / best / 1e6 / 8000:.4f}")
