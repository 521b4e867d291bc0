import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gpu_util import device_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth, batch
# parity: G6-like set through kernel 4 with and without the window table
ds = synth.g6(n_hay=64, hay_len=4096)
o = Oracle("reference"); o.add_patterns((ds.pat_blob, ds.pat_off)); pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
for opts in (dict(kernel=4), dict(kernel=4, win_walk=0)):
    res = resource_for(ds, **opts)
    p, r = device_match(res, ds)
    print(opts, "win_table", int(res.stat("win_table")), "ok", bool(np.array_equal(p, pe) and np.array_equal(r, re_)), len(r), len(re_))
    api.ahocorasick_deinit(res)
base = synth.density_set("cfg3")
for per in (1, 32, 512):
    dd = synth.plant_density(base, per, 8192)
    sub = synth.Dataset(dd.name, dd.pat_blob, dd.pat_off, dd.hay_blob[:int(dd.hay_off[32])], dd.hay_off[:33])
    pe, re_ = o.match_batch(sub.hay_blob, sub.hay_off, True) if False else (None, None)
o.close()
o = Oracle("reference"); o.add_patterns((base.pat_blob, base.pat_off))
for per in (1, 32, 512):
    dd = synth.plant_density(base, per, 8192)
    sub = synth.Dataset(dd.name, dd.pat_blob, dd.pat_off, dd.hay_blob[:int(dd.hay_off[32])], dd.hay_off[:33])
    pe, re_ = o.match_batch(sub.hay_blob, sub.hay_off, True)
    res = resource_for(sub, kernel=4)
    for fa_ in (True, False):
        pe, re_ = o.match_batch(sub.hay_blob, sub.hay_off, fa_)
        p, r = device_match(res, sub, fa_)
        print("cfg3", per, fa_, "ok", bool(np.array_equal(p, pe) and np.array_equal(r, re_)), len(r))
    # timing on 64 MiB
    reps = (64 << 20) // len(dd.hay_blob)
    d_blob = torch.from_numpy(dd.hay_blob).cuda().repeat(reps).contiguous()
    nh = reps * dd.n_haystacks
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
    batch.bind_current_stream(res)
    for ww in (1, 0):
        res.set_option("win_walk", ww)
        for _ in range(3): total = batch.match_device(res, d_blob, d_off, True)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); total = batch.match_device(res, d_blob, d_off, True); best = min(best, time.perf_counter() - t0)
        print(f"  cfg3 {per}/8KiB kernel 4 win_walk={ww}: {best*1e3:.3f} ms per 64 MiB, records {total}, chunk {int(res.stat('chunk_bytes'))}")
    api.ahocorasick_deinit(res)
