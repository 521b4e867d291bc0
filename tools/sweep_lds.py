"""LDS node-tile sizing sweep (BASELINE configs[2] asks for one): scan-kernel time vs the LDS budget of
the hot tile, for the headline automaton (25,581 states, 7 columns) and the config-3 automaton
(2.27 M states, 256 columns).  python tools/sweep_lds.py [cfg1|cfg3]
cfg3 (round 6): the sweep is taken on the TABLE WALK (kernel 4, the whole-item walk; kernel 1, the optimistic LDS-tile walk) -- the
automatic choice runs the filter on this set, whose LDS holds a bitmap, not rows -- and next to the row tiles stands the form
that is not a row tile at all: the two-byte window table (158 KiB: trie depth <= 2 as 65,536 16-bit cells + child lists, k_walk_win)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
if which == "cfg1":
    ds = synth.bench_batch(8192); hay_len = 8192; nh = 1 << 18
else:
    ds = synth.g6(n_hay=4096, hay_len=4096); hay_len = 4096; nh = 1 << 15
pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // ds.n_haystacks).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hay_len
for lds in (4096, 16384, 32768, 65536, 98304, 131072, 163328):
    res = api.ahocorasick_init(pats)
    res.set_option("lds_bytes", lds)
    api.ahocorasick_finalize(res)
    res.set_option("profile", 2)
    batch.bind_current_stream(res)
    ms = []
    for _ in range(5):
        total = batch.match_device(res, d_blob, d_off)
        ms.append(res.stat("scan_ms"))
    k = float(np.mean(ms[2:]))
    print(f"{which} lds_bytes {lds:6d}: hot states {int(res.stat('hot_states')):6d} of {int(res.stat('states'))}, kernel {k:8.3f} ms = "
          f"{d_blob.numel()/k/1e6:8.1f} GB/s, records {total}", flush=True)
    api.ahocorasick_deinit(res)

if which == "cfg3":
    # the same batch through the table walks by LDS form: row tiles of every size (a row of this set is 1 KiB: the whole LDS holds
    # 159 rows, not even trie depth 1 -- every byte step is a request to the L2 whatever the tile), then the window table
    print("cfg3, table walks (scan kernel ms per %d MiB):" % (d_blob.numel() >> 20), flush=True)
    for kern, lds, win in ((4, 4096, 0), (4, 32768, 0), (4, 79872, 0), (1, 16384, 0), (1, 65536, 0), (1, 163328, 0), (4, 163328, 1)):
        res = api.ahocorasick_init(pats)
        res.set_option("lds_bytes", lds)
        res.set_option("kernel", kern)
        api.ahocorasick_finalize(res)
        res.set_option("win_walk", win)
        res.set_option("profile", 2)
        batch.bind_current_stream(res)
        ms = []
        for _ in range(5):
            total = batch.match_device(res, d_blob, d_off)
            ms.append(res.stat("scan_ms"))
        k = float(np.mean(ms[2:]))
        form = "window table (65,536 cells + lists, 158 KiB)" if win else f"row tile of {int(res.stat('hot_states'))} rows ({lds} B)"
        print(f"  kernel {kern} {form:48s} {k:8.3f} ms = {d_blob.numel()/k/1e6:8.1f} GB/s, records {total}", flush=True)
        api.ahocorasick_deinit(res)
