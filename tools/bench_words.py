"""TABLE-mode check: 2048 patterns of 4..12 lowercase letters over letter+space text (the keyword
use-case of the reference's README), 2^17 x 8 KiB haystacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
alpha = bytes(range(ord("a"), ord("z") + 1)) + b"     "
ds = synth.generate("words", 33, 2048, 4, 12, 8192, 8192, alpha)
pats = [p for p in ds.patterns()]
res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(pats)])
api.ahocorasick_finalize(res)
res.set_option("profile", 2)
print({k: res.stat(k) for k in ("states", "hot_states", "columns", "class_mode", "entry_bytes", "lds_image_bytes", "max_pattern_length")})
nh = 1 << 17
d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(nh // 8192).contiguous()
d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
batch.bind_current_stream(res)
light = len(sys.argv) > 1 and sys.argv[1] == "light"   # counter passes: the exact walk only
for slab, refill, kern, wilp in (((16, 0, 4, 4),) if light else ((16, 0, 1, 8), (16, 0, 4, 1), (16, 0, 4, 4), (16, 0, 4, 8), (8, 0, 3, 4))):
    res.set_option("slab_events", slab); res.set_option("refill", refill); res.set_option("kernel", kern); res.set_option("walk_group", wilp)
    ms, tot = [], []
    for _ in range(3 if light else 6):
        total = batch.match_device(res, d_blob, d_off)
        ms.append(res.stat("scan_ms")); tot.append(res.stat("total_ms"))
    print(f"kernel {kern} walk_group {wilp} slab {slab} refill {refill}: scan kernel {np.mean(ms[2:]):.3f} ms, whole call {np.mean(tot[2:]):.3f} ms = {d_blob.numel()/np.mean(tot[2:])/1e6:.0f} GB/s, records {total} ({total/nh:.1f}/haystack), overflow items {int(res.stat('overflow_items'))} of {int(res.stat('items'))}, walked bytes {int(res.stat('walked_bytes'))}", flush=True)
