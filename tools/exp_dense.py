"""Dense anchor form on the keyword set of bench.py: scan / decide+emit kernel times on the real text and on text that never
hits (upper case): what the hashing alone costs and what the hit handling adds.  python tools/exp_dense.py [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth

opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
only = opts.pop("text", "")   # text=kw: the keyword text only (counter passes)
alpha = bytes(range(ord("a"), ord("z") + 1)) + b"     "
ds = synth.generate("words", 33, 2048, 4, 12, 8192, 8192, alpha)
nh = 1 << 17
for label, blob in [x for x in (("keyword text", ds.hay_blob), ("upper-case text (no hit window)", np.where(ds.hay_blob >= 97, ds.hay_blob - 32, ds.hay_blob).astype(np.uint8))) if not only or x[0].startswith("keyword")]:
    res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(ds.patterns())])
    for k, v in opts.items():
        res.set_option(k, int(v))
    res.set_option("profile", 2)
    d_blob = torch.from_numpy(blob).cuda().repeat(nh // 8192).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * 8192
    batch.bind_current_stream(res)
    for _ in range(3):
        total = batch.match_device(res, d_blob, d_off)
    scan, walk = res.stat("scan_ms"), res.stat("walk_ms")
    res.set_option("profile", 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{label:34s}: call {dt*1e3:.3f} ms, scan {scan:.3f} ms, walk {walk:.3f} ms, records {total}, kernel {int(res.stat('kernel_used'))}, variant {int(res.stat('anchor_variant'))}, "
          f"fallback {int(res.stat('fallback_bytes'))}", flush=True)
    api.ahocorasick_deinit(res)
    del d_blob
    torch.cuda.empty_cache()
