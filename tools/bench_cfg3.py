"""Config-3 shape (65,536 patterns of 8..64 B over Sigma=256) on one GPU: build time, table
size, scan rate on 4 KiB haystacks.  Analysis tool, not the headline bench."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth, _native as N

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
opts = dict(a.split("=") for a in sys.argv[2:])
HL = int(opts.pop("hay_len", 4096))   # hay_len=4001: every haystack starts unaligned
ds = synth.g6(n_hay=4096, hay_len=HL)
t0 = time.time()
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
t1 = time.time()
BUILD_OPTS = ("lds_bytes", "max_table_bytes", "max_flat_outputs", "compact", "dense_bytes", "anchor_choice", "verify")
for k, v in opts.items():
    if k in BUILD_OPTS:
        res.set_option(k, int(v))
api.ahocorasick_finalize(res)
t2 = time.time()
res.set_option("profile", 2)
for k, v in opts.items():
    if k not in BUILD_OPTS:
        res.set_option(k, int(v))
print(f"add {t1-t0:.2f}s finalize {t2-t1:.2f}s states {int(res.stat('states'))} table {res.stat('table_bytes')/1e9:.2f} GB "
      f"hot {int(res.stat('hot_states'))} cols {int(res.stat('columns'))} mode {int(res.stat('class_mode'))}")
tile = torch.from_numpy(ds.hay_blob).cuda()
reps = max(1, nh // 4096)
d_blob = tile.repeat(reps).contiguous()
d_off = torch.arange(reps * 4096 + 1, dtype=torch.int64, device="cuda") * HL
batch.bind_current_stream(res)
t0 = time.time()
total = batch.match_device(res, d_blob, d_off)
torch.cuda.synchronize()
print(f"first call (upload {res.stat('table_bytes')/1e9:.2f} GB + scan): {time.time()-t0:.2f}s, records {total}")
for _ in range(3):
    t0 = time.time()
    total = batch.match_device(res, d_blob, d_off)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"scan {d_blob.numel()/dt/1e9:.2f} GB/s whole-job ({dt*1e3:.2f} ms), kernel {res.stat('scan_ms'):.2f} ms = {d_blob.numel()/res.stat('scan_ms')/1e6:.2f} GB/s, "
          f"walk {res.stat('walk_ms'):.3f} ms, fallback {int(res.stat('fallback_bytes'))} B of {int(res.stat('walked_bytes'))} flagged, records {total}, compact {int(res.stat('compact'))} dense rows {int(res.stat('dense_rows'))}")
