"""Which haystacks of a ragged config-3 batch make k_walk_runs fall back to a walk (fallback_bytes > 0)?
    python tools/dbg_fallback.py [hay_len]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
HL = int(sys.argv[1]) if len(sys.argv) > 1 else 4001
ds = synth.g6(n_hay=4096, hay_len=HL)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
res.set_option("small_path", 0)
d_blob = torch.from_numpy(ds.hay_blob).cuda()
off = np.arange(4097, dtype=np.int64) * HL
batch.bind_current_stream(res)
total = batch.match_device(res, d_blob, torch.from_numpy(off).cuda())
print("all:", total, "fallback", res.stat("fallback_bytes"), "walked", res.stat("walked_bytes"))
bad = []
for h in range(4096):
    o = torch.tensor([off[h], off[h + 1]], dtype=torch.int64, device="cuda")
    batch.match_device(res, d_blob, o)
    if res.stat("fallback_bytes") > 0:
        bad.append((h, int(off[h]) & 15, int(off[h + 1]) & 15, res.stat("fallback_bytes"), (37 * h + 11) % (HL - 64)))
print(len(bad), "haystacks fall back; (h, start & 15, end & 15, bytes, plant at):", bad[:40])
