"""Experiment: is the per-process slow/fast mode an HBM channel effect of the 8 KiB power-of-two
stride between concurrently read streams?  Same haystacks, packed (stride 8192) vs padded strides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth
ds = synth.bench_batch(8192)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
res.set_option("profile", 2)
nh = 1 << 19
tile = torch.from_numpy(ds.hay_blob).cuda().view(8192, 8192)
batch.bind_current_stream(res)
for stride in (8192, 8192 + 128, 8192 + 256, 8192 + 1024, 8192):
    buf = torch.zeros((nh, stride), dtype=torch.uint8, device="cuda")
    buf[:, :8192] = tile.repeat(nh // 8192, 1)
    # offsets array with gaps is not expressible (offsets are contiguous by contract): emulate by making
    # each haystack `stride` long -- the pad bytes are zero (outside the alphabet) so they cost one root step each
    d_blob = buf.view(-1)
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * stride
    res.set_option("chunk_bytes", stride)
    ms = []
    for _ in range(8):
        total = batch.match_device(res, d_blob, d_off)
        ms.append(res.stat("scan_ms"))
    print(f"stride {stride}: kernel {np.mean(ms[2:]):.4f} ms for {d_blob.numel()/2**30:.3f} GiB -> {d_blob.numel()/np.mean(ms[2:])/1e6:.1f} GB/s, records {total}", flush=True)
    del buf, d_blob
