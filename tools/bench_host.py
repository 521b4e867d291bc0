"""PCIe-inclusive rate of the host-buffer entry (acmi_match_batch: what the Zend shim calls) and the
latency of single-haystack acmi_match calls.  Analysis tool; the contract bench is bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from php_aho_corasick_amd import api, synth
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
ds = synth.bench_batch(8192)
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
blob = np.tile(ds.hay_blob, nh // 8192)
off = np.arange(nh + 1, dtype=np.uint64) * 8192
api.match_arrays(res, blob, off)  # warm-up: device init, table upload, workspace
for _ in range(3):
    t0 = time.perf_counter()
    prefix, rec = api.match_arrays(res, blob, off)
    dt = time.perf_counter() - t0
    print(f"host-buffer batch: {blob.nbytes/2**20:.0f} MiB in {dt*1e3:.1f} ms = {blob.nbytes/dt/1e9:.2f} GB/s (H2D + scan + D2H of {len(rec)} records)")
one = ds.haystack(3)
api.ahocorasick_match(one, res)
t0 = time.perf_counter()
n = 200
for _ in range(n):
    r = api.ahocorasick_match(one, res)
dt = (time.perf_counter() - t0) / n
print(f"single ahocorasick_match of 8 KiB: {dt*1e6:.0f} us per call ({len(one)/dt/1e6:.1f} MB/s), {len(r)} records")
api.ahocorasick_deinit(res)
