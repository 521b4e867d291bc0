"""One ahocorasick_match call at a time (k_match_small): phase stamps (ACMI_SMALL_TRACE=1) and the call's wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from php_aho_corasick_amd import api, synth
ds = synth.g2()
res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
one = ds.haystack(3)
for _ in range(5):
    api.ahocorasick_match(one, res)
os.environ.pop("ACMI_SMALL_TRACE", None)
t0 = time.perf_counter()
for _ in range(300):
    api.ahocorasick_match(one, res)
print(f"single 8 KiB match: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")

import ctypes as C
import numpy as np
from php_aho_corasick_amd import _native as N
buf = np.frombuffer(one, dtype=np.uint8)
recs, n = C.POINTER(N.Record)(), C.c_size_t(0)
args = (res._h, buf.ctypes.data, len(one), 1, C.byref(recs), C.byref(n))
for _ in range(5):
    N.lib.acmi_match(*args)
t0 = time.perf_counter()
for _ in range(300):
    N.lib.acmi_match(*args)
print(f"the same through acmi_match alone (ctypes, no result arrays built): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
