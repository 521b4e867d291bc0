"""BASELINE configs[0]/[1] at their LITERAL size: 2048 x 16 B patterns, 256 haystacks x 8192 B (2 MiB),
init + build + 256 matches + deinit as examples/benchmark.php times it (:55-72), vs one batch call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from php_aho_corasick_amd import api, synth
ds = synth.g2()
pats = [{"id": k, "value": ds.pattern(k), "aux": [1]} for k in range(ds.n_patterns)]
hays = [ds.haystack(h) for h in range(ds.n_haystacks)]
c = api.ahocorasick_init(pats); api.ahocorasick_match(hays[0], c); api.ahocorasick_deinit(c)   # warm the process
for _ in range(3):
    t0 = time.perf_counter()
    c = api.ahocorasick_init(pats)
    t1 = time.perf_counter()
    api.ahocorasick_finalize(c)
    t2 = time.perf_counter()
    n = sum(len(api.ahocorasick_match(h, c)) for h in hays)
    t3 = time.perf_counter()
    res = api.ahocorasick_match_batch(hays, c)
    t4 = time.perf_counter()
    api.ahocorasick_deinit(c)
    t5 = time.perf_counter()
    print(f"init {1e3*(t1-t0):.1f} ms, finalize {1e3*(t2-t1):.1f} ms, 256 x ahocorasick_match {1e3*(t3-t2):.1f} ms ({n} records), "
          f"one ahocorasick_match_batch {1e3*(t4-t3):.2f} ms ({sum(len(r) for r in res)} records), deinit {1e3*(t5-t4):.1f} ms; "
          f"benchmark.php-equivalent sample = {1e3*(t3-t0 + t5-t4):.1f} ms")
