"""First records where k_walk_chain and the oracle differ on one of tests/test_gpu_chain.py's ragged sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import device_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

hmax, lmax = int(sys.argv[1]), int(sys.argv[2])
ds = synth._planted_ragged(f"CR{hmax}", 900 + hmax, 400, 1, lmax, 300, 0, hmax, b"abc", 4)
o = Oracle("port"); o.add_patterns((ds.pat_blob, ds.pat_off))
pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
for geometry in ({}, dict(tile_w=16, tile_p=9), dict(tile_w=64, tile_p=64), dict(tile_w=32, tile_p=1), dict(tile_w=32, tile_p=12), dict(tile_w=32, tile_p=16)):
    res = resource_for(ds, kernel=6, small_path=0, lds_bytes=2048, **geometry)
    for call in range(2):
        p, r = device_match(res, ds)
        bad = np.nonzero((r != re_).any(axis=1))[0] if len(r) == len(re_) else np.arange(1)
        print(geometry, "call", call, "chain", res.stat("chain_walk_used"), "chunk", res.stat("chunk_bytes"), "records", len(r), len(re_), "differing", bad.size)
        if bad.size and call == 1:
            # runs of differing records: (first index, length, haystack, first wanted position, haystack length)
            runs, start = [], bad[0]
            for a, b in zip(bad[:-1], bad[1:]):
                if b != a + 1:
                    runs.append((start, a - start + 1)); start = b
            runs.append((start, bad[-1] - start + 1))
            for s, n in runs[:12]:
                h = int(re_[s][0])
                print("   run at", s, "len", n, "hay", h, "hay_len", int(ds.hay_off[h + 1] - ds.hay_off[h]), "hay_off%16", int(ds.hay_off[h]) % 16, "want pos", int(re_[s][1]), "got pos", int(r[s][1]), "got hay", int(r[s][0]))
    api.ahocorasick_deinit(res)
