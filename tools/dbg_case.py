import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from gpu_util import host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth
name = sys.argv[1] if len(sys.argv) > 1 else "R6"
ds = synth.NAMED[name]()
o = Oracle("port"); o.add_patterns((ds.pat_blob, ds.pat_off))
for opts in (dict(kernel=2, chunk_bytes=1024), dict(kernel=2, chunk_bytes=1024, verify=0), dict(kernel=2, chunk_bytes=1024, verify=0, early_stop=0)):
    res = resource_for(ds, **opts)
    for fa in (True, False):
        p, r = host_match(res, ds, fa)
        pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, fa)
        ok = np.array_equal(p, pe) and np.array_equal(r, re_)
        print(opts, fa, "ok" if ok else "MISMATCH", len(r), len(re_), "Lmax", res.stat("max_pattern_length"), "min", res.stat("min_pattern_length"), "q", res.stat("anchor_q"))
        if not ok:
            sa = set(map(tuple, r.tolist())); sb = set(map(tuple, re_.tolist()))
            print(" extra", sorted(sa - sb)[:5], " missing", sorted(sb - sa)[:5])
            h = sorted(sb - sa)[0][0]
            print("   haystack", h, "offset", int(ds.hay_off[h]), "len", int(ds.hay_off[h+1]-ds.hay_off[h]), "records there:", [tuple(x) for x in re_.tolist() if x[0]==h and abs(x[1]-sorted(sb-sa)[0][1]) < 100])
            for rec in sorted(sb - sa)[:3]:
                print("   missing pattern", rec[3], ds.pattern(rec[3])[:70], "len", len(ds.pattern(rec[3])))
    api.ahocorasick_deinit(res)
